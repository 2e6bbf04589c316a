#!/bin/bash
# round 6: learning curves / soaks of the wide and deep nets on the k-slice build of the cooperative split product (GPU box)
#   -> gpurun_out/curves/r06_wide_*.csv + r06_wide_summary.txt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/curves
S=gpurun_out/curves/r06_wide_summary.txt
: > $S
run() {   # tag, -- args
  tag=$1; shift; shift
  t0=$(date +%s.%N)
  timeout 900 python examples/run_trpo.py "$@" --quiet --csv gpurun_out/curves/r06_wide_$tag.csv 2>&1 | grep -v amdgpu.ids | tail -1
  t1=$(date +%s.%N)
  python - <<PY >> $S
import csv, math
r=list(csv.DictReader(open("gpurun_out/curves/r06_wide_$tag.csv")))
fin=all(math.isfinite(float(v)) for x in r for k,v in x.items() if k in ("AverageReturn","MeanKL","LossAfter","LossBefore"))
print("%-34s iters %3d wall %6.1f s  AverageReturn first / mean of last 5: %9.3f / %9.3f   max MeanKL %.5f  all finite: %s" % ("$tag", len(r), $t1 - $t0, float(r[0]["AverageReturn"]), sum(float(x["AverageReturn"]) for x in r[-5:])/5, max(float(x["MeanKL"]) for x in r), fin))
PY
}
run swimmer_100_50_25 -- --env swimmer --n-envs 4096 --n-itr 200 --hidden 100,50,25
run swimmer_128_64 -- --env swimmer --n-envs 4096 --n-itr 100 --hidden 128,64
run swimmer_128_128 -- --env swimmer --n-envs 4096 --n-itr 100 --hidden 128
run cheetah_100_50_25 -- --env half_cheetah --n-envs 1024 --n-itr 100 --hidden 100,50,25 --gae-lambda 0.97
run swimmer_32_ref -- --env swimmer --n-envs 4096 --n-itr 100 --hidden 32
cat $S
