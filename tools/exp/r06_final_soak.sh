#!/bin/bash
# end-of-round soak on the final library (f16 split products, two wavefronts per SIMD step kernel): 2000 iterations of the
# headline env / net, 1000 of C5's env / net -> gpurun_out/curves/long/r06_final_*.csv + summary
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/curves/long
S=gpurun_out/curves/long/r06_final_soak_summary.txt
: > $S
run() {
  tag=$1; shift
  t0=$(date +%s.%N)
  timeout 1500 python examples/run_trpo.py "$@" --quiet --csv gpurun_out/curves/long/r06_final_$tag.csv 2>&1 | grep -v amdgpu.ids | tail -1
  t1=$(date +%s.%N)
  python - <<PY >> $S
import csv, math
r=list(csv.DictReader(open("gpurun_out/curves/long/r06_final_$tag.csv")))
fin=all(math.isfinite(float(v)) for x in r for v in x.values() if v not in ("", None))
print("%-20s iters %4d wall %6.1f s  AverageReturn first / mean of last 5: %9.3f / %9.3f   max MeanKL %.5f  all finite: %s" % ("$tag", len(r), $t1 - $t0, float(r[0]["AverageReturn"]), sum(float(x["AverageReturn"]) for x in r[-5:])/5, max(float(x["MeanKL"]) for x in r), fin))
PY
}
run swimmer --env swimmer --n-envs 4096 --n-itr 2000
run half_cheetah --env half_cheetah --n-envs 1024 --n-itr 1000 --hidden 64 --gae-lambda 0.97
cat $S
