#!/bin/bash
# round 6: learning under the two-way f16 split Fisher-vector product (the library's choice) and under the three-way bf16 one
# (RLLAB_FVP_SPLIT=5), same seeds: the headline env / net, C5's env / net, and a long soak of the headline (GPU box)
#   -> gpurun_out/curves/r06_splith_*.csv + r06_splith_summary.txt
mkdir -p gpurun_out/curves
S=gpurun_out/curves/r06_splith_summary.txt
: > $S
run() {   # tag, env value of RLLAB_FVP_SPLIT ("" = unset), -- args
  tag=$1; val=$2; shift; shift; shift
  t0=$(date +%s.%N)
  if [ -n "$val" ]; then export RLLAB_FVP_SPLIT=$val; else unset RLLAB_FVP_SPLIT; fi
  timeout 1200 python examples/run_trpo.py "$@" --quiet --csv gpurun_out/curves/r06_splith_$tag.csv 2>&1 | grep -v amdgpu.ids | tail -1
  unset RLLAB_FVP_SPLIT
  t1=$(date +%s.%N)
  python - <<PY >> $S
import csv, math
r=list(csv.DictReader(open("gpurun_out/curves/r06_splith_$tag.csv")))
fin=all(math.isfinite(float(v)) for x in r for v in x.values() if v not in ("", None))
print("%-34s iters %4d wall %6.1f s  AverageReturn first / mean of last 5: %9.3f / %9.3f   max MeanKL %.5f  all finite: %s" % ("$tag", len(r), $t1 - $t0, float(r[0]["AverageReturn"]), sum(float(x["AverageReturn"]) for x in r[-5:])/5, max(float(x["MeanKL"]) for x in r), fin))
PY
}
run swimmer_f16 "" -- --env swimmer --n-envs 4096 --n-itr 200
run swimmer_bf16 5 -- --env swimmer --n-envs 4096 --n-itr 200
run cheetah_f16 "" -- --env half_cheetah --n-envs 1024 --n-itr 200 --hidden 64 --gae-lambda 0.97
run cheetah_bf16 5 -- --env half_cheetah --n-envs 1024 --n-itr 200 --hidden 64 --gae-lambda 0.97
run walker_f16 "" -- --env walker2d --n-envs 1024 --n-itr 100 --hidden 64 --gae-lambda 0.97
run cartpole_f16 "" -- --env cartpole --n-envs 4096 --n-itr 50
run swimmer_f16_soak "" -- --env swimmer --n-envs 4096 --n-itr 1500
cat $S
