#!/bin/bash
# round 6: learning curves under the penalty and under MuJoCo's soft-constraint limit / contact models (GPU box)
#   -> gpurun_out/curves/r06_*.csv + r06_summary.txt
mkdir -p gpurun_out/curves
S=gpurun_out/curves/r06_summary.txt
: > $S
run() {   # tag, -- args
  tag=$1; shift; shift
  t0=$(date +%s.%N)
  timeout 900 python examples/run_trpo.py "$@" --quiet --csv gpurun_out/curves/r06_$tag.csv 2>&1 | grep -v amdgpu.ids | tail -1
  t1=$(date +%s.%N)
  python - <<PY >> $S
import csv
r=list(csv.DictReader(open("gpurun_out/curves/r06_$tag.csv")))
print("%-34s iters %3d wall %6.1f s  AverageReturn first / mean of last 5: %9.3f / %9.3f   max MeanKL %.5f" % ("$tag", len(r), $t1 - $t0, float(r[0]["AverageReturn"]), sum(float(x["AverageReturn"]) for x in r[-5:])/5, max(float(x["MeanKL"]) for x in r)))
PY
}
run cheetah_penalty -- --env half_cheetah --n-envs 1024 --n-itr 100 --hidden 64 --gae-lambda 0.97
run cheetah_mujoco -- --env half_cheetah --n-envs 1024 --n-itr 100 --hidden 64 --gae-lambda 0.97 --limit-model mujoco --contact-model mujoco
run cheetah_mujoco_limits_only -- --env half_cheetah --n-envs 1024 --n-itr 100 --hidden 64 --gae-lambda 0.97 --limit-model mujoco
run walker_penalty -- --env walker2d --n-envs 1024 --n-itr 100 --hidden 64 --gae-lambda 0.97
run walker_mujoco -- --env walker2d --n-envs 1024 --n-itr 100 --hidden 64 --gae-lambda 0.97 --limit-model mujoco --contact-model mujoco
run hopper_penalty -- --env hopper --n-envs 1024 --n-itr 100 --hidden 32 --gae-lambda 0.97
run hopper_mujoco -- --env hopper --n-envs 1024 --n-itr 100 --hidden 32 --gae-lambda 0.97 --limit-model mujoco --contact-model mujoco
cat $S
