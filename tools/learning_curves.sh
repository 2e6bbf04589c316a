#!/bin/bash
# TRPO learning curves of every HIP-native env (run on the GPU box): gpurun_out/curves/trpo_<env>.csv
mkdir -p gpurun_out/curves
for cfg in "swimmer 4096 150 32 1.0" "half_cheetah 1024 150 64 0.97" "walker2d 1024 100 64 0.97" "hopper 1024 100 32 0.97" "inverted_double_pendulum 1024 60 32 1.0" "cartpole 1024 40 32 1.0" "double_pendulum 1024 60 32 1.0" "cartpole_swingup 1024 80 32 1.0"; do
  set -- $cfg
  t0=$(date +%s.%N)
  timeout 400 python examples/run_trpo.py --env $1 --n-envs $2 --n-itr $3 --hidden $4 --gae-lambda $5 --quiet --csv gpurun_out/curves/trpo_$1.csv 2>&1 | tail -1
  t1=$(date +%s.%N)
  python - <<PY
import csv
r=list(csv.DictReader(open("gpurun_out/curves/trpo_$1.csv")))
print("$1", "envs $2 iters", len(r), "wall %.1f s" % ($t1 - $t0), "AverageReturn first / mean of last 5: %.3f / %.3f" % (float(r[0]["AverageReturn"]), sum(float(x["AverageReturn"]) for x in r[-5:])/5))
PY
done
