#!/bin/bash
# long TRPO runs on the split-operand Fisher-vector product (GPU box): gpurun_out/curves/long/r03_split_<env>.csv
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/curves/long
for cfg in "swimmer 4096 600 32 1.0" "hopper 1024 300 32 0.97" "half_cheetah 1024 300 32 0.97" "walker2d 1024 300 32 0.97"; do
  set -- $cfg
  t0=$(date +%s.%N)
  timeout 500 python examples/run_trpo.py --env $1 --n-envs $2 --n-itr $3 --hidden $4 --gae-lambda $5 --quiet --csv gpurun_out/curves/long/r03_split_$1.csv > /dev/null 2>&1
  t1=$(date +%s.%N)
  python - <<PY
import csv, math
r=list(csv.DictReader(open("gpurun_out/curves/long/r03_split_$1.csv")))
ret=[float(x["AverageReturn"]) for x in r]
print("%-14s envs %5d iters %4d wall %5.1f s  AverageReturn first / best / mean of last 10: %9.3f / %9.3f / %9.3f   all finite: %s   max MeanKL %.4f" % (
    "$1", $2, len(r), $t1 - $t0, ret[0], max(ret), sum(ret[-10:])/10, all(math.isfinite(v) for v in ret), max(float(x["MeanKL"]) for x in r)))
PY
done
