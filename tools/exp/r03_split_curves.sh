#!/bin/bash
# TRPO with the split-operand Fisher-vector product against the same run on the f32 matrix instructions (RLLAB_FVP_SPLIT=0),
# same seeds, for every env whose policy takes the split kernel (GPU box): gpurun_out/curves/r03_split_<env>{,_f32}.csv
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/curves
for cfg in "swimmer 4096 100" "cartpole 1024 40" "double_pendulum 1024 60" "inverted_double_pendulum 1024 60" "cartpole_swingup 1024 80"; do
  set -- $cfg
  timeout 400 python examples/run_trpo.py --env $1 --n-envs $2 --n-itr $3 --hidden 32 --quiet --csv gpurun_out/curves/r03_split_$1.csv > /dev/null 2>&1
  RLLAB_FVP_SPLIT=0 timeout 400 python examples/run_trpo.py --env $1 --n-envs $2 --n-itr $3 --hidden 32 --quiet --csv gpurun_out/curves/r03_split_$1_f32.csv > /dev/null 2>&1
  python - <<PY
import csv
a=list(csv.DictReader(open("gpurun_out/curves/r03_split_$1.csv"))); b=list(csv.DictReader(open("gpurun_out/curves/r03_split_$1_f32.csv")))
ra=[float(x["AverageReturn"]) for x in a]; rb=[float(x["AverageReturn"]) for x in b]
d10=max(abs(x-y)/max(1.0,abs(y)) for x,y in zip(ra[:10],rb[:10]))
print("%-26s envs %5d iters %3d  AverageReturn first / mean of last 5:  split %9.3f / %9.3f   f32 %9.3f / %9.3f   max rel. difference over the first 10 iterations %.1e   max MeanKL %.4f / %.4f" % (
    "$1", $2, len(ra), ra[0], sum(ra[-5:])/5, rb[0], sum(rb[-5:])/5, d10, max(float(x["MeanKL"]) for x in a), max(float(x["MeanKL"]) for x in b)))
PY
done
