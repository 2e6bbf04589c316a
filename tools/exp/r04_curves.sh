#!/bin/bash
# learning curves on the round-4 kernels (GPU box): the one-env-per-wavefront rollout (HalfCheetah, Walker2D at 1024
# envs), the four-wavefront wide rollout + cooperative split products (Swimmer, (128, 128) and (100, 50, 25)),
# adaptive_std with a wide mean network; gpurun_out/curves/r04_*.csv
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/curves
run() { name=$1; shift; t0=$(date +%s.%N); timeout 600 python examples/run_trpo.py "$@" --quiet --csv gpurun_out/curves/r04_$name.csv 2>&1 | tail -1; t1=$(date +%s.%N)
  python - <<PY
import csv
r=list(csv.DictReader(open("gpurun_out/curves/r04_$name.csv")))
print("$name", "iters", len(r), "wall %.1f s" % ($t1 - $t0), "AverageReturn first / mean of last 5: %.3f / %.3f" % (float(r[0]["AverageReturn"]), sum(float(x["AverageReturn"]) for x in r[-5:])/5), "max MeanKL %.5f" % max(float(x["MeanKL"]) for x in r))
PY
}
run half_cheetah_64_64 --env half_cheetah --n-envs 1024 --n-itr 150 --hidden 64 --gae-lambda 0.97
run walker2d_64_64 --env walker2d --n-envs 1024 --n-itr 100 --hidden 64 --gae-lambda 0.97
run swimmer_128_128 --env swimmer --n-envs 4096 --n-itr 60 --hidden 128,128
run swimmer_100_50_25 --env swimmer --n-envs 4096 --n-itr 60 --hidden 100,50,25
run swimmer_100_50_25_adaptive_std --env swimmer --n-envs 1024 --n-itr 40 --hidden 100,50,25 --adaptive-std
