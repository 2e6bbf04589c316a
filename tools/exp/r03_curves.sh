#!/bin/bash
# round-3 learning evidence: TRPO through the unchanged BatchPolopt.train() loop with policies that take the new
# kernel families (GPU box): gpurun_out/curves/r03_*.csv
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/curves
run() {  # name env n_envs n_itr extra...
  name=$1; shift; env=$1; shift; n=$1; shift; it=$1; shift
  t0=$(date +%s.%N)
  timeout 600 python examples/run_trpo.py --env $env --n-envs $n --n-itr $it --quiet --csv gpurun_out/curves/r03_$name.csv "$@" 2>&1 | tail -1
  t1=$(date +%s.%N)
  python - <<PY
import csv
r=list(csv.DictReader(open("gpurun_out/curves/r03_$name.csv")))
print("$name", "envs $n iters", len(r), "wall %.1f s" % ($t1 - $t0), "AverageReturn first / mean of last 5: %.3f / %.3f" % (float(r[0]["AverageReturn"]), sum(float(x["AverageReturn"]) for x in r[-5:])/5), "max MeanKL %.4f" % max(float(x["MeanKL"]) for x in r))
PY
}
run swimmer_100_50_25 swimmer 4096 100 --hidden 100,50,25
run half_cheetah_128_128 half_cheetah 1024 100 --hidden 128,128 --gae-lambda 0.97
run cartpole_adaptive_std cartpole 1024 40 --adaptive-std
