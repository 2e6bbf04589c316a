#!/bin/bash
# soak of the one-body-per-lane two-leg program: 1000 TRPO iterations of HalfCheetah and Walker2D (1024 envs, (64, 64), GAE)
# on the one-env-per-wavefront rollout, and 300 at 4096 envs on the 16-envs-per-wavefront shape; every logged number finite,
# MeanKL <= step size
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/curves/long
for cfg in "half_cheetah 1024 1000" "walker2d 1024 1000" "half_cheetah 4096 300"; do
  set -- $cfg
  t0=$(date +%s.%N)
  timeout 600 python examples/run_trpo.py --env $1 --n-envs $2 --n-itr $3 --hidden 64 --gae-lambda 0.97 --quiet --csv gpurun_out/curves/long/r05_body_lanes_$1_$2.csv 2>&1 | grep -v amdgpu.ids | tail -1
  t1=$(date +%s.%N)
  python - <<PY
import csv, math
r=list(csv.DictReader(open("gpurun_out/curves/long/r05_body_lanes_$1_$2.csv")))
bad=[(i,k) for i,x in enumerate(r) for k,v in x.items() if v not in ("", None) and not math.isfinite(float(v))]
print("$1 envs $2 iters", len(r), "wall %.1f s" % ($t1 - $t0), "non-finite entries:", len(bad), bad[:3], "AverageReturn first / best / mean of last 10: %.1f / %.1f / %.1f" % (float(r[0]["AverageReturn"]), max(float(x["AverageReturn"]) for x in r), sum(float(x["AverageReturn"]) for x in r[-10:])/10), "max MeanKL %.5f" % max(float(x["MeanKL"]) for x in r))
PY
done
