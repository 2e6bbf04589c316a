#!/bin/bash
# end-of-round soak of the headline env on the final library: 2000 TRPO iterations of Swimmer (4096 envs), every logged value
# finite, MeanKL <= step size; the two-rank code path of bench.py on one device (ranks share the GPU: not a scaling figure)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/curves/long
t0=$(date +%s.%N)
timeout 600 python examples/run_trpo.py --env swimmer --n-envs 4096 --n-itr 2000 --quiet --csv gpurun_out/curves/long/r05_swimmer_4096_2000.csv 2>&1 | grep -v amdgpu.ids | tail -1
t1=$(date +%s.%N)
python - <<PY
import csv, math
r=list(csv.DictReader(open("gpurun_out/curves/long/r05_swimmer_4096_2000.csv")))
bad=[(i,k) for i,x in enumerate(r) for k,v in x.items() if v not in ("", None) and not math.isfinite(float(v))]
print("swimmer 4096 envs iters", len(r), "wall %.1f s" % ($t1 - $t0), "non-finite:", len(bad), "AverageReturn first / itr 100 / mean of last 10: %.2f / %.2f / %.2f" % (float(r[0]["AverageReturn"]), float(r[100]["AverageReturn"]), sum(float(x["AverageReturn"]) for x in r[-10:])/10), "max MeanKL %.5f" % max(float(x["MeanKL"]) for x in r))
PY
env RLLAB_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('two ranks on one device:', d['n_gpus'], 'ranks', round(d['ms_per_step'],2), 'ms', d.get('collectives_per_iter'), 'collectives/iter', d.get('update_sum_path'), d.get('phase_ms_per_rank'))"
