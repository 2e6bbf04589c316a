#!/bin/bash
# the 16-envs-per-wavefront shape of the two-legged envs under the one-body-per-lane program (V2 on a quad), against the
# one-env-per-wavefront shape, at 1024 / 4096 / 8192 envs
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05n
mkdir -p $O
for n in 1024 4096 8192; do
  for w in 0 1; do
    env RLLAB_TWO_LEG_WAVE_KERNEL=$w python bench.py --workload cheetah1024_trpo_gae --n-envs $n --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_n${n}_wave$w.json
  done
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["phase_ms"].items()}, d["roofline"]["kernel"][:48])
    except Exception as e: print(f, "ERR", e)
PY
