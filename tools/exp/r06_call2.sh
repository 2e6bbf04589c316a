#!/bin/bash
# round 6, call 2: GAE launch shapes (A: 16 envs x 63 chunks of 8 steps, B: 16 x 32 x 16, C: 32 x 32 x 16), the baseline
# fit on a side stream (headline both ways)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
python tools/exp/with_libs.py tools/exp/scan_time.py 2>&1 | grep -v "^\[build\]" > $O/r06_c2_scan_ab.txt
cat $O/r06_c2_scan_ab.txt
python -m pytest tests/test_gpu_process_parity.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r06_c2_bench_side_$i.json
env RLLAB_LFB_INLINE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r06_c2_bench_inline_$i.json
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r06_c2_bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["phase_ms"].items()})
PY
