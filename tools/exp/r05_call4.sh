#!/bin/bash
# round 5: the (64, 64) split product -- parity, then C5's shard both ways
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05d
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fvp_split.py tests/test_gpu_update_parity.py tests/test_gpu_csplit.py -q -x > $O/pytest.log 2>&1
tail -8 $O/pytest.log
python bench.py --workload cheetah1024_trpo_gae --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c5_split64.json
RLLAB_FVP_SPLIT=0 python bench.py --workload cheetah1024_trpo_gae --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c5_f32.json
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c3.json
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["phase_ms"].items()}, d.get("roofline_mfma",{}).get("frac"), d.get("roofline_mfma",{}).get("avg_launch_ms"), d.get("update_ms_and_backtracks_per_iteration"))
    except Exception as e:
        print(f, "unreadable", e)
PY
