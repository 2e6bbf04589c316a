#!/bin/bash
# round 5, first GPU call: the dot2 residual (exactness + issue rate), the split product's instruction-diet variants,
# the whole GPU suite, the headline with the line search decided on the device / on the host, C5's shard
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05a
mkdir -p $O
tools/ubench/dot2_residual > $O/dot2_residual.log 2>&1
python tools/exp/fvp_split_ab.py > $O/split_ab.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
tail -5 $O/pytest.log
for i in 1 2; do
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_devls_$i.json
RLLAB_DEVICE_LINE_SEARCH=0 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_hostls_$i.json
done
python bench.py --workload cheetah1024_trpo_gae --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c5_devls.json
RLLAB_DEVICE_LINE_SEARCH=0 python bench.py --workload cheetah1024_trpo_gae --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c5_hostls.json
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["phase_ms"].items()}, d.get("roofline_mfma",{}).get("frac"), d.get("roofline_mfma",{}).get("avg_launch_ms"), d.get("backtracks"))
    except Exception as e:
        print(f, "unreadable", e)
PY
cat $O/dot2_residual.log $O/split_ab.log
