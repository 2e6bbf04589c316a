#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05f
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
tail -12 $O/pytest.log
for w in swimmer4096_trpo cheetah1024_trpo_gae cartpole4096_vpg; do
python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$w.json
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["phase_ms"].items()}, d.get("roofline_mfma",{}).get("frac"), d.get("roofline_mfma",{}).get("avg_launch_ms"), d["roofline"]["kernel"][:60], d["roofline"]["wavefronts"])
    except Exception as e:
        print(f, "unreadable", e)
PY
