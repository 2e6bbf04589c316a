#!/bin/bash
# round 5, second GPU call: dot2 residual with register selectors, the split product's timing ablations, the whole GPU
# suite, the headline with the line search decided on the device / on the host, a kernel timeline of both
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05b
mkdir -p $O
tools/ubench/dot2_residual > $O/dot2_residual.log 2>&1
python tools/exp/fvp_split_ab.py > $O/split_ablations.log 2>&1
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
tail -15 $O/pytest.log
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_devls.json
RLLAB_DEVICE_LINE_SEARCH=0 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_hostls.json
for mode in 3 0; do
  P=/tmp/prof_ls$mode; rm -rf $P
  RLLAB_DEVICE_LINE_SEARCH=$mode rocprofv3 --kernel-trace --stats --output-format csv -d $P -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof_ls$mode.log 2>&1
  python profiles/summarize.py timeline $P $O/timeline_ls$mode.csv
  python profiles/summarize.py stats $P $O/kernel_stats_ls$mode.csv
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["phase_ms"].items()}, d.get("roofline_mfma",{}).get("frac"), d.get("roofline_mfma",{}).get("avg_launch_ms"), d.get("update_ms_and_backtracks_per_iteration"))
    except Exception as e:
        print(f, "unreadable", e)
PY
cat $O/dot2_residual.log; grep -v amdgpu.ids $O/split_ablations.log
