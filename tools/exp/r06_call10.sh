#!/bin/bash
# round 6, call 10: the cooperative product with the late fetch as the default: parity of the wide family + the update tests,
# product timing against the library of the round's start (build/exp/lib_head.so), and the two wide bench lines
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 1500 python -m pytest tests/test_gpu_csplit.py tests/test_gpu_wide_nets.py tests/test_gpu_update_parity.py tests/test_gpu_policy_options.py -m gpu -q -x 2>&1 | tail -6 > $O/r06_c10_pytest.log
cat $O/r06_c10_pytest.log
CS_FORCE=2 timeout 900 python tools/exp/with_libs.py tools/exp/csplit_time.py "13,2,128-128,2048000;13,2,100-50-25,2048000;13,2,128-64,2048000;13,2,128-128-64,2048000;20,6,128-64-32,512000" only 2>&1 | grep -v "^\[build\]\|amdgpu.ids" > $O/r06_c10_csplit_ab.txt
cat $O/r06_c10_csplit_ab.txt
for h in 100,50,25 128,128; do
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --hidden $h 2>/dev/null | tail -1 > $O/r06_c10_bench_hidden_${h//,/_}.json
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r06_c10_bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["phase_ms"].items()}, d.get("roofline_mfma",{}).get("frac"), d.get("roofline_mfma",{}).get("avg_launch_ms"))
PY
