#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fvp_split.py tests/test_gpu_update_parity.py tests/test_gpu_process_parity.py -q -x > $O/pytest.log 2>&1
tail -6 $O/pytest.log
python tools/exp/fvp_split_ab.py > $O/split_variants.log 2>&1
grep -v amdgpu.ids $O/split_variants.log
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench.json
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["phase_ms"].items()}, d.get("roofline_mfma",{}).get("frac"), d.get("roofline_mfma",{}).get("avg_launch_ms"), d.get("update_ms_and_backtracks_per_iteration"))
PY
