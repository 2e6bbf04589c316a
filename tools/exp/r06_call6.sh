#!/bin/bash
# round 6, call 6: fvp_split_kernel with the observations + weight as XPIECES 1 KB LDS-direct pieces (was 8 KB0 + 1 pieces
# of 256 bytes): parity, then A/B against the library before the change (build/exp/lib_before_xpieces.so), then the headline
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_fvp_split.py tests/test_gpu_update_parity.py -m gpu -q -x 2>&1 | tail -4 > $O/r06_c6_pytest.log
cat $O/r06_c6_pytest.log
timeout 900 python tools/exp/fvp_split_ab.py 2>&1 | grep -v "^\[build\]\|amdgpu.ids" > $O/r06_c6_split_ab.txt
cat $O/r06_c6_split_ab.txt
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r06_c6_bench_$i.json
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r06_c6_bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["phase_ms"].items()}, d.get("roofline_mfma",{}).get("frac"))
PY
