#!/bin/bash
# round 6, call 18: after the bias-slot / input-transposition scales: the split tests, the whole GPU suite, accuracy / time check
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 1500 python -m pytest tests/test_gpu_fvp_split.py -m gpu -q 2>&1 | tail -15 > $O/r06_c18_pytest_split.log
cat $O/r06_c18_pytest_split.log
timeout 2400 python -m pytest tests -m gpu -q --deselect tests/test_gpu_fvp_split.py 2>&1 | tail -15 > $O/r06_c18_pytest_rest.log
cat $O/r06_c18_pytest_rest.log
timeout 900 python tools/exp/fvp_splith_check.py 2>&1 | grep -v "^\[build\]\|amdgpu.ids" > $O/r06_c18_splith_check.txt
python - <<PY
import json
for l in open("gpurun_out/r06_c18_splith_check.txt"):
    try: d=json.loads(l)
    except Exception: print(l.strip()); continue
    print(d["shape"], d["B"], d["obs_scale"], d["vec_scale"], {k:(d[k]["variant"], "%.2e"%d[k]["max_err"], d[k]["ms"]) for k in ("f16x2","bf16x3","f32")})
PY
