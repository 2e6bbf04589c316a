#!/bin/bash
# round 6, call 16: the two-way f16 split product after the prologue / asm repairs: accuracy again, then fvp() of the
# shipped library (bf16 three-way and f16 two-way) against the ablation / variant builds of policy_splith_kernels.hip
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python tools/exp/fvp_splith_check.py 2>&1 | grep -v "^\[build\]\|amdgpu.ids" > $O/r06_c16_splith_check.txt
python - <<PY
import json
for l in open("gpurun_out/r06_c16_splith_check.txt"):
    try: d=json.loads(l)
    except Exception: print(l.strip()); continue
    print(d["shape"], d["B"], d["obs_scale"], d["vec_scale"], {k:(d[k]["variant"], "%.2e"%d[k]["max_err"], d[k]["ms"]) for k in ("f16x2","bf16x3","f32")})
PY
timeout 900 python tools/exp/fvp_split_ab.py 2>&1 | grep -v "^\[build\]\|amdgpu.ids" > $O/r06_c16_ab32.txt
cat $O/r06_c16_ab32.txt
timeout 900 python tools/exp/fvp_split_ab.py 64 2>&1 | grep -v "^\[build\]\|amdgpu.ids" > $O/r06_c16_ab64.txt
cat $O/r06_c16_ab64.txt
