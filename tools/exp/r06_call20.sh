#!/bin/bash
# round 6, call 20: the wide-head (32, 32) shapes ((20, 3), (20, 6), (21, 6): one wavefront per SIMD) on the two-way f16 split:
# parity, then fvp() of both arithmetics at (20, 6) on 512 k samples
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 1500 python -m pytest tests/test_gpu_fvp_split.py tests/test_gpu_csplit.py tests/test_gpu_update_parity.py -m gpu -q 2>&1 | tail -8 > $O/r06_c20_pytest.log
cat $O/r06_c20_pytest.log
timeout 900 python tools/exp/fvp_split_ab.py wide 2>&1 | grep -v "^\[build\]\|amdgpu.ids" > $O/r06_c20_ab_wide.txt
cat $O/r06_c20_ab_wide.txt
