#!/bin/bash
# round 6, call 11: wide_pass_kernel with the next tile's observations / weight / head inputs one tile ahead (WIDE_PREFETCH):
# parity of the wide family, loss / gradient / product passes timed against the library before (build/exp/lib_before_wpf.so)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 1500 python -m pytest tests/test_gpu_wide_nets.py tests/test_gpu_csplit.py tests/test_gpu_adaptive_std.py tests/test_gpu_policy_options.py -m gpu -q -x 2>&1 | tail -4 > $O/r06_c11_pytest.log
cat $O/r06_c11_pytest.log
timeout 900 python tools/exp/policy_time.py --configs "13,2,100-50-25,2048000;13,2,128-128,2048000;20,6,128-64,512000" 2>&1 | grep -v "^\[build\]\|amdgpu.ids" > $O/r06_c11_policy_time.txt
cat $O/r06_c11_policy_time.txt
