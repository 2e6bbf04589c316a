#!/bin/bash
# round 6, call 7: csplit_fvp_kernel with k-slices for the narrow layers (CS_KSPLIT): parity of every cooperative shape,
# then the product timed per library: build/exp/lib_head.so (before), lib_base.so (new source, CS_KSPLIT=0), lib_ksplit.so
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 1200 python -m pytest tests/test_gpu_csplit.py tests/test_gpu_wide_nets.py -m gpu -q -x 2>&1 | tail -6 > $O/r06_c7_pytest.log
cat $O/r06_c7_pytest.log
CS_FORCE=2 timeout 900 python tools/exp/with_libs.py tools/exp/csplit_time.py "13,2,128-128,2048000;13,2,100-50-25,2048000;13,2,128-64,2048000;13,2,128-128-64,2048000;13,2,64-32,2048000;20,6,128-64-32,512000" only 2>&1 | grep -v "^\[build\]\|amdgpu.ids" > $O/r06_c7_csplit_ab.txt
cat $O/r06_c7_csplit_ab.txt
