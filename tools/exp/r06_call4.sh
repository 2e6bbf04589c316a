#!/bin/bash
# round 6, call 4: fvp_split16_kernel (16-sample tiles, four wavefronts per SIMD): parity, then time against fvp_split_kernel
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_fvp_split.py -m gpu -q -x -k "sixteen" 2>&1 | tail -15 > $O/r06_c4_pytest.log
cat $O/r06_c4_pytest.log
timeout 600 python tools/exp/fvp_split16_time.py 2>&1 | grep -v "^\[build\]" > $O/r06_c4_split16_time.txt
cat $O/r06_c4_split16_time.txt
