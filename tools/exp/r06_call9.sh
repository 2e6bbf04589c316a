#!/bin/bash
# round 6, call 9: csplit_fvp_kernel timing A/B over the libraries in build/exp
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
CS_FORCE=2 timeout 900 python tools/exp/with_libs.py tools/exp/csplit_time.py "13,2,128-128,2048000;13,2,100-50-25,2048000;13,2,128-64,2048000;13,2,128-128-64,2048000;20,6,128-64-32,512000" only 2>&1 | grep -v "^\[build\]\|amdgpu.ids" > $O/r06_c9_csplit_ab.txt
cat $O/r06_c9_csplit_ab.txt
