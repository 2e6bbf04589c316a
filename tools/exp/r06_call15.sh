#!/bin/bash
# round 6, call 15: the two-way f16 split product, first run: the split forms bit for bit (ubench), accuracy against float64
# and the two other arithmetics over shapes / vector scales / observation scales, timings
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
./tools/ubench/f16_split > $O/r06_f16_split.txt 2>&1; tail -4 $O/r06_f16_split.txt
timeout 900 python tools/exp/fvp_splith_check.py 2>&1 | grep -v "^\[build\]\|amdgpu.ids" > $O/r06_c15_splith_check.txt
cat $O/r06_c15_splith_check.txt
