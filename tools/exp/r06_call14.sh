#!/bin/bash
# round 6, call 14: what a two-part / three-term product would buy (RL_ABL_HALF, timing only): fvp() of the shipped library
# against build/exp/lib_half.so, (32, 32) at 2.048 M samples and (64, 64) at 512 k
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 600 python tools/exp/fvp_split_ab.py 2>&1 | grep -v "^\[build\]\|amdgpu.ids" > $O/r06_c14_half32.txt
cat $O/r06_c14_half32.txt
timeout 600 python tools/exp/fvp_split_ab.py 64 2>&1 | grep -v "^\[build\]\|amdgpu.ids" > $O/r06_c14_half64.txt
cat $O/r06_c14_half64.txt
