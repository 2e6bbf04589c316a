#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05j
mkdir -p $O
python tools/exp/fvp_split_ab.py 64 > $O/split64_ablations.log 2>&1
grep -v amdgpu.ids $O/split64_ablations.log
python tools/exp/fallback_probe.py > $O/fallback_probe.log 2>&1
grep -v amdgpu.ids $O/fallback_probe.log
