#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05l
mkdir -p $O
python -m pytest tests/test_gpu_env_options.py -x -q > $O/pytest_env_options.log 2>&1
tail -3 $O/pytest_env_options.log
python tools/exp/fallback_probe_envs.py > $O/fallback_probe_envs.log 2>&1
grep -v amdgpu.ids $O/fallback_probe_envs.log
bash tools/exp/r05_curves.sh > $O/curves.log 2>&1
tail -12 $O/curves.log
