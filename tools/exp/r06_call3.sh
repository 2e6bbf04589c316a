#!/bin/bash
# round 6, call 3: review item 7 -- one hidden layer of 65 .. 128 units and a one-layer log-std network on the kernels
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests/test_gpu_adaptive_std.py tests/test_gpu_policy_options.py -m gpu -q -x 2>&1 | tail -15 > $O/r06_c3_pytest.log
cat $O/r06_c3_pytest.log
python tools/exp/fallback_probe.py > $O/r06_fallback_probe_policies.txt 2>&1
grep -v amdgpu.ids $O/r06_fallback_probe_policies.txt | tail -30
