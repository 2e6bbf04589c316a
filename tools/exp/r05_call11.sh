#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05k
mkdir -p $O
python -m pytest tests/test_gpu_env_options.py tests/test_gpu_process_parity.py tests/test_gpu_update_parity.py -x -q > $O/pytest_new.log 2>&1
tail -4 $O/pytest_new.log
python tools/exp/lib_ab.py \
  "python bench.py --workload cheetah1024_trpo_gae --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c \"import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('C5', round(d['ms_per_step'],3), d['phase_ms'], d['roofline']['avg_launch_ms'])\"" \
  "python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c \"import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('C3', round(d['ms_per_step'],3), d['phase_ms'], d['roofline']['avg_launch_ms'])\"" \
  > $O/slp_ab.log 2>&1
cat $O/slp_ab.log
cp rllab_amd/librllab_amd.so /tmp/lib_keep.so
cp build/exp/lib_env_slp.so rllab_amd/librllab_amd.so; touch rllab_amd/librllab_amd.so
python -m pytest tests/test_gpu_env_parity.py tests/test_gpu_env_options.py -x -q > $O/pytest_slp_parity.log 2>&1
tail -4 $O/pytest_slp_parity.log
cp /tmp/lib_keep.so rllab_amd/librllab_amd.so
