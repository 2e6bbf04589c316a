#!/bin/bash
# A/B: rl_cg_step reads the product with plain loads (it is an earlier launch's output) instead of agent-scope atomic loads
# (production library) against the library before (build/exp/lib_agent_loads_in_cg_step.so); CG parity tests; kernel stats
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05p
mkdir -p $O
python -m pytest tests/test_gpu_update_parity.py tests/test_gpu_policy_options.py tests/test_gpu_wide_nets.py tests/test_gpu_two_rank.py tests/test_gpu_rccl.py -q -x > $O/pytest.log 2>&1
tail -3 $O/pytest.log
stats() {
  P=/tmp/prof_$1; rm -rf $P; mkdir -p $P
  rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -- python bench.py --workload $2 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  python profiles/summarize.py stats $P/stats $O/$1_kernel_stats.csv
  grep "cg_step" $O/$1_kernel_stats.csv | cut -c1-160
}
python tools/exp/lib_ab.py \
  "python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c \"import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('C3', round(d['ms_per_step'],3), d['phase_ms'])\"" \
  "python bench.py --workload cheetah1024_trpo_gae --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c \"import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('C5', round(d['ms_per_step'],3), d['phase_ms'])\"" \
  > $O/ab.log 2>&1
grep -v "^==" $O/ab.log
echo "-- new library"
stats new_c5 cheetah1024_trpo_gae
stats new_c3 swimmer4096_trpo
cp build/exp/lib_agent_loads_in_cg_step.so /tmp/old.so; cp rllab_amd/librllab_amd.so /tmp/new.so
cp /tmp/old.so rllab_amd/librllab_amd.so; touch rllab_amd/librllab_amd.so
echo "-- old library"
stats old_c5 cheetah1024_trpo_gae
cp /tmp/new.so rllab_amd/librllab_amd.so
