#!/bin/bash
# A/B: the loss pass (policy_pass_kernel<.., MODE_LOSS>) at four wavefronts per SIMD (production library) against the library
# before (build/exp/lib_loss_pass_at_its_old_occupancy.so: two for the 32-unit nets of <= 13 inputs, one otherwise)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05q
mkdir -p $O
python -m pytest tests/test_gpu_update_parity.py tests/test_gpu_policy_options.py tests/test_gpu_regressor.py tests/test_gpu_reference_pins.py tests/test_gpu_shard_rehearsal.py tests/test_gpu_two_rank.py -q -x > $O/pytest.log 2>&1
tail -3 $O/pytest.log
stats() {
  P=/tmp/prof_$1; rm -rf $P; mkdir -p $P
  rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -- python bench.py --workload $2 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  python profiles/summarize.py stats $P/stats $O/$1_kernel_stats.csv
  grep "policy_pass_kernel" $O/$1_kernel_stats.csv | grep "0, false" | cut -c1-170
}
python tools/exp/lib_ab.py \
  "python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c \"import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('C3', round(d['ms_per_step'],3), d['phase_ms'])\"" \
  "python bench.py --workload cheetah1024_trpo_gae --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c \"import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('C5', round(d['ms_per_step'],3), d['phase_ms'])\"" \
  "python bench.py --workload cartpole4096_vpg --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c \"import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('C2', round(d['ms_per_step'],3), d['phase_ms'])\"" \
  > $O/ab.log 2>&1
grep -v "^==" $O/ab.log
echo "-- new library: loss-pass launches (calls, total, avg, min, max ns)"
stats new_c3 swimmer4096_trpo
stats new_c5 cheetah1024_trpo_gae
cp rllab_amd/librllab_amd.so /tmp/new.so
cp build/exp/lib_loss_pass_at_its_old_occupancy.so rllab_amd/librllab_amd.so; touch rllab_amd/librllab_amd.so
echo "-- old library"
stats old_c3 swimmer4096_trpo
stats old_c5 cheetah1024_trpo_gae
cp /tmp/new.so rllab_amd/librllab_amd.so
