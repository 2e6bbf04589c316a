#!/bin/bash
# A/B: the distribution head's per-sample inputs (advantage, action, old mean) fetched one tile ahead in policy_pass_kernel
# (production library) against the library before the change (build/exp/lib_before_head_prefetch.so); update parity tests
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05o
mkdir -p $O
python -m pytest tests/test_gpu_update_parity.py tests/test_gpu_policy_options.py tests/test_gpu_regressor.py tests/test_gpu_reference_pins.py -q -x > $O/pytest.log 2>&1
tail -3 $O/pytest.log
python tools/exp/lib_ab.py \
  "python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c \"import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('C3', round(d['ms_per_step'],3), d['phase_ms'])\"" \
  "python bench.py --workload cheetah1024_trpo_gae --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c \"import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('C5', round(d['ms_per_step'],3), d['phase_ms'])\"" \
  "python bench.py --workload cartpole4096_vpg --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c \"import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('C2', round(d['ms_per_step'],3), d['phase_ms'])\"" \
  > $O/ab.log 2>&1
cat $O/ab.log
P=/tmp/prof_r05o; rm -rf $P; mkdir -p $P
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python profiles/summarize.py stats $P/stats $O/kernel_stats.csv
grep "policy_pass" $O/kernel_stats.csv | cut -c1-200
