#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05i
mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_policy_options.py tests/test_gpu_wide_nets.py tests/test_gpu_env_parity.py tests/test_gpu_rccl.py tests/test_gpu_two_rank.py tests/test_gpu_peer_allreduce.py -q > $O/pytest.log 2>&1
tail -8 $O/pytest.log
for i in 1 2; do python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c3_$i.json; done
P=/tmp/prof_i; rm -rf $P
rocprofv3 --kernel-trace --stats --output-format csv -d $P -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
python profiles/summarize.py stats $P $O/kernel_stats.csv
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_c3_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(round(d["ms_per_step"],3), d["phase_ms"])
PY
head -8 $O/kernel_stats.csv | cut -c1-140
