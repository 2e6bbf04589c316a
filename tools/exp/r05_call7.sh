#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05g
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_two_rank.py tests/test_gpu_shard_rehearsal.py tests/test_gpu_process_parity.py tests/test_gpu_peer_allreduce.py tests/test_gpu_update_parity.py -q > $O/pytest.log 2>&1
tail -12 $O/pytest.log
python bench.py --workload cartpole4096_vpg --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/bench_c2.json
python - <<PY
import json
d=json.loads(open("$O/bench_c2.json").read().strip().splitlines()[-1])
print(round(d["ms_per_step"],3), d["phase_ms"], d["per_rank_ms"], d["rank_skew_ms"])
PY
