#!/bin/bash
# after the waits behind the trajectory stores left the rollout loops: rollout parity of every shape, then every bench
# line a rollout kernel moves
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_wide_nets.py tests/test_gpu_env_parity.py tests/test_gpu_shard_rehearsal.py tests/test_gpu_env_options.py tests/test_gpu_adaptive_std.py -x -q 2>&1 | tail -5 > gpurun_out/r04g_tests.log
tail -3 gpurun_out/r04g_tests.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04g_bench_line.json
for w in cartpole4096_vpg cheetah1024_trpo_gae double_pendulum4096_trpo; do
  python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04g_bench_$w.json
done
python bench.py --n-envs 16384 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04g_bench_n16384.json
python bench.py --hidden 128,128 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04g_bench_hidden_128_128.json
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r04g_bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["phase_ms"].items()}, round(d["value"]/1e6,1))
    except Exception as e: print(f, "ERR", e)
PY
