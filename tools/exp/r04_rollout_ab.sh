#!/bin/bash
# the rollout kernels touched after the first evidence run: parity tests of the shapes, then the bench lines they move
# (A/B against the shapes they replace through the RLLAB_* switches)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_wide_nets.py tests/test_gpu_env_parity.py tests/test_gpu_shard_rehearsal.py tests/test_gpu_env_options.py -x -q 2>&1 | tail -15 > gpurun_out/r04f_tests.log
tail -3 gpurun_out/r04f_tests.log
python bench.py --workload cheetah1024_trpo_gae --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04f_bench_c5.json
for h in "100,50,25" "128,128"; do
  tag=$(echo $h | tr ',' '_')
  python bench.py --hidden $h --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04f_bench_hidden_$tag.json
  env RLLAB_SWIMMER_COOP=0 python bench.py --hidden $h --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04f_bench_hidden_${tag}_one_wavefront.json
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r04f_bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["phase_ms"].items()}, d["roofline"]["kernel"][:50])
    except Exception as e: print(f, "ERR", e)
PY
