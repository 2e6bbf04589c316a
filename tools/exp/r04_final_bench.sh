#!/bin/bash
# the round's bench lines: headline (with the CPU baseline leg), side workloads, larger env counts (GPU box)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python bench.py > gpurun_out/r04_bench_line.json 2> gpurun_out/r04_bench.err
for w in cartpole4096_vpg cheetah1024_trpo_gae double_pendulum4096_trpo; do
  python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04_bench_$w.json
done
for n in 16384 65536; do
  python bench.py --n-envs $n --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04_bench_n$n.json
done
env RLLAB_DIST_FORCE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29612 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04_bench_rccl_one_rank.json
env RLLAB_DIST_FORCE=1 RLLAB_PEER_ALLREDUCE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29613 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04_bench_peer_one_rank.json
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r04_bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["phase_ms"].items()}, round(d["value"]/1e6,1), "Msteps/s", d.get("roofline",{}).get("frac"), d.get("roofline_mfma",{}).get("frac"), d.get("collectives_per_iter"), d.get("peer_reductions_per_iter"))
    except Exception as e: print(f, "ERR", e)
PY
