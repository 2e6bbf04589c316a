#!/bin/bash
# round-5 evidence, second half: the side lines earlier rounds recorded, at the final kernel sources -- larger env counts,
# wide / deep nets, one forced rank over RCCL and over the in-stream peer all-reduce, the two-rank pre-flight record
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
for n in 16384 65536; do
  python bench.py --n-envs $n --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r05_bench_n$n.json
done
for h in "100,50,25" "128,128"; do
  tag=$(echo $h | tr ',' '_')
  python bench.py --hidden $h --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r05_bench_swimmer4096_hidden_$tag.json
done
env RLLAB_DIST_FORCE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29612 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r05_bench_rccl_one_rank.json
env RLLAB_DIST_FORCE=1 RLLAB_PEER_ALLREDUCE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29613 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r05_bench_peer_one_rank.json
env RLLAB_DIST_BACKEND=gloo python tools/preflight_multigpu.py --gpus 2 2>/dev/null | tail -1 > $O/r05_preflight_two_ranks_one_device.json
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["phase_ms"].items()}, round(d["value"]/1e6,1), "Msteps/s", d.get("roofline_mfma",{}).get("frac"), d.get("collectives_per_iter"), d.get("peer_reductions_per_iter"))
    except Exception as e: print(f, "ERR", e)
PY
head -c 600 $O/r05_preflight_two_ranks_one_device.json
