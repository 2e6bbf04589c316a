#!/bin/bash
# round-5 evidence, first half (one GPU call): the whole GPU suite at HEAD, profiles of the headline and of C5's shard
# (each stamps its record of pmc_traffic.json with the kernel-source hash), the bench lines against that stamp, the
# counters of the split Fisher-vector product
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/r05_pytest_gpu.log 2>&1
tail -3 $O/r05_pytest_gpu.log
bash profiles/run_profile.sh r05 > $O/r05_profile.log 2>&1
bash profiles/run_profile.sh r05 cheetah1024_trpo_gae r05_c5 > $O/r05_profile_c5.log 2>&1
cp $O/pmc_traffic.json profiles/pmc_traffic.json
python bench.py --steps 20 --warmup 5 > $O/r05_bench_line.json 2> $O/r05_bench.err
for w in cartpole4096_vpg cheetah1024_trpo_gae double_pendulum4096_trpo; do
  python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r05_bench_$w.json
done
env RLLAB_FVP_SPLIT=0 python bench.py --workload cheetah1024_trpo_gae --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r05_bench_cheetah1024_trpo_gae_f32_products.json
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["phase_ms"].items()}, round(d["value"]/1e6,1), "Msteps/s", d.get("roofline",{}).get("frac"), d.get("roofline",{}).get("traffic"), d.get("roofline_mfma",{}).get("frac"), d.get("roofline_mfma",{}).get("avg_launch_ms"))
    except Exception as e: print(f, "ERR", e)
PY
bash tools/prof_split.sh r05_split > $O/r05_split_profile.log 2>&1
tail -12 $O/r05_split_profile.log | cut -c1-300
ls $O | head -80
