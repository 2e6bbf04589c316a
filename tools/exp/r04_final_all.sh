#!/bin/bash
# end-of-round evidence in one GPU call: profiles of the three BASELINE configs that fit one GPU (each stamps its record of
# pmc_traffic.json with the kernel-source hash), the bench lines against that stamp, the wide-net side lines, the counters
# of the Fisher-vector-product kernels, the two-rank pre-flight record
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash profiles/run_profile.sh r04 > gpurun_out/r04_profile.log 2>&1
bash profiles/run_profile.sh r04 cheetah1024_trpo_gae r04_c5 > gpurun_out/r04_profile_c5.log 2>&1
bash profiles/run_profile.sh r04 cartpole4096_vpg r04_c2 > gpurun_out/r04_profile_c2.log 2>&1
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
bash tools/exp/r04_final_bench.sh
for h in "100,50,25" "128,128"; do
  tag=$(echo $h | tr ',' '_')
  python bench.py --hidden $h --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04_bench_swimmer4096_hidden_$tag.json
  # A/B: the one-wavefront-per-group rollout the four-wavefront shape replaces; the f32-matrix-instruction products
  env RLLAB_SWIMMER_COOP=0 python bench.py --hidden $h --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04_bench_swimmer4096_hidden_${tag}_one_wavefront_rollout.json
  env RLLAB_FVP_SPLIT=0 python bench.py --hidden $h --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04_bench_swimmer4096_hidden_${tag}_f32_products.json
done
env RLLAB_TWO_LEG_WAVE_KERNEL=0 python bench.py --workload cheetah1024_trpo_gae --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04_bench_cheetah1024_trpo_gae_16_envs_per_wavefront.json
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r04_bench_swimmer4096_hidden_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["phase_ms"].items()}, d.get("roofline_mfma",{}).get("frac"), d.get("roofline_mfma",{}).get("avg_launch_ms"))
PY
bash tools/prof_split.sh r04_split > gpurun_out/r04_split_profile.log 2>&1
bash tools/prof_pmc.sh r04_csplit128 csplit python tools/exp/csplit_time.py "13,2,128-128,2048000" s > gpurun_out/r04_csplit128_profile.log 2>&1
env RLLAB_DIST_BACKEND=gloo python tools/preflight_multigpu.py --gpus 2 2>/dev/null | tail -1 > gpurun_out/r04_preflight_two_ranks_one_device.json
ls gpurun_out | head -80
