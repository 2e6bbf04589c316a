#!/bin/bash
# the round's evidence in one GPU call at the final kernel sources (after the two-way f16 split products, section 11 of
# r06_notes): r06_final_all.sh + the counters of both split arithmetics (r06_call19.sh)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/r06_pytest_gpu.log 2>&1
tail -3 $O/r06_pytest_gpu.log
bash profiles/run_profile.sh r06 > $O/r06_profile.log 2>&1
bash profiles/run_profile.sh r06 cheetah1024_trpo_gae r06_c5 > $O/r06_profile_c5.log 2>&1
bash profiles/run_profile.sh r06 cartpole4096_vpg r06_c2 > $O/r06_profile_c2.log 2>&1
cp $O/pmc_traffic.json profiles/pmc_traffic.json
python bench.py --steps 20 --warmup 5 > $O/r06_bench_line.json 2> $O/r06_bench.err
for w in cartpole4096_vpg cheetah1024_trpo_gae double_pendulum4096_trpo; do
  python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r06_bench_$w.json
done
bash tools/exp/r06_call5.sh > $O/r06_split_profile.log 2>&1
bash tools/exp/r06_call19.sh > $O/r06_splith_profile.log 2>&1
python tools/exp/fvp_split16_time.py 1 2>&1 | grep "^{" > $O/r06_split16_time.txt
bash tools/exp/r06_call8.sh > $O/r06_csplit_profile.log 2>&1
timeout 600 python tools/kernel_bench.py --configs "13,2,100-50-25,2048000;13,2,128-128,2048000;20,6,128-64,512000" 2>&1 | grep "^{" > $O/r06_wide_kernel_bench.txt
for n in 16384 65536; do
  python bench.py --n-envs $n --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r06_bench_n$n.json
done
python bench.py --workload cheetah1024_trpo_gae --n-envs 16384 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r06_bench_cheetah16384.json
python bench.py --workload cheetah1024_trpo_gae --n-envs 65536 --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r06_bench_cheetah65536.json
for h in "100,50,25" "128,128"; do
  tag=$(echo $h | tr ',' '_')
  python bench.py --hidden $h --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r06_bench_swimmer4096_hidden_$tag.json
done
env RLLAB_DIST_FORCE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29612 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r06_bench_rccl_one_rank.json
env RLLAB_DIST_FORCE=1 RLLAB_PEER_ALLREDUCE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29613 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r06_bench_peer_one_rank.json
env RLLAB_DIST_BACKEND=gloo python tools/preflight_multigpu.py --gpus 2 2>/dev/null | tail -1 > $O/r06_preflight_two_ranks_one_device.json
python tools/exp/fallback_probe.py 2>&1 | grep -v amdgpu.ids > $O/r06_fallback_probe_policies.txt
python tools/exp/fallback_probe_envs.py 2>&1 | grep -v amdgpu.ids > $O/r06_fallback_probe_envs.txt
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r06_bench_*.json")):
    if f.endswith("_a.json") or "_c2_" in f or "_c6_" in f: continue
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["phase_ms"].items()}, round(d["value"]/1e6,1), "Msteps/s", (d.get("roofline") or {}).get("frac"), (d.get("roofline") or {}).get("traffic"), (d.get("roofline") or {}).get("issue_frac"), (d.get("roofline_mfma") or {}).get("frac"), (d.get("roofline_mfma") or {}).get("avg_launch_ms"))
    except Exception as e: print(f, "ERR", e)
PY
tail -3 $O/r06_pytest_gpu.log
