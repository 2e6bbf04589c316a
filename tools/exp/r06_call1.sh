#!/bin/bash
# round 6, call 1: the two-legged envs after the role-major eight-lane order (bit-exact tests first), the per-step
# boundary kernel at 4 M envs, the scans' rooflines in the bench line, cheetah at 16 384 / 65 536 envs
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
python -m pytest tests/test_gpu_env_parity.py tests/test_gpu_env_options.py tests/test_gpu_process_parity.py -m gpu -x -q > $O/r06_c1_pytest.log 2>&1
tail -3 $O/r06_c1_pytest.log
python tools/step_kernel_roofline.py --kinds 0,2,3,5,6 > $O/r06_step_kernel_roofline.jsonl 2>/dev/null
python - <<PY
import json
for l in open("gpurun_out/r06_step_kernel_roofline.jsonl"):
    try:
        d = json.loads(l); print(d["kernel"][:40], round(d["avg_launch_ms"], 3), "ms", round(d["frac"], 3))
    except Exception as e: print("ERR", l[:100])
PY
python bench.py --steps 20 --warmup 5 --cpu-budget 6 > $O/r06_bench_line_a.json 2> $O/r06_bench_a.err
python bench.py --workload cheetah1024_trpo_gae --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r06_bench_cheetah1024_trpo_gae_a.json
python bench.py --workload cheetah1024_trpo_gae --n-envs 16384 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r06_bench_cheetah16384_a.json
python bench.py --workload cheetah1024_trpo_gae --n-envs 65536 --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r06_bench_cheetah65536_a.json
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r06_bench_*_a.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["phase_ms"].items()}, round(d["value"]/1e6,1), "Msteps/s", d.get("roofline",{}).get("kernel","")[:40], d.get("roofline_mfma",{}).get("frac"), d.get("roofline_mfma",{}).get("avg_launch_ms"))
        for r in d.get("roofline_scan", []): print("   scan", r["kernel"][:30], round(r["avg_launch_ms"]*1e3,1), "us", round(r["frac"],3))
        for r in d.get("roofline_step_kernel", []): print("   step", r["kernel"][:40], round(r["avg_launch_ms"],3), "ms", round(r["frac"],3))
    except Exception as e: print(f, "ERR", e)
PY
