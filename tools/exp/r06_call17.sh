#!/bin/bash
# round 6, call 17: the two-way f16 split product as the library's choice: the whole GPU suite, then the headline and C5
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 1500 python -m pytest tests/test_gpu_fvp_split.py -m gpu -q -x 2>&1 | tail -15 > $O/r06_c17_pytest_split.log
cat $O/r06_c17_pytest_split.log
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $O/r06_c17_pytest_all.log
cat $O/r06_c17_pytest_all.log
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r06_c17_bench_$i.json
RLLAB_FVP_SPLIT=5 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r06_c17_bench_bf16_$i.json
python bench.py --workload cheetah1024_trpo_gae --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r06_c17_bench_c5_$i.json
RLLAB_FVP_SPLIT=5 python bench.py --workload cheetah1024_trpo_gae --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r06_c17_bench_c5_bf16_$i.json
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r06_c17_bench_*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    print(f.split("/")[-1], round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["phase_ms"].items()}, d.get("roofline_mfma",{}).get("frac"), d.get("roofline_mfma",{}).get("avg_launch_ms"))
PY
