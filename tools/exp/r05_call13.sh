#!/bin/bash
# round 5, one-body-per-lane two-leg program on the device: the whole GPU suite (every launch shape of HalfCheetah /
# Walker2D is replayed against the host build), C5's bench line, the rollout kernel's counters, learning curves
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05m
mkdir -p $O gpurun_out/curves
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1
tail -15 $O/pytest_gpu.log
for k in 1 2; do
python bench.py --workload cheetah1024_trpo_gae --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c5_$k.json
done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c3.json
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["phase_ms"].items()}, d["roofline"]["kernel"][:60], d["roofline"].get("avg_launch_ms"))
    except Exception as e: print(f, "ERR", e)
PY
P=/tmp/prof_r05m; rm -rf $P; mkdir -p $P
BENCH2="python bench.py --workload cheetah1024_trpo_gae --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $P/sq -- $BENCH2 > /dev/null 2>&1
python profiles/summarize.py pmc $P/sq $O/c5_pmc_sq.csv
grep -h "two_leg\|^kernel" $O/c5_pmc_sq.csv | cut -c1-300
for cfg in "half_cheetah 1024 100 64 0.97" "walker2d 1024 100 64 0.97"; do
  set -- $cfg
  timeout 300 python examples/run_trpo.py --env $1 --n-envs $2 --n-itr $3 --hidden $4 --gae-lambda $5 --quiet --csv gpurun_out/curves/r05_body_lanes_$1.csv 2>&1 | grep -v amdgpu.ids | tail -1
  python - <<PY
import csv
r=list(csv.DictReader(open("gpurun_out/curves/r05_body_lanes_$1.csv")))
print("$1", "iters", len(r), "AverageReturn first / mean of last 5: %.3f / %.3f" % (float(r[0]["AverageReturn"]), sum(float(x["AverageReturn"]) for x in r[-5:])/5), "max MeanKL %.5f" % max(float(x["MeanKL"]) for x in r))
PY
done
