#!/usr/bin/env python
"""Bench lines of the workloads a rollout-kernel experiment moves: headline and C5 (phase times), one line each.
Used under tools/exp/with_libs.py to compare experiment builds of env_kernels.hip on one box."""
import json, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for args in (["--steps", "20", "--warmup", "5"], ["--workload", "cheetah1024_trpo_gae", "--steps", "10", "--warmup", "3"]):
    out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline"] + args, cwd=ROOT, capture_output=True, text=True).stdout
    try:
        d = json.loads(out.strip().splitlines()[-1])
        print(d["config"]["workload"], round(d["ms_per_step"], 3), {k: round(v, 3) for k, v in d["phase_ms"].items()}, flush=True)
    except Exception as e:
        print("ERR", e, out[-300:])
