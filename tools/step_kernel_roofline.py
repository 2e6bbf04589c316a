#!/usr/bin/env python
"""HBM roofline of the per-step VecEnv boundary kernel (rl_vecenv_step).

The fused rollout keeps env state in registers for the whole horizon and is issue-bound, so
its HBM fraction says little.  The kernel the north star's "fraction of the HBM roofline on the step kernel"
applies to is the one-transition-per-launch boundary ``rl_vecenv_step``: it reads state + action and writes
state + obs + reward + done every launch.  This script times it at a size that fills the chip (default 4 M
envs = 65 536 wavefronts) and prints one JSON line per env kind:
algorithmic bytes per env-step = 4 (2 S + Da + Do + 1) + 1 (SURVEY 8d, S = persisted state floats) + 8 (ts
read + write), achieved GB/s from HIP-event timing on the launch stream, fraction of the 8 TB/s peak.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-envs", type=int, default=1 << 22)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--kinds", default="0,1,4,7,2,3")
    args = ap.parse_args()
    import torch
    import __graft_entry__
    __graft_entry__.build()
    from rllab_amd import _lib
    from rllab_amd.envs.hip_env import HipVecEnv
    names = {0: "cartpole", 1: "double_pendulum", 2: "swimmer", 3: "half_cheetah", 4: "cartpole_swingup",
             5: "walker2d", 6: "hopper", 7: "inverted_double_pendulum"}
    for kind in [int(k) for k in args.kinds.split(",")]:
        n = args.n_envs
        v = HipVecEnv(kind, n, 0, normalize=True, seed=1)
        q = v.q
        v.reset()
        act = (torch.rand((q["act_dim"], n), device=v.device) * 2 - 1).contiguous()

        def launch():
            _lib.check(_lib.lib.rl_vecenv_step(
                kind, n, 1, 1.0, 0, 1, _lib.ptr(v.state), _lib.ptr(v.ts), _lib.ptr(act), None, v.seed,
                v.step_counter, 0, None, _lib.ptr(v._obs), _lib.ptr(v._reward), _lib.ptr(v._done), _lib.stream_ptr()),
                "rl_vecenv_step")
            v.step_counter += 1
        for _ in range(args.warmup):
            launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.steps):
            launch()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.steps
        S, Do, Da = q["state_dim"], q["obs_dim"], q["act_dim"]
        step_bytes = 4 * (2 * S + Da + Do + 1) + 1 + 8
        gbs = step_bytes * n / (ms * 1e-3) / 1e9
        print(json.dumps({"kernel": "vecenv_step_kernel<%s>" % names[kind], "n_envs": n, "avg_launch_ms": ms,
                          "env_steps_per_s": n / (ms * 1e-3), "bytes_per_env_step": step_bytes,
                          "roofline": {"bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s",
                                       "frac": gbs / 8000.0}}))
        del v, act
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
