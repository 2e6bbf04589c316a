#!/usr/bin/env python
"""Per-kernel timings (HIP events on the launch stream) of the C-ABI entry points at the
BASELINE.json sizes: rollout, GAE scan, loss/KL, gradient, Fisher-vector product.
Prints one JSON line per kernel.  Run on the GPU box: python tools/kernel_bench.py"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="13,2,32,2048000;20,6,64,512000;4,1,32,409600")
    ap.add_argument("--rollout", action="store_true")
    args = ap.parse_args()
    import __graft_entry__
    __graft_entry__.build()
    from rllab_amd.envs.env_spec import EnvSpec
    from rllab_amd.policies.gaussian_mlp_policy import GaussianMLPPolicy
    from rllab_amd.spaces import Box
    for cfg in args.configs.split(";"):
        do, da, h, B = cfg.split(",")
        do, da, B = int(do), int(da), int(B)
        hidden = tuple(int(x) for x in h.split("-")) if "-" in h else (int(h), int(h))   # "100-50-25": wide / deep nets
        np.random.seed(0)
        spec = EnvSpec(Box(-np.ones(do), np.ones(do)), Box(-np.ones(da), np.ones(da)))
        pol = GaussianMLPPolicy(spec, hidden_sizes=hidden)
        ops = pol.fused_ops()
        dev = pol.flat_params.device
        g = torch.Generator(device=dev).manual_seed(0)
        obs = torch.randn(do, B, device=dev, generator=g)
        with torch.no_grad():
            mean = pol.mean_planes(obs, pol.flat_params)
        ls = pol.effective_log_std().detach()
        act = mean + torch.randn(da, B, device=dev, generator=g)
        adv = torch.randn(B, device=dev, generator=g)
        w = torch.ones(B, device=dev)
        inp = (obs, act, adv, mean, ls.reshape(-1, 1), w, 1.0 / B)
        v = torch.randn(pol.flat_params.numel(), device=dev, dtype=torch.float64, generator=g)
        sizes = (do,) + hidden + (da,)
        fwd_flops = 2 * sum(sizes[i] * sizes[i + 1] for i in range(len(sizes) - 1))
        for name, fn, mult in (("loss_kl", lambda: (pol.note_raw_write(), ops.loss_stats(inp)), 1.0),
                               ("grad", lambda: ops.loss_grad(inp), 3.0),
                               ("fvp", lambda: ops.fvp(inp, v), 6.0)):
            ms = timeit(fn)
            print(json.dumps(dict(kernel=name, net=[do, da, list(hidden)], samples=B, ms=round(ms, 4),
                                  tflops=round(mult * fwd_flops * B / ms / 1e9, 2),
                                  gbps=round(4 * (do + 3 * da + 1) * B / ms / 1e6, 1))))
        # what TRPO launches: the gradient pass leaves its activations, the products read them back
        ms = timeit(lambda: ops.loss_grad(inp, keep_activations=True))
        print(json.dumps(dict(kernel="grad+cache", net=[do, da, list(hidden)], samples=B, ms=round(ms, 4))))
        ops.loss_grad(inp, keep_activations=True)
        ms = timeit(lambda: ops.fvp(inp, v))
        print(json.dumps(dict(kernel="fvp cached", net=[do, da, list(hidden)], samples=B, ms=round(ms, 4),
                              tflops=round(5.0 * fwd_flops * B / ms / 1e9, 2), variant=ops.fvp_variant(inp))))
        ops.release()


if __name__ == "__main__":
    main()
