#!/usr/bin/env python
"""A/B of the Fisher-vector product of the wide / deep nets with and without the activation cache, interleaved
(clock drift hits both alike): tools/exp/wide_cache_ab.py [hidden ...]   e.g. 100-50-25 128-128"""
import os, sys, json
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tools.kernel_bench import timeit          # noqa: E402
from rllab_amd.envs.env_spec import EnvSpec    # noqa: E402
from rllab_amd.policies.gaussian_mlp_policy import GaussianMLPPolicy   # noqa: E402
from rllab_amd.spaces import Box               # noqa: E402

do, da, B = 13, 2, 2048000
for h in (sys.argv[1:] or ["100-50-25", "128-128"]):
    hidden = tuple(int(x) for x in h.split("-"))
    np.random.seed(0)
    pol = GaussianMLPPolicy(EnvSpec(Box(-np.ones(do), np.ones(do)), Box(-np.ones(da), np.ones(da))), hidden_sizes=hidden)
    ops = pol.fused_ops()
    dev = pol.flat_params.device
    g = torch.Generator(device=dev).manual_seed(0)
    obs = torch.randn(do, B, device=dev, generator=g)
    with torch.no_grad():
        mean = pol.mean_planes(obs, pol.flat_params)
    act = mean + torch.randn(da, B, device=dev, generator=g)
    inp = (obs, act, torch.randn(B, device=dev, generator=g), mean, pol.effective_log_std().detach().reshape(-1, 1),
           torch.ones(B, device=dev), 1.0 / B)
    v = torch.randn(pol.flat_params.numel(), device=dev, dtype=torch.float64, generator=g)
    rows = []
    for rep in range(4):
        ops.loss_grad(inp)                       # no cache
        plain = timeit(lambda: ops.fvp(inp, v), iters=10)
        ops.loss_grad(inp, keep_activations=True)
        cached = timeit(lambda: ops.fvp(inp, v), iters=10)
        rows.append((round(plain, 3), round(cached, 3)))
    gp = timeit(lambda: ops.loss_grad(inp), iters=10)
    gc = timeit(lambda: ops.loss_grad(inp, keep_activations=True), iters=10)
    print(json.dumps(dict(hidden=hidden, fvp_plain_cached_ms=rows, grad_ms=round(gp, 3), grad_cache_ms=round(gc, 3))))
    ops.release()
