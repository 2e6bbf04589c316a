#!/usr/bin/env python
"""The two-way f16 split Fisher-vector product (variant 4) against float64 autograd, the bf16 three-way split (variant 1,
RLLAB_FVP_SPLIT=5) and the f32 matrix instructions (variant 0), with timings.  GPU box only."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import test_gpu_update_parity as U
from tests import test_gpu_fvp_split as T

def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

def run(do, da, h, B, obs_scale=1.0, vec_scale=1.0, time_B=None):
    pol = U._policy(do, da, h)
    ops = pol.fused_ops()
    inp = list(U._inputs(pol, B, old_equals_new=True))
    if obs_scale != 1.0:
        inp[0] = inp[0] * obs_scale
        with torch.no_grad():
            inp[3] = pol.mean_planes(inp[0].double(), pol.flat_params.double()).float()     # old mean == new mean
    inp = tuple(inp)
    rng = np.random.RandomState(7)
    vs = [torch.as_tensor(rng.randn(pol.flat_params.numel()) * vec_scale, device="cuda") for _ in range(2)]
    want = T._f64_products(pol, inp, vs) if B <= 70000 else T._f64_products_chunked(pol, inp, vs)
    ops.loss_grad(inp, keep_activations=True)
    res = {}
    for name, env in (("f16x2", None), ("bf16x3", "5"), ("f32", "0")):
        if env is None: os.environ.pop("RLLAB_FVP_SPLIT", None)
        else: os.environ["RLLAB_FVP_SPLIT"] = env
        var = ops.fvp_variant(inp)
        got = [ops.fvp(inp, v) for v in vs]
        errs = [float((g - w).abs().max()) / float(w.abs().max()) for g, w in zip(got, want)]
        nrm = [float((g - w).norm() / w.norm()) for g, w in zip(got, want)]
        ms = timeit(lambda: ops.fvp(inp, vs[0]))
        res[name] = dict(variant=var, max_err=max(errs), norm_err=max(nrm), ms=round(ms, 4), finite=bool(all(torch.isfinite(g).all() for g in got)))
    os.environ.pop("RLLAB_FVP_SPLIT", None)
    print(json.dumps(dict(shape=(do, da, h), B=B, obs_scale=obs_scale, vec_scale=vec_scale, absmax=float(ops._absmax), **res)), flush=True)

if __name__ == "__main__":
    run(13, 2, 32, 64000)
    run(20, 6, 64, 64000)
    run(4, 1, 32, 4096)
    run(13, 2, 32, 64000, vec_scale=1e-6)
    run(13, 2, 32, 64000, vec_scale=1e4)
    run(20, 6, 64, 64000, vec_scale=1e-5)
    run(13, 2, 32, 64000, obs_scale=300.0)
    run(13, 2, 32, 64000, obs_scale=1e-3)
    run(13, 2, 32, 2048000)
    run(20, 6, 64, 512000)
