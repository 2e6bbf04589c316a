#!/usr/bin/env python
"""Cached Fisher-vector product: the f32-matrix-instruction kernel (RLLAB_FVP_SPLIT=0) against the split-operand bf16
kernel, back to back in one process, and their errors against a float64 product of a small batch."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from tests import test_gpu_update_parity as U

def timed(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

shapes = [(13, 2, 2048000), (4, 1, 409600), (13, 2, 8192000), (20, 6, 512000), (20, 3, 512000)]
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
for do, da, B in shapes:
    pol = U._policy(do, da, 32)
    ops = pol.fused_ops()
    inp = U._inputs(pol, B, ragged=False, old_equals_new=True)
    v = torch.randn(pol.flat_params.numel(), device="cuda", dtype=torch.float64)
    ops.loss_grad(inp, keep_activations=True)
    out = dict(net=[do, da, 32], B=B)
    for tag, val, wps in (("f32_ms", "0", "2"), ("split_wps1_ms", "1", "1"), ("split_ms", "1", "2"), ("f32_again_ms", "0", "2"),
                          ("split_wps1_again_ms", "1", "1"), ("split_again_ms", "1", "2")):
        os.environ["RLLAB_FVP_SPLIT"] = val
        os.environ["RLLAB_FVP_SPLIT_WPS"] = wps
        out[tag] = round(timed(lambda: ops.fvp(inp, v)), 4)
    flops = 71 * 4096 * (B / 32)
    out["split_frac_of_f32_matrix_peak"] = round(flops / (min(out["split_again_ms"], out["split_wps1_again_ms"]) * 1e-3) / 157.3e12, 3)
    print(json.dumps(out), flush=True)
