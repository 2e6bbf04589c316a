#!/usr/bin/env python
"""Cached FVP at small batches: what a launch costs beyond its tiles (GPU box).  B = 65 536 is one tile per wavefront."""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests import test_gpu_update_parity as U
pol = U._policy(13, 2, 32)
ops = pol.fused_ops()
for B in (2048, 65536, 131072, 262144, 524288, 1048576, 2048000):
    inp = U._inputs(pol, B, ragged=False, old_equals_new=True)
    v = torch.randn(pol.flat_params.numel(), device="cuda", dtype=torch.float64)
    ops.release()
    ops.loss_grad(inp, keep_activations=True)
    for _ in range(5): ops.fvp(inp, v)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ops.fvp(inp, v)
    e1.record(); torch.cuda.synchronize()
    print(json.dumps(dict(B=B, tiles_per_wave=B / 32 / 2048.0, fvp_us=round(e0.elapsed_time(e1) / 50 * 1e3, 2))))
