#!/usr/bin/env python
"""Cached Fisher-vector products of the 64-unit / wide nets: the launches of one CG iteration (_fvp_into: product kernel
+ row reduction) timed back to back, split-operand cooperative kernel vs the f32-matrix-instruction kernels.
    python tools/exp/csplit_time.py ["20,6,64-64,512000;13,2,128-128,2048000"] [only_split]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from tests import test_gpu_update_parity as U
from tests.test_gpu_csplit import _policy

def timed(fn, n=30):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

cfgs = sys.argv[1] if len(sys.argv) > 1 else "20,6,64-64,512000;13,2,128-128,2048000;13,2,100-50-25,2048000"
only_split = len(sys.argv) > 2
for cfg in cfgs.split(";"):
    do, da, h, B = cfg.split(",")
    do, da, B, hidden = int(do), int(da), int(B), tuple(int(x) for x in h.split("-"))
    pol = _policy(do, da, hidden)
    ops = pol.fused_ops()
    inp = U._inputs(pol, B, ragged=False, old_equals_new=True)
    v = torch.randn(pol.flat_params.numel(), device="cuda", dtype=torch.float64)
    ops.loss_grad(inp, keep_activations=True)
    ws_, keep_ = ops._workspace(v.device), ops._batch(inp)
    v32 = ops.layout.pack(v.to(torch.float32)).contiguous()
    out64 = torch.empty(ops.n_kernel, dtype=torch.float64, device=v.device)
    out = dict(net=[do, da, list(hidden)], B=B)
    for tag, val in ((("split_ms", os.environ.get("CS_FORCE", "1")),) if only_split else (("f32_ms", "0"), ("split_ms", "1"), ("f32_again_ms", "0"), ("split_again_ms", "1"))):
        os.environ["RLLAB_FVP_SPLIT"] = val
        out[tag] = round(timed(lambda: ops._fvp_into(keep_[0], ws_, v32, out64, inp)), 4)
    os.environ.pop("RLLAB_FVP_SPLIT")
    out["variant"] = ops.fvp_variant(inp)
    print(json.dumps(out), flush=True)
