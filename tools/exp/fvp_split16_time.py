#!/usr/bin/env python
"""fvp_split_kernel (two wavefronts per SIMD, 32-sample tiles) against fvp_split16_kernel (RLLAB_FVP_SPLIT=3: four per
SIMD, 16-sample tiles), interleaved in one process, at the headline batch.  GPU box."""
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

for do, da, B in [(13, 2, 2048000), (4, 1, 409600), (13, 2, 8192000)][:int(sys.argv[1]) if len(sys.argv) > 1 else 3]:
    pol = U._policy(do, da, 32)
    ops = pol.fused_ops()
    inp = U._inputs(pol, B, ragged=False, old_equals_new=True)
    v = torch.randn(pol.flat_params.numel(), device="cuda", dtype=torch.float64)
    ops.loss_grad(inp, keep_activations=True)
    out = dict(net=[do, da, 32], B=B)
    for rnd in range(3):
        for tag, val, wps, abl in (("split32", "1", "0", "0"), ("split16_wps4", "3", "4", "0"), ("split16_wps3", "3", "3", "0"),
                                   ("split16_wps4_no_acts_fetch", "3", "4", "1"), ("split16_wps4_no_x_fetch", "3", "4", "2"),
                                   ("split16_wps4_no_fetch", "3", "4", "3"), ("split16_wps3_no_fetch", "3", "3", "3")):
            os.environ["RLLAB_FVP_SPLIT"] = val
            os.environ["RLLAB_FVP_SPLIT_WPS"] = wps
            os.environ["RLLAB_SPLIT16_ABLATE"] = abl
            out.setdefault(tag + "_ms", []).append(round(timed(lambda: ops.fvp(inp, v)), 4))
    print(json.dumps(out), flush=True)
