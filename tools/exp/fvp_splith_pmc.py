#!/usr/bin/env python
"""Ten launches each of the three-way bf16 split product and of the two-way f16 split product at the headline batch ((13, 2),
(32, 32), 2.048 M samples) and at C5's shard ((20, 6), (64, 64), 512 k), for rocprofv3 --pmc (tools/exp/r06_call19.sh).  GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from tests import test_gpu_update_parity as U
for do, da, h, B in ((13, 2, 32, 2048000), (20, 6, 64, 512000)):
    pol = U._policy(do, da, h)
    ops = pol.fused_ops()
    inp = U._inputs(pol, B, ragged=False, old_equals_new=True)
    v = torch.randn(pol.flat_params.numel(), device="cuda", dtype=torch.float64)
    ops.loss_grad(inp, keep_activations=True)
    for val in ("5", None):
        os.environ.pop("RLLAB_FVP_SPLIT", None)
        if val:
            os.environ["RLLAB_FVP_SPLIT"] = val
        for _ in range(10):
            ops.fvp(inp, v)
        torch.cuda.synchronize()
    ops.release()
