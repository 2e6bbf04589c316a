#!/usr/bin/env python
"""Ten launches of csplit_fvp_kernel per net at 2.048 M samples, for rocprofv3 --pmc (tools/exp/r06_call8.sh).  GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from tests import test_gpu_update_parity as U
from tests.test_gpu_csplit import _policy
for hidden in ((100, 50, 25), (128, 128)):
    pol = _policy(13, 2, hidden)
    ops = pol.fused_ops()
    inp = U._inputs(pol, 2048000, ragged=False, old_equals_new=True)
    v = torch.randn(pol.flat_params.numel(), device="cuda", dtype=torch.float64)
    ops.loss_grad(inp, keep_activations=True)
    for _ in range(10):
        ops.fvp(inp, v)
    torch.cuda.synchronize()
