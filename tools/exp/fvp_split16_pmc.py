#!/usr/bin/env python
"""Ten launches each of fvp_split_kernel<13, 2, 2>, fvp_split16_kernel<13, 2, 4> and <13, 2, 3> at the headline batch, for
rocprofv3 --pmc (tools/exp/r06_call5.sh).  GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from tests import test_gpu_update_parity as U
pol = U._policy(13, 2, 32)
ops = pol.fused_ops()
inp = U._inputs(pol, 2048000, ragged=False, old_equals_new=True)
v = torch.randn(pol.flat_params.numel(), device="cuda", dtype=torch.float64)
ops.loss_grad(inp, keep_activations=True)
for val, wps in (("1", "0"), ("3", "4"), ("3", "3")):
    os.environ["RLLAB_FVP_SPLIT"], os.environ["RLLAB_FVP_SPLIT_WPS"] = val, wps
    for _ in range(10):
        ops.fvp(inp, v)
    torch.cuda.synchronize()
