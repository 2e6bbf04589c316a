#!/usr/bin/env python
"""Time the CACHED Fisher-vector product (what TRPO's CG loop launches) and the gradient / loss passes for every
experiment library build/exp/lib_*.so (GPU box scratch copy only)."""
import glob, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r'''
import sys, json, numpy as np, torch
sys.path.insert(0, %r)
from tests import test_gpu_update_parity as U
for do, da, h, B in ((13, 2, 32, 2048000), (4, 1, 32, 409600)):
    pol = U._policy(do, da, h)
    ops = pol.fused_ops()
    inp = U._inputs(pol, B, ragged=False, old_equals_new=True)
    v = torch.randn(pol.flat_params.numel(), device="cuda", dtype=torch.float64)
    ops.loss_grad(inp, keep_activations=True)
    def t(fn, n=30):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    fvp = t(lambda: ops.fvp(inp, v))
    assert ops._acts_tag is not None
    print(json.dumps(dict(net=[do, da, h], B=B, fvp_cached_ms=round(fvp, 4))))
''' % ROOT
target = os.path.join(ROOT, "rllab_amd", "librllab_amd.so")
shutil.copy(target, target + ".orig")
try:
    for lib in [target + ".orig"] + sorted(glob.glob(os.path.join(ROOT, "build", "exp", "lib_*.so"))):
        shutil.copy(lib, target)
        print("==", os.path.basename(lib), flush=True)
        subprocess.call([sys.executable, "-c", CODE])
finally:
    shutil.copy(target + ".orig", target)
