#!/usr/bin/env python
"""Time the split Fisher-vector product for the production library and every experiment build build/exp/lib_*.so,
interleaved over three rounds in ONE process per library (the first round of a process runs slow).  GPU box only."""
import glob, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r'''
import sys, json, os, torch
sys.path.insert(0, %r)
from tests import test_gpu_update_parity as U
H64 = len(sys.argv) > 1 and sys.argv[1] == "64"
WIDE = len(sys.argv) > 1 and sys.argv[1] == "wide"        # (20, 6) on (32, 32): one wavefront per SIMD
pol = U._policy(20, 6, 64) if H64 else U._policy(20, 6, 32) if WIDE else U._policy(13, 2, 32)
ops = pol.fused_ops()
inp = U._inputs(pol, 512000 if (H64 or WIDE) else 2048000, ragged=False, old_equals_new=True)
v = torch.randn(pol.flat_params.numel(), device="cuda", dtype=torch.float64)
ops.loss_grad(inp, keep_activations=True)
def t(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print(json.dumps(dict(split_ms=[round(t(lambda: ops.fvp(inp, v)), 4) for _ in range(4)])))
''' % ROOT
target = os.path.join(ROOT, "rllab_amd", "librllab_amd.so")
shutil.copy(target, target + ".orig")
try:
    # (the shipped library twice: on the three-way bf16 split, RLLAB_FVP_SPLIT=5, and on its own choice)
    for lib, env in [(target + ".orig", "5"), (target + ".orig", None)] + \
            [(l, None) for l in sorted(glob.glob(os.path.join(ROOT, "build", "exp", "lib_*.so")))] + \
            [(target + ".orig", None), (target + ".orig", "5")]:
        shutil.copy(lib, target)
        e = dict(os.environ)
        e.pop("RLLAB_FVP_SPLIT", None)
        if env:
            e["RLLAB_FVP_SPLIT"] = env
        print("==", os.path.basename(lib), "RLLAB_FVP_SPLIT=%s" % env if env else "", flush=True)
        subprocess.call([sys.executable, "-c", CODE] + sys.argv[1:], env=e)
finally:
    shutil.copy(target + ".orig", target)
