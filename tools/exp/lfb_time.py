#!/usr/bin/env python
"""rl_lfb_normal_eq alone at the BASELINE sizes: HIP-event time per launch and a checksum of its output (two builds must
agree bit for bit)."""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from rllab_amd import _lib

def timed(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

torch.manual_seed(0)
for (T, n, do) in [(500, 4096, 13), (500, 1024, 20), (100, 4096, 4), (500, 16384, 13), (97, 333, 21)]:
    dev = "cuda"
    B = T * n
    obs = (torch.randn(do, B, device=dev) * 4.0).contiguous()
    tin = torch.randint(0, T, (B,), device=dev, dtype=torch.int32)
    ret = torch.randn(B, device=dev)
    valid = (torch.rand(B, device=dev) < 0.9).to(torch.uint8)
    F = 2 * do + 4
    ws = torch.empty(512 * 64 * 64 * 8, dtype=torch.uint8, device=dev)
    out = torch.empty((F + 1) * F, dtype=torch.float64, device=dev)
    call = lambda: _lib.check(_lib.lib.rl_lfb_normal_eq(B, do, _lib.ptr(obs), _lib.ptr(tin), _lib.ptr(ret), _lib.ptr(valid), _lib.ptr(ws),
                                                        ws.numel(), _lib.ptr(out), 0, _lib.stream_ptr()), "rl_lfb_normal_eq")
    us = timed(call)
    print("T %d n %d Do %d: lfb_normal_eq %.1f us  sha %s" % (T, n, do, us, hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]), flush=True)
