#!/usr/bin/env python
"""rl_gae / rl_path_scan alone at the BASELINE sizes: HIP-event time per launch, algorithmic GB/s."""
import os, sys
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

for (T, n, do) in [(500, 4096, 13), (500, 1024, 20), (100, 4096, 4), (500, 16384, 13), (500, 65536, 13)]:
    dev = "cuda"
    r = torch.randn(T, n, device=dev); v = torch.randn(T, n, device=dev, dtype=torch.float64)
    d = (torch.rand(T, n, device=dev) < 0.01).to(torch.uint8)
    obs = torch.randn(do, T, n, device=dev); cf = torch.randn(2 * do + 4, device=dev, dtype=torch.float64)
    adv = torch.empty(T, n, device=dev); ret = torch.empty_like(adv); und = torch.empty_like(adv)
    tin = torch.empty(T, n, device=dev, dtype=torch.int32); valid = torch.empty(T, n, device=dev, dtype=torch.uint8)
    vals = torch.empty(T, n, device=dev, dtype=torch.float64)
    g = timed(lambda: _lib.check(_lib.lib.rl_gae(T, n, _lib.ptr(r), _lib.ptr(v), _lib.ptr(d), 0.99, 0.97, _lib.ptr(adv), _lib.ptr(ret), _lib.ptr(und), _lib.stream_ptr())))
    p = timed(lambda: _lib.check(_lib.lib.rl_path_scan(T, n, do, _lib.ptr(d), _lib.ptr(obs), _lib.ptr(cf), 1, _lib.ptr(tin), _lib.ptr(valid), _lib.ptr(vals), _lib.stream_ptr())))
    B = T * n
    print("T %d n %d Do %d: gae %.1f us (%.2f TB/s)  path_scan %.1f us (%.2f TB/s)" % (T, n, do, g, 25 * B / g / 1e6, p, (4 * do + 14) * B / p / 1e6), flush=True)
