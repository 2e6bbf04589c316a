#!/usr/bin/env python
"""Host latency of one small device -> host read behind a short kernel: event.synchronize() against a spin on
event.query() (what misc/device_io.py::HostRead.get waits with).  GPU box."""
import time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
x = torch.zeros(1 << 20, device="cuda")
out = torch.zeros(4, dtype=torch.float64, device="cuda")
buf = torch.empty(4, dtype=torch.float64, pin_memory=True)
def once(spin):
    x.add_(1.0)                      # ~10 us of device work
    out.copy_(x[:4])
    t0 = time.perf_counter()
    buf.copy_(out, non_blocking=True)
    ev = torch.cuda.Event(); ev.record()
    if spin:
        while not ev.query():
            pass
    else:
        ev.synchronize()
    v = buf.numpy().copy()
    return time.perf_counter() - t0
for spin in (False, True, False, True):
    for _ in range(200): once(spin)
    torch.cuda.synchronize()
    ts = sorted(once(spin) for _ in range(2000))
    print("spin" if spin else "synchronize", "median %.1f us, p90 %.1f us" % (ts[1000] * 1e6, ts[1800] * 1e6))
