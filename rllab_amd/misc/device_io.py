"""Small device -> host reads that do not stall the launch queue.

A ``float(tensor)`` / ``.cpu()`` on a device tensor is a blocking copy: the host waits for everything
queued before it and only then goes on preparing the next launches, so every such read leaves the GPU idle
for the Python time that follows.  ``read_async`` starts the copy into pinned memory and returns a handle;
the caller keeps launching and calls ``get()`` where the value is really needed (one event wait).
"""
import numpy as np
import torch

_RING = {}
_RING_SIZE = 16


def _pinned(dtype, numel):
    key = (dtype, numel)
    ring = _RING.get(key)
    if ring is None:
        ring = _RING[key] = [[torch.empty(numel, dtype=dtype, pin_memory=True) for _ in range(_RING_SIZE)], 0]
    bufs, i = ring
    ring[1] = (i + 1) % _RING_SIZE
    return bufs[i]


class HostRead(object):
    """Handle of one pending read; ``get()`` returns a numpy copy of the values."""

    def __init__(self, buf, event, shape):
        self._buf, self._event, self._shape = buf, event, shape
        self._value = None

    def get(self):
        if self._value is None:
            if self._event is not None:
                self._event.synchronize()
            self._value = self._buf.numpy().reshape(self._shape).copy()
            self._buf = self._event = None
        return self._value


def read_async(t):
    """Start copying the (small) tensor ``t`` to the host; returns a ``HostRead``."""
    t = t.detach()
    if not t.is_cuda:
        return HostRead(t.reshape(-1).clone(), None, tuple(t.shape))
    flat = t.reshape(-1)
    buf = _pinned(flat.dtype, flat.numel())
    buf.copy_(flat, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    return HostRead(buf, ev, tuple(t.shape))


def upload_async(values, dtype, device):
    """Host values -> new device tensor through pinned memory, without blocking the host."""
    a = np.ascontiguousarray(np.asarray(values))
    src = torch.from_numpy(a).to(dtype)
    if device.type != "cuda":
        return src.clone()
    buf = _pinned(dtype, src.numel())
    buf.copy_(src.reshape(-1))
    return buf.to(device, non_blocking=True).reshape(a.shape)
