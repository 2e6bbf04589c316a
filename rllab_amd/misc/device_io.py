"""Small device -> host reads that do not stall the launch queue.

A ``float(tensor)`` / ``.cpu()`` on a device tensor is a blocking copy: the host waits for everything
queued before it and only then goes on preparing the next launches, so every such read leaves the GPU idle
for the Python time that follows.  ``read_async`` starts the copy into pinned memory and returns a handle;
the caller keeps launching and calls ``get()`` where the value is really needed (one event wait).
"""
import numpy as np
import torch

_FREE = {}      # (dtype, numel) -> pinned buffers not owned by a pending read


def _acquire(dtype, numel):
    free = _FREE.setdefault((dtype, numel), [])
    return free.pop() if free else torch.empty(numel, dtype=dtype, pin_memory=True)


def _release(buf):
    free = _FREE.setdefault((buf.dtype, buf.numel()), [])
    if len(free) < 32:
        free.append(buf)


class HostRead(object):
    """Handle of one pending read; ``get()`` returns a numpy copy of the values.  The pinned buffer belongs to
    the handle until it is read (or dropped), however long that takes -- a read may stay pending across an
    iteration (LinearFeatureBaseline) while many others come and go."""

    def __init__(self, buf, event, shape, pooled):
        self._buf, self._event, self._shape, self._pooled = buf, event, shape, pooled
        self._value = None

    def get(self):
        if self._value is None:
            if self._event is not None:
                self._event.synchronize()
            self._value = self._buf.numpy().reshape(self._shape).copy()
            self._drop()
        return self._value

    def _drop(self):
        if self._buf is not None and self._pooled:
            if self._event is None or self._event.query():
                _release(self._buf)        # an unfinished copy must not see its target reused: let it go instead
        self._buf = self._event = None

    def __del__(self):
        try:
            self._drop()
        except Exception:
            pass


def read_async(t):
    """Start copying the (small) tensor ``t`` to the host; returns a ``HostRead``."""
    t = t.detach()
    if not t.is_cuda:
        return HostRead(t.reshape(-1).clone(), None, tuple(t.shape), False)
    flat = t.reshape(-1)
    buf = _acquire(flat.dtype, flat.numel())
    buf.copy_(flat, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    return HostRead(buf, ev, tuple(t.shape), True)


def upload_async(values, dtype, device):
    """Host values -> new device tensor through pinned memory, without blocking the host."""
    a = np.ascontiguousarray(np.asarray(values))
    src = torch.from_numpy(a).to(dtype)
    if device.type != "cuda":
        return src.clone()
    buf = torch.empty(src.numel(), dtype=dtype, pin_memory=True)   # stays alive until the copy engine is done with it
    buf.copy_(src.reshape(-1))
    return buf.to(device, non_blocking=True).reshape(a.shape)
