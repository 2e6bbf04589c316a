"""Numeric helpers of rllab/misc/special.py used on the hot path.

``discount_cumsum`` (reference :107-111, scipy.lfilter in float64) runs as the
HIP scan kernel ``rl_discount_cumsum`` (csrc/scan_kernels.hip): f64 accumulation,
f32 storage.  ``explained_variance_1d`` (reference :51-59) is a handful of
device reductions in float64.
"""
import numpy as np
import torch

from rllab_amd import _lib


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("rllab_amd.misc.special: no HIP device -- the scan kernels have no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def discount_cumsum(x, discount, dones=None):
    """y[t] = x[t] + discount * y[t+1] along axis 0.

    ``x``: 1-D or [T, n] array / tensor (n independent columns).  ``dones``
    (optional, same shape, uint8/bool) marks path ends inside a column.
    Returns the same container type as ``x`` (float32 on the device, cast back
    to the input dtype for numpy inputs).
    """
    is_np = not torch.is_tensor(x)
    dev = _device()
    xt = torch.as_tensor(np.ascontiguousarray(x) if is_np else x)
    shape = xt.shape
    xt = xt.to(device=dev, dtype=torch.float32).reshape(shape[0], -1).contiguous()
    T, n = xt.shape
    dt = None
    if dones is not None:
        dt = torch.as_tensor(np.ascontiguousarray(dones) if not torch.is_tensor(dones) else dones)
        dt = dt.to(device=dev).to(torch.uint8).reshape(T, n).contiguous()
    y = torch.empty_like(xt)
    _lib.check(_lib.lib.rl_discount_cumsum(T, n, _lib.ptr(xt), _lib.ptr(dt), float(discount), _lib.ptr(y),
                                           _lib.stream_ptr()), "rl_discount_cumsum")
    y = y.reshape(shape)
    if is_np:
        return y.cpu().numpy().astype(np.asarray(x).dtype if np.asarray(x).dtype.kind == 'f' else np.float64)
    return y


def discount_return(x, discount):
    x = np.asarray(x)
    return np.sum(x * (discount ** np.arange(len(x))))


def explained_variance_1d(ypred, y, weights=None):
    """1 - Var[y - ypred] / (Var[y] + 1e-8) with the reference's zero-variance
    special cases.  Accepts numpy arrays or torch tensors; optional 0/1 weights
    restrict the statistic to valid samples of a dense batch."""
    yp = torch.as_tensor(ypred).to(torch.float64).reshape(-1)
    yt = torch.as_tensor(y).to(torch.float64).reshape(-1).to(yp.device)
    if weights is not None:
        w = torch.as_tensor(weights).to(torch.float64).reshape(-1).to(yp.device)
    else:
        w = torch.ones_like(yt)
    cnt = w.sum()

    def var(v):
        m = (v * w).sum() / cnt
        return (((v - m) ** 2) * w).sum() / cnt
    vary = float(var(yt))
    if np.isclose(vary, 0):
        return 0 if float(var(yp)) > 0 else 1
    return 1 - float(var(yt - yp)) / (vary + 1e-8)
