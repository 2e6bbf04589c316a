"""Thin helpers over ``torch.distributed`` for the env-sharded data-parallel run
(one process per GPU, backend "nccl" == RCCL over xGMI; "gloo" in CPU tests).

The path has exactly one exchange family -- small sum/min/max all-reduces of
statistics, normal equations, the flat gradient and each Fisher-vector product
(SURVEY.md section 8e).  Every rank applies the identical parameter update, so no
broadcast is needed after the initial parameter sync.
"""
import os
import time

import torch
import torch.distributed as dist

# RLLAB_DIST_FORCE=1: treat an initialised world of ONE rank as distributed, so a 1-GPU box drives every
# collective of the path through the real backend (RCCL) -- tests/test_gpu_rccl.py
_FORCE = bool(os.environ.get("RLLAB_DIST_FORCE"))

# Collective accounting (bench.py's "collectives_per_iter" / "collective_ms_per_iter").  Counting is
# free; timing brackets every collective with a device synchronise on both sides, so it is only ever
# switched on for a few extra iterations AFTER the timed region.
_acct = dict(count=0, seconds=0.0, timing=False, bytes=0)


def reset_accounting(timing=False):
    _acct.update(count=0, seconds=0.0, timing=bool(timing), bytes=0)


def accounting():
    return dict(_acct)


class _account(object):
    __slots__ = ("t", "cuda")

    def __init__(self, t):
        _acct["count"] += 1
        _acct["bytes"] += t.numel() * t.element_size()
        self.cuda = t.is_cuda

    def __enter__(self):
        if _acct["timing"]:
            if self.cuda:
                torch.cuda.synchronize()
            self.t = time.perf_counter()

    def __exit__(self, *exc):
        if _acct["timing"]:
            if self.cuda:
                torch.cuda.synchronize()
            _acct["seconds"] += time.perf_counter() - self.t


def is_distributed():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _FORCE)


def backend():
    return dist.get_backend() if (dist.is_available() and dist.is_initialized()) else None


def world_size():
    return dist.get_world_size() if is_distributed() else 1


def rank():
    return dist.get_rank() if is_distributed() else 0


def _via_host(t):
    """gloo (CPU tests, or two ranks sharing one GPU in tests/) moves device tensors through the
    host; nccl (= RCCL, the production backend) reduces them in place over xGMI."""
    return t.is_cuda and dist.get_backend() == "gloo"


def _all_reduce(t, op):
    if is_distributed():
        with _account(t):
            if _via_host(t):
                h = t.cpu()
                dist.all_reduce(h, op=op)
                t.copy_(h)
            else:
                dist.all_reduce(t, op=op)
    return t


def all_reduce_sum_(t):
    """In-place sum all-reduce (no-op on a single process).  Returns ``t``."""
    return _all_reduce(t, dist.ReduceOp.SUM)


def all_reduce_min_(t):
    return _all_reduce(t, dist.ReduceOp.MIN)


def all_reduce_max_(t):
    return _all_reduce(t, dist.ReduceOp.MAX)


def broadcast_(t, src=0):
    if is_distributed():
        with _account(t):
            if _via_host(t):
                h = t.cpu()
                dist.broadcast(h, src=src)
                t.copy_(h)
            else:
                dist.broadcast(t, src=src)
    return t


def all_gather_rows(t):
    """[n] tensor -> [world, n] with row r = rank r's values (one collective; every rank sees the same rows in
    the same order, so whatever is folded from them -- sums, minima, maxima -- is identical everywhere).  Used
    where a quantity needs both a sum and a min / max: one all-gather instead of two all-reduces."""
    t = t.reshape(-1)
    if not is_distributed():
        return t.unsqueeze(0)
    w = dist.get_world_size()
    with _account(t):
        if dist.get_backend() == "gloo":        # CPU tests, or ranks sharing one GPU in tests/: through the host
            h = t.cpu()
            rows = [torch.empty_like(h) for _ in range(w)]
            dist.all_gather(rows, h)
            return torch.stack(rows).to(t.device)
        # nccl (= RCCL): the flat form, one launch.  No fallback around the collective itself: a rank that
        # caught an error here and issued a different collective would leave its peers hanging.
        t = t.contiguous()
        out = torch.empty(w * t.numel(), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t)
        return out.view(w, t.numel())


def sums(*scalars):
    """All-reduce a handful of 0-d float64 tensors in ONE message; returns a list of
    0-d tensors holding the global sums."""
    packed = torch.stack([s.to(torch.float64).reshape(()) for s in scalars])
    all_reduce_sum_(packed)
    return list(packed.unbind(0))
