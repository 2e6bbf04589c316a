"""Thin helpers over ``torch.distributed`` for the env-sharded data-parallel run
(one process per GPU, backend "nccl" == RCCL over xGMI; "gloo" in CPU tests).

The path has exactly one exchange family -- small sum/min/max all-reduces of
statistics, normal equations, the flat gradient and each Fisher-vector product
(SURVEY.md section 8e).  Every rank applies the identical parameter update, so no
broadcast is needed after the initial parameter sync.
"""
import torch
import torch.distributed as dist


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


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
    if dist.get_backend() == "gloo":            # CPU tests, or ranks sharing one GPU in tests/: through the host
        h = t.cpu()
        rows = [torch.empty_like(h) for _ in range(w)]
        dist.all_gather(rows, h)
        return torch.stack(rows).to(t.device)
    t = t.contiguous()
    try:
        out = torch.empty(w * t.numel(), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t)
        return out.view(w, t.numel())
    except (RuntimeError, AttributeError):      # a backend without the flat form: the list form is universal
        rows = [torch.empty_like(t) for _ in range(w)]
        dist.all_gather(rows, t)
        return torch.stack(rows)


def sums(*scalars):
    """All-reduce a handful of 0-d float64 tensors in ONE message; returns a list of
    0-d tensors holding the global sums."""
    packed = torch.stack([s.to(torch.float64).reshape(()) for s in scalars])
    all_reduce_sum_(packed)
    return list(packed.unbind(0))
