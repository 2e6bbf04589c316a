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
_acct = dict(count=0, seconds=0.0, timing=False, bytes=0, peer=0)


def reset_accounting(timing=False):
    _acct.update(count=0, seconds=0.0, timing=bool(timing), bytes=0, peer=0)


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


# -- one-shot peer all-reduce inside the update's launch train (csrc/peer_kernels.hip) ---------------------------------
# The sums on CG's critical path -- the flat gradient and each Fisher-vector product, float64 vectors of P doubles --
# need not return to the host: every rank writes its row into every peer's mailbox (hipIpc-mapped device memory, xGMI
# on a multi-GPU node), raises a flag and sums the world's rows in rank order, one small launch on the update's stream.
# Opt-in (RLLAB_PEER_ALLREDUCE=1) until an 8-GPU run has confirmed it against RCCL; everything else (statistics,
# normal equations, loss sums) stays a torch.distributed collective.
_peer = None
_peer_refused = None      # why the peer path was refused by the start-up checks (all ranks agree), or None


def _agree(ok):
    """min over ranks of a local success flag, through the backend (every rank takes the same branch afterwards)."""
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32,
                        device="cuda" if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(int(flag.item()))


class PeerReducer(object):
    """Mailboxes of the world, exchanged once at start-up (host side, torch.distributed.all_gather_object).

    Construction is a COLLECTIVE that cannot leave a rank behind: every stage that may fail on one rank only
    (fine-grained allocation, hipIpc export, mapping a peer's handle across devices) ends in an all-reduce-min of the
    stage's success, and a failure anywhere makes every rank release what it holds and raise ``PeerUnavailable`` --
    ``peer_reducer()`` then logs the reason and the run stays on the backend's all-reduce (RCCL)."""

    class PeerUnavailable(RuntimeError):
        pass

    def __init__(self, max_n=1 << 16):
        import ctypes
        from rllab_amd import _lib
        self._lib, self._ct = _lib, ctypes
        self.rank, self.world, self.max_n = dist.get_rank(), dist.get_world_size(), int(max_n)
        if self.world > 8:
            raise RuntimeError("peer all-reduce: one node, at most 8 ranks (got %d)" % self.world)
        self._own, self._opened, self.err = None, [], None
        why = None
        # stage 1: own mailbox (fine-grained device memory) and its IPC handle
        handle = (ctypes.c_char * 64)()
        try:
            nbytes = _lib.lib.rl_peer_mailbox_bytes(self.world, self.max_n)
            own = ctypes.c_void_p()
            _lib.check(_lib.lib.rl_peer_alloc(nbytes, ctypes.byref(own)), "rl_peer_alloc")
            self._own = own
            _lib.check(_lib.lib.rl_peer_export(own, handle), "rl_peer_export")
        except RuntimeError as e:
            why = str(e)
        if not _agree(why is None):
            self._release()
            raise PeerReducer.PeerUnavailable(why or "a peer could not allocate / export its mailbox")
        # stage 2: map every peer's mailbox (across DEVICES on a multi-GPU node: hipIpcOpenMemHandle + peer access)
        handles = [None] * self.world
        dist.all_gather_object(handles, bytes(handle.raw))
        self._table = (ctypes.c_void_p * self.world)()
        try:
            for r in range(self.world):
                if r == self.rank:
                    self._table[r] = self._own.value
                else:
                    p = ctypes.c_void_p()
                    buf = ctypes.create_string_buffer(handles[r], 64)
                    _lib.check(_lib.lib.rl_peer_open(buf, ctypes.byref(p)), "rl_peer_open")
                    self._table[r] = p.value
                    self._opened.append(p)
        except RuntimeError as e:
            why = str(e)
        if not _agree(why is None):
            self._release()
            raise PeerReducer.PeerUnavailable(why or "a peer could not map a mailbox of the world")
        self.err = torch.zeros(1, dtype=torch.int32, device="cuda")
        self.seq = 0
        self.count = 0
        self._pending = None                # (async read of the error flag, reduction count it covers)
        dist.barrier()                      # nobody writes into a mailbox that is not mapped everywhere yet
        # stage 3: one reduction of known rows, bit for bit against the rank-ordered sum of the gathered rows
        ok, why = self.self_check()
        if not _agree(ok):
            self._release()
            raise PeerReducer.PeerUnavailable(why or "a peer's check reduction differed from the rank-ordered sum")
        self.count = 0                      # (`count` is the reductions of the RUN; the sequence number goes on)

    def _release(self):
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        for p in self._opened:
            self._lib.lib.rl_peer_close(p)
        self._opened = []
        if self._own is not None:
            self._lib.lib.rl_peer_free(self._own)
            self._own = None

    def self_check(self, n=1572):
        """One peer reduction of per-rank pseudo-random float64 rows against (a) the sum of the same rows gathered by
        the backend and added in rank order -- what the kernel defines -- bit for bit, and (b) for integer-valued rows
        (exact in any order) the backend's own all-reduce, bit for bit.  Local verdict + reason; callers make it
        collective with ``_agree``."""
        n = min(int(n), self.max_n)
        g = torch.Generator(device="cpu").manual_seed(1234 + self.rank)
        real = torch.randn(n, generator=g, dtype=torch.float64)
        ints = torch.randint(-(1 << 40), 1 << 40, (n,), generator=g).to(torch.float64)
        why = None
        for name, row in (("real-valued", real), ("integer-valued", ints)):
            mine = row.cuda()
            got = self.all_reduce_sum_(mine.clone())
            rows = [torch.empty_like(row) for _ in range(self.world)]
            if dist.get_backend() == "gloo":
                dist.all_gather(rows, row)
                rows = [r.cuda() for r in rows]
            else:
                rows = [torch.empty_like(mine) for _ in range(self.world)]
                dist.all_gather(rows, mine)
            want = torch.zeros_like(mine)
            for r in rows:                                   # rank order, starting from 0.0: the kernel's own order
                want = want + r
            torch.cuda.synchronize()
            if int(self.err.item()) != 0:
                why = "check reduction (%s rows) timed out waiting for rank %d" % (name, int(self.err.item()) - 1)
                break
            if not torch.equal(got, want):
                why = "check reduction (%s rows) differs from the rank-ordered sum in %d of %d entries" % (
                    name, int((got != want).sum()), n)
                break
            if name == "integer-valued":
                back = mine.clone()
                if dist.get_backend() == "gloo":
                    h = back.cpu()
                    dist.all_reduce(h)
                    back = h.cuda()
                else:
                    dist.all_reduce(back)
                if not torch.equal(got, back):
                    why = "check reduction (integer-valued rows) differs from the backend's all-reduce"
                    break
        return why is None, why

    def all_reduce_sum_(self, t):
        assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and t.numel() <= self.max_n
        self.seq += 1
        self.count += 1
        lib = self._lib
        lib.check(lib.lib.rl_peer_allreduce_sum(t.numel(), lib.ptr(t), self.rank, self.world, self._table, self.max_n,
                                                self.seq, lib.ptr(self.err), lib.peer_spin_limit(), lib.stream_ptr()),
                  "rl_peer_allreduce_sum")
        return t

    def _raise(self, e, upto):
        raise RuntimeError("peer all-reduce: rank %d never delivered a row (a reduction <= %d of this process gave up "
                           "waiting; the sums since then are not the world's)" % (e - 1, upto))

    def check(self):
        """Blocking: raises if any reduction so far gave up waiting for a peer."""
        e = int(self.err.item())
        if e:
            self._raise(e, self.seq)

    def poll(self):
        """Non-blocking form for the training loop, once per iteration: starts reading the error flag behind the
        launches queued so far and raises on what the PREVIOUS poll read (that copy finished an iteration ago), so
        a peer that stopped answering ends the run within one iteration instead of training on partial sums."""
        from rllab_amd.misc.device_io import read_async
        prev, self._pending = self._pending, (read_async(self.err), self.seq)
        if prev is not None:
            e = int(prev[0].get()[0])
            if e:
                self._raise(e, prev[1])

    def close(self):
        torch.cuda.synchronize()
        e = int(self.err.item())            # the reductions no poll has covered yet
        dist.barrier()
        self._release()
        if e:
            self._raise(e, self.seq)


def peer_reducer():
    """The process's PeerReducer when RLLAB_PEER_ALLREDUCE=1 and the run is distributed, else None (created on first
    use: every rank reaches its first sharded gradient at the same point of the same program)."""
    global _peer, _peer_refused
    if _peer is None and _peer_refused is None and os.environ.get("RLLAB_PEER_ALLREDUCE") and is_distributed() \
            and torch.cuda.is_available():
        try:
            _peer = PeerReducer()
        except PeerReducer.PeerUnavailable as e:
            # every rank is here (the constructor's stages are collective): the run stays on the backend's all-reduce
            _peer_refused = str(e)
            import sys
            sys.stderr.write("[rllab_amd] rank %d: RLLAB_PEER_ALLREDUCE=1 but the peer path is unavailable (%s); "
                             "using the %s all-reduce\n" % (dist.get_rank(), _peer_refused, dist.get_backend()))
    return _peer


def peer_status():
    """("peer" | "backend", reason): which path ``update_sum_`` takes and, when the peer path was asked for and refused,
    why (bench.py prints it)."""
    if _peer is not None:
        return "peer", None
    return "backend", _peer_refused


def peer_poll():
    """Once per training iteration (BatchPolopt.train_iteration): see PeerReducer.poll.  Free when the peer path is off."""
    if _peer is not None:
        _peer.poll()


def peer_shutdown():
    """Unmap the mailboxes (collective: every rank calls it); raises if a reduction gave up waiting for a peer."""
    global _peer, _peer_refused
    _peer_refused = None
    if _peer is not None:
        pr, _peer = _peer, None
        pr.close()


def update_sum_(t):
    """Sum over ranks of a float64 device vector on the update's critical path (gradient, Fisher-vector product):
    the in-stream peer all-reduce when enabled, the backend's all-reduce otherwise."""
    if not is_distributed():
        return t
    pr = peer_reducer()
    if pr is not None and t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and t.numel() <= pr.max_n:
        _acct["peer"] += 1                  # in-stream reductions are not host-issued collectives: counted apart
        return pr.all_reduce_sum_(t)
    return all_reduce_sum_(t)


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
