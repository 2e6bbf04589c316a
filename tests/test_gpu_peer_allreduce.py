"""rl_peer_allreduce_sum at the C ABI, inside ONE process: the ranks of a world are launches on separate streams over
mailboxes of this process (the flag protocol does not care who owns a mailbox); and what happens when a peer never
arrives -- the bounded spin, the sticky error word, PeerReducer.poll / close raising.

The multi-process forms (hipIpc mappings, 2 and 4 ranks on one GPU, bit-identical parameters) are in
tests/test_gpu_two_rank.py; SURVEY.md section 8e is the contract ("one-shot P2P write + fixed-order local sum")."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mailboxes(lib, world, max_n):
    nbytes = lib.lib.rl_peer_mailbox_bytes(world, max_n)
    assert nbytes == 2 * 8 * 8 + 2 * world * max_n * 8
    boxes = []
    for _ in range(world):
        p = ctypes.c_void_p()
        lib.check(lib.lib.rl_peer_alloc(nbytes, ctypes.byref(p)), "rl_peer_alloc")
        boxes.append(p)
    table = (ctypes.c_void_p * world)(*[b.value for b in boxes])
    return boxes, table


def _free(lib, boxes):
    torch.cuda.synchronize()
    for b in boxes:
        lib.check(lib.lib.rl_peer_free(b), "rl_peer_free")


# (worlds of 2 and 3 only: the launches of a world must be RESIDENT together, and streams of one process share four
# hardware queues by default -- four or eight ranks are the multi-process tests' business)
@pytest.mark.parametrize("world,n", [(2, 1572), (2, 65536), (3, 1), (3, 5900)])
def test_ranks_as_streams_sum_in_rank_order(world, n, monkeypatch):
    from rllab_amd import _lib
    monkeypatch.setenv("RLLAB_PEER_SPIN_LIMIT", "4000000")   # a scheduling surprise fails in about a second
    max_n = 1 << 16
    boxes, table = _mailboxes(_lib, world, max_n)
    rng = np.random.RandomState(world * 1000 + n)
    err = torch.zeros(1, dtype=torch.int32, device="cuda")
    # The launches of a world must be RESIDENT together.  HIP maps a process's streams onto a few hardware queues in
    # creation order; which queue a new stream gets depends on how many streams the tests before this one created, and
    # two ranks on ONE queue serialise (rank 0 then waits for a peer that cannot start).  Probe for a set of streams
    # whose kernels do overlap -- a tiny two-way reduction per candidate set -- before the real runs.
    streams, keep = None, []
    for attempt in range(6):
        cand = [torch.cuda.Stream() for _ in range(world)]
        keep += cand
        err.zero_()
        probe = [torch.ones(1, dtype=torch.float64, device="cuda") for _ in range(world)]
        torch.cuda.synchronize()
        monkeypatch.setenv("RLLAB_PEER_SPIN_LIMIT", "200000")
        for r in range(world):
            with torch.cuda.stream(cand[r]):
                _lib.check(_lib.lib.rl_peer_allreduce_sum(1, _lib.ptr(probe[r]), r, world, table, max_n, 100 + attempt,
                                                          _lib.ptr(err), _lib.peer_spin_limit(), _lib.stream_ptr()),
                           "rl_peer_allreduce_sum")
        torch.cuda.synchronize()
        if int(err.item()) == 0 and all(float(p_[0]) == world for p_ in probe):
            streams = cand
            break
        keep.append(torch.cuda.Stream())                     # shift the round-robin by one and try again
    assert streams is not None, "no set of %d streams of this process runs concurrently" % world
    monkeypatch.setenv("RLLAB_PEER_SPIN_LIMIT", "4000000")
    err.zero_()
    try:
        for seq in range(200, 205):                          # both slots, several times over
            rows = rng.randn(world, n) * 10.0 ** rng.randint(-3, 4, size=(world, 1))
            xs = [torch.as_tensor(rows[r], device="cuda") for r in range(world)]
            torch.cuda.synchronize()
            for r in range(world):
                with torch.cuda.stream(streams[r]):
                    _lib.check(_lib.lib.rl_peer_allreduce_sum(n, _lib.ptr(xs[r]), r, world, table, max_n, seq,
                                                              _lib.ptr(err), _lib.peer_spin_limit(), _lib.stream_ptr()),
                               "rl_peer_allreduce_sum")
            torch.cuda.synchronize()
            want = np.zeros(n)
            for r in range(world):                           # the kernel's order: rank 0 first
                want = want + rows[r]
            for r in range(world):
                assert np.array_equal(xs[r].cpu().numpy(), want), (seq, r)
        assert int(err.item()) == 0
    finally:
        _free(_lib, boxes)


def test_a_peer_that_never_arrives_sets_the_error_word(monkeypatch):
    from rllab_amd import _lib
    monkeypatch.setenv("RLLAB_PEER_SPIN_LIMIT", "2000")      # milliseconds instead of seconds
    world, max_n, n = 2, 1024, 8
    boxes, table = _mailboxes(_lib, world, max_n)
    err = torch.zeros(1, dtype=torch.int32, device="cuda")
    x = torch.arange(n, dtype=torch.float64, device="cuda")
    try:
        _lib.check(_lib.lib.rl_peer_allreduce_sum(n, _lib.ptr(x), 0, world, table, max_n, 1, _lib.ptr(err),
                                                  _lib.peer_spin_limit(), _lib.stream_ptr()), "rl_peer_allreduce_sum")
        torch.cuda.synchronize()                             # the launch COMPLETES: nothing hangs
        assert int(err.item()) == 1 + 1                      # 1 + the rank whose flag never came
        # and its output is poisoned, not a sum over stale rows: CG / the line search reject a NaN step
        assert bool(torch.isnan(x).all())
        x = torch.arange(n, dtype=torch.float64, device="cuda")
        # sticky: a later, complete reduction does not clear it
        y = torch.ones(n, dtype=torch.float64, device="cuda")
        s1 = torch.cuda.Stream()
        with torch.cuda.stream(s1):
            _lib.check(_lib.lib.rl_peer_allreduce_sum(n, _lib.ptr(y), 1, world, table, max_n, 2, _lib.ptr(err),
                                                      _lib.peer_spin_limit(), _lib.stream_ptr()), "rl_peer_allreduce_sum")
        _lib.check(_lib.lib.rl_peer_allreduce_sum(n, _lib.ptr(x), 0, world, table, max_n, 2, _lib.ptr(err),
                                                  _lib.peer_spin_limit(), _lib.stream_ptr()), "rl_peer_allreduce_sum")
        torch.cuda.synchronize()
        assert int(err.item()) != 0
    finally:
        _free(_lib, boxes)


def test_poll_raises_one_iteration_later_and_close_raises_at_the_end():
    """PeerReducer.poll is what BatchPolopt.train_iteration calls: non-blocking, it reports what the previous poll read."""
    from rllab_amd.sampler import dist as D

    class Stub(D.PeerReducer):
        def __init__(self):                                  # no world: only the error word and the bookkeeping
            self.err = torch.zeros(1, dtype=torch.int32, device="cuda")
            self.seq, self.count, self._pending = 0, 0, None

    pr = Stub()
    pr.poll()
    pr.poll()                                                # clean so far
    pr.seq = 7
    pr.err.fill_(3)                                          # a reduction gave up on rank 2
    pr.poll()                                                # starts the read that sees it ...
    with pytest.raises(RuntimeError, match="rank 2 never delivered"):
        pr.poll()                                            # ... and the next poll raises
    with pytest.raises(RuntimeError, match="rank 2 never delivered"):
        pr.check()
