"""Multi-GPU pre-flight of the one exchange family of the sharded path: run by every rank
after ``init_process_group`` -- ``bench.py --gpus N`` does it before warm-up and prints the record in its JSON line,
``tools/preflight_multigpu.py`` is the stand-alone form -- so that the first lease of a multi-GPU node is not spent
debugging.  The reference has no counterpart (its sampler is one host process pool,
rllab/sampler/parallel_sampler.py:98-126); what is checked is this engine's own contract:

  * which device every rank sits on and the ``hipDeviceCanAccessPeer`` matrix between those devices;
  * the in-stream peer all-reduce (csrc/peer_kernels.hip) across the ranks' devices: mailboxes allocated fine-grained,
    exported with hipIpc and mapped by every peer, one reduction of known rows compared BIT FOR BIT with the rank-ordered
    sum of the rows gathered by the backend, and -- for integer-valued rows, exact in any order -- with the backend's
    (RCCL's) own all-reduce;
  * 100-call latency of both paths on a gradient-sized row (P doubles), device-timed;
  * the decision every rank takes: ``peer`` only if it was asked for (RLLAB_PEER_ALLREDUCE=1) and every stage passed
    on EVERY rank (all-reduce-min of the verdicts), else the backend -- logged, never silent.

Everything here is collective and cannot leave a rank behind: a stage that fails on one rank is agreed on by all
(`dist._agree`) before anybody moves on.
"""
import os
import sys

import torch
import torch.distributed as dist

from rllab_amd.sampler import dist as D


def _gather_obj(x):
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, x)
    return out


def _time_calls(fn, calls):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(calls):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / calls          # microseconds per call


def run(n=1572, calls=100, log=True):
    """Collective.  Returns the record (identical decision on every rank; latencies are this rank's)."""
    assert dist.is_initialized() and torch.cuda.is_available()
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.cuda.current_device()
    devices = _gather_obj(dev)
    n_dev = torch.cuda.device_count()
    access = [[(i == j) or bool(torch.cuda.can_device_access_peer(i, j)) for j in range(n_dev)] for i in range(n_dev)]
    rec = {
        "world": world, "backend": dist.get_backend(), "device_of_rank": devices, "devices_on_node": n_dev,
        "distinct_devices": len(set(devices)) == world,
        "peer_access_matrix": access,
        "peer_access_between_ranks": all(access[a][b] for a in devices for b in devices),
        "row_doubles": int(n), "calls": int(calls),
    }
    asked = bool(os.environ.get("RLLAB_PEER_ALLREDUCE"))
    # -- the peer path: built exactly as the product builds it (its constructor IS the staged, collective check) ------
    peer, why = None, None
    existing = D._peer
    try:
        peer = existing if existing is not None else D.PeerReducer()
    except D.PeerReducer.PeerUnavailable as e:
        why = str(e)
    rec["peer_check"] = {"passed": peer is not None, "reason": why,
                         "what": "mailboxes (fine-grained, hipIpc) mapped by every rank; one reduction of real-valued and "
                                 "of integer-valued rows == the rank-ordered sum of the gathered rows, bit for bit; the "
                                 "integer-valued one == the backend's all-reduce, bit for bit"}
    # -- latency of both paths on a gradient-sized float64 row (device-timed; the backend's through the host for gloo) --
    row = torch.randn(n, dtype=torch.float64, device="cuda")
    via_host = dist.get_backend() == "gloo"

    def backend_call():
        if via_host:
            h = row.cpu()
            dist.all_reduce(h)
            row.copy_(h)
        else:
            dist.all_reduce(row)
        row.mul_(1.0 / world)           # (keeps the values bounded over 100 calls)
    dist.barrier()
    rec["backend_allreduce_us"] = _time_calls(backend_call, calls)
    rec["peer_allreduce_us"] = None
    if peer is not None:
        def peer_call():
            peer.all_reduce_sum_(row)
            row.mul_(1.0 / world)
        dist.barrier()
        rec["peer_allreduce_us"] = _time_calls(peer_call, calls)
        ok = int(peer.err.item()) == 0
        if not D._agree(ok):
            rec["peer_check"].update(passed=False, reason="a reduction of the latency loop timed out waiting for a peer")
            if existing is None:
                peer._release()
            peer = None
    if peer is not None and existing is None:
        peer.close()                      # the product creates its own on first use (same stages, same verdict)
    passed = peer is not None
    rec["decision"] = "peer" if (asked and passed) else "backend"
    rec["peer_requested"] = asked
    if asked and not passed:
        rec["fallback"] = "RLLAB_PEER_ALLREDUCE=1 but the peer check failed (%s): every rank stays on the %s all-reduce" % (
            rec["peer_check"]["reason"], dist.get_backend())
    if log and rank == 0:
        sys.stderr.write("[preflight] world %d on devices %s (distinct: %s), peer access between ranks: %s, peer check: %s%s, "
                         "backend all-reduce %.1f us, peer all-reduce %s us -> %s\n" % (
                             world, devices, rec["distinct_devices"], rec["peer_access_between_ranks"],
                             "passed" if passed else "FAILED", "" if passed else " (%s)" % rec["peer_check"]["reason"],
                             rec["backend_allreduce_us"],
                             "%.1f" % rec["peer_allreduce_us"] if rec["peer_allreduce_us"] is not None else "n/a",
                             rec["decision"]))
    return rec


def run_isolated(n=1572, calls=100, timeout=240.0):
    """The same check in CHILD processes (one per rank, their own process group on MASTER_PORT + 17), so that nothing
    the peer path can do on a topology it has never seen -- a fault on a cross-device mapping, a hang -- can take the
    caller's run with it.  Collective over the caller's group: every rank spawns `tools/preflight_multigpu.py` with its
    own RANK / LOCAL_RANK, waits (bounded), and the ranks agree on the outcome.  A failed or timed-out pre-flight clears
    RLLAB_PEER_ALLREDUCE in EVERY rank's environment (logged): the run then stays on the backend's all-reduce."""
    import json
    import subprocess
    rank, world = dist.get_rank(), dist.get_world_size()
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ)
    env["MASTER_ADDR"] = os.environ.get("MASTER_ADDR", "127.0.0.1")
    env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29533")) + 17)
    env["RANK"], env["WORLD_SIZE"] = str(rank), str(world)
    env["LOCAL_RANK"] = os.environ.get("LOCAL_RANK", str(rank))
    env.pop("RLLAB_DIST_FORCE", None)
    for k in list(env):
        if k.startswith("TORCHELASTIC_") or k in ("GROUP_RANK", "ROLE_RANK", "ROLE_WORLD_SIZE", "LOCAL_WORLD_SIZE"):
            env.pop(k)
    cmd = [sys.executable, os.path.join(root, "tools", "preflight_multigpu.py"), "--gpus", str(world), "--row", str(int(n)),
           "--calls", str(int(calls))]
    rec, why = None, None
    try:
        p = subprocess.run(cmd, env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout,
                           universal_newlines=True)
        if p.returncode != 0:
            why = "pre-flight child of rank %d exited with %d: %s" % (rank, p.returncode, p.stderr.strip()[-300:])
        elif rank == 0:
            lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
            rec = json.loads(lines[-1]) if lines else None
            if rec is None:
                why = "pre-flight child of rank 0 printed no record"
    except subprocess.TimeoutExpired:
        why = "pre-flight child of rank %d did not finish within %.0f s" % (rank, timeout)
    ran = D._agree(why is None)
    recs = _gather_obj((rec, why))
    rec = next((r for r, _ in recs if r is not None), None)
    whys = [w for _, w in recs if w]
    if not ran or rec is None:
        rec = {"world": world, "backend": dist.get_backend(), "decision": "backend",
               "peer_check": {"passed": False, "reason": "; ".join(whys) or "no record"}, "isolated": True}
    rec["isolated"] = True
    if rec.get("decision") != "peer" and os.environ.get("RLLAB_PEER_ALLREDUCE"):
        os.environ.pop("RLLAB_PEER_ALLREDUCE")
        if rank == 0:
            sys.stderr.write("[preflight] RLLAB_PEER_ALLREDUCE=1 withdrawn on every rank: %s\n"
                             % (rec.get("fallback") or rec["peer_check"].get("reason")))
    return rec
