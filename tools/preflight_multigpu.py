#!/usr/bin/env python
"""tools/preflight_multigpu.py -- the multi-GPU pre-flight of rllab_amd/sampler/preflight.py as a command.

    python tools/preflight_multigpu.py --gpus 8            # one rank per GPU of this node over RCCL (self-launch)
    RLLAB_DIST_BACKEND=gloo python tools/preflight_multigpu.py --gpus 2     # test mode: ranks may share a device

Prints ONE JSON line on rank 0: device of every rank, hipDeviceCanAccessPeer matrix, the verdict of the staged peer
all-reduce check (fine-grained mailboxes, hipIpc mapping across the ranks' devices, one reduction bit for bit against the
rank-ordered sum of the gathered rows and against the backend's all-reduce), 100-call latency of both paths, and the
path every rank will take (all-reduce-min of the verdicts).  `bench.py --gpus N` runs the same check before warm-up.
Takes seconds; spend them first on a fresh multi-GPU lease.  profiles/r0N_preflight_*.json keeps the records."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=2)
    ap.add_argument("--calls", type=int, default=100)
    ap.add_argument("--row", type=int, default=1572, help="doubles per reduced row (default: the headline policy's P)")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    if "WORLD_SIZE" not in os.environ:
        import bench
        cmd = bench.self_launch_argv(args.gpus, sys.argv[1:])
        cmd[cmd.index(os.path.join(ROOT, "bench.py"))] = os.path.abspath(__file__)
        sys.stderr.write("[preflight] launching %d ranks: %s\n" % (args.gpus, " ".join(cmd)))
        sys.stderr.flush()
        os.execv(cmd[0], cmd)
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "the pre-flight needs HIP devices"
    backend = os.environ.get("RLLAB_DIST_BACKEND", "nccl")
    if backend == "nccl" and world > torch.cuda.device_count():
        sys.exit("preflight: %d ranks over RCCL on a node with %d GPUs" % (world, torch.cuda.device_count()))
    dev = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev)
    kw = dict(device_id=torch.device("cuda", dev)) if backend == "nccl" else {}
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    dist.barrier()
    from rllab_amd.sampler import preflight
    rec = preflight.run(n=args.row, calls=args.calls)
    lat = [None] * world
    dist.all_gather_object(lat, (rec["backend_allreduce_us"], rec["peer_allreduce_us"]))
    rec["per_rank_latency_us"] = [{"backend": b, "peer": p} for b, p in lat]
    import ctypes
    ctypes.CDLL(None).fflush(None)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(rec))
        sys.stdout.flush()
    os._exit(0)


if __name__ == "__main__":
    main()
