#!/usr/bin/env python
"""bench.py -- the headline measurement (BASELINE.json): env steps/s and TRPO iteration
time of the batched rollout + TRPO update at 4096 Swimmer-style envs per MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: either pre-launched as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`,
   or plainly -- bench.py then re-executes itself under torch.distributed.run with N ranks on 127.0.0.1)

One "step" = one full TRPO iteration of BatchPolopt.train (sampler.obtain_samples ->
process_samples -> log_diagnostics -> optimize_policy) on a batch of
4096 envs x 500 steps per GPU, synthetic data (random-init GaussianMLPPolicy(32,32),
reset states from the in-kernel Philox stream).  ``value`` = env steps of ALL ranks
/ wall time of the K timed iterations (nothing skipped: the update runs every
iteration), max over ranks, bracketed by barrier + synchronize.

Extra objects on the JSON line:
  roofline     -- the dominant kernel (fused rollout): algorithmic bytes per launch
                  (145 B step + 72 B trajectory record per env-step)
                  / average launch duration measured live with HIP events on the
                  launch stream.  The kernel fuses 50 physics sub-steps per env-step in
                  registers and is VALU/latency bound, so the HBM fraction is tiny by
                  construction; `valu_tflops` gives the other axis.
  roofline_scan -- the HBM-bound kernels of process_samples the north star names (discounted-return / GAE scan, path
                  scan + baseline prediction, the baseline's normal equations): SURVEY.md 8d algorithmic bytes per sample
                  x the batch / HIP-event time of the launch, live, against 8 TB/s.
  roofline_step_kernel -- rl_vecenv_step (the per-step boundary kernel, state through HBM every
                  launch) at 4 M envs for Cartpole and for the workload's env: the HBM-bound
                  regime of the step kernel, timed live.
  cpu_baseline -- kind "reference": the reference's UNMODIFIED parallel_sampler / stateful_pool /
                  rollout / NormalizedEnv (staged byte for byte under oracle/_ref by
                  oracle/make_ref.py, driven by oracle/ref_sampler.py) on all host cores and
                  on one, timed on a bounded sample in the same run (rank 0, N == 1 only); the
                  re-typed port (oracle/cpu_sampler.py) rides along as a cross-check.
  peer_reductions_per_iter -- with RLLAB_PEER_ALLREDUCE=1: sums of the gradient / Fisher-vector products done by the
                  in-stream peer all-reduce (csrc/peer_kernels.hip) instead of host-issued collectives.
  ranks / backend / collectives_per_iter / collective_ms_per_iter -- what torch.distributed
                  actually saw (N > 1), the collectives one iteration issues and their
                  host-bracketed cost measured on extra iterations after the timed region.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (env module, env class, horizon T, hidden, algo, gae_lambda, algorithmic bytes / env-step)
    "swimmer4096_trpo": dict(env="swimmer", n_envs=4096, T=500, hidden=(32, 32), algo="trpo", lam=1.0,
                             step_bytes=145, record_bytes=72),
    "cartpole4096_vpg": dict(env="cartpole", n_envs=4096, T=100, hidden=(32, 32), algo="vpg", lam=1.0,
                             step_bytes=153, record_bytes=28),
    # parity-config side lines (not the headline): DoublePendulum, and BASELINE config C5's per-GPU shard
    # (8192 envs / 8 GPUs = 1024 envs per GPU, GaussianMLPPolicy(64,64), TRPO + GAE lambda 0.97)
    "double_pendulum4096_trpo": dict(env="double_pendulum", n_envs=4096, T=100, hidden=(32, 32), algo="trpo",
                                     lam=1.0, step_bytes=4 * (2 * 17 + 1 + 6 + 1) + 1, record_bytes=36),
    "cheetah1024_trpo_gae": dict(env="half_cheetah", n_envs=1024, T=500, hidden=(64, 64), algo="trpo", lam=0.97,
                                 step_bytes=253, record_bytes=132),
}
ENVS = {  # name -> (module, class, rl_env_kind)
    "cartpole": ("rllab_amd.envs.box2d.cartpole_env", "CartpoleEnv", 0),
    "double_pendulum": ("rllab_amd.envs.box2d.double_pendulum_env", "DoublePendulumEnv", 1),
    "swimmer": ("rllab_amd.envs.mujoco.swimmer_env", "SwimmerEnv", 2),
    "half_cheetah": ("rllab_amd.envs.mujoco.half_cheetah_env", "HalfCheetahEnv", 3),
}


def step_kernel_roofline(torch, kind, n=1 << 22, steps=20, warmup=3):
    from rllab_amd import _lib
    from rllab_amd.envs.hip_env import HipVecEnv
    name = {v[2]: k for k, v in ENVS.items()}.get(kind, str(kind))
    v = HipVecEnv(kind, n, 0, normalize=True, seed=1)
    q = v.q
    v.reset()
    act = (torch.rand((q["act_dim"], n), device=v.device) * 2 - 1).contiguous()

    def launch():
        _lib.check(_lib.lib.rl_vecenv_step(
            kind, n, 1, 1.0, 0, 1, _lib.ptr(v.state), _lib.ptr(v.ts), _lib.ptr(act), None, v.seed, v.step_counter, 0,
            None, _lib.ptr(v._obs), _lib.ptr(v._reward), _lib.ptr(v._done), _lib.stream_ptr()), "rl_vecenv_step")
        v.step_counter += 1
    for _ in range(warmup):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(steps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    step_bytes = 4 * (2 * q["state_dim"] + q["act_dim"] + q["obs_dim"] + 1) + 1 + 8
    gbs = step_bytes * n / (ms * 1e-3) / 1e9
    del v, act
    torch.cuda.empty_cache()
    out = {"kernel": "vecenv_step_kernel<%s> (rl_vecenv_step, one transition per launch)" % name, "n_envs": n,
           "bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0,
           "avg_launch_ms": ms, "bytes_per_env_step": step_bytes, "env_steps_per_s": n / (ms * 1e-3)}
    # the vector axis: instructions per launch from the kernel's own counters (a builder-run pass of tools/
    # step_kernel_roofline.py under --pmc, profiles/run_profile.sh; stamped with the kernel-source hash) x 64 lanes over
    # THIS run's launch time, against the vector peak in lane-instructions (157.3 TFLOP/s counts an FMA twice)
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    cls = {"cartpole": "Cartpole", "double_pendulum": "DoublePendulum", "swimmer": "Swimmer",
           "half_cheetah": "HalfCheetah"}.get(name)
    rec = json.load(open(tpath)).get("step_kernels") if os.path.exists(tpath) else None
    if rec and cls in rec.get("kernels", {}):
        k = rec["kernels"][cls]
        if rec.get("kernel_source_hash") != kernel_source_hash():
            out["valu_source"] = "stale: the step kernels' counters were taken from kernel sources %s, this build is %s" % (
                rec.get("kernel_source_hash"), kernel_source_hash())
        elif k["n_envs"] == n:
            tfl = k["insts_valu"] * 64.0 / (ms * 1e-3) / 1e12
            out.update({"valu_tflops": tfl, "valu_peak_tflops": 157.3 / 2.0, "valu_frac": tfl / (157.3 / 2.0),
                        "valu_unit": "T lane-instructions/s (SQ_INSTS_VALU x 64 / launch time; an FMA counts once)",
                        "valu_insts_per_env_step": k["insts_valu"] * 64.0 / n,
                        "valu_issue_frac": k["active_inst_valu"] / k["wave_cycles"] if k.get("wave_cycles") else None,
                        "valu_source": rec["source"] + "; NOT measured in this run -- only the launch time is"})
    return out


def scan_rooflines(torch, algo, traj, reps=20):
    """The three scans of process_samples on the batch just sampled, each timed alone with HIP events on torch's current
    stream (the stream the launches go to: _lib.stream_ptr()); bytes per sample = what the algorithm has to move once."""
    from rllab_amd import _lib
    from rllab_amd.sampler.base import path_scan
    T, N, do = traj.T, traj.N, traj.obs_dim
    B = T * N
    dev = traj.device
    base = algo.baseline
    coeffs = base.dense_coeffs() if hasattr(base, "dense_coeffs") else None
    if coeffs is None:
        coeffs = torch.zeros(2 * do + 4, dtype=torch.float64, device=dev)
    coeffs = torch.as_tensor(coeffs, dtype=torch.float64, device=dev).contiguous()
    tin, valid, values = path_scan(traj, True, coeffs)
    adv = torch.empty((T, N), dtype=torch.float32, device=dev)
    ret, und = torch.empty_like(adv), torch.empty_like(adv)
    gamma, lam = float(algo.discount), float(algo.gae_lambda)

    def gae():
        _lib.check(_lib.lib.rl_gae(T, N, _lib.ptr(traj.rewards), _lib.ptr(values), _lib.ptr(traj.dones), gamma, lam,
                                   _lib.ptr(adv), _lib.ptr(ret), _lib.ptr(und), _lib.stream_ptr()), "rl_gae")

    def scan():
        path_scan(traj, True, coeffs)
    out = []

    def timed(name, fn, bytes_per_sample, what):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        gbs = bytes_per_sample * B / (ms * 1e-3) / 1e9
        out.append({"kernel": name, "bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0,
                    "avg_launch_ms": ms, "bytes_per_sample": bytes_per_sample, "samples": B, "bytes": what})
    timed("gae_reg_kernel (rl_gae: advantages, discounted and undiscounted returns, one segmented reverse scan)", gae, 25,
          "reward f32 + baseline f64 + done u8 in, three f32 planes out")
    timed("path_scan_reg_kernel (rl_path_scan: step-in-path index, whole-path validity, baseline prediction)", scan,
          4 * do + 14, "done u8 + %d observation planes f32 in, index i32 + valid u8 + prediction f64 out" % do)
    if hasattr(base, "normal_eq_dense"):
        keep = (traj.valid, traj.tin, traj.returns)
        traj.valid, traj.tin, traj.returns = valid, tin, ret
        gae()
        timed("lfb_normal_eq kernel (rl_lfb_normal_eq: Phi^T Phi and Phi^T y in float64, features rebuilt per tile)",
              lambda: base.normal_eq_dense(traj), 4 * do + 9,
              "%d observation planes + return f32 + index i32 + valid u8 in; the (2 Do + 4)^2 sums out" % do)
        traj.valid, traj.tin, traj.returns = keep
    return out


def self_launch_argv(n_gpus, argv, port=None):
    """The command a plain `python bench.py --gpus N` turns itself into: one rank per GPU of this node
    under torch.distributed.run, rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    if port is None:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def kernel_source_hash():
    """sha256 over the HIP sources the library is built from: stamps profiles/pmc_traffic.json so a counter
    reading can never be quoted for a kernel it was not taken from."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "rllab_amd", "csrc", "*.h*"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="swimmer4096_trpo", choices=sorted(WORKLOADS))
    ap.add_argument("--n-envs", type=int, default=None, help="envs per GPU (default: the workload's 4096)")
    ap.add_argument("--hidden", default=None, help="policy hidden sizes, e.g. 100,50,25 (side lines; default: the "
                                                   "workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=15.0, help="seconds of CPU-port sampling")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # (RLLAB_DIST_BACKEND=gloo is the test mode in which ranks may share a device)
        if args.gpus > torch.cuda.device_count() and os.environ.get("RLLAB_DIST_BACKEND", "nccl") == "nccl":
            sys.exit("bench.py --gpus %d: this node has %d GPUs" % (args.gpus, torch.cuda.device_count()))
        cmd = self_launch_argv(args.gpus, sys.argv[1:])
        sys.stderr.write("[bench] launching %d ranks: %s\n" % (args.gpus, " ".join(cmd)))
        sys.stderr.flush()
        os.execv(cmd[0], cmd)
    if args.gpus != world:
        sys.exit("bench.py --gpus %d under WORLD_SIZE=%d" % (args.gpus, world))
    device_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    if world > 1 or os.environ.get("RLLAB_DIST_FORCE"):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        backend = os.environ.get("RLLAB_DIST_BACKEND", "nccl")   # nccl == RCCL over xGMI; gloo only for tests
        kw = dict(device_id=torch.device("cuda", device_index)) if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)

    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    if world > 1:
        dist.barrier()

    from rllab_amd.algos.trpo import TRPO
    from rllab_amd.algos.vpg import VPG
    from rllab_amd.baselines.linear_feature_baseline import LinearFeatureBaseline
    from rllab_amd.envs.normalized_env import normalize
    from rllab_amd.misc import ext, logger
    from rllab_amd.policies.gaussian_mlp_policy import GaussianMLPPolicy
    from rllab_amd.sampler import dist as D

    wl = dict(WORKLOADS[args.workload])
    if args.hidden:
        wl["hidden"] = tuple(int(h) for h in args.hidden.split(","))
    n_envs = args.n_envs or wl["n_envs"]
    T = wl["T"]
    ext.set_seed(1)
    logger.set_quiet(True)
    import importlib
    mod, cls, env_kind = ENVS[wl["env"]]
    EnvCls = getattr(importlib.import_module(mod), cls)
    env = normalize(EnvCls())
    policy = GaussianMLPPolicy(env_spec=env.spec, hidden_sizes=wl["hidden"])
    # identical theta on every rank: BatchPolopt.start_worker broadcasts rank 0's (algos/batch_polopt.py)
    baseline = LinearFeatureBaseline(env_spec=env.spec)
    common = dict(env=env, policy=policy, baseline=baseline, batch_size=n_envs * T, max_path_length=T,
                  n_itr=10 ** 9, discount=0.99, gae_lambda=wl["lam"], sampler_args=dict(n_envs=n_envs))
    algo = TRPO(step_size=0.01, **common) if wl["algo"] == "trpo" else VPG(**common)
    preflight_rec = None
    if world > 1:
        # multi-GPU pre-flight (rllab_amd/sampler/preflight.py; seconds): devices, peer-access matrix, the in-stream
        # peer all-reduce across the ranks' devices bit for bit against the backend, 100-call latency of both, and the
        # path every rank takes (all-reduce-min of the verdicts) -- before any warm-up, reported in the JSON line.
        # In child processes: a topology the peer path has never seen cannot take the measurement with it.
        from rllab_amd.sampler import preflight
        preflight_rec = preflight.run_isolated(n=policy.flat_params.numel())
    algo.start_worker()
    algo.init_opt()

    ev = lambda: torch.cuda.Event(enable_timing=True)
    last = {}

    timed_events = []
    host_stamps = []   # host clock at the phase boundaries (enqueue side), for RLLAB_BENCH_HOSTTIMES=1

    pending = {}

    def launch_rollout(itr):
        e0, e1 = ev(), ev()
        e0.record()
        paths = algo.sampler.obtain_samples(itr)        # one asynchronous launch
        e1.record()
        return e0, e1, paths

    def iteration(itr, timed, prefetch_next):
        # events only: per-phase times are read after the timed region, so the loop carries no
        # measurement synchronisation of its own.  As in BatchPolopt.train_iteration, the NEXT iteration's rollout
        # is enqueued right after the update, before the host writes this iteration's log -- except across the
        # boundaries of the timed region, which therefore contains exactly K rollouts, K process_samples, K updates.
        h = [time.perf_counter()]
        e = [None, None, ev(), ev()]
        h.append(time.perf_counter())
        e[0], e[1], paths = (pending.pop(itr) if itr in pending else launch_rollout(itr))[:3]
        h.append(time.perf_counter())
        algo._update_follows = True       # (as train_iteration: process_samples may start the update's first pass)
        try:
            samples = algo.sampler.process_samples(itr, paths)
        finally:
            algo._update_follows = False
        algo.log_diagnostics(paths)
        h.append(time.perf_counter())
        e[2].record()
        # an optimizer that decides its line search on the device calls this hook when the whole update is enqueued and
        # before it reads the outcome (BatchPolopt.train_iteration wires sampler.prefetch the same way): the update's
        # end event and the next rollout go in there; a rollout queued at parameters that moved on afterwards (no
        # candidate among the device-decided ones accepted, or the step rejected) is thrown away and redone
        version = getattr(algo.policy, "param_version", lambda: None)

        def queue_next():
            e[3].record()
            pending[itr + 1] = launch_rollout(itr + 1) + (version(),)
        algo._after_update_enqueued = queue_next if prefetch_next else None
        try:
            algo.optimize_policy(itr, samples)
        finally:
            algo._after_update_enqueued = None
        h.append(time.perf_counter())
        nxt = pending.get(itr + 1)
        if nxt is None or nxt[3] is None or nxt[3] != version():
            e[3] = ev()
            e[3].record()
            if prefetch_next:
                pending[itr + 1] = launch_rollout(itr + 1) + (version(),)
        last["samples"] = samples
        logger.dump_tabular()
        h.append(time.perf_counter())
        if timed:
            timed_events.append((e, getattr(algo.optimizer, "last_backtrack_iters", None)))
            host_stamps.append(h)

    itr_base = [0]

    def timed_run(warmup, steps):
        """``warmup`` untimed + ``steps`` timed iterations, bracketed by barrier + synchronize on both sides; returns the
        run's record (max-over-ranks wall time, phase times of THIS rank and of every rank, collectives)."""
        del timed_events[:], host_stamps[:]
        pending.clear()
        base = itr_base[0]
        for w in range(warmup):
            iteration(base + w, False, w + 1 < warmup)
        # a generation-2 pass of Python's cyclic GC walks every object torch has created (~35 ms here)
        # and would land inside one timed iteration: collect now and freeze the survivors
        import gc
        gc.collect()
        gc.freeze()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        D.reset_accounting(timing=False)
        t0 = time.perf_counter()
        for k in range(steps):
            iteration(base + warmup + k, True, k + 1 < steps)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed_ = time.perf_counter() - t0
        acct = D.accounting()
        rec = dict(collectives_per_iter=acct["count"] / float(steps),
                   peer_reductions_per_iter=acct.get("peer", 0) / float(steps),   # in-stream rl_peer_allreduce_sum launches
                   collective_bytes_per_iter=acct["bytes"] / float(steps), collective_ms_per_iter=None,
                   update_sum_path=D.peer_status()[0] if dist.is_initialized() else None)
        extra = 0
        if D.is_distributed():
            # a few extra iterations with every collective bracketed by a device synchronise (host clock):
            # what the exchange family costs per iteration, outside the timed region
            D.reset_accounting(timing=True)
            extra = 3
            for k in range(extra):
                iteration(base + warmup + steps + k, False, False)
            torch.cuda.synchronize()
            rec["collective_ms_per_iter"] = D.accounting()["seconds"] * 1e3 / extra
            D.reset_accounting(timing=False)
        itr_base[0] = base + warmup + steps + extra
        ph = dict(sample=0.0, process=0.0, update=0.0)
        rec["rollout_ms"], rec["per_iter"] = [], []
        for e, backtracks in timed_events:
            rec["rollout_ms"].append(e[0].elapsed_time(e[1]))
            ph["sample"] += e[0].elapsed_time(e[1])
            ph["process"] += e[1].elapsed_time(e[2])
            ph["update"] += e[2].elapsed_time(e[3])
            rec["per_iter"].append((round(e[2].elapsed_time(e[3]), 3), backtracks))
        rec["phase_ms"] = {k: v / steps for k, v in ph.items()}
        # every rank's wall time and phase times in ONE gather: the line carries the per-rank table and the spread
        # (max - min) of each column -- a straggling rank or a slow link shows here, not only in the max
        mine = torch.tensor([elapsed_ / steps * 1e3, rec["phase_ms"]["sample"], rec["phase_ms"]["process"],
                             rec["phase_ms"]["update"]], dtype=torch.float64, device="cuda")
        rows = D.all_gather_rows(mine).cpu().numpy()
        rec["elapsed"] = float(rows[:, 0].max()) * steps / 1e3          # the MAX over ranks, as the contract says
        cols = ("iteration", "sample", "process", "update")
        rec["per_rank_ms"] = [dict(rank=r, **{c: round(float(rows[r, j]), 4) for j, c in enumerate(cols)})
                              for r in range(rows.shape[0])]
        rec["rank_skew_ms"] = {c: round(float(rows[:, j].max() - rows[:, j].min()), 4) for j, c in enumerate(cols)}
        return rec

    run = timed_run(args.warmup, args.steps)
    # N > 1: the SAME invocation then times the other reduction path of the update's sums (gradient, Fisher-vector
    # products) -- the in-stream peer all-reduce over hipIpc mailboxes (csrc/peer_kernels.hip) against the backend's
    # all-reduce (RCCL) -- so that one lease of an 8-GPU node answers which is faster.  The headline fields are the
    # backend's unless RLLAB_PEER_ALLREDUCE=1 asked for the peer path from the start; the other path's record rides
    # along as `other_sum_path`.  (Skipped when the pre-flight refused the peer path, or with RLLAB_BENCH_ONE_PATH=1.)
    other = None
    if world > 1 and not os.environ.get("RLLAB_BENCH_ONE_PATH") and run["update_sum_path"] == "backend" \
            and not os.environ.get("RLLAB_PEER_ALLREDUCE"):
        os.environ["RLLAB_PEER_ALLREDUCE"] = "1"
        try:
            D.peer_reducer()                       # collective constructor: every rank is here
            if D.peer_status()[0] == "peer":
                o = timed_run(1, args.steps)
                other = dict(update_sum_path="peer", ms_per_step=o["elapsed"] / args.steps * 1e3,
                             value=world * n_envs * T * args.steps / o["elapsed"], phase_ms=o["phase_ms"],
                             collectives_per_iter=o["collectives_per_iter"],
                             peer_reductions_per_iter=o["peer_reductions_per_iter"],
                             collective_ms_per_iter=o["collective_ms_per_iter"], per_rank_ms=o["per_rank_ms"],
                             rank_skew_ms=o["rank_skew_ms"])
            else:
                other = dict(update_sum_path="peer", refused=D.peer_status()[1])
        finally:
            D.peer_shutdown()
            os.environ.pop("RLLAB_PEER_ALLREDUCE", None)
    elapsed, phase_ms, rollout_ms, per_iter = run["elapsed"], run["phase_ms"], run["rollout_ms"], run["per_iter"]
    collectives_per_iter, peer_reductions_per_iter = run["collectives_per_iter"], run["peer_reductions_per_iter"]
    collective_bytes_per_iter, collective_ms_per_iter = run["collective_bytes_per_iter"], run["collective_ms_per_iter"]

    steps_per_iter = world * n_envs * T
    value = steps_per_iter * args.steps / elapsed
    avg_rollout_s = (sum(rollout_ms) / len(rollout_ms)) * 1e-3
    alg_bytes = (wl["step_bytes"] + wl["record_bytes"]) * n_envs * T
    achieved = alg_bytes / avg_rollout_s / 1e9
    # HBM traffic of the dominant kernel from the committed rocprofv3 --pmc summary of this command
    # (profiles/run_profile.sh; separate FETCH_SIZE / WRITE_SIZE passes), bytes per launch
    traffic, traffic_src = None, None
    compute_axis = None      # the rollout's vector-instruction counters, same stamped file
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath) and not args.hidden:
        rec = json.load(open(tpath)).get(args.workload)
        if rec and rec.get("n_envs") == n_envs:
            if rec.get("kernel_source_hash") == kernel_source_hash():
                traffic, traffic_src = rec["rollout_bytes_per_launch"], rec["source"] + \
                    "; a builder-run pass committed as profiles/pmc_traffic.json, gated on the kernel-source hash -- not " \
                    "measured in this run (valu_* and issue_slot_floor_ms come from the same file: compute_source)"
                if rec.get("rollout_insts_valu"):
                    compute_axis = rec
            else:
                traffic_src = ("stale: profiles/pmc_traffic.json was taken from kernel sources %s, this build is %s "
                               "(re-run profiles/run_profile.sh)" % (rec.get("kernel_source_hash"), kernel_source_hash()))

    # second roofline: the matrix-core kernel of the update (Fisher-vector product), timed live
    do, da, h = policy.obs_dim, policy.action_dim, wl["hidden"][0]
    ht, ks0, ks1 = h // 32, (do + 2) // 2, 16 * (h // 32)
    # algorithmic matrix instructions of one FVP per 32-sample tile: tangent forward + back-propagation + the
    # W1 outer product (the forward pass is read back from the gradient pass's activation cache): what TRPO's
    # CG loop actually launches
    cached = True      # both net widths keep the activations (32 units: LDS-direct prefetch, 64: register prefetch)
    mfma_per_tile = (0 if cached else ht * (ks0 + ks1)) + ht * (ks0 + 2 * ks1) + ht * ks1 + ht * ht * 16
    layout = policy.kernel_layout()
    wide = layout is not None and layout.wide        # (the HIP envs' (obs, action) pairs all have the equal-width kernels)
    if wide:
        # the cooperative kernels of the wide / deep nets: algorithmic work of one FVP in multiply-adds per sample, on
        # the REAL layer sizes (padding is the kernels' cost, not the algorithm's): tangent (two products per layer
        # beyond the first) + back-propagation + the outer products; the forward pass is read back from the gradient
        # pass's activation cache, as for the equal-width nets
        ins = (do + 1,) + tuple(wl["hidden"])
        pw = sum(ins[l] * ins[l + 1] for l in range(len(ins) - 1))
        first = ins[0] * ins[1]
        macs = (2 * pw - first) + (pw - first) + pw
        mfma_per_tile = macs * 32 / 2048.0             # one v_mfma_f32_32x32x2_f32 = 2048 multiply-adds
    # ... and ONE convention for both families beside it: every multiply-add of the product on the REAL layer sizes, the
    # output layer's three products and the input layer's outer product included (the 71-count above omits them)
    ins_r = (do + 1,) + tuple(wl["hidden"])
    pw_r = sum(ins_r[l] * ins_r[l + 1] for l in range(len(ins_r) - 1))
    first_r = ins_r[0] * ins_r[1]
    macs_real = (2 * pw_r - first_r) + (pw_r - first_r) + pw_r + 4 * ins_r[-1] * da
    mfma_real_per_tile = macs_real * 32 / 2048.0
    fvp_ms = None
    fvp_variant = 0
    ops = policy.fused_ops() if wl["algo"] == "trpo" else None
    if ops is not None:
        from rllab_amd.algos.npo import npo_inputs
        inp = npo_inputs(policy, last["samples"])
        v = torch.randn(policy.flat_params.numel(), device="cuda", dtype=torch.float64)
        ops.loss_grad(inp, keep_activations=True)     # as ConjugateGradientOptimizer.optimize does before CG
        # which arithmetic the library runs these products in (0: f32 matrix instructions, 1: bf16 matrix
        # instructions on three-way split f32 operands -- csrc/policy_split_kernels.hip, 4: f16 matrix instructions on
        # two-way split operands -- csrc/policy_splith_kernels.hip)
        fvp_variant = ops.fvp_variant(inp)
        def timed20(fn):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = ev(), ev()
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / 20
        # (i) the public call: f64 vector in, f64 product out (conversion / layout kernels around the product) -- the
        #     number rounds 1 and 2 quoted; (ii) exactly what the CG loop launches per iteration (FusedGaussianMLPOps.
        #     _cg_loop -> _fvp_into: the product kernel + the partial-row reduction on the f32 direction CG keeps)
        fvp_call_ms = timed20(lambda: ops.fvp(inp, v))
        ws_, keep_ = ops._workspace(v.device), ops._batch(inp)
        v32 = ops.layout.pack(v.to(torch.float32)).contiguous()
        out64 = torch.empty(ops.n_kernel, dtype=torch.float64, device=v.device)
        fvp_ms = timed20(lambda: ops._fvp_into(keep_[0], ws_, v32, out64, inp))
        ops.release()
    # which rollout kernel rl_rollout_gaussian_mlp launched, in which shape: the launcher's own plan (rl_rollout_plan_query --
    # the launch rules as data; nothing is re-derived here)
    plan = algo.sampler.vec_env.rollout_plan(policy, T)
    assert plan is not None, "bench.py times the fused rollout"
    envs_per_wave, n_waves = plan.envs_per_wavefront, plan.wavefronts
    ROLLOUT_NOTES = {
        1: "fused policy + env step + record; %s" % ("one env per lane" if envs_per_wave == 64 else
                                                     "16 envs per wavefront, the physics replicated in four lanes"),
        2: "fused wide / deep policy + env step + record; weight fragments in LDS",
        3: "fused mean + log-std networks + env step + record",
        4: "fused policy + env step + record; 16 envs per wavefront, four lanes per env in the physics sub-steps",
        5: "fused wide / deep policy + env step + record; 16 envs per wavefront, four lanes per env in the physics sub-steps",
        6: "fused wide / deep policy + env step + record; FOUR wavefronts per group of 16 envs: the layers split by output "
           "units on 16 x 16 x 4 matrix tiles, activations through LDS; four lanes per env in the physics sub-steps",
        7: "fused policy + env step + record; ONE env per wavefront: the policy's units on the lanes, one body per lane in "
           "the physics sub-steps, the trajectory stored lane-distributed",
        8: "fused policy + env step + record; 16 envs per wavefront, a lane group per env with one body (of both legs) per lane in the "
           "physics sub-steps",
        9: "fused wide / deep policy + env step + record; 16 envs per wavefront, one body (of both legs) per lane in the physics sub-steps",
    }
    rollout_name = "%s (%s)" % (plan.name.decode(), ROLLOUT_NOTES.get(plan.kernel, ""))
    out = {
        "metric": "env steps/sec over full TRPO iterations (sample + process + update), 4096 envs per MI355X",
        "value": value, "unit": "env_steps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": args.workload, "env": wl["env"], "n_envs_per_gpu": n_envs,
                   "max_path_length": T, "policy": "GaussianMLPPolicy%s" % (wl["hidden"],),
                   "algo": wl["algo"], "samples_per_iteration": steps_per_iter,
                   "parallelism": "env-sharded dp%d" % world},
        "ranks": D.world_size() if dist.is_initialized() else 1,
        "backend": ("%s (RCCL)" % D.backend() if D.backend() == "nccl" else D.backend()) if dist.is_initialized()
        else None,
        "collectives_per_iter": collectives_per_iter, "collective_bytes_per_iter": collective_bytes_per_iter,
        "peer_reductions_per_iter": peer_reductions_per_iter,
        "collective_ms_per_iter": collective_ms_per_iter,
        "update_sum_path": run["update_sum_path"],
        "preflight": preflight_rec,
        "trpo_iter_ms": elapsed / args.steps * 1e3,
        "phase_ms": phase_ms,
        "per_rank_ms": run["per_rank_ms"], "rank_skew_ms": run["rank_skew_ms"], "other_sum_path": other,
        "update_ms_and_backtracks_per_iteration": per_iter,
        "sampler_env_steps_per_s": world * n_envs * T / avg_rollout_s,
        "roofline": {"kernel": rollout_name, "bound": "hbm",
                     "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                     "traffic": traffic, "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": avg_rollout_s * 1e3,
                     "wavefronts": n_waves, "simds": 1024,
                     "envs_per_wavefront": envs_per_wave,
                     "note": "issue-bound, not HBM-bound: %d env(s) per wavefront => %d wavefronts on 1024 SIMDs, each "
                             "a single instruction stream (physics sub-steps fused in registers); a lone wavefront "
                             "pays 4 cycles per issue slot whatever it issues (vector, scalar, each s_nop wait "
                             "state; packed f32 5, transcendental 8 -- tools/ubench/valu_latency.hip), so launch "
                             "time = issue slots per wavefront x 4 cycles x T and the HBM fraction is small by "
                             "construction (SURVEY.md 8d, DESIGN.md 3.1)" % (envs_per_wave, n_waves)},
    }
    if compute_axis is not None:
        # the compute axis from the kernel's own counters (profiles/pmc_traffic.json, stamped with the kernel-source
        # hash): vector instructions x 64 lanes / launch time against the vector peak in the same unit (157.3 TFLOP/s
        # counts a fused multiply-add as two: 78.65 T lane-instructions/s), and the floor of this design -- a lone
        # wavefront pays 4 cycles per issued vector instruction -- at the clock the launch ran at
        insts, waves_c = compute_axis["rollout_insts_valu"], compute_axis["rollout_waves"]
        clock_hz = compute_axis["rollout_gui_active"] / 8.0 / avg_rollout_s
        per_wave_step = insts / waves_c / T
        floor_ms = per_wave_step * 4.0 * T / clock_hz * 1e3
        out["roofline"].update({
            # what binds this kernel: the issue slots of a lone wavefront.  issue_frac = the launch's issue-slot floor
            # (vector instructions per wavefront x 4 cycles, at the clock the launch ran at) / the measured launch time
            "bound": "issue", "issue_frac": floor_ms / (avg_rollout_s * 1e3),
            "hbm_frac": achieved / 8000.0,
            "valu_tflops": insts * 64.0 / avg_rollout_s / 1e12, "valu_peak_tflops": 157.3 / 2.0,
            "valu_unit": "T lane-instructions/s (SQ_INSTS_VALU x 64 / launch time; an FMA counts once)",
            "valu_insts_per_wavefront_and_env_step": per_wave_step,
            "clock_ghz": clock_hz / 1e9,
            "issue_slot_floor_ms": floor_ms,
            "valu_issue_frac": compute_axis["rollout_active_inst_valu"] / compute_axis["rollout_wave_cycles"],
            "compute_source": "rocprofv3 --pmc SQ_INSTS_VALU / SQ_WAVES / SQ_WAVE_CYCLES / SQ_ACTIVE_INST_VALU / "
                              "GRBM_GUI_ACTIVE, a builder-run pass of this command committed as profiles/pmc_traffic.json "
                              "(%s) and gated on the kernel-source hash; NOT measured in this run -- valu_*, clock_ghz and "
                              "issue_slot_floor_ms combine those counters with this run's launch time" % compute_axis["source"]})
    if fvp_ms is not None:
        tiles = (n_envs * T + 31) // 32
        tf = tiles * mfma_per_tile * 4096 / (fvp_ms * 1e-3) / 1e12
        if fvp_variant == 1:
            # the split product: the SAME algorithmic f32 work (mfma_per_tile f32-matrix-instruction equivalents per tile)
            # priced against the same f32 matrix peak, so that the number is comparable with earlier rounds; what the
            # bf16 pipe executes for it is 6 cross terms per product + 21 transposition products per tile
            kern = "%s (Fisher-vector product, v_mfma_f32_32x32x16_bf16 on three-way split f32 operands)" % (
                "fvp_split_kernel" if h == 32 else "fvp_split64_kernel")
            # six terms per 32 x 32 x 16 block product + the transposition products of the sample-axis operands
            kb0 = (do + 1 + 15) // 16
            bf16_mfma = 6 * (ht * kb0 + 3 * ht * 2 * ht + 2 * ht * ht + 2 * ht) + 3 * (3 * 2 * ht + kb0)
            extra = {"arithmetic": "f32 operands split hi + mid + lo (exact), six bf16 cross terms per product, f32 "
                                   "accumulation: dropped terms <= 2^-23 |a b| worst case, 2^-28 mean (tests/test_split_arithmetic.py, test_gpu_fvp_split.py)",
                     "bf16_mfma_per_32_samples": bf16_mfma,
                     "bf16_pipe_frac": tiles * bf16_mfma * 32768 / (fvp_ms * 1e-3) / 2.5e15}
        elif fvp_variant == 4:
            # the two-way f16 split (csrc/policy_splith_kernels.hip): the same algorithmic f32 work against the same peak;
            # what the f16 pipe executes is three cross terms per product + 14 transposition products per tile
            kern = "%s (Fisher-vector product, v_mfma_f32_32x32x16_f16 on two-way split f32 operands)" % (
                "fvp_splith_kernel" if h == 32 else "fvp_splith64_kernel")
            kb0 = (do + 1 + 15) // 16
            f16_mfma = 3 * (ht * kb0 + 3 * ht * 2 * ht + 2 * ht * ht + 2 * ht) + 2 * (3 * 2 * ht + kb0)
            extra = {"arithmetic": "f32 operands split hi + 2^-11 lo' (f16 parts, lo scaled to stay normal), three cross terms per "
                                   "product, f32 accumulation, per-launch power-of-two operand scales with worst-case bounds; "
                                   "closer to float64 than an f32 fma chain (tools/ubench/f16_split.hip, "
                                   "tests/test_gpu_fvp_split.py)",
                     "f16_mfma_per_32_samples": f16_mfma,
                     "f16_pipe_frac": tiles * f16_mfma * 32768 / (fvp_ms * 1e-3) / 2.5e15}
        elif fvp_variant == 2:
            kern = ("csplit_fvp_kernel (Fisher-vector product of a wide / deep net, cooperative tiling, "
                    "v_mfma_f32_32x32x16_bf16 on three-way split f32 operands; parts images in LDS, transposing reads)")
            extra = {"arithmetic": "f32 operands split hi + mid + lo (exact), six bf16 cross terms per product, f32 "
                                   "accumulation (tests/test_split_arithmetic.py, test_gpu_csplit.py)"}
        else:
            kern = ("wide_pass_kernel<FVP>" if wide else "policy_pass_kernel<FVP>") + \
                " (Fisher-vector product, v_mfma_f32_32x32x2_f32)"
            extra = {}
        out["roofline_mfma"] = dict({"kernel": kern,
                                     "bound": "mfma", "achieved": tf, "peak": 157.3, "unit": "TFLOP/s",
                                     "frac": tf / 157.3, "avg_launch_ms": fvp_ms, "avg_public_call_ms": fvp_call_ms,
                                     "frac_public_call": tf * fvp_ms / fvp_call_ms / 157.3,
                                     "mfma_per_32_samples": mfma_per_tile,
                                     "mfma_per_32_samples_real_macs": mfma_real_per_tile,
                                     "frac_real_macs": tiles * mfma_real_per_tile * 4096 / (fvp_ms * 1e-3) / 1e12 / 157.3,
                                     "activations": "read from the gradient pass's cache" if cached else "recomputed",
                                     "note": "algorithmic f32 flops (mfma_per_32_samples x 4096 per tile) against the "
                                             "f32-input MFMA peak = the f32 vector peak (MI355X_MICROARCH.md); "
                                             "avg_launch_ms = product kernel + partial-row reduce kernel, the launches of "
                                             "one CG iteration (_fvp_into); avg_public_call_ms = rl-level fvp() with its "
                                             "f64 <-> f32 conversion kernels, the quantity rounds 1-2 reported as frac; "
                                             "frac_real_macs counts every multiply-add of the product on the real layer "
                                             "sizes (output layer and input-layer outer product included) -- one "
                                             "convention for the equal-width and the wide families"}, **extra)
    if os.environ.get("RLLAB_BENCH_HOSTTIMES") and rank == 0:
        names = ["events", "obtain_samples (enqueue)", "process_samples (incl. its wait)", "optimize_policy (incl. waits)",
                 "dump_tabular", "loop overhead to next iteration"]
        n = len(host_stamps)
        acc = [0.0] * 6
        for i, h in enumerate(host_stamps):
            for j in range(5):
                acc[j] += h[j + 1] - h[j]
            if i + 1 < n:
                acc[5] += host_stamps[i + 1][0] - h[5]
        for nm, v in zip(names, acc):
            sys.stderr.write("host %-36s %8.1f us / iteration\n" % (nm, v / n * 1e6))
    if rank == 0 and world == 1:
        # The per-step VecEnv boundary kernel (rl_vecenv_step) at a chip-filling size: the one kernel of the path
        # the HBM roofline really applies to (the fused rollout keeps state in registers).  Timed live with HIP
        # events, same recipe as tools/step_kernel_roofline.py; algorithmic bytes per env-step =
        # 4 (2 S + Da + Do + 1) + 1 + 8 (ts read + write).
        out["roofline_step_kernel"] = [step_kernel_roofline(torch, k) for k in sorted({0, env_kind})]
        out["roofline_scan"] = scan_rooflines(torch, algo, last["samples"]["_traj"])
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # (1) the reference's own sampler, unmodified, in a child process (rllab must resolve to the reference there)
        from oracle import cpu_sampler, ref_sampler
        theta_host = policy.get_param_values()
        ref, ref_error = None, None
        try:
            ref = ref_sampler.timed_reference(env_kind, theta_host, T, budget_s=args.cpu_budget, hidden=wl["hidden"])
        except Exception as err:      # oracle/_ref not staged on this box: say so and report the port, never crash
            ref_error = "%s: %s" % (type(err).__name__, str(err).splitlines()[0] if str(err) else "")
            sys.stderr.write("[bench] reference sampler unavailable (%s); reporting the port\n" % ref_error)
        # (2) cross-check: the re-typed port of the same sampler on the same cores (short)
        base = cpu_sampler.timed_baseline(env_kind, theta_host, T,
                                          budget_s=min(6.0, args.cpu_budget) if ref else args.cpu_budget,
                                          hidden=wl["hidden"])
        if ref is None:
            out["cpu_baseline"] = {
                "value": base["steps_per_s"], "unit": "env_steps/s", "cores": base["cores"], "kind": "port",
                "hw_threads": os.cpu_count(), "physical_cores": ref_sampler.physical_cores(),
                "cpu_model": ref_sampler.cpu_model(),
                "sample": "%d env steps in %.1f s of oracle/cpu_sampler.py (the reference's sampler re-typed); the "
                          "reference's own modules were not available: %s" % (base["steps"], base["seconds"], ref_error),
                "one_core_env_steps_per_s": base["steps_per_s_1core"]}
            ref = dict(steps_per_s=base["steps_per_s"])
        if "cpu_baseline" not in out:
            ratio = base["steps_per_s"] / ref["steps_per_s"]
            note = ("oracle/cpu_sampler.py (the same sampler re-typed: mp.Pool + Manager counter, one ctypes call per "
                    "step with the action map in C)")
            if not 0.5 <= ratio <= 2.0:
                note += ("; differs from the reference by more than 2x: the port skips the reference's per-step Python "
                         "layers (NormalizedEnv.step, Box.flatten, Step namedtuple, tensor_utils stacking)")
            out["cpu_baseline"] = {
                "value": ref["steps_per_s"], "unit": "env_steps/s", "cores": ref["cores"], "kind": "reference",
                # "cores" = worker processes used = one per HARDWARE THREAD of the box (os.cpu_count()); the sockets'
                # physical cores are half of that on an SMT-2 part
                "hw_threads": os.cpu_count(), "physical_cores": ref_sampler.physical_cores(),
                "cpu_model": ref["cpu_model"],
                "sample": "%d env steps (%d paths) in %.1f s of the reference's unmodified parallel_sampler.sample_paths "
                          "/ StatefulPool.run_collect / rollout / NormalizedEnv (%s, staged by oracle/make_ref.py) with "
                          "n_parallel = %d worker processes; env = float64 host build of this repo's dynamics behind the "
                          "reference Env interface, policy = batch-1 NumPy MLP behind the reference Policy interface "
                          "(pybox2d / MuJoCo 1.31 / Theano are absent): sampler only, an upper bound on the true "
                          "reference stack" % (ref["steps"], ref["n_paths"], ref["seconds"], ref["ref_root"],
                                               ref["cores"]),
                "one_core_env_steps_per_s": ref["steps_per_s_1core"],
                "reference_modules": ref["modules"],
                "port_cross_check": {
                    "value": base["steps_per_s"], "one_core_env_steps_per_s": base["steps_per_s_1core"],
                    "cores": base["cores"], "port_over_reference": ratio, "note": note}}
        if wl["algo"] == "trpo":
            # the rest of the reference iteration on the CPU, on (a bounded prefix of) the paths just sampled:
            # BaseSampler.process_samples and ConjugateGradientOptimizer.optimize as restated in oracle/
            from oracle import cpu_iteration
            it = cpu_iteration.timed_process_and_update(base["paths"], theta_host, wl["hidden"],
                                                        gae_lambda=wl["lam"])
            scale = steps_per_iter / float(it["samples"])
            out["cpu_baseline"]["iteration"] = {
                "samples": it["samples"], "process_s": it["process_s"], "update_s": it["update_s"],
                "torch_threads": it["torch_threads"],
                "est_iteration_s_at_bench_batch": steps_per_iter / ref["steps_per_s"]
                + (it["process_s"] + it["update_s"]) * scale,
                "note": "oracle/cpu_iteration.py: reference process_samples (numpy port) + TRPO update (reference "
                        "control flow, float64 torch-CPU closures in place of the compiled Theano functions) on a "
                        "prefix of the sampled paths; the estimate scales both linearly to the bench batch and adds "
                        "the sampling time at the reference sampler's measured rate"}
    dist_on = dist.is_initialized()
    if dist_on:
        pr = D.peer_reducer()
        if pr is not None:
            pr.check()                      # no in-stream reduction gave up waiting for a peer
            D.peer_shutdown()
        # RCCL leaves a banner (its version, the HIP version, hostname, library path) in the C stdio buffer of every
        # rank; flushed at exit it would follow the JSON line.  Every rank pushes it out BEFORE the last barrier, rank 0
        # prints the line after it, and all leave without the exit-time flushes: the one JSON line is the last line
        # of the job's stdout.
        import ctypes
        libc = ctypes.CDLL(None)
        libc.fflush(None)
        sys.stdout.flush()
        dist.barrier()
        dist.destroy_process_group()
        libc.fflush(None)
    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
    if dist_on:
        sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
