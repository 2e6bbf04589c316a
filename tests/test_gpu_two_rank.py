"""Env-sharded data parallelism on the REAL kernels: two ranks (gloo, both on cuda:0 -- the GPU box
has one device; the production backend is nccl = RCCL) each roll out half of the envs and run one
TRPO iteration; parameters, logged statistics and baseline coefficients must equal a single process
that owns all the envs.  Philox counters are keyed by the GLOBAL env index, so the union of the two
shards' trajectories is the single-process batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

KEYS = ["AverageReturn", "AverageDiscountedReturn", "ExplainedVariance", "NumTrajs", "StdReturn", "MaxReturn",
        "MinReturn", "LossBefore", "LossAfter", "MeanKLBefore", "MeanKL"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run(n_envs, n_itr=2, env_name="swimmer", pinned=False, hidden=(32, 32)):
    """``pinned``: one CG iteration and a single line-search candidate (max_backtracks=1, accept_violation) -- the
    update then is a smooth function of the all-reduced sums (no ill-conditioned 10-step Krylov recursion, no
    discrete 0.8x choice), so the sharded run must reproduce the single-process PARAMETERS to summation order."""
    from rllab_amd.algos.trpo import TRPO
    from rllab_amd.baselines.linear_feature_baseline import LinearFeatureBaseline
    from rllab_amd.envs.box2d.cartpole_env import CartpoleEnv
    from rllab_amd.envs.mujoco.swimmer_env import SwimmerEnv
    from rllab_amd.envs.normalized_env import normalize
    from rllab_amd.misc import ext, logger
    from rllab_amd.policies.gaussian_mlp_policy import GaussianMLPPolicy
    from rllab_amd.sampler import dist as D
    ext.set_seed(3)
    logger.set_quiet(True)
    env = normalize(SwimmerEnv() if env_name == "swimmer" else CartpoleEnv())
    policy = GaussianMLPPolicy(env_spec=env.spec, hidden_sizes=hidden)
    D.broadcast_(policy.flat_params)
    baseline = LinearFeatureBaseline(env_spec=env.spec)
    T = 40
    opt_args = dict(cg_iters=1, max_backtracks=1, accept_violation=True) if pinned else None
    algo = TRPO(env=env, policy=policy, baseline=baseline, batch_size=n_envs * T, max_path_length=T, n_itr=n_itr,
                discount=0.99, step_size=0.01, sampler_args=dict(n_envs=n_envs, seed=17), optimizer_args=opt_args)
    algo.start_worker()
    algo.init_opt()
    stats = []
    probes = None
    for itr in range(n_itr):
        paths = algo.sampler.obtain_samples(itr)
        sd = algo.sampler.process_samples(itr, paths)
        if itr == 0:
            # the all-reduced building blocks of the update at a fixed evaluation point
            from rllab_amd.algos.npo import npo_inputs
            ops = policy.fused_ops()
            inp = npo_inputs(policy, sd)
            v = torch.as_tensor(np.random.RandomState(0).randn(policy.flat_params.numel()), device="cuda")
            probes = np.concatenate([ops.loss_stats(inp).cpu().numpy(), ops.loss_grad(inp).cpu().numpy(),
                                     ops.fvp(inp, v).cpu().numpy()])
            ops.release()
        algo.optimize_policy(itr, sd)
        tab = logger.get_tabular()
        stats.append([float(tab[k]) for k in KEYS])
        logger.dump_tabular()
    torch.cuda.synchronize()
    return policy.get_param_values(), np.array(stats), np.asarray(baseline.get_param_values()), probes


def _worker(rank, world, port, outdir, env_name):
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    theta_p, stats_p, _, _ = _run(64, n_itr=3, env_name=env_name, pinned=True)
    np.save(os.path.join(outdir, "theta_pinned_%d.npy" % rank), theta_p)
    np.save(os.path.join(outdir, "stats_pinned_%d.npy" % rank), stats_p)
    theta, stats, coef, probes = _run(64, env_name=env_name)
    np.save(os.path.join(outdir, "probes_%d.npy" % rank), probes)
    np.save(os.path.join(outdir, "theta_%d.npy" % rank), theta)
    np.save(os.path.join(outdir, "stats_%d.npy" % rank), stats)
    np.save(os.path.join(outdir, "coef_%d.npy" % rank), coef)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("env_name", ["swimmer", "cartpole"])
def test_two_ranks_on_the_kernels_equal_one_process(tmp_path, env_name):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), env_name), nprocs=world, join=True)
    want_theta, want_stats, want_coef, want_probes = _run(128, env_name=env_name)
    t0, t1 = (np.load(str(tmp_path / ("theta_%d.npy" % r))) for r in range(2))
    assert np.array_equal(t0, t1)                                  # identical parameters on every rank, no broadcast
    s0 = np.load(str(tmp_path / "stats_0.npy"))
    # iteration 0: both runs sample with the same parameters -> the same trajectories; sampling
    # statistics agree to float64 reduction order, the update to the f32 partial-sum order
    n_samp = KEYS.index("LossBefore")
    assert np.allclose(s0[0, :n_samp], want_stats[0, :n_samp], rtol=1e-9, atol=1e-9), (s0[0], want_stats[0])
    # loss / KL, flat gradient and a Fisher-vector product: sums of per-rank terms, one all-reduce each
    p0, p1 = (np.load(str(tmp_path / ("probes_%d.npy" % r))) for r in range(2))
    assert np.array_equal(p0, p1)
    assert np.abs(p0 - want_probes).max() <= 2e-5 * np.abs(want_probes).max(), np.abs(p0 - want_probes).max()
    # the accepted step itself goes through 10 CG iterations on an ill-conditioned Fisher matrix, which
    # amplifies the f32 summation-order difference between the two batch partitions, and the backtracking
    # line search turns a KL that sits on the trust-region boundary into a discrete 0.8x choice: both runs
    # must take a valid TRPO step of the same scale, not the same step
    for row in (s0[0], want_stats[0]):
        loss_before, loss_after, kl_before, kl = row[n_samp:]
        assert loss_after < loss_before and 0.0 < kl <= 0.01 * (1 + 1e-6) and abs(kl_before) < 1e-6, row
    assert abs(s0[0, n_samp] - want_stats[0, n_samp]) < 1e-6
    ratio = s0[0, [n_samp + 1, n_samp + 3]] / want_stats[0, [n_samp + 1, n_samp + 3]]
    assert np.all((ratio > 0.5) & (ratio < 2.0)), (s0[0], want_stats[0])
    c0 = np.load(str(tmp_path / "coef_0.npy"))
    assert np.all(np.isfinite(s0)) and np.all(np.isfinite(c0))
    # parameters after two updates: the second rollout already runs on (slightly) different
    # parameters, so only closeness relative to the update size is meaningful
    theta_init = _initial_theta(env_name)
    step = np.abs(want_theta - theta_init).max()
    assert step > 0 and np.abs(t0 - want_theta).max() <= 0.75 * step
    # pinned control flow (cg_iters=1, one line-search candidate): THREE iterations of the sharded run reproduce the
    # single-process parameters and every logged number to the f32 summation order of the batch partition
    want_tp, want_sp, _, _ = _run(128, n_itr=3, env_name=env_name, pinned=True)
    tp0, tp1 = (np.load(str(tmp_path / ("theta_pinned_%d.npy" % r))) for r in range(2))
    sp0 = np.load(str(tmp_path / "stats_pinned_0.npy"))
    assert np.array_equal(tp0, tp1)
    moved = np.abs(want_tp - theta_init).max()
    assert moved > 0 and np.abs(tp0 - want_tp).max() <= 2e-3 * moved, (np.abs(tp0 - want_tp).max(), moved)
    assert np.allclose(sp0[0], want_sp[0], rtol=2e-4, atol=1e-7), (sp0[0], want_sp[0])
    assert np.allclose(sp0[:, :n_samp], want_sp[:, :n_samp], rtol=5e-3, atol=1e-4), (sp0, want_sp)


def _initial_theta(env_name):
    from rllab_amd.envs.box2d.cartpole_env import CartpoleEnv
    from rllab_amd.envs.mujoco.swimmer_env import SwimmerEnv
    from rllab_amd.envs.normalized_env import normalize
    from rllab_amd.misc import ext
    from rllab_amd.policies.gaussian_mlp_policy import GaussianMLPPolicy
    ext.set_seed(3)
    env = normalize(SwimmerEnv() if env_name == "swimmer" else CartpoleEnv())
    return GaussianMLPPolicy(env_spec=env.spec, hidden_sizes=(32, 32)).get_param_values()


def _peer_worker(rank, world, port, outdir, peer):
    """One TRPO iteration with the default optimizer settings (cg_iters 10): parameters + what crossed ranks how."""
    torch.cuda.set_device(0)
    if peer:
        os.environ["RLLAB_PEER_ALLREDUCE"] = "1"
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    from rllab_amd.algos.trpo import TRPO
    from rllab_amd.baselines.linear_feature_baseline import LinearFeatureBaseline
    from rllab_amd.envs.mujoco.swimmer_env import SwimmerEnv
    from rllab_amd.envs.normalized_env import normalize
    from rllab_amd.misc import ext, logger
    from rllab_amd.policies.gaussian_mlp_policy import GaussianMLPPolicy
    from rllab_amd.sampler import dist as D
    ext.set_seed(3)
    logger.set_quiet(True)
    env = normalize(SwimmerEnv())
    policy = GaussianMLPPolicy(env_spec=env.spec, hidden_sizes=(32, 32))
    D.broadcast_(policy.flat_params)
    T, n = 40, 64
    algo = TRPO(env=env, policy=policy, baseline=LinearFeatureBaseline(env_spec=env.spec), batch_size=n * T,
                max_path_length=T, n_itr=3, discount=0.99, step_size=0.01, sampler_args=dict(n_envs=n, seed=17))
    algo.start_worker()
    algo.init_opt()
    counts = []
    for itr in range(3):
        D.reset_accounting()
        paths = algo.sampler.obtain_samples(itr)
        sd = algo.sampler.process_samples(itr, paths)
        algo.optimize_policy(itr, sd)
        acct = D.accounting()
        counts.append([acct["count"], acct["peer"]])
        logger.dump_tabular()
    torch.cuda.synchronize()
    if peer:
        assert D.peer_reducer() is not None and D.peer_reducer().count == sum(c[1] for c in counts)
        D.peer_poll()
        D.peer_reducer().check()                    # no reduction gave up waiting for its peer
    tag = "peer" if peer else "host"
    np.save(os.path.join(outdir, "theta_%s_%d.npy" % (tag, rank)), policy.get_param_values())
    np.save(os.path.join(outdir, "counts_%s_%d.npy" % (tag, rank)), np.array(counts))
    D.peer_shutdown()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_four_ranks_with_the_in_stream_peer_allreduce(tmp_path):
    """The same protocol with FOUR processes on the one GPU: every rank writes into three peers' mailboxes and waits for
    three flags per reduction, the two slots alternate 33 times.  All four ranks end on bit-identical parameters, and
    they are the host-backend run's to the rounding of a four-term sum taken in a different order."""
    world = 4
    mp.spawn(_peer_worker, args=(world, _free_port(), str(tmp_path), True), nprocs=world, join=True)
    mp.spawn(_peer_worker, args=(world, _free_port(), str(tmp_path), False), nprocs=world, join=True)
    tp = [np.load(str(tmp_path / ("theta_peer_%d.npy" % r))) for r in range(world)]
    th = [np.load(str(tmp_path / ("theta_host_%d.npy" % r))) for r in range(world)]
    for r in range(1, world):
        assert np.array_equal(tp[0], tp[r]) and np.array_equal(th[0], th[r])
    # gloo's ring adds the four rows in its own order; the peer kernel in rank order: same sums to float64 rounding,
    # which three TRPO iterations (CG on an ill-conditioned matrix, a discrete line search) may amplify a little
    theta0 = _initial_theta("swimmer")
    moved = np.abs(th[0] - theta0).max()
    assert moved > 0 and np.abs(tp[0] - th[0]).max() <= 1e-3 * moved, (np.abs(tp[0] - th[0]).max(), moved)
    cp = np.load(str(tmp_path / "counts_peer_0.npy"))
    assert np.all(cp[:, 1] == 11) and np.all(cp[:, 0] <= 5), cp


@pytest.mark.timeout(600)
def test_two_ranks_with_the_in_stream_peer_allreduce(tmp_path):
    """RLLAB_PEER_ALLREDUCE=1: the gradient and the ten Fisher-vector products of a TRPO update are summed across ranks
    by rl_peer_allreduce_sum -- every rank writes its row into the peer's hipIpc-mapped mailbox, flags it and sums in
    rank order, on the update's stream -- instead of eleven host-issued collectives.  Two processes share the one
    GPU of the box (the mailboxes are then plain device memory; on a node they are peer memory over xGMI).  The
    parameters after three iterations are BIT-IDENTICAL on both ranks and to the run that reduces through the host
    backend, and at most six collectives per iteration are left to torch.distributed."""
    world = 2
    mp.spawn(_peer_worker, args=(world, _free_port(), str(tmp_path), True), nprocs=world, join=True)
    mp.spawn(_peer_worker, args=(world, _free_port(), str(tmp_path), False), nprocs=world, join=True)
    tp = [np.load(str(tmp_path / ("theta_peer_%d.npy" % r))) for r in range(world)]
    th = [np.load(str(tmp_path / ("theta_host_%d.npy" % r))) for r in range(world)]
    assert np.array_equal(tp[0], tp[1]) and np.array_equal(th[0], th[1])
    assert np.array_equal(tp[0], th[0])             # rank-order sum of two rows == the backend's a + b, bit for bit
    cp = np.load(str(tmp_path / "counts_peer_0.npy"))
    ch = np.load(str(tmp_path / "counts_host_0.npy"))
    assert np.all(cp[:, 1] == 11) and np.all(ch[:, 1] == 0)          # gradient + cg_iters products, every iteration
    assert np.all(cp[:, 0] <= 5), cp                    # [statistics | normal equations], the gradient's loss sums, 2-3 candidates
    # the host-backend run: ten products + the gradient, which there shares ONE all-gather with its loss sums (round 5)
    assert np.all(ch[:, 0] == cp[:, 0] + 10), (ch, cp)


def _wide_worker(rank, world, port, outdir):
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    theta, stats, _, probes = _run(64, n_itr=2, pinned=True, hidden=(100, 50, 25))
    np.save(os.path.join(outdir, "wide_theta_%d.npy" % rank), theta)
    np.save(os.path.join(outdir, "wide_probes_%d.npy" % rank), probes)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_with_a_wide_policy_equal_one_process(tmp_path):
    """The sharded path with the cooperative-workgroup kernels and the lane-group wide rollout ((100, 50, 25) -> (128, 64, 32)):
    loss / gradient / Fisher-vector product summed over two shards equal the single-process values, and with the control
    flow pinned two iterations end on the single-process parameters to the f32 summation order of the batch partition."""
    world = 2
    mp.spawn(_wide_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    want_theta, _, _, want_probes = _run(128, n_itr=2, pinned=True, hidden=(100, 50, 25))
    t0, t1 = (np.load(str(tmp_path / ("wide_theta_%d.npy" % r))) for r in range(2))
    p0, p1 = (np.load(str(tmp_path / ("wide_probes_%d.npy" % r))) for r in range(2))
    assert np.array_equal(t0, t1) and np.array_equal(p0, p1)
    assert np.abs(p0 - want_probes).max() <= 2e-5 * np.abs(want_probes).max(), np.abs(p0 - want_probes).max()
    from rllab_amd.envs.mujoco.swimmer_env import SwimmerEnv
    from rllab_amd.envs.normalized_env import normalize
    from rllab_amd.misc import ext
    from rllab_amd.policies.gaussian_mlp_policy import GaussianMLPPolicy
    ext.set_seed(3)
    theta_init = GaussianMLPPolicy(env_spec=normalize(SwimmerEnv()).spec, hidden_sizes=(100, 50, 25)).get_param_values()
    moved = np.abs(want_theta - theta_init).max()
    assert moved > 0 and np.abs(t0 - want_theta).max() <= 5e-3 * moved, (np.abs(t0 - want_theta).max(), moved)
