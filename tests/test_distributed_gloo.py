"""world_size-2 ``gloo`` tests (CPU) of the env-sharded data-parallel path: every statistic,
the baseline normal equations, the flat gradient and each Hessian-vector product are SUM
all-reduced, so two ranks holding half of the batch each must end up with the parameters a
single process computes on the whole batch -- identical on both ranks, no broadcast."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _spec(do, da):
    from rllab_amd.envs.env_spec import EnvSpec
    from rllab_amd.spaces import Box
    return EnvSpec(Box(-np.ones(do), np.ones(do)), Box(-np.ones(da), np.ones(da)))


def _make_problem(seed=0, B=600, do=5, da=2):
    from rllab_amd.policies.gaussian_mlp_policy import GaussianMLPPolicy
    np.random.seed(seed)
    pol = GaussianMLPPolicy(_spec(do, da), hidden_sizes=(8, 8))
    rng = np.random.RandomState(seed + 1)
    obs = torch.as_tensor(rng.randn(do, B).astype(np.float32))
    with torch.no_grad():
        mean = pol.mean_planes(obs)
    ls = pol.effective_log_std().detach()
    act = mean + torch.exp(ls)[:, None] * torch.as_tensor(rng.randn(da, B).astype(np.float32))
    adv = torch.as_tensor(rng.randn(B).astype(np.float32))
    w = torch.ones(B)
    w[torch.as_tensor(rng.rand(B) < 0.1)] = 0.0
    return pol, (obs, act, adv, mean, ls.reshape(-1, 1), w)


def _shard(inputs, rank, world):
    obs, act, adv, mean, ls, w = inputs
    B = obs.shape[-1]
    lo, hi = rank * B // world, (rank + 1) * B // world
    sl = lambda x: x[..., lo:hi].contiguous()
    return sl(obs), sl(act), sl(adv), sl(mean), ls, sl(w)


def _closures(pol):
    from rllab_amd.algos.npo import NPO
    from rllab_amd.optimizers.conjugate_gradient_optimizer import ConjugateGradientOptimizer
    algo = NPO.__new__(NPO)
    algo.policy, algo.truncate_local_is_ratio, algo.step_size, algo.use_fused = pol, None, 0.01, False
    captured = {}

    class _Opt(object):
        def update_opt(self, **kw):
            captured.update(kw)
    algo.optimizer = _Opt()
    algo.init_opt()
    return captured["loss"], captured["leq_constraint"][0]


def _run_trpo(pol, shard_inputs):
    from rllab_amd.misc import logger
    from rllab_amd.optimizers.conjugate_gradient_optimizer import ConjugateGradientOptimizer
    from rllab_amd.sampler import dist as D
    logger.set_quiet(True)
    cnt = D.all_reduce_sum_(shard_inputs[-1].double().sum())
    inputs = tuple(shard_inputs) + (1.0 / cnt,)
    loss, kl = _closures(pol)
    opt = ConjugateGradientOptimizer()
    opt.update_opt(loss=loss, target=pol, leq_constraint=(kl, 0.01), inputs=None)
    before = (opt.loss(inputs), opt.constraint_val(inputs))
    opt.optimize(inputs)
    return pol.get_param_values(), before, (opt.loss(inputs), opt.constraint_val(inputs))


def _run_vpg(pol, shard_inputs):
    from rllab_amd.misc import logger
    from rllab_amd.optimizers.first_order_optimizer import FirstOrderOptimizer
    from rllab_amd.sampler import dist as D
    logger.set_quiet(True)
    cnt = D.all_reduce_sum_(shard_inputs[-1].double().sum())
    inputs = tuple(shard_inputs) + (1.0 / cnt,)
    dist_ = pol.distribution

    def surr(flat, obs, act, adv, om, ols, w, inv):
        new = pol.dist_info_planes(obs, flat)
        return -(dist_.log_likelihood_sym(act, new, axis=0) * adv * w).sum() * inv.to(torch.float32)
    opt = FirstOrderOptimizer(batch_size=None, max_epochs=1)
    opt.update_opt(surr, target=pol)
    opt.optimize(inputs)
    opt.optimize(inputs)
    return pol.get_param_values()


def _traj_from(rng, T, N, do):
    from rllab_amd.sampler.trajectories import Trajectories
    done = torch.as_tensor((rng.rand(T, N) < 0.1).astype(np.uint8))
    tr = Trajectories(torch.as_tensor(rng.randn(do, T, N).astype(np.float32)), torch.zeros(1, T, N),
                      torch.zeros(1, T, N), torch.zeros(1), torch.as_tensor(rng.randn(T, N).astype(np.float32)),
                      done, T)
    tr.valid = tr.valid_mask(True)
    tr.returns = torch.as_tensor(rng.randn(T, N).astype(np.float32))
    return tr


def _worker(rank, world, port, outdir):
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    from rllab_amd.baselines.linear_feature_baseline import LinearFeatureBaseline
    from rllab_amd.sampler import dist as D
    from rllab_amd.sampler.base import merge_stats
    assert D.is_distributed() and D.world_size() == world and D.rank() == rank
    # statistics
    s, = D.sums(torch.tensor(float(rank + 1)))
    assert float(s) == 3.0
    assert float(D.all_reduce_min_(torch.tensor(float(rank)))) == 0.0
    assert float(D.all_reduce_max_(torch.tensor(float(rank)))) == 1.0
    # rl_sample_stats rows of two shards -> global row: 15 sum columns, then min / max / min / max / min
    row = torch.arange(20, dtype=torch.float64) + 100.0 * rank
    row[15], row[16], row[17], row[18], row[19] = -5.0 - rank, 7.0 + rank, -3.0 + rank, 2.0 - rank, 4.0 + rank
    got = merge_stats(row).numpy()
    assert np.array_equal(got[:15], 2 * np.arange(15) + 100.0)
    assert got[15] == -6.0 and got[16] == 8.0 and got[17] == -3.0 and got[18] == 2.0 and got[19] == 4.0
    # TRPO update on half of the batch
    pol, inputs = _make_problem()
    theta, before, after = _run_trpo(pol, _shard(inputs, rank, world))
    np.save(os.path.join(outdir, "trpo_%d.npy" % rank), theta)
    np.save(os.path.join(outdir, "trpo_stats_%d.npy" % rank), np.array(before + after))
    pol2, inputs2 = _make_problem(seed=5)
    np.save(os.path.join(outdir, "vpg_%d.npy" % rank), _run_vpg(pol2, _shard(inputs2, rank, world)))
    # baseline normal equations
    rng = np.random.RandomState(100 + rank)
    b = LinearFeatureBaseline(None)
    b.fit_dense(_traj_from(rng, 30, 8, 3), all_reduce=D.all_reduce_sum_)
    np.save(os.path.join(outdir, "coef_%d.npy" % rank), b.get_param_values())
    # initial parameter sync of BatchPolopt.start_worker: ranks start from DIFFERENT np.random states (the
    # example scripts set no seed) and must leave with rank 0's parameters; collectives are counted
    from rllab_amd.algos.batch_polopt import BatchPolopt
    from rllab_amd.baselines.zero_baseline import ZeroBaseline
    from rllab_amd.policies.gaussian_mlp_policy import GaussianMLPPolicy
    np.random.seed(1000 + rank)
    pol3 = GaussianMLPPolicy(_spec(5, 2), hidden_sizes=(8, 8))

    class _Algo(object):
        policy, baseline = pol3, ZeroBaseline(None)
    np.save(os.path.join(outdir, "init_before_%d.npy" % rank), pol3.get_param_values())
    D.reset_accounting()
    BatchPolopt.sync_initial_parameters(_Algo())
    assert D.accounting()["count"] == 1 and D.accounting()["bytes"] == 4 * pol3.flat_params.numel()
    np.save(os.path.join(outdir, "init_after_%d.npy" % rank), pol3.get_param_values())
    from rllab_amd.misc import logger
    assert logger.is_primary() == (rank == 0)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_update_equals_single_process(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    # single-process references on the full batch
    pol, inputs = _make_problem()
    theta0 = pol.get_param_values()
    want, before, after = _run_trpo(pol, inputs)
    t0, t1 = (np.load(str(tmp_path / ("trpo_%d.npy" % r))) for r in range(2))
    assert np.array_equal(t0, t1)                                # identical update on every rank
    step = np.abs(want - theta0).max()
    assert step > 0 and np.abs(t0 - want).max() <= 1e-4 * step
    s0 = np.load(str(tmp_path / "trpo_stats_0.npy"))
    assert np.allclose(s0, np.array(before + after), rtol=1e-5, atol=1e-7)
    assert after[0] < before[0] and after[1] <= 0.01             # loss improved inside the trust region
    pol2, inputs2 = _make_problem(seed=5)
    want_v = _run_vpg(pol2, inputs2)
    v0, v1 = (np.load(str(tmp_path / ("vpg_%d.npy" % r))) for r in range(2))
    assert np.array_equal(v0, v1) and np.abs(v0 - want_v).max() <= 1e-6
    # baseline: both ranks solve the same all-reduced normal equations == fit on the union
    from rllab_amd.baselines.linear_feature_baseline import LinearFeatureBaseline
    c0, c1 = (np.load(str(tmp_path / ("coef_%d.npy" % r))) for r in range(2))
    assert np.array_equal(c0, c1)
    trs = [_traj_from(np.random.RandomState(100 + r), 30, 8, 3) for r in range(2)]
    phi = torch.cat([LinearFeatureBaseline._features_dense(t) * t.valid.reshape(1, -1).double() for t in trs], 1)
    phi_raw = torch.cat([LinearFeatureBaseline._features_dense(t) for t in trs], 1)
    y = torch.cat([t.returns.reshape(-1).double() for t in trs])
    ref = LinearFeatureBaseline(None)._solve((phi @ phi_raw.t()).numpy(), (phi @ y).numpy())
    assert np.allclose(c0, ref, rtol=1e-8, atol=1e-10)
    # initial parameter broadcast
    b0, b1 = (np.load(str(tmp_path / ("init_before_%d.npy" % r))) for r in range(2))
    a0, a1 = (np.load(str(tmp_path / ("init_after_%d.npy" % r))) for r in range(2))
    assert not np.array_equal(b0, b1) and np.array_equal(a0, b0) and np.array_equal(a1, b0)


def _peer_refusal_worker(rank, world, port, outdir):
    """No GPU here: rank 0's mailbox stage is made to 'succeed' (stubbed), rank 1's fails for real -- every rank must
    leave the constructor with PeerUnavailable after the SAME stage, nobody waits for a peer that gave up."""
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    from rllab_amd import _lib
    from rllab_amd.sampler import dist as D
    released = []
    if rank == 0:
        class _FakeLib(object):
            def __getattr__(self, name):
                if name == "rl_peer_mailbox_bytes":
                    return lambda w, n: 1024
                if name in ("rl_peer_alloc", "rl_peer_export"):
                    return lambda *a: 0
                if name in ("rl_peer_free", "rl_peer_close"):
                    return lambda *a: released.append(name) or 0
                return getattr(_lib.lib, name)
        real = _lib.lib
        _lib.lib = _FakeLib()
    orig_sync = torch.cuda.synchronize
    torch.cuda.synchronize = lambda *a, **k: None
    orig_avail = torch.cuda.is_available
    try:
        try:
            D.PeerReducer(max_n=64)
            outcome = "constructed"
        except D.PeerReducer.PeerUnavailable as e:
            outcome = "refused: %s" % e
    finally:
        torch.cuda.synchronize = orig_sync
        if rank == 0:
            _lib.lib = real
    with open(os.path.join(outdir, "peer_%d.txt" % rank), "w") as fh:
        fh.write(outcome + "\n" + ",".join(released))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_peer_reducer_refuses_on_every_rank_when_one_rank_cannot_build_its_mailbox(tmp_path):
    """The pre-flight's contract (every rank takes the same branch): a stage of the peer
    all-reduce's construction that fails on ONE rank ends in PeerUnavailable on EVERY rank, after an all-reduce-min of
    the stage's verdict, with the successful rank's mailbox released."""
    world, port = 2, _free_port()
    mp.spawn(_peer_refusal_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    out = [open(str(tmp_path / ("peer_%d.txt" % r))).read().split("\n") for r in range(2)]
    assert out[0][0].startswith("refused") and out[1][0].startswith("refused"), out
    assert "rl_peer_free" in out[0][1]                     # rank 0 released the mailbox it had allocated
    assert "fine-grained" in out[1][0] or "rl_peer_alloc" in out[1][0] or "HIP" in out[1][0], out[1][0]
