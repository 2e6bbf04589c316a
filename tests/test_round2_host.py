"""Host-side behaviour added in round 2 (CPU): mini-batch normalisation of FirstOrderOptimizer,
the recorded (old) log_std never aliasing the parameter vector, rank-0-only logging, the
self-launch command of ``bench.py --gpus N`` and collective accounting of sampler/dist.py."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cpu_policy(do=5, da=2, h=8, min_std=1e-6):
    from rllab_amd.envs.env_spec import EnvSpec
    from rllab_amd.policies.gaussian_mlp_policy import GaussianMLPPolicy
    from rllab_amd.spaces import Box
    np.random.seed(0)
    spec = EnvSpec(Box(-np.ones(do), np.ones(do)), Box(-np.ones(da), np.ones(da)))
    return GaussianMLPPolicy(spec, hidden_sizes=(h, h), min_std=min_std)


@pytest.mark.parametrize("min_std", [1e-6, None])
def test_recorded_log_std_is_a_copy(min_std):
    """With min_std=None ``effective_log_std`` is a view into flat_params; what a rollout records as the OLD
    log_std must not move when the optimizer rewrites the parameters in place."""
    pol = _cpu_policy(min_std=min_std)
    rec = pol.recorded_log_std()
    before = rec.clone()
    with torch.no_grad():
        pol.flat_params.add_(1.0)
    assert torch.equal(rec, before)
    assert rec.data_ptr() != pol.effective_log_std().data_ptr()
    assert float((pol.effective_log_std() - before).abs().max()) == pytest.approx(1.0)


def test_minibatch_loss_is_the_minibatch_mean(quiet_logger):
    """FirstOrderOptimizer(batch_size=b): each step's loss is normalised by the mini-batch's own weight sum
    (the reference compiles a mean over whatever slice it is fed, first_order_optimizer.py:112-114), not by
    the full batch's 1/W."""
    from rllab_amd.optimizers.first_order_optimizer import FirstOrderOptimizer
    pol = _cpu_policy()
    B, b = 64, 16
    rng = np.random.RandomState(1)
    obs = torch.as_tensor(rng.randn(5, B).astype(np.float32))
    act = torch.as_tensor(rng.randn(2, B).astype(np.float32))
    adv = torch.as_tensor(rng.randn(B).astype(np.float32))
    w = torch.ones(B)
    w[::5] = 0.0
    seen = []
    dist_ = pol.distribution

    def surr(flat, obs, act, adv, w, inv):
        seen.append((float(w.sum()), float(inv), obs.shape[-1]))
        new = pol.dist_info_planes(obs, flat)
        return -(dist_.log_likelihood_sym(act, new, axis=0) * adv * w).sum() * inv.to(torch.float32)
    opt = FirstOrderOptimizer(batch_size=b, max_epochs=1)
    opt.update_opt(surr, target=pol, weighted_mean_inputs=True)      # what algos/vpg.py declares
    opt.optimize((obs, act, adv, w, torch.tensor(1.0 / float(w.sum()), dtype=torch.float64)))
    steps = [s for s in seen if s[2] == b]
    assert len(steps) == B // b
    for wsum, inv, _ in steps:
        assert inv == pytest.approx(1.0 / wsum)
    full = [s for s in seen if s[2] == B]
    assert full and all(inv == pytest.approx(1.0 / float(w.sum())) for _, inv, _ in full)
    # the convention is the caller's declaration, never inferred from tensor-ness: without it (and for extra_inputs
    # appended behind the declared inputs) the last input reaches the closure untouched
    marker = torch.tensor(123.0, dtype=torch.float64)
    got = []

    def plain(flat, xs, ys, tag):
        got.append(float(tag))
        return ((flat[:1] * xs).sum() - ys.sum()) ** 2
    opt2 = FirstOrderOptimizer(batch_size=b, max_epochs=1)
    opt2.update_opt(plain, target=pol)
    opt2.optimize((adv, w, marker))
    assert got and all(v == 123.0 for v in got)
    del seen[:]
    opt3 = FirstOrderOptimizer(batch_size=b, max_epochs=1)
    opt3.update_opt(lambda flat, obs, act, adv, w, inv, extra: surr(flat, obs, act, adv, w, inv) * float(extra == 5.0),
                    target=pol, weighted_mean_inputs=True)
    opt3.optimize((obs, act, adv, w, torch.tensor(1.0, dtype=torch.float64)),
                  extra_inputs=(torch.tensor(5.0, dtype=torch.float64),))
    assert all(inv == pytest.approx(1.0 / wsum) for wsum, inv, n in seen if n == b)


def test_logger_writes_only_on_the_primary_process(tmp_path, capsys):
    from rllab_amd.misc import logger
    csv_path, snap = str(tmp_path / "progress.csv"), str(tmp_path)
    try:
        logger.set_primary(False)
        logger.add_tabular_output(csv_path)
        logger.set_snapshot_dir(snap)
        logger.log("hello from a secondary rank")
        logger.record_tabular("AverageReturn", 1.5)
        assert logger.get_tabular() == {"AverageReturn": "1.5"}      # every rank still sees its row
        logger.dump_tabular()
        logger.save_itr_params(0, dict(itr=0))
        assert capsys.readouterr().out == "" and not os.path.exists(csv_path)
        assert not any(f.endswith(".pkl") for f in os.listdir(snap))
        logger.set_primary(True)
        logger.add_tabular_output(csv_path)
        logger.record_tabular("AverageReturn", 2.5)
        logger.dump_tabular()
        logger.save_itr_params(0, dict(itr=0))
        assert "AverageReturn" in capsys.readouterr().out
        assert open(csv_path).read().split() == ["AverageReturn", "2.5"]
        assert os.path.exists(os.path.join(snap, "itr_0.pkl"))
    finally:
        logger.remove_tabular_output(csv_path)
        logger.set_snapshot_dir(None)
        logger.set_primary(None)


def test_logger_snapshot_modes(tmp_path):
    from rllab_amd.misc import logger
    try:
        logger.set_snapshot_dir(str(tmp_path))
        logger.set_snapshot_mode("gap")
        logger.set_snapshot_gap(3)
        for itr in range(7):
            logger.save_itr_params(itr, dict(itr=itr))
        assert sorted(os.listdir(str(tmp_path))) == ["itr_0.pkl", "itr_3.pkl", "itr_6.pkl"]
        logger.set_snapshot_mode("last")
        logger.save_itr_params(9, dict(itr=9))
        assert "params.pkl" in os.listdir(str(tmp_path))
        logger.set_snapshot_mode("bogus")
        with pytest.raises(NotImplementedError):
            logger.save_itr_params(0, {})
    finally:
        logger.set_snapshot_mode("all")
        logger.set_snapshot_gap(1)
        logger.set_snapshot_dir(None)


def test_bench_self_launch_command():
    """`python bench.py --gpus 4 ...` re-executes itself as one rank per GPU under torch.distributed.run on
    127.0.0.1 with the user's arguments passed through."""
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.self_launch_argv(4, ["--gpus", "4", "--steps", "7", "--warmup", "2"], port=29999)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]
    free = bench.self_launch_argv(2, [])
    assert 1024 < int(free[free.index("--master-port") + 1]) < 65536
    assert len(bench.kernel_source_hash()) == 16


def test_collective_accounting_is_a_noop_without_a_process_group():
    from rllab_amd.sampler import dist as D
    D.reset_accounting()
    t = torch.ones(3)
    D.all_reduce_sum_(t)
    D.broadcast_(t)
    assert D.all_gather_rows(t).shape == (1, 3)
    assert D.accounting()["count"] == 0 and D.world_size() == 1 and D.backend() is None


def test_kernel_layout_zero_padding_is_exact():
    """policies/kernel_layout.py: a (h0, h1) policy presented to the H-wide kernels by zero padding computes the same
    mean (padded units are tanh(0) = 0 and feed nothing), pack / unpack are inverse on the real entries."""
    from rllab_amd.policies.kernel_layout import KernelLayout, tile_for
    assert tile_for((32, 32)) == 32 and tile_for((16, 8)) == 32 and tile_for((50, 25)) == 64
    assert tile_for((64, 64)) == 64 and tile_for((65, 8)) is None and tile_for((32,)) is None
    for hs in [(16, 16), (50, 25), (7, 64), (32, 32)]:
        pol = _cpu_policy(do=5, da=2, h=8)
        from rllab_amd.policies.gaussian_mlp_policy import GaussianMLPPolicy
        np.random.seed(1)
        pol = GaussianMLPPolicy(pol.env_spec if hasattr(pol, "env_spec") else pol._env_spec, hidden_sizes=hs)
        lay = KernelLayout(pol)
        H = lay.H
        theta = lay.theta()
        assert theta.numel() == 5 * H + H + H * H + H + H * 2 + 4
        if lay.exact:
            assert theta.data_ptr() == pol.flat_params.data_ptr()
            continue
        # forward pass of the padded net with plain torch == the policy's own
        o = 0
        W0 = theta[o:o + 5 * H].view(5, H); o += 5 * H
        b0 = theta[o:o + H]; o += H
        W1 = theta[o:o + H * H].view(H, H); o += H * H
        b1 = theta[o:o + H]; o += H
        W2 = theta[o:o + H * 2].view(H, 2); o += H * 2
        b2 = theta[o:o + 2]; o += 2
        ls = theta[o:o + 2]
        x = torch.randn(5, 11)
        h = torch.tanh(W1.t() @ torch.tanh(W0.t() @ x + b0[:, None]) + b1[:, None])
        assert torch.allclose(W2.t() @ h + b2[:, None], pol.mean_planes(x), atol=1e-6)
        assert torch.equal(ls, pol.effective_log_std())
        assert int((theta != 0).sum()) <= pol.flat_params.numel()
        v = torch.randn(pol.flat_params.numel(), dtype=torch.float64)
        assert torch.equal(lay.unpack(lay.pack(v)), v) and lay.pack(v).numel() == lay.P_pad
        # the padded copy follows in-place parameter updates
        with torch.no_grad():
            pol.flat_params.mul_(2.0)
        assert torch.equal(lay.unpack(lay.theta()), pol.flat_params)
        pol.note_raw_write()
        assert lay.theta() is lay._theta


def test_adaptive_std_policy_forward_layout_and_pickle():
    """GaussianMLPPolicy(adaptive_std=True): log_std = a second MLP on the observation, floored at log(min_std)
    (gaussian_mlp_policy.py:73-101); parameters = mean network then std network in Lasagne order; the reference's
    regression test constructs exactly this (tests/regression_tests/test_issue_3.py:12-29)."""
    import pickle
    from rllab_amd.core.network import MLP, rectify
    from rllab_amd.envs.env_spec import EnvSpec
    from rllab_amd.policies.gaussian_mlp_policy import GaussianMLPPolicy
    from rllab_amd.spaces import Box
    np.random.seed(3)
    spec = EnvSpec(Box(-np.ones(4), np.ones(4)), Box(-np.ones(2), np.ones(2)))
    pol = GaussianMLPPolicy(spec, hidden_sizes=(8, 6), adaptive_std=True, std_hidden_sizes=(5,), min_std=0.5,
                            std_share_network=True)
    assert pol.state_dependent_std and not pol.fusable and pol.kernel_layout() is None
    shapes = pol.get_param_shapes()
    assert shapes == [(4, 8), (8,), (8, 6), (6,), (6, 2), (2,), (4, 5), (5,), (5, 2), (2,)]
    theta = pol.get_param_values()
    parts = pol.flat_to_params(theta)
    obs = np.random.randn(7, 4)
    h = np.tanh(np.tanh(obs @ parts[0] + parts[1]) @ parts[2] + parts[3])
    mean = h @ parts[4] + parts[5]
    ls = np.maximum(np.tanh(obs @ parts[6] + parts[7]) @ parts[8] + parts[9], np.log(0.5))
    d = pol.dist_info(obs)
    np.testing.assert_allclose(d["mean"], mean, atol=1e-5)
    np.testing.assert_allclose(d["log_std"], ls, atol=1e-5)
    assert (d["log_std"] >= np.log(0.5) - 1e-7).all()
    a, info = pol.get_action(obs[0])
    assert a.shape == (2,) and info["log_std"].shape == (2,)
    acts, infos = pol.get_actions(obs)
    assert acts.shape == (7, 2) and infos["log_std"].shape == (7, 2)
    with pytest.raises(AttributeError):
        pol.effective_log_std()
    pol2 = pickle.loads(pickle.dumps(pol))
    assert np.array_equal(pol2.get_param_values(), theta) and pol2.state_dependent_std
    # custom networks: MLP descriptions whose layer sizes / nonlinearities are taken over
    pol3 = GaussianMLPPolicy(spec, mean_network=MLP((4,), 2, (9,), rectify), std_network=MLP((4,), 2, (3, 3), torch.tanh))
    assert pol3.hidden_sizes == (9,) and pol3.get_param_shapes()[-4:] == [(3, 3), (3,), (3, 2), (2,)]
    assert not pol3.fusable
