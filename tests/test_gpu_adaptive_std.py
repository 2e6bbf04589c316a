"""GaussianMLPPolicy(adaptive_std=True) -- the log-std is a second network on the observation
(rllab/policies/gaussian_mlp_policy.py:60-98; the reference's tests/regression_tests/test_issue_3.py:12-29 runs TRPO on
it) -- with loss / KL / gradient / Fisher-vector product on the HIP kernels: both networks through rl_mlp_forward /
rl_mlp_backward, the Gaussian head on planes (rl_gaussian_head / rl_gaussian_fisher).  Parity against float64 autograd
of the reference formulas (npo.py:72-82, diagonal_gaussian.py:14-69): 2e-5, Fisher-vector product 5e-5."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# (Do, Da, mean hidden, std hidden); a number H = (H, H).  The last four run (partly) on the cooperative kernels'
# OUT / OUT_TAN / BWD modes: wide and deep networks, and equal-width ones on an (obs, action) pair the one-wavefront-per-
# tile kernels are not instantiated for
SHAPES = [(4, 1, 32, 32), (13, 2, 32, 32), (13, 2, 64, 32), (20, 6, 32, 64),
          (13, 2, (128, 64, 32), 32), (20, 6, (128, 128), (128, 64)), (17, 3, 64, 32), (11, 1, (64, 32), (32, 64, 32)),
          # widths that are not tile sizes: each network zero-padded per layer (rllab's own (100, 50, 25), a narrow std net)
          (13, 2, (100, 50, 25), (20, 20)), (20, 6, (48, 48), (100, 50, 25)), (4, 1, (8, 8), (5, 7, 9)),
          # one hidden layer (round 6): the network's kernel copy is (H, H) with W1 = I, an identity layer
          (13, 2, (32, 32), (16,)), (20, 6, (100,), (16,)), (4, 1, (20,), (32, 32))]


def _policy(do, da, hm, hs, min_std=1e-6, seed=0):
    from rllab_amd.envs.env_spec import EnvSpec
    from rllab_amd.policies.gaussian_mlp_policy import GaussianMLPPolicy
    from rllab_amd.spaces import Box
    np.random.seed(seed)
    spec = EnvSpec(Box(-np.ones(do), np.ones(do)), Box(-np.ones(da), np.ones(da)))
    hm = (hm, hm) if isinstance(hm, int) else tuple(hm)
    hs = (hs, hs) if isinstance(hs, int) else tuple(hs)
    pol = GaussianMLPPolicy(spec, hidden_sizes=hm, adaptive_std=True, std_hidden_sizes=hs, min_std=min_std)
    theta = pol.get_param_values()
    theta += 0.1 * np.random.randn(theta.size)
    pol.set_param_values(theta)
    return pol


def _inputs(pol, B, seed=1, old_equals_new=False):
    rng = np.random.RandomState(seed)
    dev = pol.flat_params.device
    do, da = pol.obs_dim, pol.action_dim
    obs = torch.as_tensor(rng.randn(do, B).astype(np.float32), device=dev)
    with torch.no_grad():
        d = pol.dist_info_planes(obs.double(), pol.flat_params.double())
    if old_equals_new:
        old_mean, old_ls = d["mean"].float(), d["log_std"].float()
    else:
        old_mean = (d["mean"] + 0.05 * torch.as_tensor(rng.randn(da, B), device=dev)).float()
        old_ls = (d["log_std"] + 0.03 * torch.as_tensor(rng.randn(da, B), device=dev)).float()
    act = old_mean + torch.exp(old_ls) * torch.as_tensor(rng.randn(da, B).astype(np.float32), device=dev)
    adv = torch.as_tensor(rng.randn(B).astype(np.float32), device=dev)
    w = torch.ones(B, dtype=torch.float32, device=dev)
    w[torch.as_tensor(rng.rand(B) < 0.1, device=dev)] = 0.0
    w[0] = 1.0
    return (obs, act, adv, old_mean, old_ls, w, 1.0 / w.double().sum())


def _closures(pol):
    dist = pol.distribution

    def surr(flat, obs, act, adv, om, ols, w, inv):
        new = pol.dist_info_planes(obs.double(), flat)
        lr = dist.likelihood_ratio_sym(act.double(), dict(mean=om.double(), log_std=ols.double()), new, axis=0)
        return -(lr * adv.double() * w.double()).sum() * inv

    def kl(flat, obs, act, adv, om, ols, w, inv):
        new = pol.dist_info_planes(obs.double(), flat)
        return (dist.kl_sym(dict(mean=om.double(), log_std=ols.double()), new, axis=0) * w.double()).sum() * inv

    def vpg(flat, obs, act, adv, om, ols, w, inv):
        new = pol.dist_info_planes(obs.double(), flat)
        return -(dist.log_likelihood_sym(act.double(), new, axis=0) * adv.double() * w.double()).sum() * inv
    return surr, kl, vpg


@pytest.mark.parametrize("do,da,hm,hs", SHAPES)
@pytest.mark.parametrize("B", [63, 1000, 70001])
def test_loss_kl_grad_vs_float64_autograd(do, da, hm, hs, B):
    pol = _policy(do, da, hm, hs)
    ops = pol.fused_ops()
    assert ops is not None and type(ops).__name__ == "FusedAdaptiveStdOps"
    inp = _inputs(pol, B)
    assert ops.accepts(inp)
    surr, kl, vpg = _closures(pol)
    flat64 = pol.flat_params.detach().double().requires_grad_(True)
    l64, k64, v64 = surr(flat64, *inp), kl(flat64, *inp), vpg(flat64, *inp)
    s = ops.loss_stats(inp)
    assert abs(float(-s[0]) - float(l64.detach())) <= 2e-5 * max(1.0, abs(float(l64.detach())))
    assert abs(float(s[1]) - float(k64.detach())) <= 2e-5 * max(1e-2, abs(float(k64.detach())))
    assert abs(float(-s[2]) - float(v64.detach())) <= 2e-5 * max(1.0, abs(float(v64.detach())))
    g64 = torch.autograd.grad(l64, flat64, retain_graph=True)[0]
    g = ops.loss_grad(inp)
    assert float((g - g64).abs().max()) <= 2e-5 * max(1e-3, float(g64.abs().max()))
    n_mean = ops.nets[0][1]
    assert float(g[n_mean:].abs().max()) > 0                      # the std network gets its gradient
    gv64 = torch.autograd.grad(v64, flat64, retain_graph=True)[0]
    gv = ops.loss_grad(inp, vpg=True)
    assert float((gv - gv64).abs().max()) <= 2e-5 * max(1e-3, float(gv64.abs().max()))
    # PPO's penalised objective from the same passes
    obj = l64 + 2.5 * k64
    go64 = torch.autograd.grad(obj, flat64)[0].cpu().numpy()
    val, go = ops.value_and_grad(inp, 2.5)
    assert abs(val - float(obj.detach())) <= 2e-5 * max(1.0, abs(float(obj.detach())))
    assert np.abs(go - go64).max() <= 3e-5 * max(1e-3, np.abs(go64).max())


@pytest.mark.parametrize("do,da,hm,hs", SHAPES)
def test_fvp_equals_kl_hessian_at_theta_old(do, da, hm, hs):
    pol = _policy(do, da, hm, hs)
    ops = pol.fused_ops()
    inp = _inputs(pol, 5000, old_equals_new=True)
    _, kl, _ = _closures(pol)
    rng = np.random.RandomState(3)
    flat64 = pol.flat_params.detach().double().requires_grad_(True)
    with torch.no_grad():
        d64 = pol.dist_info_planes(inp[0].double(), flat64.detach())
    inp64 = (inp[0], inp[1], inp[2], d64["mean"], d64["log_std"], inp[5], inp[6])
    g = torch.autograd.grad(kl(flat64, *inp64), flat64, create_graph=True)[0]
    for trial in range(2):
        v = torch.as_tensor(rng.randn(flat64.numel()), device=flat64.device)
        hv64 = torch.autograd.grad((g * v).sum(), flat64, retain_graph=True)[0]
        hv = ops.fvp(inp, v)
        assert float((hv - hv64).abs().max()) <= 5e-5 * float(hv64.abs().max())


def test_min_std_floor_has_zero_derivative():
    """Where the log-std network's output is below log(min_std) the floor is active (gaussian_mlp_policy.py:100-101):
    the std network gets no gradient from those entries -- with a floor above every output, none at all."""
    pol = _policy(13, 2, 32, 32, min_std=50.0)
    ops = pol.fused_ops()
    inp = _inputs(pol, 4000)
    surr, _, _ = _closures(pol)
    flat64 = pol.flat_params.detach().double().requires_grad_(True)
    g64 = torch.autograd.grad(surr(flat64, *inp), flat64)[0]
    g = ops.loss_grad(inp)
    n_mean = ops.nets[0][1]
    assert float(g64[n_mean:].abs().max()) == 0.0 and float(g[n_mean:].abs().max()) == 0.0
    assert float((g - g64).abs().max()) <= 2e-5 * float(g64.abs().max())


def test_trpo_with_adaptive_std_updates_on_the_kernels(quiet_logger):
    """The configuration of the reference's test_issue_3 (TRPO + adaptive_std on Cartpole), several iterations: the
    optimizer runs the fused passes (device CG included) and the std network moves."""
    from rllab.algos.trpo import TRPO
    from rllab.baselines.zero_baseline import ZeroBaseline
    from rllab.envs.box2d.cartpole_env import CartpoleEnv
    from rllab.misc import ext, logger
    from rllab.policies.gaussian_mlp_policy import GaussianMLPPolicy
    ext.set_seed(4)
    env = CartpoleEnv()
    policy = GaussianMLPPolicy(env_spec=env.spec, adaptive_std=True)
    algo = TRPO(env=env, policy=policy, baseline=ZeroBaseline(env_spec=env.spec), batch_size=256 * 50,
                max_path_length=50, n_itr=3, sampler_args=dict(n_envs=256))
    algo.start_worker()
    algo.init_opt()
    assert type(algo.optimizer._fused).__name__ == "FusedAdaptiveStdOps"
    theta0 = policy.get_param_values().copy()
    for itr in range(3):
        paths = algo.sampler.obtain_samples(itr)
        sd = algo.sampler.process_samples(itr, paths)
        from rllab_amd.algos.npo import npo_inputs
        assert algo.optimizer._fused.accepts(npo_inputs(policy, sd))
        algo.optimize_policy(itr, sd)
        tab = logger.get_tabular()
        assert float(tab["MeanKL"]) <= 0.0101 and float(tab["LossAfter"]) < float(tab["LossBefore"])
        logger.dump_tabular()
    moved = np.abs(policy.get_param_values() - theta0)
    n_mean = algo.optimizer._fused.nets[0][1]
    assert moved[:n_mean].max() > 0 and moved[n_mean:].max() > 0


def test_trpo_with_adaptive_std_and_free_form_widths_stays_on_the_kernels(quiet_logger):
    """rllab's (100, 50, 25) mean network with adaptive_std (std net (32, 32) by default): both networks are zero-padded
    per layer for the kernels -- the rollout is the fused one, the update runs the fused passes, MeanKL holds."""
    from rllab.algos.trpo import TRPO
    from rllab.baselines.linear_feature_baseline import LinearFeatureBaseline
    from rllab.envs.mujoco.swimmer_env import SwimmerEnv
    from rllab.envs.normalized_env import normalize
    from rllab.misc import ext, logger
    from rllab.policies.gaussian_mlp_policy import GaussianMLPPolicy
    ext.set_seed(5)
    env = normalize(SwimmerEnv())
    policy = GaussianMLPPolicy(env_spec=env.spec, hidden_sizes=(100, 50, 25), adaptive_std=True)
    algo = TRPO(env=env, policy=policy, baseline=LinearFeatureBaseline(env_spec=env.spec), batch_size=256 * 60,
                max_path_length=60, n_itr=3, sampler_args=dict(n_envs=256))
    algo.start_worker()
    algo.init_opt()
    assert type(algo.optimizer._fused).__name__ == "FusedAdaptiveStdOps"
    assert algo.sampler._takes_fused_rollout(policy)
    assert [h for _, _, h in algo.optimizer._fused.nets] == [(128, 64, 32), (32, 32, 0)]
    theta0 = policy.get_param_values().copy()
    for itr in range(3):
        paths = algo.sampler.obtain_samples(itr)
        sd = algo.sampler.process_samples(itr, paths)
        algo.log_diagnostics(paths)
        algo.optimize_policy(itr, sd)
        tab = logger.get_tabular()
        assert float(tab["MeanKL"]) <= 0.0101 and float(tab["LossAfter"]) < float(tab["LossBefore"])
        assert abs(float(tab["MeanKLBefore"])) < 1e-6
        logger.dump_tabular()
    assert np.isfinite(policy.get_param_values()).all() and np.abs(policy.get_param_values() - theta0).max() > 0


@pytest.mark.parametrize("kind,hm,hs", [(0, (32, 32), (32, 32)), (2, (64, 64), (32, 32)), (3, (128, 64, 32), (32, 32)),
                                        (6, (32, 32), (64, 64)), (2, (100, 50, 25), (20, 20)), (0, (8, 8), (5, 7, 9)),
                                        # one hidden layer (round 6): (H, H) with the identity as second layer
                                        (2, (32, 32), (16,)), (3, (100,), (16,)), (0, (20,), (32, 32))])
@pytest.mark.parametrize("epw", ["16", "64"])
def test_fused_rollout_with_a_log_std_network(kind, hm, hs, epw, monkeypatch):
    """rl_rollout_gaussian_mlp with rl_rollout_args.theta_std: mean AND log-std network evaluated in the kernel every
    step (gaussian_mlp_policy.py:60-98,132-137), the floored log-stds recorded as agent_info.  Env dynamics replayed on
    the host build bit for bit; means and log-stds against a float64 torch forward of the two networks;
    action == mean + eps * exp(log_std)."""
    from rllab_amd import _lib
    from rllab_amd.envs.env_spec import EnvSpec
    from rllab_amd.envs.hip_env import HipVecEnv
    from rllab_amd.policies.gaussian_mlp_policy import GaussianMLPPolicy
    from rllab_amd.spaces import Box
    from oracle.replay import replay_check
    monkeypatch.setenv("RLLAB_ROLLOUT_EPW", epw)
    q = _lib.env_query(kind)
    np.random.seed(2)
    spec = EnvSpec(Box(-1e6 * np.ones(q["obs_dim"]), 1e6 * np.ones(q["obs_dim"])),
                   Box(-np.ones(q["act_dim"]), np.ones(q["act_dim"])))
    policy = GaussianMLPPolicy(spec, hidden_sizes=hm, adaptive_std=True, std_hidden_sizes=hs, min_std=1.0)
    theta = policy.get_param_values()
    policy.set_param_values(theta + 0.2 * np.random.randn(theta.size))      # log-stds on both sides of the floor
    assert policy.kernel_layout() is None and policy.rollout_networks() is not None
    rng = np.random.RandomState(1)
    n, T, mpl = 130, 40, 17
    v = HipVecEnv(kind, n, mpl, normalize=True, seed=11)
    eps = rng.randn(q["act_dim"], T, n).astype(np.float32)
    draws = (rng.randn if q["reset_is_normal"] else rng.rand)(T + 1, q["reset_draws"], n).astype(np.float32)
    traj = v.rollout(policy, T, reset_at_start=True, eps=eps, reset_draws=draws)
    torch.cuda.synchronize()
    assert traj.log_std is None and traj.log_std_planes.shape == (q["act_dim"], T, n)
    assert replay_check(v, traj, max_envs=n, reset_draws=draws) == n * T
    obs64 = traj.obs.reshape(q["obs_dim"], -1).double()
    with torch.no_grad():
        d64 = policy.dist_info_planes(obs64, policy.flat_params.double())
    got_m = traj.means.reshape(q["act_dim"], -1).double()
    got_ls = traj.log_std_planes.reshape(q["act_dim"], -1).double()
    assert float((got_m - d64["mean"]).abs().max()) <= 2e-5
    assert float((got_ls - d64["log_std"]).abs().max()) <= 2e-5
    at_floor = float((got_ls == 0.0).double().mean())           # min_std = 1: the floor log(min_std) is exactly 0
    assert 0.0 < at_floor < 1.0 and float(got_ls.min()) >= 0.0      # the floor is active somewhere, not everywhere
    act64 = got_m + torch.as_tensor(eps, device=got_m.device).reshape(q["act_dim"], -1).double() * torch.exp(got_ls)
    # (the kernel's exp is the hardware's: a few ulp of a std of up to ~3, times |eps| up to ~4)
    err = (traj.actions.reshape(q["act_dim"], -1).double() - act64).abs() / act64.abs().clamp_min(1.0)
    assert float(err.max()) <= 1e-5


def test_adaptive_std_is_sampled_by_the_fused_rollout(quiet_logger):
    """TRPO + adaptive_std end to end with NOTHING left to the per-transition loop: one rollout launch per iteration."""
    from rllab.algos.trpo import TRPO
    from rllab.baselines.linear_feature_baseline import LinearFeatureBaseline
    from rllab.envs.box2d.cartpole_env import CartpoleEnv
    from rllab.envs.normalized_env import normalize
    from rllab.misc import ext, logger
    from rllab.policies.gaussian_mlp_policy import GaussianMLPPolicy
    ext.set_seed(6)
    env = normalize(CartpoleEnv())
    policy = GaussianMLPPolicy(env_spec=env.spec, adaptive_std=True)
    algo = TRPO(env=env, policy=policy, baseline=LinearFeatureBaseline(env_spec=env.spec), batch_size=512 * 100,
                max_path_length=100, n_itr=12, sampler_args=dict(n_envs=512))
    algo.start_worker()
    algo.init_opt()
    assert algo.sampler._takes_fused_rollout(policy) and type(algo.optimizer._fused).__name__ == "FusedAdaptiveStdOps"
    rets = []
    for itr in range(12):
        paths = algo.sampler.obtain_samples(itr)
        assert getattr(algo.sampler, "_step_graph", None) is None               # the hipGraph loop was never built
        sd = algo.sampler.process_samples(itr, paths)
        algo.log_diagnostics(paths)
        algo.optimize_policy(itr, sd)
        tab = logger.get_tabular()
        rets.append(float(tab["AverageReturn"]))
        assert float(tab["MeanKL"]) <= 0.0101 and np.isfinite(float(tab["Entropy"]))
        assert 0.0 < float(tab["AveragePolicyStd"]) < 10.0
        logger.dump_tabular()
    assert np.mean(rets[-3:]) > 1.5 * np.mean(rets[:3]), rets


def test_two_wide_networks_that_do_not_fit_lds_are_sampled_stepwise_and_updated_on_the_kernels(quiet_logger):
    """GaussianMLPPolicy(adaptive_std=True) with a (128, 128) mean net and a (128, 128) log-std net on a 20-observation
    env: the fused rollout would need 172 KB of LDS for the two networks' weight fragments (a CU has 160) -- the sampler asks
    the library (rl_rollout_lds_bytes) and takes the per-transition loop instead of failing; the update still runs on the
    kernels (both networks through the cooperative OUT / OUT_TAN / BWD modes)."""
    from rllab.algos.trpo import TRPO
    from rllab.baselines.linear_feature_baseline import LinearFeatureBaseline
    from rllab.envs.mujoco.hopper_env import HopperEnv
    from rllab.envs.normalized_env import normalize
    from rllab.misc import ext, logger
    from rllab.policies.gaussian_mlp_policy import GaussianMLPPolicy
    ext.set_seed(8)
    env = normalize(HopperEnv())
    policy = GaussianMLPPolicy(env_spec=env.spec, hidden_sizes=(128, 128), adaptive_std=True, std_hidden_sizes=(128, 128))
    algo = TRPO(env=env, policy=policy, baseline=LinearFeatureBaseline(env_spec=env.spec), batch_size=256 * 40,
                max_path_length=40, n_itr=2, sampler_args=dict(n_envs=256))
    algo.start_worker()
    algo.init_opt()
    assert policy.rollout_networks() is not None                       # a kernel shape ...
    assert not algo.sampler._takes_fused_rollout(policy)               # ... that does not fit the LDS of a CU on this env
    assert type(algo.optimizer._fused).__name__ == "FusedAdaptiveStdOps"
    theta0 = policy.get_param_values().copy()
    for itr in range(2):
        paths = algo.sampler.obtain_samples(itr)
        sd = algo.sampler.process_samples(itr, paths)
        algo.optimize_policy(itr, sd)
        tab = logger.get_tabular()
        assert float(tab["MeanKL"]) <= 0.0101 and np.isfinite(float(tab["LossAfter"]))
        logger.dump_tabular()
    assert np.abs(policy.get_param_values() - theta0).max() > 0
    # and a pair of networks that does fit keeps the one-launch rollout
    small = GaussianMLPPolicy(env_spec=env.spec, hidden_sizes=(64, 64), adaptive_std=True, std_hidden_sizes=(32, 32))
    assert algo.sampler.vec_env.takes_rollout_of(small)
