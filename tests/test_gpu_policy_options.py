"""GaussianMLPPolicy(hidden_sizes=...) is free-form in the reference (gaussian_mlp_policy.py:24): two tanh hidden
layers of any sizes up to 64 stay on the HIP kernels by zero padding (policies/kernel_layout.py); adaptive_std
(:60-98, tests/regression_tests/test_issue_3.py:12-29) and NormalizedEnv(normalize_obs / normalize_reward)
(normalized_env.py:33-49) run through the per-transition sampler."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HIDDEN = [(16, 16), (50, 25), (8, 64), (64, 32), (33, 33)]


def _policy(kind, hidden, seed=0):
    from tests.test_gpu_env_parity import _make_policy
    pol = _make_policy(kind, hidden, seed)
    theta = pol.get_param_values()
    theta += 0.1 * np.random.RandomState(seed).randn(theta.size)
    pol.set_param_values(theta)
    return pol


@pytest.mark.parametrize("hidden", HIDDEN)
@pytest.mark.parametrize("kind", [0, 2, 3])
def test_fused_rollout_with_padded_hidden_sizes(kind, hidden):
    """The fused rollout on the padded layout: recorded means == float64 torch forward of the REAL (unpadded) net,
    env dynamics replay bit-exactly on the host."""
    from rllab_amd.envs.hip_env import HipVecEnv
    from oracle.replay import replay_check
    pol = _policy(kind, hidden)
    lay = pol.kernel_layout()
    assert lay is not None and not lay.exact and lay.H == (32 if max(hidden) <= 32 else 64)
    rng = np.random.RandomState(1)
    n, T = 70, 25
    v = HipVecEnv(kind, n, 11, normalize=True, seed=5)
    q = v.q
    eps = rng.randn(q["act_dim"], T, n).astype(np.float32)
    draws = (rng.randn if q["reset_is_normal"] else rng.rand)(T + 1, q["reset_draws"], n).astype(np.float32)
    traj = v.rollout(pol, T, reset_at_start=True, eps=eps, reset_draws=draws)
    assert replay_check(v, traj, max_envs=n, reset_draws=draws) == n * T
    with torch.no_grad():
        mean64 = pol.mean_planes(traj.obs.reshape(q["obs_dim"], -1).double(), pol.flat_params.double())
    assert float((traj.means.reshape(q["act_dim"], -1).double() - mean64).abs().max()) <= 1e-5


@pytest.mark.parametrize("hidden", HIDDEN)
@pytest.mark.parametrize("do,da", [(4, 1), (13, 2), (20, 6)])
def test_update_kernels_with_padded_hidden_sizes(do, da, hidden):
    """loss / KL / gradient / Fisher-vector product of the fused kernels on the padded layout against float64
    autograd of the real net (same bars as the exact-size tests)."""
    from tests import test_gpu_update_parity as U
    from rllab_amd.envs.env_spec import EnvSpec
    from rllab_amd.policies.gaussian_mlp_policy import GaussianMLPPolicy
    from rllab_amd.spaces import Box
    np.random.seed(2)
    spec = EnvSpec(Box(-np.ones(do), np.ones(do)), Box(-np.ones(da), np.ones(da)))
    pol = GaussianMLPPolicy(spec, hidden_sizes=hidden)
    th = pol.get_param_values()
    pol.set_param_values(th + 0.1 * np.random.randn(th.size))
    ops = pol.fused_ops()
    assert ops is not None and ops.n_kernel > pol.flat_params.numel()
    inp = U._inputs(pol, 3001)
    surr, kl, vpg = U._closures(pol)
    flat64 = pol.flat_params.detach().double().requires_grad_(True)
    l64, k64 = surr(flat64, *inp), kl(flat64, *inp)
    s = ops.loss_stats(inp)
    assert abs(float(-s[0]) - float(l64)) <= 2e-5 * max(1.0, abs(float(l64)))
    assert abs(float(s[1]) - float(k64)) <= 2e-5 * max(1e-2, abs(float(k64)))
    g64 = torch.autograd.grad(l64, flat64)[0]
    g = ops.loss_grad(inp)
    assert g.shape == g64.shape
    assert float((g - g64).abs().max()) <= 2e-5 * max(1e-3, float(g64.abs().max()))
    # Fisher-vector product at theta_old == theta_new
    inp0 = U._inputs(pol, 3001, old_equals_new=True)
    flat64 = pol.flat_params.detach().double().requires_grad_(True)
    with torch.no_grad():
        om64 = pol.mean_planes(inp0[0].double(), flat64.detach())
    inp64 = (inp0[0], inp0[1], inp0[2], om64, pol.effective_log_std().detach().double().reshape(-1, 1), inp0[5], inp0[6])
    gk = torch.autograd.grad(kl(flat64, *inp64), flat64, create_graph=True)[0]
    v = torch.as_tensor(np.random.RandomState(3).randn(flat64.numel()), device=flat64.device)
    hv64 = torch.autograd.grad((gk * v).sum(), flat64)[0]
    hv = ops.fvp(inp0, v)
    assert float((hv - hv64).abs().max()) <= 5e-5 * float(hv64.abs().max())
    # the parameter copy of the kernels follows the line search's raw writes
    prev = pol.flat_params.detach().clone()
    step = torch.ones_like(prev, dtype=torch.float64) * 1e-3
    ops.line_search_point(prev, step, 1.0)
    l_new = float(surr(pol.flat_params.detach().double(), *inp))
    assert abs(float(-ops.loss_stats(inp)[0]) - l_new) <= 2e-5 * max(1.0, abs(l_new))


@pytest.mark.parametrize("hidden", [(16, 16), (50, 25)])
def test_trpo_runs_fused_with_free_form_hidden_sizes(hidden, quiet_logger):
    from rllab.algos.trpo import TRPO
    from rllab.baselines.linear_feature_baseline import LinearFeatureBaseline
    from rllab.envs.box2d.cartpole_env import CartpoleEnv
    from rllab.envs.normalized_env import normalize
    from rllab.misc import ext, logger
    from rllab.policies.gaussian_mlp_policy import GaussianMLPPolicy
    ext.set_seed(1)
    env = normalize(CartpoleEnv())
    policy = GaussianMLPPolicy(env_spec=env.spec, hidden_sizes=hidden)
    algo = TRPO(env=env, policy=policy, baseline=LinearFeatureBaseline(env_spec=env.spec), batch_size=256 * 100,
                max_path_length=100, n_itr=15, discount=0.99, step_size=0.01, sampler_args=dict(n_envs=256))
    algo.start_worker()
    algo.init_opt()
    assert algo.optimizer._fused is not None                  # the HIP update path, not autograd
    rets = []
    for itr in range(15):
        paths = algo.sampler.obtain_samples(itr)
        assert paths.traj.log_std is not None and paths.traj.log_std_planes is None     # fused rollout
        sd = algo.sampler.process_samples(itr, paths)
        algo.log_diagnostics(paths)
        algo.optimize_policy(itr, sd)
        tab = logger.get_tabular()
        rets.append(float(tab["AverageReturn"]))
        assert float(tab["MeanKL"]) <= 0.0101
        logger.dump_tabular()
    assert np.mean(rets[-3:]) > 2.0 * np.mean(rets[:3]), rets


def test_adaptive_std_policy_trains(quiet_logger):
    """tests/regression_tests/test_issue_3.py of the reference (TRPO + GaussianMLPPolicy(adaptive_std=True) +
    ZeroBaseline on CartpoleEnv, batch 100, one iteration), then a longer run that must learn."""
    from rllab.algos.trpo import TRPO
    from rllab.baselines.zero_baseline import ZeroBaseline
    from rllab.envs.box2d.cartpole_env import CartpoleEnv
    from rllab.misc import ext, logger
    from rllab.policies.gaussian_mlp_policy import GaussianMLPPolicy
    ext.set_seed(2)
    env = CartpoleEnv()
    policy = GaussianMLPPolicy(env_spec=env.spec, adaptive_std=True)
    before = policy.get_param_values()
    algo = TRPO(env=env, policy=policy, baseline=ZeroBaseline(env_spec=env.spec), batch_size=100, n_itr=1)
    algo.train()
    after = policy.get_param_values()
    assert np.isfinite(after).all() and np.abs(after - before).max() > 0
    n_mean = sum(int(np.prod(s)) for s in policy.get_param_shapes()[:6])
    assert np.abs(after[n_mean:] - before[n_mean:]).max() > 0          # the std network moved too
    from rllab.envs.normalized_env import normalize
    env = normalize(CartpoleEnv())
    policy = GaussianMLPPolicy(env_spec=env.spec, adaptive_std=True, std_hidden_sizes=(16, 16))
    algo = TRPO(env=env, policy=policy, baseline=ZeroBaseline(env_spec=env.spec), batch_size=128 * 100,
                max_path_length=100, n_itr=12, discount=0.99, step_size=0.01, sampler_args=dict(n_envs=128))
    algo.start_worker()
    algo.init_opt()
    rets = []
    for itr in range(12):
        paths = algo.sampler.obtain_samples(itr)
        assert paths.traj.log_std_planes is not None                    # per-sample agent_info["log_std"]
        sd = algo.sampler.process_samples(itr, paths)
        assert sd["agent_infos"]["log_std"].shape == sd["agent_infos"]["mean"].shape
        algo.log_diagnostics(paths)
        algo.optimize_policy(itr, sd)
        tab = logger.get_tabular()
        rets.append(float(tab["AverageReturn"]))
        assert float(tab["MeanKL"]) <= 0.0101 and np.isfinite(float(tab["Entropy"]))
        assert 0.0 < float(tab["AveragePolicyStd"]) < 10.0
        logger.dump_tabular()
    assert np.mean(rets[-3:]) > 1.5 * np.mean(rets[:3]), rets


def test_sampler_takes_running_normalisation_through_the_transition_loop(quiet_logger):
    """NormalizedEnv(normalize_obs=True, normalize_reward=True) keeps ``env.vectorized``; its arithmetic is pinned to the
    reference's own classes in tests/test_gpu_reference_vecenv.py.  Here: with obs_noise-free Cartpole and a policy that
    has a fused kernel the sampler uses the fused rollout; a policy without one goes through reset() / step()."""
    from rllab.envs.box2d.cartpole_env import CartpoleEnv
    from rllab.envs.normalized_env import normalize
    env = normalize(CartpoleEnv(), scale_reward=0.1, normalize_obs=True, normalize_reward=True, obs_alpha=0.01,
                    reward_alpha=0.02)
    assert env.vectorized
    from rllab.algos.vpg import VPG
    from rllab.baselines.zero_baseline import ZeroBaseline
    from rllab.policies.gaussian_mlp_policy import GaussianMLPPolicy
    for hidden, fused in (((32, 32), True), ((100, 50, 25), False)):
        policy = GaussianMLPPolicy(env_spec=env.spec, hidden_sizes=hidden)
        algo = VPG(env=env, policy=policy, baseline=ZeroBaseline(env_spec=env.spec), batch_size=32 * 30,
                   max_path_length=30, n_itr=2, sampler_args=dict(n_envs=32))
        algo.start_worker()
        assert algo.sampler.sampling_path(policy)[0].startswith("fused rollout kernel") == fused
        algo.shutdown_worker()
        algo.train()
        assert np.isfinite(policy.get_param_values()).all()


@pytest.mark.parametrize("do,da,h", [(4, 1, 32), (13, 2, 32), (20, 6, 64)])
@pytest.mark.parametrize("penalty", [0.0, 2.5])
def test_penalised_surrogate_value_and_gradient(do, da, h, penalty):
    """PenaltyLbfgsOptimizer's objective on a policy (PPO, rllab/algos/ppo.py:8-22): surrogate loss + penalty * mean KL
    and its gradient from ONE kernel pass (rl_policy_grad_loss with kl_penalty) against float64 autograd."""
    from tests import test_gpu_update_parity as U
    pol = U._policy(do, da, h)
    ops = pol.fused_ops()
    inp = U._inputs(pol, 5003)
    surr, kl, _ = U._closures(pol)
    flat64 = pol.flat_params.detach().double().requires_grad_(True)
    obj = surr(flat64, *inp) + penalty * kl(flat64, *inp)
    g64 = torch.autograd.grad(obj, flat64)[0].cpu().numpy()
    val, g = ops.value_and_grad(inp, penalty)
    assert abs(val - float(obj.detach())) <= 2e-5 * max(1.0, abs(float(obj.detach())))
    assert np.abs(g - g64).max() <= 3e-5 * max(1e-3, np.abs(g64).max())


def test_ppo_runs_on_the_kernels_and_learns(quiet_logger):
    from rllab.algos.ppo import PPO
    from rllab.baselines.linear_feature_baseline import LinearFeatureBaseline
    from rllab.envs.box2d.cartpole_env import CartpoleEnv
    from rllab.envs.normalized_env import normalize
    from rllab.misc import ext, logger
    from rllab.policies.gaussian_mlp_policy import GaussianMLPPolicy
    ext.set_seed(5)
    env = normalize(CartpoleEnv())
    policy = GaussianMLPPolicy(env_spec=env.spec, hidden_sizes=(32, 32))
    algo = PPO(env=env, policy=policy, baseline=LinearFeatureBaseline(env_spec=env.spec), batch_size=256 * 100,
               max_path_length=100, n_itr=10, discount=0.99, step_size=0.01, sampler_args=dict(n_envs=256))
    algo.start_worker()
    algo.init_opt()
    assert algo.optimizer._fused is not None and hasattr(algo.optimizer._fused, "value_and_grad")
    rets = []
    for itr in range(10):
        paths = algo.sampler.obtain_samples(itr)
        sd = algo.sampler.process_samples(itr, paths)
        algo.log_diagnostics(paths)
        algo.optimize_policy(itr, sd)
        tab = logger.get_tabular()
        rets.append(float(tab["AverageReturn"]))
        assert float(tab["MeanKL"]) <= 0.0101 and float(tab["LossAfter"]) <= float(tab["LossBefore"]) + 1e-7
        logger.dump_tabular()
    assert np.mean(rets[-2:]) > 1.5 * np.mean(rets[:2]), rets


# ---- one hidden layer, rectify layers (round 5): the equal-width two-layer kernels with per-layer activation codes ----------
# gaussian_mlp_policy.py:21-69 / network.py:36-101 take any hidden_sizes and hidden_nonlinearity; a one-hidden-layer policy's
# kernel copy is the two-layer net with an identity second layer (policies/kernel_layout.py)
def _nl(name):
    from rllab_amd.core.network import rectify
    return dict(tanh=torch.tanh, rectify=rectify, torch_relu=torch.relu)[name]


LAYERED = [((32,), "tanh"), ((20,), "tanh"), ((64,), "tanh"), ((32, 32), "rectify"), ((16,), "rectify"),
           ((50, 25), "torch_relu"),
           ((100,), "tanh"), ((128,), "tanh")]     # round 6: 65 .. 128 units on the cooperative family's (128, 128) shape


def _layered_policy(do_or_kind, da, hidden, nl, seed=0, by_kind=False):
    from rllab_amd import _lib
    from rllab_amd.envs.env_spec import EnvSpec
    from rllab_amd.policies.gaussian_mlp_policy import GaussianMLPPolicy
    from rllab_amd.spaces import Box
    if by_kind:
        q = _lib.env_query(do_or_kind)
        do, da = q["obs_dim"], q["act_dim"]
    else:
        do = do_or_kind
    np.random.seed(seed)
    spec = EnvSpec(Box(-1e6 * np.ones(do), 1e6 * np.ones(do)), Box(-np.ones(da), np.ones(da)))
    pol = GaussianMLPPolicy(spec, hidden_sizes=hidden, hidden_nonlinearity=_nl(nl))
    th = pol.get_param_values()
    pol.set_param_values(th + 0.1 * np.random.RandomState(seed).randn(th.size))
    return pol


@pytest.mark.parametrize("hidden,nl", LAYERED)
@pytest.mark.parametrize("kind", [0, 2, 3])
def test_fused_rollout_with_one_hidden_layer_or_rectify_layers(kind, hidden, nl, rollout_shape_all):
    """Every launch shape of the equal-width rollouts: recorded means == float64 torch forward of the REAL net (one layer /
    rectify), env dynamics replay bit-exactly on the host, and the launcher's plan says it is a fused kernel."""
    from rllab_amd.envs.hip_env import HipVecEnv
    from oracle.replay import replay_check
    pol = _layered_policy(kind, None, hidden, nl, by_kind=True)
    lay = pol.kernel_layout()
    assert lay is not None and lay.layer_activations != 0
    assert lay.identity_layer == (len(hidden) == 1)
    rng = np.random.RandomState(1)
    n, T = 70, 25
    v = HipVecEnv(kind, n, 11, normalize=True, seed=5)
    plan = v.rollout_plan(pol, T)
    assert plan is not None and plan.kernel in ((1, 4, 7, 8) if max(hidden) <= 64 else (2, 5, 6, 9)), plan and plan.kernel
    q = v.q
    eps = rng.randn(q["act_dim"], T, n).astype(np.float32)
    draws = (rng.randn if q["reset_is_normal"] else rng.rand)(T + 1, q["reset_draws"], n).astype(np.float32)
    traj = v.rollout(pol, T, reset_at_start=True, eps=eps, reset_draws=draws)
    assert replay_check(v, traj, max_envs=n, reset_draws=draws) == n * T
    with torch.no_grad():
        mean64 = pol.mean_planes(traj.obs.reshape(q["obs_dim"], -1).double(), pol.flat_params.double())
    assert float((traj.means.reshape(q["act_dim"], -1).double() - mean64).abs().max()) <= 1e-5


@pytest.fixture(params=["auto", "64", "generic16"])
def rollout_shape_all(request, monkeypatch):
    for k in ("RLLAB_ROLLOUT_EPW", "RLLAB_SWIMMER_LANE_KERNEL", "RLLAB_TWO_LEG_LANE_KERNEL"):
        monkeypatch.delenv(k, raising=False)
    if request.param == "64":
        monkeypatch.setenv("RLLAB_ROLLOUT_EPW", "64")
        monkeypatch.setenv("RLLAB_SWIMMER_LANE_KERNEL", "1")
    elif request.param == "generic16":
        monkeypatch.setenv("RLLAB_ROLLOUT_EPW", "16")
        monkeypatch.setenv("RLLAB_SWIMMER_LANE_KERNEL", "1")
    return request.param


@pytest.mark.parametrize("hidden,nl", LAYERED)
@pytest.mark.parametrize("do,da", [(4, 1), (13, 2), (20, 6)])
def test_update_kernels_with_one_hidden_layer_or_rectify_layers(do, da, hidden, nl):
    """loss / KL / gradients / Fisher-vector product of the kernels (per-layer activation codes) against float64 autograd of
    the real net -- the bars of the tanh tests; the product runs on the f32 matrix instructions (the split-operand kernels
    take tanh layers), with and without the activation cache."""
    from tests import test_gpu_update_parity as U
    pol = _layered_policy(do, da, hidden, nl, seed=2)
    ops = pol.fused_ops()
    assert ops is not None and ops.layout.layer_activations != 0
    inp = U._inputs(pol, 3008)
    surr, kl, vpg = U._closures(pol)
    flat64 = pol.flat_params.detach().double().requires_grad_(True)
    l64, k64, v64 = surr(flat64, *inp), kl(flat64, *inp), vpg(flat64, *inp)
    s = ops.loss_stats(inp)
    assert abs(float(-s[0]) - float(l64)) <= 2e-5 * max(1.0, abs(float(l64)))
    assert abs(float(s[1]) - float(k64)) <= 2e-5 * max(1e-2, abs(float(k64)))
    g64 = torch.autograd.grad(l64, flat64, retain_graph=True)[0]
    g = ops.loss_grad(inp)
    assert g.shape == g64.shape
    assert float((g - g64).abs().max()) <= 2e-5 * max(1e-3, float(g64.abs().max()))
    gv64 = torch.autograd.grad(v64, flat64)[0]
    gv = ops.loss_grad(inp, vpg=True)
    assert float((gv - gv64).abs().max()) <= 2e-5 * max(1e-3, float(gv64.abs().max()))
    inp0 = U._inputs(pol, 3008, old_equals_new=True)
    flat64 = pol.flat_params.detach().double().requires_grad_(True)
    with torch.no_grad():
        om64 = pol.mean_planes(inp0[0].double(), flat64.detach())
    inp64 = (inp0[0], inp0[1], inp0[2], om64, pol.effective_log_std().detach().double().reshape(-1, 1), inp0[5], inp0[6])
    gk = torch.autograd.grad(kl(flat64, *inp64), flat64, create_graph=True)[0]
    v = torch.as_tensor(np.random.RandomState(3).randn(flat64.numel()), device=flat64.device)
    hv64 = torch.autograd.grad((gk * v).sum(), flat64)[0]
    hv = ops.fvp(inp0, v)
    assert float((hv - hv64).abs().max()) <= 5e-5 * float(hv64.abs().max())
    ops.loss_grad(inp0, keep_activations=True)                  # ... and from the cached activations (whole tiles: 3008 = 94 x 32)
    assert ops.fvp_variant(inp0) == 0                           # not the split kernels: they evaluate tanh layers
    hv_c = ops.fvp(inp0, v)
    assert torch.equal(hv_c, hv)
    # the device CG (products in the kernels' vector, where a one-hidden-layer policy's identity layer W1 = I sits as
    # constants) == krylov.cg on the same product in the REAL parameter space: the constants' Fisher rows stay out of it
    from rllab_amd.misc import krylov
    g = ops.loss_grad(inp0)
    x_dev, _ = ops.cg(inp0, g, 10, 1e-5)
    x_ref = krylov.cg(lambda p_: ops.fvp(inp0, p_) + 1e-5 * p_, g, cg_iters=10)
    assert float((x_dev - x_ref).abs().max()) <= 1e-6 * float(x_ref.abs().max()), float((x_dev - x_ref).abs().max())


def test_policies_that_still_leave_the_kernels_say_why():
    pol = _layered_policy(13, 2, (200,), "tanh")
    assert pol.kernel_layout() is None and "one hidden layer" in pol.why_no_kernel_layout()
    pol = _layered_policy(13, 2, (100,), "rectify")                 # (rectify layers: the equal-width kernels only)
    assert pol.kernel_layout() is None
    pol = _layered_policy(17, 2, (32,), "tanh")                     # not an (obs, action) pair of a HIP-native env
    assert pol.kernel_layout() is None and "pairs" in pol.why_no_kernel_layout()
    pol = _layered_policy(13, 2, (128, 128), "rectify")
    assert pol.kernel_layout() is None and "rectify" in pol.why_no_kernel_layout()
    from rllab_amd.policies.gaussian_mlp_policy import GaussianMLPPolicy
    from rllab_amd.envs.env_spec import EnvSpec
    from rllab_amd.spaces import Box
    spec = EnvSpec(Box(-np.ones(13), np.ones(13)), Box(-np.ones(2), np.ones(2)))
    pol = GaussianMLPPolicy(spec, hidden_sizes=(32, 32), hidden_nonlinearity=torch.sigmoid)
    assert pol.kernel_layout() is None and "sigmoid" in pol.why_no_kernel_layout()


@pytest.mark.parametrize("hidden,nl", [((32,), "tanh"), ((32, 32), "rectify"), ((100,), "tanh")])
def test_trpo_learns_on_the_kernels_with_one_hidden_layer_or_rectify(hidden, nl, quiet_logger):
    from rllab.algos.trpo import TRPO
    from rllab.baselines.linear_feature_baseline import LinearFeatureBaseline
    from rllab.envs.box2d.cartpole_env import CartpoleEnv
    from rllab.envs.normalized_env import normalize
    from rllab.misc import ext, logger
    from rllab.policies.gaussian_mlp_policy import GaussianMLPPolicy
    ext.set_seed(1)
    env = normalize(CartpoleEnv())
    policy = GaussianMLPPolicy(env_spec=env.spec, hidden_sizes=hidden, hidden_nonlinearity=_nl(nl))
    algo = TRPO(env=env, policy=policy, baseline=LinearFeatureBaseline(env_spec=env.spec), batch_size=256 * 100,
                max_path_length=100, n_itr=15, discount=0.99, step_size=0.01, sampler_args=dict(n_envs=256))
    algo.start_worker()
    algo.init_opt()
    assert algo.optimizer._fused is not None                  # the HIP update path, not autograd
    assert algo.sampler.sampling_path(policy)[0].startswith("fused rollout kernel")
    rets = []
    for itr in range(15):
        paths = algo.sampler.obtain_samples(itr)
        assert paths.traj.log_std is not None and paths.traj.log_std_planes is None     # fused rollout
        sd = algo.sampler.process_samples(itr, paths)
        algo.log_diagnostics(paths)
        algo.optimize_policy(itr, sd)
        tab = logger.get_tabular()
        rets.append(float(tab["AverageReturn"]))
        assert float(tab["MeanKL"]) <= 0.0101
        logger.dump_tabular()
    assert np.mean(rets[-3:]) > 2.0 * np.mean(rets[:3]), rets


# ---- NormalizedEnv(normalize_obs / normalize_reward) inside the fused rollout (round 5: rl_running_norm) -----------------
def test_trpo_with_running_normalisation_stays_on_the_fused_rollout(quiet_logger):
    from rllab.algos.trpo import TRPO
    from rllab.baselines.linear_feature_baseline import LinearFeatureBaseline
    from rllab.envs.box2d.cartpole_env import CartpoleEnv
    from rllab.envs.normalized_env import normalize
    from rllab.misc import ext, logger
    from rllab.policies.gaussian_mlp_policy import GaussianMLPPolicy
    ext.set_seed(1)
    env = normalize(CartpoleEnv(), normalize_obs=True, normalize_reward=True)
    policy = GaussianMLPPolicy(env_spec=env.spec, hidden_sizes=(32, 32))
    algo = TRPO(env=env, policy=policy, baseline=LinearFeatureBaseline(env_spec=env.spec), batch_size=256 * 100,
                max_path_length=100, n_itr=12, discount=0.99, step_size=0.01, sampler_args=dict(n_envs=256))
    algo.start_worker()
    algo.init_opt()
    assert algo.sampler.sampling_path(policy)[0].startswith("fused rollout kernel")
    lens = []
    for itr in range(12):
        paths = algo.sampler.obtain_samples(itr)
        sd = algo.sampler.process_samples(itr, paths)
        algo.log_diagnostics(paths)
        algo.optimize_policy(itr, sd)
        tab = logger.get_tabular()
        lens.append(256 * 100 / float(tab["NumTrajs"]))          # rewards are whitened: episode length is the progress measure
        assert float(tab["MeanKL"]) <= 0.0101
        logger.dump_tabular()
    assert np.mean(lens[-3:]) > 1.5 * np.mean(lens[:3]), lens
    # the estimates moved, and they travel with the env (snapshot semantics of round 3)
    v = algo.sampler.vec_env
    assert float((v.obs_var - 1.0).abs().max()) > 1e-3
