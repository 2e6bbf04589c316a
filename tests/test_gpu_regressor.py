"""GaussianMLPRegressor / GaussianMLPBaseline on the HIP kernels (SURVEY.md 8 f1,
rllab/regressors/gaussian_mlp_regressor.py:107-143, rllab/baselines/gaussian_mlp_baseline.py:10-47): the fit objective
-- negative log-likelihood of whitened targets plus penalty x mean KL to the previous prediction -- evaluated by
rl_policy_grad_loss(vpg = 1, kl_penalty) on rectify / tanh nets, against float64 autograd of the regressor's own
closures (the restated reference formulas), and a whole fit against the autograd-driven fit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _regressor(din, nonlin, seed=0, **kw):
    from rllab_amd.regressors.gaussian_mlp_regressor import GaussianMLPRegressor
    np.random.seed(seed)
    return GaussianMLPRegressor(input_shape=(din,), output_dim=1, hidden_nonlinearity=nonlin, name="vf", **kw)


def _problem(reg, B, seed=1, ragged=True):
    rng = np.random.RandomState(seed)
    dev = reg.flat_params.device
    xs = torch.as_tensor((rng.randn(reg.input_dim, B) * 2.0 + 0.5).astype(np.float32), device=dev)
    ys = torch.as_tensor((np.sin(xs[:2].sum(0).cpu().numpy()) * 3.0 + rng.randn(B) * 0.3 + 5.0).astype(np.float32)[None, :],
                         device=dev)
    w = torch.ones(B, device=dev)
    if ragged:
        w[torch.as_tensor(rng.rand(B) < 0.15, device=dev)] = 0.0
    # whitening statistics and non-trivial parameters
    reg._x_mean, reg._x_std = xs.mean(1, keepdim=True), xs.std(1, keepdim=True) + 1e-8
    reg._y_mean, reg._y_std = ys.mean(1, keepdim=True), ys.std(1, keepdim=True) + 1e-8
    th = reg.get_param_values()
    reg.set_param_values(th + 0.05 * rng.randn(th.size))
    old_means, old_log_stds = reg.pdists_planes(xs)
    old_means = old_means + torch.as_tensor(0.2 * rng.randn(1, B).astype(np.float32), device=dev)
    old_log_stds = old_log_stds + 0.1
    inv = (1.0 / w.double().sum()).to(torch.float32)
    return (xs, ys, old_means, old_log_stds, w, inv)


@pytest.mark.parametrize("din", [4, 13, 20])
@pytest.mark.parametrize("nonlin_name", ["rectify", "tanh"])
@pytest.mark.parametrize("penalty", [0.0, 0.7])
def test_fit_objective_and_gradient_vs_float64_autograd(din, nonlin_name, penalty):
    from rllab_amd.core import network
    reg = _regressor(din, getattr(network, nonlin_name), use_trust_region=True)
    assert reg._fused is not None
    inputs = _problem(reg, 4099)
    opt = reg._optimizer
    flat64 = reg.flat_params.detach().double().requires_grad_(True)
    in64 = tuple(t.double() if torch.is_tensor(t) else t for t in inputs)
    # the closures whiten with float32 statistics; evaluate them in float64
    reg._x_mean, reg._x_std, reg._y_mean, reg._y_std = (t.double() for t in (reg._x_mean, reg._x_std, reg._y_mean,
                                                                             reg._y_std))
    l64 = opt._loss(flat64, *in64)
    k64 = opt._constraint(flat64, *in64)
    g64 = torch.autograd.grad(l64 + penalty * k64, flat64)[0]
    reg._x_mean, reg._x_std, reg._y_mean, reg._y_std = (t.float() for t in (reg._x_mean, reg._x_std, reg._y_mean,
                                                                            reg._y_std))
    nll, kl = reg._fused.loss_and_kl(inputs)
    assert abs(nll - float(l64)) <= 2e-5 * max(1.0, abs(float(l64)))
    assert abs(kl - float(k64)) <= 2e-5 * max(1e-2, abs(float(k64)))
    val, g = reg._fused.value_and_grad(inputs, penalty)
    assert abs(val - float(l64 + penalty * k64)) <= 2e-5 * max(1.0, abs(float(l64 + penalty * k64)))
    assert np.abs(g - g64.cpu().numpy()).max() <= 3e-5 * max(1e-3, float(g64.abs().max()))
    assert opt.loss(inputs) == pytest.approx(nll) and opt.constraint_val(inputs) == pytest.approx(kl)


@pytest.mark.parametrize("use_trust_region", [True, False])
def test_whole_fit_on_the_kernels_matches_the_autograd_fit(use_trust_region, quiet_logger):
    """Same data, same start: L-BFGS driven by the HIP objective lands where L-BFGS driven by torch autograd lands
    (float32 objective in both cases -- trajectories agree to optimisation tolerance, not bit for bit)."""
    from rllab_amd.core.network import rectify
    rng = np.random.RandomState(4)
    B = 20000
    xs = torch.as_tensor(rng.randn(13, B).astype(np.float32), device="cuda")
    ys = torch.as_tensor((2.0 * np.tanh(xs[0].cpu().numpy()) - xs[1].cpu().numpy() ** 2 + 0.1 * rng.randn(B))
                         .astype(np.float32)[None, :], device="cuda")
    fits = {}
    for fused in (True, False):
        reg = _regressor(13, rectify, seed=7, use_trust_region=use_trust_region, step_size=0.05)
        assert reg._fused is not None
        if not fused:
            reg._fused = None
            reg._optimizer._fused = None
        for _ in range(3):                                   # successive fits move through the trust region
            reg.fit_planes(xs, ys)
        pred = reg.predict_planes(xs)
        fits[fused] = (float(((pred - ys) ** 2).mean()), pred)
    mse_hip, mse_ref = fits[True][0], fits[False][0]
    var = float(ys.var())
    assert mse_hip < 0.5 * var and mse_ref < 0.5 * var                  # both learned the function
    assert abs(mse_hip - mse_ref) <= 0.15 * max(mse_ref, 0.02 * var), (mse_hip, mse_ref)


def test_gaussian_mlp_baseline_learns_the_value_function(quiet_logger):
    """TRPO + GaussianMLPBaseline (the neural value function of the rllab benchmark paper) on the HIP Cartpole: the
    baseline fit runs on the kernels, logs the reference's vf_ keys, and its explained variance becomes positive."""
    from rllab.algos.trpo import TRPO
    from rllab.baselines.gaussian_mlp_baseline import GaussianMLPBaseline
    from rllab.envs.box2d.cartpole_env import CartpoleEnv
    from rllab.envs.normalized_env import normalize
    from rllab.misc import ext, logger
    from rllab.policies.gaussian_mlp_policy import GaussianMLPPolicy
    ext.set_seed(1)
    env = normalize(CartpoleEnv())
    policy = GaussianMLPPolicy(env_spec=env.spec, hidden_sizes=(32, 32))
    baseline = GaussianMLPBaseline(env_spec=env.spec)
    assert baseline._regressor._fused is not None
    algo = TRPO(env=env, policy=policy, baseline=baseline, batch_size=256 * 100, max_path_length=100, n_itr=8,
                discount=0.99, step_size=0.01, sampler_args=dict(n_envs=256))
    algo.start_worker()
    algo.init_opt()
    evs, rets = [], []
    for itr in range(8):
        paths = algo.sampler.obtain_samples(itr)
        sd = algo.sampler.process_samples(itr, paths)
        algo.log_diagnostics(paths)
        algo.optimize_policy(itr, sd)
        tab = logger.get_tabular()
        for key in ("vf_LossBefore", "vf_LossAfter", "vf_dLoss", "vf_MeanKL"):
            assert key in tab, (key, sorted(tab))
        assert float(tab["vf_LossAfter"]) <= float(tab["vf_LossBefore"]) + 1e-6
        assert float(tab["vf_MeanKL"]) <= 0.0101
        evs.append(float(tab["ExplainedVariance"]))
        rets.append(float(tab["AverageReturn"]))
        logger.dump_tabular()
    assert max(evs[2:]) > 0.2, evs
    assert np.mean(rets[-2:]) > np.mean(rets[:2]), rets
