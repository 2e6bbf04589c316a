"""Wide and deep GaussianMLPPolicy nets on the HIP kernels (csrc/policy_wide_kernels.hip, rollout_wide_kernel).

``GaussianMLPPolicy(hidden_sizes=...)`` is free-form in the reference (rllab/policies/gaussian_mlp_policy.py:21-58,
rllab/core/network.py:36-101; rllab's own MuJoCo experiments use (100, 50, 25)).  Two or three tanh layers of up to
128 units stay on the fused path: zero-padded to 32 / 64 / 128 per layer (policies/kernel_layout.py), sampled by
the fused rollout and updated by the cooperative-workgroup passes.  The parity matrix of
tests/test_gpu_update_parity.py (float64 autograd of the reference formulas, 2e-5; Fisher-vector product 5e-5)
extended to these shapes, the rollout replayed bit for bit on the host build, and a TRPO iteration end to end."""
import numpy as np
import pytest
import torch

from tests import test_gpu_update_parity as U

pytestmark = pytest.mark.gpu

# (obs_dim, act_dim, hidden_sizes): padded to (128,64,32), (128,128), (128,64,32), (128,128), (64,64,64), (128,64),
# (32,128), (64,128,64)
WIDE_SHAPES = [(13, 2, (100, 50, 25)), (13, 2, (128, 128)), (20, 6, (100, 50, 25)), (20, 6, (128, 128)),
               (4, 1, (64, 64, 64)), (21, 6, (128, 64)), (11, 1, (32, 128)), (6, 1, (40, 100, 40))]


def _policy(do, da, hidden, seed=0):
    from rllab_amd.envs.env_spec import EnvSpec
    from rllab_amd.policies.gaussian_mlp_policy import GaussianMLPPolicy
    from rllab_amd.spaces import Box
    np.random.seed(seed)
    spec = EnvSpec(Box(-np.ones(do), np.ones(do)), Box(-np.ones(da), np.ones(da)))
    pol = GaussianMLPPolicy(spec, hidden_sizes=hidden)
    theta = pol.get_param_values()
    theta += 0.1 * np.random.randn(theta.size)
    pol.set_param_values(theta)
    return pol


def test_the_reference_experiment_sizes_stay_fused():
    from rllab_amd.policies.kernel_layout import padded_sizes
    assert padded_sizes((100, 50, 25)) == (128, 64, 32) and padded_sizes((128, 128)) == (128, 128)
    assert padded_sizes((48, 20)) == (64, 64) and padded_sizes((32, 32)) == (32, 32)        # the equal-width family
    assert padded_sizes((129, 32)) is None and padded_sizes((8, 8, 8, 8)) is None
    assert padded_sizes((32,)) == (32, 32) and padded_sizes((40,)) == (64, 64)      # one layer: + identity
    assert padded_sizes((65,)) == (128, 128) and padded_sizes((129,)) is None       # (65 .. 128 units: the cooperative family, round 6)
    for hidden in ((100, 50, 25), (128, 128)):
        pol = _policy(13, 2, hidden)
        assert pol.kernel_layout() is not None and pol.kernel_layout().wide
        assert pol.fused_ops() is not None
    lay = _policy(13, 2, (100, 50, 25)).kernel_layout()
    assert lay.hidden3 == (128, 64, 32) and not lay.exact
    v = torch.arange(lay.P, dtype=torch.float64, device="cuda")
    assert torch.equal(lay.unpack(lay.pack(v)), v) and float(lay.pack(v).abs().sum()) == float(v.sum())
    assert _policy(13, 2, (128, 128)).kernel_layout().exact


@pytest.mark.parametrize("do,da,hidden", WIDE_SHAPES)
@pytest.mark.parametrize("B", [1, 63, 1000, 70001])
def test_loss_kl_grad_vs_float64_autograd(do, da, hidden, B):
    pol = _policy(do, da, hidden)
    ops = pol.fused_ops()
    assert ops is not None and ops.layout.wide
    inp = U._inputs(pol, B)
    surr, kl, vpg = U._closures(pol)
    flat64 = pol.flat_params.detach().double().requires_grad_(True)
    l64, k64, v64 = surr(flat64, *inp), kl(flat64, *inp), vpg(flat64, *inp)
    s = ops.loss_stats(inp)
    assert abs(float(-s[0]) - float(l64.detach())) <= 2e-5 * max(1.0, abs(float(l64.detach())))
    assert abs(float(s[1]) - float(k64.detach())) <= 2e-5 * max(1e-2, abs(float(k64.detach())))
    assert abs(float(-s[2]) - float(v64.detach())) <= 2e-5 * max(1.0, abs(float(v64.detach())))
    g64 = torch.autograd.grad(l64, flat64, retain_graph=True)[0]
    g = ops.loss_grad(inp)
    assert float((g - g64).abs().max()) <= 2e-5 * max(1e-3, float(g64.abs().max()))
    # the same pass hands back the loss sums (rl_policy_grad_loss)
    ops.release()
    g2 = ops.loss_grad(inp, with_loss=True)
    assert torch.equal(g2, g)
    s2 = ops.loss_stats(inp)
    assert torch.allclose(s2, s, rtol=1e-12, atol=1e-12)
    gv64 = torch.autograd.grad(v64, flat64)[0]
    gv = ops.loss_grad(inp, vpg=True)
    assert float((gv - gv64).abs().max()) <= 2e-5 * max(1e-3, float(gv64.abs().max()))


@pytest.mark.parametrize("do,da,hidden", WIDE_SHAPES)
def test_fvp_equals_kl_hessian_at_theta_old(do, da, hidden):
    pol = _policy(do, da, hidden)
    ops = pol.fused_ops()
    B = 5000
    inp = U._inputs(pol, B, old_equals_new=True)
    _, kl, _ = U._closures(pol)
    rng = np.random.RandomState(3)
    flat64 = pol.flat_params.detach().double().requires_grad_(True)
    with torch.no_grad():
        om64 = pol.mean_planes(inp[0].double(), flat64.detach())
    inp64 = (inp[0], inp[1], inp[2], om64, pol.effective_log_std().detach().double().reshape(-1, 1), inp[5], inp[6])
    g = torch.autograd.grad(kl(flat64, *inp64), flat64, create_graph=True)[0]
    for trial in range(2):
        v = torch.as_tensor(rng.randn(flat64.numel()), device=flat64.device)
        hv64 = torch.autograd.grad((g * v).sum(), flat64, retain_graph=True)[0]
        hv = ops.fvp(inp, v)
        assert float((hv - hv64).abs().max()) <= 5e-5 * float(hv64.abs().max())
    # symmetric: u . F v == v . F u
    u = torch.as_tensor(rng.randn(flat64.numel()), device=flat64.device)
    v = torch.as_tensor(rng.randn(flat64.numel()), device=flat64.device)
    a, b = float(u.dot(ops.fvp(inp, v))), float(v.dot(ops.fvp(inp, u)))
    assert abs(a - b) <= 1e-4 * max(abs(a), abs(b))


@pytest.mark.parametrize("do,da,hidden", WIDE_SHAPES + [(7, 3, (32, 32)), (17, 8, (64, 64))])
@pytest.mark.parametrize("B", [1, 63, 1000, 70001])
def test_fvp_on_cached_activations_is_the_same_product(do, da, hidden, B):
    """The cooperative kernels' form of tests/test_gpu_update_parity.py's cache test: rl_policy_grad with
    batch.activations set leaves every layer's activations in device memory (rl_policy_activation_bytes: one float per
    sample and PADDED hidden unit), rl_policy_fvp then runs no forward chain -- the same gradient and the same product
    bit for bit, through CG as well; a parameter update drops the cache."""
    pol = _policy(do, da, hidden)
    ops = pol.fused_ops()
    assert ops.wide_kernels
    inp = U._inputs(pol, B, old_equals_new=True)
    v = torch.as_tensor(np.random.RandomState(5).randn(pol.flat_params.numel()), device="cuda")
    want_g, want_hv = ops.loss_grad(inp), ops.fvp(inp, v)
    assert ops._acts_tag is None
    g = ops.loss_grad(inp, keep_activations=True)
    padded = ops.dims[2] + ops.dims[3] + ops.dims[4]
    assert ops._acts_tag is not None and ops._acts.numel() == ((B + 31) // 32) * 32 * padded * 4
    hv = ops.fvp(inp, v)
    assert torch.equal(g, want_g)
    assert torch.equal(hv, want_hv)
    x, xhx = ops.cg(inp, want_g, 3, 1e-5)
    ops._acts_tag = None
    x2, xhx2 = ops.cg(inp, want_g, 3, 1e-5)
    assert torch.equal(x, x2) and torch.equal(xhx, xhx2)
    ops.loss_grad(inp, keep_activations=True)
    with torch.no_grad():
        pol.flat_params.add_(0.01)
    hv_new = ops.fvp(inp, v)
    ops._acts_tag = None
    assert torch.equal(hv_new, ops.fvp(inp, v)) and not torch.equal(hv_new, want_hv)
    ops.loss_grad(inp, vpg=True, keep_activations=True)
    assert ops._acts_tag is None


@pytest.mark.parametrize("do,da,hidden", [(13, 2, (100, 50, 25)), (20, 6, (128, 128))])
def test_penalised_surrogate_value_and_gradient(do, da, hidden):
    """PPO's objective (surrogate + penalty * mean KL, penalty_lbfgs_optimizer.py:66-79) from one pass."""
    pol = _policy(do, da, hidden)
    ops = pol.fused_ops()
    inp = U._inputs(pol, 5003)
    surr, kl, _ = U._closures(pol)
    flat64 = pol.flat_params.detach().double().requires_grad_(True)
    obj = surr(flat64, *inp) + 2.5 * kl(flat64, *inp)
    g64 = torch.autograd.grad(obj, flat64)[0].cpu().numpy()
    val, g = ops.value_and_grad(inp, 2.5)
    assert abs(val - float(obj.detach())) <= 2e-5 * max(1.0, abs(float(obj.detach())))
    assert np.abs(g - g64).max() <= 3e-5 * max(1e-3, np.abs(g64).max())


def test_device_cg_beyond_the_register_cached_size():
    """(20 -> 128 -> 128 -> 6) has 20 108 parameters: rl_cg_init / rl_cg_step / rl_trpo_step take the re-reading form
    (n > 16 384).  The device CG must be krylov.cg (rllab/misc/krylov.py:7-39) on the same Fisher-vector products."""
    from rllab_amd.misc import krylov
    pol = _policy(20, 6, (128, 128))
    assert pol.flat_params.numel() > 16384
    ops = pol.fused_ops()
    inp = U._inputs(pol, 20000, old_equals_new=True)
    g = ops.loss_grad(inp)
    x, xhx = ops.cg(inp, g, 6, 1e-5)
    want = krylov.cg(lambda p: ops.fvp(inp, p) + 1e-5 * p, g, cg_iters=6)
    assert float((x - want).abs().max()) <= 1e-6 * float(want.abs().max())
    step, stats = ops.cg_step_vector(inp, g, 6, 1e-5, 0.01)
    beta = float(stats[1])
    assert abs(float(stats[0]) - float(xhx)) <= 1e-5 * abs(float(xhx))
    assert float((step - beta * x).abs().max()) <= 1e-9 * float(step.abs().max())
    assert abs(beta - np.sqrt(2 * 0.01 / (float(stats[0]) + 1e-8))) <= 1e-12 * beta


@pytest.mark.parametrize("kind,hidden", [(2, (100, 50, 25)), (0, (128, 128)), (3, (100, 50, 25)), (6, (64, 32, 32))])
@pytest.mark.parametrize("epw", ["auto", "16", "64"])
def test_fused_rollout_of_a_wide_policy(kind, hidden, epw, monkeypatch):
    """The wide policies' rollouts in every launch shape -- "auto": what a run takes (the lane-group kernels of the
    Swimmer and the two-legged envs with the policy's weight fragments in LDS, 16 envs per wavefront for the rest),
    "16" / "64": the generic rollout_wide_kernel shapes: env dynamics replayed on the host build from the recorded
    actions, bit for bit; recorded means against a float64 torch forward of the same parameters."""
    from rllab_amd.envs.hip_env import HipVecEnv
    from oracle.replay import replay_check
    from tests.test_gpu_env_parity import _make_policy
    monkeypatch.delenv("RLLAB_ROLLOUT_EPW", raising=False)
    if epw != "auto":
        monkeypatch.setenv("RLLAB_ROLLOUT_EPW", epw)
    rng = np.random.RandomState(1)
    n, T, mpl = 130, 40, 17
    policy = _make_policy(kind, hidden)
    assert policy.kernel_layout() is not None and policy.kernel_layout().wide
    v = HipVecEnv(kind, n, mpl, normalize=True, seed=11)
    q = v.q
    eps = rng.randn(q["act_dim"], T, n).astype(np.float32)
    draws = (rng.randn if q["reset_is_normal"] else rng.rand)(T + 1, q["reset_draws"], n).astype(np.float32)
    traj = v.rollout(policy, T, reset_at_start=True, eps=eps, reset_draws=draws)
    torch.cuda.synchronize()
    assert replay_check(v, traj, max_envs=n, reset_draws=draws) == n * T
    obs64 = traj.obs.reshape(q["obs_dim"], -1).double()
    with torch.no_grad():
        mean64 = policy.mean_planes(obs64, policy.flat_params.double())
    got = traj.means.reshape(q["act_dim"], -1).double()
    assert float((got - mean64).abs().max()) <= 2e-5
    std = torch.exp(policy.effective_log_std().double())[:, None]
    act64 = got + torch.as_tensor(eps, device=got.device).reshape(q["act_dim"], -1).double() * std
    assert float((traj.actions.reshape(q["act_dim"], -1).double() - act64).abs().max()) <= 1e-6


@pytest.mark.parametrize("hidden", [(128, 128), (100, 50, 25), (64, 32), (32, 64, 128), (128, 128, 128)])
@pytest.mark.parametrize("coop", ["1", "0"])
def test_swimmer_rollout_with_the_network_split_over_four_wavefronts(hidden, coop, monkeypatch):
    """rollout_swimmer_quad_coop_kernel (four wavefronts per group of 16 envs, the layers split by output units on
    16 x 16 x 4 matrix tiles, activations through LDS) against the one-wavefront-per-group kernel it replaces for small
    launches (RLLAB_SWIMMER_COOP=0): the same trajectories up to the rounding of the means (so: every recorded
    transition replays on the host bit for bit from ITS recorded action, means within 2e-5 of float64), every layer
    width in every position, and the (128, 128, 128) net whose weights do not fit the cooperative layout's LDS and
    falls back."""
    from rllab_amd.envs.hip_env import HipVecEnv
    from oracle.replay import replay_check
    from tests.test_gpu_env_parity import _make_policy
    monkeypatch.delenv("RLLAB_ROLLOUT_EPW", raising=False)
    monkeypatch.setenv("RLLAB_SWIMMER_COOP", coop)
    rng = np.random.RandomState(3)
    n, T, mpl = 70, 60, 23                       # 4 full groups + one of 6 envs; resets inside the launch
    policy = _make_policy(2, hidden)
    v = HipVecEnv(2, n, mpl, normalize=True, seed=5)
    q = v.q
    eps = rng.randn(q["act_dim"], T, n).astype(np.float32)
    draws = (rng.randn if q["reset_is_normal"] else rng.rand)(T + 1, q["reset_draws"], n).astype(np.float32)
    traj = v.rollout(policy, T, reset_at_start=True, eps=eps, reset_draws=draws)
    torch.cuda.synchronize()
    assert replay_check(v, traj, max_envs=n, reset_draws=draws) == n * T
    obs64 = traj.obs.reshape(q["obs_dim"], -1).double()
    with torch.no_grad():
        mean64 = policy.mean_planes(obs64, policy.flat_params.double())
    got = traj.means.reshape(q["act_dim"], -1).double()
    assert float((got - mean64).abs().max()) <= 2e-5
    if coop == "1":
        # Philox noise and resets inside the kernel: the same streams in both shapes -- the first transition (before
        # the means' rounding can steer the trajectories apart) agrees: same reset state, same policy noise
        a = HipVecEnv(2, n, mpl, normalize=True, seed=5).rollout(policy, T, reset_at_start=True)
        monkeypatch.setenv("RLLAB_SWIMMER_COOP", "0")
        v2 = HipVecEnv(2, n, mpl, normalize=True, seed=5)
        b = v2.rollout(policy, T, reset_at_start=True)
        torch.cuda.synchronize()
        assert torch.equal(a.obs[:, 0], b.obs[:, 0])
        za, zb = (a.actions - a.means)[:, 0], (b.actions - b.means)[:, 0]
        assert float((za - zb).abs().max()) <= 1e-6 and float(za.abs().max()) > 0.1
        assert float((a.means[:, 0] - b.means[:, 0]).abs().max()) <= 4e-5
        assert bool(torch.isfinite(a.actions).all()) and bool(torch.isfinite(a.rewards).all())


@pytest.mark.parametrize("hidden", [(100, 50, 25), (128, 128)])
def test_trpo_on_the_kernels_with_the_reference_experiment_net(hidden, quiet_logger):
    """TRPO on the Swimmer with the (100, 50, 25) / (128, 128) policy: fused rollout, fused update, device CG."""
    from rllab.algos.trpo import TRPO
    from rllab.baselines.linear_feature_baseline import LinearFeatureBaseline
    from rllab.envs.mujoco.swimmer_env import SwimmerEnv
    from rllab.envs.normalized_env import normalize
    from rllab.misc import ext, logger
    from rllab.policies.gaussian_mlp_policy import GaussianMLPPolicy
    ext.set_seed(2)
    env = normalize(SwimmerEnv())
    policy = GaussianMLPPolicy(env_spec=env.spec, hidden_sizes=hidden)
    algo = TRPO(env=env, policy=policy, baseline=LinearFeatureBaseline(env_spec=env.spec), batch_size=256 * 100,
                max_path_length=100, n_itr=4, discount=0.99, step_size=0.01, sampler_args=dict(n_envs=256))
    algo.start_worker()
    algo.init_opt()
    assert algo.optimizer._fused is not None and algo.sampler._takes_fused_rollout(policy)
    theta0 = policy.get_param_values().copy()
    for itr in range(4):
        paths = algo.sampler.obtain_samples(itr)
        sd = algo.sampler.process_samples(itr, paths)
        algo.log_diagnostics(paths)
        algo.optimize_policy(itr, sd)
        tab = logger.get_tabular()
        assert float(tab["MeanKL"]) <= 0.0101 and float(tab["LossAfter"]) < float(tab["LossBefore"])
        assert abs(float(tab["MeanKLBefore"])) < 1e-6
        logger.dump_tabular()
    assert np.isfinite(policy.get_param_values()).all() and np.abs(policy.get_param_values() - theta0).max() > 0


@pytest.mark.parametrize("do,da,h", [(7, 3, 32), (17, 8, 64), (30, 1, 32), (3, 2, 48)])
def test_any_observation_and_action_width_stays_on_the_kernels(do, da, h):
    """The one-wavefront-per-tile kernels are instantiated for the HIP envs' (obs_dim, action_dim) pairs; an equal-width
    net on ANY other pair (obs_dim <= 30, action_dim <= 8: a user's own env sampled through BatchSampler, say) is
    dispatched to the cooperative kernels, which take the widths at run time.  Same parity bar."""
    pol = _policy(do, da, (h, h))
    ops = pol.fused_ops()
    assert ops is not None and ops.wide_kernels and not ops.layout.wide
    inp = U._inputs(pol, 9001)
    surr, kl, vpg = U._closures(pol)
    flat64 = pol.flat_params.detach().double().requires_grad_(True)
    l64, k64 = surr(flat64, *inp), kl(flat64, *inp)
    s = ops.loss_stats(inp)
    assert abs(float(-s[0]) - float(l64.detach())) <= 2e-5 * max(1.0, abs(float(l64.detach())))
    assert abs(float(s[1]) - float(k64.detach())) <= 2e-5 * max(1e-2, abs(float(k64.detach())))
    g64 = torch.autograd.grad(l64, flat64)[0]
    g = ops.loss_grad(inp, keep_activations=True)
    assert ops._acts_tag is not None                        # the cooperative kernels keep their activations too
    assert float((g - g64).abs().max()) <= 2e-5 * max(1e-3, float(g64.abs().max()))
    inp0 = U._inputs(pol, 5000, old_equals_new=True)
    with torch.no_grad():
        om64 = pol.mean_planes(inp0[0].double(), flat64.detach())
    inp64 = (inp0[0], inp0[1], inp0[2], om64, pol.effective_log_std().detach().double().reshape(-1, 1), inp0[5], inp0[6])
    gk = torch.autograd.grad(kl(flat64, *inp64), flat64, create_graph=True)[0]
    v = torch.as_tensor(np.random.RandomState(3).randn(flat64.numel()), device=flat64.device)
    hv64 = torch.autograd.grad((gk * v).sum(), flat64)[0]
    hv = ops.fvp(inp0, v)
    assert float((hv - hv64).abs().max()) <= 5e-5 * float(hv64.abs().max())
    x, _ = ops.cg(inp0, ops.loss_grad(inp0), 3, 1e-5)
    assert bool(torch.isfinite(x).all())
