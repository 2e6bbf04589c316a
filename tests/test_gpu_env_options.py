"""Constructor options of the reference's env classes as run-time options of the kernels (``rl_env_cfg``):
ctrl_cost_coeff / alive_coeff (swimmer_env.py:15-21, walker2d_env.py:21-27, hopper_env.py:27-35), action_noise
(box2d_env.py:219-226, mujoco_env.py:175-187), obs_noise and frame_skip (box2d_env.py:30-58,194-218),
position_only (:185-192), random_start (inverted_double_pendulum_env.py:20) and the engine's
reset_pole_follows_cart.  Every option: GPU kernel == host build of the same dynamics, bit for bit, with the noise
draws injected on both sides; plus what the option must DO (reference semantics restated in numpy)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# (env kind, options): every option at a non-default value on every env that has it
CASES = [
    (0, dict(action_noise=0.3)), (0, dict(obs_noise=0.05)), (0, dict(frame_skip=3)), (0, dict(flags=1)),
    (0, dict(action_noise=0.2, obs_noise=0.1, frame_skip=2, flags=1)),
    (1, dict(action_noise=0.1, obs_noise=0.02)), (1, dict(frame_skip=1)), (1, dict(frame_skip=4)),
    (1, dict(link_len=1.37)), (1, dict(link_len=0.6, action_noise=0.1, frame_skip=3)),
    (4, dict(action_noise=0.5, obs_noise=0.3, flags=1)),
    (2, dict(ctrl_cost_coeff=0.7)), (2, dict(action_noise=0.25)), (2, dict(ctrl_cost_coeff=0.0, action_noise=0.1)),
    (2, dict(flags=4)), (2, dict(flags=4, action_noise=0.2)),     # SwimmerEnv(limit_model="mujoco"): RL_CFG_LIMIT_MUJOCO
    (3, dict(action_noise=0.4)),
    (5, dict(ctrl_cost_coeff=0.3, action_noise=0.05)),
    (6, dict(ctrl_cost_coeff=0.2, alive_coeff=2.5, action_noise=0.05)),
    # limit_model / contact_model = "mujoco" (RL_CFG_LIMIT_MUJOCO = 4, RL_CFG_CONTACT_MUJOCO = 8): csrc/dyn_mjc.h
    (3, dict(flags=4)), (3, dict(flags=8)), (3, dict(flags=12)), (3, dict(flags=12, action_noise=0.3)),
    (5, dict(flags=12)), (5, dict(flags=8, ctrl_cost_coeff=0.3)), (6, dict(flags=12)), (6, dict(flags=4, alive_coeff=0.5)),
    (7, dict(flags=2)), (7, dict(action_noise=0.2)),
]


def _draws(rng, q, n):
    return (rng.randn(q["reset_draws"], n) if q["reset_is_normal"] else rng.rand(q["reset_draws"], n)).astype(np.float32)


@pytest.mark.parametrize("kind,cfg", CASES)
def test_vecenv_step_with_options_bit_exact(kind, cfg):
    from rllab_amd.envs.hip_env import HipVecEnv
    from oracle import host_env as H
    rng = np.random.RandomState(5)
    n, mpl = 193, 20
    gpu = HipVecEnv(kind, n, mpl, normalize=True, scale_reward=1.0, seed=3, cfg=cfg)
    cpu = H.HostVecEnv(kind, n, mpl, normalize=True, scale_reward=1.0, cfg=cfg)
    q = gpu.q
    d0, zo0 = _draws(rng, q, n), rng.randn(q["obs_dim"], n).astype(np.float32)
    og = gpu.reset(draws=d0, obs_noise_z=zo0)
    oc = cpu.reset(d0, obs_z=zo0)
    assert np.array_equal(og.t().cpu().numpy().view(np.uint32), oc.view(np.uint32))
    # and the options change something (against the default-option executor on the same inputs)
    ref = HipVecEnv(kind, n, mpl, normalize=True, scale_reward=1.0, seed=3)
    ref.reset(draws=d0)
    differs = False
    n_done = 0
    for t in range(45):
        a = rng.randn(n, q["act_dim"]).astype(np.float32)
        dr, za, zo = _draws(rng, q, n), rng.randn(q["act_dim"], n).astype(np.float32), \
            rng.randn(q["obs_dim"], n).astype(np.float32)
        og, rg, dg, _ = gpu.step(torch.as_tensor(a, device=gpu.device), reset_draws=dr, action_noise_z=za, obs_noise_z=zo)
        oc, rc, dc = cpu.step(a.T, dr, act_z=za, obs_z=zo)
        assert np.array_equal(gpu.state.cpu().numpy().view(np.uint32), cpu.state.view(np.uint32)), t
        assert np.array_equal(og.t().cpu().numpy().view(np.uint32), oc.view(np.uint32)), t
        assert np.array_equal(rg.cpu().numpy().view(np.uint32), rc.view(np.uint32)), t
        assert np.array_equal(dg.cpu().numpy(), dc.astype(bool)), t
        orf, rrf, _, _ = ref.step(torch.as_tensor(a, device=gpu.device), reset_draws=dr)
        differs = differs or not (torch.equal(orf, og) and torch.equal(rrf, rg))
        n_done += int(dc.sum())
    assert differs and n_done > 0


@pytest.mark.parametrize("kind,cfg,hidden", [
    (0, dict(action_noise=0.2, obs_noise=0.05, frame_skip=2, flags=1), (32, 32)),
    (1, dict(action_noise=0.1, obs_noise=0.02), (32, 32)),
    (1, dict(link_len=0.75, action_noise=0.05), (32, 32)),
    (2, dict(ctrl_cost_coeff=0.4, action_noise=0.15), (32, 32)),      # the lane-group (quad) kernel
    (2, dict(ctrl_cost_coeff=0.4, action_noise=0.15), (64, 64)),
    (2, dict(flags=4), (32, 32)),                                     # soft-constraint joint limits: the scalar program
    (3, dict(action_noise=0.3), (64, 64)),
    (6, dict(ctrl_cost_coeff=0.2, alive_coeff=0.5, action_noise=0.05), (32, 32)),
    # MuJoCo's soft-constraint limits and contacts (csrc/dyn_mjc.h): the env-per-lane kernels
    (3, dict(flags=12), (64, 64)), (3, dict(flags=8, action_noise=0.2), (32, 32)), (5, dict(flags=12), (32, 32)),
    (6, dict(flags=12), (32, 32)),
])
def test_fused_rollout_with_options_replays_on_the_host(kind, cfg, hidden):
    from rllab_amd.envs.hip_env import HipVecEnv
    from oracle.replay import replay_check
    from tests.test_gpu_env_parity import _make_policy
    rng = np.random.RandomState(2)
    n, T, mpl = 97, 30, 13
    policy = _make_policy(kind, hidden)
    v = HipVecEnv(kind, n, mpl, normalize=True, seed=11, cfg=cfg)
    q = v.q
    eps = rng.randn(q["act_dim"], T, n).astype(np.float32)
    draws = (rng.randn if q["reset_is_normal"] else rng.rand)(T + 1, q["reset_draws"], n).astype(np.float32)
    if kind == 2 and cfg.get("flags", 0) & 4:
        draws *= 150.0           # limit model: start far outside the reset distribution (hinge angles ~ 1.5 N(0,1) rad
                                 # against limits of 1.745), so that the limit rows are active from the first sub-step
    za = rng.randn(T, q["act_dim"], n).astype(np.float32)
    zo = rng.randn(T + 1, q["obs_dim"], n).astype(np.float32)
    traj = v.rollout(policy, T, reset_at_start=True, eps=eps, reset_draws=draws, action_noise_z=za, obs_noise_z=zo)
    torch.cuda.synchronize()
    assert replay_check(v, traj, max_envs=n, reset_draws=draws, action_noise_z=za, obs_noise_z=zo) == n * T
    # the same rollout without options differs (the options reached the fused kernel)
    v0 = HipVecEnv(kind, n, mpl, normalize=True, seed=11)
    t0 = v0.rollout(policy, T, reset_at_start=True, eps=eps, reset_draws=draws)
    assert not (torch.equal(t0.rewards, traj.rewards) and torch.equal(t0.obs, traj.obs))


def test_option_semantics_against_the_reference_formulas():
    """What each option does, restated from the reference with numpy on the kernel's own outputs."""
    from rllab_amd.envs.hip_env import HipVecEnv
    rng = np.random.RandomState(9)
    n = 64
    a = rng.uniform(-1, 1, size=(n, 2)).astype(np.float32)
    d0 = rng.randn(10, n).astype(np.float32)

    def one_step(cfg, **kw):
        v = HipVecEnv(2, n, 0, normalize=True, seed=1, cfg=cfg)
        v.reset(draws=d0)
        o, r, d, _ = v.step(torch.as_tensor(a, device=v.device), **kw)
        return o.clone(), r.clone().double().cpu().numpy(), v
    # ctrl_cost_coeff: reward = comvel_x - 0.5 c sum((action / scaling)^2); normalised actions in [-1, 1] map to
    # lb + (a + 1) (ub - lb) / 2 = 50 a, scaling = 50  (swimmer_env.py:37-43)
    o1, r1, v1 = one_step(dict(ctrl_cost_coeff=0.0))
    o2, r2, _ = one_step(dict(ctrl_cost_coeff=0.8))
    assert torch.equal(o1, o2)
    np.testing.assert_allclose(r1 - r2, 0.5 * 0.8 * (a.astype(np.float64) ** 2).sum(1), rtol=0, atol=2e-6)
    # get_body_comvel: the reward's forward term (swimmer_env.py:41), get_body_com: the observation's tail
    com = v1.com().double().cpu().numpy()
    np.testing.assert_allclose(com[:, 2], r1, rtol=0, atol=1e-6)
    np.testing.assert_allclose(com[:, :2], o1[:, 10:12].double().cpu().numpy(), rtol=0, atol=1e-6)
    # action_noise: ctrl = action + 0.5 (ub - lb) sigma z, the reward's cost term still sees the clean action
    z = rng.randn(2, n).astype(np.float32)
    o3, r3, _ = one_step(dict(ctrl_cost_coeff=0.8, action_noise=0.01), action_noise_z=z)
    v_shift = HipVecEnv(2, n, 0, normalize=True, seed=1, cfg=dict(ctrl_cost_coeff=0.8))
    v_shift.reset(draws=d0)
    a_shift = a + 0.01 * z.T                        # 50 (a + sigma z) = 50 a + 0.5 * 100 * sigma z
    o4, r4, _, _ = v_shift.step(torch.as_tensor(a_shift, device=v_shift.device))
    np.testing.assert_allclose(o3.cpu().numpy(), o4.cpu().numpy(), rtol=0, atol=5e-5)
    cost = lambda act: 0.5 * 0.8 * (act.astype(np.float64) ** 2).sum(1)
    np.testing.assert_allclose(r3 + cost(a), r4.double().cpu().numpy() + cost(a_shift), rtol=0, atol=5e-5)


def test_in_kernel_noise_streams_have_the_requested_scale():
    """Without injected draws the kernels draw from Philox (seed, env, step, purpose): obs noise is N(0, sigma^2) around
    the clean observation, differs per env / step / seed, and is reproducible for a fixed seed."""
    from rllab_amd.envs.hip_env import HipVecEnv
    n, sigma = 4096, 0.25
    d0 = np.random.RandomState(0).rand(4, n).astype(np.float32)
    a = torch.zeros((n, 1), device="cuda")

    def run(seed, cfg):
        v = HipVecEnv(0, n, 0, normalize=True, seed=seed, cfg=cfg, auto_reset=False)
        first = v.reset(draws=d0).clone()
        return first, v.step(a)[0].clone()
    clean0, clean1 = run(1, {})
    n0, n1 = run(1, dict(obs_noise=sigma))
    m0, m1 = run(1, dict(obs_noise=sigma))
    assert torch.equal(n0, m0) and torch.equal(n1, m1)                 # reproducible
    o0, o1 = run(2, dict(obs_noise=sigma))
    assert not torch.equal(o0, n0)                                     # seed-dependent
    for noisy, clean in ((n0, clean0), (n1, clean1)):
        z = ((noisy - clean) / sigma).double().cpu().numpy()
        assert abs(z.mean()) < 0.03 and abs(z.std() - 1.0) < 0.03
        assert abs(np.corrcoef(z[:-1, 0], z[1:, 0])[0, 1]) < 0.06      # neighbouring envs are independent
    z0 = ((n0 - clean0) / sigma).double().cpu().numpy()
    z1 = ((n1 - clean1) / sigma).double().cpu().numpy()
    assert abs(np.corrcoef(z0[:, 1], z1[:, 1])[0, 1]) < 0.06           # consecutive steps are independent
    # action noise moves the cart: same state, zero action, different pushes
    _, p1 = run(1, dict(action_noise=0.5))
    assert float((p1 - clean1).abs().max()) > 1e-3


def test_position_only_and_the_env_classes(quiet_logger):
    """Box2DEnv(position_only=True): observations keep the position-typed <state> entries (cartpole: cart x, pole
    angle); such an env stays on the fused rollout (test_position_only_on_the_fused_rollout) and TRPO runs on it.  Reward-coefficient
    and noise options reach the kernels through the rllab classes."""
    from rllab.algos.trpo import TRPO
    from rllab.baselines.linear_feature_baseline import LinearFeatureBaseline
    from rllab.envs.box2d.cartpole_env import CartpoleEnv
    from rllab.envs.mujoco.swimmer_env import SwimmerEnv
    from rllab.envs.normalized_env import normalize
    from rllab.misc import ext, logger
    from rllab.policies.gaussian_mlp_policy import GaussianMLPPolicy
    ext.set_seed(3)
    env = normalize(CartpoleEnv(position_only=True, obs_noise=0.01))
    assert env.observation_space.flat_dim == 2 and env.spec.observation_space.flat_dim == 2
    o = env.reset()
    assert o.shape == (2,)
    o2, r, d, _ = env.step(np.array([0.3]))
    assert o2.shape == (2,) and np.isfinite(r)
    policy = GaussianMLPPolicy(env_spec=env.spec, hidden_sizes=(32, 32))
    algo = TRPO(env=env, policy=policy, baseline=LinearFeatureBaseline(env_spec=env.spec), batch_size=64 * 50,
                max_path_length=50, n_itr=3, discount=0.99, step_size=0.01, sampler_args=dict(n_envs=64))
    algo.train()
    assert np.isfinite(policy.get_param_values()).all()
    assert algo.sampler.sampling_path(policy)[0].startswith("fused rollout kernel")
    sw = SwimmerEnv(ctrl_cost_coeff=0.0)
    sw.reset()
    _, r0, _, _ = sw.step(np.array([40.0, -40.0]))
    np.testing.assert_allclose(sw.get_body_comvel("torso")[0], r0, atol=1e-6)      # no control cost left
    assert sw.get_body_comvel("torso").shape == (3,) and sw.get_body_com("torso")[2] == 0.0


@pytest.mark.parametrize("kind,ids", [(0, (0, 2)), (1, (0, 1, 3, 4))])
@pytest.mark.parametrize("hidden", [(32, 32), (16, 16), (64, 64), (100, 50, 25)])
def test_position_only_on_the_fused_rollout(kind, ids, hidden):
    """Box2DEnv(position_only=True) (box2d_env.py:219-227: noise on the full observation, then the filter) in ONE launch:
    the policy built on the kept rows runs as the same net on the full observation with zero first-layer rows at the
    dropped entries -- actions, rewards and dones are those of a full-observation executor running that net, bit for
    bit; the batch holds the kept rows; recorded means == float64 forward of the policy on them."""
    from rllab_amd.envs.env_spec import EnvSpec
    from rllab_amd.envs.hip_env import HipVecEnv
    from rllab_amd.policies.gaussian_mlp_policy import GaussianMLPPolicy
    from rllab_amd.spaces import Box
    from oracle.replay import replay_check
    n, T, mpl = 70, 25, 11
    cfg = dict(obs_noise=0.05)
    pos = HipVecEnv(kind, n, mpl, normalize=True, seed=5, cfg=cfg, position_ids=ids)
    full = HipVecEnv(kind, n, mpl, normalize=True, seed=5, cfg=cfg)
    q = full.q
    do, da, kept = q["obs_dim"], q["act_dim"], len(ids)
    assert pos.obs_rows == kept

    def build(d):
        np.random.seed(0)
        spec = EnvSpec(Box(-1e6 * np.ones(d), 1e6 * np.ones(d)), Box(-np.ones(da), np.ones(da)))
        return GaussianMLPPolicy(spec, hidden_sizes=hidden)
    pol, pol_full = build(kept), build(do)
    th = pol.get_param_values()
    th = th + 0.1 * np.random.RandomState(3).randn(th.size)
    pol.set_param_values(th)
    h0 = hidden[0]
    w0 = np.zeros((do, h0))
    w0[list(ids)] = th[:kept * h0].reshape(kept, h0)
    pol_full.set_param_values(np.concatenate([w0.reshape(-1), th[kept * h0:]]))
    assert pos.takes_rollout_of(pol)
    rng = np.random.RandomState(1)
    eps = rng.randn(da, T, n).astype(np.float32)
    draws = (rng.randn if q["reset_is_normal"] else rng.rand)(T + 1, q["reset_draws"], n).astype(np.float32)
    oz = rng.randn(T + 1, do, n).astype(np.float32)
    a = pos.rollout(pol, T, eps=eps, reset_draws=draws, obs_noise_z=oz)
    b = full.rollout(pol_full, T, eps=eps, reset_draws=draws, obs_noise_z=oz)
    assert a.obs.shape == (kept, T, n) and a.obs.is_contiguous()
    assert torch.equal(a.obs, b.obs[list(ids)])
    for x, y in ((a.actions, b.actions), (a.means, b.means), (a.rewards, b.rewards), (a.dones, b.dones)):
        assert torch.equal(x, y)
    assert torch.equal(pos.state, full.state)
    assert torch.equal(pos._filtered(pos._obs), full._obs[list(ids)].t())
    assert replay_check(full, b, max_envs=n, reset_draws=draws, obs_noise_z=oz) == n * T
    with torch.no_grad():
        mean64 = pol.mean_planes(a.obs.reshape(kept, -1).double(), pol.flat_params.double())
    assert float((a.means.reshape(da, -1).double() - mean64).abs().max()) <= 1e-5


@pytest.mark.parametrize("name", ["half_cheetah", "walker2d", "hopper"])
def test_soft_constraint_models_through_the_env_classes(name, quiet_logger):
    """HalfCheetahEnv / Walker2DEnv / HopperEnv(limit_model="mujoco", contact_model="mujoco"): the options reach the kernels
    as rl_env_cfg flags, such an env is sampled by the fused env-per-lane rollout (the one-body-per-lane kernels are built
    for the penalty models), its first batch replays on the host build bit for bit, and TRPO steps on it."""
    import importlib
    from rllab.algos.trpo import TRPO
    from rllab.baselines.linear_feature_baseline import LinearFeatureBaseline
    from rllab.envs.normalized_env import normalize
    from rllab.misc import ext
    from rllab.policies.gaussian_mlp_policy import GaussianMLPPolicy
    from rllab_amd import _lib
    cls = dict(half_cheetah="HalfCheetahEnv", walker2d="Walker2DEnv", hopper="HopperEnv")[name]
    Env = getattr(importlib.import_module("rllab.envs.mujoco.%s_env" % name), cls)
    with pytest.raises(ValueError):
        Env(contact_model="box2d")
    ext.set_seed(2)
    env = normalize(Env(limit_model="mujoco", contact_model="mujoco"))
    policy = GaussianMLPPolicy(env_spec=env.spec, hidden_sizes=(32, 32))
    algo = TRPO(env=env, policy=policy, baseline=LinearFeatureBaseline(env_spec=env.spec), batch_size=96 * 40,
                max_path_length=40, n_itr=2, discount=0.99, step_size=0.01, sampler_args=dict(n_envs=96, seed=5))
    algo.start_worker()
    algo.init_opt()
    v = algo.sampler.vec_env
    assert int(v.cfg.flags) == _lib.CFG_LIMIT_MUJOCO | _lib.CFG_CONTACT_MUJOCO
    assert algo.sampler.sampling_path(policy)[0].startswith("fused rollout kernel")
    assert v.rollout_plan(policy, 40).name.decode().startswith("rollout_kernel<")
    pickled = __import__("pickle").loads(__import__("pickle").dumps(env))
    assert pickled.wrapped_env.contact_model == "mujoco" and pickled.wrapped_env.limit_model == "mujoco"
    theta0 = policy.get_param_values()
    for itr in range(2):
        paths = algo.sampler.obtain_samples(itr)
        sd = algo.sampler.process_samples(itr, paths)
        algo.optimize_policy(itr, sd)
    assert np.isfinite(policy.get_param_values()).all() and np.abs(policy.get_param_values() - theta0).max() > 0
