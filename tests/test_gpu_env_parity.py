"""GPU parity tests proper (``-m gpu``): every call goes through the C ABI of
librllab_amd.so; the checker is the host oracle (oracle/)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ENV_KINDS = [0, 1, 2, 3, 4, 5, 6, 7]
OBS_INVERTIBLE = [0, 2, 4]   # kinds whose reset state can be rebuilt from the observation


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda", 0)


def test_philox_device_matches_host():
    from rllab_amd import _lib
    from oracle import host_env as H
    n = 1000
    out = torch.zeros(4 * n, dtype=torch.int32, device=_dev())
    args = (12345, 7, 0xdeadbeef, 0x52455345, 0x1234abcd, 0x9e3779b9)
    _lib.check(_lib.lib.rl_debug_philox(*args, n, _lib.ptr(out), _lib.stream_ptr()))
    got = out.cpu().numpy().view(np.uint32).reshape(n, 4)
    want = H.philox(*args, n)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("kind", ENV_KINDS)
@pytest.mark.parametrize("normalize", [False, True])
def test_vecenv_step_bit_exact(kind, normalize):
    """reset + 60 lock-step transitions with auto-reset and a horizon: every state
    plane, observation, reward and done flag equals the host build bit for bit."""
    from rllab_amd.envs.hip_env import HipVecEnv
    from oracle import host_env as H
    rng = np.random.RandomState(0)
    n, mpl = 257, 25
    gpu = HipVecEnv(kind, n, mpl, normalize=normalize, scale_reward=0.5, seed=3)
    cpu = H.HostVecEnv(kind, n, mpl, normalize=normalize, scale_reward=0.5)
    q = gpu.q
    draw = (lambda: rng.randn(q["reset_draws"], n)) if q["reset_is_normal"] else \
        (lambda: rng.rand(q["reset_draws"], n))
    d0 = draw().astype(np.float32)
    o_gpu = gpu.reset(draws=d0)
    o_cpu = cpu.reset(d0)
    assert np.array_equal(o_gpu.t().cpu().numpy().view(np.uint32), o_cpu.view(np.uint32))
    lb, ub = H.HostEnv(kind).q, None
    n_done = 0
    for t in range(60):
        scale = 1.0 if normalize else 10.0
        a = (rng.randn(n, q["act_dim"]) * scale).astype(np.float32)
        dr = draw().astype(np.float32)
        og, rg, dg, _ = gpu.step(torch.as_tensor(a, device=gpu.device), reset_draws=dr)
        oc, rc, dc = cpu.step(a.T, dr)
        assert np.array_equal(gpu.state.cpu().numpy().view(np.uint32), cpu.state.view(np.uint32)), t
        assert np.array_equal(og.t().cpu().numpy().view(np.uint32), oc.view(np.uint32)), t
        assert np.array_equal(rg.cpu().numpy().view(np.uint32), rc.view(np.uint32)), t
        assert np.array_equal(dg.cpu().numpy(), dc.astype(bool)), t
        assert np.array_equal(gpu.ts.cpu().numpy(), cpu.ts), t
        n_done += int(dc.sum())
    assert n_done > 0  # the auto-reset branch (env done or horizon) was exercised


def _make_policy(kind, hidden=(32, 32), seed=0):
    from rllab_amd import _lib
    from rllab_amd.envs.env_spec import EnvSpec
    from rllab_amd.policies.gaussian_mlp_policy import GaussianMLPPolicy
    from rllab_amd.spaces import Box
    q = _lib.env_query(kind)
    np.random.seed(seed)
    spec = EnvSpec(Box(-1e6 * np.ones(q["obs_dim"]), 1e6 * np.ones(q["obs_dim"])),
                   Box(-np.ones(q["act_dim"]), np.ones(q["act_dim"])))
    return GaussianMLPPolicy(spec, hidden_sizes=hidden)


@pytest.fixture(params=["auto", "64", "wpb4", "leg16"])
def rollout_shape(request, monkeypatch):
    """The launch shapes of the fused rollout: "auto" = at these sizes 16 envs per wavefront (policy on 16x16x4 tiles,
    physics replicated on four lanes; the Swimmer four lanes per env) in single-wavefront workgroups -- and ONE ENV PER
    WAVEFRONT for the two-legged envs (policy units on the lanes, rollout_two_leg_wave_kernel); "64" = one env per lane;
    "wpb4" = the auto shapes in workgroups of four wavefronts (what launches beyond 256 wavefronts use);
    "leg16" = the two-legged envs in their 16-envs-per-wavefront shape (what they use beyond 2048 envs;
    csrc/env_kernels.hip)."""
    monkeypatch.delenv("RLLAB_ROLLOUT_EPW", raising=False)
    monkeypatch.delenv("RLLAB_ROLLOUT_WPB", raising=False)
    monkeypatch.delenv("RLLAB_TWO_LEG_WAVE_KERNEL", raising=False)
    if request.param == "64":
        monkeypatch.setenv("RLLAB_ROLLOUT_EPW", "64")
    elif request.param == "wpb4":
        monkeypatch.setenv("RLLAB_ROLLOUT_WPB", "4")
    elif request.param == "leg16":
        monkeypatch.setenv("RLLAB_TWO_LEG_WAVE_KERNEL", "0")
    return request.param


@pytest.mark.parametrize("kind", ENV_KINDS)
@pytest.mark.parametrize("hidden", [(32, 32), (64, 64)])
def test_fused_rollout_injected_noise(kind, hidden, rollout_shape):
    """Fused rollout with injected policy noise and reset draws:
       * env dynamics replayed on the host oracle from the recorded actions: bit-exact;
       * recorded means vs a float64 torch forward of the same theta: <= 1e-5;
       * recorded action == mean + eps * exp(log_std) (<= 1 ulp-ish, fma vs mul+add)."""
    from rllab_amd.envs.hip_env import HipVecEnv
    from oracle.replay import replay_check
    rng = np.random.RandomState(1)
    n, T, mpl = 130, 40, 17
    policy = _make_policy(kind, hidden)
    v = HipVecEnv(kind, n, mpl, normalize=True, seed=11)
    q = v.q
    eps = rng.randn(q["act_dim"], T, n).astype(np.float32)
    draws = (rng.randn if q["reset_is_normal"] else rng.rand)(T + 1, q["reset_draws"], n).astype(np.float32)
    traj = v.rollout(policy, T, reset_at_start=True, eps=eps, reset_draws=draws)
    torch.cuda.synchronize()
    assert replay_check(v, traj, max_envs=n, reset_draws=draws) == n * T
    assert int(traj.dones.sum()) > 0
    # policy parity
    obs64 = traj.obs.reshape(q["obs_dim"], -1).double()
    with torch.no_grad():
        mean64 = policy.mean_planes(obs64, policy.flat_params.double())
    got = traj.means.reshape(q["act_dim"], -1).double()
    assert float((got - mean64).abs().max()) <= 1e-5
    std = torch.exp(policy.effective_log_std().double())[:, None]
    act64 = got + torch.as_tensor(eps, device=got.device).reshape(q["act_dim"], -1).double() * std
    assert float((traj.actions.reshape(q["act_dim"], -1).double() - act64).abs().max()) <= 1e-6
    # reset draws were consumed from the right slice: the first observation of every env is the
    # host oracle's reset() of slice 0 (bit-exact)
    from oracle import host_env as H
    o0 = traj.obs[:, 0, :].cpu().numpy()
    for i in range(0, n, 13):
        want = H.HostEnv(kind, np.float32).reset(draws[0, :, i])
        assert np.array_equal(o0[:, i].view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("kind,scale", [(2, 30.0), (2, 100.0), (3, 30.0), (5, 30.0), (6, 30.0)])
def test_fused_rollout_from_wild_states(kind, scale, rollout_shape):
    """The same bit-exact replay from far outside the reset distribution (injected reset draws scaled by 30 / 100:
    hinges beyond their limits, body rates of tens of rad/s, strong policy noise): the penalty branches, the packed
    pair arithmetic, the reciprocal of the solve and the carried sines of the Swimmer's lane-group program see large
    arguments, and every recorded transition must still be the host build's, bit for bit."""
    from rllab_amd.envs.hip_env import HipVecEnv
    from oracle.replay import replay_check
    rng = np.random.RandomState(7)
    n, T, mpl = 96, 60, 25
    policy = _make_policy(kind, seed=3)
    v = HipVecEnv(kind, n, mpl, normalize=True, seed=2)
    q = v.q
    eps = (3.0 * rng.randn(q["act_dim"], T, n)).astype(np.float32)
    draws = (scale * (rng.randn if q["reset_is_normal"] else rng.rand)(T + 1, q["reset_draws"], n)).astype(np.float32)
    traj = v.rollout(policy, T, reset_at_start=True, eps=eps, reset_draws=draws)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(traj.obs).all()) and bool(torch.isfinite(traj.rewards).all())
    assert replay_check(v, traj, max_envs=n, reset_draws=draws) == n * T


@pytest.mark.parametrize("kind", OBS_INVERTIBLE)
def test_fused_rollout_production_rng(kind, rollout_shape):
    """Production mode (in-kernel Philox): dynamics still replay bit-exactly, the
    policy noise has the right moments, and two runs with the same seed / counter agree."""
    from rllab_amd.envs.hip_env import HipVecEnv
    from oracle.replay import replay_check
    n, T = 4096, 50
    policy = _make_policy(kind)
    v = HipVecEnv(kind, n, T, normalize=True, seed=5)
    traj = v.rollout(policy, T)
    v2 = HipVecEnv(kind, n, T, normalize=True, seed=5)
    traj2 = v2.rollout(policy, T)
    torch.cuda.synchronize()
    assert torch.equal(traj.actions, traj2.actions) and torch.equal(traj.obs, traj2.obs)
    replay_check(v, traj, max_envs=32)
    z = (traj.actions - traj.means) / torch.exp(traj.log_std)[:, None, None]
    z = z.double().reshape(-1)
    assert abs(float(z.mean())) < 0.01 and abs(float(z.std()) - 1.0) < 0.01
    assert abs(float((z ** 4).mean()) - 3.0) < 0.1
    # a different seed gives a different stream
    v3 = HipVecEnv(kind, n, T, normalize=True, seed=6)
    assert not torch.equal(v3.rollout(policy, T).actions, traj.actions)


def _ref_scan(x, c):
    y = np.zeros_like(x, dtype=np.float64)
    acc = np.zeros(x.shape[1])
    for t in range(x.shape[0] - 1, -1, -1):
        acc = x[t] + c[t] * acc
        y[t] = acc
    return y


@pytest.mark.parametrize("T,n", [(1, 1), (7, 3), (100, 257), (500, 1000), (513, 64)])
def test_gae_kernel_vs_float64_loop(T, n):
    from rllab_amd import _lib
    rng = np.random.RandomState(T * 1000 + n)
    dev = _dev()
    r = rng.randn(T, n).astype(np.float32)
    v = rng.randn(T, n) * 3
    done = (rng.rand(T, n) < 0.05).astype(np.uint8)
    gamma, lam = 0.99, 0.97
    adv = torch.empty((T, n), dtype=torch.float32, device=dev)
    ret = torch.empty((T, n), dtype=torch.float32, device=dev)
    tr, tv, td = (torch.as_tensor(x, device=dev) for x in (r, v, done))
    _lib.check(_lib.lib.rl_gae(T, n, _lib.ptr(tr), _lib.ptr(tv), _lib.ptr(td), gamma, lam, _lib.ptr(adv),
                               _lib.ptr(ret), None, _lib.stream_ptr()))
    end = done.astype(bool).copy()
    end[-1] = True
    keep = 1.0 - end
    vnext = np.vstack([v[1:], np.zeros((1, n))])
    delta = r.astype(np.float64) + gamma * vnext * keep - v
    want_adv = _ref_scan(delta, gamma * lam * keep)
    want_ret = _ref_scan(r.astype(np.float64), gamma * keep)
    scale = max(1.0, np.abs(want_adv).max())
    assert np.abs(adv.cpu().numpy() - want_adv).max() <= 1e-5 * scale
    assert np.abs(ret.cpu().numpy() - want_ret).max() <= 1e-5 * max(1.0, np.abs(want_ret).max())
    # values=None path
    _lib.check(_lib.lib.rl_gae(T, n, _lib.ptr(tr), None, _lib.ptr(td), gamma, 1.0, _lib.ptr(adv),
                               _lib.ptr(ret), None, _lib.stream_ptr()))
    assert np.abs(adv.cpu().numpy() - want_ret).max() <= 1e-5 * max(1.0, np.abs(want_ret).max())


def test_discount_cumsum_special():
    from rllab_amd.misc import special
    # known answer from the reference run in the survey container
    y = special.discount_cumsum(np.arange(5, dtype=np.float64), 0.9)
    assert np.allclose(y, [7.3314, 8.146, 7.94, 6.6, 4.0], atol=1e-5)
    rng = np.random.RandomState(0)
    x = rng.randn(500, 33)
    want = _ref_scan(x, np.full_like(x, 0.99))
    assert np.abs(special.discount_cumsum(x, 0.99) - want).max() <= 1e-5 * np.abs(want).max()


def test_smoke_entry():
    import __graft_entry__
    __graft_entry__.smoke()


def test_single_env_state_api():
    """State-level methods of the reference env bases on the 1-env executor: get_current_obs == the observation
    step / reset returned, set_state(get_state()) is the identity, reset_mujoco / inject_action_noise consume
    np.random like the reference, get_body_com('torso') is the COM the observation carries."""
    from rllab_amd.envs.box2d.cartpole_env import CartpoleEnv
    from rllab_amd.envs.mujoco.hopper_env import HopperEnv
    from rllab_amd.envs.mujoco.swimmer_env import SwimmerEnv
    env = CartpoleEnv()
    o0 = env.reset()
    assert np.array_equal(env.get_current_obs(), o0)
    o1 = env.step(np.array([0.3]))[0]
    assert np.array_equal(env.get_current_obs(), o1)
    st = env.get_state()
    assert st.shape == (16,)
    o2 = env.step(np.array([-0.7]))[0]
    env.set_state(st)                                   # rewind: the same step again gives the same observation
    assert np.array_equal(env.get_current_obs(), o1)
    assert np.array_equal(env.step(np.array([-0.7]))[0], o2)

    sw = SwimmerEnv()
    sw.reset()
    np.random.seed(4)
    sw.reset_mujoco()
    np.random.seed(4)
    want = np.concatenate([np.random.normal(size=5) * 0.01, np.random.normal(size=5) * 0.1])
    obs = sw.get_current_obs()
    assert np.allclose(obs[:10], want.astype(np.float32), rtol=0, atol=0)
    assert np.array_equal(sw.get_body_com("torso"), obs[-3:]) and obs[-1] == 0.0
    sw.reset_mujoco(init_state=np.arange(20) * 0.01)    # [qpos, qvel, qacc, ctrl]: the first ten entries count
    assert np.allclose(sw.get_current_obs()[:10], (np.arange(10) * 0.01).astype(np.float32), atol=0)
    np.random.seed(9)
    a = sw.inject_action_noise(np.array([0.2, -0.1]))
    assert np.array_equal(a, np.array([0.2, -0.1]))     # scale 0 ...
    np.random.seed(9)
    np.random.normal(size=2)
    nxt = np.random.rand()
    np.random.seed(9)
    sw.inject_action_noise(np.zeros(2))
    assert np.random.rand() == nxt                      # ... but the draw was made
    # subtree COM / COM velocity of the torso come from rl_vecenv_com; other bodies are not exported
    assert sw.get_body_comvel("torso").shape == (3,) and sw.get_body_comvel("torso")[2] == 0.0
    hp = HopperEnv()
    hp.reset()
    com = hp.get_body_com("torso")
    assert com.shape == (3,) and com[1] == 0.0 and 0.5 < com[2] < 1.5          # (x, 0, z): the hopper stands ~1 m tall
    with pytest.raises(NotImplementedError):
        sw.get_body_com("mid")


def test_stepwise_rollout_through_a_hip_graph(quiet_logger, monkeypatch):
    """A policy without a fused rollout (an architecture the kernels are not built for) is sampled one transition at
    a time; that loop is captured into a hipGraph and replayed.  The recorded batch must be a valid rollout: env
    dynamics replay bit-exactly on the host, recorded means are the policy's, the noise is fresh in every step and
    every call, episodes reset with fresh draws, and it agrees in distribution with the eager loop."""
    import time
    from oracle.replay import replay_check
    from rllab_amd.algos.trpo import TRPO
    from rllab_amd.baselines.linear_feature_baseline import LinearFeatureBaseline
    from rllab_amd.envs.box2d.cartpole_env import CartpoleEnv
    from rllab_amd.envs.normalized_env import normalize
    from rllab_amd.misc import ext
    from rllab_amd.policies.gaussian_mlp_policy import GaussianMLPPolicy
    ext.set_seed(3)
    torch.manual_seed(3)
    env = normalize(CartpoleEnv())
    pol = GaussianMLPPolicy(env_spec=env.spec, hidden_sizes=(16, 16, 16, 16))    # four layers: no fused kernel
    assert pol.kernel_layout() is None
    n, T = 512, 60
    algo = TRPO(env=env, policy=pol, baseline=LinearFeatureBaseline(env_spec=env.spec), batch_size=n * T,
                max_path_length=T, n_itr=1, sampler_args=dict(n_envs=n, seed=3))
    algo.start_worker()
    s = algo.sampler
    tr = s.obtain_samples(0).traj
    assert getattr(s, "_step_graph", None) is not None                      # the graph path was taken
    # (Cartpole terminates: the loop runs on past T lock steps until n * T samples are in finished paths)
    assert tr.N == n and T <= tr.T < 2 * T and int(tr.dones.sum()) >= n and int(s._finished_by_step(tr)[-1]) >= n * T
    assert replay_check(s.vec_env, tr, max_envs=32) > 0
    with torch.no_grad():
        mean64 = pol.mean_planes(tr.obs.reshape(4, -1).double(), pol.flat_params.double())
    assert float((tr.means.reshape(1, -1).double() - mean64).abs().max()) <= 1e-5
    z = ((tr.actions - tr.means) / torch.exp(tr.log_std)[:, None, None]).double().reshape(-1)
    assert abs(float(z.mean())) < 0.02 and abs(float(z.std()) - 1.0) < 0.02
    assert not torch.equal(tr.actions[:, 0], tr.actions[:, 1])               # noise differs from step to step
    tr2 = s.obtain_samples(1).traj
    assert not torch.equal(tr2.actions, tr.actions) and not torch.equal(tr2.obs[:, 0], tr.obs[:, 0])
    # (no host replay of the second batch: Cartpole's warm-start impulses persist across reset() and are not part of
    #  the observation the replay starts from -- only a fresh executor's first rollout can be replayed)
    # same distribution as the eager loop (selected by the RLLAB_NO_GRAPH switch): mean episode length within a few per cent
    monkeypatch.setenv("RLLAB_NO_GRAPH", "1")
    tr3 = s.obtain_samples(2).traj
    monkeypatch.delenv("RLLAB_NO_GRAPH")
    ep = lambda t: float(t.dones.sum()) / (t.N * t.T)          # episodes ended per env-step
    assert abs(ep(tr3) - ep(tr2)) <= 0.15 * ep(tr2)
    # and the point of it: fewer launches per transition
    s.use_graph = True
    torch.cuda.synchronize()
    t0 = time.time(); s.obtain_samples(3); torch.cuda.synchronize(); t_graph = time.time() - t0
    s.use_graph = False
    t0 = time.time(); s.obtain_samples(4); torch.cuda.synchronize(); t_eager = time.time() - t0
    print("stepwise rollout %d envs x %d steps: hipGraph %.2f ms, eager %.2f ms" % (n, T, t_graph * 1e3, t_eager * 1e3))
    assert t_graph < t_eager


@pytest.mark.parametrize("kind,T", [(0, 23), (2, 23), (3, 23), (3, 150), (5, 70)])
def test_policy_noise_stream_is_the_same_in_every_launch_shape(kind, T, monkeypatch):
    """In-kernel policy noise is a function of (seed, global env index, step) only: the lane-group shapes draw four
    steps at once on the four replicas of an env, the one-env-per-wavefront shape of the two-legged envs 64 steps at once
    on its 64 lanes (T = 150, 70: more than one such group, the last partly unused), the env-per-lane shape one step per
    launch iteration -- the standardised noise (action - mean) / std must agree (to the rounding of the two policy
    forward passes)."""
    from rllab_amd.envs.hip_env import HipVecEnv
    policy = _make_policy(kind, (32, 32))
    n = 100                           # T = 23: not a multiple of four, the last group of draws is partly unused

    def run():
        v = HipVecEnv(kind, n, 9, normalize=True, seed=21)
        tr = v.rollout(policy, T, reset_at_start=True)
        std = torch.exp(policy.effective_log_std())[:, None, None]
        return ((tr.actions - tr.means) / std).cpu().numpy(), tr.obs.cpu().numpy()
    monkeypatch.delenv("RLLAB_ROLLOUT_EPW", raising=False)
    monkeypatch.delenv("RLLAB_SWIMMER_LANE_KERNEL", raising=False)
    monkeypatch.delenv("RLLAB_TWO_LEG_WAVE_KERNEL", raising=False)
    z16, o16 = run()
    monkeypatch.setenv("RLLAB_ROLLOUT_EPW", "64")
    monkeypatch.setenv("RLLAB_SWIMMER_LANE_KERNEL", "1")
    z64, o64 = run()
    assert np.array_equal(o16[:, 0], o64[:, 0])                       # same reset draws
    assert np.abs(z16[:, 0] - z64[:, 0]).max() < 1e-5                 # step 0: same observation, same noise
    # later steps: trajectories drift apart by rounding, the noise does not depend on them
    assert np.abs(z16 - z64).max() < 2e-3 and abs(z16.std() - 1.0) < 0.05
    assert np.abs(z16[:, 1:] - z16[:, :-1]).max() > 0.1               # fresh per step
