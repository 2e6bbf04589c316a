"""GPU parity of sample post-processing (rl_gae + device statistics + dense baseline) against
fixtures produced by the REAL reference ``BaseSampler.process_samples`` /
``LinearFeatureBaseline`` (tests/golden/, oracle/make_golden.py), and size-independent
properties at BASELINE.json's full sizes."""
import os

import numpy as np
import pytest
import torch

from tests.golden_util import STAT_KEYS, load, unpack_paths

pytestmark = pytest.mark.gpu


class _Algo(object):
    def __init__(self, g, act_dim):
        from rllab_amd.baselines.linear_feature_baseline import LinearFeatureBaseline
        from rllab_amd.distributions.diagonal_gaussian import DiagonalGaussian
        self.baseline = LinearFeatureBaseline(env_spec=None)
        self.discount, self.gae_lambda = float(g["discount"]), float(g["gae_lambda"])
        self.center_adv, self.positive_adv = bool(g["center_adv"]), bool(g["positive_adv"])
        self.whole_paths = True

        class _P(object):
            recurrent = False
            distribution = DiagonalGaussian(act_dim)
        self.policy = _P()


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_process_samples_matches_reference_within_1e5(tag, quiet_logger):
    """returns / advantages within 1e-5 of the reference's float64 post-processing, over two
    iterations (zero baseline, then the fitted LinearFeatureBaseline), tabular stats included."""
    from rllab_amd.misc import logger
    from rllab_amd.sampler.base import BaseSampler
    g = load("process_samples_" + tag)
    paths = unpack_paths(g)
    algo = _Algo(g, paths[0]["actions"].shape[1])
    sampler = BaseSampler(algo)
    for it, (ka, kr, kc, ks) in enumerate([("adv1", "ret1", "coeffs1", "stats1"), ("adv2", "ret2", "coeffs2", "stats2")]):
        sd = sampler.process_samples(it, [dict(p) for p in paths])
        tab = logger.get_tabular()
        logger.dump_tabular()
        adv = sd["advantages"].double().cpu().numpy()
        ret = sd["returns"].double().cpu().numpy()
        # pack_paths lays one path per column (t-major flattening) -> regroup per path
        tr = sd["_traj"]
        lens = g["lens"]
        cols = np.concatenate([np.full(L, i) for i, L in enumerate(lens)])
        rows = np.concatenate([np.arange(L) for L in lens])
        dense_adv = tr.advantages.double().cpu().numpy()[rows, cols]
        dense_ret = tr.returns.double().cpu().numpy()[rows, cols]
        assert np.abs(dense_ret - g[kr]).max() <= 1e-5 * max(1.0, np.abs(g[kr]).max())
        assert np.abs(dense_adv - g[ka]).max() <= 1e-5 * max(1.0, np.abs(g[ka]).max())
        assert adv.shape == g[ka].shape and ret.shape == g[kr].shape
        assert np.isclose(np.sort(adv), np.sort(dense_adv)).all()
        pred_ref = np.concatenate([_feat(p).dot(g[kc]) for p in paths])
        pred_got = np.concatenate([_feat(p).dot(algo.baseline.get_param_values()) for p in paths])
        assert np.abs(pred_got - pred_ref).max() <= 1e-5 * max(1.0, np.abs(pred_ref).max())
        want = dict(zip(STAT_KEYS, g[ks]))
        for k in STAT_KEYS:
            assert np.isclose(float(tab[k]), want[k], rtol=2e-5, atol=2e-5), (k, tab[k], want[k])


def _feat(path):
    o = np.clip(path["observations"], -10, 10)
    l = len(path["rewards"])
    al = np.arange(l).reshape(-1, 1) / 100.0
    return np.concatenate([o, o ** 2, al, al ** 2, al ** 3, np.ones((l, 1))], axis=1)


def test_gae_full_size_properties():
    """4096 x 500 (config C3): closed-form geometric series, linearity, and segment isolation."""
    from rllab_amd import _lib
    dev = torch.device("cuda", 0)
    T, n, gamma = 500, 4096, 0.99
    ones = torch.ones((T, n), device=dev)
    done = torch.zeros((T, n), dtype=torch.uint8, device=dev)
    adv = torch.empty_like(ones)
    ret = torch.empty_like(ones)

    def gae(r, v, d, lam=1.0):
        _lib.check(_lib.lib.rl_gae(T, n, _lib.ptr(r), _lib.ptr(v), _lib.ptr(d), gamma, lam, _lib.ptr(adv),
                                   _lib.ptr(ret), None, _lib.stream_ptr()))
        return adv.clone(), ret.clone()
    _, r1 = gae(ones, None, done)
    k = torch.arange(T, 0, -1, device=dev, dtype=torch.float64)
    closed = (1 - gamma ** k) / (1 - gamma)
    assert float((r1[:, 0].double() - closed).abs().max()) <= 1e-5 * float(closed.max())
    assert torch.equal(r1, r1[:, :1].expand(T, n))
    x = torch.randn((T, n), device=dev)
    y = torch.randn((T, n), device=dev)
    _, rx = gae(x, None, done)
    _, ry = gae(y, None, done)
    _, rxy = gae(x + 2 * y, None, done)
    assert float((rxy - (rx + 2 * ry)).abs().max()) <= 1e-3
    # a done flag isolates what follows it: changing rewards after a boundary leaves earlier returns alone
    d2 = done.clone()
    d2[249] = 1
    _, ra = gae(x, None, d2)
    x2 = x.clone()
    x2[250:] += 5.0
    _, rb = gae(x2, None, d2)
    assert torch.equal(ra[:250], rb[:250]) and not torch.equal(ra[250:], rb[250:])
    # with V == exact returns and lambda = 1 the advantage is zero
    v = r1.double()
    a0, _ = gae(ones, v, done)
    assert float(a0.abs().max()) <= 1e-4


def test_full_size_swimmer_rollout_properties(quiet_logger):
    """Config C3 at full size (4096 envs x 500 steps): determinism, one path per env, finite
    values, and encode->process->decode consistency of the dense batch."""
    from rllab_amd.algos.trpo import TRPO
    from rllab_amd.baselines.linear_feature_baseline import LinearFeatureBaseline
    from rllab_amd.envs.mujoco.swimmer_env import SwimmerEnv
    from rllab_amd.envs.normalized_env import normalize
    from rllab_amd.misc import ext, logger
    from rllab_amd.policies.gaussian_mlp_policy import GaussianMLPPolicy
    ext.set_seed(7)
    env = normalize(SwimmerEnv())
    pol = GaussianMLPPolicy(env_spec=env.spec, hidden_sizes=(32, 32))
    algo = TRPO(env=env, policy=pol, baseline=LinearFeatureBaseline(env_spec=env.spec), batch_size=4096 * 500,
                max_path_length=500, n_itr=1, sampler_args=dict(n_envs=4096, seed=7))
    algo.start_worker()
    algo.init_opt()
    paths = algo.sampler.obtain_samples(0)
    tr = paths.traj
    assert (tr.T, tr.N) == (500, 4096) and len(paths) == 4096
    assert int(tr.dones.sum()) == 4096 and bool(tr.dones[-1].all())        # horizon ends every path
    assert all(bool(torch.isfinite(x).all()) for x in (tr.obs, tr.actions, tr.means, tr.rewards))
    sd = algo.sampler.process_samples(0, paths)
    tab = logger.get_tabular()
    assert int(tab["NumTrajs"]) == 4096
    adv = tr.advantages.double()
    assert abs(float(adv.mean())) < 1e-6 and abs(float(adv.std(unbiased=False)) - 1.0) < 1e-5
    assert sd["observations"].shape == (4096 * 500, 13) and sd["advantages"].shape == (4096 * 500,)
    # same seed, same counters -> identical batch
    algo2 = TRPO(env=env, policy=pol, baseline=LinearFeatureBaseline(env_spec=env.spec), batch_size=4096 * 500,
                 max_path_length=500, n_itr=1, sampler_args=dict(n_envs=4096, seed=7))
    algo2.start_worker()
    tr2 = algo2.sampler.obtain_samples(0).traj
    assert torch.equal(tr.obs, tr2.obs) and torch.equal(tr.rewards, tr2.rewards)
    # one TRPO step keeps the KL inside the trust region and does not increase the loss
    algo.optimize_policy(0, sd)
    tab = logger.get_tabular()
    assert float(tab["MeanKL"]) <= 0.01 + 1e-6 and float(tab["LossAfter"]) <= float(tab["LossBefore"])
    logger.dump_tabular()


def test_full_size_cheetah_c5_shard(quiet_logger):
    """Config C5's per-GPU shard at full size (HalfCheetah-style, 1024 envs x 500 steps, GaussianMLPPolicy(64,64),
    TRPO + GAE lambda 0.97): ALL 1024 x 500 recorded transitions replay bit-exactly on the host build of the dynamics (the
    "fp32 tolerance check vs CPU rollout" of the config, at tolerance zero), the GAE plane matches a float64 loop
    on sampled columns within 1e-5, and one TRPO step keeps the KL inside the trust region."""
    from oracle.replay import replay_check
    from rllab_amd.algos.trpo import TRPO
    from rllab_amd.baselines.linear_feature_baseline import LinearFeatureBaseline
    from rllab_amd.envs.mujoco.half_cheetah_env import HalfCheetahEnv
    from rllab_amd.envs.normalized_env import normalize
    from rllab_amd.misc import ext, logger
    from rllab_amd.policies.gaussian_mlp_policy import GaussianMLPPolicy
    ext.set_seed(3)
    env = normalize(HalfCheetahEnv())
    pol = GaussianMLPPolicy(env_spec=env.spec, hidden_sizes=(64, 64))
    gamma, lam = 0.99, 0.97
    algo = TRPO(env=env, policy=pol, baseline=LinearFeatureBaseline(env_spec=env.spec), batch_size=1024 * 500,
                max_path_length=500, n_itr=1, discount=gamma, gae_lambda=lam, sampler_args=dict(n_envs=1024, seed=3))
    algo.start_worker()
    algo.init_opt()
    from rllab_amd.sampler.trajectories import PathList
    v = algo.sampler.vec_env
    # reset draws are injected (the observation leaves out the root x, so the host replay needs the draws);
    # the policy noise is the production in-kernel Philox stream
    q = v.q
    draws = (np.random.RandomState(0).randn if q["reset_is_normal"] else np.random.RandomState(0).rand)(
        501, q["reset_draws"], 1024).astype(np.float32)
    # second iteration: the baseline is fitted, so the GAE plane is not just discounted returns
    for itr in range(2):
        tr = v.rollout(pol, 500, reset_at_start=True, reset_draws=draws)
        paths = PathList(tr)
        sd = algo.sampler.process_samples(itr, paths)
        if itr == 0:
            algo.optimize_policy(itr, sd)
            logger.dump_tabular()
    assert (tr.T, tr.N) == (500, 1024) and int(tr.dones.sum()) == 1024
    assert replay_check(v, tr, max_envs=1024, reset_draws=draws) == 1024 * 500      # every env of the shard, every step
    # GAE vs a float64 loop (sampler/base.py:57-66) on a few env columns; advantages were centred afterwards
    r = tr.rewards.double().cpu().numpy()
    v = tr.baselines.double().cpu().numpy()
    cols = [0, 17, 511, 1023]
    raw = np.zeros((500, len(cols)))
    for j, c in enumerate(cols):
        acc = 0.0
        for t in range(499, -1, -1):
            nxt = v[t + 1, c] if t < 499 else 0.0
            acc = (r[t, c] + gamma * nxt - v[t, c]) + gamma * lam * acc
            raw[t, j] = acc
    got = tr.advantages.double().cpu().numpy()[:, cols]
    # undo the centring with the batch's own moments: (raw - mean) / (std + 1e-8)
    a, b = np.polyfit(raw.reshape(-1), got.reshape(-1), 1)
    assert np.abs(a * raw + b - got).max() <= 1e-5 * max(1.0, np.abs(got).max())
    algo.optimize_policy(1, sd)
    tab = logger.get_tabular()
    assert float(tab["MeanKL"]) <= 0.01 + 1e-6 and float(tab["LossAfter"]) <= float(tab["LossBefore"])
    logger.dump_tabular()


def test_cartpole_whole_paths_drop_incomplete_tails(quiet_logger):
    """Ragged case: early terminations, auto-reset, trailing incomplete paths dropped when
    whole_paths (reference default) and kept as truncated paths otherwise."""
    from rllab_amd.algos.vpg import VPG
    from rllab_amd.baselines.linear_feature_baseline import LinearFeatureBaseline
    from rllab_amd.envs.box2d.cartpole_env import CartpoleEnv
    from rllab_amd.envs.normalized_env import normalize
    from rllab_amd.misc import ext, logger
    from rllab_amd.policies.gaussian_mlp_policy import GaussianMLPPolicy
    ext.set_seed(3)
    env = normalize(CartpoleEnv())
    pol = GaussianMLPPolicy(env_spec=env.spec, hidden_sizes=(32, 32))
    for whole in (True, False):
        algo = VPG(env=env, policy=pol, baseline=LinearFeatureBaseline(env_spec=env.spec), batch_size=512 * 100,
                   max_path_length=100, n_itr=1, whole_paths=whole, sampler_args=dict(n_envs=512, seed=3))
        algo.start_worker()
        algo.init_opt()
        paths = algo.sampler.obtain_samples(0)
        sd = algo.sampler.process_samples(0, paths)
        tr = paths.traj
        n_valid = int(tr.valid.sum())
        # at least batch_size samples in finished paths; whole_paths=False cuts the list to exactly batch_size
        assert (512 * 100 <= n_valid < tr.B) if whole else (n_valid == 512 * 100)
        assert sd["observations"].shape[0] == n_valid
        lens = [len(p["rewards"]) for p in list(paths)[:50]]
        assert max(lens) <= 100 and min(lens) >= 1
        # every listed path ends with a done flag when whole_paths
        env_i, t0, t1 = paths.index()
        if whole:
            assert bool(tr.dones[t1, env_i].all())
        theta0 = pol.get_param_values()
        algo.optimize_policy(0, sd)
        assert np.isfinite(pol.get_param_values()).all() and np.abs(pol.get_param_values() - theta0).max() > 0
        logger.dump_tabular()


def _np_tin_valid(done, whole_paths=True):
    T, n = done.shape
    tin = np.zeros((T, n), np.int64)
    valid = np.zeros((T, n), bool)
    for i in range(n):
        start = 0
        for t in range(T):
            if t > 0 and done[t - 1, i]:
                start = t
            tin[t, i] = t - start
        seen = False
        for t in range(T - 1, -1, -1):
            seen = seen or bool(done[t, i])
            valid[t, i] = seen or not whole_paths
    return tin, valid


def _np_features(obs, tin):
    """[B, F] float64 features in the reference's order (linear_feature_baseline.py:16-19)."""
    o = np.clip(obs.astype(np.float64), -10, 10)
    al = tin.astype(np.float64)[:, None] / 100.0
    return np.concatenate([o, o ** 2, al, al ** 2, al ** 3, np.ones_like(al)], axis=1)


@pytest.mark.parametrize("T,n,do", [(1, 1, 4), (7, 3, 6), (100, 257, 13), (513, 64, 20)])
@pytest.mark.parametrize("whole_paths", [True, False])
@pytest.mark.parametrize("valu_form", [False, True])
def test_path_scan_and_normal_equations_vs_numpy(T, n, do, whole_paths, valu_form, monkeypatch):
    """rl_path_scan (path index, validity, fused baseline prediction) and rl_lfb_normal_eq
    (Phi^T W Phi, Phi^T W y) against per-column numpy loops / a float64 feature matrix; the normal equations in both
    of their forms (f64 matrix instructions by default, RLLAB_LFB_VALU=1: the f64 vector-ALU kernel)."""
    if valu_form:
        monkeypatch.setenv("RLLAB_LFB_VALU", "1")
    else:
        monkeypatch.delenv("RLLAB_LFB_VALU", raising=False)
    from rllab_amd import _lib
    from rllab_amd.sampler.base import _workspace
    rng = np.random.RandomState(T * 100 + n)
    dev = torch.device("cuda", 0)
    done = (rng.rand(T, n) < 0.07).astype(np.uint8)
    obs = (rng.randn(do, T, n) * 6).astype(np.float32)          # some values beyond the +-10 clip
    F = 2 * do + 4
    coeffs = rng.randn(F)
    t_done, t_obs = torch.as_tensor(done, device=dev), torch.as_tensor(obs, device=dev)
    tin = torch.empty((T, n), dtype=torch.int32, device=dev)
    valid = torch.empty((T, n), dtype=torch.uint8, device=dev)
    values = torch.empty((T, n), dtype=torch.float64, device=dev)
    cf = torch.as_tensor(coeffs, device=dev)
    _lib.check(_lib.lib.rl_path_scan(T, n, do, _lib.ptr(t_done), _lib.ptr(t_obs), _lib.ptr(cf), int(whole_paths),
                                     _lib.ptr(tin), _lib.ptr(valid), _lib.ptr(values), _lib.stream_ptr()))
    want_tin, want_valid = _np_tin_valid(done, whole_paths)
    assert np.array_equal(tin.cpu().numpy(), want_tin)
    assert np.array_equal(valid.cpu().numpy().astype(bool), want_valid)
    phi = _np_features(obs.reshape(do, -1).T, want_tin.reshape(-1))
    want_v = phi.dot(coeffs)
    assert np.abs(values.cpu().numpy().reshape(-1) - want_v).max() <= 1e-11 * max(1.0, np.abs(want_v).max())
    # normal equations over the valid samples
    ret = rng.randn(T, n).astype(np.float32) * 30
    t_ret = torch.as_tensor(ret, device=dev)
    ws = _workspace(dev, do)
    out = torch.empty((F + 1) * F, dtype=torch.float64, device=dev)
    _lib.check(_lib.lib.rl_lfb_normal_eq(T * n, do, _lib.ptr(t_obs), _lib.ptr(tin), _lib.ptr(t_ret), _lib.ptr(valid),
                                         _lib.ptr(ws), ws.numel(), _lib.ptr(out),
                                         1 if os.environ.get("RLLAB_LFB_VALU") else 0, _lib.stream_ptr()))
    w = want_valid.reshape(-1).astype(np.float64)
    gram = (phi * w[:, None]).T.dot(phi)
    rhs = (phi * w[:, None]).T.dot(ret.reshape(-1).astype(np.float64))
    got = out.cpu().numpy()
    assert np.abs(got[:F * F].reshape(F, F) - gram).max() <= 1e-11 * max(1.0, np.abs(gram).max())
    assert np.abs(got[F * F:] - rhs).max() <= 1e-11 * max(1.0, np.abs(rhs).max())


@pytest.mark.parametrize("T,n", [(1, 1), (9, 7), (50, 100), (500, 4096)])
def test_sample_stats_and_adv_finish_vs_numpy(T, n):
    from rllab_amd import _lib
    from rllab_amd.sampler.base import _workspace
    B = T * n
    rng = np.random.RandomState(B % 1000)
    dev = torch.device("cuda", 0)
    ret = (rng.randn(B) * 20 + 300).astype(np.float32)
    base = rng.randn(B) * 20 + 290
    adv = (rng.randn(B) * 3 + 1).astype(np.float32)
    und = (rng.randn(B) * 50 + 500).astype(np.float32)
    x = (rng.randn(B) * 2).astype(np.float32)                  # the "progress" observation component
    done = (rng.rand(T, n) < 0.1).astype(np.uint8)
    tin2, valid2 = _np_tin_valid(done, True)
    if valid2.sum() == 0:
        done[-1, 0] = 1
        tin2, valid2 = _np_tin_valid(done, True)
    tin, valid = tin2.reshape(-1).astype(np.int32), valid2.reshape(-1).astype(np.uint8)
    t = lambda a: torch.as_tensor(a, device=dev)
    ws = _workspace(dev, 13)
    out = torch.empty(20, dtype=torch.float64, device=dev)
    args = [t(ret), t(base), t(adv), t(und), t(tin), t(valid)]
    tx = t(x)
    _lib.check(_lib.lib.rl_sample_stats(B, *[_lib.ptr(a) for a in args], 280.0, 450.0, _lib.ptr(tx), n, _lib.ptr(ws),
                                        ws.numel(), _lib.ptr(out), _lib.stream_ptr()))
    s = out.cpu().numpy()
    m = valid.astype(bool)
    st = m & (tin == 0)
    r64, a64, u64 = ret.astype(np.float64), adv.astype(np.float64), und.astype(np.float64)
    # per-path progress: x at the path's last step minus x at its first step, valid paths only
    x2 = x.reshape(T, n).astype(np.float64)
    progs = []
    for i in range(n):
        for tt in range(T):
            last = (tt == T - 1) or tin2[tt + 1, i] == 0
            if valid2[tt, i] and last:
                progs.append(x2[tt, i] - x2[tt - tin2[tt, i], i])
    progs = np.array(progs)
    want = [m.sum(), (r64[m] - 280).sum(), ((r64[m] - 280) ** 2).sum(), (base[m] - 280).sum(), ((base[m] - 280) ** 2).sum(),
            (r64[m] - base[m]).sum(), ((r64[m] - base[m]) ** 2).sum(), a64[m].sum(), (a64[m] ** 2).sum(), st.sum(),
            (u64[st] - 450).sum(), ((u64[st] - 450) ** 2).sum(), r64[st].sum(), progs.sum(), (progs ** 2).sum(),
            a64[m].min(), u64[st].max(), u64[st].min(), progs.max(), progs.min()]
    assert len(progs) == st.sum()
    for i, (g_, w_) in enumerate(zip(s, want)):
        assert abs(g_ - w_) <= 1e-10 * max(1.0, abs(w_)), (i, g_, w_)
    # the variances process_samples derives from these sums equal numpy's two-pass np.var
    cnt = s[0]
    assert np.isclose(s[2] / cnt - (s[1] / cnt) ** 2, r64[m].var(), rtol=1e-9, atol=1e-9)
    assert np.isclose(s[8] / cnt - (s[7] / cnt) ** 2, a64[m].var(), rtol=1e-9, atol=1e-9)
    # progress == NULL leaves its columns at the identities
    _lib.check(_lib.lib.rl_sample_stats(B, *[_lib.ptr(a) for a in args], 280.0, 450.0, None, 0, _lib.ptr(ws),
                                        ws.numel(), _lib.ptr(out), _lib.stream_ptr()))
    s2 = out.cpu().numpy()
    assert s2[13] == 0.0 and s2[14] == 0.0 and s2[18] == -np.inf and s2[19] == np.inf
    mean, denom = a64[m].mean(), a64[m].std() + 1e-8
    o = torch.empty(B, dtype=torch.float32, device=dev)
    _lib.check(_lib.lib.rl_adv_finish(B, _lib.ptr(args[2]), _lib.ptr(args[5]), mean, denom, 0.25, _lib.ptr(o),
                                      _lib.stream_ptr()))
    want_o = np.where(m, (a64 - mean) / denom + 0.25, 0.0)
    assert np.abs(o.cpu().numpy() - want_o).max() <= 1e-6 * max(1.0, np.abs(want_o).max())


def test_forward_progress_from_stats_kernel_matches_path_index(quiet_logger):
    """AverageForwardProgress & co. (swimmer_env.py:48-62) as computed inside rl_sample_stats equal the
    per-path gather over the path index (torch nonzero / indexing) on a real Swimmer rollout."""
    from rllab_amd.algos.trpo import TRPO
    from rllab_amd.baselines.zero_baseline import ZeroBaseline
    from rllab_amd.envs.mujoco.swimmer_env import SwimmerEnv
    from rllab_amd.envs.normalized_env import normalize
    from rllab_amd.misc import ext, logger
    from rllab_amd.policies.gaussian_mlp_policy import GaussianMLPPolicy
    ext.set_seed(7)
    env = normalize(SwimmerEnv())
    policy = GaussianMLPPolicy(env_spec=env.spec, hidden_sizes=(32, 32))
    algo = TRPO(env=env, policy=policy, baseline=ZeroBaseline(env_spec=env.spec), batch_size=100 * 60, max_path_length=37,
                n_itr=1, sampler_args=dict(n_envs=100))
    algo.start_worker()
    paths = algo.sampler.obtain_samples(0)
    # horizon 37 inside 37-step rollouts: make the rollout longer than a path so several paths per column exist
    traj = algo.sampler.vec_env.rollout(policy, 100, reset_at_start=True)
    from rllab_amd.sampler.base import process_dense
    from rllab_amd.sampler.trajectories import PathList
    sd = process_dense(algo, 0, traj)
    logger.dump_tabular()
    got = traj.progress_stats
    e, t0, t1 = PathList(traj).index()
    comx = traj.obs[traj.obs_dim - 3].double()
    progs = (comx[t1, e] - comx[t0, e]).cpu().numpy()
    assert len(progs) == 200 and got is not None       # 2 complete 37-step paths per column, the tail dropped
    want = (progs.mean(), progs.max(), progs.min(), progs.std())
    assert np.allclose(got, want, rtol=1e-9, atol=1e-12)
    env.log_diagnostics(PathList(traj))
    tab = logger.get_tabular()
    assert np.isclose(float(tab["AverageForwardProgress"]), want[0]) and np.isclose(float(tab["StdForwardProgress"]), want[3])
    logger.dump_tabular()


def test_prefetched_rollout_changes_nothing(quiet_logger, monkeypatch):
    """BatchPolopt.train enqueues iteration k + 1's rollout right after update k (sampler.prefetch), before the host
    writes log and snapshot: same launches in the same order -- parameters and every logged number are identical to
    the run that launches the rollout at the top of the next iteration."""
    from rllab_amd.algos.trpo import TRPO
    from rllab_amd.baselines.linear_feature_baseline import LinearFeatureBaseline
    from rllab_amd.envs.mujoco.swimmer_env import SwimmerEnv
    from rllab_amd.envs.normalized_env import normalize
    from rllab_amd.misc import ext, logger
    from rllab_amd.policies.gaussian_mlp_policy import GaussianMLPPolicy
    from rllab_amd.sampler.vectorized_sampler import VectorizedSampler

    def run(prefetch):
        ext.set_seed(4)
        env = normalize(SwimmerEnv())
        pol = GaussianMLPPolicy(env_spec=env.spec, hidden_sizes=(32, 32))
        algo = TRPO(env=env, policy=pol, baseline=LinearFeatureBaseline(env_spec=env.spec), batch_size=64 * 50,
                    max_path_length=50, n_itr=4, discount=0.99, step_size=0.01, sampler_args=dict(n_envs=64, seed=9))
        calls = []
        orig = VectorizedSampler.prefetch
        if prefetch:
            monkeypatch.setattr(VectorizedSampler, "prefetch", lambda self, itr: (calls.append(itr), orig(self, itr))[1])
        else:
            monkeypatch.setattr(VectorizedSampler, "prefetch", lambda self, itr: calls.append(("skipped", itr)))
        rows = []
        orig_dump = logger.dump_tabular
        monkeypatch.setattr(logger, "dump_tabular", lambda *a, **k: (rows.append(logger.get_tabular()), orig_dump(*a, **k))[1])
        algo.train()
        monkeypatch.setattr(logger, "dump_tabular", orig_dump)
        monkeypatch.setattr(VectorizedSampler, "prefetch", orig)
        return pol.get_param_values(), rows, calls
    th_a, rows_a, calls_a = run(True)
    th_b, rows_b, calls_b = run(False)
    # (twice per iteration since round 5: from the optimizer's hook once the update is enqueued -- the line search is
    # decided on the device -- and from train_iteration afterwards, where it finds the batch already queued)
    assert calls_a == [1, 1, 2, 2, 3, 3] and calls_b == [("skipped", i) for i in (1, 1, 2, 2, 3, 3)]
    assert np.array_equal(th_a, th_b)
    for ra, rb in zip(rows_a, rows_b):
        for k in ("AverageReturn", "LossBefore", "LossAfter", "MeanKL", "Entropy"):
            assert ra[k] == rb[k], (k, ra[k], rb[k])
