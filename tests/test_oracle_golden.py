"""The numpy oracle (oracle/np_reference.py) pinned against fixtures produced by the REAL
reference functions (oracle/make_golden.py, run in the build container where
/root/reference exists).  CPU only."""
import numpy as np
import pytest

from oracle import np_reference as R
from tests.golden_util import STAT_KEYS, load, unpack_paths


def test_discount_cumsum_and_stats():
    g = load("special_util")
    assert np.allclose(R.discount_cumsum(g["x"], 0.99), g["dc_099"], rtol=1e-12, atol=1e-12)
    assert np.allclose(R.discount_cumsum(g["x"], 0.5), g["dc_05"], rtol=1e-12, atol=1e-12)
    assert np.allclose(R.discount_cumsum(g["X2"], 0.97), g["dc2_097"], rtol=1e-12, atol=1e-12)
    assert np.allclose(g["arange5"], [7.3314, 8.146, 7.94, 6.6, 4.0])  # value quoted in SURVEY.md 8c
    assert np.isclose(R.explained_variance_1d(g["yp"], g["y"]), g["ev"], rtol=1e-12)
    assert R.explained_variance_1d(g["yp"], np.ones(500)) == g["ev_const"] == 0
    assert R.explained_variance_1d(np.zeros(500), np.ones(500)) == g["ev_const_both"] == 1
    assert np.allclose(R.center_advantages(g["adv"]), g["centered"], rtol=1e-12)
    assert np.allclose(R.shift_advantages_to_positive(g["adv"]), g["shifted"], rtol=1e-12)


def test_diagonal_gaussian():
    g = load("diagonal_gaussian")
    assert np.allclose(R.gaussian_kl(g["om"], g["ols"], g["nm"], g["nls"]), g["kl"], rtol=1e-12)
    assert np.allclose(R.gaussian_log_likelihood(g["xs"], g["nm"], g["nls"]), g["logli"], rtol=1e-12)
    assert np.allclose(R.gaussian_entropy(g["nls"]), g["entropy"], rtol=1e-12)
    # docs/user/experiments.rst:81 -- entropy of a 1-D unit Gaussian
    assert np.isclose(R.gaussian_entropy(np.zeros((1, 1)))[0], 1.41894, atol=1e-5)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_process_samples_two_iterations(tag):
    g = load("process_samples_" + tag)
    paths = unpack_paths(g)
    base = R.LinearFeatureBaseline()
    kw = dict(discount=float(g["discount"]), gae_lambda=float(g["gae_lambda"]),
              center_adv=bool(g["center_adv"]), positive_adv=bool(g["positive_adv"]))
    out1, st1 = R.process_samples([dict(p) for p in paths], base, **kw)
    assert np.allclose(out1["advantages"], g["adv1"], rtol=1e-9, atol=1e-9)
    assert np.allclose(out1["returns"], g["ret1"], rtol=1e-12)
    assert np.allclose(base._coeffs, g["coeffs1"], rtol=1e-6, atol=1e-8)
    assert np.allclose([st1[k] for k in STAT_KEYS], g["stats1"], rtol=1e-9)
    out2, st2 = R.process_samples([dict(p) for p in paths], base, **kw)
    assert np.allclose(out2["advantages"], g["adv2"], rtol=1e-6, atol=1e-7)
    assert np.allclose(out2["returns"], g["ret2"], rtol=1e-12)
    assert np.allclose([st2[k] for k in STAT_KEYS], g["stats2"], rtol=1e-6)


def test_truncate_paths_matches_reference_test():
    """tests/test_sampler.py of the reference: (100, 50) truncated at 130 -> (100, 30)."""
    g = load("truncate_paths")
    mk = lambda n: dict(observations=np.zeros((n, 1)), actions=np.zeros((n, 1)), rewards=np.zeros(n),
                        env_infos=dict(), agent_infos=dict(lala=np.zeros(n)))
    paths = [mk(100), mk(50)]
    t130 = R.truncate_paths(paths, 130)
    assert [len(p["rewards"]) for p in t130] == list(g["lens130"]) == [100, 30]
    assert [len(p["agent_infos"]["lala"]) for p in t130] == list(g["info130"])
    assert [len(p["rewards"]) for p in R.truncate_paths(paths, 90)] == list(g["lens90"]) == [90]
    assert len(paths[1]["rewards"]) == 50  # input not mutated


def test_krylov_cg():
    g = load("krylov_cg")
    A, b = g["A"], g["b"]
    assert np.allclose(R.cg(lambda v: A.dot(v), b, 10), g["x10"], rtol=1e-10)
    assert np.allclose(R.cg(lambda v: A.dot(v), b, 3), g["x3"], rtol=1e-10)
    assert np.allclose(R.cg(lambda v: 4.0 * v, b, 10), g["x_early"], rtol=1e-12)
    # krylov.test_cg of the reference: CG solves a small SPD system
    rng = np.random.RandomState(0)
    M = rng.randn(5, 5)
    M = M.T.dot(M)
    rhs = rng.randn(5)
    assert np.allclose(M.dot(R.cg(lambda v: M.dot(v), rhs, 5)), rhs)


def test_normalized_action_map():
    g = load("normalized_env")
    got = np.array([R.normalized_action(a, g["lb"], g["ub"]) for a in g["acts"]])
    assert np.allclose(got, g["scaled"], rtol=1e-15)
    assert np.allclose(g["rews"], 0.25)


@pytest.mark.parametrize("tag", ["easy", "backtrack", "reject"])
def test_cg_optimizer_control_flow(tag):
    g = load("cg_optimizer")
    Hm, Cm, gvec = g["Hm"], g["Cm"], g["gvec"]
    theta0, quartic, delta = g["theta0_" + tag], float(g["quartic_" + tag]), float(g["delta_" + tag])

    def loss(th):
        d = th - theta0
        if quartic < 0:
            return float(gvec.dot(d)) ** 2 + 1.0
        return float(gvec.dot(d) + 0.5 * d.dot(Hm).dot(d) + quartic * np.sum(d ** 4))
    cons = lambda th: float(0.5 * (th - theta0).dot(Cm).dot(th - theta0))
    got, info = R.cg_optimize(theta0.copy(), loss, lambda th: gvec, cons, lambda th, x: Cm.dot(x), delta)
    assert np.allclose(got, g["theta1_" + tag], rtol=1e-10, atol=1e-12)
    assert info["rejected"] == (tag == "reject")
    if tag == "backtrack":
        assert info["backtrack_iters"] > 0


def test_flat_param_layout():
    g = load("flat_params")
    pol = R.NumpyGaussianMLP(4, 1, (32, 32))
    pol.set_param_values(g["flat"])
    layers, log_std = pol.unflatten()
    for (W, b), kW, kb in zip(layers, ["W0", "W1", "W2"], ["b0", "b1", "b2"]):
        assert np.array_equal(W, g[kW]) and np.array_equal(b, g[kb])
    assert np.array_equal(log_std, g["log_std"])


# -- lasagne.updates.adam (third party, absent): hand-derived known answers ------------------------------------------------
def test_adam_first_steps_by_hand():
    """With m = v = 0:  m_1 = (1 - b1) g,  v_1 = (1 - b2) g^2,  a_1 = lr sqrt(1 - b2) / (1 - b1), so the first step is
    lr g / (|g| + eps / sqrt(1 - b2)); under a CONSTANT gradient m_t = (1 - b1^t) g and v_t = (1 - b2^t) g^2, so
    every step is lr g / (|g| + eps / sqrt(1 - b2^t)) -- written out here without the recurrences."""
    lr, b1, b2, eps = 1e-2, 0.9, 0.999, 1e-8
    g = np.array([3.0, -0.5, 1e-3, -2e-9, 0.0])
    theta = np.array([1.0, 2.0, -3.0, 0.5, 7.0])
    st = R.Adam([theta], learning_rate=lr)
    want = theta.copy()
    for t in range(1, 6):
        theta = R.adam_step(st, theta, g)
        want = want - lr * g / (np.abs(g) + eps / np.sqrt(1.0 - b2 ** t))
        assert np.allclose(theta, want, rtol=1e-13, atol=0), t
        assert np.allclose(st.m[0], (1 - b1 ** t) * g, rtol=1e-13) and np.allclose(st.v[0], (1 - b2 ** t) * g * g, rtol=1e-12)
    assert st.t == 5.0 and theta[4] == 7.0          # a zero gradient never moves its parameter (0 / (0 + eps))


def test_adam_two_steps_with_changing_gradient_by_hand():
    """Second step spelled out:  m_2 = b1 (1-b1) g1 + (1-b1) g2,  v_2 = b2 (1-b2) g1^2 + (1-b2) g2^2,
    a_2 = lr sqrt(1 - b2^2) / (1 - b1^2)."""
    lr, b1, b2, eps = 1e-3, 0.9, 0.999, 1e-8
    g1, g2 = np.array([0.3, -1.2]), np.array([-0.7, 0.4])
    th0 = np.array([0.1, 0.2])
    st = R.Adam([th0])
    th1 = R.adam_step(st, th0, g1)
    th2 = R.adam_step(st, th1, g2)
    m2 = b1 * (1 - b1) * g1 + (1 - b1) * g2
    v2 = b2 * (1 - b2) * g1 ** 2 + (1 - b2) * g2 ** 2
    a2 = lr * np.sqrt(1 - b2 ** 2) / (1 - b1 ** 2)
    assert np.allclose(th2, th1 - a2 * m2 / (np.sqrt(v2) + eps), rtol=1e-14, atol=0)
    # one counter for all arrays (lasagne shares t_prev): two arrays stepped together == the flat vector
    st2 = R.Adam([th0[:1], th0[1:]])
    a, b = st2.step([th0[:1], th0[1:]], [g1[:1], g1[1:]])
    a, b = st2.step([a, b], [g2[:1], g2[1:]])
    assert np.array_equal(np.concatenate([a, b]), th2)


def test_vpg_gradient_against_central_differences():
    """oracle's hand-written back-propagation of vpg.py's surrogate == central differences of its own loss."""
    rng = np.random.RandomState(3)
    pol = R.NumpyGaussianMLP(4, 2, (8, 8))
    theta = 0.3 * rng.randn(pol.n_params)
    theta[-1] = np.log(1e-6) - 1.0                 # one log_std below the floor: its gradient is blocked
    obs, act, adv = rng.randn(50, 4), rng.randn(50, 2), rng.randn(50)
    w = (rng.rand(50) > 0.2).astype(np.float64)
    loss, g = R.vpg_surrogate_and_grad(pol, theta, obs, act, adv, w)
    num = np.zeros_like(theta)
    for i in range(theta.size):
        e = np.zeros_like(theta)
        e[i] = 1e-6
        num[i] = (R.vpg_surrogate_and_grad(pol, theta + e, obs, act, adv, w)[0] -
                  R.vpg_surrogate_and_grad(pol, theta - e, obs, act, adv, w)[0]) / 2e-6
    assert g[-1] == 0.0
    assert np.allclose(g, num, rtol=2e-6, atol=1e-4 * np.abs(g).max())
    # the loss is the reference formula on the oracle's own log-likelihood
    mean, ls = pol.dist_info(obs, theta)
    assert np.isclose(loss, -np.sum(w * adv * R.gaussian_log_likelihood(act, mean, ls)) / w.sum(), rtol=1e-13)
