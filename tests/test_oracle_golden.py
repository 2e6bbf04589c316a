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
