"""Host-side logic of the product on CPU tensors (no kernel launches): API seams,
parameter layout, optimizer control flow against the golden fixtures of the real
reference."""
import os
import pickle

import numpy as np
import pytest
import torch

from tests.golden_util import load


def _spec(do, da):
    from rllab_amd.envs.env_spec import EnvSpec
    from rllab_amd.spaces import Box
    return EnvSpec(Box(-np.ones(do), np.ones(do)), Box(-np.ones(da), np.ones(da)))


def test_rllab_alias_is_same_module():
    import rllab.algos.trpo as a
    import rllab_amd.algos.trpo as b
    from rllab.envs.normalized_env import normalize
    from rllab_amd.envs.normalized_env import NormalizedEnv
    assert a is b and normalize is NormalizedEnv
    # every import path used by examples/trpo_cartpole.py and examples/trpo_swimmer.py
    from rllab.algos.trpo import TRPO  # noqa: F401
    from rllab.baselines.linear_feature_baseline import LinearFeatureBaseline  # noqa: F401
    from rllab.envs.box2d.cartpole_env import CartpoleEnv  # noqa: F401
    from rllab.envs.mujoco.swimmer_env import SwimmerEnv  # noqa: F401
    from rllab.policies.gaussian_mlp_policy import GaussianMLPPolicy  # noqa: F401


def test_policy_flat_layout_matches_reference_flatten_tensors():
    from rllab_amd.policies.gaussian_mlp_policy import GaussianMLPPolicy
    g = load("flat_params")
    pol = GaussianMLPPolicy(_spec(4, 1), hidden_sizes=(32, 32))
    assert [p.shape for p in pol.get_params(trainable=True)] == [(4, 32), (32,), (32, 32), (32,), (32, 1), (1,), (1,)]
    pol.set_param_values(g["flat"])
    for p, key in zip(pol.get_params(), ["W0", "b0", "W1", "b1", "W2", "b2", "log_std"]):
        assert np.allclose(p.get_value(), g[key], atol=1e-7)
    assert pol.get_param_values().shape == (1250,)
    vals = pol.flat_to_params(g["flat"], trainable=True)
    assert np.array_equal(vals[2], g["W1"])
    # forward against the numpy oracle policy with the same flat vector
    from oracle import np_reference as R
    ref = R.NumpyGaussianMLP(4, 1, (32, 32))
    ref.set_param_values(g["flat"])
    obs = np.random.RandomState(0).randn(17, 4)
    mean, log_std = ref.dist_info(obs)
    d = pol.dist_info(obs)
    assert np.allclose(d["mean"], mean, atol=2e-6) and np.allclose(d["log_std"], log_std, atol=1e-7)


def test_policy_init_and_pickle_roundtrip():
    from rllab_amd.policies.gaussian_mlp_policy import GaussianMLPPolicy
    np.random.seed(3)
    pol = GaussianMLPPolicy(_spec(13, 2), hidden_sizes=(32, 32), init_std=0.5)
    th = pol.get_param_values()
    assert th.shape == (1572,)
    W0 = th[:13 * 32]
    assert np.abs(W0).max() <= np.sqrt(6.0 / (13 + 32)) + 1e-6          # Glorot-uniform bound
    assert np.allclose(th[13 * 32:13 * 32 + 32], 0) and np.allclose(th[-2:], np.log(0.5))
    clone = pickle.loads(pickle.dumps(pol))
    assert np.array_equal(clone.get_param_values(), th)
    # learn_std=False removes log_std from the trainable set (Lasagne tag semantics)
    fixed = GaussianMLPPolicy(_spec(13, 2), learn_std=False)
    assert fixed.get_param_values(trainable=True).shape == (1570,)
    fixed.set_param_values(np.ones(1570), trainable=True)
    assert np.allclose(fixed.get_param_values()[:1570], 1) and np.allclose(fixed.get_param_values()[1570:], 0)
    # entropy of the initial policy: 0.5*ln(2 pi e) per dim (docs/user/experiments.rst:81)
    unit = GaussianMLPPolicy(_spec(4, 1))
    ent = unit.distribution.entropy(dict(log_std=unit.dist_info(np.zeros((1, 4)))["log_std"]))
    assert np.isclose(ent[0], 1.41894, atol=1e-5)


def test_serializable_clone_and_spaces():
    from rllab_amd.core.serializable import Serializable
    from rllab_amd.envs.normalized_env import NormalizedEnv
    from rllab_amd.spaces import Box

    class E(object):
        observation_space = Box(-np.ones(3), np.ones(3))
        action_space = Box(np.array([-2.0]), np.array([4.0]))
    ne = NormalizedEnv(E(), scale_reward=0.5)
    c = Serializable.clone(ne, scale_reward=2.0)
    assert c._scale_reward == 2.0 and ne._scale_reward == 0.5
    b = Box(-1.0, 1.0, (3, 4))
    assert b.flat_dim == 12 and b.flatten(np.zeros((3, 4))).shape == (12,)
    assert b.flatten_n(np.zeros((5, 3, 4))).shape == (5, 12) and b.unflatten_n(np.zeros((5, 12))).shape == (5, 3, 4)
    assert b.contains(np.zeros((3, 4))) and not b.contains(2 * np.ones((3, 4)))
    assert ne.action_space == Box(-np.ones(1), np.ones(1))


def test_normalized_env_numpy_path_vs_reference():
    from rllab_amd.envs.base import Env, Step
    from rllab_amd.envs.normalized_env import normalize
    from rllab_amd.spaces import Box
    g = load("normalized_env")

    class E(Env):
        action_space = Box(g["lb"], g["ub"])
        observation_space = Box(-np.ones(3), np.ones(3))

        def reset(self):
            return np.zeros(3)

        def step(self, a):
            self.last = np.array(a)
            return Step(np.zeros(3), 2.5, False)
    e = E()
    ne = normalize(e, scale_reward=0.1)
    for a, want, r in zip(g["acts"], g["scaled"], g["rews"]):
        _, rew, _, _ = ne.step(a)
        assert np.allclose(e.last, want) and np.isclose(rew, r)
    assert not ne.vectorized


def test_diagonal_gaussian_and_krylov_torch_vs_reference():
    from rllab_amd.distributions.diagonal_gaussian import DiagonalGaussian
    from rllab_amd.misc import krylov
    g = load("diagonal_gaussian")
    d = DiagonalGaussian(3)
    assert np.allclose(d.kl(dict(mean=g["om"], log_std=g["ols"]), dict(mean=g["nm"], log_std=g["nls"])), g["kl"])
    assert np.allclose(d.log_likelihood(g["xs"], dict(mean=g["nm"], log_std=g["nls"])), g["logli"])
    assert np.allclose(d.entropy(dict(mean=g["nm"], log_std=g["nls"])), g["entropy"])
    # planes layout (axis=0) gives the same numbers
    t = lambda x: torch.as_tensor(x.T.copy())
    kl0 = d.kl_sym(dict(mean=t(g["om"]), log_std=t(g["ols"])), dict(mean=t(g["nm"]), log_std=t(g["nls"])), axis=0)
    assert np.allclose(kl0.numpy(), g["kl"])
    k = load("krylov_cg")
    A = torch.as_tensor(k["A"])
    assert np.allclose(krylov.cg(lambda v: A @ v, k["b"], cg_iters=10).numpy(), k["x10"], rtol=1e-9)
    assert np.allclose(krylov.cg(lambda v: A @ v, k["b"], cg_iters=3).numpy(), k["x3"], rtol=1e-9)
    # early-exit branch (rdotr < 1e-10 after one iteration) realised by the `active` flag
    assert np.allclose(krylov.cg(lambda v: 4.0 * v, k["b"], cg_iters=10).numpy(), k["x_early"], rtol=1e-12)


class _Quad(object):
    """Minimal Parameterized stand-in: a flat float64 vector on the CPU."""

    def __init__(self, theta):
        self.flat_params = torch.tensor(theta, dtype=torch.float64)

    def _flat_index(self, **tags):
        return None

    def get_param_values(self, **tags):
        return self.flat_params.numpy().copy()

    def set_param_values(self, v, **tags):
        self.flat_params.copy_(torch.as_tensor(np.asarray(v), dtype=torch.float64))


@pytest.mark.parametrize("tag", ["easy", "backtrack", "reject"])
def test_cg_optimizer_control_flow_vs_reference(tag, quiet_logger):
    """ConjugateGradientOptimizer.optimize (device CG, torch closures) reproduces the parameter
    vector the REAL reference optimizer produced on the same toy problem."""
    from rllab_amd.optimizers.conjugate_gradient_optimizer import ConjugateGradientOptimizer
    g = load("cg_optimizer")
    Hm, Cm, gvec = (torch.as_tensor(g[k]) for k in ("Hm", "Cm", "gvec"))
    theta0 = torch.as_tensor(g["theta0_" + tag])
    quartic, delta = float(g["quartic_" + tag]), float(g["delta_" + tag])

    def loss(flat, dummy):
        d = flat - theta0
        if quartic < 0:
            # value never improves, gradient is the reference's injected gvec
            return (gvec.dot(d)) ** 2 + 1.0 + (gvec.dot(d) - gvec.dot(d).detach())
        return gvec.dot(d) + 0.5 * d @ Hm @ d + quartic * (d ** 4).sum()

    def cons(flat, dummy):
        d = flat - theta0
        return 0.5 * d @ Cm @ d
    target = _Quad(g["theta0_" + tag])
    opt = ConjugateGradientOptimizer()
    opt.update_opt(loss=loss, target=target, leq_constraint=(cons, delta), inputs=None)
    opt.optimize((torch.zeros(4, 1),))
    assert np.allclose(target.get_param_values(), g["theta1_" + tag], rtol=1e-8, atol=1e-10)


def test_first_order_optimizer_lasagne_adam_step(quiet_logger):
    from rllab_amd.optimizers.first_order_optimizer import FirstOrderOptimizer
    theta0 = np.array([1.0, -2.0, 0.5])
    target = _Quad(theta0)
    c = torch.tensor([3.0, -1.0, 2.0], dtype=torch.float64)
    opt = FirstOrderOptimizer(batch_size=None, max_epochs=1, learning_rate=1e-3)
    opt.update_opt(lambda flat, x: (c * flat).sum() + 0.0 * x.sum(), target=target)
    x = torch.zeros(1, 5, dtype=torch.float64)
    opt.optimize((x,))
    # first Adam step: m = 0.1 g, v = 0.001 g^2, a_1 = lr*sqrt(1-b2)/(1-b1) => step = lr * g/(|g| + eps')
    g = c.numpy()
    a1 = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
    want = theta0 - a1 * (0.1 * g) / (np.sqrt(0.001 * g * g) + 1e-8)
    assert np.allclose(target.get_param_values(), want, rtol=1e-12)
    opt.optimize((x,))  # moments persist across calls (vpg: created once in update_opt)
    assert opt._updater.t == 2


def test_trajectories_segmentation_and_pathlist_cpu():
    from rllab_amd.sampler.trajectories import PathList, Trajectories
    T, N = 6, 3
    done = torch.zeros(T, N, dtype=torch.uint8)
    done[1, 0] = 1          # env 0: paths [0,1], [2..5] (incomplete)
    done[5, 1] = 1          # env 1: one complete path
    done[0, 2] = 1
    done[3, 2] = 1          # env 2: [0], [1..3], [4,5] (incomplete)
    obs = torch.arange(2 * T * N, dtype=torch.float32).reshape(2, T, N)
    act = torch.zeros(1, T, N)
    tr = Trajectories(obs, act, act.clone(), torch.zeros(1), torch.ones(T, N), done, 6)
    env, t0, t1, complete = tr.segments()
    assert list(zip(env.tolist(), t0.tolist(), t1.tolist(), complete.tolist())) == [
        (0, 0, 1, True), (0, 2, 5, False), (1, 0, 5, True), (2, 0, 0, True), (2, 1, 3, True), (2, 4, 5, False)]
    assert tr.time_in_path()[:, 2].tolist() == [0, 0, 1, 2, 0, 1]
    v = tr.valid_mask(whole_paths=True)
    assert v[:, 0].tolist() == [True, True, False, False, False, False]
    assert v[:, 1].all() and v[:, 2].tolist() == [True, True, True, True, False, False]
    assert tr.valid_mask(whole_paths=False).all()
    tr.valid = v
    paths = PathList(tr)
    assert len(paths) == 4
    p = paths[3]
    assert p["observations"].shape == (3, 2) and p["rewards"].shape == (3,)
    assert np.array_equal(p["observations"][:, 0], obs[0, 1:4, 2].numpy())
    assert [len(q["rewards"]) for q in paths] == [2, 6, 1, 3]


def test_rollout_and_truncate_paths_generic_env():
    """rollout() / truncate_paths keep the reference semantics for arbitrary Python envs
    (tests/test_sampler.py of the reference)."""
    from rllab_amd.envs.base import Env, Step
    from rllab_amd.sampler.utils import rollout, truncate_paths
    from rllab_amd.spaces import Box
    g = load("truncate_paths")

    class Count(Env):
        observation_space = Box(-np.ones(1), np.ones(1))
        action_space = Box(-np.ones(1), np.ones(1))

        def reset(self):
            self.t = 0
            return np.zeros(1)

        def step(self, a):
            self.t += 1
            return Step(np.array([self.t / 10.0]), 1.0, self.t >= 7, k=self.t)

    class Pol(object):
        def reset(self):
            pass

        def get_action(self, o):
            return np.array([0.5]), dict(mean=np.zeros(1), log_std=np.zeros(1))
    p = rollout(Count(), Pol(), max_path_length=100)
    assert len(p["rewards"]) == 7 and p["observations"].shape == (7, 1) and p["env_infos"]["k"].tolist() == list(range(1, 8))
    assert len(rollout(Count(), Pol(), max_path_length=3)["rewards"]) == 3
    mk = lambda n: dict(observations=np.zeros((n, 1)), actions=np.zeros((n, 1)), rewards=np.zeros(n),
                        env_infos=dict(), agent_infos=dict(lala=np.zeros(n)))
    paths = [mk(100), mk(50)]
    assert [len(q["rewards"]) for q in truncate_paths(paths, 130)] == list(g["lens130"])
    assert [len(q["rewards"]) for q in truncate_paths(paths, 90)] == list(g["lens90"])
    assert len(paths[-1]["rewards"]) == 50


def test_logger_tabular_and_csv(tmp_path):
    from rllab_amd.misc import logger
    logger.set_quiet(True)
    f = str(tmp_path / "progress.csv")
    logger.add_tabular_output(f)
    logger.record_tabular("Iteration", 0)
    logger.record_tabular("AverageReturn", 1.5)
    with logger.prefix("itr #0 | "):
        logger.log("hello")
    logger.dump_tabular()
    logger.record_tabular("Iteration", 1)
    logger.record_tabular("AverageReturn", 2.5)
    logger.dump_tabular()
    logger.remove_tabular_output(f)
    logger.set_quiet(False)
    assert open(f).read().splitlines() == ["Iteration,AverageReturn", "0,1.5", "1,2.5"]


def test_unsupported_options_fail_loudly():
    from rllab_amd.algos.npo import NPO
    from rllab_amd.envs.box2d.cartpole_env import CartpoleEnv
    from rllab_amd.envs.mujoco.swimmer_env import SwimmerEnv
    from rllab_amd.policies.gaussian_mlp_policy import GaussianMLPPolicy
    # options that name a different world / model have no kernel; the ones the kernels honour are accepted
    with pytest.raises(NotImplementedError):
        CartpoleEnv(template_args=dict(noise=True))
    with pytest.raises(NotImplementedError):
        SwimmerEnv(file_path="/tmp/other.xml")
    with pytest.raises(ValueError):
        CartpoleEnv(obs_noise=-0.1)
    env = CartpoleEnv(obs_noise=0.1, action_noise=0.05, frame_skip=2, position_only=True)
    assert env.observation_space.flat_dim == 2 and env.frame_skip == 2 and env.obs_noise == 0.1
    sw = SwimmerEnv(ctrl_cost_coeff=0.5, action_noise=0.1)
    assert sw.ctrl_cost_coeff == 0.5 and sw._cfg["ctrl_cost_coeff"] == 0.5 and sw._cfg["action_noise"] == 0.1
    import pickle
    sw2 = pickle.loads(pickle.dumps(sw))                  # options survive the ctor-args pickle
    assert sw2.ctrl_cost_coeff == 0.5 and sw2.action_noise == 0.1
    cp = pickle.loads(pickle.dumps(CartpoleEnv(reset_pole_follows_cart=True, obs_noise=0.2)))
    assert cp.reset_pole_follows_cart and cp._cfg["flags"] == 1 and cp.obs_noise == 0.2
    from rllab_amd.envs.box2d.double_pendulum_env import DoublePendulumEnv
    np.random.seed(11)
    want = (np.random.rand() - 0.5) + 1
    np.random.seed(11)
    dp = DoublePendulumEnv(template_args=dict(noise=True))          # one random link length per env object (:17-21)
    assert dp.link_len == want and 0.5 <= dp.link_len < 1.5 and abs(dp._cfg["link_len"] - want) < 1e-12
    assert pickle.loads(pickle.dumps(DoublePendulumEnv(template_args=dict(link_len=1.25)))).link_len == 1.25
    assert DoublePendulumEnv().link_len == 1 and DoublePendulumEnv().frame_skip == 2
    with pytest.raises(NotImplementedError):
        DoublePendulumEnv(template_args=dict(gravity=3))
    ad = GaussianMLPPolicy(_spec(4, 1), adaptive_std=True)     # the reference's tests/regression_tests/test_issue_3.py
    assert ad.state_dependent_std and ad.kernel_layout() is None
    # NPO's default optimizer is the reference's PenaltyLbfgsOptimizer (npo.py:27-30)
    from rllab_amd.optimizers.penalty_lbfgs_optimizer import PenaltyLbfgsOptimizer
    assert isinstance(NPO(env=None, policy=None, baseline=None).optimizer, PenaltyLbfgsOptimizer)
    if not torch.cuda.is_available():
        env = CartpoleEnv()
        with pytest.raises(RuntimeError, match="no HIP device"):
            env.reset()  # no CPU fallback for env kernels


def _lbfgs_problem(g, theta0):
    A, C, b = (torch.as_tensor(g[k]) for k in ("A", "C", "b"))
    th0 = torch.as_tensor(theta0)

    def loss(flat, dummy):
        d = flat - th0
        return b.dot(d) + 0.5 * d @ A @ d + 0.1 * (d ** 4).sum()

    def cons(flat, dummy):
        d = flat - th0
        return 0.5 * d @ C @ d
    return loss, cons


def test_lbfgs_optimizer_vs_reference(quiet_logger):
    """LbfgsOptimizer.optimize == the REAL reference optimizer on the same toy problem (same
    scipy fmin_l_bfgs_b trajectory, including the last-evaluation quirk)."""
    from rllab_amd.optimizers.lbfgs_optimizer import LbfgsOptimizer
    g = load("lbfgs_optimizers")
    loss, _ = _lbfgs_problem(g, g["lbfgs_theta0"])
    target = _Quad(g["lbfgs_theta0"])
    opt = LbfgsOptimizer(max_opt_itr=20)
    opt.update_opt(loss=loss, target=target, inputs=None)
    opt.optimize((torch.zeros(1),))
    assert np.allclose(target.get_param_values(), g["lbfgs_theta1"], rtol=1e-7, atol=1e-9)


@pytest.mark.parametrize("tag,kw", [("tight", {}), ("loose", {}), ("fixed", dict(adapt_penalty=False)),
                                    ("few", dict(max_penalty_itr=3))])
def test_penalty_lbfgs_optimizer_control_flow_vs_reference(tag, kw, quiet_logger):
    """The adaptive-penalty search (grow until the constraint holds / shrink until it is crossed /
    give up and restore) lands on the parameters and the penalty of the REAL reference."""
    from rllab_amd.optimizers.penalty_lbfgs_optimizer import PenaltyLbfgsOptimizer
    g = load("lbfgs_optimizers")
    loss, cons = _lbfgs_problem(g, g["pen_theta0_" + tag])
    target = _Quad(g["pen_theta0_" + tag])
    opt = PenaltyLbfgsOptimizer(**kw)
    opt.update_opt(loss=loss, target=target, leq_constraint=(cons, float(g["pen_eps_" + tag])), inputs=None)
    x = (torch.zeros(1),)
    opt.optimize(x)
    assert np.allclose(target.get_param_values(), g["pen_theta1_" + tag], rtol=1e-6, atol=1e-8)
    assert opt._penalty == float(g["pen_penalty_" + tag])
    assert np.isclose(opt.constraint_val(x), float(g["pen_cons_" + tag]), rtol=1e-5, atol=1e-12)


def test_gaussian_mlp_regressor_and_baseline_fit(quiet_logger):
    """GaussianMLPRegressor (rectify MLP + free log_std, whitened inputs / outputs, trust-region
    PenaltyLbfgs fit) learns a smooth function; GaussianMLPBaseline maps paths to it.  Repeated fits
    move the prediction towards the targets while each fit stays inside its KL bound."""
    from rllab_amd.baselines.gaussian_mlp_baseline import GaussianMLPBaseline
    from rllab_amd.envs.env_spec import EnvSpec
    from rllab_amd.misc import logger
    from rllab_amd.regressors.gaussian_mlp_regressor import GaussianMLPRegressor
    from rllab_amd.spaces import Box
    np.random.seed(0)
    rng = np.random.RandomState(0)
    xs = rng.randn(2000, 3)
    ys = (np.sin(xs[:, :1]) + 0.5 * xs[:, 1:2] * xs[:, 2:3]) * 10 + 50
    reg = GaussianMLPRegressor(input_shape=(3,), output_dim=1, name="vf", step_size=0.05)
    err = []
    for _ in range(12):
        reg.fit(xs, ys)
        tab = logger.get_tabular()
        assert float(tab["vf_LossAfter"]) <= float(tab["vf_LossBefore"]) + 1e-9
        assert float(tab["vf_MeanKL"]) <= 0.05 + 1e-6
        logger.dump_tabular()
        err.append(float(np.mean((reg.predict(xs) - ys) ** 2)))
    assert err[-1] < 0.25 * err[0] and err[-1] < 0.5 * np.var(ys)
    assert reg.predict(xs[:5]).shape == (5, 1) and reg.predict_log_likelihood(xs[:5], ys[:5]).shape == (5,)
    # flat parameter round trip (Parameterized contract) and the baseline wrapper
    theta = reg.get_param_values()
    assert theta.shape == (3 * 32 + 32 + 32 * 32 + 32 + 32 + 1 + 1,)
    spec = EnvSpec(Box(-np.ones(3), np.ones(3)), Box(-np.ones(1), np.ones(1)))
    b = GaussianMLPBaseline(spec, regressor_args=dict(use_trust_region=False))
    paths = [dict(observations=xs[i:i + 100], returns=ys[i:i + 100, 0], rewards=np.zeros(100)) for i in range(0, 2000, 100)]
    b.fit(paths)
    logger.dump_tabular()
    p = b.predict(paths[3])
    assert p.shape == (100,) and np.mean((p - paths[3]["returns"]) ** 2) < np.var(ys)
    b.set_param_values(b.get_param_values() * 0.0)
    assert np.all(b.get_param_values() == 0.0)


def test_linear_feature_baseline_dense_fit_is_deferred_but_identical():
    """fit_dense only starts the read of the normal equations; the solve happens at the first access --
    get_param_values, predict, pickling -- and gives the per-path fit's coefficients."""
    from rllab_amd.baselines.linear_feature_baseline import LinearFeatureBaseline
    from rllab_amd.sampler.trajectories import Trajectories
    rng = np.random.RandomState(0)
    T, N, do = 12, 5, 3
    obs = torch.as_tensor(rng.randn(do, T, N).astype(np.float32))
    dones = torch.zeros((T, N), dtype=torch.uint8)
    dones[T - 1] = 1
    traj = Trajectories(obs, torch.zeros((1, T, N)), torch.zeros((1, T, N)), torch.zeros(1),
                        torch.zeros((T, N)), dones, T)
    traj.valid = torch.ones((T, N), dtype=torch.bool)
    traj.returns = torch.as_tensor(rng.randn(T, N).astype(np.float32))
    b = LinearFeatureBaseline(None)
    b.fit_dense(traj)
    assert b._pending is not None and b._coeffs_value is None          # nothing solved yet
    clone = pickle.loads(pickle.dumps(b))                               # pickling resolves the pending fit
    assert b._pending is None and clone._pending is None
    paths = [dict(observations=obs[:, :, i].t().numpy().astype(np.float64), rewards=np.zeros(T),
                  returns=traj.returns[:, i].numpy().astype(np.float64)) for i in range(N)]
    ref = LinearFeatureBaseline(None)
    ref.fit(paths)
    assert np.allclose(b.get_param_values(), ref.get_param_values(), rtol=1e-6, atol=1e-8)
    assert np.array_equal(clone.get_param_values(), b.get_param_values())
    assert np.allclose(clone.predict(paths[0]), ref.predict(paths[0]), atol=1e-6)
    # a snapshot written before the fit became asynchronous still loads
    legacy = LinearFeatureBaseline(None)
    legacy.__setstate__({"_coeffs": np.arange(3.0), "_reg_coeff": 1e-5})
    assert np.array_equal(legacy.get_param_values(), np.arange(3.0))


def test_batch_dataset_and_flatten_tensor_variables():
    """BatchDataset (minibatch_dataset.py:4-38): every sample once per pass, reshuffled between passes, extras
    appended; planes with the sample axis last are sliced on that axis, per-batch items handed through."""
    from rllab_amd.misc import ext
    from rllab_amd.optimizers.minibatch_dataset import BatchDataset
    x, y = np.arange(10).reshape(10, 1), np.arange(10) * 2
    ds = BatchDataset([x, y], batch_size=4, extra_inputs=["e"])
    assert ds.number_batches == 3
    np.random.seed(0)
    first = [b for b in ds.iterate()]
    assert [len(b[1]) for b in first] == [4, 4, 2] and all(b[2] == "e" for b in first)
    assert sorted(np.concatenate([b[1] for b in first]).tolist()) == list(range(0, 20, 2))
    assert all(np.array_equal(b[0][:, 0] * 2, b[1]) for b in first)
    second = [b for b in ds.iterate()]
    assert not np.array_equal(np.concatenate([b[1] for b in first]), np.concatenate([b[1] for b in second]))
    whole = list(BatchDataset([x, y], batch_size=None).iterate())
    assert len(whole) == 1 and whole[0][0] is x
    planes, row, inv = torch.arange(20.).reshape(2, 10), torch.ones(2, 1), torch.tensor(0.1)
    b = next(BatchDataset([planes, torch.arange(10.), row, inv], batch_size=3, sample_axis=-1).iterate())
    assert b[0].shape == (2, 3) and b[1].shape == (3,) and b[2] is row and b[3] is inv
    assert torch.equal(b[0][0], b[1])
    flat = ext.flatten_tensor_variables([torch.ones(2, 3), torch.zeros(4)])
    assert flat.shape == (10,) and float(flat.sum()) == 6.0


def test_small_helper_apis(tmp_path):
    """tensor_utils / logger helpers that scripts written against the reference call."""
    import json
    from rllab_amd.misc import logger, tensor_utils as tu
    d = dict(a=np.arange(24.).reshape(2, 3, 4), b=dict(c=np.arange(6.).reshape(2, 3)))
    f = tu.flatten_first_axis_tensor_dict(d)
    assert f["a"].shape == (6, 4) and f["b"]["c"].shape == (6,)
    np.random.seed(0)
    sub = tu.concat_tensor_list_subsample([np.arange(10), np.arange(100, 104)], 0.5)
    assert len(sub) == 5 + 2 and set(sub[:5]) <= set(range(10)) and set(sub[5:]) <= set(range(100, 104))
    sd = tu.concat_tensor_dict_list_subsample([dict(x=np.arange(4), y=dict(z=np.arange(4)))] * 2, 0.5)
    assert sd["x"].shape == (4,) and sd["y"]["z"].shape == (4,)
    assert abs(sum(tu.high_res_normalize([0.1, 0.2, 0.3])) - 1.0) < 1e-15
    logger.record_tabular_misc_stat("Ret", [1.0, 2.0, 6.0])
    logger.record_tabular_misc_stat("Len", [], placement='front')
    tab = logger.get_tabular()
    assert float(tab["RetAverage"]) == 3.0 and float(tab["RetMedian"]) == 2.0 and float(tab["RetMax"]) == 6.0
    assert np.isnan(float(tab["AverageLen"]))
    logger.dump_tabular()
    logger.log_variant(str(tmp_path / "v" / "variant.json"), dict(lr=0.1, env=object))
    assert json.load(open(str(tmp_path / "v" / "variant.json")))["lr"] == 0.1


def test_reparam_action_and_optimize_gen(quiet_logger):
    from rllab_amd.envs.env_spec import EnvSpec
    from rllab_amd.optimizers.first_order_optimizer import FirstOrderOptimizer
    from rllab_amd.policies.gaussian_mlp_policy import GaussianMLPPolicy
    from rllab_amd.spaces import Box
    np.random.seed(0)
    pol = GaussianMLPPolicy(EnvSpec(Box(-np.ones(3), np.ones(3)), Box(-np.ones(2), np.ones(2))), hidden_sizes=(8, 8))
    obs = np.random.randn(5, 3)
    d = pol.dist_info(obs)
    act = d["mean"] + np.exp(d["log_std"]) * np.random.randn(5, 2)
    # at unchanged parameters the reparameterised action is the action itself
    got = pol.get_reparam_action_sym(obs, act, d).detach().cpu().numpy()
    assert np.allclose(got, act, atol=1e-5)
    # optimize_gen yields between mini-batch steps; optimize() drains it
    target = pol
    x = torch.as_tensor(np.random.randn(3, 40), dtype=torch.float32, device=pol.flat_params.device)

    def loss(flat, xs):
        return (pol.mean_planes(xs, flat) ** 2).mean()
    opt = FirstOrderOptimizer(max_epochs=2, batch_size=10, learning_rate=1e-2)
    opt.update_opt(loss, target, [x])
    l0 = opt.loss([x])
    assert sum(1 for _ in opt.optimize_gen([x], yield_itr=0)) == 8     # 2 epochs x 4 mini-batches
    assert opt.loss([x]) < l0


def test_preconditioned_cg_solves_and_matches_plain_cg_with_identity():
    from rllab_amd.misc import krylov
    rng = np.random.RandomState(0)
    A = rng.randn(12, 12)
    A = torch.as_tensor(A @ A.T + 12 * np.eye(12))
    b = torch.as_tensor(rng.randn(12))
    f = lambda v: A @ v
    x = krylov.preconditioned_cg(f, lambda v: v / torch.diagonal(A), b, cg_iters=40, residual_tol=1e-30)
    assert float((A @ x - b).abs().max()) < 1e-8
    same = krylov.preconditioned_cg(f, lambda v: v.clone(), b, cg_iters=5)
    assert torch.allclose(same, krylov.cg(f, b, cg_iters=5), rtol=1e-12, atol=1e-14)


def test_device_io_and_fold_stats_on_cpu():
    """read_async / upload_async degrade to immediate copies without a device; fold_stats sums the additive
    statistics columns over ranks and folds the extrema columns."""
    from rllab_amd.misc.device_io import read_async, upload_async
    from rllab_amd.sampler import base as B
    t = torch.arange(6.).reshape(2, 3)
    r = read_async(t)
    t += 1                                              # the handle holds its own copy
    assert np.array_equal(r.get(), np.arange(6.).reshape(2, 3)) and r.get() is r.get()
    u = upload_async(np.arange(4.), torch.float64, torch.device("cpu"))
    assert u.dtype == torch.float64 and u.tolist() == [0.0, 1.0, 2.0, 3.0]
    rows = np.zeros((2, 20))
    rows[0, B._COUNT], rows[1, B._COUNT] = 10, 5
    rows[0, B._ADVMIN], rows[1, B._ADVMIN] = -1.0, -3.0
    rows[0, B._UNDMAX], rows[1, B._UNDMAX] = 7.0, 2.0
    rows[0, B._PROGMIN], rows[1, B._PROGMIN] = 0.5, 0.25
    out = B.fold_stats(rows)
    assert out[B._COUNT] == 15 and out[B._ADVMIN] == -3.0 and out[B._UNDMAX] == 7.0 and out[B._PROGMIN] == 0.25
    assert np.array_equal(B.fold_stats(rows[:1]), rows[0])


def test_sliced_fun_is_the_sample_weighted_mean():
    """ext.sliced_fun (contract: rllab/misc/ext.py:341-370): chunks of max(1, n // k) samples, a ragged tail as one
    more chunk, outputs averaged with the chunk lengths as weights, container kind of the output preserved."""
    from rllab_amd.misc import ext
    x = np.arange(10.0)
    y = x * x
    calls = []

    def f(xs, ys, scale):
        calls.append(len(xs))
        return xs.mean() * scale, np.array([ys.mean(), ys.max()])
    out = ext.sliced_fun(f, 3)([x, y], [2.0])
    assert calls == [3, 3, 3, 1] and isinstance(out, tuple)
    assert abs(out[0] - 2.0 * x.mean()) < 1e-12                       # a mean of means with the right weights
    assert abs(out[1][0] - y.mean()) < 1e-12
    assert abs(out[1][1] - (3 * 4 + 3 * 25 + 3 * 64 + 81) / 10.0) < 1e-12   # max is NOT averageable: weighted as is
    assert ext.sliced_fun(lambda xs: xs.sum(), 1)([x]) == x.sum()      # one slice: the plain call, bare value
    as_list = ext.sliced_fun(lambda xs: [xs.mean()], 2)([x], ())
    assert isinstance(as_list, list) and abs(as_list[0] - 4.5) < 1e-12
    assert abs(ext.sliced_fun(lambda xs: xs.mean(), 100)([x]) - 4.5) < 1e-12   # more slices than samples: size 1
    assert ext.extract(dict(a=1, b=2), "b", "a") == (2, 1)
    assert ext.extract([dict(a=1), dict(a=3)], "a") == ([1, 3],)


def test_sliced_fun_against_the_reference_fixture():
    """tests/golden/sliced_fun.npz: outputs of the REAL rllab.misc.ext.sliced_fun (oracle/make_golden.py)."""
    from rllab_amd.misc import ext
    from tests.golden_util import load
    g = load("sliced_fun")

    def target(xs, ys, w):
        return (xs.dot(w) * ys).mean(), np.array([xs.mean(axis=0).sum(), (ys ** 2).mean()])
    for k in (1, 2, 4, 5, 23, 40):
        a, b = ext.sliced_fun(target, k)([g["x"], g["y"]], [g["w"]])
        assert np.allclose(a, g["k%d_0" % k], rtol=1e-13, atol=0) and np.allclose(b, g["k%d_1" % k], rtol=1e-13, atol=0)
    assert np.isclose(ext.sliced_fun(lambda xs: xs.mean(), 4)([g["y"]]), g["bare"], rtol=1e-13)


def test_log_dir_switch(tmp_path):
    """RLLAB_LOG_DIR moves config.LOG_DIR (read at import: checked in a child interpreter)."""
    import subprocess
    import sys
    env = dict(os.environ, RLLAB_LOG_DIR=str(tmp_path))
    out = subprocess.check_output([sys.executable, "-c", "from rllab_amd import config; print(config.LOG_DIR)"],
                                  env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.decode().strip() == str(tmp_path)


def test_swimmer_limit_model_option_travels_with_the_env():
    """SwimmerEnv(limit_model="mujoco") sets rl_env_cfg flag RL_CFG_LIMIT_MUJOCO, survives pickling (snapshots,
    workers) and cloning; anything but the two models is rejected."""
    from rllab_amd import _lib
    from rllab_amd.envs.mujoco.swimmer_env import SwimmerEnv
    from rllab_amd.envs.normalized_env import normalize
    assert SwimmerEnv()._cfg.get("flags", 0) == 0 and SwimmerEnv().limit_model == "penalty"
    env = SwimmerEnv(limit_model="mujoco", ctrl_cost_coeff=0.02)
    assert env._cfg["flags"] == _lib.CFG_LIMIT_MUJOCO and env._cfg["ctrl_cost_coeff"] == 0.02
    back = pickle.loads(pickle.dumps(normalize(env)))
    assert back.wrapped_env.limit_model == "mujoco" and back.wrapped_env._cfg["flags"] == _lib.CFG_LIMIT_MUJOCO
    with pytest.raises(ValueError):
        SwimmerEnv(limit_model="lcp")


@pytest.mark.parametrize("mod,cls", [("half_cheetah_env", "HalfCheetahEnv"), ("walker2d_env", "Walker2DEnv"), ("hopper_env", "HopperEnv")])
def test_legged_constraint_model_options_travel_with_the_env(mod, cls):
    """HalfCheetahEnv / Walker2DEnv / HopperEnv(limit_model=.., contact_model=..) set RL_CFG_LIMIT_MUJOCO / RL_CFG_CONTACT_MUJOCO,
    default to the penalty models, survive pickling; unknown models are rejected."""
    import importlib
    from rllab_amd import _lib
    from rllab_amd.envs.normalized_env import normalize
    Env = getattr(importlib.import_module("rllab_amd.envs.mujoco." + mod), cls)
    assert Env()._cfg.get("flags", 0) == 0 and (Env().limit_model, Env().contact_model) == ("penalty", "penalty")
    assert Env(limit_model="mujoco")._cfg["flags"] == _lib.CFG_LIMIT_MUJOCO
    assert Env(contact_model="mujoco")._cfg["flags"] == _lib.CFG_CONTACT_MUJOCO
    env = Env(limit_model="mujoco", contact_model="mujoco")
    assert env._cfg["flags"] == _lib.CFG_LIMIT_MUJOCO | _lib.CFG_CONTACT_MUJOCO == 12
    back = pickle.loads(pickle.dumps(normalize(env)))
    assert back.wrapped_env._cfg["flags"] == 12 and back.wrapped_env.contact_model == "mujoco"
    with pytest.raises(ValueError):
        Env(contact_model="lcp")
    with pytest.raises(ValueError):
        Env(limit_model="hard")


def test_ext_helpers_behave_like_the_reference_module():
    """The Theano-free helpers of rllab/misc/ext.py (:23-40, :71-120, :151-182, :302-338, :373-391), each held to
    what the reference's does on hand-worked cases -- including its quirks: a scan starts from ``base`` only when
    ``base`` is truthy, ``lazydict.set`` replaces the thunk but not a value already computed."""
    import random
    import torch
    from rllab_amd.misc import ext
    f = lambda acc, x: acc * 2 + x
    assert ext.scanl(f, [1, 2, 3]) == [1, 4, 11]
    assert ext.scanl(f, [1, 2, 3], 0) == [1, 4, 11]                   # 0 is "no base"
    assert ext.scanl(f, [1, 2, 3], 5) == [11, 24, 51]
    assert ext.scanr(f, [1, 2, 3]) == [3, 7, 9]                       # f(x, acc) over the reversed list
    assert ext.scanr(f, [1, 2, 3], 1) == [7, 11, 13] and ext.scanl(f, []) == []
    assert list(ext.iscanl(lambda a, b: a + b, iter([1, 1, 1]))) == [1, 2, 3]
    assert ext.compact(dict(a=1, b=None)) == dict(a=1) and ext.compact([None, 0, None, 2]) == [0, 2] and ext.compact(3) == 3
    assert ext.extract_dict(dict(a=1, b=2), "b", "zz") == dict(b=2)
    assert ext.flatten([[1, 2], [], [3]]) == [1, 2, 3]
    n = [0]

    def thunk():
        n[0] += 1
        return 7
    ld = ext.lazydict(a=thunk)
    assert n[0] == 0 and ld["a"] == 7 and ld["a"] == 7 and n[0] == 1
    assert ld.get("a") == 7 and ld.get("zz", 4) == 4
    ld["b"] = lambda: 9
    assert ext.extract(ld, "b", "a") == (9, 7)
    ld.set("b", lambda: 10)
    assert ld["b"] == 9
    ad = ext.AttrDict(a=1)
    ad.b = 2
    assert ad["b"] == 2 and ad.a == 1 and dict(ad) == dict(a=1, b=2)
    p1 = dict(states=np.zeros((4, 2)), rewards=np.arange(4.0), only_here=np.ones(4))
    p2 = dict(states=np.ones((3, 2)), rewards=np.arange(3.0))
    both = ext.concat_paths(p1, p2)
    assert sorted(both) == ["rewards", "states"] and both["states"].shape == (7, 2) and both["rewards"][4] == 0.0
    assert ext.path_len(p1) == 4 and ext.path_len(ext.truncate_path(p1, 2)) == 2
    random.seed(0)
    order = list(ext.shuffled(range(10)))
    assert sorted(order) == list(range(10)) and order != list(range(10))
    random.seed(0)
    assert list(ext.shuffled(range(10))) == order
    assert ext.flatten_shape_dim((2, 3, 4)) == 24 and ext.flatten_shape_dim(()) == 1
    x = np.array([[1.0, 10.0], [3.0, 30.0]])
    assert np.allclose(ext.stdize(x, eps=0.0), [[-1, -1], [1, 1]])
    y = np.arange(5)
    xs = np.arange(10.0).reshape(5, 2)
    got = list(ext.iterate_minibatches_generic([xs, y], 2))
    assert [len(b[1]) for b in got] == [2, 2, 1] and np.array_equal(got[2][0], xs[4:]) and np.array_equal(got[1][1], [2, 3])
    assert len(list(ext.iterate_minibatches_generic([xs, y]))) == 1
    np.random.seed(1)
    sh = list(ext.iterate_minibatches_generic([xs, y], 2, shuffle=True))
    perm = np.concatenate([b[1] for b in sh])
    np.random.seed(1)
    want = np.arange(5)
    np.random.shuffle(want)
    assert np.array_equal(perm, want) and all(np.array_equal(b[0], xs[b[1]]) for b in sh)
    ts = [torch.arange(6.0).reshape(2, 3), torch.arange(4.0)]
    back = ext.unflatten_tensor_variables(ext.flatten_tensor_variables(ts), [t.shape for t in ts], ts)
    assert all(torch.equal(a, b) for a, b in zip(ts, back))
