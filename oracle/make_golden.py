#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REAL reference functions (TEST
INFRASTRUCTURE; needs /root/reference, so it runs only in the build container).

The reference is imported unmodified behind an import shim: stub modules for the
third-party packages that are not installed (theano, lasagne, Box2D, mako, pyprind,
cached_property, path, pygame) -- none of the functions exercised here calls into
them -- plus the joblib ``MemmapingPool`` spelling and ``_ast.Num`` (removed from
modern Python; conjugate_gradient_optimizer.py:10 imports it).

Every fixture stores the seeded inputs next to the reference outputs, so tests can
replay them through (a) the numpy oracle (oracle/np_reference.py) on CPU and (b) the
HIP path on the GPU box, where /root/reference does not exist.
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def install_shims():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import ref_shim
    ref_shim.install(REF)


def synth_paths(rng, n_paths, obs_dim, act_dim, max_len):
    paths = []
    for _ in range(n_paths):
        L = int(rng.randint(1, max_len + 1))
        paths.append(dict(
            observations=rng.randn(L, obs_dim) * 3.0,
            actions=rng.randn(L, act_dim),
            rewards=rng.randn(L) + 1.0,
            agent_infos=dict(mean=rng.randn(L, act_dim), log_std=np.tile(rng.randn(act_dim) * 0.1, (L, 1))),
            env_infos=dict()))
    return paths


def pack(paths):
    """Flatten a list of path dicts into arrays + offsets for storage."""
    lens = np.array([len(p["rewards"]) for p in paths])
    cat = lambda f: np.concatenate([f(p) for p in paths])
    return dict(lens=lens, observations=cat(lambda p: p["observations"]), actions=cat(lambda p: p["actions"]),
                rewards=cat(lambda p: p["rewards"]), mean=cat(lambda p: p["agent_infos"]["mean"]),
                log_std=cat(lambda p: p["agent_infos"]["log_std"]))


def main():
    install_shims()
    os.makedirs(OUT, exist_ok=True)
    from rllab.misc import special, krylov, tensor_utils
    from rllab.algos import util
    from rllab.baselines.linear_feature_baseline import LinearFeatureBaseline
    from rllab.distributions.diagonal_gaussian import DiagonalGaussian
    from rllab.sampler.base import BaseSampler
    from rllab.sampler import parallel_sampler
    from rllab.envs.normalized_env import NormalizedEnv
    from rllab.envs.base import Env, Step
    from rllab.spaces.box import Box
    import rllab.misc.logger as ref_logger
    ref_logger.log = lambda *a, **k: None

    rng = np.random.RandomState(20260921)

    # ---- special / util ------------------------------------------------------------
    x = rng.randn(137)
    X2 = rng.randn(50, 7)
    y, yp = rng.randn(500), rng.randn(500)
    adv = rng.randn(1000) * 3 + 1
    np.savez(os.path.join(OUT, "special_util.npz"),
             x=x, dc_099=special.discount_cumsum(x, 0.99), dc_05=special.discount_cumsum(x, 0.5),
             X2=X2, dc2_097=special.discount_cumsum(X2, 0.97),
             arange5=special.discount_cumsum(np.arange(5.0), 0.9),
             y=y, yp=yp, ev=special.explained_variance_1d(yp, y),
             ev_const=special.explained_variance_1d(yp, np.ones(500)),
             ev_const_both=special.explained_variance_1d(np.zeros(500), np.ones(500)),
             adv=adv, centered=util.center_advantages(adv), shifted=util.shift_advantages_to_positive(adv))

    # ---- DiagonalGaussian ------------------------------------------------------------
    d = DiagonalGaussian(3)
    om, ols, nm, nls, xs = (rng.randn(64, 3), rng.randn(64, 3) * 0.3, rng.randn(64, 3), rng.randn(64, 3) * 0.3,
                            rng.randn(64, 3))
    np.savez(os.path.join(OUT, "diagonal_gaussian.npz"), om=om, ols=ols, nm=nm, nls=nls, xs=xs,
             kl=d.kl(dict(mean=om, log_std=ols), dict(mean=nm, log_std=nls)),
             logli=d.log_likelihood(xs, dict(mean=nm, log_std=nls)),
             entropy=d.entropy(dict(mean=nm, log_std=nls)))

    # ---- LinearFeatureBaseline + process_samples -----------------------------------------
    class _Policy(object):
        recurrent = False
        distribution = DiagonalGaussian(2)

    class _Algo(object):
        pass

    for tag, (n_paths, od, ad, ml, lam, center, positive) in dict(
            a=(40, 4, 1, 100, 1.0, True, False), b=(25, 13, 2, 60, 0.97, True, True),
            c=(3, 6, 1, 5, 0.9, False, False)).items():
        paths = synth_paths(rng, n_paths, od, ad, ml)
        packed = pack(paths)
        algo = _Algo()
        algo.baseline = LinearFeatureBaseline(env_spec=None)
        algo.policy = _Policy()
        algo.policy.distribution = DiagonalGaussian(ad)
        algo.discount, algo.gae_lambda = 0.99, lam
        algo.center_adv, algo.positive_adv = center, positive
        sampler = BaseSampler(algo)
        ref_logger._tabular[:] = []
        out1 = sampler.process_samples(0, [dict(p, agent_infos=dict(p["agent_infos"])) for p in paths])
        tab1 = dict(ref_logger._tabular)
        coeffs1 = algo.baseline._coeffs.copy()
        # second pass with the now-fitted baseline (non-zero predictions)
        paths2 = [dict(observations=p["observations"], actions=p["actions"], rewards=p["rewards"],
                       agent_infos=p["agent_infos"], env_infos={}) for p in paths]
        ref_logger._tabular[:] = []
        out2 = sampler.process_samples(1, paths2)
        tab2 = dict(ref_logger._tabular)
        keys = ['AverageDiscountedReturn', 'AverageReturn', 'ExplainedVariance', 'NumTrajs', 'Entropy',
                'Perplexity', 'StdReturn', 'MaxReturn', 'MinReturn']
        np.savez(os.path.join(OUT, "process_samples_%s.npz" % tag),
                 discount=0.99, gae_lambda=lam, center_adv=center, positive_adv=positive,
                 adv1=out1["advantages"], ret1=out1["returns"], coeffs1=coeffs1,
                 adv2=out2["advantages"], ret2=out2["returns"], coeffs2=algo.baseline._coeffs.copy(),
                 stats1=np.array([float(tab1[k]) for k in keys]),
                 stats2=np.array([float(tab2[k]) for k in keys]), stat_keys=np.array(keys), **packed)

    # ---- truncate_paths (tests/test_sampler.py of the reference) ---------------------------
    tp = [dict(observations=np.zeros((100, 1)), actions=np.zeros((100, 1)), rewards=np.zeros(100),
               env_infos=dict(), agent_infos=dict(lala=np.zeros(100))),
          dict(observations=np.zeros((50, 1)), actions=np.zeros((50, 1)), rewards=np.zeros(50),
               env_infos=dict(), agent_infos=dict(lala=np.zeros(50)))]
    t130 = parallel_sampler.truncate_paths(tp, 130)
    t90 = parallel_sampler.truncate_paths(tp, 90)
    np.savez(os.path.join(OUT, "truncate_paths.npz"),
             lens130=np.array([len(p["rewards"]) for p in t130]), lens90=np.array([len(p["rewards"]) for p in t90]),
             info130=np.array([len(p["agent_infos"]["lala"]) for p in t130]))

    # ---- krylov.cg ---------------------------------------------------------------------
    A = rng.randn(30, 30)
    A = A.T.dot(A) + 0.1 * np.eye(30)
    b = rng.randn(30)
    np.savez(os.path.join(OUT, "krylov_cg.npz"), A=A, b=b,
             x10=krylov.cg(lambda v: A.dot(v), b, cg_iters=10),
             x3=krylov.cg(lambda v: A.dot(v), b, cg_iters=3),
             x_early=krylov.cg(lambda v: 4.0 * v, b, cg_iters=10))   # converges in 1 iteration

    # ---- NormalizedEnv action map ----------------------------------------------------------
    class _E(Env):
        def __init__(self):
            self.last = None
        action_space = Box(np.array([-10.0, -2.0]), np.array([10.0, 6.0]))
        observation_space = Box(-np.ones(3), np.ones(3))

        def reset(self):
            return np.zeros(3)

        def step(self, a):
            self.last = np.array(a)
            return Step(np.zeros(3), 2.5, False)
    e = _E()
    ne = NormalizedEnv(e, scale_reward=0.1)
    acts = rng.randn(20, 2) * 1.5
    scaled, rews = [], []
    for a in acts:
        _, r, _, _ = ne.step(a)
        scaled.append(e.last.copy())
        rews.append(r)
    np.savez(os.path.join(OUT, "normalized_env.npz"), acts=acts, scaled=np.array(scaled), rews=np.array(rews),
             lb=e.action_space.low, ub=e.action_space.high)

    # ---- ConjugateGradientOptimizer.optimize control flow -------------------------------------
    from rllab.optimizers.conjugate_gradient_optimizer import ConjugateGradientOptimizer

    class _Target(object):
        def __init__(self, theta):
            self.theta = theta.copy()

        def get_param_values(self, **tags):
            return self.theta.copy()

        def set_param_values(self, v, **tags):
            self.theta = np.array(v, dtype=np.float64)

        def flat_to_params(self, x, **tags):
            return [x]

    n = 12
    Hm = rng.randn(n, n)
    Hm = Hm.T.dot(Hm) / n + 0.05 * np.eye(n)
    Cm = rng.randn(n, n)
    Cm = Cm.T.dot(Cm) / n + 0.05 * np.eye(n)
    gvec = rng.randn(n)
    results = {}
    for tag, (delta, quartic) in dict(easy=(0.01, 0.0), backtrack=(0.5, 40.0), reject=(0.01, -1.0)).items():
        theta0 = rng.randn(n) * 0.1
        target = _Target(theta0)

        def loss_fn(th, quartic=quartic, theta0=theta0):
            d = th - theta0
            if quartic < 0:   # loss that never improves -> rejection branch
                return float(gvec.dot(d)) ** 2 + 1.0
            return float(gvec.dot(d) + 0.5 * d.dot(Hm).dot(d) + quartic * np.sum(d ** 4))

        def cons_fn(th, theta0=theta0):
            d = th - theta0
            return float(0.5 * d.dot(Cm).dot(d))

        opt = ConjugateGradientOptimizer()

        class _Hvp(object):
            def update_opt(self, *a, **k):
                pass

            def build_eval(self, inputs):
                return lambda x: Cm.dot(x) + 1e-5 * x
        opt._hvp_approach = _Hvp()
        opt._target = target
        opt._max_constraint_val = delta
        opt._constraint_name = "c"
        grad_fn = (lambda q: (lambda: gvec if q >= 0 else 2 * gvec * 0.0 + gvec))(quartic)
        opt._opt_fun = dict(
            f_loss=lambda *a: loss_fn(target.theta),
            f_grad=lambda *a: grad_fn(),
            f_constraint=lambda *a: cons_fn(target.theta),
            f_loss_constraint=lambda *a: (loss_fn(target.theta), cons_fn(target.theta)))
        opt.optimize((np.zeros((4, 1)),))
        results["theta0_" + tag] = theta0
        results["theta1_" + tag] = target.theta.copy()
        results["delta_" + tag] = delta
        results["quartic_" + tag] = quartic
    np.savez(os.path.join(OUT, "cg_optimizer.npz"), Hm=Hm, Cm=Cm, gvec=gvec, **results)

    # ---- flat parameter layout helpers -----------------------------------------------------
    shapes = [(4, 32), (32,), (32, 32), (32,), (32, 1), (1,), (1,)]
    tensors = [rng.randn(*s) for s in shapes]
    flat = tensor_utils.flatten_tensors(tensors)
    np.savez(os.path.join(OUT, "flat_params.npz"), flat=flat, W0=tensors[0], b0=tensors[1], W1=tensors[2],
             b1=tensors[3], W2=tensors[4], b2=tensors[5], log_std=tensors[6])
    # ---- ext.sliced_fun (rllab/misc/ext.py:341-370) ------------------------------------------
    from rllab.misc import ext as ref_ext
    sx, sy = rng.randn(23, 3), rng.randn(23)
    sw = rng.randn(3)

    def sliced_target(xs, ys, w):
        return (xs.dot(w) * ys).mean(), np.array([xs.mean(axis=0).sum(), (ys ** 2).mean()])
    sl = {}
    for k in (1, 2, 4, 5, 23, 40):
        a, b = ref_ext.sliced_fun(sliced_target, k)([sx, sy], [sw])
        sl["k%d_0" % k], sl["k%d_1" % k] = a, b
    bare = ref_ext.sliced_fun(lambda xs: xs.mean(), 4)([sy])
    np.savez(os.path.join(OUT, "sliced_fun.npz"), x=sx, y=sy, w=sw, bare=bare, **sl)
    print("golden fixtures written to", OUT, sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
