#!/usr/bin/env python
"""Golden fixtures for LbfgsOptimizer / PenaltyLbfgsOptimizer: the REAL reference classes
(rllab/optimizers/lbfgs_optimizer.py, penalty_lbfgs_optimizer.py) run on analytic toy problems through
their ``_opt_fun`` hooks (plain attributes; TEST INFRASTRUCTURE, needs /root/reference).
    python oracle/make_golden_lbfgs.py  ->  tests/golden/lbfgs_optimizers.npz"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.make_golden import OUT, install_shims  # noqa: E402


class _Target(object):
    def __init__(self, theta):
        self.theta = theta.copy()

    def get_param_values(self, **tags):
        return self.theta.copy()

    def set_param_values(self, v, **tags):
        self.theta = np.array(v, dtype=np.float64)


def main():
    install_shims()
    from rllab.optimizers.lbfgs_optimizer import LbfgsOptimizer
    from rllab.optimizers.penalty_lbfgs_optimizer import PenaltyLbfgsOptimizer
    from rllab.misc import logger
    logger.log = lambda *a, **k: None
    rng = np.random.RandomState(7)
    n = 10
    A = rng.randn(n, n)
    A = A.T.dot(A) / n + 0.1 * np.eye(n)
    C = rng.randn(n, n)
    C = C.T.dot(C) / n + 0.1 * np.eye(n)
    b = rng.randn(n)
    out = dict(A=A, C=C, b=b)

    def loss_fn(th, th0):        # convex, non-quadratic
        d = th - th0
        return float(b.dot(d) + 0.5 * d.dot(A).dot(d) + 0.1 * np.sum(d ** 4))

    def loss_grad(th, th0):
        d = th - th0
        return b + A.dot(d) + 0.4 * d ** 3

    def cons_fn(th, th0):
        d = th - th0
        return float(0.5 * d.dot(C).dot(d))

    def cons_grad(th, th0):
        return C.dot(th - th0)

    # ---- LbfgsOptimizer ----------------------------------------------------------------------
    th0 = rng.randn(n) * 0.1
    target = _Target(th0)
    opt = LbfgsOptimizer(max_opt_itr=20)
    opt._target = target
    opt._opt_fun = dict(f_loss=lambda *a: loss_fn(target.theta, th0),
                        f_opt=lambda *a: [np.float64(loss_fn(target.theta, th0)), loss_grad(target.theta, th0)])
    opt.optimize([np.zeros(1)])
    out.update(lbfgs_theta0=th0, lbfgs_theta1=target.theta.copy())

    # ---- PenaltyLbfgsOptimizer: tight / loose constraint, adaptive and fixed penalty ---------------
    for tag, (eps, kw) in dict(tight=(1e-3, {}), loose=(5.0, {}), fixed=(1e-2, dict(adapt_penalty=False)),
                               few=(1e-4, dict(max_penalty_itr=3))).items():
        th0 = rng.randn(n) * 0.1
        target = _Target(th0)
        opt = PenaltyLbfgsOptimizer(**kw)
        opt._target = target
        opt._max_constraint_val = eps
        opt._constraint_name = "c"

        def f_opt(*a, th0=th0, target=target):
            pen = a[-1]
            return [np.float64(loss_fn(target.theta, th0) + pen * cons_fn(target.theta, th0)),
                    loss_grad(target.theta, th0) + pen * cons_grad(target.theta, th0)]

        def f_pen(*a, th0=th0, target=target):
            pen = a[-1]
            l, c = loss_fn(target.theta, th0), cons_fn(target.theta, th0)
            return l + pen * c, l, c
        opt._opt_fun = dict(f_loss=lambda *a, th0=th0, target=target: loss_fn(target.theta, th0),
                            f_constraint=lambda *a, th0=th0, target=target: cons_fn(target.theta, th0),
                            f_penalized_loss=f_pen, f_opt=f_opt)
        opt.optimize([np.zeros(1)])
        out.update({"pen_theta0_" + tag: th0, "pen_theta1_" + tag: target.theta.copy(), "pen_eps_" + tag: eps,
                    "pen_penalty_" + tag: float(opt._penalty), "pen_cons_" + tag: cons_fn(target.theta, th0)})
    np.savez(os.path.join(OUT, "lbfgs_optimizers.npz"), **out)
    print({k: (v if np.ndim(v) == 0 else np.shape(v)) for k, v in out.items()})


if __name__ == "__main__":
    main()
