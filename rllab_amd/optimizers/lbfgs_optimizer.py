"""LbfgsOptimizer (API and control flow of rllab/optimizers/lbfgs_optimizer.py:8-90).

``loss`` is a closure ``f(flat_params, *inputs) -> 0-d tensor`` (the reference hands a Theano
expression and compiles ``f_loss`` / ``f_opt``); value and gradient are evaluated with torch
autograd on the device, summed over env shards (one all-reduce each), and handed to
``scipy.optimize.fmin_l_bfgs_b`` in float64 exactly as the reference does (:87-90) -- including
its quirk that the target keeps the parameters of the LAST function evaluation.
"""
import time

import numpy as np
import scipy.optimize
import torch

from rllab_amd.core.serializable import Serializable
from rllab_amd.sampler import dist as D


def value_and_grad(fn, target, inputs):
    """float64 (value, flat gradient over the trainable parameters) of ``fn`` at target's parameters."""
    flat = target.flat_params.detach().clone().requires_grad_(True)
    val = fn(flat, *inputs)
    (g,) = torch.autograd.grad(val, flat, allow_unused=True)
    if g is None:
        g = torch.zeros_like(flat)
    idx = target._flat_index(trainable=True)
    if idx is not None:
        g = g[idx]
    packed = torch.cat([val.detach().reshape(1).to(torch.float64), g.to(torch.float64)])
    D.all_reduce_sum_(packed)
    host = packed.cpu().numpy()
    return float(host[0]), host[1:].copy()


class LbfgsOptimizer(Serializable):
    """Unconstrained optimisation via L-BFGS."""

    def __init__(self, max_opt_itr=20, callback=None):
        Serializable.quick_init(self, locals())
        self._max_opt_itr = max_opt_itr
        self._loss = None
        self._target = None
        self._callback = callback

    def update_opt(self, loss, target, inputs=None, extra_inputs=None, gradients=None, *args, **kwargs):
        """``fused`` (keyword): an object with ``accepts(inputs)``, ``loss_and_kl(inputs)`` and
        ``value_and_grad(inputs, penalty)`` evaluating the same loss with HIP kernels
        (regressors/fused_regressor_ops.py); the closure stays the definition and the fallback."""
        self._target = target
        self._loss = loss
        self._fused = kwargs.get("fused")

    def _fused_for(self, inputs):
        f = getattr(self, "_fused", None)
        return f if (f is not None and f.accepts(inputs)) else None

    def loss(self, inputs, extra_inputs=None):
        inputs = tuple(inputs) + tuple(extra_inputs or ())
        if self._fused_for(inputs) is not None:
            return self._fused.loss_and_kl(inputs)[0]
        with torch.no_grad():
            v = self._loss(self._target.flat_params, *inputs).to(torch.float64)
        return float(D.all_reduce_sum_(v))

    def optimize(self, inputs, extra_inputs=None):
        inputs = tuple(inputs) + tuple(extra_inputs or ())

        fused = self._fused_for(inputs)

        def f_opt_wrapper(flat_params):
            self._target.set_param_values(flat_params, trainable=True)
            if fused is not None:
                return fused.value_and_grad(inputs)
            return value_and_grad(self._loss, self._target, inputs)

        itr = [0]
        start_time = time.time()
        if self._callback:
            def opt_callback(params):
                self._callback(dict(loss=self.loss(inputs), params=params, itr=itr[0],
                                    elapsed=time.time() - start_time))
                itr[0] += 1
        else:
            opt_callback = None
        scipy.optimize.fmin_l_bfgs_b(
            func=f_opt_wrapper, x0=np.asarray(self._target.get_param_values(trainable=True), dtype=np.float64),
            maxiter=self._max_opt_itr, callback=opt_callback)
