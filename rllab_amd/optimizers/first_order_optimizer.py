"""FirstOrderOptimizer (API of rllab/optimizers/first_order_optimizer.py:14-137).

VPG uses it with ``batch_size=None, max_epochs=1`` (rllab/algos/vpg.py:26-34): one
full-batch Adam step per iteration, moments persisting across iterations.  The
update rule is Lasagne's ``adam`` (third party, pinned fork in the reference's
environment.yml): t += 1; a_t = lr*sqrt(1-b2^t)/(1-b1^t); m = b1*m + (1-b1)*g;
v = b2*v + (1-b2)*g^2; theta -= a_t * m / (sqrt(v) + eps), b1=0.9, b2=0.999,
eps=1e-8.  State lives on the device; gradients are sum-all-reduced across ranks.
"""
import time

import numpy as np
import torch

from rllab_amd.core.serializable import Serializable
from rllab_amd.misc import logger
from rllab_amd.optimizers.minibatch_dataset import BatchDataset
from rllab_amd.sampler import dist as D


class _Adam(object):
    def __init__(self, learning_rate=1e-3, beta1=0.9, beta2=0.999, epsilon=1e-8):
        self.lr, self.b1, self.b2, self.eps = learning_rate, beta1, beta2, epsilon
        self.t = 0
        self.m = None
        self.v = None

    def step(self, param, grad):
        if self.m is None:
            self.m = torch.zeros_like(grad)
            self.v = torch.zeros_like(grad)
        self.t += 1
        a_t = self.lr * np.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        self.m = self.b1 * self.m + (1 - self.b1) * grad
        self.v = self.b2 * self.v + (1 - self.b2) * grad * grad
        return param - a_t * self.m / (torch.sqrt(self.v) + self.eps)

    def step_in_place(self, theta32, grad):
        """The same step written straight into the float32 device parameter vector by rl_adam_step (one launch
        instead of ~10 small tensor kernels); float64 arithmetic and moments as in ``step``."""
        from rllab_amd import _lib
        grad = grad.to(torch.float64).contiguous()
        if self.m is None:
            self.m = torch.zeros_like(grad)
            self.v = torch.zeros_like(grad)
        self.t += 1
        a_t = self.lr * np.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        _lib.check(_lib.lib.rl_adam_step(theta32.numel(), _lib.ptr(theta32), _lib.ptr(grad), _lib.ptr(self.m),
                                         _lib.ptr(self.v), float(a_t), self.b1, self.b2, self.eps,
                                         _lib.stream_ptr()), "rl_adam_step")


class _SGD(object):
    def __init__(self, learning_rate=1e-3):
        self.lr = learning_rate

    def step(self, param, grad):
        return param - self.lr * grad


def adam(learning_rate=1e-3, **kw):
    return _Adam(learning_rate=learning_rate, **kw)


def sgd(learning_rate=1e-3, **kw):
    return _SGD(learning_rate=learning_rate)


class FirstOrderOptimizer(Serializable):
    reports_before_values = True     # optimize() leaves (loss, mean KL, max KL) at its starting point in last_before

    def __init__(self, update_method=adam, learning_rate=1e-3, max_epochs=1000, tolerance=1e-6,
                 batch_size=32, callback=None, verbose=False, **kwargs):
        Serializable.quick_init(self, locals())
        self._loss = None
        self._target = None
        self._callback = callback
        self._update_factory = lambda: update_method(learning_rate=learning_rate)
        self._updater = None
        self._max_epochs = max_epochs
        self._tolerance = tolerance
        self._batch_size = batch_size
        self._verbose = verbose

    def update_opt(self, loss, target, inputs=None, extra_inputs=None, gradients=None, **kwargs):
        """``loss`` is a closure ``loss(flat_params, *inputs) -> 0-d tensor``.  The
        update state (Adam moments) is created here, once, like the reference's
        shared variables (:62-65)."""
        self._target = target
        self._loss = loss
        self._updater = self._update_factory()
        self._fused = kwargs.get("fused")
        # opt-in of the caller (algos/vpg.py passes weighted_mean_inputs=True): the inputs END with
        # [..., weights, 1 / W] and a mini-batch's mean must be renormalised by its own weight sum.  Never inferred
        # from the inputs' types: (xs, ys) inputs or appended extra_inputs keep their last entry.
        self._weighted_mean_inputs = bool(kwargs.get("weighted_mean_inputs", False))

    def loss(self, inputs, extra_inputs=None):
        inputs = tuple(inputs) + tuple(extra_inputs or ())
        if getattr(self, "_fused", None) is not None and self._fused.accepts(inputs):
            return -self._fused.loss_stats_host(inputs)[2]
        with torch.no_grad():
            v = self._loss(self._target.flat_params, *inputs).to(torch.float64)
        return float(D.all_reduce_sum_(v))

    def _step(self, inputs, with_loss=False):
        target = self._target
        if getattr(self, "_fused", None) is not None and self._fused.accepts(inputs):
            g = self._fused.loss_grad(inputs, vpg=True, with_loss=with_loss)
            if with_loss:
                self._pre_step_stats = self._fused.loss_stats_deferred(inputs)   # served by the gradient pass
        else:
            flat = target.flat_params.detach().clone().requires_grad_(True)
            g = torch.autograd.grad(self._loss(flat, *inputs), flat)[0]
            g = D.all_reduce_sum_(g.to(torch.float64))
        idx = target._flat_index(trainable=True)
        flat = target.flat_params
        if idx is None and flat.is_cuda and flat.dtype == torch.float32 and hasattr(self._updater, "step_in_place"):
            self._updater.step_in_place(flat.detach(), g)
            # the write bypassed torch: evaluation caches key on (tensor version, the fused ops' own epoch)
            if getattr(self, "_fused", None) is not None:
                self._fused._epoch += 1
            else:
                flat.add_(0)
            return
        theta = target.flat_params.detach().to(torch.float64)
        if idx is not None:
            g, theta = g[idx], theta[idx]
        target.set_param_values(self._updater.step(theta, g), trainable=True)

    def optimize(self, inputs, extra_inputs=None, callback=None, **kwargs):
        for _ in self.optimize_gen(inputs, extra_inputs=extra_inputs, callback=callback, **kwargs):
            pass

    def optimize_gen(self, inputs, extra_inputs=None, callback=None, yield_itr=None, **kwargs):
        """``optimize`` as a generator: with ``yield_itr`` set, control returns to the caller after every
        ``yield_itr + 1`` mini-batch steps (first_order_optimizer.py:78-134)."""
        inputs = tuple(inputs) + tuple(extra_inputs or ())
        if len(inputs) == 0:
            raise NotImplementedError
        # Full batch on the fused kernels: the gradient pass of an epoch's single step also produces the loss at the
        # parameters it starts from (rl_policy_grad_loss), read when the epoch's new loss is read -- no separate
        # loss pass and no blocking read before the step.  ``last_before`` = (loss, mean KL, max KL) at the start.
        fused_full = (self._batch_size is None and getattr(self, "_fused", None) is not None
                      and self._fused.accepts(inputs))
        self.last_before = None
        last_loss = None if fused_full else self.loss(inputs)
        if last_loss is not None:
            self.last_before = (last_loss, None, None)
        start_time = time.time()
        n_main = len(inputs) - len(tuple(extra_inputs or ()))             # extra_inputs ride behind, untouched
        dataset = BatchDataset(inputs, self._batch_size, sample_axis=-1)   # planes: the sample axis is last
        itr = 0
        for epoch in range(self._max_epochs):
            for batch in dataset.iterate(update=True):
                before = None
                if self._batch_size is not None and getattr(self, "_weighted_mean_inputs", False) and n_main >= 2:
                    # declared input convention (algos/npo.py, algos/vpg.py): [..., weights, 1 / W].  A mini-batch
                    # is normalised by ITS OWN (global, all-reduced) weight sum -- the reference's compiled loss is
                    # the mean over whatever slice it is given (first_order_optimizer.py:112-114)
                    w_at, inv_at = n_main - 2, n_main - 1
                    cnt = D.all_reduce_sum_(batch[w_at].to(torch.float64).sum())
                    batch[inv_at] = (1.0 / cnt.clamp_min(1.0)).to(
                        batch[inv_at].dtype if torch.is_tensor(batch[inv_at]) else torch.float64)
                if fused_full and last_loss is None:
                    self._step(tuple(batch), with_loss=True)       # records the loss / KL sums of the old parameters
                else:
                    self._step(tuple(batch))
                if yield_itr is not None and itr % (yield_itr + 1) == 0:
                    yield
                itr += 1
            if fused_full and last_loss is None:
                before = self._pre_step_stats
            new_loss = self.loss(inputs)
            if before is not None:
                s0 = before()
                last_loss = -s0[2]
                self.last_before = (last_loss, s0[1], s0[3])
            if self._verbose:
                logger.log("Epoch %d, loss %s" % (epoch, new_loss))
            if self._callback or callback:
                cb = dict(loss=new_loss, params=self._target.get_param_values(trainable=True),
                          itr=epoch, elapsed=time.time() - start_time)
                if self._callback:
                    self._callback(cb)
                if callback:
                    callback(**cb)
            if abs(last_loss - new_loss) < self._tolerance:
                break
            last_loss = new_loss
