"""HIP-kernel evaluation of a GaussianMLPRegressor's fit objective (csrc/policy_kernels.hip through the C ABI).

The regressor fits a diagonal Gaussian -- mean = MLP(hidden layers), one free log_std row -- by minimising the
negative log-likelihood of whitened targets, optionally under a mean-KL trust region to its previous prediction
(rllab/regressors/gaussian_mlp_regressor.py:107-143).  With ``observations -> whitened inputs``, ``actions ->
whitened targets``, ``advantages -> 1`` that is exactly the policy kernels' log-likelihood pass:

    rl_policy_grad_loss(vpg = 1, kl_penalty = p)   ->   d/dtheta [ -sum w logp + p sum w KL(old || new) ] / W
                                                         and the sums  sum w logp,  sum w KL

so one launch gives L-BFGS its value and gradient (optimizers/lbfgs_optimizer.py, penalty_lbfgs_optimizer.py).
The hidden nonlinearity is the regressor's: rectify (the reference default, RL_ACT_RECTIFY) or tanh.  Results are
sums over env shards, all-reduced here.
"""
import ctypes

import numpy as np
import torch

from rllab_amd import _lib
from rllab_amd.core.network import rectify, tanh
from rllab_amd.sampler import dist as D

_INPUT_DIMS = (4, 6, 11, 13, 20, 21)


class FusedRegressorOps(object):
    def __init__(self, regressor):
        self.reg = regressor
        net = regressor._mean_network
        self.dims = (net.input_dim, net.output_dim, net.hidden_sizes[0], net.hidden_sizes[1], 0)
        self.activation = _lib.ACT_RECTIFY if net.hidden_nonlinearity is rectify else _lib.ACT_TANH
        self._ws = None
        self._bound = None

    @staticmethod
    def supported(regressor):
        net = regressor._mean_network
        return (regressor.flat_params.is_cuda and regressor.flat_params.dtype == torch.float32
                and net.output_dim == 1 and tuple(net.hidden_sizes) == (32, 32) and net.input_dim in _INPUT_DIMS
                and (net.hidden_nonlinearity is rectify or net.hidden_nonlinearity is tanh)
                and net.output_nonlinearity is None and regressor._log_std_param.tags["trainable"])

    def accepts(self, inputs):
        return len(inputs) == 6 and torch.is_tensor(inputs[0]) and inputs[0].is_cuda

    # inputs = (xs [Din,B], ys [1,B], old_means [1,B], old_log_stds [1,B], w [B], inv) in OUTPUT units, as the
    # regressor's closures take them; the kernels see whitened planes, built once per fit
    def bind(self, inputs):
        key = tuple(id(t) for t in inputs) + tuple(t.data_ptr() for t in (self.reg._x_mean, self.reg._y_mean))
        if self._bound is not None and self._bound[0] == key:
            return self._bound[1]
        xs, ys, old_means, old_log_stds, w, inv = inputs
        r = self.reg
        f32 = torch.float32
        nx = ((xs - r._x_mean) / r._x_std).to(f32).contiguous()
        ny = ((ys - r._y_mean) / r._y_std).to(f32).contiguous()
        om = ((old_means - r._y_mean) / r._y_std).to(f32).contiguous()
        ols = (old_log_stds[:, :1] - torch.log(r._y_std)).reshape(-1).to(f32).contiguous()   # one constant row
        ones = torch.ones(nx.shape[-1], dtype=f32, device=nx.device)
        wv = w.to(f32).contiguous()
        theta = r.flat_params.detach()
        assert theta.is_contiguous()
        b = _lib.PolicyBatch(
            n_samples=nx.shape[-1], obs_dim=self.dims[0], act_dim=self.dims[1], hidden0=self.dims[2],
            hidden1=self.dims[3], inv_count=float(inv), log_min_std=-1e30, theta=theta.data_ptr(), obs=nx.data_ptr(),
            actions=ny.data_ptr(), advantages=ones.data_ptr(), old_means=om.data_ptr(), old_log_std=ols.data_ptr(),
            weights=wv.data_ptr(), activations=None, kl_penalty=0.0, activation=self.activation)
        if self._ws is None or self._ws.device != nx.device:
            n = _lib.lib.rl_policy_workspace_bytes(*self.dims)
            self._ws = torch.empty(n, dtype=torch.uint8, device=nx.device)
        bound = (b, (nx, ny, om, ols, ones, wv, theta) + tuple(inputs), float(inv))
        self._bound = (key, bound)
        return bound

    def release(self):
        self._bound = None

    def _sums(self, out4, inv):
        h = D.all_reduce_sum_(out4[:3].clone()).cpu().numpy() * inv      # [lr-sum (unused), KL, logp] / W
        return -float(h[2]), float(h[1])                                  # (negative log-likelihood, mean KL)

    def loss_and_kl(self, inputs):
        """(mean negative log-likelihood, mean KL(old || new)) at the regressor's current parameters."""
        b, keep, inv = self.bind(inputs)
        out4 = torch.empty(4, dtype=torch.float64, device=keep[0].device)
        b.kl_penalty = 0.0
        _lib.check(_lib.lib.rl_policy_loss_kl(ctypes.byref(b), _lib.ptr(self._ws), self._ws.numel(), _lib.ptr(out4),
                                              _lib.stream_ptr()), "rl_policy_loss_kl")
        return self._sums(out4, inv)

    def value_and_grad(self, inputs, penalty=0.0):
        """float64 (value, flat gradient) of  NLL + penalty * mean KL  -- what scipy's L-BFGS is handed."""
        b, keep, inv = self.bind(inputs)
        dev = keep[0].device
        grad = torch.empty(self.reg.flat_params.numel(), dtype=torch.float64, device=dev)
        out4 = torch.empty(4, dtype=torch.float64, device=dev)
        b.kl_penalty = float(penalty)
        try:
            _lib.check(_lib.lib.rl_policy_grad_loss(ctypes.byref(b), 1, _lib.ptr(self._ws), self._ws.numel(),
                                                    _lib.ptr(grad), _lib.ptr(out4), _lib.stream_ptr()),
                       "rl_policy_grad_loss")
        finally:
            b.kl_penalty = 0.0
        packed = torch.cat([out4[:3], grad])
        D.all_reduce_sum_(packed)
        host = packed.cpu().numpy()
        nll, kl = -host[2] * inv, host[1] * inv
        g = host[3:].copy()
        idx = self.reg._flat_index(trainable=True)
        if idx is not None:
            g = g[idx.cpu().numpy()]
        return float(nll + penalty * kl), g
