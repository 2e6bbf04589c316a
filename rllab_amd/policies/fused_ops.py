"""HIP-kernel implementations of the update-time functions of a GaussianMLPPolicy
(csrc/policy_kernels.hip through the C ABI): surrogate loss + mean KL, flat
gradient, Fisher-vector product.  Each result is a SUM of per-rank terms already
normalised by the global sample count, all-reduced here (RCCL) so callers see
global values.  Inputs are the tuples built by ``rllab_amd.algos.npo.npo_inputs``.
"""
import ctypes
import math

import torch

from rllab_amd import _lib
from rllab_amd.core.serializable import Serializable
from rllab_amd.sampler import dist as D


class FusedGaussianMLPOps(object):
    def __init__(self, policy):
        self.policy = policy
        hs = tuple(policy.hidden_sizes)
        assert len(hs) == 2 and policy.fusable
        self.dims = (policy.obs_dim, policy.action_dim, hs[0], hs[1])
        self._ws = None
        self._cache_key = None
        self._cache_val = None

    @staticmethod
    def supported(policy):
        hs = tuple(getattr(policy, "hidden_sizes", ()))
        return (getattr(policy, "fusable", False) and len(hs) == 2 and hs[0] == hs[1] and hs[0] in (32, 64)
                and policy.flat_params.is_cuda and policy.learn_std
                and (policy.obs_dim, policy.action_dim) in ((4, 1), (6, 1), (13, 2), (20, 6)))

    def accepts(self, inputs):
        """The kernels take ONE old log_std row (state-independent std)."""
        return inputs[4].numel() == self.dims[1] and inputs[0].is_cuda

    def _workspace(self, device):
        if self._ws is None or self._ws.device != device:
            n = _lib.lib.rl_policy_workspace_bytes(*self.dims)
            self._ws = torch.empty(n, dtype=torch.uint8, device=device)
        return self._ws

    def _batch(self, inputs, theta=None):
        obs, act, adv, old_mean, old_log_std, w, inv_count = inputs
        theta = self.policy.flat_params.detach() if theta is None else theta
        keep = [t.contiguous() for t in (obs, act, adv, old_mean, old_log_std.reshape(-1).float(), w, theta)]
        obs, act, adv, old_mean, old_ls, w, theta = keep
        pol = self.policy
        b = _lib.PolicyBatch(
            n_samples=obs.shape[-1], obs_dim=self.dims[0], act_dim=self.dims[1], hidden0=self.dims[2],
            hidden1=self.dims[3], inv_count=float(inv_count),
            log_min_std=math.log(pol.min_std) if pol.min_std is not None else -1e30,
            theta=theta.data_ptr(), obs=obs.data_ptr(), actions=act.data_ptr(), advantages=adv.data_ptr(),
            old_means=old_mean.data_ptr(), old_log_std=old_ls.data_ptr(), weights=w.data_ptr())
        return b, keep

    def loss_stats(self, inputs):
        """[sum w lr adv, sum w KL, sum w logp adv] * inv_count (global) and max KL, as a
        float64 device tensor of 4."""
        b, keep = self._batch(inputs)
        ws = self._workspace(keep[0].device)
        out = torch.empty(4, dtype=torch.float64, device=keep[0].device)
        _lib.check(_lib.lib.rl_policy_loss_kl(ctypes.byref(b), _lib.ptr(ws), ws.numel(), _lib.ptr(out),
                                              _lib.stream_ptr()), "rl_policy_loss_kl")
        sums = out[:3] * float(inputs[-1])
        D.all_reduce_sum_(sums)
        mx = D.all_reduce_max_(out[3:4].clone())
        return torch.cat([sums, mx])

    def loss_and_kl(self, inputs):
        s = self.loss_stats(inputs)
        return -s[0], s[1]

    def loss_grad(self, inputs, vpg=False):
        b, keep = self._batch(inputs)
        ws = self._workspace(keep[0].device)
        out = torch.empty(self.policy.flat_params.numel(), dtype=torch.float64, device=keep[0].device)
        _lib.check(_lib.lib.rl_policy_grad(ctypes.byref(b), int(vpg), _lib.ptr(ws), ws.numel(), _lib.ptr(out),
                                           _lib.stream_ptr()), "rl_policy_grad")
        return D.all_reduce_sum_(out)

    def fvp(self, inputs, vec):
        b, keep = self._batch(inputs)
        ws = self._workspace(keep[0].device)
        v = vec.to(torch.float32).contiguous()
        out = torch.empty(self.policy.flat_params.numel(), dtype=torch.float64, device=keep[0].device)
        _lib.check(_lib.lib.rl_policy_fvp(ctypes.byref(b), _lib.ptr(v), _lib.ptr(ws), ws.numel(), _lib.ptr(out),
                                          _lib.stream_ptr()), "rl_policy_fvp")
        return D.all_reduce_sum_(out)

    def hvp_approach(self):
        return FusedFisherHvp(self)


class FusedFisherHvp(Serializable):
    """HVP plug-in (interface of PerlmutterHvp, rllab/optimizers/
    conjugate_gradient_optimizer.py:13-55) backed by the fused Fisher-vector-product
    kernel.  Exact for the mean-KL Hessian at theta_new == theta_old, the only point
    where TRPO evaluates it."""

    def __init__(self, ops=None):
        self.ops = ops
        self.target = None
        self.reg_coeff = None

    def update_opt(self, f, target, inputs, reg_coeff):
        self.target, self.reg_coeff = target, reg_coeff

    def build_eval(self, inputs, trainable_index=None):
        ops, reg = self.ops, self.reg_coeff
        n = self.target.flat_params.numel()

        def eval(x):
            if trainable_index is None:
                full = x
            else:
                full = torch.zeros(n, dtype=x.dtype, device=x.device)
                full[trainable_index] = x
            hx = ops.fvp(inputs, full)
            if trainable_index is not None:
                hx = hx[trainable_index]
            return hx + reg * x
        return eval
