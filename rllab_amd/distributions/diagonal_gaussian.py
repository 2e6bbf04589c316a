"""Diagonal Gaussian (mirrors rllab/distributions/diagonal_gaussian.py:6-96).

The ``*_sym`` methods of the reference build Theano graphs; here they are the
same formulas on torch tensors (autograd plays the role of the symbolic graph).
The plain methods accept numpy arrays or torch tensors.  ``axis`` is the action
axis: -1 for the reference's ``[B, Da]`` layout, 0 for the engine's dense
``[Da, B]`` planes.
"""
import numpy as np
import torch

from rllab_amd.distributions.base import Distribution

_LOG_2PI = float(np.log(2 * np.pi))
_ENT_CONST = float(np.log(np.sqrt(2 * np.pi * np.e)))


def _t(x):
    return x if torch.is_tensor(x) else torch.as_tensor(np.asarray(x))


class DiagonalGaussian(Distribution):
    def __init__(self, dim):
        self._dim = dim

    @property
    def dim(self):
        return self._dim

    # -- torch ("symbolic") forms -------------------------------------------
    def kl_sym(self, old_dist_info_vars, new_dist_info_vars, axis=-1):
        old_means, old_log_stds = old_dist_info_vars["mean"], old_dist_info_vars["log_std"]
        new_means, new_log_stds = new_dist_info_vars["mean"], new_dist_info_vars["log_std"]
        old_std = torch.exp(old_log_stds)
        new_std = torch.exp(new_log_stds)
        # {(mu1 - mu2)^2 + s1^2 - s2^2} / (2 s2^2 + 1e-8) + ln(s2 / s1)   (reference :30-34)
        numerator = (old_means - new_means) ** 2 + old_std ** 2 - new_std ** 2
        denominator = 2 * new_std ** 2 + 1e-8
        return torch.sum(numerator / denominator + new_log_stds - old_log_stds, dim=axis)

    def log_likelihood_sym(self, x_var, dist_info_vars, axis=-1):
        means, log_stds = dist_info_vars["mean"], dist_info_vars["log_std"]
        zs = (x_var - means) / torch.exp(log_stds)
        dim = means.shape[axis]
        bl = torch.broadcast_to(log_stds, means.shape)
        return -torch.sum(bl, dim=axis) - 0.5 * torch.sum(zs ** 2, dim=axis) - 0.5 * dim * _LOG_2PI

    def likelihood_ratio_sym(self, x_var, old_dist_info_vars, new_dist_info_vars, axis=-1):
        logli_new = self.log_likelihood_sym(x_var, new_dist_info_vars, axis=axis)
        logli_old = self.log_likelihood_sym(x_var, old_dist_info_vars, axis=axis)
        return torch.exp(logli_new - logli_old)

    def entropy_sym(self, dist_info_var, axis=-1):
        return torch.sum(dist_info_var["log_std"] + _ENT_CONST, dim=axis)

    # -- numeric forms (numpy in -> numpy out, tensor in -> tensor out) -------
    def _wrap(self, fn, ref, *dicts_or_arrays):
        is_np = not torch.is_tensor(ref)
        out = fn()
        return out.cpu().numpy() if is_np else out

    def kl(self, old_dist_info, new_dist_info, axis=-1):
        o = {k: _t(v).to(torch.float64) for k, v in old_dist_info.items()}
        n = {k: _t(v).to(torch.float64) for k, v in new_dist_info.items()}
        return self._wrap(lambda: self.kl_sym(o, n, axis=axis), old_dist_info["mean"])

    def log_likelihood(self, xs, dist_info, axis=-1):
        d = {k: _t(v).to(torch.float64) for k, v in dist_info.items()}
        return self._wrap(lambda: self.log_likelihood_sym(_t(xs).to(torch.float64), d, axis=axis), xs)

    def entropy(self, dist_info, axis=-1):
        d = {"log_std": _t(dist_info["log_std"]).to(torch.float64)}
        return self._wrap(lambda: self.entropy_sym(d, axis=axis), dist_info["log_std"])

    def sample(self, dist_info):
        means, log_stds = np.asarray(dist_info["mean"]), np.asarray(dist_info["log_std"])
        rnd = np.random.normal(size=means.shape)
        return rnd * np.exp(log_stds) + means

    @property
    def dist_info_keys(self):
        return ["mean", "log_std"]
