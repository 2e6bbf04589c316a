"""What the policy-gradient algorithms ask of an action distribution
(interface of rllab/distributions/base.py).  ``*_sym`` members take / return torch tensors and are
differentiable (the reference's Theano expressions); the others work on numpy arrays.  Distribution
parameters travel as ``dist_info`` dicts keyed by ``dist_info_keys``."""
import abc


class Distribution(abc.ABC):
    @property
    @abc.abstractmethod
    def dim(self):
        """Number of action dimensions."""

    @property
    @abc.abstractmethod
    def dist_info_keys(self):
        """Names of the per-sample parameters, e.g. ["mean", "log_std"]."""

    @abc.abstractmethod
    def kl_sym(self, old_dist_info_vars, new_dist_info_vars):
        """KL(old || new) per sample, differentiable in the new parameters."""

    @abc.abstractmethod
    def kl(self, old_dist_info, new_dist_info):
        """KL(old || new) per sample on numpy arrays."""

    @abc.abstractmethod
    def likelihood_ratio_sym(self, x_var, old_dist_info_vars, new_dist_info_vars):
        """p_new(x) / p_old(x) per sample."""

    @abc.abstractmethod
    def log_likelihood_sym(self, x_var, dist_info_vars):
        """log p(x) per sample, differentiable."""

    @abc.abstractmethod
    def log_likelihood(self, xs, dist_info):
        """log p(x) per sample on numpy arrays."""

    @abc.abstractmethod
    def entropy(self, dist_info):
        """Entropy per sample."""
