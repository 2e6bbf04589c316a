"""NormalizedEnv / ``normalize`` (mirrors rllab/envs/normalized_env.py:11-103).

For a HIP-native wrapped env the action map ``lb + (a+1)/2*(ub-lb)`` + clip and
``scale_reward`` are fused into the step kernel (``normalize`` / ``scale_reward``
arguments of the C ABI); the numpy implementation below serves arbitrary Python
envs exactly like the reference.  Running obs/reward normalisation
(``normalize_obs`` / ``normalize_reward``, off by default) is a sequential EMA
over a single env's stream and is only available on the numpy path.
"""
import numpy as np

from rllab_amd import spaces
from rllab_amd.core.serializable import Serializable
from rllab_amd.envs.base import Step
from rllab_amd.envs.proxy_env import ProxyEnv
from rllab_amd.spaces.box import Box


class _RunningMoments(object):
    """Exponential moving estimates of mean and variance of a stream (the reference's
    _update_*_estimate, normalized_env.py:33-49)."""

    def __init__(self, shape, alpha):
        self.alpha = alpha
        self.mean = np.zeros(shape) if shape else 0.
        self.var = np.ones(shape) if shape else 1.

    def update(self, x):
        a = self.alpha
        self.mean = (1 - a) * self.mean + a * x
        self.var = (1 - a) * self.var + a * np.square(x - self.mean)

    @property
    def std(self):
        return np.sqrt(self.var) + 1e-8


def unit_to_bounds(action, lb, ub):
    """Map an action in [-1, 1]^n affinely onto [lb, ub] and clip (normalized_env.py:81-83)."""
    return np.clip(lb + (np.asarray(action) + 1.) * 0.5 * (ub - lb), lb, ub)


class NormalizedEnv(ProxyEnv, Serializable):
    def __init__(self, env, scale_reward=1., normalize_obs=False, normalize_reward=False,
                 obs_alpha=0.001, reward_alpha=0.001):
        Serializable.quick_init(self, locals())
        ProxyEnv.__init__(self, env)
        self._scale_reward = scale_reward
        self._normalize_obs, self._normalize_reward = normalize_obs, normalize_reward
        self._obs_stats = _RunningMoments((env.observation_space.flat_dim,), obs_alpha)
        self._reward_stats = _RunningMoments((), reward_alpha)

    # -- running normalisation (numpy path only) ---------------------------------------------------------
    def _whiten_obs(self, obs):
        self._obs_stats.update(self._wrapped_env.observation_space.flatten(obs))
        return (obs - self._obs_stats.mean) / self._obs_stats.std

    def _rescale_reward(self, reward):
        self._reward_stats.update(reward)
        return reward / self._reward_stats.std

    def __getstate__(self):
        d = Serializable.__getstate__(self)
        d["_obs_mean"], d["_obs_var"] = self._obs_stats.mean, self._obs_stats.var
        return d

    def __setstate__(self, d):
        Serializable.__setstate__(self, d)
        self._obs_stats.mean, self._obs_stats.var = d["_obs_mean"], d["_obs_var"]

    # -- Env interface -------------------------------------------------------------------------------------
    def _inner_is_box(self):
        return isinstance(self._wrapped_env.action_space, Box)

    @property
    def action_space(self):
        inner = self._wrapped_env.action_space
        if self._inner_is_box():
            ones = np.ones(inner.shape)
            return spaces.Box(-ones, ones)
        return inner

    def reset(self):
        obs = self._wrapped_env.reset()
        return self._whiten_obs(obs) if self._normalize_obs else obs

    def step(self, action):
        if self._inner_is_box():
            action = unit_to_bounds(action, *self._wrapped_env.action_space.bounds)
        next_obs, reward, done, info = self._wrapped_env.step(action)
        if self._normalize_obs:
            next_obs = self._whiten_obs(next_obs)
        if self._normalize_reward:
            reward = self._rescale_reward(reward)
        return Step(next_obs, reward * self._scale_reward, done, **info)

    def __str__(self):
        return "Normalized: %s" % self._wrapped_env

    # -- vectorised boundary: the affine action map and scale_reward are fused into the step kernels -------
    @property
    def vectorized(self):
        return bool(getattr(self._wrapped_env, "vectorized", False)) and \
            not (self._normalize_obs or self._normalize_reward)

    def vec_env_executor(self, n_envs, max_path_length, **kwargs):
        if not self.vectorized:
            raise NotImplementedError("NormalizedEnv: wrapped env is not vectorized (or running "
                                      "obs/reward normalisation is on)")
        return self._wrapped_env.vec_env_executor(
            n_envs=n_envs, max_path_length=max_path_length, normalize=True,
            scale_reward=float(self._scale_reward), **kwargs)


normalize = NormalizedEnv
