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


class NormalizedEnv(ProxyEnv, Serializable):
    def __init__(self, env, scale_reward=1., normalize_obs=False, normalize_reward=False,
                 obs_alpha=0.001, reward_alpha=0.001):
        Serializable.quick_init(self, locals())
        ProxyEnv.__init__(self, env)
        self._scale_reward = scale_reward
        self._normalize_obs = normalize_obs
        self._normalize_reward = normalize_reward
        self._obs_alpha = obs_alpha
        self._obs_mean = np.zeros(env.observation_space.flat_dim)
        self._obs_var = np.ones(env.observation_space.flat_dim)
        self._reward_alpha = reward_alpha
        self._reward_mean = 0.
        self._reward_var = 1.

    def _update_obs_estimate(self, obs):
        flat_obs = self.wrapped_env.observation_space.flatten(obs)
        a = self._obs_alpha
        self._obs_mean = (1 - a) * self._obs_mean + a * flat_obs
        self._obs_var = (1 - a) * self._obs_var + a * np.square(flat_obs - self._obs_mean)

    def _update_reward_estimate(self, reward):
        a = self._reward_alpha
        self._reward_mean = (1 - a) * self._reward_mean + a * reward
        self._reward_var = (1 - a) * self._reward_var + a * np.square(reward - self._reward_mean)

    def _apply_normalize_obs(self, obs):
        self._update_obs_estimate(obs)
        return (obs - self._obs_mean) / (np.sqrt(self._obs_var) + 1e-8)

    def _apply_normalize_reward(self, reward):
        self._update_reward_estimate(reward)
        return reward / (np.sqrt(self._reward_var) + 1e-8)

    def reset(self):
        ret = self._wrapped_env.reset()
        return self._apply_normalize_obs(ret) if self._normalize_obs else ret

    def __getstate__(self):
        d = Serializable.__getstate__(self)
        d["_obs_mean"] = self._obs_mean
        d["_obs_var"] = self._obs_var
        return d

    def __setstate__(self, d):
        Serializable.__setstate__(self, d)
        self._obs_mean = d["_obs_mean"]
        self._obs_var = d["_obs_var"]

    @property
    def action_space(self):
        if isinstance(self._wrapped_env.action_space, Box):
            ub = np.ones(self._wrapped_env.action_space.shape)
            return spaces.Box(-1 * ub, ub)
        return self._wrapped_env.action_space

    def step(self, action):
        if isinstance(self._wrapped_env.action_space, Box):
            lb, ub = self._wrapped_env.action_space.bounds
            scaled_action = lb + (np.asarray(action) + 1.) * 0.5 * (ub - lb)
            scaled_action = np.clip(scaled_action, lb, ub)
        else:
            scaled_action = action
        next_obs, reward, done, info = self._wrapped_env.step(scaled_action)
        if self._normalize_obs:
            next_obs = self._apply_normalize_obs(next_obs)
        if self._normalize_reward:
            reward = self._apply_normalize_reward(reward)
        return Step(next_obs, reward * self._scale_reward, done, **info)

    def __str__(self):
        return "Normalized: %s" % self._wrapped_env

    # -- vectorised boundary --------------------------------------------------
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
