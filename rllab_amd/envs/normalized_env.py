"""NormalizedEnv / ``normalize`` (mirrors rllab/envs/normalized_env.py:11-103).

For a HIP-native wrapped env the action map ``lb + (a+1)/2*(ub-lb)`` + clip and
``scale_reward`` are fused into the step kernel (``normalize`` / ``scale_reward``
arguments of the C ABI); the numpy implementation below serves arbitrary Python
envs exactly like the reference.  Running obs/reward normalisation
(``normalize_obs`` / ``normalize_reward``, off by default) is an exponential moving
estimate over each env copy's own stream (the reference's VecEnvExecutor holds n
pickled copies of the env, each with its own estimates,
sandbox/rocky/tf/envs/vec_env_executor.py:8-14): on the vectorised path
``NormalizingVecEnv`` keeps those estimates as [., n] device planes around the HIP executor.
"""
import numpy as np
import torch

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

    def _pull_live_estimates(self):
        """A live vectorised executor owns the running estimates (one per env copy, on the device): before this env
        is pickled -- every snapshot pickles ``algo.env`` -- take env copy 0's, so a policy trained on whitened
        observations is replayed against the estimator it was trained with, not one restarting at mean 0 / var 1."""
        ref = getattr(self, "_live_vec", None)
        vec = ref() if ref is not None else None
        if vec is not None:
            vec.write_back(self)

    def __getstate__(self):
        self._pull_live_estimates()
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
        return bool(getattr(self._wrapped_env, "vectorized", False))

    def vec_env_executor(self, n_envs, max_path_length, **kwargs):
        if not self.vectorized:
            raise NotImplementedError("NormalizedEnv: wrapped env is not vectorized")
        if not (self._normalize_obs or self._normalize_reward):
            return self._wrapped_env.vec_env_executor(
                n_envs=n_envs, max_path_length=max_path_length, normalize=True,
                scale_reward=float(self._scale_reward), **kwargs)
        # running normalisation sits between the env's reward and scale_reward (:85-92): the kernel leaves the
        # reward unscaled and the wrapper applies normalisation, then the scale.  The inner executor does NOT
        # auto-reset: the wrapper has to see the terminal observation before the reset one (below).
        import weakref
        inner = self._wrapped_env.vec_env_executor(n_envs=n_envs, max_path_length=max_path_length, normalize=True,
                                                   scale_reward=1.0, auto_reset=False, **kwargs)
        vec = NormalizingVecEnv(inner, float(self._scale_reward), self._normalize_obs, self._normalize_reward,
                                self._obs_stats.alpha, self._reward_stats.alpha, owner=self)
        self._live_vec = weakref.ref(vec)
        return vec


class NormalizingVecEnv(object):
    """The VecEnvExecutor surface over a HIP executor with NormalizedEnv's running estimates per env copy:
        mean <- (1 - a) mean + a x;  var <- (1 - a) var + a (x - mean)^2     (normalized_env.py:33-49, float64)
        obs -> (obs - mean) / (sqrt(var) + 1e-8);   reward -> reward / (sqrt(var_r) + 1e-8), then * scale_reward.
    Update order of the reference's VecEnvExecutor over n NormalizedEnv copies (vec_env_executor.py:16-28): every
    copy's ``step`` feeds its estimate the observation it produced -- the TERMINAL one included -- and a copy that is
    done is then ``reset``, which feeds the estimate once more with the reset observation and returns that one
    whitened.  So the inner executor steps without auto-reset, the estimates see the step's observations of all
    copies, then the finished copies are reset under a mask and only their estimates see the reset observations.
    Estimates start from the owning NormalizedEnv's (a snapshot's) and env copy 0's are written back to it when it
    is pickled or the executor terminates.  The (32,32) / (64,64) policies are sampled by the fused rollout, which feeds,
    applies and writes back the same estimate planes inside the kernel (``rollout``, rl_running_norm); every other policy
    through ``reset`` / ``step`` below, one transition at a time."""
    graphable = False          # the sampler's hipGraph loop talks to the raw executor's buffers, not to step() below
    stateful_rollouts = True   # a rollout advances the estimates for good: the sampler never launches one speculatively

    def __init__(self, inner, scale_reward, normalize_obs, normalize_reward, obs_alpha, reward_alpha, owner=None):
        self.inner = inner
        self.scale_reward_outer = float(scale_reward)
        self.normalize_obs, self.normalize_reward = bool(normalize_obs), bool(normalize_reward)
        self.obs_alpha, self.reward_alpha = float(obs_alpha), float(reward_alpha)
        f64 = dict(dtype=torch.float64, device=inner.device)
        self.obs_mean = torch.zeros((inner.obs_rows, inner.n), **f64)
        self.obs_var = torch.ones((inner.obs_rows, inner.n), **f64)
        self.reward_mean = torch.zeros(inner.n, **f64)
        self.reward_var = torch.ones(inner.n, **f64)
        self._owner = owner
        if owner is not None:                         # resume from the owner's (snapshotted) estimates
            self.obs_mean += torch.as_tensor(np.asarray(owner._obs_stats.mean, dtype=np.float64), **f64).reshape(-1, 1)
            self.obs_var *= torch.as_tensor(np.asarray(owner._obs_stats.var, dtype=np.float64), **f64).reshape(-1, 1)
            self.reward_mean += float(owner._reward_stats.mean)
            self.reward_var *= float(owner._reward_stats.var)

    def __getattr__(self, name):                      # n, q, device, obs_rows, max_path_length, position_ids, ...
        if name == "inner":
            raise AttributeError(name)
        return getattr(self.inner, name)

    @property
    def num_envs(self):
        return self.inner.n

    def takes_rollout_of(self, policy):
        """The fused rollout under running normalisation: the generic kernels of the (32,32) / (64,64) policies carry the
        estimates in registers (rl_running_norm); everything else is sampled through reset() / step()."""
        if self.inner.position_ids is not None:
            return False
        return self.inner.rollout_plan(policy, norm=(self.normalize_obs, self.normalize_reward)) is not None

    def rollout_plan(self, policy, horizon=None):
        return self.inner.rollout_plan(policy, horizon, norm=(self.normalize_obs, self.normalize_reward))

    def rollout(self, policy, horizon, reset_at_start=True, **kwargs):
        """One launch for the whole horizon (HipVecEnv.rollout) with this wrapper's running estimates fed, applied and
        written back by the kernel in the reference's order: observations and rewards of the returned batch are the
        whitened ones, as ``step`` returns them.  A continuation (``reset_at_start=False``) starts from the whitened
        observation the previous launch left in the executor's buffer; no estimate is fed twice."""
        return self.inner.rollout(policy, horizon, reset_at_start=reset_at_start, norm=self,
                                  scale_reward=self.scale_reward_outer, **kwargs)

    def write_back(self, env):
        """Env copy 0's estimates -> the NormalizedEnv that gets pickled (its ``_obs_mean`` / ``_obs_var`` state)."""
        env._obs_stats.mean = self.obs_mean[:, 0].cpu().numpy().copy()
        env._obs_stats.var = self.obs_var[:, 0].cpu().numpy().copy()
        env._reward_stats.mean = float(self.reward_mean[0])
        env._reward_stats.var = float(self.reward_var[0])

    def _whiten(self, obs_n, only=None):
        """Feed the estimates ``obs_n`` [n, Do] and whiten it; ``only`` [n] bool: just those env copies move."""
        if not self.normalize_obs:
            return obs_n
        x = obs_n.t().to(torch.float64)
        a = self.obs_alpha
        mean = self.obs_mean * (1 - a) + a * x
        var = self.obs_var * (1 - a) + a * (x - mean) ** 2
        if only is not None:
            mean = torch.where(only, mean, self.obs_mean)
            var = torch.where(only, var, self.obs_var)
        self.obs_mean, self.obs_var = mean, var
        return ((x - mean) / (torch.sqrt(var) + 1e-8)).t().to(torch.float32)

    def reset(self, *args, **kwargs):
        return self._whiten(self.inner.reset(*args, **kwargs))

    def step(self, action_n, reset_draws=None, **kwargs):
        """``reset_draws`` [R, n]: injected draws of the resets this step triggers (parity runs)."""
        is_np = not torch.is_tensor(action_n)
        a = torch.as_tensor(np.asarray(action_n), device=self.inner.device) if is_np else action_n
        obs, rew, done, info = self.inner.step(a, **kwargs)      # no auto-reset: terminal observations
        obs = self._whiten(obs)
        done = done.clone()
        fresh = self._whiten(self.inner.reset(mask=done, draws=reset_draws), only=done)   # done copies only; a launch over a mask of
        obs = torch.where(done.unsqueeze(1), fresh, obs)                # zeros when nobody finished (no host sync)
        r = rew.to(torch.float64)
        if self.normalize_reward:
            al = self.reward_alpha
            self.reward_mean = self.reward_mean * (1 - al) + al * r
            self.reward_var = self.reward_var * (1 - al) + al * (r - self.reward_mean) ** 2
            r = r / (torch.sqrt(self.reward_var) + 1e-8)
        r = (r * self.scale_reward_outer).to(torch.float32)
        if is_np:
            return (obs.cpu().numpy().astype(np.float64), r.cpu().numpy().astype(np.float64), done.cpu().numpy(), info)
        return obs, r, done, info

    def terminate(self):
        if self._owner is not None:
            self.write_back(self._owner)
        self.inner.terminate()


normalize = NormalizedEnv
