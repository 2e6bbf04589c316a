"""HIP-native envs: the Python face of the env kernels.

``HipVecEnv`` is the vectorised boundary of sandbox/rocky/tf/envs/
vec_env_executor.py:8-48 (``reset() -> obs_n``, ``step(action_n) -> (obs_n,
rewards, dones, env_infos)``, ``num_envs``, ``terminate()``) with every per-env
Python loop replaced by one kernel launch over SoA state planes in HBM.
``HipEnv`` is the single-env ``Env`` face (reset/step on a 1-env executor) that
keeps scripts written against rllab/envs/base.py working.
"""
import ctypes
import math

import numpy as np
import torch

from rllab_amd import _lib, spaces
from rllab_amd.core.serializable import Serializable
from rllab_amd.envs.base import Env, Step
from rllab_amd.misc import ext
from rllab_amd.sampler.trajectories import Trajectories

BIG = 1e6


def _require_device():
    if not torch.cuda.is_available():
        raise RuntimeError("rllab_amd: no HIP device visible -- env kernels have no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _fresh_seed():
    s = ext.get_seed()
    if s is None:
        s = int(np.random.randint(0, 2 ** 31 - 1))
    return int(s)


class HipVecEnv(object):
    """n lock-step copies of one env kind on the current HIP device."""

    def __init__(self, kind, n_envs, max_path_length, normalize=False, scale_reward=1.0, seed=None,
                 env_offset=0, action_space=None, observation_space=None, auto_reset=True, cfg=None,
                 position_ids=None):
        """``cfg``: dict of env options (fields of ``rl_env_cfg``: ctrl_cost_coeff, alive_coeff, action_noise,
        obs_noise, frame_skip, flags) on top of the env's defaults.  ``position_ids``: Box2DEnv(position_only=True)
        -- the observation rows callers see (box2d_env.py:185-192,228-230); the kernels always produce the full
        observation, the filter is a row selection of their output."""
        self.kind = kind
        self.cfg_overrides = dict(cfg or {})
        self.cfg = _lib.env_default_cfg(kind, **self.cfg_overrides)
        self.position_ids = None if position_ids is None else [int(i) for i in position_ids]
        self.n = int(n_envs)
        self.max_path_length = int(max_path_length) if max_path_length is not None else 0
        self.normalize = bool(normalize)
        self.scale_reward = float(scale_reward)
        self.seed = _fresh_seed() if seed is None else int(seed)
        self.env_offset = int(env_offset)
        self.auto_reset = bool(auto_reset)
        self.q = _lib.env_query(kind)
        # can a path end before max_path_length?  (rl_env_terminates: the sampler counts finished samples only then)
        self.terminates = bool(self.q["terminates"])
        self.device = _require_device()
        self.state = torch.zeros((self.q["state_dim"], self.n), dtype=torch.float32, device=self.device)
        self.ts = torch.zeros((self.n,), dtype=torch.int32, device=self.device)
        self._obs = torch.zeros((self.q["obs_dim"], self.n), dtype=torch.float32, device=self.device)
        self._reward = torch.zeros((self.n,), dtype=torch.float32, device=self.device)
        self._done = torch.zeros((self.n,), dtype=torch.uint8, device=self.device)
        self.step_counter = 0  # global step index: RNG counter base
        self._action_space, self._observation_space = action_space, observation_space
        self._pos_index = None if self.position_ids is None else torch.as_tensor(self.position_ids, device=self.device)

    def __getstate__(self):
        d = dict(self.__dict__)
        d["cfg"] = None                    # ctypes structs with pointers do not pickle: rebuilt from the overrides
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)
        self.cfg = _lib.env_default_cfg(self.kind, **self.cfg_overrides)

    def _cfg_with(self, action_noise_z=None, obs_noise_z=None):
        """The launch's rl_env_cfg: the env's options plus (parity runs) injected noise draws."""
        c = self.cfg
        c.action_noise_z = action_noise_z.data_ptr() if action_noise_z is not None else None
        c.obs_noise_z = obs_noise_z.data_ptr() if obs_noise_z is not None else None
        return ctypes.byref(c)

    def _filtered(self, obs_rows):
        """[Do, n] observation planes -> what the caller sees ([n, Do'] view; position rows only when filtering)."""
        if self._pos_index is not None:
            obs_rows = obs_rows.index_select(0, self._pos_index)
        return obs_rows.t()

    def _plane(self, z, shape):
        if z is None:
            return None
        z = torch.as_tensor(z, dtype=torch.float32, device=self.device).contiguous()
        assert tuple(z.shape) == tuple(shape), (tuple(z.shape), tuple(shape))
        return z

    # -- VecEnvExecutor surface ------------------------------------------------
    @property
    def num_envs(self):
        return self.n

    @property
    def obs_rows(self):
        """Observation dimension callers see (the position rows only under position_only)."""
        return self.q["obs_dim"] if self.position_ids is None else len(self.position_ids)

    @property
    def action_space(self):
        return self._action_space

    @property
    def observation_space(self):
        return self._observation_space

    def terminate(self):
        pass

    def reset(self, mask=None, draws=None, obs_noise_z=None):
        """Reset all envs (or those in ``mask``); returns obs_n as an [n, Do] view.  ``obs_noise_z`` [Do, n]:
        injected N(0,1) draws of the observation noise (parity runs; otherwise the in-kernel Philox stream)."""
        if draws is not None:
            draws = torch.as_tensor(draws, dtype=torch.float32, device=self.device).contiguous()
            assert draws.shape == (self.q["reset_draws"], self.n)
        if mask is not None:
            mask = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
        oz = self._plane(obs_noise_z, (self.q["obs_dim"], self.n))
        _lib.check(_lib.lib.rl_vecenv_reset(
            self.kind, self.n, _lib.ptr(self.state), _lib.ptr(self.ts), _lib.ptr(mask), _lib.ptr(draws),
            self.seed, self.step_counter, self.env_offset, self._cfg_with(None, oz), _lib.ptr(self._obs),
            _lib.stream_ptr()), "rl_vecenv_reset")
        self.step_counter += 1
        return self._filtered(self._obs)

    def step(self, action_n, reset_draws=None, action_noise_z=None, obs_noise_z=None):
        """One lock-step transition.  ``action_n``: [n, Da] numpy array or tensor.
        numpy in -> numpy out; tensor in -> device tensors out (views of buffers
        that the next call overwrites).  ``action_noise_z`` [Da, n] / ``obs_noise_z`` [Do, n]: injected
        N(0,1) draws of the env's own noises (parity runs)."""
        is_np = not torch.is_tensor(action_n)
        a = torch.as_tensor(np.asarray(action_n) if is_np else action_n)
        a = a.to(device=self.device, dtype=torch.float32).reshape(self.n, self.q["act_dim"]).t().contiguous()
        if reset_draws is not None:
            reset_draws = torch.as_tensor(reset_draws, dtype=torch.float32, device=self.device).contiguous()
        az = self._plane(action_noise_z, (self.q["act_dim"], self.n))
        oz = self._plane(obs_noise_z, (self.q["obs_dim"], self.n))
        _lib.check(_lib.lib.rl_vecenv_step(
            self.kind, self.n, int(self.normalize), self.scale_reward, self.max_path_length,
            int(self.auto_reset), _lib.ptr(self.state), _lib.ptr(self.ts), _lib.ptr(a), _lib.ptr(reset_draws), self.seed,
            self.step_counter, self.env_offset, self._cfg_with(az, oz), _lib.ptr(self._obs), _lib.ptr(self._reward),
            _lib.ptr(self._done), _lib.stream_ptr()), "rl_vecenv_step")
        self.step_counter += 1
        obs, rew, done = self._filtered(self._obs), self._reward, self._done.bool()
        if is_np:
            return (obs.cpu().numpy().astype(np.float64), rew.cpu().numpy().astype(np.float64),
                    done.cpu().numpy(), dict())
        return obs, rew, done, dict()

    # -- state access (single source of truth: the SoA planes on the device) ------
    def observe(self):
        """obs_n of the state planes as they are (no transition): rl_vecenv_observe."""
        _lib.check(_lib.lib.rl_vecenv_observe(self.kind, self.n, _lib.ptr(self.state), _lib.ptr(self._obs),
                                              _lib.stream_ptr()), "rl_vecenv_observe")
        return self._filtered(self._obs)

    def com(self):
        """[n, 4] = (forward, up) position and velocity of the torso subtree's centre of mass of every env
        (MujocoEnv.get_body_com / get_body_comvel, mujoco_env.py:232-238): rl_vecenv_com."""
        out = torch.empty((4, self.n), dtype=torch.float32, device=self.device)
        _lib.check(_lib.lib.rl_vecenv_com(self.kind, self.n, _lib.ptr(self.state), _lib.ptr(out), _lib.stream_ptr()),
                   "rl_vecenv_com")
        return out.t()

    def get_state(self):
        """[n, S] host copy of the persisted state (layout per env kind: csrc/dyn_*.h)."""
        return self.state.t().cpu().numpy().astype(np.float64)

    def set_state(self, state):
        """Overwrite the persisted state from an [n, S] (or [S] for one env) array; steps since reset -> 0."""
        st = torch.as_tensor(np.asarray(state, dtype=np.float32).reshape(self.n, self.q["state_dim"]))
        self.state.copy_(st.t().contiguous())
        self.ts.zero_()
        self.observe()          # a rollout that carries on (reset_at_start=False) starts from this buffer

    # -- fused rollout ---------------------------------------------------------
    def takes_rollout_of(self, policy):
        """True when ``rollout(policy, ..)`` has a kernel for this policy ON THIS ENV: the policy offers a kernel
        layout (or its two networks) and the weight fragments of a wide / deep / dual-network shape fit the LDS of a CU
        next to the env's observation tile (rl_rollout_lds_bytes) -- e.g. a (128,128) mean net + a (128,128) log-std
        net on a 20-observation env needs 172 KB and is sampled through the per-transition loop instead."""
        return self.rollout_plan(policy) is not None

    def rollout_plan(self, policy, horizon=None, norm=None):
        """``_lib.RolloutPlan`` -- kernel, envs per wavefront, wavefronts, workgroup shape, LDS -- of ``rollout(policy, ..)``
        on this executor under the current launch options, from the launcher itself (rl_rollout_plan_query), or None
        when there is no fused kernel for the policy here."""
        layout = policy.kernel_layout() if hasattr(policy, "kernel_layout") else None
        dual = policy.rollout_networks() if (layout is None and hasattr(policy, "rollout_networks")) else None
        if layout is None and dual is None:
            return None
        T = int(horizon if horizon is not None else max(1, self.max_path_length))
        flags = int(self.cfg.flags)
        hs, hs_std = (layout.hidden3, (0, 0, 0)) if layout is not None else (tuple(dual[1]), tuple(dual[3]))
        # asked once per (sizes, launch options): the sampler asks before every rollout
        _lib.launch_opts()
        acts = layout.layer_activations if layout is not None else int(dual[4])
        acts_std = 0 if layout is not None else int(dual[5])
        key = (hs, hs_std, T, flags, acts, acts_std, norm, float(self.cfg.obs_noise), bytes(_lib._OPTS))
        cache = self.__dict__.setdefault("_plan_cache", {})
        if key not in cache:
            if len(cache) > 64:
                cache.clear()
            cache[key] = _lib.rollout_plan(self.kind, self.n, T, hs, hs_std, cfg_flags=flags, layer_activations=acts,
                                           norm=norm, obs_noise=float(self.cfg.obs_noise), std_layer_activations=acts_std)
        return cache[key]

    def _first_layer_on_full_obs(self, theta, h0):
        """Kernel-layout parameters of a net built on the position rows -> the same net on the full observation:
        W0 [kept, h0] scattered into [obs_dim, h0] (flat order W0, b0, ...: policies/kernel_layout.py), the rest as is."""
        kept, do = len(self.position_ids), self.q["obs_dim"]
        w0 = torch.zeros((do, h0), dtype=torch.float32, device=self.device)
        w0.index_copy_(0, self._pos_index, theta[:kept * h0].view(kept, h0))
        return torch.cat([w0.view(-1), theta[kept * h0:]])

    def rollout(self, policy, horizon, reset_at_start=True, eps=None, reset_draws=None, action_noise_z=None,
                obs_noise_z=None, norm=None, scale_reward=None):
        """``horizon`` lock-step iterations of get_actions -> step -> record ->
        auto-reset in ONE launch (rl_rollout_gaussian_mlp).  Returns
        ``Trajectories``.  ``eps`` [Da, T, n] / ``reset_draws`` [T+1, R, n] / ``action_noise_z`` [T, Da, n] /
        ``obs_noise_z`` [T+1, Do, n] inject pre-generated noise (parity runs).  ``reset_at_start=False``: the envs carry
        on from their state, step count and the observation the previous launch / ``step`` / ``reset`` ended on."""
        if self.position_ids is not None and norm is not None:
            raise NotImplementedError("position_only observations under running normalisation: the estimates are over "
                                      "the kept rows; sample through the per-transition path")
        T, n = int(horizon), self.n
        do, da = self.q["obs_dim"], self.q["act_dim"]
        layout = policy.kernel_layout() if hasattr(policy, "kernel_layout") else None
        dual = policy.rollout_networks() if (layout is None and hasattr(policy, "rollout_networks")) else None
        if layout is None and dual is None:
            raise NotImplementedError("fused rollout needs a GaussianMLPPolicy with two or three tanh hidden layers of "
                                      "at most 128 units (policies/kernel_layout.py)")
        hs = layout.hidden3 if dual is None else dual[1]
        dev = self.device
        f32 = dict(dtype=torch.float32, device=dev)
        obs = torch.empty((do, T, n), **f32)
        act = torch.empty((da, T, n), **f32)
        mean = torch.empty((da, T, n), **f32)
        rew = torch.empty((T, n), **f32)
        done = torch.empty((T, n), dtype=torch.uint8, device=dev)
        # the parameters in the kernels' layout (zero-padded hidden units); with a log-std network: the two networks
        theta = layout.theta() if dual is None else dual[0]
        theta_std = None if dual is None else dual[2]
        hs_std = (0, 0, 0) if dual is None else dual[3]
        log_stds = None if dual is None else torch.empty((da, T, n), **f32)
        if self.position_ids is not None:
            # Box2DEnv(position_only=True): the env kernel produces (and noises) the full observation and the policy was
            # built on the kept rows (box2d_env.py:219-227).  The kernel copy of the first layer gets zero rows at the
            # dropped observations -- products with 0, the kept rows' sums unchanged -- and the batch keeps the kept rows.
            theta = self._first_layer_on_full_obs(theta, hs[0])
            if theta_std is not None:
                theta_std = self._first_layer_on_full_obs(theta_std, hs_std[0])
        assert theta.is_cuda and theta.dtype == torch.float32 and theta.is_contiguous()
        if eps is not None:
            eps = torch.as_tensor(eps, **f32).contiguous()
            assert eps.shape == (da, T, n)
        if reset_draws is not None:
            reset_draws = torch.as_tensor(reset_draws, **f32).contiguous()
            assert reset_draws.shape == (T + 1, self.q["reset_draws"], n)
        az = self._plane(action_noise_z, (T, da, n))
        oz = self._plane(obs_noise_z, (T + 1, do, n))
        log_min_std = math.log(policy.min_std) if policy.min_std is not None else -1e30
        self._cfg_with(az, oz)
        args = _lib.RolloutArgs(
            kind=self.kind, n_envs=n, horizon=T, max_path_length=self.max_path_length,
            normalize=int(self.normalize), reset_at_start=int(reset_at_start),
            hidden0=hs[0], hidden1=hs[1], hidden2=hs[2], env_offset=self.env_offset,
            scale_reward=self.scale_reward, log_min_std=log_min_std, seed=self.seed,
            step_counter=self.step_counter,
            state=self.state.data_ptr(), ts=self.ts.data_ptr(), theta=theta.data_ptr(),
            eps=eps.data_ptr() if eps is not None else None,
            reset_draws=reset_draws.data_ptr() if reset_draws is not None else None,
            obs=obs.data_ptr(), actions=act.data_ptr(), means=mean.data_ptr(),
            rewards=rew.data_ptr(), dones=done.data_ptr(), last_obs=self._obs.data_ptr(),
            cfg=ctypes.pointer(self.cfg),
            theta_std=None if theta_std is None else theta_std.data_ptr(),
            log_stds=None if log_stds is None else log_stds.data_ptr(),
            std_hidden0=hs_std[0], std_hidden1=hs_std[1], std_hidden2=hs_std[2],
            layer_activations=layout.layer_activations if layout is not None else int(dual[4]),
            std_layer_activations=0 if dual is None else int(dual[5]), opts=_lib.launch_opts())
        if scale_reward is not None:          # NormalizingVecEnv: its outer scale (the inner executor's is 1)
            args.scale_reward = float(scale_reward)
        nrm = None
        if norm is not None:
            # NormalizedEnv(normalize_obs / normalize_reward): the wrapper's per-env running estimates (float64 device
            # planes), fed, applied and written back by the kernel in the reference's order (rl_running_norm)
            nrm = _lib.RunningNorm(
                obs_mean=norm.obs_mean.data_ptr(), obs_var=norm.obs_var.data_ptr(),
                reward_mean=norm.reward_mean.data_ptr(), reward_var=norm.reward_var.data_ptr(),
                obs_alpha=norm.obs_alpha, reward_alpha=norm.reward_alpha,
                normalize_obs=int(norm.normalize_obs), normalize_reward=int(norm.normalize_reward))
            args.norm = ctypes.addressof(nrm)
        _lib.check(_lib.lib.rl_rollout_gaussian_mlp(ctypes.byref(args), _lib.stream_ptr()),
                   "rl_rollout_gaussian_mlp")
        self.step_counter += T + 1
        if self._pos_index is not None:
            obs = obs.index_select(0, self._pos_index)
        return Trajectories(obs, act, mean, policy.recorded_log_std(), rew, done,
                            self.max_path_length, log_std_planes=log_stds)


class HipEnv(Env, Serializable):
    """Base of the HIP-native envs; subclasses set ``KIND``."""
    KIND = None

    def __init__(self, cfg=None, position_ids=None):
        """``cfg``: env options for the kernels (dict of rl_env_cfg fields; validated by the library on first use);
        ``position_ids``: observation rows kept by Box2DEnv(position_only=True)."""
        q = _lib.env_query(self.KIND)
        self._q = q
        self._cfg = dict(cfg or {})
        _lib.env_default_cfg(self.KIND, **self._cfg)      # unknown option names fail here, at construction
        self._position_ids = None if position_ids is None else list(position_ids)
        lb, ub = _lib.env_action_bounds(self.KIND)
        self._action_space = spaces.Box(lb, ub)
        ob = BIG * np.ones(q["obs_dim"] if self._position_ids is None else len(self._position_ids))
        self._observation_space = spaces.Box(-ob, ob)
        self._single = None

    @property
    def action_space(self):
        return self._action_space

    @property
    def observation_space(self):
        return self._observation_space

    @property
    def action_bounds(self):
        return self.action_space.bounds

    @property
    def vectorized(self):
        return True

    def vec_env_executor(self, n_envs, max_path_length, normalize=False, scale_reward=1.0, seed=None,
                         env_offset=0, auto_reset=True):
        return HipVecEnv(self.KIND, n_envs, max_path_length, normalize=normalize,
                         scale_reward=scale_reward, seed=seed, env_offset=env_offset,
                         action_space=self.action_space, observation_space=self.observation_space,
                         auto_reset=auto_reset, cfg=self._cfg, position_ids=self._position_ids)

    # -- single-env face (rllab/envs/base.py) on a 1-env executor --------------
    def _one(self):
        if self._single is None:
            self._single = self.vec_env_executor(1, 0, auto_reset=False)
        return self._single

    def reset(self):
        return self._one().reset()[0].cpu().numpy().astype(np.float64)

    def step(self, action):
        v = self._one()
        a = np.asarray(action, dtype=np.float64).reshape(1, -1)
        obs, rew, done, _ = v.step(a)  # auto_reset off: terminal observation, caller resets
        return Step(observation=obs[0], reward=float(rew[0]), done=bool(done[0]))

    def get_current_obs(self):
        """Observation of the single env's present state (box2d_env.py:210-218, mujoco_env.py:118-131)."""
        return self._one().observe()[0].cpu().numpy().astype(np.float64)

    def get_state(self):
        """The persisted state vector of the single env (kernel layout of this env kind, csrc/dyn_*.h)."""
        return self._one().get_state()[0]

    def set_state(self, state):
        self._one().set_state(np.asarray(state).reshape(1, -1))

    def terminate(self):
        self._single = None
