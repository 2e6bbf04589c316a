"""MujocoEnv base (API of rllab/envs/mujoco/mujoco_env.py:36-238).

The reference loads an MJCF into the proprietary MuJoCo 1.31 binary through ctypes
(rllab/mujoco_py); importing this module needs no MuJoCo: each task's model is
compiled into a HIP kernel (csrc/dyn_planar.h + dyn_<task>.h).  ``action_noise`` and the
tasks' reward coefficients are run-time options of the kernels (``rl_env_cfg``); options
that name a different model are rejected loudly.
"""
import numpy as np

from rllab_amd import _lib

from rllab_amd.envs.hip_env import HipEnv


class MujocoEnv(HipEnv):
    FILE = None
    # observation component whose change over a path is the "ForwardProgress" diagnostic
    # (obs[-3] = x of the subtree COM: swimmer_env.py:48-62, half_cheetah_env.py:48-56)
    progress_obs_index = -3

    def __init__(self, action_noise=0.0, file_path=None, template_args=None, **engine_cfg):
        """``action_noise``: ctrl = action + 0.5 (ub - lb) * action_noise * N(0,1) (mujoco_env.py:175-187), drawn by
        the kernels (Philox stream; ``np.random`` in the reference).  ``file_path`` / ``template_args`` would name a
        different MJCF: only the task's own model is compiled into a kernel."""
        if file_path is not None or template_args is not None:
            raise NotImplementedError(
                "%s: file_path / template_args describe a different model; only the task's own MJCF constants are "
                "compiled into the HIP kernel of this env" % type(self).__name__)
        if action_noise < 0:
            raise ValueError("action_noise must be >= 0")
        self.action_noise = float(action_noise)
        HipEnv.__init__(self, cfg=dict(engine_cfg, action_noise=self.action_noise))

    @staticmethod
    def _constraint_flags(limit_model, contact_model, kwargs):
        """Engine options of the legged envs (the reference delegates both to MuJoCo 1.31): "penalty" = the spring-damper
        models of csrc/dyn_two_legs.h / dyn_legged.h (default; the one-body-per-lane rollout kernels), "mujoco" = MuJoCo's
        documented soft-constraint model with the MJCF's own solref / solimp / friction (csrc/dyn_mjc.h: projected
        Gauss-Seidel over the active limit and pyramidal contact rows), on the env-per-lane kernels."""
        for name, val in (("limit_model", limit_model), ("contact_model", contact_model)):
            if val not in ("penalty", "mujoco"):
                raise ValueError("%s=%r: 'penalty' or 'mujoco'" % (name, val))
        flags = int(kwargs.get("flags", 0))
        if limit_model == "mujoco":
            flags |= _lib.CFG_LIMIT_MUJOCO
        if contact_model == "mujoco":
            flags |= _lib.CFG_CONTACT_MUJOCO
        return dict(kwargs, flags=flags) if flags else kwargs

    # -- state-level API of the reference base class (mujoco_env.py:109-238) --------------------------------------
    @property
    def _nq(self):
        return self._q["state_dim"] // 2          # state = [qpos, qvel] for the planar MuJoCo-style envs

    def reset_mujoco(self, init_state=None):
        """qpos / qvel <- init + N(0, 0.01) / N(0, 0.1) drawn from np.random like the reference (:109-116), or the
        leading [qpos, qvel] of ``init_state`` (qacc / ctrl entries, if present, carry no state here)."""
        nq = self._nq
        if init_state is None:
            qpos = np.random.normal(size=nq) * 0.01
            qvel = np.random.normal(size=nq) * 0.1
            state = np.concatenate([qpos, qvel])
        else:
            state = np.asarray(init_state, dtype=np.float64).reshape(-1)[:2 * nq]
        self.set_state(state)

    def inject_action_noise(self, action):
        """action + 0.5 (ub - lb) * action_noise * N(0, 1); the draw happens even at scale 0, as in the reference
        (:175-185), so scripts that share np.random with the env see the same stream."""
        action = np.asarray(action, dtype=np.float64)
        noise = self.action_noise * np.random.normal(size=action.shape)
        lb, ub = self.action_bounds
        return action + 0.5 * (ub - lb) * noise

    def _torso_com(self, body_name):
        if body_name != "torso":
            raise NotImplementedError("%s: the kernels export the torso subtree (= whole model) only"
                                      % type(self).__name__)
        return self._one().com()[0].cpu().numpy().astype(np.float64)   # forward, up, d/dt forward, d/dt up

    def get_body_com(self, body_name):
        """Subtree centre of mass of the torso, ``model.data.com_subtree[idx]`` (mujoco_env.py:232-234): (x, y, z)
        with the planar model's forward / up coordinates in the slots its observation uses."""
        c = self._torso_com(body_name)
        return self._planar_to_xyz(c[0], c[1])

    def get_body_comvel(self, body_name):
        """Subtree linear momentum / subtree mass (mujoco_env.py:236-238, mjcore.py:58-81) -- the quantity whose
        forward component is the reward's ``comvel_x``."""
        c = self._torso_com(body_name)
        return self._planar_to_xyz(c[2], c[3])

    # the swimmer moves in the x-y plane (z = 0); the legged models in x-z (y = 0)
    PLANE = "xz"

    def _planar_to_xyz(self, forward, up):
        return np.array([forward, up, 0.0]) if self.PLANE == "xy" else np.array([forward, 0.0, up])

    def _log_forward_progress(self, paths):
        """Average/Max/Min/StdForwardProgress = obs[-1][-3] - obs[0][-3] per path
        (swimmer_env.py:48-62, half_cheetah_env.py:48-56), computed on the device for
        a dense batch."""
        import numpy as np
        import torch
        from rllab_amd.misc import logger
        from rllab_amd.sampler import dist as D
        from rllab_amd.sampler.trajectories import PathList
        if isinstance(paths, PathList) and paths.traj.progress_stats is not None:
            vals = paths.traj.progress_stats      # computed by rl_sample_stats inside process_samples
        elif isinstance(paths, PathList):
            env, t0, t1 = paths.index()
            comx = paths.traj.obs[paths.traj.obs_dim - 3]
            progs = (comx[t1, env] - comx[t0, env]).to(torch.float64)
            n, s = D.sums(torch.as_tensor(float(progs.numel()), dtype=torch.float64, device=progs.device),
                          progs.sum())
            if float(n) > 0:
                mean = s / n
                (ss,) = D.sums(((progs - mean) ** 2).sum())
                mx = D.all_reduce_max_(progs.max() if progs.numel() else torch.tensor(-np.inf, device=progs.device))
                mn = D.all_reduce_min_(progs.min() if progs.numel() else torch.tensor(np.inf, device=progs.device))
                vals = (float(mean), float(mx), float(mn), float(torch.sqrt(ss / n)))
            else:
                vals = (np.nan,) * 4
        elif len(paths) > 0:
            progs = [p["observations"][-1][-3] - p["observations"][0][-3] for p in paths]
            vals = (np.mean(progs), np.max(progs), np.min(progs), np.std(progs))
        else:
            vals = (np.nan,) * 4
        for k, v in zip(('AverageForwardProgress', 'MaxForwardProgress', 'MinForwardProgress',
                         'StdForwardProgress'), vals):
            logger.record_tabular(k, v)
