"""Lock-step GPU sampler -- the replacement of ``BatchSampler`` /
``parallel_sampler`` / ``stateful_pool`` (rllab/algos/batch_polopt.py:9-34,
rllab/sampler/parallel_sampler.py, rllab/sampler/stateful_pool.py).

Contract of sandbox/rocky/tf/samplers/vectorized_sampler.py:14-108: every env is
reset at the start of ``obtain_samples``, envs auto-reset when done, a path ends
at ``done`` or at ``max_path_length``; trailing unfinished paths are dropped when
``algo.whole_paths`` (reference default) and kept as truncated paths otherwise.
Differences that are the point of the rebuild: all ``n_envs`` copies advance in
one HIP launch; with a fusable GaussianMLPPolicy the whole ``max_path_length``
horizon (policy forward, action noise, env step, trajectory record, auto-reset)
is ONE launch; samples never leave the device.

Plug in with ``TRPO(..., sampler_cls=VectorizedSampler, sampler_args=dict(n_envs=4096))``
-- it is also the default sampler for vectorised envs.  Under ``torch.distributed``
each rank owns ``n_envs`` envs with global indices ``rank*n_envs ...`` (weak scaling).
"""
import math
import time

import torch

import rllab_amd.misc.logger as logger
from rllab_amd.sampler import dist as D
from rllab_amd.sampler.base import BaseSampler
from rllab_amd.sampler.trajectories import PathList, Trajectories


class VectorizedSampler(BaseSampler):
    def __init__(self, algo, n_envs=None, seed=None):
        super(VectorizedSampler, self).__init__(algo)
        self.n_envs = n_envs
        self.seed = seed
        self.vec_env = None
        self.last_sample_time = None
        self.last_num_samples = None

    def __getstate__(self):
        d = dict(self.__dict__)
        d["vec_env"] = None
        return d

    def start_worker(self):
        algo = self.algo
        n_envs = self.n_envs
        if n_envs is None:
            n_envs = max(1, int(math.ceil(algo.batch_size / float(algo.max_path_length))))
        if not getattr(algo.env, "vectorized", False):
            raise NotImplementedError("VectorizedSampler needs env.vectorized (a HIP-native env)")
        kw = dict(n_envs=n_envs, max_path_length=algo.max_path_length, env_offset=D.rank() * n_envs)
        if self.seed is not None:
            kw["seed"] = self.seed
        self.vec_env = algo.env.vec_env_executor(**kw)
        self.n_envs = n_envs

    def shutdown_worker(self):
        if self.vec_env is not None:
            self.vec_env.terminate()

    def obtain_samples(self, itr):
        algo = self.algo
        policy = algo.policy
        T = algo.max_path_length
        t_start = time.time()
        if getattr(policy, "fusable", False) and len(getattr(policy, "hidden_sizes", ())) == 2 \
                and tuple(policy.hidden_sizes) in ((32, 32), (64, 64)):
            traj = self.vec_env.rollout(policy, T, reset_at_start=True)
        else:
            traj = self._stepwise_rollout(policy, T)
        self.last_traj = traj
        self.last_num_samples = traj.B
        self.last_sample_time = time.time() - t_start  # enqueue time only; bench syncs explicitly
        return PathList(traj)

    def _stepwise_rollout(self, policy, T):
        """Generic vectorised path: one policy.get_actions + one rl_vecenv_step launch per step."""
        v = self.vec_env
        n, do, da = v.n, v.q["obs_dim"], v.q["act_dim"]
        dev = v.device
        obs_p = torch.empty((do, T, n), dtype=torch.float32, device=dev)
        act_p = torch.empty((da, T, n), dtype=torch.float32, device=dev)
        mean_p = torch.empty((da, T, n), dtype=torch.float32, device=dev)
        rew_p = torch.empty((T, n), dtype=torch.float32, device=dev)
        done_p = torch.empty((T, n), dtype=torch.uint8, device=dev)
        obs = v.reset()
        for t in range(T):
            obs_p[:, t, :] = obs.t()
            actions, info = policy.get_actions(obs)
            act_p[:, t, :] = actions.t()
            mean_p[:, t, :] = info["mean"].t()
            obs, rew, done, _ = v.step(actions)
            rew_p[t] = rew
            done_p[t] = done.to(torch.uint8)
        return Trajectories(obs_p, act_p, mean_p, policy.effective_log_std().detach(), rew_p, done_p,
                            v.max_path_length)
