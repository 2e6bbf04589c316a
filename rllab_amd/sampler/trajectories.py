"""Dense device-resident trajectory planes of one lock-step rollout, and the
lazy ``paths`` view the rllab API still expects.

Layout (all on the device, float32 unless noted; B = T*N, flat sample index
b = t*N + n):
    obs      [Do, T, N]   observation the action was computed from
    actions  [Da, T, N]
    means    [Da, T, N]   agent_info "mean"
    log_std  [Da]         agent_info "log_std" is one constant row (state-independent)
    rewards  [T, N]
    dones    [T, N] uint8 env done OR ts == max_path_length
A *path* is a maximal run of consecutive t in one env column that ends at a
done flag or at the last recorded step.  Nothing here copies B-sized data to the
host: per-path Python dicts are materialised lazily, one path at a time, only if
user code indexes ``paths[i]`` (SURVEY.md section 7 "API still wants Python paths").
"""
import numpy as np
import torch


class Trajectories(object):
    def __init__(self, obs, actions, means, log_std, rewards, dones, max_path_length, log_std_planes=None):
        self.obs, self.actions, self.means = obs, actions, means
        self.log_std, self.rewards, self.dones = log_std, rewards, dones
        # [Da, T, N] per-sample log_std, only for batches packed from arbitrary user paths whose
        # policy does not have a state-independent log_std (None for engine rollouts)
        self.log_std_planes = log_std_planes
        self.max_path_length = max_path_length
        self.T, self.N = rewards.shape
        self.obs_dim, self.act_dim = obs.shape[0], actions.shape[0]
        self._seg = None
        # filled by process_samples
        self.valid = None
        self.advantages = None
        self.returns = None
        self.baselines = None
        self.tin = None          # [T, N] int32 step index inside its path (rl_path_scan)
        self.progress_stats = None   # (mean, max, min, std) of the env's per-path progress (rl_sample_stats)
        self.count = None            # global number of valid samples, host float (read with the statistics)
        self.log_std_host = None     # the recorded log_std row on the host (read with the statistics)

    @property
    def device(self):
        return self.rewards.device

    # -- horizons made of several launches ----------------------------------------
    _PLANES = ("obs", "actions", "means", "log_std_planes")      # [D, T, N]
    _ROWS = ("rewards", "dones")                                   # [T, N]

    @classmethod
    def concat(cls, chunks):
        """Consecutive lock-step chunks of the SAME envs (the later ones launched without a reset) as one batch:
        the planes joined along the time axis."""
        if len(chunks) == 1:
            return chunks[0]
        first = chunks[0]
        cat = lambda name, dim: (None if getattr(first, name) is None
                                 else torch.cat([getattr(c, name) for c in chunks], dim=dim))
        return cls(cat("obs", 1), cat("actions", 1), cat("means", 1), first.log_std, cat("rewards", 0), cat("dones", 0),
                   first.max_path_length, log_std_planes=cat("log_std_planes", 1))

    def first_steps(self, steps):
        """The batch cut after ``steps`` lock steps (contiguous copies; ``self`` when nothing is cut)."""
        if steps >= self.T:
            return self
        cut3 = lambda x: None if x is None else x[:, :steps, :].contiguous()
        return Trajectories(cut3(self.obs), cut3(self.actions), cut3(self.means), self.log_std,
                            self.rewards[:steps].contiguous(), self.dones[:steps].contiguous(), self.max_path_length,
                            log_std_planes=cut3(self.log_std_planes))

    @property
    def B(self):
        return self.T * self.N

    # -- path segmentation ------------------------------------------------------
    def segments(self):
        """(env, t_start, t_end_inclusive, complete) int64 tensors, ordered by env
        then time.  ``complete`` is False for a trailing path cut by the end of
        the rollout rather than by a done flag."""
        if self._seg is None:
            T, N = self.T, self.N
            done = self.dones.bool()
            end = done.clone()
            end[T - 1] = True
            start = torch.ones_like(done)
            start[1:] = done[:-1]
            s_idx = torch.nonzero(start.t(), as_tuple=False)  # rows (n, t), sorted by n then t
            e_idx = torch.nonzero(end.t(), as_tuple=False)
            env = s_idx[:, 0]
            t0 = s_idx[:, 1]
            t1 = e_idx[:, 1]
            complete = done[t1, env]
            self._seg = (env, t0, t1, complete)
        return self._seg

    def time_in_path(self):
        """[T, N] int64: index of each step inside its path (0 at a path start)."""
        T, N = self.T, self.N
        done = self.dones.bool()
        start = torch.ones_like(done)
        start[1:] = done[:-1]
        t_idx = torch.arange(T, device=self.device).unsqueeze(1).expand(T, N)
        start_t = torch.where(start, t_idx, torch.zeros_like(t_idx))
        last_start = torch.cummax(start_t, dim=0).values
        return t_idx - last_start

    def valid_mask(self, whole_paths=True):
        """[T, N] bool.  With ``whole_paths`` (reference default,
        batch_polopt.py:30-34) samples of trailing incomplete paths are dropped,
        as sandbox VectorizedSampler does (vectorized_sampler.py:72-97);
        otherwise they are kept as truncated paths (truncate_paths semantics)."""
        T, N = self.T, self.N
        if not whole_paths:
            return torch.ones((T, N), dtype=torch.bool, device=self.device)
        done = self.dones.bool()
        # a step is valid iff some done flag occurs at or after it in its column
        suffix_any = torch.flip(torch.cummax(torch.flip(done, [0]).to(torch.uint8), dim=0).values, [0])
        return suffix_any.bool()


class PathList(object):
    """Lazy ``paths`` sequence over a ``Trajectories``.  ``len`` and iteration work
    like a list of path dicts; each dict is built on demand from host copies of
    ONE path's rows.  Once ``process_samples`` has set ``traj.valid`` only valid
    paths are listed (the index is resolved at first use)."""

    def __init__(self, traj, only_valid=True):
        self.traj = traj
        self.only_valid = only_valid
        self._idx = None
        self._host = None

    def index(self):
        """(env, t0, t1) device int64 tensors of the listed paths."""
        if self._idx is None:
            env, t0, t1, _complete = self.traj.segments()
            if self.only_valid and self.traj.valid is not None:
                keep = self.traj.valid[t0, env]
                env, t0, t1 = env[keep], t0[keep], t1[keep]
            self._idx = (env, t0, t1)
        return self._idx

    def __len__(self):
        return int(self.index()[0].numel())

    def _host_index(self):
        if self._host is None:
            self._host = tuple(x.cpu().numpy() for x in self.index())
        return self._host

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        env, t0, t1 = self._host_index()
        n, a, b = int(env[i]), int(t0[i]), int(t1[i]) + 1
        tr = self.traj
        L = b - a
        f64 = lambda x: x.detach().cpu().numpy().astype(np.float64)
        path = dict(
            observations=f64(tr.obs[:, a:b, n].t()),
            actions=f64(tr.actions[:, a:b, n].t()),
            rewards=f64(tr.rewards[a:b, n]),
            agent_infos=dict(mean=f64(tr.means[:, a:b, n].t()),
                             log_std=(f64(tr.log_std_planes[:, a:b, n].t()) if tr.log_std_planes is not None
                                      else np.tile(f64(tr.log_std)[None, :], (L, 1)))),
            env_infos=dict(),
        )
        if tr.advantages is not None:
            path["advantages"] = f64(tr.advantages[a:b, n])
            path["returns"] = f64(tr.returns[a:b, n])
        return path

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]
