"""Lock-step GPU sampler -- the replacement of ``BatchSampler`` /
``parallel_sampler`` / ``stateful_pool`` (rllab/algos/batch_polopt.py:9-34,
rllab/sampler/parallel_sampler.py, rllab/sampler/stateful_pool.py).

Contract of sandbox/rocky/tf/samplers/vectorized_sampler.py:14-108: every env is
reset at the start of ``obtain_samples``, envs auto-reset when done, a path ends
at ``done`` or at ``max_path_length``, and the lock-step loop runs
``while n_samples < batch_size`` (:55) with ``n_samples`` the samples of FINISHED
paths -- the batch holds at least ``batch_size`` samples in whole paths
(rllab/algos/batch_polopt.py:23-34, rllab/sampler/parallel_sampler.py:98-126) and the
paths still running when the loop stops are dropped.  With ``algo.whole_paths=False``
every recorded step counts, running paths are kept as truncated paths and the batch
is cut to exactly ``batch_size`` samples in path order (``truncate_paths``,
parallel_sampler.py:129-155).  Here: one launch of ``max_path_length`` lock steps;
for env kinds whose ``done`` is always False (Swimmer, HalfCheetah, DoublePendulum)
that is ``n_envs * max_path_length`` samples in whole paths by construction and
nothing is counted.  For env kinds that terminate, the finished samples per lock
step are counted on the device, further launches WITHOUT a reset carry the same envs
on until the count reaches ``batch_size`` (``_meet_batch_size``), and the batch is cut
after the lock step at which the reference's loop would have stopped.
Differences that are the point of the rebuild: all ``n_envs`` copies advance in
one HIP launch; with a fusable GaussianMLPPolicy the whole ``max_path_length``
horizon (policy forward, action noise, env step, trajectory record, auto-reset)
is ONE launch; samples never leave the device.

Plug in with ``TRPO(..., sampler_cls=VectorizedSampler, sampler_args=dict(n_envs=4096))``
-- it is also the default sampler for vectorised envs.  Under ``torch.distributed``
each rank owns ``n_envs`` envs with global indices ``rank*n_envs ...`` (weak scaling).
"""
import math
import os
import time

import torch

import rllab_amd.misc.logger as logger
from rllab_amd import _lib
from rllab_amd.sampler import dist as D
from rllab_amd.sampler.base import BaseSampler
from rllab_amd.sampler.trajectories import PathList, Trajectories


class VectorizedSampler(BaseSampler):
    def __init__(self, algo, n_envs=None, seed=None, use_graph=True):
        super(VectorizedSampler, self).__init__(algo)
        self.n_envs = n_envs
        self.seed = seed
        self.use_graph = use_graph     # hipGraph replay of the per-transition loop (policies without a fused rollout)
        self.vec_env = None
        self.last_sample_time = None
        self.last_num_samples = None

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
        name, why = self.sampling_path(algo.policy)
        logger.log("sampling path: %s%s" % (name, "" if why is None else " -- " + why))

    def sampling_path(self, policy):
        """(name, reason) of the way ``obtain_samples`` will sample ``policy`` on this executor: the fused rollout (one
        launch per batch), or one of the per-transition loops (5-20x slower at thousands of envs) and WHY the fused
        kernels do not take it -- logged once by ``start_worker``; tools/exp/fallback_probe.py prints it per option."""
        ve = self.vec_env
        if self._takes_fused_rollout(policy):
            return "fused rollout kernel (one launch per batch of %d envs)" % ve.n, None
        if ve is None:
            return "none (no executor yet)", None
        if not getattr(ve, "graphable", True) and ve.position_ids is not None:
            why = "Box2DEnv(position_only=True) under NormalizedEnv(normalize_obs / normalize_reward): the running " \
                  "estimates are over the kept rows, the fused rollout's over the full observation"
        elif not getattr(ve, "graphable", True):
            why = "NormalizedEnv(normalize_obs / normalize_reward): the running estimates ride in the fused rollout of the " \
                  "(32, 32) / (64, 64) policies only (and not together with obs_noise)"
        else:
            layout = policy.kernel_layout() if hasattr(policy, "kernel_layout") else None
            dual = policy.rollout_networks() if hasattr(policy, "rollout_networks") else None
            if layout is not None or dual is not None:
                why = "the policy's weight fragments do not fit the 160 KB LDS of a CU next to this env's observation tile"
            elif getattr(policy, "state_dependent_std", False):
                why = "the policy's two networks have no rollout kernel (each: two or three tanh layers of at most 128 units)"
            elif hasattr(policy, "why_no_kernel_layout"):
                why = policy.why_no_kernel_layout()
            else:
                why = "%s is not a GaussianMLPPolicy" % type(policy).__name__
        graph = self.use_graph and getattr(ve, "graphable", True) and not os.environ.get("RLLAB_NO_GRAPH") \
            and hasattr(policy, "recorded_log_std")
        name = "per-transition loop (%s), one policy evaluation + one env-step launch per step" % (
            "hipGraph replay" if graph else "eager")
        return name, why

    def shutdown_worker(self):
        if self.vec_env is not None:
            self.vec_env.terminate()

    def __getstate__(self):
        d = dict(self.__dict__)
        d["vec_env"] = None
        d.pop("_step_graph", None)
        d.pop("_prefetched", None)
        d.pop("_carry_obs", None)
        return d

    def _takes_fused_rollout(self, policy):
        """True when ``obtain_samples`` is ONE asynchronous launch for this policy (the fused rollout kernels)."""
        ve = self.vec_env
        if ve is None:
            return False
        if hasattr(ve, "takes_rollout_of"):
            # the kernels' own answer: a layout exists AND its weight fragments fit the LDS of a CU on this env
            return ve.takes_rollout_of(policy)
        return (hasattr(policy, "kernel_layout") and policy.kernel_layout() is not None) or \
            (hasattr(policy, "rollout_networks") and policy.rollout_networks() is not None)

    def prefetch(self, itr):
        """Enqueue iteration ``itr``'s rollout NOW and keep the lazy batch for ``obtain_samples(itr)``: BatchPolopt
        calls this right after the parameter update, so the device starts the next rollout while the host still
        writes the previous iteration's log lines and snapshot.  The batch is only handed out if the parameters
        have not changed in between -- which needs ``policy.param_version()`` -- and only the fused rollout is one
        asynchronous launch: for any other policy (no version to check, or the host-bound per-transition loop,
        where nothing overlaps) this is a no-op.  An optimizer that decides its line search on the device calls this
        BEFORE it knows the outcome: a batch launched at parameters that were then rejected is dropped, and the
        executor's RNG counter goes back to where it was, so a run samples the same noise with or without the
        speculation.  An executor that carries state a dropped batch would have advanced for good -- NormalizedEnv's
        running estimates (``stateful_rollouts``) -- is never prefetched: the reference feeds each estimate exactly
        once per sampled transition, and a snapshot taken after the prefetch would hold estimates one batch ahead.
        ``BatchPolopt(prefetch_rollout=False)`` turns it off altogether."""
        policy = self.algo.policy
        pre = getattr(self, "_prefetched", None)
        if pre is not None and pre[0] == itr and hasattr(policy, "param_version") and pre[1] == policy.param_version():
            return                               # already queued at these very parameters
        self._drop_prefetched()
        if not hasattr(policy, "param_version") or not self._takes_fused_rollout(policy):
            return
        if getattr(self.vec_env, "stateful_rollouts", False):
            return
        counter = self.vec_env.step_counter
        first = self._rollout_chunk(policy, self.algo.max_path_length, True)
        self._prefetched = (itr, policy.param_version(), first, counter)

    def _drop_prefetched(self):
        """A batch launched at parameters that have moved on is thrown away; the RNG counter it consumed is given back."""
        pre = getattr(self, "_prefetched", None)
        self._prefetched = None
        if pre is not None and self.vec_env is not None:
            self.vec_env.step_counter = pre[3]

    def obtain_samples(self, itr):
        algo = self.algo
        policy = algo.policy
        pre = getattr(self, "_prefetched", None)
        first = None
        if pre is not None:
            if pre[0] == itr and pre[1] is not None and pre[1] == policy.param_version():
                first = pre[2]
                self._prefetched = None
            else:
                self._drop_prefetched()
        t_start = time.time()
        if first is None:
            first = self._rollout_chunk(policy, algo.max_path_length, True)
        traj = self._meet_batch_size(policy, first)
        self.last_num_samples = traj.B
        self.last_sample_time = time.time() - t_start  # enqueue time only (+ the count read of a terminating env)
        return PathList(traj)

    def _rollout_chunk(self, policy, steps, first):
        """``steps`` lock steps of every env as one ``Trajectories``; ``first``: every env is reset before (the start
        of ``obtain_samples``), otherwise the envs carry on where the previous chunk left them."""
        graphable = getattr(self.vec_env, "graphable", True)
        if self._takes_fused_rollout(policy):
            return self.vec_env.rollout(policy, steps, reset_at_start=first)
        if self.use_graph and graphable and not os.environ.get("RLLAB_NO_GRAPH") and hasattr(policy, "recorded_log_std"):
            try:
                return self._stepwise_rollout_graph(policy, steps, first)
            except RuntimeError as err:
                # a policy whose get_actions cannot be captured (host round trips, data-dependent control flow):
                # say so once and sample it with the eager loop from here on
                logger.log("hipGraph capture of the per-transition loop failed (%s); using the eager loop"
                           % str(err).split("\n")[0])
                self.use_graph = False
                self._step_graph = None
                torch.cuda.synchronize()
        return self._stepwise_rollout(policy, steps, first)

    # -- the batch-size contract ------------------------------------------------------------------------------------
    MAX_EXTENSIONS = 64      # further launches before giving up on an env that never finishes a path

    def _path_index(self, traj):
        """(tin int32 [T, N] step index inside its path, valid bool [T, N] sample of a path that ends in the batch)."""
        T, N = traj.T, traj.N
        tin = torch.empty((T, N), dtype=torch.int32, device=traj.device)
        valid = torch.empty((T, N), dtype=torch.uint8, device=traj.device)
        _lib.check(_lib.lib.rl_path_scan(T, N, 0, _lib.ptr(traj.dones), None, None, 1, _lib.ptr(tin), _lib.ptr(valid),
                                         None, _lib.stream_ptr()), "rl_path_scan")
        return tin, valid.view(torch.bool)

    def _finished_by_step(self, traj):
        """[T] int64 on the device: samples in the paths that ended at or before lock step t -- ``n_samples`` of the
        reference's loop after that step (vectorized_sampler.py:55,98-99)."""
        tin, _ = self._path_index(traj)
        ended = (tin.to(torch.int64) + 1) * traj.dones.to(torch.int64)
        return torch.cumsum(ended.sum(dim=1), 0)

    def _meet_batch_size(self, policy, first):
        """Carry the envs of ``first`` (``max_path_length`` lock steps from a reset) on until the batch holds
        ``algo.batch_size`` samples in finished paths, and cut it after the lock step at which the reference's loop
        stops.  ``algo.whole_paths=False``: the first ``batch_size`` of those samples in path order (env by env), the
        last kept path truncated -- what ``truncate_paths`` leaves of the list (batch_polopt.py:30-34).
        Sharded over ranks (each with ``n_envs`` envs and ``batch_size`` samples of the job's world x batch_size): the
        count is the sum over the shards -- one sum all-reduce of the [T] per-step counts per look, for env kinds that
        terminate -- so every rank carries on by the same number of lock steps and cuts at the same one, and the job
        samples exactly what one process with all the envs would."""
        algo, v = self.algo, self.vec_env
        world, rank = D.world_size(), D.rank()
        want, N, T = int(algo.batch_size) * world, first.N * world, int(algo.max_path_length)
        chunks = [first]
        if not getattr(v, "terminates", True) and first.T == T and T > 0:
            # done is always False: every path is max_path_length steps, a round of T lock steps ends them all together
            # (nothing to count, nothing read back: the launch stays asynchronous)
            rounds = max(1, -(-want // (N * T)))
            for _ in range(rounds - 1):
                chunks.append(self._rollout_chunk(policy, T, False))
            traj = Trajectories.concat(chunks)
            if not algo.whole_paths and N * traj.T > want:
                self._keep_first(traj, want - rank * traj.B, None)
            return traj
        for _ in range(self.MAX_EXTENSIONS):
            traj = Trajectories.concat(chunks)
            chunks = [traj]
            finished = D.all_reduce_sum_(self._finished_by_step(traj)).cpu()   # the one host read of such a batch
            hit = torch.nonzero(finished >= want)
            if hit.numel() > 0:
                traj = traj.first_steps(int(hit[0]) + 1)
                if not algo.whole_paths and int(finished[int(hit[0])]) > want:
                    whole = self._path_index(traj)[1]
                    mine = whole.sum().reshape(1)
                    before = int(D.all_gather_rows(mine)[:rank].sum()) if world > 1 else 0   # path order: rank by rank
                    self._keep_first(traj, want - before, whole)
                return traj
            short = want - int(finished[-1])
            # paths still running are finished by the continuation and count then; aim a little past the shortfall
            k = max(8, -(-3 * short // (2 * N)))
            chunks.append(self._rollout_chunk(policy, min(k, T) if T > 0 else k, False))
        raise RuntimeError("VectorizedSampler: %d samples in finished paths after %d further launches, batch_size is %d "
                           "(an env that never ends a path, and no max_path_length?)"
                           % (int(finished[-1]), self.MAX_EXTENSIONS, want))

    @staticmethod
    def _keep_first(traj, want, whole):
        """``traj.valid`` <- the first ``want`` samples of the finished paths in path order (env by env, then time);
        the sample the cut falls on ends its path (``truncate_paths``, parallel_sampler.py:129-155).  ``whole``
        [T, N] bool: samples of paths that end in the batch (None: all of them)."""
        T, N = traj.T, traj.N
        dev = traj.device
        want = max(0, min(int(want), T * N))      # (a shard behind the cut keeps nothing, one before it everything)
        if whole is None:
            rank = (torch.arange(N, device=dev).unsqueeze(0) * T + torch.arange(T, device=dev).unsqueeze(1)) + 1
            keep = rank <= want
        else:
            rank = torch.cumsum(whole.t().reshape(-1).to(torch.int64), 0).reshape(N, T).t()   # 1-based, env-major
            keep = whole & (rank <= want)
        traj.valid = keep.contiguous()
        traj.dones[keep & (rank == want)] = 1

    # -- arbitrary vectorised policies: one transition at a time --------------------------------------------------
    # T x (policy kernels, noise, copies, one env-step kernel) is a launch-bound loop: at 4096 envs every kernel in
    # it runs for microseconds.  One transition is captured into a hipGraph (torch.cuda.CUDAGraph; the env step and
    # the counter bump are nodes like any other because they are launched on the capturing stream) and replayed T
    # times; what changes from replay to replay lives in device memory -- the write position t_idx, the RNG counter
    # of the env step (rl_vecenv_step_graph reads it from a device word) and torch's own graph-safe generator state.
    def _graph_for(self, policy, T):
        v = self.vec_env
        key = (id(policy), T, v.n)
        g = getattr(self, "_step_graph", None)
        if g is not None and g["key"] == key:
            return g
        n, do, da = v.n, v.obs_rows, v.q["act_dim"]
        dev = v.device
        f32 = dict(dtype=torch.float32, device=dev)
        st = dict(key=key,
                  obs_p=torch.empty((do, T, n), **f32), act_p=torch.empty((da, T, n), **f32),
                  mean_p=torch.empty((da, T, n), **f32), rew_p=torch.empty((T, n), **f32),
                  done_p=torch.empty((T, n), dtype=torch.uint8, device=dev),
                  a_planes=torch.zeros((da, n), **f32),
                  # agent_info["log_std"] per sample: only a policy with a log-std network needs the planes
                  ls_p=torch.empty((da, T, n), **f32) if getattr(policy, "state_dependent_std", False) else None,
                  t_idx=torch.zeros(1, dtype=torch.int64, device=dev),
                  counter=torch.zeros(1, dtype=torch.int64, device=dev))

        def one_transition():
            obs = v._filtered(v._obs)                                  # [n, Do] view of the executor's buffer
            st["obs_p"].index_copy_(1, st["t_idx"], obs.t().unsqueeze(1))
            actions, info = policy.get_actions(obs)
            st["a_planes"].copy_(actions.t())
            st["act_p"].index_copy_(1, st["t_idx"], st["a_planes"].unsqueeze(1))
            st["mean_p"].index_copy_(1, st["t_idx"], info["mean"].t().unsqueeze(1))
            if st["ls_p"] is not None:
                st["ls_p"].index_copy_(1, st["t_idx"], info["log_std"].t().unsqueeze(1))
            _lib.check(_lib.lib.rl_vecenv_step_graph(
                v.kind, n, int(v.normalize), v.scale_reward, v.max_path_length, int(v.auto_reset),
                _lib.ptr(v.state), _lib.ptr(v.ts), _lib.ptr(st["a_planes"]), v.seed, _lib.ptr(st["counter"]),
                v.env_offset, v._cfg_with(), _lib.ptr(v._obs), _lib.ptr(v._reward), _lib.ptr(v._done),
                _lib.stream_ptr()), "rl_vecenv_step_graph")
            st["rew_p"].index_copy_(0, st["t_idx"], v._reward.unsqueeze(0))
            st["done_p"].index_copy_(0, st["t_idx"], v._done.unsqueeze(0))
            st["t_idx"].add_(1)
            _lib.check(_lib.lib.rl_counter_add(_lib.ptr(st["counter"]), 1, _lib.stream_ptr()), "rl_counter_add")

        # warm-up on a side stream (allocator pools, lazy initialisation), then the capture itself; neither may be
        # counted as a transition, so position and env state are restored around them
        snap = (v.state.clone(), v.ts.clone(), v._obs.clone())
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(2):
                st["t_idx"].zero_()
                one_transition()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        st["t_idx"].zero_()
        with torch.no_grad(), torch.cuda.graph(graph):
            one_transition()
        v.state.copy_(snap[0]); v.ts.copy_(snap[1]); v._obs.copy_(snap[2])
        st["graph"] = graph
        self._step_graph = st
        return st

    def _stepwise_rollout_graph(self, policy, steps, first=True):
        v = self.vec_env
        if first:
            v.reset()
        T = max(int(steps), int(self.algo.max_path_length))      # the graph's planes: one full horizon
        st = self._graph_for(policy, T)
        st["t_idx"].zero_()
        st["counter"].fill_(v.step_counter)
        for _ in range(steps):
            st["graph"].replay()
        v.step_counter += steps
        self._carry_obs = None                                    # (the executor's own buffer holds the last observation)
        # the planes belong to the graph: hand out copies, as the eager path hands out fresh tensors
        cp3 = lambda x: x[:, :steps, :].clone()
        return Trajectories(cp3(st["obs_p"]), cp3(st["act_p"]), cp3(st["mean_p"]),
                            policy.recorded_log_std(), st["rew_p"][:steps].clone(), st["done_p"][:steps].clone(),
                            v.max_path_length, log_std_planes=None if st["ls_p"] is None else cp3(st["ls_p"]))

    def _stepwise_rollout(self, policy, T, first=True):
        """Generic vectorised path: one policy.get_actions + one rl_vecenv_step launch per step."""
        v = self.vec_env
        n, do, da = v.n, v.obs_rows, v.q["act_dim"]
        dev = v.device
        obs_p = torch.empty((do, T, n), dtype=torch.float32, device=dev)
        act_p = torch.empty((da, T, n), dtype=torch.float32, device=dev)
        mean_p = torch.empty((da, T, n), dtype=torch.float32, device=dev)
        rew_p = torch.empty((T, n), dtype=torch.float32, device=dev)
        done_p = torch.empty((T, n), dtype=torch.uint8, device=dev)
        ls_p = torch.empty((da, T, n), dtype=torch.float32, device=dev) \
            if getattr(policy, "state_dependent_std", False) else None
        if first:
            obs = v.reset()
        else:         # carry on from the observation the previous chunk ended on (whitened, under running normalisation)
            obs = getattr(self, "_carry_obs", None)
            if obs is None:
                obs = v._filtered(v._obs)
        for t in range(T):
            obs_p[:, t, :] = obs.t()
            actions, info = policy.get_actions(obs)
            act_p[:, t, :] = actions.t()
            mean_p[:, t, :] = info["mean"].t()
            if ls_p is not None:
                ls_p[:, t, :] = info["log_std"].t()
            obs, rew, done, _ = v.step(actions)
            rew_p[t] = rew
            done_p[t] = done.to(torch.uint8)
        self._carry_obs = obs
        return Trajectories(obs_p, act_p, mean_p, policy.recorded_log_std(), rew_p, done_p,
                            v.max_path_length, log_std_planes=ls_p)
