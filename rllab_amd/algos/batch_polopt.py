"""BatchPolopt driver and the in-process BatchSampler
(API of rllab/algos/batch_polopt.py:9-161).

``train()`` is the reference's loop: obtain_samples -> process_samples ->
log_diagnostics -> optimize_policy -> snapshot -> dump_tabular.  The default
``sampler_cls`` is the lock-step GPU ``VectorizedSampler`` when the env is
HIP-native (``env.vectorized``); otherwise the ``BatchSampler`` below, which rolls a Python
env through ``sampler/parallel_sampler.py`` (in-process, or on its CPU worker pool) like the reference.
"""
import time

import rllab_amd.misc.logger as logger
from rllab_amd.algos.base import RLAlgorithm
from rllab_amd.sampler.base import BaseSampler
from rllab_amd.sampler import dist as D
from rllab_amd.sampler import parallel_sampler
from rllab_amd.sampler.utils import truncate_paths


class BatchSampler(BaseSampler):
    """Whole-path sampler for arbitrary Python envs (reference: batch_polopt.py:9-34): through the CPU worker
    pool when ``parallel_sampler.initialize(n_parallel > 1)`` was called, in-process otherwise."""

    def __init__(self, algo):
        self.algo = algo

    def start_worker(self):
        parallel_sampler.populate_task(self.algo.env, self.algo.policy, scope=self.algo.scope)

    def shutdown_worker(self):
        parallel_sampler.terminate_task(scope=self.algo.scope)

    def obtain_samples(self, itr):
        algo = self.algo
        paths = parallel_sampler.sample_paths(policy_params=algo.policy.get_param_values(),
                                              max_samples=algo.batch_size, max_path_length=algo.max_path_length,
                                              scope=algo.scope)
        if algo.whole_paths:
            return paths
        return truncate_paths(paths, algo.batch_size)


class BatchPolopt(RLAlgorithm):
    def __init__(self, env, policy, baseline, scope=None, n_itr=500, start_itr=0, batch_size=5000,
                 max_path_length=500, discount=0.99, gae_lambda=1, plot=False, pause_for_plot=False,
                 center_adv=True, positive_adv=False, store_paths=False, whole_paths=True,
                 sampler_cls=None, sampler_args=None, prefetch_rollout=True, **kwargs):
        self.env = env
        self.policy = policy
        self.baseline = baseline
        self.scope = scope
        self.n_itr = n_itr
        self.current_itr = start_itr
        self.batch_size = batch_size
        self.max_path_length = max_path_length
        self.discount = discount
        self.gae_lambda = gae_lambda
        self.plot = plot
        self.pause_for_plot = pause_for_plot
        self.center_adv = center_adv
        self.positive_adv = positive_adv
        self.store_paths = store_paths
        self.whole_paths = whole_paths
        # engine option: enqueue the next iteration's rollout right after the update (VectorizedSampler.prefetch)
        self.prefetch_rollout = prefetch_rollout
        if plot:
            raise NotImplementedError("plotting is out of scope (SURVEY.md section 2, row 27)")
        if sampler_cls is None:
            if getattr(env, "vectorized", False) and getattr(policy, "vectorized", False):
                from rllab_amd.sampler.vectorized_sampler import VectorizedSampler
                sampler_cls = VectorizedSampler
            else:
                sampler_cls = BatchSampler
        if sampler_args is None:
            sampler_args = dict()
        self.sampler = sampler_cls(self, **sampler_args)
        self.itr_times = []

    def start_worker(self):
        self.sync_initial_parameters()
        self.sampler.start_worker()

    def sync_initial_parameters(self):
        """One process per GPU, every process runs the same script: the example scripts set no seed and the
        policy initialises from np.random, so rank 0's parameters (policy, and a parametric baseline's) are
        broadcast once.  From here on every rank applies the identical all-reduced update and no further
        broadcast is needed.  Logging and snapshots are rank 0's alone (misc/logger.py)."""
        from rllab_amd.sampler import dist as D
        if not D.is_distributed():
            return
        import torch
        for owner in (self.policy, self.baseline):
            flat = getattr(owner, "flat_params", None)
            if isinstance(flat, torch.Tensor):
                D.broadcast_(flat)
            elif hasattr(owner, "get_param_values") and hasattr(owner, "set_param_values"):
                try:
                    val = owner.get_param_values()
                except NotImplementedError:
                    continue
                if val is None or getattr(val, "size", 0) == 0:
                    continue
                t = torch.as_tensor(val, dtype=torch.float64)
                if torch.cuda.is_available() and D.backend() == "nccl":
                    t = t.cuda()
                D.broadcast_(t)
                owner.set_param_values(t.cpu().numpy())

    def shutdown_worker(self):
        self.sampler.shutdown_worker()

    def train(self):
        self.start_worker()
        self.init_opt()
        # everything allocated so far lives for the whole run: keep it out of the cyclic GC's
        # generation-2 walks, which otherwise stall one iteration in ~10 by tens of milliseconds
        import gc
        gc.collect()
        gc.freeze()
        for itr in range(self.current_itr, self.n_itr):
            self.train_iteration(itr)
        self.shutdown_worker()
        D.peer_shutdown()

    def train_iteration(self, itr):
        """One pass of the reference's loop body (batch_polopt.py:119-132)."""
        itr_start = time.time()
        with logger.prefix('itr #%d | ' % itr):
            paths = self.sampler.obtain_samples(itr)
            # process_samples may hand the optimizer its inputs early (prefetch_update): only here, where optimize_policy
            # follows on the same batch -- a caller that processes samples on its own pays no extra pass or collective
            self._update_follows = True
            try:
                samples_data = self.sampler.process_samples(itr, paths)
            finally:
                self._update_follows = False
            self.log_diagnostics(paths)
            # the next rollout depends on nothing but the updated parameters: an optimizer that decides its line search
            # on the device calls this hook once the whole update is enqueued (before it reads the outcome), so the
            # rollout starts the moment the accepted candidate's pass ends; any other optimizer leaves it to the call
            # below (a no-op after the hook ran at the final parameter version)
            want_next = itr + 1 < self.n_itr and hasattr(self.sampler, "prefetch") and not self.store_paths \
                and getattr(self, "prefetch_rollout", True)
            self._after_update_enqueued = (lambda: self.sampler.prefetch(itr + 1)) if want_next else None
            try:
                self.optimize_policy(itr, samples_data)
            finally:
                self._after_update_enqueued = None
            D.peer_poll()           # in-stream peer all-reduce (RLLAB_PEER_ALLREDUCE=1): did a peer stop answering?
            if want_next:
                self.sampler.prefetch(itr + 1)
            logger.log("saving snapshot...")
            params = self.get_itr_snapshot(itr, samples_data)
            self.current_itr = itr + 1
            params["algo"] = self
            if self.store_paths:
                params["paths"] = samples_data["paths"]
            logger.save_itr_params(itr, params)
            logger.log("saved")
            self.itr_times.append(time.time() - itr_start)   # includes the ENQUEUE of the prefetched rollout only
            logger.record_tabular('ItrTime', self.itr_times[-1])
            logger.dump_tabular(with_prefix=False)
        return samples_data

    def log_diagnostics(self, paths):
        self.env.log_diagnostics(paths)
        self.policy.log_diagnostics(paths)
        self.baseline.log_diagnostics(paths)

    def init_opt(self):
        raise NotImplementedError

    def get_itr_snapshot(self, itr, samples_data):
        raise NotImplementedError

    def optimize_policy(self, itr, samples_data):
        raise NotImplementedError
