"""Natural Policy Optimization (API of rllab/algos/npo.py:10-132).

``init_opt`` hands the optimizer two closures instead of Theano expressions:
    surr_loss(theta) = - sum_b w_b * lr_b * adv_b / W      lr = p_theta(a|o) / p_old(a|o)
    mean_kl(theta)   =   sum_b w_b * KL(old_b || theta) / W
(reference :72-82; w = 0/1 validity weights of the dense batch, W = global number
of valid samples, so a sum all-reduce over env shards yields the global mean).
Inputs follow the reference's order [obs, actions, advantages, *state_infos,
old mean, old log_std] (:84-88) with the two batch-normalisation extras appended:
[..., weights, 1/W].  All per-sample tensors are "planes" with the sample axis
LAST (obs [Do, B], actions [Da, B], ...).
"""
import os

import torch

import rllab_amd.misc.logger as logger
from rllab_amd.algos.batch_polopt import BatchPolopt
from rllab_amd.sampler import dist as D


def npo_inputs(policy, samples_data):
    """Build the optimizer input tuple from a processed dense batch."""
    traj = samples_data["_traj"]
    B = traj.B
    w = traj.valid.reshape(B).to(torch.float32)
    if getattr(traj, "count", None) is not None:
        # process_samples already read the global number of valid samples (rl_sample_stats): no second
        # reduction + blocking read for 1 / W
        cnt = torch.tensor(float(traj.count), dtype=torch.float64)
    else:
        cnt = D.all_reduce_sum_(w.to(torch.float64).sum())
    old_ls = traj.log_std.reshape(-1, 1) if traj.log_std_planes is None \
        else traj.log_std_planes.reshape(traj.act_dim, B)
    return (traj.obs.reshape(traj.obs_dim, B), traj.actions.reshape(traj.act_dim, B),
            traj.advantages.reshape(B), traj.means.reshape(traj.act_dim, B), old_ls, w, (1.0 / cnt))


def log_update_path(policy, fused, why=None):
    """One log line naming where loss / gradient / Fisher-vector products of the update run, and why when it is not
    the HIP kernels (the autograd path costs 10-20x at the batch sizes this engine samples)."""
    if fused is not None:
        logger.log("update path: HIP kernels (%s)" % type(fused).__name__)
        return
    if why is None and hasattr(policy, "why_no_kernel_layout"):
        why = policy.why_no_kernel_layout()
    logger.log("update path: torch autograd%s" % ("" if why is None else " -- " + why))


def pick_optimizer(optimizer, optimizer_args, default_cls, **default_args):
    """The optimizer an NPO variant runs with: the one handed in, else ``default_cls`` built from the
    variant's defaults overridden by ``optimizer_args``."""
    if optimizer is not None:
        return optimizer
    return default_cls(**dict(default_args, **(optimizer_args or {})))


class NPO(BatchPolopt):
    def __init__(self, optimizer=None, optimizer_args=None, step_size=0.01,
                 truncate_local_is_ratio=None, **kwargs):
        if optimizer is None:
            from rllab_amd.optimizers.penalty_lbfgs_optimizer import PenaltyLbfgsOptimizer
            optimizer = pick_optimizer(None, optimizer_args, PenaltyLbfgsOptimizer)   # reference default (npo.py:27-30)
        self.optimizer = optimizer
        self.step_size = step_size
        self.truncate_local_is_ratio = truncate_local_is_ratio
        super(NPO, self).__init__(**kwargs)

    def init_opt(self):
        if self.policy.recurrent:
            raise NotImplementedError("recurrent policies are outside the hot path built here")
        policy = self.policy
        dist = policy.distribution
        trunc = self.truncate_local_is_ratio

        def _new_dist(flat, obs):
            return policy.dist_info_planes(obs, flat)

        def surr_loss(flat, obs, act, adv, old_mean, old_log_std, w, inv_count):
            new = _new_dist(flat, obs)
            old = dict(mean=old_mean, log_std=old_log_std)
            lr = dist.likelihood_ratio_sym(act, old, new, axis=0)
            if trunc is not None:
                lr = torch.clamp(lr, max=trunc)
            return -(lr * adv * w).sum() * inv_count.to(lr.dtype)

        def mean_kl(flat, obs, act, adv, old_mean, old_log_std, w, inv_count):
            new = _new_dist(flat, obs)
            old = dict(mean=old_mean, log_std=old_log_std)
            kl = dist.kl_sym(old, new, axis=0)
            return (kl * w).sum() * inv_count.to(kl.dtype)

        fused = None
        if trunc is None and hasattr(policy, "fused_ops") and getattr(self, "use_fused", True):
            fused = policy.fused_ops()
        log_update_path(policy, fused, "truncate_local_is_ratio is set (the kernels evaluate the plain likelihood ratio)"
                        if trunc is not None else None)
        self.optimizer.update_opt(loss=surr_loss, target=policy, leq_constraint=(mean_kl, self.step_size),
                                  inputs=None, constraint_name="mean_kl", fused=fused)
        return dict()

    def prefetch_update(self, samples_data):
        """Called by ``process_samples`` once the advantages are on the device: hands the optimizer its inputs early so
        that its first pass runs while the host logs (optimizers/conjugate_gradient_optimizer.py::prefetch)."""
        self._prefetched = None
        if not hasattr(self.optimizer, "prefetch") or not getattr(self, "prefetch_update_enabled", True) or \
                os.environ.get("RLLAB_UPDATE_PREFETCH", "1")[:1] == "0":       # (A/B timing)
            return
        values = npo_inputs(self.policy, samples_data)
        self._prefetched = (samples_data, values)
        self.optimizer.prefetch(values)

    def optimize_policy(self, itr, samples_data):
        pre, self._prefetched = getattr(self, "_prefetched", None), None
        all_input_values = pre[1] if pre is not None and pre[0] is samples_data else npo_inputs(self.policy, samples_data)
        if getattr(self.optimizer, "reports_before_values", False):
            # (train_iteration's hook: what to enqueue behind the update before its outcome is read -- the next rollout)
            self.optimizer._after_enqueue = getattr(self, "_after_update_enqueued", None)
            try:
                self.optimizer.optimize(all_input_values)
            finally:
                self.optimizer._after_enqueue = None
            loss_before, mean_kl_before = self.optimizer.last_before
        else:
            loss_before = self.optimizer.loss(all_input_values)
            mean_kl_before = self.optimizer.constraint_val(all_input_values)
            self.optimizer.optimize(all_input_values)
        mean_kl = self.optimizer.constraint_val(all_input_values)
        loss_after = self.optimizer.loss(all_input_values)
        logger.record_tabular('LossBefore', loss_before)
        logger.record_tabular('LossAfter', loss_after)
        logger.record_tabular('MeanKLBefore', mean_kl_before)
        logger.record_tabular('MeanKL', mean_kl)
        logger.record_tabular('dLoss', loss_before - loss_after)
        fused = getattr(self.optimizer, "_fused", None)
        if fused is not None:
            fused.release()   # drop the cached batch descriptor (it keeps the batch tensors alive)
        return dict()

    def get_itr_snapshot(self, itr, samples_data):
        return dict(itr=itr, policy=self.policy, baseline=self.baseline, env=self.env)
