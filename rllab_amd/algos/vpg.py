"""Vanilla Policy Gradient (API of rllab/algos/vpg.py:11-139).

surr_obj = - mean(log p_theta(a|o) * adv); one full-batch Adam step per iteration
by default (``FirstOrderOptimizer(batch_size=None, max_epochs=1)``); logs
LossBefore / LossAfter / MeanKL / MaxKL.
"""
import torch

import rllab_amd.misc.logger as logger
from rllab_amd.algos.batch_polopt import BatchPolopt
from rllab_amd.algos.npo import log_update_path, npo_inputs
from rllab_amd.core.serializable import Serializable
from rllab_amd.optimizers.first_order_optimizer import FirstOrderOptimizer
from rllab_amd.sampler import dist as D


class VPG(BatchPolopt, Serializable):
    def __init__(self, env, policy, baseline, optimizer=None, optimizer_args=None, **kwargs):
        Serializable.quick_init(self, locals())
        if optimizer is None:
            default_args = dict(batch_size=None, max_epochs=1)
            optimizer_args = default_args if optimizer_args is None else dict(default_args, **optimizer_args)
            optimizer = FirstOrderOptimizer(**optimizer_args)
        self.optimizer = optimizer
        self.opt_info = None
        super(VPG, self).__init__(env=env, policy=policy, baseline=baseline, **kwargs)

    def init_opt(self):
        if self.policy.recurrent:
            raise NotImplementedError("recurrent policies are outside the hot path built here")
        policy = self.policy
        dist = policy.distribution

        def surr_obj(flat, obs, act, adv, old_mean, old_log_std, w, inv_count):
            new = policy.dist_info_planes(obs, flat)
            logli = dist.log_likelihood_sym(act, new, axis=0)
            return -(logli * adv * w).sum() * inv_count.to(logli.dtype)

        def f_kl(inputs):
            obs, act, adv, old_mean, old_log_std, w, inv_count = inputs
            with torch.no_grad():
                new = policy.dist_info_planes(obs)
                kl = dist.kl_sym(dict(mean=old_mean, log_std=old_log_std), new, axis=0)
                mean_kl = D.all_reduce_sum_(((kl * w).sum() * inv_count.to(kl.dtype)).to(torch.float64))
                neg = torch.full_like(kl, -float("inf"))
                max_kl = D.all_reduce_max_(torch.where(w > 0, kl, neg).max().to(torch.float64))
            return float(mean_kl), float(max_kl)

        fused = policy.fused_ops() if hasattr(policy, "fused_ops") and getattr(self, "use_fused", True) else None
        log_update_path(policy, fused)
        if fused is not None:
            def f_kl(inputs):  # noqa: F811  (HIP kernel version of the same statistic)
                s = fused.loss_stats_host(inputs)      # the evaluation's one host read (shared with loss())
                return s[1], s[3]
        self.optimizer.update_opt(surr_obj, target=policy, inputs=None, fused=fused, weighted_mean_inputs=True)
        self.opt_info = dict(f_kl=f_kl)

    def optimize_policy(self, itr, samples_data):
        logger.log("optimizing policy")
        inputs = npo_inputs(self.policy, samples_data)
        loss_before = None
        if not getattr(self.optimizer, "reports_before_values", False):
            loss_before = self.optimizer.loss(inputs)
        self.optimizer.optimize(inputs)
        if loss_before is None:
            before = getattr(self.optimizer, "last_before", None)
            # (a mini-batch or non-fused run has evaluated the loss up front itself and cached nothing for us)
            loss_before = before[0] if before is not None else float("nan")
        loss_after = self.optimizer.loss(inputs)
        logger.record_tabular("LossBefore", loss_before)
        logger.record_tabular("LossAfter", loss_after)
        mean_kl, max_kl = self.opt_info['f_kl'](inputs)
        logger.record_tabular('MeanKL', mean_kl)
        logger.record_tabular('MaxKL', max_kl)
        fused = getattr(self.optimizer, "_fused", None)
        if fused is not None:
            fused.release()

    def get_itr_snapshot(self, itr, samples_data):
        return dict(itr=itr, policy=self.policy, baseline=self.baseline, env=self.env)
