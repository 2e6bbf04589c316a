"""Penalized Policy Optimization (API of rllab/algos/ppo.py:6-19): NPO with PenaltyLbfgsOptimizer."""
from rllab_amd.algos.npo import NPO
from rllab_amd.optimizers.penalty_lbfgs_optimizer import PenaltyLbfgsOptimizer


class PPO(NPO):
    def __init__(self, optimizer=None, optimizer_args=None, **kwargs):
        if optimizer is None:
            if optimizer_args is None:
                optimizer_args = dict()
            optimizer = PenaltyLbfgsOptimizer(**optimizer_args)
        super(PPO, self).__init__(optimizer=optimizer, **kwargs)
