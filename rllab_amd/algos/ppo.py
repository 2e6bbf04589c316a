"""PPO, penalised policy optimisation (API of rllab/algos/ppo.py:6-19): the KL bound of NPO enforced
by an adaptive penalty under L-BFGS instead of a trust-region step."""
from rllab_amd.algos.npo import NPO, pick_optimizer
from rllab_amd.optimizers.penalty_lbfgs_optimizer import PenaltyLbfgsOptimizer


class PPO(NPO):
    def __init__(self, optimizer=None, optimizer_args=None, **kwargs):
        NPO.__init__(self, optimizer=pick_optimizer(optimizer, optimizer_args, PenaltyLbfgsOptimizer), **kwargs)
