"""TNPG, truncated natural policy gradient (API of rllab/algos/tnpg.py:6-22): TRPO's direction with
the full step taken once -- ``max_backtracks=1`` unless ``optimizer_args`` says otherwise."""
from rllab_amd.algos.npo import NPO, pick_optimizer
from rllab_amd.optimizers.conjugate_gradient_optimizer import ConjugateGradientOptimizer


class TNPG(NPO):
    def __init__(self, optimizer=None, optimizer_args=None, **kwargs):
        chosen = pick_optimizer(optimizer, optimizer_args, ConjugateGradientOptimizer, max_backtracks=1)
        NPO.__init__(self, optimizer=chosen, **kwargs)
