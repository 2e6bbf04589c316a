"""TRPO (API of rllab/algos/trpo.py:6-20): natural policy optimisation whose step comes from
conjugate gradient on the Fisher matrix followed by a backtracking line search."""
from rllab_amd.algos.npo import NPO, pick_optimizer
from rllab_amd.optimizers.conjugate_gradient_optimizer import ConjugateGradientOptimizer


class TRPO(NPO):
    def __init__(self, optimizer=None, optimizer_args=None, **kwargs):
        NPO.__init__(self, optimizer=pick_optimizer(optimizer, optimizer_args, ConjugateGradientOptimizer), **kwargs)
