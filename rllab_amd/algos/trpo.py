"""Trust Region Policy Optimization = NPO + ConjugateGradientOptimizer
(API of rllab/algos/trpo.py:6-20)."""
from rllab_amd.algos.npo import NPO
from rllab_amd.optimizers.conjugate_gradient_optimizer import ConjugateGradientOptimizer


class TRPO(NPO):
    def __init__(self, optimizer=None, optimizer_args=None, **kwargs):
        if optimizer is None:
            if optimizer_args is None:
                optimizer_args = dict()
            optimizer = ConjugateGradientOptimizer(**optimizer_args)
        super(TRPO, self).__init__(optimizer=optimizer, **kwargs)
