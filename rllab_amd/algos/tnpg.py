"""Truncated Natural Policy Gradient (API of rllab/algos/tnpg.py:6-22): NPO with the
conjugate-gradient optimizer and a single line-search step."""
from rllab_amd.algos.npo import NPO
from rllab_amd.optimizers.conjugate_gradient_optimizer import ConjugateGradientOptimizer


class TNPG(NPO):
    def __init__(self, optimizer=None, optimizer_args=None, **kwargs):
        if optimizer is None:
            default_args = dict(max_backtracks=1)
            if optimizer_args is None:
                optimizer_args = default_args
            else:
                optimizer_args = dict(default_args, **optimizer_args)
            optimizer = ConjugateGradientOptimizer(**optimizer_args)
        super(TNPG, self).__init__(optimizer=optimizer, **kwargs)
