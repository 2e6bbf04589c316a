"""Algorithm base classes (mirror rllab/algos/base.py)."""


class Algorithm(object):
    pass


class RLAlgorithm(Algorithm):
    def train(self):
        raise NotImplementedError
