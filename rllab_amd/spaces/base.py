"""Space interface (mirrors rllab/spaces/base.py)."""


class Space(object):
    def sample(self, seed=0):
        raise NotImplementedError

    def contains(self, x):
        raise NotImplementedError

    def flatten(self, x):
        raise NotImplementedError

    def unflatten(self, x):
        raise NotImplementedError

    def flatten_n(self, xs):
        raise NotImplementedError

    def unflatten_n(self, xs):
        raise NotImplementedError

    @property
    def flat_dim(self):
        raise NotImplementedError

    def new_tensor_variable(self, name, extra_dims):
        raise NotImplementedError("symbolic variables do not exist in rllab_amd (torch closures instead)")
