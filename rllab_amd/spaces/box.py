"""Box: an axis-aligned box in R^n, possibly unbounded (API of rllab/spaces/box.py:8-77).
``Box(low, high)`` takes two arrays of one shape, ``Box(lo, hi, shape)`` broadcasts two scalars."""
import numpy as np

from rllab_amd.spaces.base import Space


class Box(Space):
    def __init__(self, low, high, shape=None):
        if shape is not None:
            if not (np.isscalar(low) and np.isscalar(high)):
                raise AssertionError("Box(low, high, shape): low / high must be scalars when a shape is given")
            low, high = np.full(shape, low, dtype=np.float64), np.full(shape, high, dtype=np.float64)
        else:
            low, high = np.asarray(low), np.asarray(high)
            if low.shape != high.shape:
                raise AssertionError("Box: low %s and high %s differ in shape" % (low.shape, high.shape))
        self.low, self.high = low, high

    # -- geometry -----------------------------------------------------------------------------------
    shape = property(lambda self: self.low.shape)
    bounds = property(lambda self: (self.low, self.high))
    flat_dim = property(lambda self: int(np.prod(self.low.shape)))

    def sample(self):
        return np.random.uniform(low=self.low, high=self.high, size=self.shape)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all((self.low <= x) & (x <= self.high)))

    # -- (un)flattening: one element, or a batch with the leading axis kept -----------------------------
    def flatten(self, x):
        return np.asarray(x).reshape(-1)

    def unflatten(self, x):
        return np.asarray(x).reshape(self.shape)

    def flatten_n(self, xs):
        xs = np.asarray(xs)
        return xs.reshape(len(xs), -1)

    def unflatten_n(self, xs):
        xs = np.asarray(xs)
        return xs.reshape((len(xs),) + self.shape)

    # -- value semantics ------------------------------------------------------------------------------
    def __eq__(self, other):
        return isinstance(other, Box) and self.shape == other.shape and \
            np.allclose(self.low, other.low) and np.allclose(self.high, other.high)

    def __hash__(self):
        return hash((self.low.tobytes(), self.high.tobytes()))

    def __repr__(self):
        return "Box" + str(self.shape)
