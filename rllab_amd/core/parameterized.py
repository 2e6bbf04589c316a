"""Flat-parameter get/set contract (mirrors rllab/core/parameterized.py:15-84).

In the reference a parameter is a Theano shared variable; here it is a ``Param``:
a named view into ONE flat float32 device vector owned by the parameterized
object, laid out exactly like ``flatten_tensors`` would concatenate the
reference's parameters (W stored ``[in, out]`` row-major).  ``get_param_values``
/ ``set_param_values`` therefore are single device<->host copies of P floats.
"""
from contextlib import contextmanager

import numpy as np
import torch

from rllab_amd.core.serializable import Serializable

load_params = True


@contextmanager
def suppress_params_loading():
    global load_params
    load_params = False
    yield
    load_params = True


class Param(object):
    """A named, shaped slice of the owner's flat parameter vector."""

    def __init__(self, name, shape, offset, trainable=True, regularizable=True):
        self.name = name
        self.shape = tuple(shape)
        self.size = int(np.prod(self.shape))
        self.offset = offset
        self.tags = dict(trainable=trainable, regularizable=regularizable)
        self._owner = None

    def view(self, flat):
        return flat[self.offset:self.offset + self.size].view(self.shape)

    def get_value(self, borrow=False):
        return self.view(self._owner.flat_params).detach().cpu().numpy()

    def set_value(self, value):
        v = torch.as_tensor(np.asarray(value), dtype=self._owner.flat_params.dtype)
        self.view(self._owner.flat_params).copy_(v.reshape(self.shape))


class Parameterized(Serializable):
    def __init__(self):
        self._cached_params = {}
        self._cached_param_dtypes = {}
        self._cached_param_shapes = {}

    def get_params_internal(self, **tags):
        raise NotImplementedError

    def get_params(self, **tags):
        tag_tuple = tuple(sorted(tags.items(), key=lambda x: x[0]))
        if tag_tuple not in self._cached_params:
            self._cached_params[tag_tuple] = self.get_params_internal(**tags)
        return self._cached_params[tag_tuple]

    def get_param_dtypes(self, **tags):
        return [np.dtype("float32") for _ in self.get_params(**tags)]

    def get_param_shapes(self, **tags):
        return [p.shape for p in self.get_params(**tags)]

    def _flat_index(self, **tags):
        """Index tensor selecting the tagged parameters inside the flat vector
        (None when the selection is the whole vector)."""
        params = self.get_params(**tags)
        total = self.flat_params.numel()
        if sum(p.size for p in params) == total:
            return None
        idx = np.concatenate([np.arange(p.offset, p.offset + p.size) for p in params]) \
            if params else np.zeros(0, dtype=np.int64)
        return torch.as_tensor(idx, dtype=torch.long, device=self.flat_params.device)

    def get_param_values(self, **tags):
        idx = self._flat_index(**tags)
        flat = self.flat_params.detach()
        if idx is not None:
            flat = flat[idx]
        return flat.cpu().numpy().astype(np.float64)

    def set_param_values(self, flattened_params, **tags):
        tags.pop("debug", None)
        idx = self._flat_index(**tags)
        v = torch.as_tensor(np.asarray(flattened_params) if not torch.is_tensor(flattened_params)
                            else flattened_params)
        v = v.to(device=self.flat_params.device, dtype=self.flat_params.dtype)
        with torch.no_grad():
            if idx is None:
                self.flat_params.copy_(v)
            else:
                self.flat_params[idx] = v

    def flat_to_params(self, flattened_params, **tags):
        out, n = [], 0
        for shape in self.get_param_shapes(**tags):
            size = int(np.prod(shape))
            out.append(flattened_params[n:n + size].reshape(shape))
            n += size
        return out

    def __getstate__(self):
        d = Serializable.__getstate__(self)
        d["params"] = self.get_param_values()
        return d

    def __setstate__(self, d):
        Serializable.__setstate__(self, d)
        if load_params:
            self.set_param_values(d["params"])
