"""The helpers of rllab/misc/ext.py the hot path uses, and nothing else of that module: ``extract`` (:14-20),
``set_seed`` (:188-206), ``is_iterable`` (:209-210), ``flatten_tensor_variables`` (:297-299), ``sliced_fun``
(:341-370).  The reference file is the contract (what goes in, what comes out), not the text."""
import random

import numpy as np

seed_ = None


def extract(x, *keys):
    """``extract(d, 'a', 'b') -> (d['a'], d['b'])``; for a list of dicts each entry is the list over the dicts."""
    if isinstance(x, dict):
        return tuple(x[k] for k in keys)
    if isinstance(x, list):
        return tuple([item[k] for item in x] for k in keys)
    raise NotImplementedError("extract() takes a dict or a list of dicts, not %s" % type(x).__name__)


def set_seed(seed):
    """Seeds ``random`` / ``np.random`` / torch, and becomes the key of the
    in-kernel Philox streams (samplers read ``get_seed()``)."""
    global seed_
    seed %= 4294967294
    seed_ = seed
    random.seed(seed)
    np.random.seed(seed)
    import torch
    torch.manual_seed(seed)


def get_seed():
    return seed_


class _SampleWeightedMean(object):
    """Accumulator behind ``sliced_fun``: sum_k n_k * out_k / sum_k n_k, per output position."""

    def __init__(self):
        self.weight = 0
        self.sums = None
        self.shape = None          # None: bare value; tuple / list: the container type f returned

    def add(self, out, n):
        if isinstance(out, (tuple, list)):
            self.shape, values = type(out), list(out)
        else:
            self.shape, values = None, [out]
        terms = [np.asarray(v) * n for v in values]
        self.sums = terms if self.sums is None else [s + t for s, t in zip(self.sums, terms)]
        self.weight += n

    def result(self):
        means = [s / self.weight for s in self.sums]
        return means[0] if self.shape is None else self.shape(means)


def sliced_fun(f, n_slices):
    """``sliced_fun(f, k)(sliced_inputs, non_sliced_inputs=None)``: cut every array of ``sliced_inputs`` into
    chunks of ``max(1, n // k)`` samples along the first axis, call ``f(*chunk, *non_sliced_inputs)`` per chunk and
    return the sample-weighted mean of what it returned -- a bare value, or a tuple / list of the same kind
    (contract: rllab/misc/ext.py:341-370).  A ragged tail is one more, shorter chunk, weighted by its own length.
    On a 288 GB part no batch of this path needs slicing, so the optimizers keep ``num_slices`` for signature
    compatibility and evaluate whole batches; this function serves callers that slice host-side closures."""
    def sliced_f(sliced_inputs, non_sliced_inputs=None):
        rest = list(non_sliced_inputs) if non_sliced_inputs is not None else []
        n = len(sliced_inputs[0])
        chunk = max(1, n // n_slices)
        acc = _SampleWeightedMean()
        for lo in range(0, n, chunk):
            part = [v[lo:lo + chunk] for v in sliced_inputs]
            acc.add(f(*(part + rest)), len(part[0]))
        return acc.result()
    return sliced_f


def is_iterable(obj):
    """rllab/misc/ext.py:209-210."""
    return isinstance(obj, str) or getattr(obj, '__iter__', False)


def flatten_tensor_variables(ts):
    """One flat vector from a list of tensors (rllab/misc/ext.py:297-299, which concatenates flattened Theano
    variables; here torch tensors, order preserved, graph kept)."""
    import torch
    return torch.cat([torch.reshape(t, (-1,)) for t in ts])
