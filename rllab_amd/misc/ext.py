"""The helpers of rllab/misc/ext.py that do not need Theano: ``extract`` (:14-20), ``extract_dict`` / ``flatten`` /
``compact`` (:23-40), ``lazydict`` (:71-90), the scans (:93-120), ``AttrDict`` (:151-154), ``is_iterable`` (:157-158),
the path helpers (:162-172), ``shuffled`` (:175-182), ``set_seed`` / ``get_seed`` (:188-210),
``flatten_tensor_variables`` / ``unflatten_tensor_variables`` (:297-338, over torch tensors), ``flatten_shape_dim``
(:302-303), ``sliced_fun`` (:341-370), ``stdize`` / ``iterate_minibatches_generic`` (:373-391).  Gone on purpose: the
Theano graph helpers (``compile_function``, ``cached_function``, ``new_tensor*``, ``flatten_hessian``,
``print_lasagne_layer``) -- there is no symbolic graph here.  The reference file is the contract (what goes in, what
comes out), not the text; tests/test_host_logic.py holds each helper to the behaviour of the reference's."""
import functools
import operator
import random

import numpy as np

seed_ = None


class lazydict(object):
    """Values given as thunks, evaluated on first read and remembered (ext.py:71-90).  ``d[k] = thunk`` / ``set``
    replace the THUNK; a value already computed for ``k`` stays -- the reference's behaviour, kept."""

    def __init__(self, **thunks):
        self._lazy_dict = dict(thunks)
        self._dict = {}

    def __getitem__(self, key):
        try:
            return self._dict[key]
        except KeyError:
            value = self._dict[key] = self._lazy_dict[key]()
            return value

    def __setitem__(self, key, thunk):
        self.set(key, thunk)

    def get(self, key, default=None):
        return self[key] if key in self._lazy_dict else default

    def set(self, key, thunk):
        self._lazy_dict[key] = thunk


def extract(x, *keys):
    """``extract(d, 'a', 'b') -> (d['a'], d['b'])``; for a list of dicts each entry is the list over the dicts."""
    if isinstance(x, (dict, lazydict)):
        return tuple(x[k] for k in keys)
    if isinstance(x, list):
        return tuple([item[k] for item in x] for k in keys)
    raise NotImplementedError("extract() takes a dict or a list of dicts, not %s" % type(x).__name__)


def extract_dict(x, *keys):
    """The sub-dictionary of ``x`` over those of ``keys`` it has (ext.py:23-24)."""
    return dict((k, x[k]) for k in keys if k in x)


def flatten(xs):
    """One level of nesting removed: ``[[1, 2], [3]] -> [1, 2, 3]`` (ext.py:27-28)."""
    out = []
    for inner in xs:
        out.extend(inner)
    return out


def compact(x):
    """A dict without its ``None`` values, a list without its ``None`` elements, anything else as it is
    (ext.py:31-40)."""
    if isinstance(x, dict):
        return {k: v for k, v in x.items() if v is not None}
    if isinstance(x, list):
        return [v for v in x if v is not None]
    return x


def _iscan(f, items, base, flip):
    # the reference starts from ``base`` only when it is truthy (ext.py:93-112: ``if base or started``): a base of 0
    # or None means "start from the first item"
    seeded = bool(base)
    for item in items:
        if seeded:
            base = f(item, base) if flip else f(base, item)
        else:
            base, seeded = item, True
        yield base


def iscanl(f, l, base=None):
    """Running left fold, as a generator: ``x0, f(x0, x1), f(f(x0, x1), x2), ...`` (ext.py:93-101)."""
    return _iscan(f, l, base, False)


def iscanr(f, l, base=None):
    """Running right fold over the reversed sequence: ``xn, f(xn-1, xn), ...`` (ext.py:104-112)."""
    return _iscan(f, list(l)[::-1], base, True)


def scanl(f, l, base=None):
    return list(iscanl(f, l, base))


def scanr(f, l, base=None):
    return list(iscanr(f, l, base))


class AttrDict(dict):
    """A dict whose keys are also its attributes (ext.py:151-154)."""

    def __init__(self, *args, **kwargs):
        dict.__init__(self, *args, **kwargs)
        self.__dict__ = self


def truncate_path(p, t):
    """Every array of the path cut to its first ``t`` steps (ext.py:162-163)."""
    return {k: v[:t] for k, v in p.items()}


def concat_paths(p1, p2):
    """The keys both paths have, their arrays joined along time (ext.py:166-168)."""
    return {k: np.concatenate([p1[k], p2[k]]) for k in list(p1.keys()) if k in p2}


def path_len(p):
    return len(p["states"])


def shuffled(sequence):
    """The items in a random order drawn from ``random`` (swap-with-last draw of ext.py:175-182, so a seeded
    ``random`` yields the reference's order)."""
    deck = list(sequence)
    while deck:
        i = random.randint(0, len(deck) - 1)
        card, deck[i] = deck[i], deck[-1]
        deck.pop()
        yield card


def flatten_shape_dim(shape):
    """Number of elements of an array of that shape (ext.py:302-303)."""
    return functools.reduce(operator.mul, shape, 1)


def stdize(data, eps=1e-6):
    """Columns shifted to zero mean and scaled by (std + eps) (ext.py:373-374)."""
    return (data - np.mean(data, axis=0)) / (np.std(data, axis=0) + eps)


def iterate_minibatches_generic(input_lst=None, batchsize=None, shuffle=False):
    """Minibatches ``[a[idx] for a in input_lst]`` of ``batchsize`` rows (the last one may be shorter; ``None`` = one
    batch of everything); ``shuffle`` permutes the rows once with ``np.random.shuffle`` (ext.py:377-391)."""
    n = len(input_lst[0])
    assert all(len(a) == n for a in input_lst)
    if batchsize is None:
        batchsize = n
    order = None
    if shuffle:
        order = np.arange(n)
        np.random.shuffle(order)
    for lo in range(0, n, batchsize):
        rows = order[lo:lo + batchsize] if shuffle else slice(lo, lo + batchsize)
        yield [a[rows] for a in input_lst]


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


def unflatten_tensor_variables(flatarr, shapes, symb_arrs=None):
    """The inverse: consecutive pieces of ``flatarr`` in the given shapes (ext.py:320-338; ``symb_arrs`` carried the
    Theano broadcast patterns and is accepted and ignored)."""
    out, lo = [], 0
    for shape in shapes:
        size = int(np.prod(list(shape)))
        out.append(flatarr[lo:lo + size].reshape(tuple(shape)))
        lo += size
    return out
