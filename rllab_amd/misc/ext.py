"""The few helpers of rllab/misc/ext.py the hot path uses: ``extract`` (:14-20),
``set_seed`` (:188-206), ``sliced_fun`` (:341-370), ``lazydict``."""
import random

import numpy as np

seed_ = None


def extract(x, *keys):
    if isinstance(x, dict):
        return tuple(x[k] for k in keys)
    elif isinstance(x, list):
        return tuple([xi[k] for xi in x] for k in keys)
    raise NotImplementedError


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


class lazydict(object):
    def __init__(self, **kwargs):
        self._lazy_dict = kwargs
        self._dict = {}

    def __getitem__(self, key):
        if key not in self._dict:
            self._dict[key] = self._lazy_dict[key]()
        return self._dict[key]

    def set(self, key, value):
        self._lazy_dict[key] = value


def sliced_fun(f, n_slices):
    """Evaluate ``f`` on ``n_slices`` slices of the sample axis and return the
    sample-weighted mean (reference :341-370).  With ``n_slices == 1`` it is a
    plain call."""
    def sliced_f(sliced_inputs, non_sliced_inputs=None):
        if non_sliced_inputs is None:
            non_sliced_inputs = []
        non_sliced_inputs = list(non_sliced_inputs)
        n_paths = len(sliced_inputs[0])
        slice_size = max(1, n_paths // n_slices)
        ret_vals = None
        was_tuple = was_seq = False
        for start in range(0, n_paths, slice_size):
            inputs_slice = [v[start:start + slice_size] for v in sliced_inputs]
            out = f(*(inputs_slice + non_sliced_inputs))
            was_seq = isinstance(out, (tuple, list))
            was_tuple = isinstance(out, tuple)
            outs = list(out) if was_seq else [out]
            scaled = [np.asarray(v) * len(inputs_slice[0]) for v in outs]
            ret_vals = scaled if ret_vals is None else [x + y for x, y in zip(ret_vals, scaled)]
        ret_vals = [v / n_paths for v in ret_vals]
        if not was_seq:
            return ret_vals[0]
        return tuple(ret_vals) if was_tuple else ret_vals
    return sliced_f


def is_iterable(obj):
    """rllab/misc/ext.py:209-210."""
    return isinstance(obj, str) or getattr(obj, '__iter__', False)


def flatten_tensor_variables(ts):
    """One flat vector from a list of tensors (rllab/misc/ext.py:297-299, which concatenates flattened Theano
    variables; here torch tensors, order preserved, graph kept)."""
    import torch
    return torch.cat([torch.reshape(t, (-1,)) for t in ts])


# -- small generic helpers scripts written against rllab/misc/ext.py use -----------------------------------------
class AttrDict(dict):
    """dict whose items are also attributes (ext.py:42-46)."""

    def __init__(self, *args, **kwargs):
        super(AttrDict, self).__init__(*args, **kwargs)
        self.__dict__ = self


def extract_dict(x, *keys):
    return {k: x[k] for k in keys if k in x}


def compact(x):
    """Drop the None entries of a dict / list (ext.py:32-39)."""
    if isinstance(x, dict):
        return {k: v for k, v in x.items() if v is not None}
    if isinstance(x, list):
        return [v for v in x if v is not None]
    return x


def flatten(xs):
    """One level of nesting removed."""
    return [x for group in xs for x in group]


def shuffled(sequence):
    """Generator over a random permutation of the sequence (np.random)."""
    import numpy as np
    for i in np.random.permutation(len(sequence)):
        yield sequence[i]


def path_len(p):
    return len(p["states"]) if "states" in p else len(p["rewards"])


def concat_paths(p1, p2):
    import numpy as np
    return {k: np.concatenate([p1[k], p2[k]]) for k in p1.keys() if k in p2}


def truncate_path(p, t):
    return {k: v[:t] for k, v in p.items()}


def iterate_minibatches_generic(input_lst=None, batchsize=None, shuffle=False):
    """Aligned mini-batches of several arrays (ext.py:158-176); batchsize None = one batch of everything."""
    import numpy as np
    n = len(input_lst[0])
    if batchsize is None:
        batchsize = n
    assert all(len(x) == n for x in input_lst)
    order = np.random.permutation(n) if shuffle else np.arange(n)
    for start in range(0, n, batchsize):
        sel = order[start:start + batchsize]
        yield [x[sel] for x in input_lst]


def stdize(data, eps=1e-6):
    import numpy as np
    return (data - np.mean(data, axis=0)) / (np.std(data, axis=0) + eps)


def scanl(f, xs, init):
    """Running left fold: [init, f(init, x0), f(f(init, x0), x1), ...]."""
    out = [init]
    for x in xs:
        out.append(f(out[-1], x))
    return out


def scanr(f, xs, init):
    """Running right fold, aligned like the reference: result[i] folds xs[i:]; result[-1] = init."""
    out = [init]
    for x in reversed(list(xs)):
        out.append(f(x, out[-1]))
    return out[::-1]


def unflatten_tensor_variables(flatarr, shapes, symb_arrs=None):
    """Inverse of flatten_tensor_variables for torch tensors (ext.py:302-311)."""
    out, n = [], 0
    for shape in shapes:
        size = 1
        for s in shape:
            size *= int(s)
        out.append(flatarr[n:n + size].reshape(tuple(shape)))
        n += size
    return out
