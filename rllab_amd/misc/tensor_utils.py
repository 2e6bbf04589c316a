"""Host-side container shuffling for API users: flat <-> shaped parameter lists, and the
list-of-dicts <-> dict-of-arrays conversions of path data (API of rllab/misc/tensor_utils.py:6-150).
Nested dicts are handled by one recursive mapper.  The engine keeps trajectories as dense device planes
and never calls these on the hot path."""
import numpy as np


def _map_leaves(fn, tree):
    """Apply ``fn`` to every non-dict leaf of a (possibly nested) dict."""
    return {k: (_map_leaves(fn, v) if isinstance(v, dict) else fn(v)) for k, v in tree.items()}


def _zip_leaves(fn, trees):
    """Combine a list of identically-keyed nested dicts leaf by leaf: ``fn`` receives the list of leaves."""
    first = trees[0]
    return {k: (_zip_leaves(fn, [t[k] for t in trees]) if isinstance(first[k], dict) else fn([t[k] for t in trees]))
            for k in first}


# -- flat parameter vectors -----------------------------------------------------------------------
def flatten_tensors(tensors):
    tensors = list(tensors)
    if not tensors:
        return np.asarray([])
    return np.concatenate([np.asarray(t).ravel() for t in tensors])


def unflatten_tensors(flattened, tensor_shapes):
    flattened = np.asarray(flattened)
    out, start = [], 0
    for shape in tensor_shapes:
        n = int(np.prod(shape))
        out.append(flattened[start:start + n].reshape(shape))
        start += n
    return out


# -- padding --------------------------------------------------------------------------------------
def pad_tensor(x, max_len, mode='zero'):
    x = np.asarray(x)
    fill = x[-1] if mode == 'last' else np.zeros_like(x[0])
    tail = np.broadcast_to(fill, (max_len - len(x),) + x.shape[1:])
    return np.concatenate([x, tail])


def pad_tensor_n(xs, max_len):
    out = np.zeros((len(xs), max_len) + xs[0].shape[1:], dtype=xs[0].dtype)
    for row, x in zip(out, xs):
        row[:len(x)] = x
    return out


def pad_tensor_dict(tensor_dict, max_len, mode='zero'):
    return _map_leaves(lambda v: pad_tensor(v, max_len, mode=mode), tensor_dict)


# -- stacking / concatenating / splitting / truncating -----------------------------------------------
def stack_tensor_list(tensor_list):
    return np.array(tensor_list)


def stack_tensor_dict_list(tensor_dict_list):
    if not tensor_dict_list:
        return dict()
    return _zip_leaves(stack_tensor_list, tensor_dict_list)


def concat_tensor_list(tensor_list):
    return np.concatenate(tensor_list, axis=0)


def concat_tensor_dict_list(tensor_dict_list):
    return _zip_leaves(concat_tensor_list, tensor_dict_list)


def split_tensor_dict_list(tensor_dict):
    """dict of arrays (leading axis n) -> list of n dicts; ``None`` for an empty dict, as the reference."""
    if not tensor_dict:
        return None
    columns = {k: (split_tensor_dict_list(v) if isinstance(v, dict) else v) for k, v in tensor_dict.items()}
    n = len(next(iter(columns.values())))
    return [{k: col[i] for k, col in columns.items()} for i in range(n)]


def truncate_tensor_list(tensor_list, truncated_len):
    return tensor_list[:truncated_len]


def truncate_tensor_dict(tensor_dict, truncated_len):
    return _map_leaves(lambda v: v[:truncated_len], tensor_dict)


def flatten_first_axis_tensor_dict(tensor_dict):
    """Merge the two leading axes of every leaf: [a, b, ...] -> [a * b, ...]."""
    return _map_leaves(lambda v: v.reshape((-1,) + v.shape[2:]), tensor_dict)


def concat_tensor_list_subsample(tensor_list, f):
    """Concatenation of a random fraction ``f`` (rounded up, without replacement, np.random) of every tensor's rows."""
    picks = [t[np.random.choice(len(t), int(np.ceil(len(t) * f)), replace=False)] for t in tensor_list]
    return np.concatenate(picks, axis=0)


def concat_tensor_dict_list_subsample(tensor_dict_list, f):
    return _zip_leaves(lambda leaves: concat_tensor_list_subsample(leaves, f), tensor_dict_list)


def high_res_normalize(probs):
    """Probabilities rescaled to sum to one in Python float arithmetic (np.random.multinomial is picky)."""
    vals = [float(p) for p in probs]
    total = sum(vals)
    return [v / total for v in vals]
