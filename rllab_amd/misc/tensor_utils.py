"""List-of-dicts <-> dict-of-arrays helpers (mirrors the functions of
rllab/misc/tensor_utils.py:6-150 that sit on the sampler path).  These shuffle
host-side containers for API users; the engine itself keeps trajectories as
dense device planes and never calls them on the hot path."""
import numpy as np


def flatten_tensors(tensors):
    if len(tensors) > 0:
        return np.concatenate([np.reshape(x, [-1]) for x in tensors])
    return np.asarray([])


def unflatten_tensors(flattened, tensor_shapes):
    sizes = [int(np.prod(s)) for s in tensor_shapes]
    indices = np.cumsum(sizes)[:-1]
    return [np.reshape(chunk, shape) for chunk, shape in zip(np.split(flattened, indices), tensor_shapes)]


def pad_tensor(x, max_len, mode='zero'):
    padding = np.zeros_like(x[0])
    if mode == 'last':
        padding = x[-1]
    return np.concatenate([x, np.tile(padding, (max_len - len(x),) + (1,) * np.ndim(x[0]))])


def pad_tensor_n(xs, max_len):
    ret = np.zeros((len(xs), max_len) + xs[0].shape[1:], dtype=xs[0].dtype)
    for idx, x in enumerate(xs):
        ret[idx][:len(x)] = x
    return ret


def pad_tensor_dict(tensor_dict, max_len, mode='zero'):
    ret = dict()
    for k, v in tensor_dict.items():
        ret[k] = pad_tensor_dict(v, max_len, mode=mode) if isinstance(v, dict) else pad_tensor(v, max_len, mode=mode)
    return ret


def stack_tensor_list(tensor_list):
    return np.array(tensor_list)


def stack_tensor_dict_list(tensor_dict_list):
    ret = dict()
    for k in list(tensor_dict_list[0].keys()):
        example = tensor_dict_list[0][k]
        if isinstance(example, dict):
            ret[k] = stack_tensor_dict_list([x[k] for x in tensor_dict_list])
        else:
            ret[k] = stack_tensor_list([x[k] for x in tensor_dict_list])
    return ret


def concat_tensor_list(tensor_list):
    return np.concatenate(tensor_list, axis=0)


def concat_tensor_dict_list(tensor_dict_list):
    ret = dict()
    for k in list(tensor_dict_list[0].keys()):
        example = tensor_dict_list[0][k]
        if isinstance(example, dict):
            ret[k] = concat_tensor_dict_list([x[k] for x in tensor_dict_list])
        else:
            ret[k] = concat_tensor_list([x[k] for x in tensor_dict_list])
    return ret


def split_tensor_dict_list(tensor_dict):
    ret = None
    for k in list(tensor_dict.keys()):
        vals = tensor_dict[k]
        if isinstance(vals, dict):
            vals = split_tensor_dict_list(vals)
        if ret is None:
            ret = [{k: v} for v in vals]
        else:
            for v, cur in zip(vals, ret):
                cur[k] = v
    return ret


def truncate_tensor_list(tensor_list, truncated_len):
    return tensor_list[:truncated_len]


def truncate_tensor_dict(tensor_dict, truncated_len):
    ret = dict()
    for k, v in tensor_dict.items():
        ret[k] = truncate_tensor_dict(v, truncated_len) if isinstance(v, dict) else truncate_tensor_list(v, truncated_len)
    return ret
