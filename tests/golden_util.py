import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def unpack_paths(g):
    """Rebuild the list of path dicts stored by oracle/make_golden.py::pack."""
    paths, o = [], 0
    for L in g["lens"]:
        L = int(L)
        paths.append(dict(observations=g["observations"][o:o + L].copy(), actions=g["actions"][o:o + L].copy(),
                          rewards=g["rewards"][o:o + L].copy(),
                          agent_infos=dict(mean=g["mean"][o:o + L].copy(), log_std=g["log_std"][o:o + L].copy()),
                          env_infos=dict()))
        o += L
    return paths


STAT_KEYS = ['AverageDiscountedReturn', 'AverageReturn', 'ExplainedVariance', 'NumTrajs', 'Entropy',
             'Perplexity', 'StdReturn', 'MaxReturn', 'MinReturn']
