"""ZeroBaseline (mirrors rllab/baselines/zero_baseline.py:6-25)."""
import numpy as np

from rllab_amd.baselines.base import Baseline


class ZeroBaseline(Baseline):
    def __init__(self, env_spec):
        pass

    def get_param_values(self, **kwargs):
        return None

    def set_param_values(self, val, **kwargs):
        pass

    def fit(self, paths):
        pass

    def predict(self, path):
        return np.zeros_like(path["rewards"])

    # dense-batch forms used by the vectorised sampler
    def predict_dense(self, traj):
        return None

    def fit_dense(self, traj, all_reduce=None):
        pass
