"""ZeroBaseline: predicts 0 everywhere, fits nothing (API of rllab/baselines/zero_baseline.py:6-25).
With it the advantages are the discounted returns themselves."""
import numpy as np

from rllab_amd.baselines.base import Baseline


class ZeroBaseline(Baseline):
    def __init__(self, env_spec=None):
        Baseline.__init__(self, env_spec)

    # per-path numpy face
    def predict(self, path):
        return np.zeros(len(path["rewards"]), dtype=np.asarray(path["rewards"]).dtype)

    def fit(self, paths):
        return None

    # dense face: None == an all-zero value plane (rl_gae skips the loads)
    def predict_dense(self, traj):
        return None

    def fit_dense(self, traj, all_reduce=None):
        return None

    def get_param_values(self, **tags):
        return None

    def set_param_values(self, val, **tags):
        return None
