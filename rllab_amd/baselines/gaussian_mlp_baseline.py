"""GaussianMLPBaseline (API of rllab/baselines/gaussian_mlp_baseline.py:10-47): the neural value
function of the rllab benchmark paper -- a GaussianMLPRegressor from observations to returns.  ``fit`` /
``predict`` keep the reference's per-path numpy semantics; the sampler uses the dense forms, which feed
the device planes of the rollout straight into the regressor (no host copies, no concatenation)."""
import numpy as np
import torch

from rllab_amd.baselines.base import Baseline
from rllab_amd.core.parameterized import Parameterized
from rllab_amd.core.serializable import Serializable
from rllab_amd.regressors.gaussian_mlp_regressor import GaussianMLPRegressor


class GaussianMLPBaseline(Baseline, Parameterized):
    def __init__(self, env_spec, subsample_factor=1., num_seq_inputs=1, regressor_args=None):
        Serializable.quick_init(self, locals())
        Baseline.__init__(self, env_spec)
        Parameterized.__init__(self)
        obs_dim = env_spec.observation_space.flat_dim * num_seq_inputs
        self._regressor = GaussianMLPRegressor(input_shape=(obs_dim,), output_dim=1, name="vf",
                                               **(regressor_args or {}))

    # -- per-path numpy face (reference semantics) -------------------------------------------------------
    def fit(self, paths):
        if hasattr(paths, "traj"):           # lazy PathList of a dense batch: stay on the device
            return self.fit_dense(paths.traj)
        xs = np.concatenate([p["observations"] for p in paths])
        ys = np.concatenate([p["returns"] for p in paths])
        self._regressor.fit(xs, ys[:, None])

    def predict(self, path):
        return self._regressor.predict(path["observations"])[:, 0]

    # -- dense device face ------------------------------------------------------------------------------------
    def predict_dense(self, traj):
        """[T, N] float64 plane of value predictions."""
        values = self._regressor.predict_planes(traj.obs.reshape(traj.obs_dim, -1))
        return values.reshape(traj.T, traj.N).to(torch.float64)

    def fit_dense(self, traj, all_reduce=None):
        weights = None if traj.valid is None else traj.valid.reshape(-1)
        self._regressor.fit_planes(traj.obs.reshape(traj.obs_dim, -1), traj.returns.reshape(1, -1), weights)

    # -- Parameterized: the regressor's parameters are the baseline's ---------------------------------------------
    def get_params_internal(self, **tags):
        return self._regressor.get_params_internal(**tags)

    def get_param_values(self, **tags):
        return self._regressor.get_param_values(**tags)

    def set_param_values(self, flattened_params, **tags):
        self._regressor.set_param_values(flattened_params, **tags)
