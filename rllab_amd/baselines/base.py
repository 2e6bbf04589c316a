"""What ``process_samples`` asks of a baseline (interface of rllab/baselines/base.py:4-39).

Two faces: the reference's per-path numpy calls ``fit(paths)`` / ``predict(path)``, and optional *dense*
forms the lock-step sampler prefers -- ``predict_dense(traj) -> [T, N] float64 plane | None`` and
``fit_dense(traj, all_reduce=None)`` -- which read the rollout's device planes directly."""


class Baseline(object):
    algorithm_parallelized = False   # the reference's flag for baselines that fit inside worker processes

    def __init__(self, env_spec):
        self._mdp_spec = env_spec

    def fit(self, paths):
        raise NotImplementedError("%s.fit" % type(self).__name__)

    def predict(self, path):
        raise NotImplementedError("%s.predict" % type(self).__name__)

    def get_param_values(self):
        raise NotImplementedError("%s.get_param_values" % type(self).__name__)

    def set_param_values(self, val):
        raise NotImplementedError("%s.set_param_values" % type(self).__name__)

    def log_diagnostics(self, paths):
        """Baselines have nothing to log by default."""
