"""Policy interfaces (mirror rllab/policies/base.py:4-81 plus the vectorised
extensions of sandbox/rocky/tf/policies/base.py:17-29)."""
from rllab_amd.core.parameterized import Parameterized


class Policy(Parameterized):
    def __init__(self, env_spec):
        Parameterized.__init__(self)
        self._env_spec = env_spec

    def get_action(self, observation):
        raise NotImplementedError

    def get_actions(self, observations):
        raise NotImplementedError

    def reset(self, dones=None):
        pass

    @property
    def vectorized(self):
        return False

    @property
    def observation_space(self):
        return self._env_spec.observation_space

    @property
    def action_space(self):
        return self._env_spec.action_space

    @property
    def env_spec(self):
        return self._env_spec

    @property
    def recurrent(self):
        return False

    def log_diagnostics(self, paths):
        pass

    @property
    def state_info_keys(self):
        return list()

    def terminate(self):
        pass


class StochasticPolicy(Policy):
    @property
    def distribution(self):
        raise NotImplementedError

    def dist_info_sym(self, obs_var, state_info_vars):
        raise NotImplementedError

    def dist_info(self, obs, state_infos):
        raise NotImplementedError
