"""Env interface and the ``Step`` tuple (mirrors rllab/envs/base.py:6-100)."""
import collections

from rllab_amd.envs.env_spec import EnvSpec


class Env(object):
    def step(self, action):
        raise NotImplementedError

    def reset(self):
        raise NotImplementedError

    @property
    def action_space(self):
        raise NotImplementedError

    @property
    def observation_space(self):
        raise NotImplementedError

    @property
    def action_dim(self):
        return self.action_space.flat_dim

    def render(self):
        pass

    def log_diagnostics(self, paths):
        pass

    @property
    def spec(self):
        spec = self.__dict__.get("_cached_spec")
        if spec is None:
            spec = EnvSpec(observation_space=self.observation_space, action_space=self.action_space)
            self.__dict__["_cached_spec"] = spec
        return spec

    @property
    def horizon(self):
        raise NotImplementedError

    def terminate(self):
        pass

    def get_param_values(self):
        return None

    def set_param_values(self, params):
        pass

    # vectorised boundary (precedent: sandbox/rocky/tf/envs/base.py:49-54)
    @property
    def vectorized(self):
        return False

    def vec_env_executor(self, n_envs, max_path_length):
        raise NotImplementedError


_Step = collections.namedtuple("Step", ["observation", "reward", "done", "info"])


def Step(observation, reward, done, **kwargs):
    return _Step(observation, reward, done, kwargs)
