"""EnvSpec: the (observation space, action space) pair policies and baselines are built from
(API of rllab/envs/env_spec.py:5-25)."""
from rllab_amd.core.serializable import Serializable


class EnvSpec(Serializable):
    __slots__ = ()

    def __init__(self, observation_space, action_space):
        Serializable.quick_init(self, locals())
        self._spaces = (observation_space, action_space)

    observation_space = property(lambda self: self._spaces[0])
    action_space = property(lambda self: self._spaces[1])

    def __repr__(self):
        return "EnvSpec(obs=%r, act=%r)" % self._spaces
