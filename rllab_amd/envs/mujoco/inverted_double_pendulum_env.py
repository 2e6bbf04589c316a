"""InvertedDoublePendulumEnv (API of rllab/envs/mujoco/inverted_double_pendulum_env.py:10-58); dynamics in
csrc/dyn_idp.h (``rl::InvertedDoublePendulum``: cart on a rail with two hinged capsule poles, built from the bodies of
vendor/mujoco_models/inverted_double_pendulum.xml.mako with noise = False)."""
from rllab_amd import _lib
from rllab_amd.core.serializable import Serializable
from rllab_amd.envs.mujoco.mujoco_env import MujocoEnv


class InvertedDoublePendulumEnv(MujocoEnv, Serializable):
    FILE = 'inverted_double_pendulum.xml.mako'
    KIND = _lib.ENV_INVERTED_DOUBLE_PENDULUM
    progress_obs_index = None      # no forward-progress diagnostic (the reference class logs none)

    def __init__(self, *args, **kwargs):
        Serializable.quick_init(self, locals())
        kwargs = dict(kwargs)
        self.random_start = kwargs.pop("random_start", True)
        super(InvertedDoublePendulumEnv, self).__init__(*args, flags=0 if self.random_start else _lib.CFG_FIXED_START,
                                                        **kwargs)
