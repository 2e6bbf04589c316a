"""CartpoleSwingupEnv (API of rllab/envs/box2d/cartpole_swingup_env.py:14-61); dynamics in
csrc/dyn_cartpole.h (``rl::CartpoleSwingup``: the Cartpole world started hanging down, reward
cos(pole angle), -100 and done beyond |x| = 3)."""
from rllab_amd import _lib
from rllab_amd.core.serializable import Serializable
from rllab_amd.envs.box2d.box2d_env import Box2DEnv


class CartpoleSwingupEnv(Box2DEnv, Serializable):
    KIND = _lib.ENV_CARTPOLE_SWINGUP
    POSITION_IDS = (0, 2)

    def __init__(self, *args, **kwargs):
        Serializable.quick_init(self, locals())
        kwargs = dict(kwargs)
        follows = bool(kwargs.pop("reset_pole_follows_cart", False))
        super(CartpoleSwingupEnv, self).__init__(None, *args, flags=_lib.CFG_POLE_FOLLOWS_CART if follows else 0,
                                                 **kwargs)
        self.max_cart_pos = 3
        self.max_reward_cart_pos = 3
