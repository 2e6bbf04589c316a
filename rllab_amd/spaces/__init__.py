from rllab_amd.spaces.base import Space
from rllab_amd.spaces.box import Box

__all__ = ["Space", "Box"]
