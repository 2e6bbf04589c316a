"""SwimmerEnv (API of rllab/envs/mujoco/swimmer_env.py:10-62); dynamics in
csrc/dyn_swimmer.h (``rl::Swimmer``, a swimmer-style planar 3-link chain built from
the constants of vendor/mujoco_models/swimmer.xml)."""
from rllab_amd import _lib
from rllab_amd.core.serializable import Serializable
from rllab_amd.envs.mujoco.mujoco_env import MujocoEnv


class SwimmerEnv(MujocoEnv, Serializable):
    FILE = 'swimmer.xml'
    ORI_IND = 2
    KIND = _lib.ENV_SWIMMER
    OBS_ENDS_WITH_TORSO_COM = True     # obs = [..., com_subtree(torso)] (get_body_com)

    PLANE = "xy"

    def __init__(self, ctrl_cost_coeff=1e-2, *args, **kwargs):
        self.ctrl_cost_coeff = ctrl_cost_coeff
        Serializable.quick_init(self, locals())
        super(SwimmerEnv, self).__init__(*args, ctrl_cost_coeff=float(ctrl_cost_coeff), **kwargs)

    def get_ori(self):
        """Heading of the first link: qpos[ORI_IND] (swimmer_env.py:32-33)."""
        return float(self.get_current_obs()[self.ORI_IND])

    def log_diagnostics(self, paths):
        self._log_forward_progress(paths)
