"""Walker2DEnv (API of rllab/envs/mujoco/walker2d_env.py:15-59); dynamics in csrc/dyn_walker.h
(``rl::Walker2D``, a walker-style planar 7-body / 9-DoF biped built from the constants of
vendor/mujoco_models/walker2d.xml)."""
from rllab_amd import _lib
from rllab_amd.core.serializable import Serializable
from rllab_amd.envs.mujoco.mujoco_env import MujocoEnv


class Walker2DEnv(MujocoEnv, Serializable):
    FILE = 'walker2d.xml'
    KIND = _lib.ENV_WALKER2D
    OBS_ENDS_WITH_TORSO_COM = True     # obs = [..., com_subtree(torso)] (get_body_com)

    def __init__(self, ctrl_cost_coeff=1e-2, limit_model="penalty", contact_model="penalty", *args, **kwargs):
        """``limit_model`` / ``contact_model``: "penalty" (default) or "mujoco" (MujocoEnv._constraint_flags;
        vendor/mujoco_models/walker2d.xml:6 sets no solver parameter: MuJoCo's defaults)."""
        self.ctrl_cost_coeff = ctrl_cost_coeff
        self.limit_model, self.contact_model = limit_model, contact_model
        Serializable.quick_init(self, locals())
        super(Walker2DEnv, self).__init__(*args, ctrl_cost_coeff=float(ctrl_cost_coeff),
                                          **self._constraint_flags(limit_model, contact_model, kwargs))

    def log_diagnostics(self, paths):
        self._log_forward_progress(paths)
