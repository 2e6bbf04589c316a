"""Box2DEnv base (API of rllab/envs/box2d/box2d_env.py:30-321).

The reference builds a pybox2d ``b2World`` from an XML template and steps it
through SWIG; here the world of each task is compiled into a HIP kernel
(csrc/dyn_*.h) and this class only carries the constructor surface.  Options
that change the simulated world (frame_skip other than the task default,
position_only, obs/action noise, custom templates) are rejected loudly instead
of being silently ignored.
"""
from rllab_amd.envs.hip_env import HipEnv


class Box2DEnv(HipEnv):
    DEFAULT_FRAME_SKIP = 1

    def __init__(self, model_path=None, frame_skip=None, position_only=False, obs_noise=0.0,
                 action_noise=0.0, template_string=None, template_args=None):
        if frame_skip is None:
            frame_skip = self.DEFAULT_FRAME_SKIP
        unsupported = []
        if frame_skip != self.DEFAULT_FRAME_SKIP:
            unsupported.append("frame_skip=%r" % (frame_skip,))
        if position_only:
            unsupported.append("position_only")
        if obs_noise != 0.0:
            unsupported.append("obs_noise")
        if action_noise != 0.0:
            unsupported.append("action_noise")
        if template_string is not None or template_args not in (None, {}):
            unsupported.append("template_string/template_args")
        if unsupported:
            raise NotImplementedError(
                "%s: options %s are not compiled into the HIP kernel of this env" %
                (type(self).__name__, ", ".join(unsupported)))
        self.frame_skip = frame_skip
        self.position_only = position_only
        self.obs_noise = obs_noise
        self.action_noise = action_noise
        HipEnv.__init__(self)
