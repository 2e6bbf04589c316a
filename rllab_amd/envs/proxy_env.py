"""ProxyEnv: an env that stands in front of another one (API of rllab/envs/proxy_env.py:5-45).

Everything the wrapper does not define itself is answered by the wrapped env: the methods of the
``Env`` interface are bound below from one table, and any other attribute (``vectorized``,
``progress_obs_index``, ``KIND`` ... of the HIP-native envs) falls through ``__getattr__``.
"""
from rllab_amd.core.serializable import Serializable
from rllab_amd.envs.base import Env

# Env-interface members a proxy answers with the inner env's implementation
_FORWARDED_CALLS = ("reset", "step", "render", "log_diagnostics", "terminate", "get_param_values",
                    "set_param_values")
_FORWARDED_VALUES = ("action_space", "observation_space", "horizon")


class ProxyEnv(Env, Serializable):
    def __init__(self, wrapped_env):
        Serializable.quick_init(self, locals())
        self._wrapped_env = wrapped_env

    @property
    def wrapped_env(self):
        return self._wrapped_env

    def __getattr__(self, name):
        # only reached when normal lookup fails; never forward private / pickling hooks
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.__dict__["_wrapped_env"], name)


def _call_through(name):
    def method(self, *args, **kwargs):
        return getattr(self._wrapped_env, name)(*args, **kwargs)
    method.__name__ = name
    return method


for _n in _FORWARDED_CALLS:
    setattr(ProxyEnv, _n, _call_through(_n))
for _n in _FORWARDED_VALUES:
    setattr(ProxyEnv, _n, property(lambda self, _n=_n: getattr(self._wrapped_env, _n)))
