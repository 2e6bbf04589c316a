"""``rllab`` import-path alias of ``rllab_amd``.

``examples/trpo_cartpole.py`` / ``examples/trpo_swimmer.py`` of rll/rllab import
``rllab.algos.trpo``, ``rllab.envs.normalized_env`` ...; this package makes those
paths resolve to the SAME module objects as ``rllab_amd.algos.trpo`` etc., so the
scripts run unchanged on the MI355X-native engine.
"""
import importlib
import importlib.abc
import importlib.util
import sys

import rllab_amd

__path__ = []  # no real submodules: everything is aliased


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    PREFIX = "rllab."

    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(self.PREFIX):
            return None
        real = "rllab_amd." + fullname[len(self.PREFIX):]
        try:
            if importlib.util.find_spec(real) is None:
                return None
        except (ImportError, ValueError):
            return None
        return importlib.util.spec_from_loader(fullname, self, is_package=True)

    def create_module(self, spec):
        return importlib.import_module("rllab_amd." + spec.name[len(self.PREFIX):])

    def exec_module(self, module):
        pass


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())

__version__ = rllab_amd.__version__
