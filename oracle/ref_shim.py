"""Import shim for running the UNMODIFIED reference modules (TEST INFRASTRUCTURE).

The reference's pure-Python hot-path modules (sampler, stateful_pool, rollout,
process_samples, NormalizedEnv, Box, krylov, ...) import -- at module level only --
third-party packages that are not installed here and cannot be (no network): theano,
lasagne, Box2D, mako, pyprind, cached_property, path, pygame, tensorflow.  None of the functions
this repo exercises calls into them, so stub modules are enough.  Two spelling fixes
for modern libraries: joblib's ``MemmapingPool`` (stateful_pool.py:1) and ``_ast.Num``
(conjugate_gradient_optimizer.py:10).

``install(root)`` puts ``root`` (``/root/reference`` in the build container, or the
staged copy ``oracle/_ref`` on the GPU box; see oracle/make_ref.py) FIRST on sys.path,
ahead of this repo's own ``rllab`` alias package, so ``import rllab`` resolves to the
reference.  Never call it in a process that also uses the product's ``rllab`` alias.
"""
import sys
import types

STUBS = ["theano", "theano.tensor", "theano.tensor.nnet", "theano.tensor.signal",
         "theano.tensor.signal.pool", "theano.tensor.extra_ops", "theano.ifelse",
         "theano.sandbox", "theano.sandbox.rng_mrg", "theano.gradient", "theano.compile",
         "theano.tensor.shared_randomstreams",
         "lasagne", "lasagne.layers", "lasagne.nonlinearities", "lasagne.init",
         "lasagne.updates", "lasagne.utils", "lasagne.random",
         "Box2D", "pygame", "pygame.locals", "mako", "mako.template", "mako.lookup", "pyprind",
         "path",
         # sandbox/rocky/tf/misc/tensor_utils.py imports it at module level; VecEnvExecutor uses the numpy helpers only
         "tensorflow"]


class _Anything(types.ModuleType):
    """A module whose every attribute is another such module and which can be called."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        sub = _Anything(self.__name__ + "." + name)
        setattr(self, name, sub)
        return sub

    def __call__(self, *a, **k):
        return _Anything("call")


def install(root=None):
    import _ast
    import ast
    if not hasattr(_ast, "Num"):
        _ast.Num = ast.Constant
    for name in STUBS:
        if name not in sys.modules:
            sys.modules[name] = _Anything(name)
    cp = types.ModuleType("cached_property")
    cp.cached_property = property
    sys.modules["cached_property"] = cp
    import joblib.pool
    if not hasattr(joblib.pool, "MemmapingPool"):
        joblib.pool.MemmapingPool = joblib.pool.MemmappingPool
    if root is not None:
        if root in sys.path:
            sys.path.remove(root)
        sys.path.insert(0, root)
