"""Constructor-argument pickling (behaviour of rllab/core/serializable.py:5-65).

A ``Serializable`` object remembers the arguments it was built with; its pickle is just those arguments
(``{"__args": ..., "__kwargs": ...}``) and unpickling runs the constructor again.  Classes opt in by
calling ``Serializable.quick_init(self, locals())`` at the top of ``__init__`` (or
``Serializable.__init__(self, *args, **kwargs)``).  ``Parameterized`` adds the flat parameter vector on top,
which is the whole snapshot format of policies and baselines.
"""
import inspect


def _ctor_signature(obj):
    spec = inspect.getfullargspec(obj.__init__)
    return spec.args[1:], spec.varargs, spec.varkw          # positional names without `self`


class Serializable(object):
    def __init__(self, *args, **kwargs):
        self.__args, self.__kwargs = args, kwargs

    def quick_init(self, locals_):
        """Record the constructor call from the ``locals()`` of ``__init__``.  Only the first call counts:
        a subclass's quick_init wins over the ones its bases issue later."""
        if getattr(self, "_serializable_initialized", False):
            return
        names, varargs, varkw = _ctor_signature(self)
        positional = tuple(locals_[n] for n in names)
        if varargs:
            positional += tuple(locals_[varargs])
        self.__args = positional
        self.__kwargs = dict(locals_[varkw]) if varkw else dict()
        self._serializable_initialized = True

    def __getstate__(self):
        return {"__args": self.__args, "__kwargs": self.__kwargs}

    def __setstate__(self, state):
        rebuilt = type(self)(*state["__args"], **state["__kwargs"])
        self.__dict__.update(rebuilt.__dict__)

    @classmethod
    def clone(cls, obj, **overrides):
        """A fresh object built like ``obj`` with some constructor arguments replaced."""
        assert isinstance(obj, Serializable)
        state = dict(obj.__getstate__())          # subclasses may carry extra entries (e.g. running statistics)
        args, kwargs = list(state["__args"]), dict(state["__kwargs"])
        names, _, _ = _ctor_signature(obj)
        for key, val in overrides.items():
            if key in names:
                args[names.index(key)] = val
            else:
                kwargs[key] = val
        twin = type(obj).__new__(type(obj))
        state["__args"], state["__kwargs"] = tuple(args), kwargs
        twin.__setstate__(state)
        return twin
