"""Constructor-argument pickling (mirrors rllab/core/serializable.py:5-65).

An object records the arguments it was constructed with; pickling stores only
those (``__args`` / ``__kwargs``) and unpickling re-runs the constructor.
"""
import inspect


class Serializable(object):
    def __init__(self, *args, **kwargs):
        self.__args = args
        self.__kwargs = kwargs

    def quick_init(self, locals_):
        """Capture ctor arguments from ``locals()`` (reference :11-34)."""
        if getattr(self, "_serializable_initialized", False):
            return
        spec = inspect.getfullargspec(self.__init__)
        kwargs = locals_[spec.varkw] if spec.varkw else dict()
        varargs = locals_[spec.varargs] if spec.varargs else tuple()
        in_order_args = [locals_[arg] for arg in spec.args][1:]
        self.__args = tuple(in_order_args) + tuple(varargs)
        self.__kwargs = kwargs
        setattr(self, "_serializable_initialized", True)

    def __getstate__(self):
        return {"__args": self.__args, "__kwargs": self.__kwargs}

    def __setstate__(self, d):
        out = type(self)(*d["__args"], **d["__kwargs"])
        self.__dict__.update(out.__dict__)

    @classmethod
    def clone(cls, obj, **kwargs):
        """Rebuild ``obj`` with some ctor arguments replaced (reference :44-65)."""
        assert isinstance(obj, Serializable)
        d = obj.__getstate__()
        spec = inspect.getfullargspec(obj.__init__)
        in_order_args = spec.args[1:]
        d["__args"] = list(d["__args"])
        d["__kwargs"] = dict(d["__kwargs"])
        for kw, val in kwargs.items():
            if kw in in_order_args:
                d["__args"][in_order_args.index(kw)] = val
            else:
                d["__kwargs"][kw] = val
        out = type(obj).__new__(type(obj))
        out.__setstate__(d)
        return out
