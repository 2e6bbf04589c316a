"""``params`` and ``such`` with the semantics the reference's tests rely on, for collection by pytest."""
import functools
import unittest


def params(*cases):
    """nose2.tools.params: one call of the test per case (a tuple = positional arguments, anything else = one
    argument).  Collected by pytest as ONE test that runs every case and reports the failing one."""
    def deco(fn):
        @functools.wraps(fn)
        def run_all():
            for case in cases:
                args = case if isinstance(case, tuple) else (case,)
                try:
                    fn(*args)
                except Exception as err:       # name the case in the failure
                    raise AssertionError("%s%r failed: %r" % (fn.__name__, args, err)) from err
        del run_all.__wrapped__                # pytest must not see the parameters of the wrapped function
        return run_all
    return deco


class _Scenario(unittest.TestCase):
    """``with such.A("...") as it:`` -- the layer object: ``it.should`` registers tests, ``it.assertX`` are unittest's."""

    def __init__(self, description):
        super(_Scenario, self).__init__("__init__")
        self.description = description
        self.cases = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def should(self, arg):
        def register(fn):
            self.cases.append(fn)
            return fn
        if callable(arg):                      # bare ``@it.should``
            return register(arg)
        return register                        # ``@it.should("description")``

    def createTests(self, namespace):
        for k, fn in enumerate(self.cases):
            name = fn.__name__ if fn.__name__.startswith("test") else "test_%s" % fn.__name__
            namespace.setdefault(name, fn)


class such(object):
    A = _Scenario
