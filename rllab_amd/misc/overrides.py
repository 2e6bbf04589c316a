"""``@overrides`` marker decorator (the reference's version, rllab/misc/overrides.py,
inspects bytecode to verify the base class; here it is a documented no-op)."""


def overrides(method):
    return method
