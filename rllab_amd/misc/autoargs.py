"""Shims for the reference's legacy CLI decorators (rllab/misc/autoargs.py).
Env modules import them; argument-parser generation is launch tooling and out of
scope, so the decorators only pass through."""


def arg(name, type=None, help=None, nargs=None, mapper=None, choices=None, prefixable=True):
    def wrap(fn):
        return fn
    return wrap


def inherit(base_func):
    def wrap(fn):
        return fn
    return wrap


def prefix(prefix_):
    def wrap(fn):
        return fn
    return wrap


def add_args(fn):
    return fn


def new_from_args(fn):
    return fn
