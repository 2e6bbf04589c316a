"""Experiment launching, local mode (public entry points of rllab/misc/instrument.py: ``stub``, ``concretize``,
``run_experiment_lite``; :30-163, 290-297, 338-470, 1340-1395 + scripts/run_experiment_lite.py:21-139).

``run_experiment_lite(task, n_parallel=..., snapshot_mode=..., seed=...)`` sets up what the
reference's ``scripts/run_experiment_lite.py`` sets up -- seed, ``data/local/<prefix>/<name>/``
with ``progress.csv`` / ``debug.log`` / ``params.json`` / ``variant.json``, snapshot dir, mode and
gap, log prefix -- and then runs the task IN THIS PROCESS: either a plain callable
``task(variant)`` (the reference's cloudpickle path) or a stubbed method call built with
``stub(globals())``.  The reference serialises the call and spawns ``python
scripts/run_experiment_lite.py`` so that it can also ship it to docker / EC2 / Kubernetes; those modes
are outside the hot path and raise.  ``n_parallel`` sizes the CPU worker pool of the generic-env sampler
(``sampler/parallel_sampler.py``), as in the reference; HIP-native envs are sampled by the lock-step GPU
sampler and never touch it.
"""
import datetime
import inspect
import json
import os
import os.path as osp
import uuid

from rllab_amd import config
from rllab_amd.misc import ext, logger


# ---- lazy experiment descriptions ---------------------------------------------------------------------------
# ``stub(globals())`` swaps every class of a script for a ``LazyClass``; "constructing" objects and "calling"
# methods on them then only records a small expression tree, which ``concretize`` evaluates later (here: in this
# process, right before the run).  Four node types:
#     LazyClass(cls)                       the class itself
#     LazyObject(cls, kwargs)              cls(**kwargs), built once and memoised
#     LazyAttr(target, name)               getattr(target, name)
#     LazyCall(target, method, args, kw)   getattr(target, method)(*args, **kw)
# (the reference's counterpart: rllab/misc/instrument.py:30-163,1340-1395; only the public entry points ``stub``,
# ``concretize`` and ``run_experiment_lite`` are contract)
class Lazy(object):
    """Base of the expression nodes.  Unknown attributes become LazyAttr nodes; a LazyAttr that is called
    becomes a LazyCall; a few operators are recorded the same way."""
    _own = ()

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return LazyAttr(self, name)

    def _record(self, method, *args):
        return LazyCall(self, method, list(args), {})

    def __getitem__(self, key):
        return self._record("__getitem__", key)

    def __add__(self, other):
        return self._record("__add__", other)

    def __rmul__(self, other):
        return self._record("__rmul__", other)

    def __pow__(self, power, modulo=None):
        return self._record("__pow__", power, modulo)

    def __getstate__(self):
        return {k: self.__dict__[k] for k in self._own}

    def __setstate__(self, state):
        self.__dict__.update(state)

    def __repr__(self):
        return "%s(%s)" % (type(self).__name__, ", ".join(repr(self.__dict__[k]) for k in self._own))


def _ctor_kwargs(cls, args, kwargs):
    """Fold positional constructor arguments into keywords (snapshots and variants are keyed by name)."""
    if not args:
        return dict(kwargs)
    names = inspect.getfullargspec(cls.__init__).args[1:]
    return dict(zip(names, args), **kwargs)


class LazyClass(Lazy):
    _own = ("cls",)

    def __init__(self, cls):
        self.__dict__["cls"] = cls

    def __call__(self, *args, **kwargs):
        return LazyObject(self.cls, _ctor_kwargs(self.cls, args, kwargs))

    def __getattr__(self, name):
        if not hasattr(self.__dict__["cls"], name):
            raise AttributeError(name)
        return Lazy.__getattr__(self, name)


class LazyObject(Lazy):
    _own = ("cls", "kwargs")

    def __init__(self, cls, kwargs):
        self.__dict__.update(cls=cls, kwargs=kwargs)

    def __getattr__(self, name):
        if not hasattr(self.__dict__["cls"], name):
            raise AttributeError("%s has no attribute %r" % (self.__dict__["cls"].__name__, name))
        return Lazy.__getattr__(self, name)


class LazyAttr(Lazy):
    _own = ("target", "name")

    def __init__(self, target, name):
        self.__dict__.update(target=target, name=name)

    def __call__(self, *args, **kwargs):
        return LazyCall(self.target, self.name, list(args), kwargs)


class LazyCall(Lazy):
    _own = ("target", "method", "args", "kwargs")

    def __init__(self, target, method, args, kwargs):
        self.__dict__.update(target=target, method=method, args=args, kwargs=kwargs)


def stub(glbs):
    """Replace every class in ``glbs`` by a LazyClass: constructor calls then build a lazy description of the
    experiment instead of the objects themselves."""
    for name, value in list(glbs.items()):
        if isinstance(value, type) and not issubclass(value, Lazy):
            glbs[name] = LazyClass(value)


def concretize(node):
    """Evaluate a lazy description (recursing through containers); anything else is returned as is."""
    if isinstance(node, LazyCall):
        fn = getattr(concretize(node.target), node.method)
        return fn(*concretize(node.args), **concretize(node.kwargs))
    if isinstance(node, LazyAttr):
        return concretize(getattr(concretize(node.target), node.name))
    if isinstance(node, LazyObject):
        memo = node.__dict__
        if "_built" not in memo:
            memo["_built"] = node.cls(**concretize(node.kwargs))
        return memo["_built"]
    if isinstance(node, LazyClass):
        return node.cls
    if isinstance(node, dict):
        return {concretize(k): concretize(v) for k, v in node.items()}
    if isinstance(node, (list, tuple)):
        return type(node)(concretize(x) for x in node)
    return node


class VariantDict(dict):
    """A variant whose entries also read as attributes (``v.lr``)."""

    def __getattr__(self, k):
        if k in self:
            return self[k]
        raise AttributeError(k)


exp_count = 0
now = datetime.datetime.now()
timestamp = now.strftime('%Y_%m_%d_%H_%M_%S')


def run_experiment_lite(stub_method_call=None, batch_tasks=None, exp_prefix="experiment", exp_name=None,
                        log_dir=None, script="scripts/run_experiment_lite.py", python_command="python",
                        mode="local", dry=False, variant=None, use_cloudpickle=None, n_parallel=1,
                        snapshot_mode="all", snapshot_gap=1, seed=None, plot=False, resume_from=None,
                        tabular_log_file="progress.csv", text_log_file="debug.log",
                        params_log_file="params.json", variant_log_file="variant.json",
                        log_tabular_only=False, **kwargs):
    """Run a task (callable or stubbed method call) with the reference's experiment bookkeeping.
    Returns the experiment's log directory."""
    global exp_count
    assert stub_method_call is not None or batch_tasks is not None or resume_from is not None, \
        "Must provide at least either stub_method_call or batch_tasks"
    if mode != "local":
        raise NotImplementedError("run_experiment_lite: only mode='local' is built (docker / ec2 / lab_kube "
                                  "launching is outside the hot path, SURVEY.md section 2)")
    if plot:
        raise NotImplementedError("plotting is out of scope")
    if batch_tasks is None:
        batch_tasks = [dict(stub_method_call=stub_method_call, exp_name=exp_name, log_dir=log_dir, variant=variant)]
    last_dir = None
    for task in batch_tasks:
        call = task.get("stub_method_call", stub_method_call)
        task_variant = task.get("variant", variant)
        exp_count += 1
        name = task.get("exp_name") or "%s_%s_%04d" % (exp_prefix, timestamp, exp_count)
        ldir = task.get("log_dir") or osp.join(config.LOG_DIR, "local", exp_prefix.replace("_", "-"), name)
        if dry:
            print("run_experiment_lite (dry): %s -> %s" % (name, ldir))
            continue
        last_dir = _run_local(call, name, ldir, task_variant, n_parallel, snapshot_mode, snapshot_gap, seed,
                              resume_from, tabular_log_file, text_log_file, params_log_file, variant_log_file,
                              log_tabular_only)
    return last_dir


def _run_local(call, exp_name, log_dir, variant, n_parallel, snapshot_mode, snapshot_gap, seed, resume_from,
               tabular_log_file, text_log_file, params_log_file, variant_log_file, log_tabular_only):
    if seed is not None:
        ext.set_seed(seed)
    if n_parallel and n_parallel > 0:
        # the CPU worker pool of the generic-env sampler (scripts/run_experiment_lite.py:73-77 of the reference);
        # HIP-native envs never use it -- their rollouts are one kernel launch -- so one worker means "inline"
        from rllab_amd.sampler import parallel_sampler
        parallel_sampler.initialize(n_parallel=n_parallel)
        if seed is not None:
            parallel_sampler.set_seed(seed)
    os.makedirs(log_dir, exist_ok=True)
    tabular_log_file = osp.join(log_dir, tabular_log_file)
    text_log_file = osp.join(log_dir, text_log_file)
    if variant is not None:
        with open(osp.join(log_dir, variant_log_file), "w") as fh:
            json.dump(dict(variant, exp_name=exp_name), fh, indent=2, sort_keys=True, default=str)
    with open(osp.join(log_dir, params_log_file), "w") as fh:
        json.dump(dict(exp_name=exp_name, n_parallel=n_parallel, snapshot_mode=snapshot_mode,
                       snapshot_gap=snapshot_gap, seed=seed, resume_from=resume_from,
                       rollout_workers="CPU pool of %d for generic Python envs; HIP-native envs use the lock-step GPU sampler"
                                       % max(1, n_parallel or 1)),
                  fh, indent=2, sort_keys=True)
    logger.add_text_output(text_log_file)
    logger.add_tabular_output(tabular_log_file)
    prev_snapshot_dir, prev_mode = logger.get_snapshot_dir(), logger.get_snapshot_mode()
    prev_gap, prev_only = logger.get_snapshot_gap(), logger.get_log_tabular_only()
    logger.set_snapshot_dir(log_dir)
    logger.set_snapshot_mode(snapshot_mode)
    logger.set_snapshot_gap(snapshot_gap)
    logger.set_log_tabular_only(log_tabular_only)
    logger.push_prefix("[%s] " % exp_name)
    try:
        if resume_from is not None:
            import joblib
            data = joblib.load(resume_from)
            assert 'algo' in data
            data['algo'].train()
        elif isinstance(call, Lazy):
            maybe_iter = concretize(call)
            if ext.is_iterable(maybe_iter):
                for _ in maybe_iter:
                    pass
        else:
            assert hasattr(call, '__call__')
            call(VariantDict(variant or {}))
    finally:
        logger.set_snapshot_mode(prev_mode)
        logger.set_snapshot_dir(prev_snapshot_dir)
        logger.set_snapshot_gap(prev_gap)
        logger.set_log_tabular_only(prev_only)
        logger.remove_tabular_output(tabular_log_file)
        logger.remove_text_output(text_log_file)
        logger.pop_prefix()
    return log_dir
