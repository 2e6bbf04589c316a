"""Experiment launching, local mode (API of rllab/misc/instrument.py:30-163, 290-297, 338-470,
1340-1395 + scripts/run_experiment_lite.py:21-139).

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
from rllab_amd.core.serializable import Serializable
from rllab_amd.misc import ext, logger


class StubBase(object):
    def __getitem__(self, item):
        return StubMethodCall(self, "__getitem__", args=[item], kwargs=dict())

    def __getattr__(self, item):
        try:
            return super(self.__class__, self).__getattribute__(item)
        except AttributeError:
            if item.startswith("__") and item.endswith("__"):
                raise
            return StubAttr(self, item)

    def __call__(self, *args, **kwargs):
        return StubMethodCall(self.obj, self.attr_name, args, kwargs)

    def __add__(self, other):
        return StubMethodCall(self, "__add__", [other], dict())

    def __rmul__(self, other):
        return StubMethodCall(self, "__rmul__", [other], dict())

    def __pow__(self, power, modulo=None):
        return StubMethodCall(self, "__pow__", [power, modulo], dict())


class StubAttr(StubBase):
    def __init__(self, obj, attr_name):
        self.__dict__["_obj"] = obj
        self.__dict__["_attr_name"] = attr_name

    @property
    def obj(self):
        return self.__dict__["_obj"]

    @property
    def attr_name(self):
        return self.__dict__["_attr_name"]

    def __str__(self):
        return "StubAttr(%s, %s)" % (str(self.obj), str(self.attr_name))


class StubMethodCall(StubBase, Serializable):
    def __init__(self, obj, method_name, args, kwargs):
        self._serializable_initialized = False
        Serializable.quick_init(self, locals())
        self.obj = obj
        self.method_name = method_name
        self.args = args
        self.kwargs = kwargs

    def __str__(self):
        return "StubMethodCall(%s, %s, %s, %s)" % (str(self.obj), str(self.method_name), str(self.args),
                                                   str(self.kwargs))


def _positional_to_kwargs(cls, args, kwargs):
    if len(args) > 0:
        spec = inspect.getfullargspec(cls.__init__)
        kwargs = dict(list(zip(spec.args[1:], args)), **kwargs)
    return kwargs


class StubClass(StubBase):
    def __init__(self, proxy_class):
        self.proxy_class = proxy_class

    def __call__(self, *args, **kwargs):
        return StubObject(self.proxy_class, **_positional_to_kwargs(self.proxy_class, args, kwargs))

    def __getstate__(self):
        return dict(proxy_class=self.proxy_class)

    def __setstate__(self, d):
        self.proxy_class = d["proxy_class"]

    def __getattr__(self, item):
        if hasattr(self.proxy_class, item):
            return StubAttr(self, item)
        raise AttributeError(item)

    def __str__(self):
        return "StubClass(%s)" % self.proxy_class


class StubObject(StubBase):
    def __init__(self, __proxy_class, *args, **kwargs):
        self.proxy_class = __proxy_class
        self.args = tuple()
        self.kwargs = _positional_to_kwargs(__proxy_class, args, kwargs)

    def __getstate__(self):
        return dict(args=self.args, kwargs=self.kwargs, proxy_class=self.proxy_class)

    def __setstate__(self, d):
        self.args, self.kwargs, self.proxy_class = d["args"], d["kwargs"], d["proxy_class"]

    def __getattr__(self, item):
        if hasattr(self.proxy_class, item):
            return StubAttr(self, item)
        raise AttributeError('Cannot get attribute %s from %s' % (item, self.proxy_class))

    def __str__(self):
        return "StubObject(%s, *%s, **%s)" % (str(self.proxy_class), str(self.args), str(self.kwargs))


def stub(glbs):
    """Replace every class in ``glbs`` by a StubClass: constructor calls then build a lazy
    description of the experiment instead of the objects themselves."""
    for k, v in list(glbs.items()):
        if isinstance(v, type) and v != StubClass:
            glbs[k] = StubClass(v)


def concretize(maybe_stub):
    if isinstance(maybe_stub, StubMethodCall):
        obj = concretize(maybe_stub.obj)
        method = getattr(obj, maybe_stub.method_name)
        return method(*concretize(maybe_stub.args), **concretize(maybe_stub.kwargs))
    elif isinstance(maybe_stub, StubClass):
        return maybe_stub.proxy_class
    elif isinstance(maybe_stub, StubAttr):
        return concretize(getattr(concretize(maybe_stub.obj), maybe_stub.attr_name))
    elif isinstance(maybe_stub, StubObject):
        if "_stub_cache" not in maybe_stub.__dict__:
            maybe_stub.__dict__["_stub_cache"] = maybe_stub.proxy_class(*concretize(maybe_stub.args),
                                                                       **concretize(maybe_stub.kwargs))
        return maybe_stub.__dict__["_stub_cache"]
    elif isinstance(maybe_stub, dict):
        return {concretize(k): concretize(v) for k, v in maybe_stub.items()}
    elif isinstance(maybe_stub, (list, tuple)):
        return maybe_stub.__class__(list(map(concretize, maybe_stub)))
    return maybe_stub


class VariantDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
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
        elif isinstance(call, StubBase):
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
