"""CPU worker-pool sampler for arbitrary Python envs -- the API of rllab/sampler/parallel_sampler.py:18-155 on
``stateful_pool.singleton_pool``.  (HIP-native envs do not use it: their sampler is the lock-step
``VectorizedSampler``; this module is what ``BatchSampler`` falls back on for envs that only exist as Python
objects, and what scripts written against the reference call: ``parallel_sampler.initialize(n_parallel)``,
``parallel_sampler.set_seed(seed)``.)

Each worker keeps its own unpickled copy of (env, policy) per ``scope``; ``sample_paths`` ships the flat policy
parameters, then lets every worker roll out whole paths until the shared sample counter reaches ``max_samples``;
worker i is seeded with ``seed + i``.
"""
import pickle

import numpy as np

from rllab_amd.misc import ext, logger
from rllab_amd.sampler.stateful_pool import SharedGlobal, singleton_pool
from rllab_amd.sampler.utils import rollout, truncate_paths  # noqa: F401  (truncate_paths is part of this module's API)


def _scoped(G, scope):
    """The worker's state for one algorithm instance (``scope`` lets several algos share the pool)."""
    if scope is None:
        return G
    scopes = G.__dict__.setdefault("scopes", {})
    if scope not in scopes:
        scopes[scope] = SharedGlobal()
        scopes[scope].worker_id = getattr(G, "worker_id", 0)
    return scopes[scope]


_get_scoped_G = _scoped   # the reference's name


def _worker_init(G, worker_id):
    G.worker_id = worker_id


def initialize(n_parallel):
    singleton_pool.initialize(n_parallel)
    singleton_pool.run_each(_worker_init, [(i,) for i in range(singleton_pool.n_parallel)])


def _worker_populate_task(G, env, policy, scope=None):
    g = _scoped(G, scope)
    g.env, g.policy = pickle.loads(env), pickle.loads(policy)


def _worker_terminate_task(G, scope=None):
    g = _scoped(G, scope)
    for name in ("env", "policy"):
        obj = getattr(g, name, None)
        if obj is not None:
            obj.terminate()
            setattr(g, name, None)


def populate_task(env, policy, scope=None):
    logger.log("Populating workers...")
    if singleton_pool.n_parallel > 1:
        blob = (pickle.dumps(env), pickle.dumps(policy), scope)
        singleton_pool.run_each(_worker_populate_task, [blob] * singleton_pool.n_parallel)
    else:   # in-process: share the caller's objects instead of copying them
        g = _scoped(singleton_pool.G, scope)
        g.env, g.policy = env, policy
    logger.log("Populated")


def terminate_task(scope=None):
    singleton_pool.run_each(_worker_terminate_task, [(scope,)] * singleton_pool.n_parallel)


def _worker_set_seed(_, seed):
    logger.log("Setting seed to %d" % seed)
    ext.set_seed(seed)


def set_seed(seed):
    singleton_pool.run_each(_worker_set_seed, [(seed + i,) for i in range(singleton_pool.n_parallel)])


def _worker_set_policy_params(G, params, scope=None):
    _scoped(G, scope).policy.set_param_values(params)


def _worker_set_env_params(G, params, scope=None):
    _scoped(G, scope).env.set_param_values(params)


def _worker_collect_one_path(G, max_path_length, scope=None):
    g = _scoped(G, scope)
    path = rollout(g.env, g.policy, max_path_length)
    return path, len(path["rewards"])


def sample_paths(policy_params, max_samples, max_path_length=np.inf, env_params=None, scope=None):
    """At least ``max_samples`` env steps of whole paths (each runs to termination or ``max_path_length``) under
    the given flat policy parameters; returns the list of path dicts."""
    n = singleton_pool.n_parallel
    singleton_pool.run_each(_worker_set_policy_params, [(policy_params, scope)] * n)
    if env_params is not None:
        singleton_pool.run_each(_worker_set_env_params, [(env_params, scope)] * n)
    return singleton_pool.run_collect(_worker_collect_one_path, threshold=max_samples,
                                      args=(max_path_length, scope), show_prog_bar=True)
