"""CPU port of the reference sampler -- the ``cpu_baseline`` of bench.py and the
sampler-level oracle of tests (TEST INFRASTRUCTURE, never imported by the product).

Restates, in the reference's own shape (pure-Python per-step loop, one process per
core, policy parameters re-sent every iteration, a shared counter the master polls
every 0.1 s):
  rollout()                         rllab/sampler/utils.py:6-43
  _worker_collect_one_path          rllab/sampler/parallel_sampler.py:92-95
  StatefulPool.run_collect + _worker_run_collect
                                    rllab/sampler/stateful_pool.py:101-188
  _worker_set_seed (seed + worker)  rllab/sampler/parallel_sampler.py:72-81
The env is the host build of the engine's dynamics (oracle/host_env.py) behind the
NormalizedEnv action map (rllab/envs/normalized_env.py:78-92) -- one ctypes call per
step, as the reference pays one SWIG/ctypes call per step -- and the policy is the
batch-1 NumPy MLP of oracle/np_reference.py.  Caveat printed with every number: the
true reference stack (Theano + pybox2d / MuJoCo 1.31) cannot be installed here, so
this is an UPPER bound on reference throughput.
"""
import multiprocessing as mp
import os
import time

import numpy as np

from oracle import host_env as H
from oracle import np_reference as R


class HostNormalizedEnv(object):
    """NormalizedEnv(<env>) with reference reset distributions drawn from np.random."""

    def __init__(self, kind):
        self.kind = kind
        self.env = H.HostEnv(kind, np.float64, normalize=True)
        self.q = self.env.q

    def reset(self):
        n = self.q["reset_draws"]
        draws = np.random.normal(size=n) if self.q["reset_is_normal"] else np.random.uniform(size=n)
        return self.env.reset(draws)

    def step(self, action):
        return self.env.step(action)


def rollout(env, agent, max_path_length):
    observations, actions, rewards, means, log_stds = [], [], [], [], []
    o = env.reset()
    agent.reset()
    path_length = 0
    while path_length < max_path_length:
        a, agent_info = agent.get_action(o)
        next_o, r, d = env.step(a)
        observations.append(np.asarray(o).flatten())
        rewards.append(r)
        actions.append(np.asarray(a).flatten())
        means.append(agent_info["mean"])
        log_stds.append(agent_info["log_std"])
        path_length += 1
        if d:
            break
        o = next_o
    return dict(observations=np.array(observations), actions=np.array(actions), rewards=np.array(rewards),
                agent_infos=dict(mean=np.array(means), log_std=np.array(log_stds)), env_infos=dict())


_G = {}


def _worker_init(kind, hidden, seed_base, counter_id):
    ident = mp.current_process()._identity
    wid = ident[0] - 1 if ident else 0
    np.random.seed(seed_base + wid)
    _G["env"] = HostNormalizedEnv(kind)
    q = _G["env"].q
    _G["policy"] = R.NumpyGaussianMLP(q["obs_dim"], q["act_dim"], hidden)


def _worker_collect(args):
    theta, max_path_length, threshold, counter, lock = args
    _G["policy"].set_param_values(theta)
    collected = []
    while True:
        with lock:
            if counter.value >= threshold:
                return collected
        path = rollout(_G["env"], _G["policy"], max_path_length)
        with lock:
            counter.value += len(path["rewards"])
        collected.append(path)


def sample_paths(kind, theta, max_samples, max_path_length, n_parallel=None, hidden=(32, 32), seed=1):
    """Collect >= max_samples env steps of whole paths with n_parallel worker processes.
    Returns (paths, seconds, n_parallel)."""
    n_parallel = n_parallel or os.cpu_count()
    if n_parallel == 1:
        np.random.seed(seed)
        env = HostNormalizedEnv(kind)
        pol = R.NumpyGaussianMLP(env.q["obs_dim"], env.q["act_dim"], hidden)
        pol.set_param_values(theta)
        t0 = time.time()
        paths, n = [], 0
        while n < max_samples:
            p = rollout(env, pol, max_path_length)
            paths.append(p)
            n += len(p["rewards"])
        return paths, time.time() - t0, 1
    ctx = mp.get_context("fork")
    manager = ctx.Manager()
    counter, lock = manager.Value('i', 0), manager.RLock()
    pool = ctx.Pool(n_parallel, initializer=_worker_init, initargs=(kind, hidden, seed, 0))
    try:
        t0 = time.time()
        res = pool.map_async(_worker_collect, [(theta, max_path_length, max_samples, counter, lock)] * n_parallel)
        while not res.ready():
            res.wait(0.1)  # the reference master polls its counter every 0.1 s
        paths = sum(res.get(), [])
        dt = time.time() - t0
    finally:
        pool.terminate()
        manager.shutdown()
    return paths, dt, n_parallel


def timed_baseline(kind, theta, max_path_length, budget_s=15.0, hidden=(32, 32), n_parallel=None):
    """Steps/s of the port on this host, on a sample sized to ~budget_s seconds of wall
    time (calibrated with a short parallel run: the pool does not scale linearly -- the
    shared counter lives in a manager process, as in the reference)."""
    n_parallel = n_parallel or os.cpu_count()
    paths, dt1, _ = sample_paths(kind, theta, max_path_length * 2, max_path_length, n_parallel=1, hidden=hidden)
    rate1 = sum(len(p["rewards"]) for p in paths) / dt1
    cal_target = max_path_length * n_parallel * 2
    paths, dt, n_par = sample_paths(kind, theta, cal_target, max_path_length, n_parallel=n_parallel, hidden=hidden)
    n = sum(len(p["rewards"]) for p in paths)
    rate = n / dt
    remaining = budget_s - dt
    if remaining > 2.0:
        target = int(rate * remaining * 0.8)
        paths, dt, n_par = sample_paths(kind, theta, target, max_path_length, n_parallel=n_parallel, hidden=hidden)
        n = sum(len(p["rewards"]) for p in paths)
    return dict(steps=n, seconds=dt, steps_per_s=n / dt, cores=n_par, steps_per_s_1core=rate1, paths=paths)
