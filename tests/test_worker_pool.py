"""CPU worker pool of the generic-env sampler: sampler/stateful_pool.py and sampler/parallel_sampler.py against
the behaviour the reference pins in tests/test_stateful_pool.py and tests/test_sampler.py, plus the pool
conventions of SURVEY.md 8b (run_each exactly once per worker, worker i seeded seed + i, worker exceptions
re-raised with their traceback, workers kept off the GPU)."""
import os

import numpy as np
import pytest


def _collect_once(_):
    return 'a', 1


def _whoami(G):
    return getattr(G, "worker_id", None), os.getpid(), os.environ.get("CUDA_VISIBLE_DEVICES")


def _boom(G, x):
    raise ValueError("boom %d" % x)


def _square(G, x):
    return x * x


def _draw(G):
    return float(np.random.rand())


@pytest.fixture
def pool():
    from rllab_amd.sampler import stateful_pool
    yield stateful_pool.singleton_pool
    stateful_pool.singleton_pool.terminate()


def test_stateful_pool_collects_to_threshold(pool):
    # reference tests/test_stateful_pool.py:8-12
    pool.initialize(n_parallel=3)
    assert tuple(pool.run_collect(_collect_once, 3, show_prog_bar=False)) == ('a', 'a', 'a')


def test_stateful_pool_over_capacity(pool):
    # reference tests/test_stateful_pool.py:15-19
    pool.initialize(n_parallel=4)
    assert len(pool.run_collect(_collect_once, 3, show_prog_bar=False)) >= 3


def test_inline_pool_and_rllab_alias(pool):
    from rllab.sampler import stateful_pool as aliased      # drop-in import path
    assert aliased.singleton_pool is pool
    pool.initialize(n_parallel=1)
    assert pool.run_collect(_collect_once, 5) == ['a'] * 5
    assert pool.run_each(_square, [(7,)]) == [49]
    assert pool.run_map(_square, [(i,) for i in range(5)]) == [0, 1, 4, 9, 16]


def test_run_each_is_once_per_worker_and_workers_stay_off_the_gpu(pool):
    from rllab_amd.sampler import parallel_sampler
    parallel_sampler.initialize(n_parallel=3)
    seen = pool.run_each(_whoami)
    assert sorted(w for w, _, _ in seen) == [0, 1, 2]                 # every worker exactly once
    assert len({pid for _, pid, _ in seen}) == 3 and os.getpid() not in {pid for _, pid, _ in seen}
    assert all(vis == "" for _, _, vis in seen)
    assert pool.run_map(_square, [(i,) for i in range(10)]) == [i * i for i in range(10)]
    assert sorted(pool.run_imap_unordered(_square, [(i,) for i in range(4)])) == [0, 1, 4, 9]
    # worker i is seeded with seed + i (parallel_sampler.py:72-81): distinct streams, reproducible
    parallel_sampler.set_seed(11)
    a = pool.run_each(_draw)
    parallel_sampler.set_seed(11)
    assert pool.run_each(_draw) == a and len(set(a)) == 3
    np.random.seed(12)
    assert a[1] == float(np.random.rand())


def test_worker_exception_reaches_the_caller_with_its_traceback(pool):
    pool.initialize(n_parallel=2)
    with pytest.raises(Exception) as err:
        pool.run_each(_boom, [(1,), (2,)])
    assert "ValueError: boom" in str(err.value) and "_boom" in str(err.value)
    assert pool.run_each(_square, [(2,), (3,)]) == [4, 9]              # the pool survives


class _LineEnv(object):
    """1-D point that walks by its action; done after a fixed number of steps."""
    def __init__(self, horizon=7):
        from rllab_amd.spaces import Box
        self.horizon, self.t, self.x = horizon, 0, 0.0
        self.observation_space = Box(-np.ones(1) * 1e6, np.ones(1) * 1e6)
        self.action_space = Box(-np.ones(1), np.ones(1))

    def reset(self):
        self.t, self.x = 0, 0.0
        return np.array([self.x])

    def step(self, a):
        self.t += 1
        self.x += float(np.asarray(a).reshape(-1)[0])
        return np.array([self.x]), -abs(self.x), self.t >= self.horizon, dict(t=self.t)

    def terminate(self):
        pass


class _GainPolicy(object):
    def __init__(self):
        self.k = np.zeros(1)

    def reset(self):
        pass

    def get_action(self, obs):
        mean = self.k * obs
        return mean + 0.1 * np.random.randn(1), dict(mean=mean)

    def get_param_values(self, **tags):
        return self.k.copy()

    def set_param_values(self, v, **tags):
        self.k = np.asarray(v, dtype=np.float64).copy()

    def terminate(self):
        pass


@pytest.mark.parametrize("n_parallel", [1, 3])
def test_parallel_sampler_sample_paths(pool, n_parallel):
    from rllab_amd.sampler import parallel_sampler
    parallel_sampler.initialize(n_parallel=n_parallel)
    parallel_sampler.set_seed(5)
    parallel_sampler.populate_task(_LineEnv(), _GainPolicy(), scope="s")
    paths = parallel_sampler.sample_paths(np.array([-0.5]), max_samples=60, max_path_length=5, scope="s")
    n = sum(len(p["rewards"]) for p in paths)
    assert n >= 60 and all(len(p["rewards"]) == 5 for p in paths)      # whole paths, cut at max_path_length
    p = paths[0]
    assert p["observations"].shape == (5, 1) and p["actions"].shape == (5, 1)
    assert np.array_equal(p["env_infos"]["t"], np.arange(1, 6))
    assert np.allclose(p["agent_infos"]["mean"], -0.5 * p["observations"])   # the shipped parameters were used
    cut = parallel_sampler.truncate_paths(paths, 58)
    assert sum(len(q["rewards"]) for q in cut) == 58
    parallel_sampler.terminate_task(scope="s")


def test_batch_sampler_uses_the_pool(pool, quiet_logger):
    """BatchSampler (generic Python env) through three workers: whole paths >= batch_size, exact batch when
    whole_paths is off."""
    from rllab_amd.algos.batch_polopt import BatchSampler
    from rllab_amd.sampler import parallel_sampler

    class _Algo(object):
        env, policy, scope = _LineEnv(), _GainPolicy(), None
        batch_size, max_path_length, whole_paths = 40, 7, True
    parallel_sampler.initialize(n_parallel=3)
    s = BatchSampler(_Algo())
    s.start_worker()
    paths = s.obtain_samples(0)
    assert sum(len(p["rewards"]) for p in paths) >= 40 and all(len(p["rewards"]) == 7 for p in paths)
    _Algo.whole_paths = False
    assert sum(len(p["rewards"]) for p in s.obtain_samples(1)) == 40
    s.shutdown_worker()
