"""Round-3 host logic (CPU only)."""
import types

import numpy as np
import pytest


class _FakeVecEnv(object):
    position_ids = None
    graphable = True


def _sampler(policy):
    from rllab_amd.sampler.vectorized_sampler import VectorizedSampler
    algo = types.SimpleNamespace(policy=policy, max_path_length=5, batch_size=10, env=None)
    s = VectorizedSampler(algo, n_envs=2)
    s.vec_env = _FakeVecEnv()
    return s


def test_prefetch_is_a_noop_without_a_parameter_version_or_a_fused_rollout():
    """A prefetched batch can only be handed out when policy.param_version() says the parameters did not move, and
    only the fused rollout is one asynchronous launch: for every other policy prefetch() must not sample at all
    (it used to roll out, discard and roll out again)."""
    calls = []

    class NoVersion(object):                       # a vectorised policy without param_version()
        def kernel_layout(self):
            return object()

    class Stepwise(object):                        # has a version, but samples through the per-transition loop
        def kernel_layout(self):
            return None

        def param_version(self):
            return 1

    class Fused(object):
        def __init__(self):
            self.v = 7

        def kernel_layout(self):
            return object()

        def param_version(self):
            return self.v

    for pol in (NoVersion(), Stepwise()):
        s = _sampler(pol)
        s.obtain_samples = lambda itr: calls.append(itr)
        s.prefetch(3)
        assert calls == [] and getattr(s, "_prefetched", None) is None

    pol = Fused()
    s = _sampler(pol)
    real = s.obtain_samples
    s.last_sample_time = 0.125

    def fake(itr):
        calls.append(itr)
        s.last_sample_time = 99.0
        return "batch%d" % itr
    s.obtain_samples = fake
    s.prefetch(4)
    assert calls == [4] and s._prefetched == (4, 7, "batch4")
    assert s.last_sample_time == 0.125             # the enqueue of the next batch is not this iteration's time
    s.prefetch(4)                                  # idempotent
    assert calls == [4]
    s.obtain_samples = real
    assert s.obtain_samples(4) == "batch4"         # same version: handed out
    assert s._prefetched is None
    s.obtain_samples = fake
    s.prefetch(5)
    pol.v = 8                                      # parameters moved: the prefetched batch must not be used
    s.obtain_samples = real
    s._takes_fused_rollout = lambda p: False
    s._stepwise_rollout = lambda p, T: types.SimpleNamespace(B=10)
    s.use_graph = False
    out = s.obtain_samples(5)
    assert out != "batch5"


def test_logger_decides_who_writes_at_write_time(tmp_path, monkeypatch):
    """run_experiment_lite opens the sinks BEFORE a user script calls init_process_group: a torchrun rank > 0 must
    neither create nor truncate progress.csv -- primary-ness comes from the launcher's RANK until the process group
    exists, files are opened by the first primary write, and it is re-checked on every write."""
    from rllab_amd.misc import logger
    csv_path, txt_path = str(tmp_path / "progress.csv"), str(tmp_path / "debug.log")
    with open(csv_path, "w") as fh:
        fh.write("AverageReturn\n9.0\n")               # rank 0 already wrote its header and a row
    logger.set_quiet(True)
    try:
        logger.set_primary(None)                         # decide lazily
        monkeypatch.setenv("RANK", "1")
        assert not logger.is_primary()
        logger.add_tabular_output(csv_path)              # a late-starting secondary rank registers the same sinks
        logger.add_text_output(txt_path)
        logger.log("x")
        logger.record_tabular("AverageReturn", 1.0)
        logger.dump_tabular()
        assert open(csv_path).read() == "AverageReturn\n9.0\n" and not (tmp_path / "debug.log").exists()
        monkeypatch.setenv("RANK", "0")
        assert logger.is_primary()
        logger.record_tabular("AverageReturn", 2.0)
        logger.dump_tabular()
        logger.log("y")
        assert open(csv_path).read().split() == ["AverageReturn", "2.0"]
        assert "y" in open(txt_path).read()
    finally:
        logger.remove_tabular_output(csv_path)
        logger.remove_text_output(txt_path)
        logger.set_primary(None)
        logger.set_quiet(False)
