"""Host-side pieces of the sampler's batch-size contract (CPU): the replay harness around the reference's OWN
``VectorizedSampler.obtain_samples`` / ``truncate_paths`` (oracle/ref_vecsampler.py) on a hand-checkable recording, and
``VectorizedSampler._keep_first`` (``whole_paths=False``) against what ``truncate_paths`` leaves of the same paths."""
import numpy as np
import pytest
import torch

from test_ref_vecenv import needs_ref


def recording():
    # env 0: paths end at t = 1 and t = 4; env 1: at t = 2 and t = 5
    d = np.zeros((6, 2), np.uint8)
    d[1, 0] = d[4, 0] = d[2, 1] = d[5, 1] = 1
    return d, np.arange(12.0).reshape(6, 2)


@needs_ref
@pytest.mark.parametrize("want,whole,steps,paths", [
    (2, True, 2, [(0, 0, 2)]),
    (3, True, 3, [(0, 0, 2), (1, 0, 3)]),
    (3, False, 3, [(0, 0, 2), (1, 0, 1)]),
    (6, True, 5, [(0, 0, 2), (1, 0, 3), (0, 2, 3)]),
    (6, False, 5, [(0, 0, 2), (1, 0, 3), (0, 2, 1)]),
    (11, True, 6, [(0, 0, 2), (1, 0, 3), (0, 2, 3), (1, 3, 3)]),
])
def test_reference_loop_on_a_replayed_recording(want, whole, steps, paths):
    from oracle import ref_vecsampler
    d, r = recording()
    out = ref_vecsampler.run(d, r, want, 100, whole)
    assert int(out["steps"]) == steps
    assert list(zip(out["env"].tolist(), out["t0"].tolist(), out["length"].tolist())) == paths


@needs_ref
def test_reference_loop_refuses_a_recording_that_is_too_short():
    from oracle import ref_vecsampler
    d, r = recording()
    with pytest.raises(RuntimeError, match="asks for lock step 6"):
        ref_vecsampler.run(d, r, 12, 100, True)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_keep_first_is_truncate_paths_in_env_order(seed):
    from rllab_amd.sampler.trajectories import PathList, Trajectories
    from rllab_amd.sampler.utils import truncate_paths
    from rllab_amd.sampler.vectorized_sampler import VectorizedSampler
    rng = np.random.RandomState(seed)
    T, N = 23, 7
    dones = torch.as_tensor((rng.rand(T, N) < 0.15).astype(np.uint8))
    rewards = torch.as_tensor(rng.randn(T, N).astype(np.float32))
    z3 = lambda d: torch.zeros((d, T, N))
    tr = Trajectories(z3(2), z3(1), z3(1), torch.zeros(1), rewards, dones.clone(), T)
    whole = tr.valid_mask(True)
    total = int(whole.sum())
    # the finished paths as a list in env order, through the product's copy of truncate_paths (pinned to the reference's
    # by tests/test_reference_tests_verbatim.py)
    tr.valid = whole
    full = [dict(rewards=p["rewards"]) for p in PathList(tr)]
    for want in (1, total // 3, total - 1, total):
        tr2 = Trajectories(z3(2), z3(1), z3(1), torch.zeros(1), rewards, dones.clone(), T)
        VectorizedSampler._keep_first(tr2, want, whole)
        assert int(tr2.valid.sum()) == want
        got = [p["rewards"] for p in PathList(tr2)]
        ref = [p["rewards"] for p in truncate_paths(full, want)]
        assert len(got) == len(ref) and all(np.array_equal(a, b) for a, b in zip(got, ref))
    # every sample valid (an env kind that never terminates): rank is the plain env-major index
    tr3 = Trajectories(z3(2), z3(1), z3(1), torch.zeros(1), rewards, torch.zeros_like(dones), T)
    VectorizedSampler._keep_first(tr3, 2 * T + 5, None)
    assert int(tr3.valid.sum()) == 2 * T + 5 and bool(tr3.valid[:, :2].all()) and int(tr3.valid[:, 2].sum()) == 5
    assert int(tr3.dones[4, 2]) == 1
