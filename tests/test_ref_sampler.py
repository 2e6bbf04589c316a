"""The reference's UNMODIFIED CPU sampler (oracle/ref_sampler.py over oracle/_ref, staged by
oracle/make_ref.py) -- the ``cpu_baseline`` of bench.py -- and its cross-checks.

CPU tests: the staged files are byte-identical to /root/reference (where mounted), the child
process really imports ``rllab.sampler.*`` from the staged tree, and the re-typed port
(oracle/cpu_sampler.py, also the sampler-level oracle of other tests) reproduces the reference
sampler's paths under the same seed -- i.e. rollout(), the NormalizedEnv action map and the
draw order of the port ARE the reference's (rllab/sampler/utils.py:6-43,
rllab/envs/normalized_env.py:78-92).
"""
import hashlib
import json
import os
import pickle

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGED = os.path.join(ROOT, "oracle", "_ref")

needs_ref = pytest.mark.skipif(not (os.path.isdir(os.path.join(STAGED, "rllab")) or os.path.isdir("/root/reference")),
                               reason="no staged reference (run oracle/make_ref.py in the build container)")


def _theta(kind, hidden=(32, 32), seed=3):
    from oracle import host_env as H
    from oracle import np_reference as R
    q = H.query(kind)
    net = R.NumpyGaussianMLP(q["obs_dim"], q["act_dim"], hidden)
    return np.random.RandomState(seed).randn(net.n_params) * 0.3


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference tree not mounted")
def test_staged_files_are_the_reference_byte_for_byte():
    man = json.load(open(os.path.join(STAGED, "MANIFEST.json")))["files"]
    assert "rllab/sampler/stateful_pool.py" in man and "rllab/sampler/parallel_sampler.py" in man
    assert "rllab/sampler/utils.py" in man and "rllab/envs/normalized_env.py" in man
    assert "examples/trpo_cartpole.py" in man and "examples/trpo_swimmer.py" in man
    for rel, digest in man.items():
        assert hashlib.sha256(open(os.path.join("/root/reference", rel), "rb").read()).hexdigest() == digest, rel
        assert hashlib.sha256(open(os.path.join(STAGED, rel), "rb").read()).hexdigest() == digest, rel


@needs_ref
@pytest.mark.parametrize("kind,T", [(0, 100), (2, 40)])
def test_port_reproduces_the_reference_sampler_paths(tmp_path, kind, T):
    from oracle import cpu_sampler, ref_sampler
    theta = _theta(kind)
    dump = str(tmp_path / "paths.pkl")
    rec = ref_sampler.run(kind, theta, T, [T * 3], n_parallel=1, seed=7, dump_paths=dump)
    assert rec["modules"]["rllab.sampler.stateful_pool"] == "rllab/sampler/stateful_pool.py"
    ref_paths = pickle.load(open(dump, "rb"))
    port_paths, _, _ = cpu_sampler.sample_paths(kind, theta, T * 3, T, n_parallel=1, seed=7)
    assert len(ref_paths) == len(port_paths) >= 3
    assert sum(len(p["rewards"]) for p in ref_paths) == rec["runs"][-1]["steps"]
    for a, b in zip(ref_paths, port_paths):
        assert a["observations"].shape == b["observations"].shape
        assert set(a) == {"observations", "actions", "rewards", "agent_infos", "env_infos"}
        for key in ("observations", "actions", "rewards"):
            np.testing.assert_allclose(a[key], b[key], rtol=0, atol=1e-11)
        np.testing.assert_allclose(a["agent_infos"]["mean"], b["agent_infos"]["mean"], rtol=0, atol=1e-11)


@needs_ref
def test_reference_pool_with_worker_processes_collects_whole_paths():
    """n_parallel = 2: StatefulPool.run_collect's threshold semantics (>= max_samples, whole paths only)
    through the reference's own worker processes."""
    from oracle import ref_sampler
    T = 25
    rec = ref_sampler.run(2, _theta(2), T, [T * 4, T * 10], n_parallel=2, seed=1)
    for run, want in zip(rec["runs"], (T * 4, T * 10)):
        assert run["steps"] >= want and run["steps"] % T == 0      # Swimmer never terminates early
        assert run["steps"] <= want + 2 * T                         # at most one extra path per worker
    assert rec["n_parallel"] == 2
