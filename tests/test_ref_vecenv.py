"""The oracle's lock-step executor (oracle/host_env.py::HostVecEnv, the C restatement ``oracle_vecenv_step`` of
sandbox/rocky/tf/envs/vec_env_executor.py:16-28 the GPU parity tests replay against) pinned to the reference's OWN
``VecEnvExecutor`` over its OWN ``NormalizedEnv`` copies, run unmodified from the staged tree (oracle/ref_vecenv.py).
CPU only.  The dynamics under both are the same host float32 build, so what is compared is the executor contract
(``ts >= max_path_length`` => done, a done copy is reset and the RESET observation returned) and NormalizedEnv's
action map / reward scale; the map is evaluated in float32 numpy by the reference and inside the step by the oracle,
hence the one-rounding tolerance.
"""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGED = os.path.join(ROOT, "oracle", "_ref", "sandbox", "rocky", "tf", "envs", "vec_env_executor.py")
needs_ref = pytest.mark.skipif(not (os.path.isfile(STAGED) or os.path.isdir("/root/reference")),
                               reason="no staged reference (run oracle/make_ref.py in the build container)")


def draws_for(q, rng, T, n):
    return (rng.randn if q["reset_is_normal"] else rng.rand)(T + 1, q["reset_draws"], n).astype(np.float32)


def grid_actions(rng, T, n, da):
    """Actions on a grid of 1/64 in [-1.5, 1.5]: with the envs' integer action bounds NormalizedEnv's map
    lb + (a + 1) * 0.5 * (ub - lb) is then EXACT in float32 whatever the order of evaluation or fusion, so the
    reference's numpy evaluation and the step's own cannot differ by a rounding (some actions clip: |a| > 1)."""
    return (rng.randint(-96, 97, size=(T, n, da)) / 64.0).astype(np.float32)


@needs_ref
@pytest.mark.parametrize("kind,mpl", [(0, 9), (0, 0), (1, 7), (2, 5), (3, 6), (4, 10), (5, 11), (6, 12), (7, 8)])
def test_oracle_lock_step_is_the_reference_executors(kind, mpl):
    """Bit for bit: observations (the RESET observation after a done), rewards (scale_reward a power of two: the
    product is exact), done flags of 40 lock steps of 6 env copies."""
    from oracle import host_env as H
    from oracle import ref_vecenv
    q = H.query(kind)
    lb, ub = H.action_bounds(kind)
    assert np.all(lb == np.round(lb)) and np.all(ub == np.round(ub)) and np.abs(ub - lb).max() < 1024
    rng = np.random.RandomState(10 + kind)
    T, n = 40, 6
    draws = draws_for(q, rng, T, n)
    actions = grid_actions(rng, T, n, q["act_dim"])
    ref = ref_vecenv.run(kind, mpl, actions, draws, scale_reward=0.25)
    assert ref["modules"]["sandbox.rocky.tf.envs.vec_env_executor"] == "sandbox/rocky/tf/envs/vec_env_executor.py"
    assert ref["modules"]["rllab.envs.normalized_env"] == "rllab/envs/normalized_env.py"
    host = H.HostVecEnv(kind, n, max_path_length=mpl, normalize=True, scale_reward=0.25, auto_reset=True)
    o = host.reset(draws[0])
    assert np.array_equal(o.T.astype(np.float64), ref["obs"][0])
    n_done = 0
    for t in range(T):
        o, r, d = host.step(actions[t].T, reset_draws=draws[t + 1])
        assert np.array_equal(d.astype(bool), ref["dones"][t]), t
        assert np.array_equal(o.T.astype(np.float64), ref["obs"][t + 1]), (t, np.abs(o.T - ref["obs"][t + 1]).max())
        assert np.array_equal(r.astype(np.float64), ref["rewards"][t]), (t, r, ref["rewards"][t])
        n_done += int(d.sum())
    assert n_done >= (n if (mpl or q_terminates(kind)) else 0)                 # resets inside step() were exercised


def q_terminates(kind):
    return kind in (0, 4, 5, 6, 7)
