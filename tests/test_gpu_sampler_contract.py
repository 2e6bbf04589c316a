"""The sampler's batch-size contract: ``obtain_samples`` returns AT LEAST ``batch_size`` samples in whole paths and
stops where the reference's lock-step loop stops (``while n_samples < self.algo.batch_size``,
sandbox/rocky/tf/samplers/vectorized_sampler.py:55; rllab/algos/batch_polopt.py:23-34,
rllab/sampler/parallel_sampler.py:98-126); ``whole_paths=False`` cuts the list to exactly ``batch_size`` samples
(``truncate_paths``, parallel_sampler.py:129-155).

The checker is the reference's OWN ``VectorizedSampler.obtain_samples`` (staged, unmodified) run over a replay of the
recorded batch (oracle/ref_vecsampler.py): it must consume exactly the lock steps the engine recorded -- one fewer and
the engine overshot, one more and the replay raises -- and return exactly the paths the engine lists.
"""
import importlib

import numpy as np
import pytest
import torch

from test_ref_vecenv import needs_ref

pytestmark = [pytest.mark.gpu, needs_ref]

ENVS = dict(cartpole=("rllab.envs.box2d.cartpole_env", "CartpoleEnv"),
            double_pendulum=("rllab.envs.box2d.double_pendulum_env", "DoublePendulumEnv"),
            swimmer=("rllab.envs.mujoco.swimmer_env", "SwimmerEnv"),
            hopper=("rllab.envs.mujoco.hopper_env", "HopperEnv"),
            walker=("rllab.envs.mujoco.walker2d_env", "Walker2DEnv"))


def make_algo(name, batch_size, T, n_envs=None, whole=True, hidden=(32, 32), norm=None, seed=3, **algo_kw):
    from rllab.algos.vpg import VPG
    from rllab.baselines.linear_feature_baseline import LinearFeatureBaseline
    from rllab.envs.normalized_env import normalize
    from rllab.misc import ext
    from rllab.policies.gaussian_mlp_policy import GaussianMLPPolicy
    ext.set_seed(seed)
    mod, cls = ENVS[name]
    env = normalize(getattr(importlib.import_module(mod), cls)(), **(norm or {}))
    pol = GaussianMLPPolicy(env_spec=env.spec, hidden_sizes=hidden)
    sampler_args = dict(seed=seed)
    if n_envs is not None:
        sampler_args["n_envs"] = n_envs
    algo = VPG(env=env, policy=pol, baseline=LinearFeatureBaseline(env_spec=env.spec), batch_size=batch_size,
               max_path_length=T, n_itr=1, whole_paths=whole, sampler_args=sampler_args, **algo_kw)
    algo.start_worker()
    algo.init_opt()
    return algo


def listed(paths):
    env, t0, t1 = (x.cpu().numpy() for x in paths.index())
    return sorted(zip(env.tolist(), t0.tolist(), (t1 - t0 + 1).tolist()))


def reference_loop(traj, batch_size, T, whole=True):
    from oracle import ref_vecsampler
    out = ref_vecsampler.run(traj.dones.cpu().numpy(), traj.rewards.cpu().numpy(), batch_size, T, whole)
    assert out["modules"]["sandbox.rocky.tf.samplers.vectorized_sampler"] == "sandbox/rocky/tf/samplers/vectorized_sampler.py"
    return out, sorted(zip(out["env"].tolist(), out["t0"].tolist(), out["length"].tolist()))


def check_contract(algo, batch_size, T):
    paths = algo.sampler.obtain_samples(0)
    sd = algo.sampler.process_samples(0, paths)
    tr = paths.traj
    n_valid = int(tr.valid.sum())
    ref, ref_paths = reference_loop(tr, batch_size, T)
    assert int(ref["steps"]) == tr.T, "reference loop ran %d lock steps, the batch has %d" % (int(ref["steps"]), tr.T)
    assert listed(paths) == ref_paths
    assert n_valid == int(ref["length"].sum()) >= batch_size
    assert sd["observations"].shape[0] == n_valid and len(sd["paths"]) == len(ref_paths)
    env_i, _t0, t1 = paths.index()
    assert bool(tr.dones[t1, env_i].all())                     # every listed path ends in a done
    # reward sums per path, located through the reference's own bookkeeping
    got = {(e, a): float(tr.rewards[a:a + l, e].double().sum()) for e, a, l in listed(paths)}
    for e, a, s in zip(ref["env"], ref["t0"], ref["reward_sums"]):
        assert abs(got[(int(e), int(a))] - s) <= 1e-9 * max(1.0, abs(s))
    return paths, sd, ref


@pytest.mark.parametrize("name,batch_size,T,n_envs", [
    ("cartpole", 4000, 100, None),          # examples/trpo_cartpole.py's sizes: n_envs = 40
    ("cartpole", 4096 * 100, 100, 4096),    # BASELINE C2
    ("cartpole", 4000, 100, 512),           # far more envs than the batch needs: the loop stops after a few lock steps
    ("hopper", 6000, 60, 100),
    ("walker", 5000, 50, 100),              # the one-leg-per-lane rollout kernels carried on without a reset
])
def test_terminating_envs_return_at_least_batch_size_in_whole_paths(name, batch_size, T, n_envs, quiet_logger):
    algo = make_algo(name, batch_size, T, n_envs)
    assert algo.sampler.sampling_path(algo.policy)[0].startswith("fused rollout kernel")
    paths, sd, ref = check_contract(algo, batch_size, T)
    tr = paths.traj
    n = algo.sampler.vec_env.n
    assert batch_size <= int(tr.valid.sum()) < batch_size + n * T
    if n_envs == 512:
        assert tr.T < T
    theta0 = algo.policy.get_param_values()
    algo.optimize_policy(0, sd)
    assert np.abs(algo.policy.get_param_values() - theta0).max() > 0


def test_envs_that_never_terminate_stay_one_asynchronous_launch(quiet_logger, monkeypatch):
    """Swimmer (done always False): n_envs x max_path_length lock steps ARE batch_size whole-path samples; nothing is
    counted, nothing read back -- and the reference's loop agrees."""
    algo = make_algo("swimmer", 64 * 50, 50, 64)

    def boom(*a, **k):
        raise AssertionError("a never-terminating env's batch must not be counted on the host")
    monkeypatch.setattr(type(algo.sampler), "_finished_by_step", boom)
    paths, _sd, _ref = check_contract(algo, 64 * 50, 50)
    assert (paths.traj.T, paths.traj.N) == (50, 64) and len(paths) == 64


def test_fewer_envs_than_the_batch_needs_run_further_rounds(quiet_logger):
    """n_envs x max_path_length < batch_size (the reference caps n_envs at 100 and loops on,
    vectorized_sampler.py:22-24,55): further rounds on the same envs, no reset in between."""
    algo = make_algo("double_pendulum", 1000, 50, 8)
    paths, _sd, ref = check_contract(algo, 1000, 50)
    assert paths.traj.T == 150 and len(paths) == 24 and int(ref["length"].sum()) == 1200


@pytest.mark.parametrize("name,batch_size,T,n_envs", [("cartpole", 4000, 100, None), ("swimmer", 3000, 50, 64)])
def test_whole_paths_false_cuts_to_exactly_batch_size(name, batch_size, T, n_envs, quiet_logger):
    """``whole_paths=False``: the finished paths are collected as always, then cut to the first ``batch_size`` samples in
    path order with the last kept path truncated -- here path order is env by env."""
    algo = make_algo(name, batch_size, T, n_envs, whole=False)
    paths = algo.sampler.obtain_samples(0)
    sd = algo.sampler.process_samples(0, paths)
    tr = paths.traj
    assert int(tr.valid.sum()) == batch_size == sd["observations"].shape[0]
    ref, ref_paths = reference_loop(tr, batch_size, T, whole=True)       # the finished paths before the cut
    assert int(ref["steps"]) == tr.T
    ref_trunc, _ = reference_loop(tr, batch_size, T, whole=False)
    assert int(ref_trunc["length"].sum()) == batch_size                 # what truncate_paths leaves: the same count
    want, room = [], batch_size
    for e, a, l in ref_paths:                                             # env-major already (sorted)
        if room <= 0:
            break
        want.append((e, a, min(l, room)))
        room -= min(l, room)
    assert listed(paths) == want
    algo.optimize_policy(0, sd)


@pytest.mark.parametrize("hidden,norm,loop", [((300,), None, "hipGraph replay"),
                                              ((100, 50, 25), dict(normalize_obs=True, normalize_reward=True), "eager")])
def test_per_transition_loops_meet_the_contract_too(hidden, norm, loop, quiet_logger):
    algo = make_algo("cartpole", 2000, 50, None, hidden=hidden, norm=norm)
    name, _why = algo.sampler.sampling_path(algo.policy)
    if not name.startswith("per-transition loop (%s)" % loop):
        pytest.skip("sampled by: %s" % name)                              # (a later round fused this shape)
    check_contract(algo, 2000, 50)


def test_running_normalisation_is_carried_over_the_further_launches(quiet_logger):
    """Fused rollout under NormalizedEnv(normalize_obs, normalize_reward): the continuation feeds every estimate once
    per transition -- the estimates after the batch are those of ONE stream of traj.T + 1 observations per env."""
    norm = dict(normalize_obs=True, normalize_reward=True, obs_alpha=0.01, reward_alpha=0.01)
    algo = make_algo("cartpole", 4000, 100, None, norm=norm)
    assert algo.sampler.sampling_path(algo.policy)[0].startswith("fused rollout kernel")
    paths, _sd, _ref = check_contract(algo, 4000, 100)
    assert paths.traj.T > 100                                             # the batch did need further launches


def test_dropped_speculative_batch_gives_its_rng_counter_back(quiet_logger):
    """A prefetched rollout whose parameters were then rejected is thrown away AND leaves no trace: the next batch is
    bit-identical to that of a run that never speculated."""
    runs = []
    for speculate in (True, False):
        algo = make_algo("swimmer", 32 * 20, 20, 32)
        algo.sampler.obtain_samples(0)
        theta = algo.policy.get_param_values()
        if speculate:
            algo.sampler.prefetch(1)
            assert algo.sampler._prefetched is not None
        algo.policy.set_param_values(theta * 0.5)                         # the step that was speculated on is reverted
        runs.append(algo.sampler.obtain_samples(1).traj)
    assert torch.equal(runs[0].obs, runs[1].obs) and torch.equal(runs[0].actions, runs[1].actions)


def test_running_normalisation_is_never_prefetched(quiet_logger):
    norm = dict(normalize_obs=True, normalize_reward=True)
    algo = make_algo("cartpole", 64 * 20, 20, 64, norm=norm)
    algo.sampler.obtain_samples(0)
    before = algo.sampler.vec_env.obs_mean.clone()
    algo.sampler.prefetch(1)
    assert getattr(algo.sampler, "_prefetched", None) is None
    assert torch.equal(before, algo.sampler.vec_env.obs_mean)
