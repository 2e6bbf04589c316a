#!/usr/bin/env python
"""Which GaussianMLPPolicy constructor options stay on the kernels (fused rollout / fused update), and what an iteration
costs when they do not: prints one line per configuration.  GPU box."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rllab.algos.trpo import TRPO
from rllab.baselines.linear_feature_baseline import LinearFeatureBaseline
from rllab.envs.mujoco.swimmer_env import SwimmerEnv
from rllab.envs.normalized_env import normalize
from rllab.misc import ext, logger
from rllab.policies.gaussian_mlp_policy import GaussianMLPPolicy
from rllab.core.network import MLP
logger.set_quiet(True)
CONFIGS = [
    ("default (32, 32)", dict()),
    ("hidden (64,)", dict(hidden_sizes=(64,))),
    ("hidden (32,)", dict(hidden_sizes=(32,))),
    ("hidden (20,) relu", dict(hidden_sizes=(20,), hidden_nonlinearity="relu")),
    ("hidden (100,)", dict(hidden_sizes=(100,))),
    ("hidden (50, 50)", dict(hidden_sizes=(50, 50))),
    ("hidden (16, 16, 16)", dict(hidden_sizes=(16, 16, 16))),
    ("hidden (200, 100)", dict(hidden_sizes=(200, 100))),
    ("learn_std=False", dict(learn_std=False)),
    ("init_std=0.5, min_std=1e-2", dict(init_std=0.5, min_std=1e-2)),
    ("std_share_network", dict(std_share_network=True)),
    ("adaptive_std, std (16,)", dict(adaptive_std=True, std_hidden_sizes=(16,))),
    ("hidden_nonlinearity=relu", dict(hidden_nonlinearity="relu")),
]
for name, kw in CONFIGS:
    try:
        ext.set_seed(1)
        env = normalize(SwimmerEnv())
        if kw.get("hidden_nonlinearity") == "relu":
            from rllab.core.network import rectify
            kw = dict(kw, hidden_nonlinearity=rectify)
        policy = GaussianMLPPolicy(env_spec=env.spec, **kw)
        algo = TRPO(env=env, policy=policy, baseline=LinearFeatureBaseline(env_spec=env.spec), batch_size=512 * 100,
                    max_path_length=100, n_itr=3, discount=0.99, step_size=0.01, sampler_args=dict(n_envs=512))
        algo.start_worker(); algo.init_opt()
        fr = algo.sampler._takes_fused_rollout(policy)
        why = algo.sampler.sampling_path(policy)[1]
        fu = type(getattr(algo.optimizer, "_fused", None)).__name__
        ts = []
        for itr in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            paths = algo.sampler.obtain_samples(itr); sd = algo.sampler.process_samples(itr, paths)
            algo.log_diagnostics(paths); algo.optimize_policy(itr, sd); torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3); logger.dump_tabular()
        print("%-32s fused rollout %-5s fused update %-22s iteration %.1f ms%s" % (
            name, fr, fu, min(ts), "" if why is None else "   [" + why[:90] + "]"), flush=True)
    except Exception as e:
        print("%-32s ERROR %s: %s" % (name, type(e).__name__, str(e)[:150]), flush=True)
