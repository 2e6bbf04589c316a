#!/usr/bin/env python
"""Which env / algorithm options keep an iteration on the fused path (one rollout launch + kernel update): one line per
configuration with the iteration time.  GPU box."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rllab.algos.trpo import TRPO
from rllab.algos.tnpg import TNPG
from rllab.algos.ppo import PPO
from rllab.algos.vpg import VPG
from rllab.baselines.linear_feature_baseline import LinearFeatureBaseline
from rllab.baselines.zero_baseline import ZeroBaseline
from rllab.baselines.gaussian_mlp_baseline import GaussianMLPBaseline
from rllab.envs.mujoco.swimmer_env import SwimmerEnv
from rllab.envs.mujoco.half_cheetah_env import HalfCheetahEnv
from rllab.envs.box2d.cartpole_env import CartpoleEnv
from rllab.envs.normalized_env import normalize
from rllab.misc import ext, logger
from rllab.policies.gaussian_mlp_policy import GaussianMLPPolicy
logger.set_quiet(True)

def lfb(env): return LinearFeatureBaseline(env_spec=env.spec)
CONFIGS = [
    ("TRPO swimmer (reference)", TRPO, lambda: normalize(SwimmerEnv()), lfb, dict()),
    ("TNPG", TNPG, lambda: normalize(SwimmerEnv()), lfb, dict()),
    ("PPO", PPO, lambda: normalize(SwimmerEnv()), lfb, dict()),
    ("VPG", VPG, lambda: normalize(SwimmerEnv()), lfb, dict()),
    ("TRPO zero baseline", TRPO, lambda: normalize(SwimmerEnv()), lambda e: ZeroBaseline(env_spec=e.spec), dict()),
    ("TRPO GaussianMLPBaseline", TRPO, lambda: normalize(SwimmerEnv()), lambda e: GaussianMLPBaseline(env_spec=e.spec), dict()),
    ("TRPO gae 0.97, no centring", TRPO, lambda: normalize(SwimmerEnv()), lfb, dict(gae_lambda=0.97, center_adv=False)),
    ("TRPO positive_adv", TRPO, lambda: normalize(SwimmerEnv()), lfb, dict(positive_adv=True)),
    ("TRPO whole_paths=False", TRPO, lambda: normalize(SwimmerEnv()), lfb, dict(whole_paths=False)),
    ("TRPO store_paths", TRPO, lambda: normalize(SwimmerEnv()), lfb, dict(store_paths=True)),
    ("TRPO un-normalized env", TRPO, lambda: SwimmerEnv(), lfb, dict()),
    ("TRPO normalize_obs", TRPO, lambda: normalize(SwimmerEnv(), normalize_obs=True), lfb, dict()),
    ("TRPO normalize_reward", TRPO, lambda: normalize(SwimmerEnv(), normalize_reward=True), lfb, dict()),
    ("TRPO scale_reward=0.1", TRPO, lambda: normalize(SwimmerEnv(), scale_reward=0.1), lfb, dict()),
    ("TRPO swimmer ctrl_cost 0.1, action_noise", TRPO, lambda: normalize(SwimmerEnv(ctrl_cost_coeff=0.1, action_noise=0.1)), lfb, dict()),
    ("TRPO cartpole obs_noise", TRPO, lambda: normalize(CartpoleEnv(obs_noise=0.05)), lfb, dict()),
    ("TRPO half_cheetah", TRPO, lambda: normalize(HalfCheetahEnv()), lfb, dict()),
    ("TRPO subsample_factor 0.5", TRPO, lambda: normalize(SwimmerEnv()), lfb, dict(optimizer_args=dict(subsample_factor=0.5))),
]
for name, cls, mk_env, mk_base, kw in CONFIGS:
    try:
        ext.set_seed(1)
        env = mk_env()
        policy = GaussianMLPPolicy(env_spec=env.spec, hidden_sizes=(32, 32))
        algo = cls(env=env, policy=policy, baseline=mk_base(env), batch_size=512 * 100, max_path_length=100, n_itr=3,
                   discount=0.99, sampler_args=dict(n_envs=512), **kw)
        algo.start_worker(); algo.init_opt()
        fr = algo.sampler._takes_fused_rollout(policy)
        fu = type(getattr(algo.optimizer, "_fused", None)).__name__
        ts = []
        for itr in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            paths = algo.sampler.obtain_samples(itr); sd = algo.sampler.process_samples(itr, paths)
            algo.log_diagnostics(paths); algo.optimize_policy(itr, sd); torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3); logger.dump_tabular()
        print("%-42s fused rollout %-5s update %-20s iteration %.1f ms" % (name, fr, fu, min(ts)), flush=True)
    except Exception as e:
        print("%-42s ERROR %s: %s" % (name, type(e).__name__, str(e)[:200]), flush=True)
