import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from rllab.algos.trpo import TRPO
from rllab.baselines.linear_feature_baseline import LinearFeatureBaseline
from rllab.envs.mujoco.swimmer_env import SwimmerEnv
from rllab.envs.normalized_env import normalize
from rllab.misc import ext, logger
from rllab.policies.gaussian_mlp_policy import GaussianMLPPolicy
logger.set_quiet(True)
for hidden in ((100, 50, 25), (32, 32)):
    ext.set_seed(1)
    env = normalize(SwimmerEnv())
    policy = GaussianMLPPolicy(env_spec=env.spec, hidden_sizes=hidden, adaptive_std=True)
    algo = TRPO(env=env, policy=policy, baseline=LinearFeatureBaseline(env_spec=env.spec), batch_size=1024 * 500,
                max_path_length=500, n_itr=5, discount=0.99, step_size=0.01, sampler_args=dict(n_envs=1024))
    algo.start_worker(); algo.init_opt()
    print(hidden, "fused rollout:", algo.sampler._takes_fused_rollout(policy), "fused update:", type(algo.optimizer._fused).__name__)
    for itr in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        paths = algo.sampler.obtain_samples(itr); torch.cuda.synchronize(); t1 = time.perf_counter()
        sd = algo.sampler.process_samples(itr, paths); algo.log_diagnostics(paths); torch.cuda.synchronize(); t2 = time.perf_counter()
        algo.optimize_policy(itr, sd); torch.cuda.synchronize(); t3 = time.perf_counter()
        logger.dump_tabular()
        print("  itr", itr, "sample %.1f ms process %.1f ms update %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
