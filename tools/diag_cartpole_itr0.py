"""Diagnostic: iteration-0 statistics of a fresh policy on the HIP Cartpole vs the CPU sampler (same theta)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rllab.algos.trpo import TRPO
from rllab.baselines.linear_feature_baseline import LinearFeatureBaseline
from rllab.envs.box2d.cartpole_env import CartpoleEnv
from rllab.envs.normalized_env import normalize
from rllab.misc import ext, logger
from rllab.policies.gaussian_mlp_policy import GaussianMLPPolicy
from oracle import cpu_sampler
logger.set_quiet(True)
for seed in (1, 2, 3):
    for n_envs, T in ((40, 100), (4096, 100)):
        ext.set_seed(seed)
        env = normalize(CartpoleEnv())
        policy = GaussianMLPPolicy(env_spec=env.spec, hidden_sizes=(32, 32))
        algo = TRPO(env=env, policy=policy, baseline=LinearFeatureBaseline(env_spec=env.spec), batch_size=n_envs * T,
                    max_path_length=T, n_itr=1, discount=0.99, step_size=0.01, sampler_args=dict(seed=seed, n_envs=n_envs))
        algo.start_worker(); algo.init_opt()
        paths = algo.sampler.obtain_samples(0)
        sd = algo.sampler.process_samples(0, paths)
        tab = logger.get_tabular(); logger.dump_tabular()
        traj = paths.traj
        done = traj.dones.cpu().numpy()
        # path lengths of complete paths per env column
        lens = []
        for n in range(min(n_envs, 512)):
            idx = np.flatnonzero(done[:, n])
            prev = -1
            for i in idx:
                lens.append(i - prev); prev = i
        lens = np.array(lens)
        print("seed", seed, "n_envs", n_envs, {k: tab[k] for k in ("AverageReturn", "StdReturn", "MinReturn", "MaxReturn", "NumTrajs")},
              "mean len", lens.mean(), "hist", np.round(np.bincount(lens)[:8] / len(lens), 3))
        algo.shutdown_worker()
    theta = policy.get_param_values()
    ps, _, _ = cpu_sampler.sample_paths(0, theta, 20000, 100, n_parallel=1, seed=seed)
    rets = np.array([p["rewards"].sum() for p in ps]); L = np.array([len(p["rewards"]) for p in ps])
    print("   cpu sampler same theta: AverageReturn %.2f Std %.2f Min %.2f Max %.2f n %d mean len %.2f hist" %
          (rets.mean(), rets.std(), rets.min(), rets.max(), len(ps), L.mean()), np.round(np.bincount(L)[:8] / len(L), 3))
