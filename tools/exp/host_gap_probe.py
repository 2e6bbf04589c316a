#!/usr/bin/env python
"""Where the host spends the time between the last line-search read and the launch of the next rollout (the device
idles meanwhile), and between the statistics read and the gradient launch: wall-clock stamps at the Python seams,
averaged over iterations.  GPU box."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rllab.algos.trpo import TRPO
from rllab.baselines.linear_feature_baseline import LinearFeatureBaseline
from rllab.envs.mujoco.half_cheetah_env import HalfCheetahEnv
from rllab.envs.normalized_env import normalize
from rllab.misc import ext, logger
from rllab.policies.gaussian_mlp_policy import GaussianMLPPolicy
from rllab_amd.optimizers import conjugate_gradient_optimizer as cgo
from rllab_amd.envs import hip_env
from rllab_amd import _lib
logger.set_quiet(True)
ext.set_seed(1)
env = normalize(HalfCheetahEnv())
policy = GaussianMLPPolicy(env_spec=env.spec, hidden_sizes=(64, 64))
algo = TRPO(env=env, policy=policy, baseline=LinearFeatureBaseline(env_spec=env.spec), batch_size=1024 * 500,
            max_path_length=500, n_itr=30, discount=0.99, gae_lambda=0.97, step_size=0.01, sampler_args=dict(n_envs=1024))
algo.start_worker(); algo.init_opt()
S = {}
def stamp(k): S.setdefault(k, []).append(time.perf_counter())
orig_lc = cgo.ConjugateGradientOptimizer._loss_constraint
def lc(self, inputs):
    r = orig_lc(self, inputs); stamp("lc_enqueued"); return r
cgo.ConjugateGradientOptimizer._loss_constraint = lc
orig_opt = cgo.ConjugateGradientOptimizer.optimize
def opt(self, *a, **k):
    r = orig_opt(self, *a, **k); stamp("optimize_returned"); return r
cgo.ConjugateGradientOptimizer.optimize = opt
orig_roll = hip_env.HipVecEnv.rollout
def roll(self, *a, **k):
    stamp("rollout_enter"); r = orig_roll(self, *a, **k); stamp("rollout_launched"); return r
hip_env.HipVecEnv.rollout = roll
pending = None
rows = []
for itr in range(25):
    paths = pending if pending is not None else algo.sampler.obtain_samples(itr)
    sd = algo.sampler.process_samples(itr, paths)
    algo.log_diagnostics(paths)
    S.clear()
    algo.optimize_policy(itr, sd); stamp("optimize_policy_returned")
    pending = algo.sampler.obtain_samples(itr + 1)
    logger.dump_tabular()
    if itr >= 5:
        t_last_lc = S["lc_enqueued"][-1]
        rows.append((S["optimize_returned"][0] - t_last_lc, S["optimize_policy_returned"][0] - S["optimize_returned"][0],
                     S["rollout_enter"][0] - S["optimize_policy_returned"][0], S["rollout_launched"][0] - S["rollout_enter"][0],
                     len(S["lc_enqueued"])))
a = np.array(rows)
print("per iteration (us): last loss-eval enqueued -> optimize() returns (incl. waiting for its result) %.0f | NPO tail %.0f | "
      "to rollout() entry %.0f | rollout() enqueue %.0f | loss evaluations %.1f" % tuple(list(a[:, :4].mean(0) * 1e6) + [a[:, 4].mean()]))
