#!/usr/bin/env python
"""Time the fused rollout launch (HIP events, 10 launches) for every experiment library build/exp/lib_*.so
(each swapped in as rllab_amd/librllab_amd.so in a child process -- run on the GPU box's scratch copy only)."""
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
import importlib
ENVS = dict(swimmer=("rllab_amd.envs.mujoco.swimmer_env", "SwimmerEnv"), half_cheetah=("rllab_amd.envs.mujoco.half_cheetah_env", "HalfCheetahEnv"),
            walker2d=("rllab_amd.envs.mujoco.walker2d_env", "Walker2DEnv"), hopper=("rllab_amd.envs.mujoco.hopper_env", "HopperEnv"),
            cartpole=("rllab_amd.envs.box2d.cartpole_env", "CartpoleEnv"), double_pendulum=("rllab_amd.envs.box2d.double_pendulum_env", "DoublePendulumEnv"))
from rllab_amd.envs.normalized_env import normalize
from rllab_amd.envs.hip_env import HipVecEnv
from rllab_amd.policies.gaussian_mlp_policy import GaussianMLPPolicy
kind, n, T, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
mod, cls = ENVS[kind]
env = normalize(getattr(importlib.import_module(mod), cls)())
np.random.seed(0)
pol = GaussianMLPPolicy(env.spec, hidden_sizes=(h, h))
vec = env.vec_env_executor(n_envs=n, max_path_length=T)
for _ in range(2): vec.rollout(pol, T)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): tr = vec.rollout(pol, T)
e1.record(); torch.cuda.synchronize()
print("%%s %%s n=%%d T=%%d: %%.3f ms per rollout  (mean reward %%.6f)" %% (sys.argv[5], kind, n, T, e0.elapsed_time(e1) / 10, float(tr.rewards.mean())))
""" % ROOT


def main():
    libs = sorted(glob.glob(os.path.join(ROOT, "build", "exp", "lib_*.so")))
    target = os.path.join(ROOT, "rllab_amd", "librllab_amd.so")
    shutil.copy(target, target + ".orig")
    cfgs = sys.argv[1:] or ["swimmer,4096,500,32"]
    try:
        for lib in [target + ".orig"] + libs:
            shutil.copy(lib, target)
            for cfg in cfgs:
                kind, n, T, h = cfg.split(",")
                subprocess.call([sys.executable, "-c", CHILD, kind, n, T, h, os.path.basename(lib)])
    finally:
        shutil.copy(target + ".orig", target)


if __name__ == "__main__":
    main()
