#!/usr/bin/env python
"""Train TRPO (or VPG) on a HIP-native env with the lock-step GPU sampler.

Equivalent of the reference's examples/trpo_cartpole.py / trpo_swimmer.py with the knobs
that matter on an MI355X exposed (number of parallel envs, iterations).  The reference
scripts themselves also run unchanged: put this repository first on PYTHONPATH.

  python examples/run_trpo.py --env cartpole --n-envs 1024 --n-itr 30
  python examples/run_trpo.py --env swimmer  --n-envs 4096 --n-itr 50
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from rllab.algos.trpo import TRPO  # noqa: E402
from rllab.algos.vpg import VPG  # noqa: E402
from rllab.baselines.linear_feature_baseline import LinearFeatureBaseline  # noqa: E402
from rllab.misc import ext  # noqa: E402
from rllab.policies.gaussian_mlp_policy import GaussianMLPPolicy  # noqa: E402


def make_env(name, position_only=False, constraint=None, **norm):
    if position_only and name not in ("cartpole", "double_pendulum", "cartpole_swingup"):
        raise SystemExit("--position-only: a Box2DEnv option")
    box = dict(position_only=True) if position_only else {}
    mj = dict(constraint or {})        # HalfCheetahEnv / Walker2DEnv / HopperEnv(limit_model=.., contact_model=..): engine options
    if mj and name not in ("half_cheetah", "walker2d", "hopper"):
        raise SystemExit("--limit-model / --contact-model: options of half_cheetah, walker2d, hopper")

    def normalize(env):                           # (the wrapper's running estimates: --normalize-obs / --normalize-reward)
        from rllab.envs.normalized_env import normalize as wrap
        return wrap(env, **norm)
    if name == "cartpole":
        from rllab.envs.box2d.cartpole_env import CartpoleEnv
        return normalize(CartpoleEnv(**box)), 100
    if name == "swimmer":
        from rllab.envs.mujoco.swimmer_env import SwimmerEnv
        return normalize(SwimmerEnv()), 500
    if name == "swimmer_mujoco_limits":     # engine option: joint limits by MuJoCo's soft-constraint model
        from rllab.envs.mujoco.swimmer_env import SwimmerEnv
        return normalize(SwimmerEnv(limit_model="mujoco")), 500
    if name == "half_cheetah":
        from rllab.envs.mujoco.half_cheetah_env import HalfCheetahEnv
        return normalize(HalfCheetahEnv(**mj)), 500
    if name == "walker2d":
        from rllab.envs.mujoco.walker2d_env import Walker2DEnv
        return normalize(Walker2DEnv(**mj)), 500
    if name == "hopper":
        from rllab.envs.mujoco.hopper_env import HopperEnv
        return normalize(HopperEnv(**mj)), 500
    if name == "inverted_double_pendulum":
        from rllab.envs.mujoco.inverted_double_pendulum_env import InvertedDoublePendulumEnv
        return normalize(InvertedDoublePendulumEnv()), 100
    if name == "double_pendulum":
        from rllab.envs.box2d.double_pendulum_env import DoublePendulumEnv
        return normalize(DoublePendulumEnv(**box)), 100
    if name == "cartpole_swingup":
        from rllab.envs.box2d.cartpole_swingup_env import CartpoleSwingupEnv
        return normalize(CartpoleSwingupEnv(**box)), 500
    raise SystemExit("unknown env %r" % name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="cartpole")
    ap.add_argument("--algo", default="trpo", choices=["trpo", "vpg"])
    ap.add_argument("--n-envs", type=int, default=1024)
    ap.add_argument("--n-itr", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--hidden", default="32", help="hidden sizes: one number H for (H, H), or a list like 100,50,25")
    ap.add_argument("--one-hidden-layer", action="store_true", help="--hidden H means (H,) instead of (H, H)")
    ap.add_argument("--nonlinearity", default="tanh", choices=["tanh", "relu"], help="hidden_nonlinearity")
    ap.add_argument("--normalize-obs", action="store_true", help="NormalizedEnv(normalize_obs=True)")
    ap.add_argument("--normalize-reward", action="store_true", help="NormalizedEnv(normalize_reward=True)")
    ap.add_argument("--position-only", action="store_true", help="Box2DEnv(position_only=True)")
    ap.add_argument("--adaptive-std", action="store_true", help="GaussianMLPPolicy(adaptive_std=True)")
    ap.add_argument("--gae-lambda", type=float, default=1.0)
    ap.add_argument("--limit-model", default=None, choices=["penalty", "mujoco"], help="legged envs: joint limits")
    ap.add_argument("--contact-model", default=None, choices=["penalty", "mujoco"], help="legged envs: floor contacts")
    ap.add_argument("--csv", default=None, help="write the tabular log (one row per iteration) to this file")
    ap.add_argument("--quiet", action="store_true")
    args = ap.parse_args()
    from rllab.misc import logger
    if args.csv:
        logger.add_tabular_output(args.csv)
    if args.quiet:
        logger.set_quiet(True)
    ext.set_seed(args.seed)
    norm = {k: True for k in ("normalize_obs", "normalize_reward") if getattr(args, k)}
    constraint = {k: getattr(args, k) for k in ("limit_model", "contact_model") if getattr(args, k)}
    env, horizon = make_env(args.env, position_only=args.position_only, constraint=constraint, **norm)
    hs = tuple(int(h) for h in str(args.hidden).split(","))
    extra = {}
    if args.nonlinearity == "relu":
        from rllab.core.network import rectify
        extra["hidden_nonlinearity"] = rectify
    policy = GaussianMLPPolicy(env_spec=env.spec,
                               hidden_sizes=hs * 2 if len(hs) == 1 and not args.one_hidden_layer else hs,
                               adaptive_std=args.adaptive_std, **extra)
    baseline = LinearFeatureBaseline(env_spec=env.spec)
    kw = dict(env=env, policy=policy, baseline=baseline, batch_size=args.n_envs * horizon,
              max_path_length=horizon, n_itr=args.n_itr, discount=0.99, gae_lambda=args.gae_lambda,
              sampler_args=dict(n_envs=args.n_envs))
    algo = TRPO(step_size=0.01, **kw) if args.algo == "trpo" else VPG(**kw)
    algo.train()


if __name__ == "__main__":
    main()
