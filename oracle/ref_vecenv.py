#!/usr/bin/env python
"""The reference's OWN lock-step executor over its OWN NormalizedEnv copies, run on recorded actions
(TEST INFRASTRUCTURE, never imported by the product; a child process because ``rllab`` must resolve
to the reference here and to the product's alias package in the test that calls it).

What runs is the reference's code, staged byte for byte by oracle/make_ref.py:

    VecEnvExecutor.reset / step              sandbox/rocky/tf/envs/vec_env_executor.py:8-33
        (ts += 1; dones[ts >= max_path_length] = True; a done copy is reset and the RESET observation returned)
    NormalizedEnv.reset / step               rllab/envs/normalized_env.py:33-92
        (affine action map + clip; running estimates fed by every observation a copy produces, the terminal one
         included, and once more by the reset observation; reward normalised, then scaled)
    tensor_utils.stack_tensor_dict_list      sandbox/rocky/tf/misc/tensor_utils.py  (imports tensorflow at module
                                             level only: stubbed by oracle/ref_shim.py)

What cannot be the reference's (pybox2d / MuJoCo 1.31 are absent): the dynamics under the wrapper --
``DrawnHostEnv``, the float32 host build of this repo's env headers (oracle/host_env.py) behind the reference's ``Env``
interface, its resets fed from a table of injected draws indexed by the lock step (slice 0 = the first reset, slice
t + 1 = a reset after step t), exactly the table the GPU rollout was given.

    python oracle/ref_vecenv.py IN.npz OUT.npz
IN : kind, max_path_length, actions [T, n, Da], reset_draws [T+1, R, n], scale_reward, normalize_obs, normalize_reward,
     obs_alpha, reward_alpha, obs_mean0 / obs_var0 [n, Do], reward_mean0 / reward_var0 [n]  (estimates to resume from)
OUT: obs [T+1, n, Do] (slot 0 = reset(), slot t + 1 = what step t returned), rewards [T, n], dones [T, n],
     obs_mean / obs_var [n, Do], reward_mean / reward_var [n]  (every copy's estimates after the last step), modules
"""
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def ref_root():
    staged = os.path.join(HERE, "_ref")
    if os.path.isfile(os.path.join(staged, "sandbox", "rocky", "tf", "envs", "vec_env_executor.py")):
        return staged
    if os.path.isdir("/root/reference/sandbox"):
        return "/root/reference"
    raise RuntimeError("no staged reference: run `python oracle/make_ref.py` in the build container")


def child_main(path_in, path_out):
    sys.path.insert(0, ROOT)
    from oracle import ref_shim
    ref_shim.install(ref_root())
    import numpy as np
    import rllab
    assert os.path.abspath(rllab.__file__).startswith(ref_root()), rllab.__file__
    from rllab.core.serializable import Serializable
    from rllab.envs.base import Env, Step
    from rllab.envs.normalized_env import NormalizedEnv
    from rllab.spaces.box import Box
    from sandbox.rocky.tf.envs.vec_env_executor import VecEnvExecutor
    from oracle import host_env as H

    z = np.load(path_in)
    kind, mpl = int(z["kind"]), int(z["max_path_length"])
    actions, draws = z["actions"], z["reset_draws"]
    T, n, _da = actions.shape
    clock = dict(slice=0)

    class DrawnHostEnv(Env, Serializable):
        """Env copy ``i``: host float32 dynamics, reset k uses the draws of the lock step it happens at."""

        def __init__(self, i):
            Serializable.quick_init(self, locals())
            self.i = i
            self._env = H.HostEnv(kind, np.float32, normalize=False)
            q = self._env.q
            lb, ub = H.action_bounds(kind)
            self._action_space = Box(lb.astype(np.float32), ub.astype(np.float32))
            self._observation_space = Box(-np.inf * np.ones(q["obs_dim"]), np.inf * np.ones(q["obs_dim"]))

        @property
        def action_space(self):
            return self._action_space

        @property
        def observation_space(self):
            return self._observation_space

        # observations and rewards leave as float64 (the float32 values, widened): the reference's envs hand out
        # float64, and NormalizedEnv's estimate updates are float64 arithmetic on them (a float32 array would make
        # numpy evaluate alpha * obs in float32)
        def reset(self):
            return self._env.reset(draws[clock["slice"], :, self.i]).astype(np.float64)

        def step(self, action):
            o, r, d = self._env.step(np.asarray(action, dtype=np.float32))
            return Step(observation=o.astype(np.float64), reward=float(np.float32(r)), done=bool(d))

    envs = []
    for i in range(n):
        e = NormalizedEnv(DrawnHostEnv(i), scale_reward=float(z["scale_reward"]), normalize_obs=bool(z["normalize_obs"]),
                          normalize_reward=bool(z["normalize_reward"]), obs_alpha=float(z["obs_alpha"]),
                          reward_alpha=float(z["reward_alpha"]))
        # resume from given estimates (what unpickling a snapshot does for the observation pair, normalized_env.py:65-68)
        e._obs_mean, e._obs_var = z["obs_mean0"][i].astype(np.float64), z["obs_var0"][i].astype(np.float64)
        e._reward_mean, e._reward_var = float(z["reward_mean0"][i]), float(z["reward_var0"][i])
        envs.append(e)
    vec = VecEnvExecutor(envs=envs, max_path_length=mpl if mpl > 0 else None)
    do = envs[0].observation_space.flat_dim
    obs = np.zeros((T + 1, n, do))
    rew, done = np.zeros((T, n)), np.zeros((T, n), dtype=bool)
    clock["slice"] = 0
    obs[0] = np.asarray(vec.reset())
    for t in range(T):
        clock["slice"] = t + 1                      # a reset inside step t draws slice t + 1
        o, r, d, _infos = vec.step(list(actions[t]))
        obs[t + 1], rew[t], done[t] = np.asarray(o), r, d
    np.savez(path_out, obs=obs, rewards=rew, dones=done,
             obs_mean=np.stack([e._obs_mean for e in envs]), obs_var=np.stack([e._obs_var for e in envs]),
             reward_mean=np.array([e._reward_mean for e in envs]), reward_var=np.array([e._reward_var for e in envs]),
             modules=json.dumps({m: os.path.relpath(sys.modules[m].__file__, ref_root())
                                 for m in ["sandbox.rocky.tf.envs.vec_env_executor", "rllab.envs.normalized_env"]}))


def run(kind, max_path_length, actions, reset_draws, scale_reward=1.0, normalize_obs=False, normalize_reward=False,
        obs_alpha=0.001, reward_alpha=0.001, obs_mean0=None, obs_var0=None, reward_mean0=None, reward_var0=None,
        timeout=600):
    """Parent side: run the reference executor in a child process, return its arrays as a dict."""
    import numpy as np
    sys.path.insert(0, ROOT) if ROOT not in sys.path else None
    from oracle import host_env as H
    T, n, _ = np.asarray(actions).shape
    do = H.query(kind)["obs_dim"]
    with tempfile.TemporaryDirectory() as tmp:
        pin, pout = os.path.join(tmp, "in.npz"), os.path.join(tmp, "out.npz")
        np.savez(pin, kind=kind, max_path_length=max_path_length, actions=np.asarray(actions, np.float32),
                 reset_draws=np.asarray(reset_draws, np.float32), scale_reward=scale_reward,
                 normalize_obs=normalize_obs, normalize_reward=normalize_reward, obs_alpha=obs_alpha,
                 reward_alpha=reward_alpha,
                 obs_mean0=np.zeros((n, do)) if obs_mean0 is None else obs_mean0,
                 obs_var0=np.ones((n, do)) if obs_var0 is None else obs_var0,
                 reward_mean0=np.zeros(n) if reward_mean0 is None else reward_mean0,
                 reward_var0=np.ones(n) if reward_var0 is None else reward_var0)
        env = dict(os.environ)
        env.pop("PYTHONPATH", None)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), pin, pout], stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, env=env, timeout=timeout, cwd=tmp, universal_newlines=True)
        if p.returncode != 0:
            raise RuntimeError("reference VecEnvExecutor child failed (rc %d):\n%s\n%s"
                               % (p.returncode, p.stdout[-2000:], p.stderr[-4000:]))
        out = dict(np.load(pout))
    out["modules"] = json.loads(str(out["modules"]))
    return out


if __name__ == "__main__":
    child_main(sys.argv[1], sys.argv[2])
