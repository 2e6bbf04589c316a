#!/usr/bin/env python
"""The reference's OWN vectorised sampler loop deciding where a batch ends (TEST INFRASTRUCTURE, never imported by the
product; a child process because ``rllab`` must resolve to the reference here).

What runs is the reference's code, staged byte for byte by oracle/make_ref.py:

    VectorizedSampler.obtain_samples         sandbox/rocky/tf/samplers/vectorized_sampler.py:43-108
        (``while n_samples < self.algo.batch_size``; a path is appended and counted when its env reports done;
         paths still running when the loop stops are dropped)
    truncate_paths                           rllab/sampler/parallel_sampler.py:129-155   (``whole_paths=False``)
    tensor_utils.* / Box.flatten_n / EnvSpec

It is driven by a REPLAY of a recorded batch: the executor it steps hands out the recorded observations, rewards and
done flags of lock step t (whatever actions it is given), the policy hands out the recorded actions; every transition
carries its (env, t) in ``env_infos`` so the returned paths can be located in the recorded planes.  If the loop asks for
a lock step the recording does not hold, the replay raises -- the recorded batch was too short for the contract.

    python oracle/ref_vecsampler.py IN.npz OUT.npz
IN : dones [T, n] uint8, rewards [T, n], batch_size, max_path_length, whole_paths
OUT: env [P], t0 [P], length [P]  (the returned paths in the reference's order), steps (lock steps the loop ran), modules
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
    if os.path.isfile(os.path.join(staged, "sandbox", "rocky", "tf", "samplers", "vectorized_sampler.py")):
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
    from rllab.envs.env_spec import EnvSpec
    from rllab.misc import logger
    from rllab.sampler.parallel_sampler import truncate_paths
    from rllab.spaces.box import Box
    from sandbox.rocky.tf.samplers.vectorized_sampler import VectorizedSampler
    logger.set_log_tabular_only(True)

    z = np.load(path_in)
    dones, rewards = z["dones"].astype(bool), z["rewards"].astype(np.float64)
    T, n = dones.shape
    clock = dict(t=0)

    class Replay(object):
        """VecEnvExecutor surface over the recording: observation of env i at lock step t is the pair (i, t)."""
        num_envs = n

        def reset(self):
            clock["t"] = 0
            return [np.array([float(i), 0.0]) for i in range(n)]

        def step(self, action_n):
            t = clock["t"]
            if t >= T:
                raise RuntimeError("the reference's loop asks for lock step %d, the recorded batch has %d" % (t, T))
            clock["t"] = t + 1
            obs = [np.array([float(i), float(t + 1)]) for i in range(n)]
            return obs, rewards[t], dones[t], dict(env=np.arange(n), t=np.full(n, t))

        def terminate(self):
            pass

    class Env(object):
        vectorized = True
        spec = EnvSpec(observation_space=Box(-np.inf * np.ones(2), np.inf * np.ones(2)), action_space=Box(-np.ones(1), np.ones(1)))

        def vec_env_executor(self, n_envs, max_path_length):
            assert n_envs == n
            return Replay()

    class Policy(object):
        def reset(self, dones=None):
            pass

        def get_actions(self, obses):
            return np.zeros((len(obses), 1)), dict(mean=np.zeros((len(obses), 1)))

    class Algo(object):
        batch_size, max_path_length = int(z["batch_size"]), int(z["max_path_length"])
        env, policy = Env(), Policy()

    sampler = VectorizedSampler(Algo(), n_envs=n)
    sampler.start_worker()
    paths = sampler.obtain_samples(0)
    steps = clock["t"]
    if not bool(z["whole_paths"]):
        paths = truncate_paths(paths, Algo.batch_size)
    sampler.shutdown_worker()
    np.savez(path_out, env=np.array([int(p["env_infos"]["env"][0]) for p in paths]),
             t0=np.array([int(p["env_infos"]["t"][0]) for p in paths]),
             length=np.array([len(p["rewards"]) for p in paths]), steps=steps,
             reward_sums=np.array([float(np.sum(p["rewards"])) for p in paths]),
             modules=json.dumps({m: os.path.relpath(sys.modules[m].__file__, ref_root())
                                 for m in ["sandbox.rocky.tf.samplers.vectorized_sampler", "rllab.sampler.parallel_sampler"]}))


def run(dones, rewards, batch_size, max_path_length, whole_paths=True, timeout=600):
    """Parent side: where does the reference's loop end on this recording?  Returns its arrays as a dict."""
    import numpy as np
    with tempfile.TemporaryDirectory() as tmp:
        pin, pout = os.path.join(tmp, "in.npz"), os.path.join(tmp, "out.npz")
        np.savez(pin, dones=np.asarray(dones, np.uint8), rewards=np.asarray(rewards, np.float64), batch_size=batch_size,
                 max_path_length=max_path_length, whole_paths=whole_paths)
        env = dict(os.environ)
        env.pop("PYTHONPATH", None)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), pin, pout], stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, env=env, timeout=timeout, cwd=tmp, universal_newlines=True)
        if p.returncode != 0:
            raise RuntimeError("reference VectorizedSampler child failed (rc %d):\n%s\n%s"
                               % (p.returncode, p.stdout[-2000:], p.stderr[-4000:]))
        out = dict(np.load(pout))
    out["modules"] = json.loads(str(out["modules"]))
    return out


if __name__ == "__main__":
    child_main(sys.argv[1], sys.argv[2])
