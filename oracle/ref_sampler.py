#!/usr/bin/env python
"""``cpu_baseline`` with ``kind: "reference"`` -- the reference's UNMODIFIED CPU sampler
timed on this box's host cores (TEST INFRASTRUCTURE, never imported by the product).

What runs is the reference's own code, staged byte for byte by oracle/make_ref.py
(``oracle/_ref/rllab/...``; in the build container /root/reference is used directly):

    parallel_sampler.initialize / populate_task / set_seed / sample_paths
                                             rllab/sampler/parallel_sampler.py:18-126
    StatefulPool.run_each / run_collect      rllab/sampler/stateful_pool.py:48-143
    rollout                                  rllab/sampler/utils.py:6-43
    NormalizedEnv (normalize)                rllab/envs/normalized_env.py:8-95
    Env / Step / EnvSpec / Box / Policy / Parameterized / Serializable / tensor_utils

What cannot be the reference's (its arithmetic lives in pybox2d / MuJoCo 1.31 / Theano,
all absent) and is supplied here behind the reference's own interfaces:
  * ``HostRefEnv(Env)``: the float64 host build of this repo's env dynamics
    (oracle/host_env.py, one ctypes call per step as the reference pays one SWIG /
    ctypes call per step), reset draws from ``np.random`` as the reference envs do;
  * ``NumpyGaussianMLPPolicy(Policy)``: batch-1 NumPy forward pass with the
    reference's flat parameter layout and ``get_action`` rule
    (rllab/policies/gaussian_mlp_policy.py:125-130).
So the number is an UPPER bound on the true reference stack's throughput.

Runs in its own process (``python oracle/ref_sampler.py ...``): ``rllab`` must resolve
to the reference here and to the product's alias package everywhere else.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
MARK = "REF_SAMPLER_JSON "


def ref_root():
    staged = os.path.join(HERE, "_ref")
    if os.path.isdir(os.path.join(staged, "rllab", "sampler")):
        return staged
    if os.path.isdir("/root/reference/rllab"):
        return "/root/reference"
    raise RuntimeError("no staged reference: run `python oracle/make_ref.py` in the build container")


# ----------------------------------------------------------------------------------------------
# child process: everything below imports the reference's `rllab`
# ----------------------------------------------------------------------------------------------
def _child_classes():
    import numpy as np
    from rllab.core.serializable import Serializable
    from rllab.envs.base import Env, Step
    from rllab.policies.base import Policy
    from rllab.spaces.box import Box
    from oracle import host_env as H
    from oracle import np_reference as R

    class HostRefEnv(Env, Serializable):
        """The host (float64) build of the engine's dynamics behind the reference's Env interface."""

        def __init__(self, kind):
            Serializable.quick_init(self, locals())
            self.kind = kind
            self._env = H.HostEnv(kind, np.float64, normalize=False)
            q = self._env.q
            lb, ub = H.action_bounds(kind)
            self._action_space = Box(lb, ub)
            self._observation_space = Box(-np.inf * np.ones(q["obs_dim"]), np.inf * np.ones(q["obs_dim"]))
            self._n_draws, self._normal = q["reset_draws"], q["reset_is_normal"]

        @property
        def action_space(self):
            return self._action_space

        @property
        def observation_space(self):
            return self._observation_space

        def reset(self):
            n = self._n_draws
            draws = np.random.normal(size=n) if self._normal else np.random.uniform(size=n)
            return self._env.reset(draws)

        def step(self, action):
            o, r, d = self._env.step(action)
            return Step(observation=o, reward=r, done=d)

    class NumpyGaussianMLPPolicy(Policy, Serializable):
        def __init__(self, env_spec, hidden_sizes=(32, 32)):
            Serializable.quick_init(self, locals())
            Policy.__init__(self, env_spec)
            self._net = R.NumpyGaussianMLP(env_spec.observation_space.flat_dim,
                                           env_spec.action_space.flat_dim, tuple(hidden_sizes))

        # Parameterized's own accessors walk Theano shared variables; the flat vector IS the state here
        def get_param_values(self, **tags):
            return self._net.get_param_values()

        def set_param_values(self, flattened_params, **tags):
            self._net.set_param_values(flattened_params)

        def get_params_internal(self, **tags):
            return []

        def get_action(self, observation):
            flat_obs = self.observation_space.flatten(observation)
            mean, log_std = [x[0] for x in self._net.dist_info([flat_obs])]
            rnd = np.random.normal(size=mean.shape)
            action = rnd * np.exp(log_std) + mean
            return action, dict(mean=mean, log_std=log_std)

    return HostRefEnv, NumpyGaussianMLPPolicy


def child_main(args):
    sys.path.insert(0, ROOT)
    from oracle import ref_shim
    ref_shim.install(ref_root())
    import numpy as np
    import rllab
    assert os.path.abspath(rllab.__file__).startswith(ref_root()), rllab.__file__
    from rllab.envs.normalized_env import normalize
    from rllab.sampler import parallel_sampler
    from rllab.sampler import stateful_pool
    HostRefEnv, NumpyGaussianMLPPolicy = _child_classes()
    # make the two classes picklable by reference for the worker processes (fork: same modules)
    me = sys.modules[__name__]
    HostRefEnv.__module__ = NumpyGaussianMLPPolicy.__module__ = __name__
    HostRefEnv.__qualname__, NumpyGaussianMLPPolicy.__qualname__ = "HostRefEnv", "NumpyGaussianMLPPolicy"
    me.HostRefEnv, me.NumpyGaussianMLPPolicy = HostRefEnv, NumpyGaussianMLPPolicy

    theta = np.load(args.theta)
    hidden = tuple(int(x) for x in args.hidden.split(","))
    env = normalize(HostRefEnv(args.kind))
    policy = NumpyGaussianMLPPolicy(env_spec=env.spec, hidden_sizes=hidden)
    assert policy.get_param_values().shape == theta.shape, (policy.get_param_values().shape, theta.shape)
    t_init = time.time()
    parallel_sampler.initialize(n_parallel=args.n_parallel)          # parallel_sampler.py:18-20
    parallel_sampler.set_seed(args.seed)                              # :84-88, worker i gets seed + i
    parallel_sampler.populate_task(env, policy)                       # :50-62
    t_init = time.time() - t_init
    out = []
    for max_samples in args.max_samples:
        t0 = time.time()
        paths = parallel_sampler.sample_paths(theta, max_samples, max_path_length=args.T)   # :98-126
        dt = time.time() - t0
        n = int(sum(len(p["rewards"]) for p in paths))
        out.append(dict(steps=n, seconds=dt, steps_per_s=n / dt, n_paths=len(paths), max_samples=max_samples))
    rec = dict(runs=out, n_parallel=args.n_parallel, init_seconds=t_init, ref_root=ref_root(),
               modules={m: os.path.relpath(sys.modules[m].__file__, ref_root())
                        for m in ["rllab.sampler.parallel_sampler", "rllab.sampler.stateful_pool",
                                  "rllab.sampler.utils", "rllab.envs.normalized_env"]})
    if args.dump_paths:
        import pickle
        pickle.dump(paths, open(args.dump_paths, "wb"))
    parallel_sampler.terminate_task()
    if stateful_pool.singleton_pool.pool is not None:
        stateful_pool.singleton_pool.pool.terminate()
    print(MARK + json.dumps(rec))
    sys.stdout.flush()


# ----------------------------------------------------------------------------------------------
# parent side (bench.py, tests)
# ----------------------------------------------------------------------------------------------
def run(kind, theta, T, max_samples, n_parallel, hidden=(32, 32), seed=1, dump_paths=None, timeout=600):
    """Run the reference sampler in a child process; ``max_samples`` may be a list (several
    ``sample_paths`` calls on the same pool -- the first one warms the workers up)."""
    import numpy as np
    if not isinstance(max_samples, (list, tuple)):
        max_samples = [max_samples]
    with tempfile.TemporaryDirectory() as tmp:
        tf = os.path.join(tmp, "theta.npy")
        np.save(tf, np.asarray(theta, dtype=np.float64))
        cmd = [sys.executable, os.path.abspath(__file__), "--child", "--kind", str(kind), "--theta", tf, "--T", str(T),
               "--n-parallel", str(n_parallel), "--hidden", ",".join(str(h) for h in hidden), "--seed", str(seed),
               "--max-samples"] + [str(int(m)) for m in max_samples]
        if dump_paths:
            cmd += ["--dump-paths", dump_paths]
        env = dict(os.environ)
        # one BLAS / OpenMP thread per worker process: the workers ARE the parallelism
        env.update(OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
        env.pop("PYTHONPATH", None)
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=timeout,
                           cwd=tmp, universal_newlines=True)
    for line in p.stdout.splitlines()[::-1]:
        if line.startswith(MARK):
            return json.loads(line[len(MARK):])
    raise RuntimeError("reference sampler child failed (rc %d):\n%s\n%s" % (p.returncode, p.stdout[-2000:],
                                                                           p.stderr[-4000:]))


def physical_cores():
    """Distinct (physical id, core id) pairs of /proc/cpuinfo: what ``os.cpu_count()`` (hardware threads) is not."""
    try:
        seen, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        if phys is not None and core is not None:
            seen.add((phys, core))
        return len(seen) or None
    except OSError:
        return None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def timed_reference(kind, theta, T, budget_s=15.0, hidden=(32, 32), n_parallel=None, dump_paths=None):
    """steps/s of the reference sampler at n_parallel = 1 and n_parallel = all host cores, on a
    sample sized to ~budget_s seconds of wall time for the parallel run."""
    n_parallel = n_parallel or os.cpu_count()
    one = run(kind, theta, T, [T * 2, T * 4], 1, hidden)["runs"][-1]
    # calibrate the pool with a short call (>= 2 paths per worker), then spend the budget
    cal = run(kind, theta, T, [T * n_parallel, T * n_parallel * 2], n_parallel, hidden)
    rate = cal["runs"][-1]["steps_per_s"]
    target = max(T * n_parallel * 2, int(rate * budget_s * 0.8))
    full = run(kind, theta, T, [T * n_parallel, target], n_parallel, hidden, dump_paths=dump_paths)
    r = full["runs"][-1]
    return dict(steps=r["steps"], seconds=r["seconds"], steps_per_s=r["steps_per_s"], cores=n_parallel,
                n_paths=r["n_paths"], steps_per_s_1core=one["steps_per_s"], one_core_steps=one["steps"],
                cpu_model=cpu_model(), ref_root=os.path.relpath(full["ref_root"], ROOT)
                if full["ref_root"].startswith(ROOT) else full["ref_root"], modules=full["modules"],
                pool_init_seconds=full["init_seconds"])


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--kind", type=int, default=2)
    ap.add_argument("--theta")
    ap.add_argument("--T", type=int, default=500)
    ap.add_argument("--n-parallel", type=int, default=1)
    ap.add_argument("--hidden", default="32,32")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-samples", type=int, nargs="+", default=[1000])
    ap.add_argument("--dump-paths")
    a = ap.parse_args()
    if a.child:
        child_main(a)
    else:
        import numpy as np
        sys.path.insert(0, ROOT)
        from oracle import host_env as H
        from oracle import np_reference as R
        q = H.query(a.kind)
        net = R.NumpyGaussianMLP(q["obs_dim"], q["act_dim"], tuple(int(x) for x in a.hidden.split(",")))
        theta = np.random.RandomState(1).randn(net.n_params) * 0.1
        print(json.dumps(timed_reference(a.kind, theta, a.T, budget_s=5.0,
                                         hidden=tuple(int(x) for x in a.hidden.split(","))), indent=1))
