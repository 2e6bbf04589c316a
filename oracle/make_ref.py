#!/usr/bin/env python
"""Stage the reference's OWN hot-path modules, unmodified, under ``oracle/_ref/``
(TEST INFRASTRUCTURE; git-ignored like a built .so, travels to the GPU box with
the gpurun snapshot -- /root/reference itself does not exist there).

    python oracle/make_ref.py        (also run by __graft_entry__.build() when
                                      /root/reference is present)

What is staged is computed, not listed: the reference modules named in ROOTS are
imported from /root/reference behind oracle/ref_shim.py and every module the import
pulled in from that tree is copied byte for byte to the same relative path
(``rllab/sampler/stateful_pool.py`` -> ``oracle/_ref/rllab/sampler/stateful_pool.py``).
Plus the two acceptance scripts ``examples/trpo_{cartpole,swimmer}.py``
(tests/test_gpu_reference_pins.py runs them verbatim through the product's ``rllab``
alias), the reference's own test files of this path (tests/test_reference_tests_verbatim.py
runs them verbatim against the engine) and a MANIFEST.json with the sha256 of every staged file.

Used by
  * oracle/ref_sampler.py -- the ``cpu_baseline`` of bench.py, ``kind: "reference"``:
    the reference's unmodified ``parallel_sampler`` / ``stateful_pool`` / ``rollout`` /
    ``NormalizedEnv`` timed on the GPU box's host cores;
  * tests that cross-check the port (oracle/cpu_sampler.py) against it;
  * oracle/ref_vecenv.py -- the reference's VecEnvExecutor over its NormalizedEnv copies on recorded actions
    (tests/test_gpu_reference_vecenv.py).
Nothing under rllab_amd/ imports it.
"""
import hashlib
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
OUT = os.path.join(HERE, "_ref")

# import roots of the sampling path -- their import
# closure inside the reference tree is what gets staged
ROOTS = [
    "rllab.sampler.parallel_sampler", "rllab.sampler.stateful_pool", "rllab.sampler.utils",
    "rllab.sampler.base", "rllab.envs.base", "rllab.envs.normalized_env", "rllab.envs.proxy_env",
    "rllab.envs.env_spec", "rllab.spaces.box", "rllab.policies.base", "rllab.core.parameterized",
    "rllab.core.serializable", "rllab.misc.tensor_utils", "rllab.misc.special", "rllab.misc.ext",
    "rllab.misc.logger", "rllab.misc.krylov", "rllab.algos.util", "rllab.baselines.linear_feature_baseline",
    "rllab.baselines.zero_baseline", "rllab.distributions.diagonal_gaussian",
    # the vectorised precedent's lock-step executor: run by oracle/ref_vecenv.py over the
    # reference's NormalizedEnv copies as the checker of HipVecEnv.step / the fused rollout's running normalisation
    "sandbox.rocky.tf.envs.vec_env_executor",
    # ... and its sampler loop, the statement of the batch-size contract (``while n_samples < batch_size``): run by
    # oracle/ref_vecsampler.py on a replay of a recorded batch
    "sandbox.rocky.tf.samplers.vectorized_sampler",
]
EXTRA_FILES = ["examples/trpo_cartpole.py", "examples/trpo_swimmer.py",
               # the reference's own tests of this path, run verbatim against the engine
               # (tests/test_reference_tests_verbatim.py)
               "tests/test_sampler.py", "tests/test_stateful_pool.py", "tests/test_baselines.py",
               "tests/regression_tests/test_issue_3.py"]


def stage(verbose=True):
    if not os.path.isdir(REF):
        raise SystemExit("oracle/make_ref.py needs %s (run it in the build container)" % REF)
    sys.path.insert(0, os.path.dirname(HERE))
    from oracle import ref_shim
    ref_shim.install(REF)
    import importlib
    for name in ROOTS:
        importlib.import_module(name)
    files = set(EXTRA_FILES)
    for mod in list(sys.modules.values()):
        f = getattr(mod, "__file__", None)
        if f and os.path.abspath(f).startswith(REF + os.sep):
            files.add(os.path.relpath(os.path.abspath(f), REF))
    # package __init__ files on the way down (some are empty, some import)
    for rel in list(files):
        d = os.path.dirname(rel)
        while d:
            init = os.path.join(d, "__init__.py")
            if os.path.exists(os.path.join(REF, init)):
                files.add(init)
            d = os.path.dirname(d)
    if os.path.isdir(OUT):
        shutil.rmtree(OUT)
    manifest = {}
    for rel in sorted(files):
        dst = os.path.join(OUT, rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(os.path.join(REF, rel), dst)
        manifest[rel] = hashlib.sha256(open(dst, "rb").read()).hexdigest()
    json.dump({"source": REF,
               "note": "unmodified reference files staged by oracle/make_ref.py as the CPU baseline / acceptance "
                       "material of tests and bench.py (VERDICT r1, item 1 and 7); a build product like a .so: "
                       "git-ignored, never imported by rllab_amd/, not part of the repository's sources",
               "files": manifest}, open(os.path.join(OUT, "MANIFEST.json"), "w"), indent=1)
    if verbose:
        sys.stderr.write("[make_ref] staged %d reference files under %s\n" % (len(manifest), OUT))
    return manifest


if __name__ == "__main__":
    stage()
