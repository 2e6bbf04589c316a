"""The reference's OWN test files of the hot path, executed verbatim against the engine.

``oracle/make_ref.py`` stages them byte for byte from /root/reference/tests (sha256 in the manifest); here they are
copied to a scratch directory -- away from the staged copy of the reference's ``rllab`` package, so that ``import
rllab`` resolves to this repo's alias package, i.e. to the engine -- and run under pytest in a child process.  The only
thing added is a stand-in for the ``nose2`` helpers two of them import (tests/ref_shims).

  CPU   tests/test_sampler.py          truncate_paths 130 -> 100 + 30               (reference file :4-31)
        tests/test_stateful_pool.py    run_collect to threshold / over capacity      (:8-19)
  GPU   tests/test_baselines.py        VPG, one iteration, with Zero / LinearFeature / GaussianMLP baselines on
                                       CartpoleEnv and a one-hidden-layer policy       (:14-26)
        tests/regression_tests/test_issue_3.py   TRPO + GaussianMLPPolicy(adaptive_std=True)   (:12-29)
"""
import hashlib
import json
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGED = os.path.join(ROOT, "oracle", "_ref")
SHIMS = os.path.join(ROOT, "tests", "ref_shims")


def _run(files, tmp_path, timeout=900):
    man_path = os.path.join(STAGED, "MANIFEST.json")
    if not os.path.exists(man_path):
        pytest.skip("oracle/_ref not staged (oracle/make_ref.py runs in the build container)")
    man = json.load(open(man_path))["files"]
    targets = []
    for rel in files:
        src = os.path.join(STAGED, rel)
        if rel not in man or not os.path.exists(src):
            pytest.skip("%s not staged" % rel)
        assert hashlib.sha256(open(src, "rb").read()).hexdigest() == man[rel], rel      # the reference's bytes
        dst = os.path.join(str(tmp_path), os.path.basename(rel))
        shutil.copyfile(src, dst)
        targets.append(dst)
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, SHIMS]))
    p = subprocess.run([sys.executable, "-m", "pytest", "-q", "-p", "no:cacheprovider", "--rootdir", str(tmp_path)]
                       + targets, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, cwd=str(tmp_path),
                       universal_newlines=True, timeout=timeout)
    return p.returncode, p.stdout


def test_reference_sampler_and_pool_tests_pass_on_the_engine(tmp_path):
    rc, out = _run(["tests/test_sampler.py", "tests/test_stateful_pool.py"], tmp_path)
    assert rc == 0 and "3 passed" in out, out[-3000:]


@pytest.mark.gpu
def test_reference_baseline_and_adaptive_std_tests_pass_on_the_engine(tmp_path):
    rc, out = _run(["tests/test_baselines.py", "tests/regression_tests/test_issue_3.py"], tmp_path)
    assert rc == 0 and "2 passed" in out, out[-4000:]
