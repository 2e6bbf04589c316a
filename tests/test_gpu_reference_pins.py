"""GPU tests tied to numbers / files the REFERENCE itself holds (the only ones that exist for
this path):

* ``tests/golden/diagonal_gaussian.npz`` (outputs of the real ``DiagonalGaussian.kl`` /
  ``log_likelihood``, oracle/make_golden.py) fed STRAIGHT through the C-ABI kernel
  ``rl_policy_loss_kl`` -- no product formula module in between;
* the iteration-0 table of ``trpo_cartpole`` the reference documents
  (docs/user/experiments.rst:81-95: AverageReturn 68.3242, MinReturn 19.9874, MeanKL 0.00305741,
  Entropy 1.41894) against a fresh policy on the HIP Cartpole;
* the two acceptance scripts ``examples/trpo_cartpole.py`` / ``examples/trpo_swimmer.py`` executed
  VERBATIM (staged byte for byte under oracle/_ref/examples by oracle/make_ref.py; sha256 checked
  against the manifest) through the ``rllab`` alias package.
"""
import hashlib
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
STAGED = os.path.join(ROOT, "oracle", "_ref")


def test_reference_diagonal_gaussian_vectors_through_rl_policy_loss_kl():
    """Per fixture row i: old = (om_i, ols_i), new = (nm_i, nls_i), action xs_i.  The kernel's new mean comes
    from its MLP, so the row is presented shifted by nm_i (KL and log-likelihood depend on means only through
    differences) to a policy whose output layer is zero and whose log_std parameter is nls_i:
        sum w KL / W            == DiagonalGaussian.kl(old, new)_i                      (reference :71-83)
        sum w logp(a) adv / W   == DiagonalGaussian.log_likelihood(xs, new)_i, adv = 1  (reference :54-60)
        sum w lr adv / W        == exp(loglik_new - loglik_old)                         (reference :62-69)
    """
    from rllab_amd import _lib
    from rllab_amd.envs.env_spec import EnvSpec
    from rllab_amd.policies.gaussian_mlp_policy import GaussianMLPPolicy
    from rllab_amd.spaces import Box
    d = np.load(os.path.join(GOLDEN, "diagonal_gaussian.npz"))
    n, da = d["om"].shape
    do, h = 20, 32
    np.random.seed(0)
    spec = EnvSpec(Box(-np.ones(do), np.ones(do)), Box(-np.ones(da), np.ones(da)))
    pol = GaussianMLPPolicy(spec, hidden_sizes=(h, h), min_std=None)
    ops = pol.fused_ops()
    assert ops is not None
    theta = pol.get_param_values()
    n_out = h * da + da
    theta[-(n_out + da):-da] = 0.0                       # Wout, bout = 0: the new mean is exactly 0
    dev = pol.flat_params.device
    obs = torch.randn(do, 1, device=dev)
    one = torch.ones(1, device=dev)
    got_kl, got_ll, got_lr = [], [], []
    for i in range(n):
        theta[-da:] = d["nls"][i]
        pol.set_param_values(theta)
        om = torch.as_tensor((d["om"][i] - d["nm"][i]).reshape(da, 1), dtype=torch.float32, device=dev)
        act = torch.as_tensor((d["xs"][i] - d["nm"][i]).reshape(da, 1), dtype=torch.float32, device=dev)
        ols = torch.as_tensor(d["ols"][i].reshape(da, 1), dtype=torch.float32, device=dev)
        s = ops.loss_stats_host((obs, act, one, om, ols, one, torch.tensor(1.0, dtype=torch.float64)))
        got_lr.append(s[0]); got_kl.append(s[1]); got_ll.append(s[2])
    # float32 kernel against float64 reference values: 1e-5 relative (north_star), floor 1e-5 absolute
    np.testing.assert_allclose(got_kl, d["kl"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(got_ll, d["logli"], rtol=2e-5, atol=2e-5)
    z_old = (d["xs"] - d["om"]) / np.exp(d["ols"])
    ll_old = -d["ols"].sum(-1) - 0.5 * (z_old ** 2).sum(-1) - 0.5 * da * np.log(2 * np.pi)
    np.testing.assert_allclose(np.log(got_lr), d["logli"] - ll_old, rtol=0, atol=5e-5)
    assert _lib.lib.rl_abi_version() >= 5


def _iteration0(seed, batch_size, **env_kwargs):
    from rllab.algos.trpo import TRPO
    from rllab.baselines.linear_feature_baseline import LinearFeatureBaseline
    from rllab.envs.box2d.cartpole_env import CartpoleEnv
    from rllab.envs.normalized_env import normalize
    from rllab.misc import ext, logger
    from rllab.policies.gaussian_mlp_policy import GaussianMLPPolicy
    ext.set_seed(seed)
    env = normalize(CartpoleEnv(**env_kwargs))
    policy = GaussianMLPPolicy(env_spec=env.spec, hidden_sizes=(32, 32))
    algo = TRPO(env=env, policy=policy, baseline=LinearFeatureBaseline(env_spec=env.spec), batch_size=batch_size,
                max_path_length=100, n_itr=1, discount=0.99, step_size=0.01, sampler_args=dict(seed=seed))
    algo.start_worker()
    algo.init_opt()
    paths = algo.sampler.obtain_samples(0)
    sd = algo.sampler.process_samples(0, paths)
    algo.log_diagnostics(paths)
    algo.optimize_policy(0, sd)
    tab = {k: float(v) for k, v in logger.get_tabular().items() if k != "Iteration"}
    logger.dump_tabular()
    algo.shutdown_worker()
    return tab


DOC = dict(AverageReturn=68.3242, StdReturn=42.6061, MinReturn=19.9874, MaxReturn=369.864, NumTrajs=1278,
           MeanKL=0.00305741, Entropy=1.41894, Perplexity=4.13273)     # docs/user/experiments.rst:81-95


def test_trpo_cartpole_iteration0_against_the_documented_log(quiet_logger):
    """examples/trpo_cartpole.py settings (horizon 100, fresh GaussianMLPPolicy(32,32), TRPO step 0.01) on the HIP
    Cartpole, 8 seeds, against the reference's documented itr-0 table (docs/user/experiments.rst:81-95, a run of
    10 000 samples: 1278 trajectories x (6.83 rewarded steps + the terminal one)).  What a fresh policy achieves is a
    property of the env dynamics (how fast the pole falls under +-10 N pushes), so this ties the hand-written
    Box2D-order island solver to numbers the reference holds.
    Exact: Entropy, Perplexity, AveragePolicyStd, ExplainedVariance (no baseline yet).  Statistical: AverageReturn
    brackets 68.3, MeanKL has the documented order, LossAfter < LossBefore.
    The table also discriminates the two readings of CartpoleEnv.reset: its MinReturn
    19.9874 is exactly "two rewarded steps, then the terminal one" and no shorter episode occurs among 1278 --
    reproduced with ``reset_pole_follows_cart=True`` (hinge closed at reset); the literal reading of
    cartpole_env.py:28-43 at HEAD (the pole body is not moved; default here) starts the hinge up to 0.12 m open,
    the first position solve swings the pole by up to 0.17 rad and ~9 % of the episodes end within two steps."""
    tabs = [_iteration0(seed, 10000) for seed in range(1, 9)]
    for t in tabs:
        assert abs(t["Entropy"] - DOC["Entropy"]) < 1e-4 and abs(t["Perplexity"] - DOC["Perplexity"]) < 1e-3
        assert abs(t["AveragePolicyStd"] - 1.0) < 1e-6 and abs(t["ExplainedVariance"]) < 1e-9
        assert 5e-4 < t["MeanKL"] <= 0.0101 and t["LossAfter"] < t["LossBefore"]
    avg = np.array([t["AverageReturn"] for t in tabs])
    print("default reset: itr-0 AverageReturn per seed", np.round(avg, 2), "MinReturn", [round(t["MinReturn"], 2) for t in tabs],
          "Std", [round(t["StdReturn"], 1) for t in tabs], "MeanKL", [round(t["MeanKL"], 5) for t in tabs])
    # documented 68.3242 is one draw; the spread over policy initialisations is several units
    assert avg.min() <= DOC["AverageReturn"] + 3.0 and avg.max() >= DOC["AverageReturn"] - 3.0, avg
    assert abs(np.median(avg) - DOC["AverageReturn"]) < 0.2 * DOC["AverageReturn"], avg
    kls = np.array([t["MeanKL"] for t in tabs])
    assert 0.3 * DOC["MeanKL"] < np.median(kls) < 3.4 * DOC["MeanKL"], kls

    closed = [_iteration0(seed, 10000, reset_pole_follows_cart=True) for seed in range(1, 9)]
    avg_c = np.array([t["AverageReturn"] for t in closed])
    mins = np.array([t["MinReturn"] for t in closed])
    stds = np.array([t["StdReturn"] for t in closed])
    print("hinge closed at reset: AverageReturn", np.round(avg_c, 2), "MinReturn", np.round(mins, 4), "Std", np.round(stds, 1))
    assert np.all(np.abs(mins - DOC["MinReturn"]) < 0.02), mins          # 19.98..20.00: the shortest possible episode
    assert avg_c.min() <= DOC["AverageReturn"] + 3.0 and avg_c.max() >= DOC["AverageReturn"] - 3.0, avg_c
    assert abs(np.median(stds) - DOC["StdReturn"]) < 0.25 * DOC["StdReturn"], stds
    # ... while the default reset produces episodes the documented run does not contain
    assert np.median([t["MinReturn"] for t in tabs]) < 10.5


@pytest.mark.parametrize("script", ["trpo_cartpole.py", "trpo_swimmer.py"])
def test_reference_example_script_runs_verbatim(script, tmp_path):
    """`python examples/<script>` of the reference, unchanged, on the MI355X engine: 40 TRPO iterations finish,
    every tabular key of the reference's log is there, MeanKL respects the trust region and AverageReturn rises."""
    path = os.path.join(STAGED, "examples", script)
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/examples not staged (oracle/make_ref.py runs in the build container)")
    man = json.load(open(os.path.join(STAGED, "MANIFEST.json")))["files"]
    assert hashlib.sha256(open(path, "rb").read()).hexdigest() == man["examples/" + script]
    env = dict(os.environ, PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable, path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env,
                       cwd=str(tmp_path), universal_newlines=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    cols = {}
    for line in p.stdout.splitlines():
        m = re.match(r".*\| (\w+)\s+(-?[\d.eE+\-]+|nan|inf)\s*$", line)
        if m:
            cols.setdefault(m.group(1), []).append(float(m.group(2)))
    for key in ["Iteration", "Entropy", "Perplexity", "AverageReturn", "StdReturn", "MaxReturn", "MinReturn",
                "AverageDiscountedReturn", "NumTrajs", "ExplainedVariance", "AveragePolicyStd", "MeanKL",
                "LossBefore", "LossAfter"]:
        assert len(cols.get(key, [])) == 40, (key, len(cols.get(key, [])))
    assert cols["Iteration"] == list(range(40))
    assert max(cols["MeanKL"]) <= 0.0101 and np.all(np.isfinite(cols["AverageReturn"]))
    ret = np.array(cols["AverageReturn"])
    print(script, "AverageReturn first/last 5:", np.round(ret[:5], 2), np.round(ret[-5:], 2))
    if script == "trpo_cartpole.py":
        assert ret[-5:].mean() > 4.0 * ret[:5].mean(), ret        # documented run: 68 -> ~1000 of 1000
    else:
        assert ret[-5:].mean() > ret[:5].mean() + 1.0, ret
