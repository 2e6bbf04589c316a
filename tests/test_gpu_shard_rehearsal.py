"""Full-size dress rehearsal of the 8-GPU configurations on ONE device (BASELINE configs C4 and C5).

The reference has one process and no sharding (rllab/sampler/parallel_sampler.py:98-126; the only per-worker rule
is seed + i, :77-81); the contract here is SURVEY 8(e): rank r owns the env index range [r n, (r + 1) n), rolls out
and evaluates on its shard, and ONE sum all-reduce per quantity makes the sharded run equal the single-process one.
On a one-GPU box that is checked at the real sizes without any collective:

  * one process rolls out all N = 8 n envs; then eight launches of n envs with ``env_offset = r n`` and the same
    seed / step counter: every shard's observations, actions, means, rewards and dones are BIT-IDENTICAL to its
    slice of the big batch (the Philox counters are keyed by the global env index);
  * each shard's ``rl_sample_stats`` row, ``rl_lfb_normal_eq`` normal equations, ``rl_policy_grad_loss`` gradient and
    loss sums, and one ``rl_policy_fvp`` partial are summed in rank order -- what the all-reduce does -- and equal the
    N-env single-process quantities: float64 statistics / normal equations to 1e-9, f32-kernel sums to 2e-5.
"""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

WORLD = 8


def _env_policy(name, hidden):
    from rllab_amd.envs.mujoco.half_cheetah_env import HalfCheetahEnv
    from rllab_amd.envs.mujoco.swimmer_env import SwimmerEnv
    from rllab_amd.envs.normalized_env import normalize
    from rllab_amd.misc import ext
    from rllab_amd.policies.gaussian_mlp_policy import GaussianMLPPolicy
    ext.set_seed(11)
    env = normalize(SwimmerEnv() if name == "swimmer" else HalfCheetahEnv())
    return env, GaussianMLPPolicy(env_spec=env.spec, hidden_sizes=hidden)


def _rollout(env, policy, n, offset, T, counter):
    v = env.vec_env_executor(n_envs=n, max_path_length=T, seed=23, env_offset=offset)
    v.step_counter = counter
    return v.rollout(policy, T, reset_at_start=True)


def _scan_and_stats(traj, coeffs, gamma, lam):
    """path index / validity / baseline prediction, GAE, the 20 statistics and the normal equations of one batch
    (what sampler/base.py::process_dense launches before anything crosses ranks)."""
    from rllab_amd import _lib
    from rllab_amd.sampler.base import _workspace, path_scan
    dev, T, N = traj.device, traj.T, traj.N
    tin, valid, base = path_scan(traj, True, coeffs)
    adv, ret, und = (torch.empty((T, N), dtype=torch.float32, device=dev) for _ in range(3))
    _lib.check(_lib.lib.rl_gae(T, N, _lib.ptr(traj.rewards), _lib.ptr(base), _lib.ptr(traj.dones), gamma, lam,
                               _lib.ptr(adv), _lib.ptr(ret), _lib.ptr(und), _lib.stream_ptr()), "rl_gae")
    valid_u8 = valid.contiguous().view(torch.uint8)
    ws = _workspace(dev, traj.obs_dim)
    st = torch.empty(20, dtype=torch.float64, device=dev)
    _lib.check(_lib.lib.rl_sample_stats(T * N, _lib.ptr(ret), _lib.ptr(base), _lib.ptr(adv), _lib.ptr(und),
                                        _lib.ptr(tin), _lib.ptr(valid_u8), 0.0, 0.0, None, N, _lib.ptr(ws),
                                        ws.numel(), _lib.ptr(st), _lib.stream_ptr()), "rl_sample_stats")
    F = 2 * traj.obs_dim + 4
    neq = torch.empty((F + 1) * F, dtype=torch.float64, device=dev)
    _lib.check(_lib.lib.rl_lfb_normal_eq(T * N, traj.obs_dim, _lib.ptr(traj.obs), _lib.ptr(tin), _lib.ptr(ret),
                                         _lib.ptr(valid_u8), _lib.ptr(ws), ws.numel(), _lib.ptr(neq), 0,
                                         _lib.stream_ptr()), "rl_lfb_normal_eq")
    return dict(valid=valid, valid_u8=valid_u8, adv=adv, ret=ret, st=st.cpu().numpy(), neq=neq.cpu().numpy())


def _update_partials(policy, traj, q, mean_a, denom, inv_count, vec):
    """Centred advantages with the GLOBAL moments, then the shard's terms of gradient, loss sums and one FVP."""
    from rllab_amd import _lib
    dev, T, N, B = traj.device, traj.T, traj.N, traj.T * traj.N
    adv = torch.empty((T, N), dtype=torch.float32, device=dev)
    _lib.check(_lib.lib.rl_adv_finish(B, _lib.ptr(q["adv"]), _lib.ptr(q["valid_u8"]), float(mean_a), float(denom),
                                      0.0, _lib.ptr(adv), _lib.stream_ptr()), "rl_adv_finish")
    inputs = (traj.obs.reshape(traj.obs_dim, B), traj.actions.reshape(traj.act_dim, B), adv.reshape(B),
              traj.means.reshape(traj.act_dim, B), traj.log_std.reshape(-1, 1), q["valid"].reshape(B).float(),
              torch.tensor(inv_count, dtype=torch.float64))
    ops = policy.fused_ops()
    g = ops.loss_grad(inputs, with_loss=True).cpu().numpy()
    sums = ops._loss_cache["out"].cpu().numpy().copy()          # this process's four sums (no rank in sight)
    hv = ops.fvp(inputs, vec).cpu().numpy()
    ops.release()
    return g, sums, hv


@pytest.mark.timeout(900)
@pytest.mark.parametrize("name,hidden,n,lam", [("swimmer", (32, 32), 4096, 1.0),        # C4: 8 x 4096, TRPO
                                               ("half_cheetah", (64, 64), 1024, 0.97)])   # C5: 8 x 1024, TRPO + GAE
def test_union_of_eight_shards_is_the_single_process_batch(name, hidden, n, lam, monkeypatch):
    from rllab_amd.sampler.base import _ADV, _ADV2, _COUNT, fold_stats
    # Bit-identity of a shard and its slice holds within ONE launch shape of the rollout (the shapes' policy forward
    # passes sum in different orders).  1024 HalfCheetah envs per rank run one env per wavefront (csrc/env_kernels.hip:
    # rollout_two_leg_wave_kernel, n <= 2048) -- on every rank of an 8-GPU run alike; the single-process reference batch of
    # 8192 envs is rolled out in the same shape here.
    monkeypatch.setenv("RLLAB_TWO_LEG_WAVE_KERNEL", "1")
    T, gamma, N = 500, 0.99, WORLD * n
    env, policy = _env_policy(name, hidden)
    counter = 3 * (T + 1)
    rng = np.random.RandomState(5)
    do = env.observation_space.flat_dim
    coeffs = rng.randn(2 * do + 4) * 0.1                       # a fitted baseline of some earlier iteration
    vec = torch.as_tensor(rng.randn(policy.flat_params.numel()), device="cuda")

    full = _rollout(env, policy, N, 0, T, counter)
    fq = _scan_and_stats(full, coeffs, gamma, lam)
    cnt = fq["st"][_COUNT]
    assert cnt > 0.9 * N * T
    mean_a = fq["st"][_ADV] / cnt
    denom = np.sqrt(max(fq["st"][_ADV2] / cnt - mean_a ** 2, 0.0)) + 1e-8
    g_full, sums_full, hv_full = _update_partials(policy, full, fq, mean_a, denom, 1.0 / cnt, vec)

    rows, neq, g, sums, hv = [], 0.0, 0.0, np.zeros(4), 0.0
    kl_max = -np.inf
    for r in range(WORLD):
        shard = _rollout(env, policy, n, r * n, T, counter)
        sl = slice(r * n, (r + 1) * n)
        # the shard IS its slice of the single-process batch, bit for bit
        for nm in ("obs", "actions", "means", "rewards", "dones"):
            assert torch.equal(getattr(shard, nm), getattr(full, nm)[..., sl]), (name, r, nm)
        sq = _scan_and_stats(shard, coeffs, gamma, lam)
        assert torch.equal(sq["valid"], fq["valid"][:, sl])
        assert torch.equal(sq["ret"], fq["ret"][:, sl]) and torch.equal(sq["adv"], fq["adv"][:, sl])
        rows.append(sq["st"])
        neq = neq + sq["neq"]                                   # rank order, like the all-reduce
        g_r, s_r, hv_r = _update_partials(policy, shard, sq, mean_a, denom, 1.0 / cnt, vec)
        g, hv = g + g_r, hv + hv_r
        sums[:3] += s_r[:3]
        kl_max = max(kl_max, s_r[3])
        del shard
    folded = fold_stats(np.stack(rows))
    fin = np.isfinite(fq["st"])                                  # no progress plane here: those extrema stay -/+ inf
    assert np.array_equal(folded[~fin], fq["st"][~fin])
    scale = np.maximum(np.abs(fq["st"][fin]), 1.0)
    assert np.all(np.abs(folded[fin] - fq["st"][fin]) <= 1e-9 * scale), (folded, fq["st"])
    assert folded[_COUNT] == cnt
    assert np.abs(neq - fq["neq"]).max() <= 1e-9 * np.abs(fq["neq"]).max()
    assert np.abs(g - g_full).max() <= 2e-5 * np.abs(g_full).max(), np.abs(g - g_full).max()
    assert np.abs(hv - hv_full).max() <= 2e-5 * np.abs(hv_full).max(), np.abs(hv - hv_full).max()
    assert np.all(np.abs(sums[:3] - sums_full[:3]) <= 2e-5 * np.maximum(np.abs(sums_full[:3]), 1.0)), (sums, sums_full)
    assert kl_max == sums_full[3]                               # an extremum folds exactly
