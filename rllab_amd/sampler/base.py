"""Sampler interface and sample post-processing.

``Sampler`` / ``BaseSampler`` keep the surface of rllab/sampler/base.py:10-46.
``process_samples`` computes what the non-recurrent branch of the reference
(:48-104, :163-182) computes -- baseline prediction, TD residuals, GAE advantages,
discounted returns, explained variance, advantage centring/shifting, entropy,
baseline fit and the tabular log -- but on dense device planes:

  * the per-path Python loop + two ``scipy.lfilter`` calls become one launch of the
    HIP scan kernel ``rl_gae`` over the whole [T, N] batch (f64 accumulate);
  * statistics are float64 device reductions (two-pass variance), optionally
    all-reduced across env shards;
  * a list of numpy path dicts (generic Python envs) is first packed into padded
    planes, so there is a single -- device -- code path.
"""
import numpy as np
import torch

import rllab_amd.misc.logger as logger
from rllab_amd import _lib
from rllab_amd.misc import special
from rllab_amd.sampler import dist as D
from rllab_amd.sampler.trajectories import PathList, Trajectories


class Sampler(object):
    def start_worker(self):
        raise NotImplementedError

    def obtain_samples(self, itr):
        raise NotImplementedError

    def process_samples(self, itr, paths):
        raise NotImplementedError

    def shutdown_worker(self):
        raise NotImplementedError


class SamplesData(dict):
    """``samples_data`` dict.  The dense batch is under ``"_traj"``; the
    reference's flat keys (observations, actions, rewards, returns, advantages,
    agent_infos, env_infos: ``[B_valid, ...]`` device tensors, sample order
    t-major) are materialised on first access."""
    _LAZY = ("observations", "actions", "rewards", "returns", "advantages", "agent_infos", "env_infos")

    def __missing__(self, key):
        if key not in self._LAZY:
            raise KeyError(key)
        tr = dict.__getitem__(self, "_traj")
        sel = None if bool(tr.valid.all()) else tr.valid.reshape(-1)

        def rows(planes):  # [D, T, N] -> [B, D]
            x = planes.reshape(planes.shape[0], -1).t()
            return x if sel is None else x[sel]

        def vec(plane):
            x = plane.reshape(-1)
            return x if sel is None else x[sel]
        if key == "observations":
            val = rows(tr.obs)
        elif key == "actions":
            val = rows(tr.actions)
        elif key == "rewards":
            val = vec(tr.rewards)
        elif key == "returns":
            val = vec(tr.returns)
        elif key == "advantages":
            val = vec(tr.advantages)
        elif key == "agent_infos":
            mean = rows(tr.means)
            val = dict(mean=mean, log_std=rows(tr.log_std_planes) if tr.log_std_planes is not None
                       else tr.log_std.unsqueeze(0).expand_as(mean))
        else:
            val = dict()
        self[key] = val
        return val


def _two_pass_var(x, w, cnt):
    """Population variance of x over weights w (0/1), global across ranks.
    Returns (mean, var) as 0-d float64 tensors."""
    (s,) = D.sums((x * w).sum())
    mean = s / cnt
    (ss,) = D.sums((((x - mean) ** 2) * w).sum())
    return mean, ss / cnt


def process_dense(algo, itr, traj, log=True):
    """Post-process one dense rollout in place and return ``SamplesData``."""
    dev = traj.device
    T, N = traj.T, traj.N
    gamma, lam = float(algo.discount), float(algo.gae_lambda)
    valid = traj.valid_mask(algo.whole_paths) if traj.valid is None else traj.valid
    traj.valid = valid
    w = valid.to(torch.float64)
    (cnt,) = D.sums(w.sum())

    baseline = algo.baseline
    if hasattr(baseline, "predict_dense"):
        base = baseline.predict_dense(traj)
    else:  # arbitrary user baseline: per-path predict on host, scattered back
        base = _predict_by_path(baseline, traj)
    adv = torch.empty((T, N), dtype=torch.float32, device=dev)
    ret = torch.empty((T, N), dtype=torch.float32, device=dev)
    base_c = base.contiguous() if base is not None else None
    _lib.check(_lib.lib.rl_gae(T, N, _lib.ptr(traj.rewards), _lib.ptr(base_c), _lib.ptr(traj.dones),
                               gamma, lam, _lib.ptr(adv), _lib.ptr(ret), _lib.stream_ptr()), "rl_gae")
    traj.returns = ret
    traj.baselines = base

    # explained variance of the baseline (special.explained_variance_1d, reference :68-71)
    ret64 = ret.to(torch.float64)
    base64 = base if base is not None else torch.zeros_like(ret64)
    _, var_y = _two_pass_var(ret64, w, cnt)
    _, var_res = _two_pass_var(ret64 - base64, w, cnt)
    _, var_pred = _two_pass_var(base64, w, cnt)
    var_y_f = float(var_y)
    if np.isclose(var_y_f, 0):
        ev = 0 if float(var_pred) > 0 else 1
    else:
        ev = 1 - float(var_res) / (var_y_f + 1e-8)

    # advantage centring / shifting (algos/util.py:7-12), statistics over valid samples
    adv64 = adv.to(torch.float64)
    if algo.center_adv:
        mean, var = _two_pass_var(adv64, w, cnt)
        adv64 = (adv64 - mean) / (torch.sqrt(var) + 1e-8)
    if algo.positive_adv:
        big = torch.full_like(adv64, float("inf"))
        mn = D.all_reduce_min_(torch.where(valid, adv64, big).min())
        adv64 = (adv64 - mn) + 1e-8
    traj.advantages = torch.where(valid, adv64, torch.zeros_like(adv64)).to(torch.float32)

    # per-path statistics
    paths = PathList(traj)
    env, t0, t1 = paths.index()
    r64 = traj.rewards.to(torch.float64)
    csum = torch.cumsum(r64, dim=0)
    before = torch.where(t0 > 0, csum[(t0 - 1).clamp(min=0), env], torch.zeros_like(csum[t0, env]))
    undisc = csum[t1, env] - before
    disc0 = ret64[t0, env]
    n_local = torch.as_tensor(float(env.numel()), dtype=torch.float64, device=dev)
    n_paths, s_disc, s_und = D.sums(n_local, disc0.sum(), undisc.sum())
    mean_und = s_und / n_paths
    (ss_und,) = D.sums(((undisc - mean_und) ** 2).sum())
    inf = torch.as_tensor(float("inf"), dtype=torch.float64, device=dev)
    mx = D.all_reduce_max_(undisc.max() if undisc.numel() else -inf)
    mn = D.all_reduce_min_(undisc.min() if undisc.numel() else inf)

    # Entropy = mean over samples of the policy entropy (reference :93)
    pdist = algo.policy.distribution
    if traj.log_std_planes is not None and hasattr(pdist, "entropy_sym"):
        e = pdist.entropy_sym(dict(log_std=traj.log_std_planes.to(torch.float64)), axis=0)
        (es,) = D.sums((e * w).sum())
        ent = float(es / cnt)
    elif traj.log_std is not None and hasattr(pdist, "entropy_sym"):
        ent = float(pdist.entropy_sym(dict(log_std=traj.log_std.to(torch.float64)), axis=0))
    else:
        ent = float("nan")

    samples_data = SamplesData(_traj=traj, paths=paths)

    if log:
        logger.log("fitting baseline...")
    if hasattr(baseline, "fit_dense"):
        baseline.fit_dense(traj, all_reduce=D.all_reduce_sum_ if D.is_distributed() else None)
    elif hasattr(baseline, 'fit_with_samples'):
        baseline.fit_with_samples(paths, samples_data)
    else:
        baseline.fit(paths)
    if log:
        logger.log("fitted")
        logger.record_tabular('Iteration', itr)
        logger.record_tabular('AverageDiscountedReturn', float(s_disc / n_paths))
        logger.record_tabular('AverageReturn', float(mean_und))
        logger.record_tabular('ExplainedVariance', ev)
        logger.record_tabular('NumTrajs', int(round(float(n_paths))))
        logger.record_tabular('Entropy', ent)
        logger.record_tabular('Perplexity', float(np.exp(ent)))
        logger.record_tabular('StdReturn', float(torch.sqrt(ss_und / n_paths)))
        logger.record_tabular('MaxReturn', float(mx))
        logger.record_tabular('MinReturn', float(mn))
    return samples_data


def _predict_by_path(baseline, traj):
    paths = PathList(traj, only_valid=False)
    env, t0, t1 = paths._host_index()
    out = np.zeros((traj.T, traj.N), dtype=np.float64)
    for i in range(len(paths)):
        out[t0[i]:t1[i] + 1, env[i]] = baseline.predict(paths[i])
    return torch.as_tensor(out, device=traj.device)


def pack_paths(paths, device=None):
    """List of numpy path dicts -> padded ``Trajectories`` (one column per path)."""
    if device is None:
        if not torch.cuda.is_available():
            raise RuntimeError("rllab_amd: no HIP device -- process_samples has no CPU fallback")
        device = torch.device("cuda", torch.cuda.current_device())
    n = len(paths)
    lens = np.array([len(p["rewards"]) for p in paths])
    T = int(lens.max())
    do = np.asarray(paths[0]["observations"]).reshape(lens[0], -1).shape[1]
    da = np.asarray(paths[0]["actions"]).reshape(lens[0], -1).shape[1]
    obs = np.zeros((do, T, n), np.float32)
    act = np.zeros((da, T, n), np.float32)
    mean = np.zeros((da, T, n), np.float32)
    rew = np.zeros((T, n), np.float32)
    done = np.zeros((T, n), np.uint8)
    valid = np.zeros((T, n), bool)
    log_std = None
    ls_planes = np.zeros((da, T, n), np.float32)
    ls_const = True
    for i, p in enumerate(paths):
        L = lens[i]
        obs[:, :L, i] = np.asarray(p["observations"]).reshape(L, -1).T
        act[:, :L, i] = np.asarray(p["actions"]).reshape(L, -1).T
        rew[:L, i] = p["rewards"]
        ai = p.get("agent_infos", {})
        if "mean" in ai:
            mean[:, :L, i] = np.asarray(ai["mean"]).reshape(L, -1).T
        if "log_std" in ai:
            ls = np.asarray(ai["log_std"]).reshape(L, -1)
            ls_planes[:, :L, i] = ls.T
            if log_std is None:
                log_std = ls[0]
            ls_const = ls_const and bool(np.all(ls == log_std[None, :]))
        done[L - 1:, i] = 1
        valid[:L, i] = True
    t = lambda x: torch.as_tensor(x, device=device)
    traj = Trajectories(t(obs), t(act), t(mean),
                        t(log_std.astype(np.float32)) if log_std is not None else None,
                        t(rew), t(done), T,
                        log_std_planes=None if (ls_const or log_std is None) else t(ls_planes))
    traj.valid = t(valid)
    return traj


class BaseSampler(Sampler):
    def __init__(self, algo):
        self.algo = algo

    def process_samples(self, itr, paths):
        if self.algo.policy.recurrent:
            raise NotImplementedError("recurrent policies are outside the hot path built here")
        traj = paths.traj if isinstance(paths, PathList) else pack_paths(paths)
        return process_dense(self.algo, itr, traj)
