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
from rllab_amd.misc.device_io import read_async, upload_async
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


_WS = {}


def _workspace(device, obs_dim):
    key = (device, obs_dim)
    if key not in _WS:
        n = _lib.lib.rl_process_workspace_bytes(int(obs_dim))
        _WS[key] = torch.empty(n, dtype=torch.uint8, device=device)
    return _WS[key]


# column indices of rl_sample_stats (include/rllab_amd.h)
(_COUNT, _RET, _RET2, _BASE, _BASE2, _RES, _RES2, _ADV, _ADV2, _NPATH, _UND, _UND2, _DISC, _PROG, _PROG2,
 _ADVMIN, _UNDMAX, _UNDMIN, _PROGMAX, _PROGMIN) = range(20)
_N_SUM = 15


def path_scan(traj, whole_paths, coeffs=None, want_values=True):
    """rl_path_scan: (tin int32 [T,N], valid bool [T,N], values f64 [T,N] or None)."""
    dev = traj.device
    T, N = traj.T, traj.N
    tin = torch.empty((T, N), dtype=torch.int32, device=dev)
    valid = torch.empty((T, N), dtype=torch.uint8, device=dev)
    values = torch.empty((T, N), dtype=torch.float64, device=dev) if (want_values and coeffs is not None) else None
    cf = None
    if coeffs is not None:
        if torch.is_tensor(coeffs) and coeffs.device == dev:
            cf = coeffs.to(torch.float64).contiguous()
        else:   # through pinned memory: a pageable upload would hold the host until the rollout ahead of it is done
            cf = upload_async(np.asarray(coeffs, dtype=np.float64), torch.float64, dev)
        assert cf.numel() == 2 * traj.obs_dim + 4
    _lib.check(_lib.lib.rl_path_scan(T, N, traj.obs_dim, _lib.ptr(traj.dones), _lib.ptr(traj.obs), _lib.ptr(cf),
                                     int(bool(whole_paths)), _lib.ptr(tin), _lib.ptr(valid), _lib.ptr(values),
                                     _lib.stream_ptr()), "rl_path_scan")
    return tin, valid.view(torch.bool), values     # 0 / 1 bytes: the same storage seen as bool


def merge_stats(st):
    """Combine the rl_sample_stats rows of all env shards on the device (kept for callers that want a tensor):
    one all-gather, then sum / min / max over the rank axis."""
    if not D.is_distributed():
        return st
    return torch.as_tensor(fold_stats(D.all_gather_rows(st).cpu().numpy()), device=st.device)


def fold_stats(rows):
    """[world, 20] host rows -> one row: additive columns summed in rank order, extrema folded."""
    rows = np.asarray(rows, dtype=np.float64).reshape(-1, 20)
    out = rows[:, :].sum(axis=0)
    for c in (_ADVMIN, _UNDMIN, _PROGMIN):
        out[c] = rows[:, c].min()
    for c in (_UNDMAX, _PROGMAX):
        out[c] = rows[:, c].max()
    return out


_SHIFT = dict(ret=0.0, und=0.0)   # last iteration's means: they only condition the one-pass variances


def process_dense(algo, itr, traj, log=True):
    """Post-process one dense rollout in place and return ``SamplesData``.

    Launches: rl_path_scan (path index + validity + baseline prediction), rl_gae (advantages,
    returns, undiscounted returns), rl_sample_stats (every moment below in one read),
    rl_adv_finish, then the baseline fit (rl_lfb_normal_eq for LinearFeatureBaseline)."""
    dev = traj.device
    T, N = traj.T, traj.N
    B = T * N
    gamma, lam = float(algo.discount), float(algo.gae_lambda)
    baseline = algo.baseline
    dense_lfb = hasattr(baseline, "dense_coeffs") and traj.obs_dim <= 21
    preset_valid = traj.valid
    tin, valid, base = path_scan(traj, algo.whole_paths, baseline.dense_coeffs() if dense_lfb else None)
    if preset_valid is not None:          # batches packed from path lists carry their own padding mask
        valid = preset_valid
    traj.valid, traj.tin = valid, tin
    if not dense_lfb:
        if hasattr(baseline, "predict_dense"):
            base = baseline.predict_dense(traj)
        else:  # arbitrary user baseline: per-path predict on host, scattered back
            base = _predict_by_path(baseline, traj)
    adv = torch.empty((T, N), dtype=torch.float32, device=dev)
    ret = torch.empty((T, N), dtype=torch.float32, device=dev)
    und = torch.empty((T, N), dtype=torch.float32, device=dev)
    base_c = base.contiguous() if base is not None else None
    _lib.check(_lib.lib.rl_gae(T, N, _lib.ptr(traj.rewards), _lib.ptr(base_c), _lib.ptr(traj.dones),
                               gamma, lam, _lib.ptr(adv), _lib.ptr(ret), _lib.ptr(und), _lib.stream_ptr()), "rl_gae")
    traj.returns = ret
    traj.baselines = base

    valid_u8 = valid.contiguous().view(torch.uint8)
    ws = _workspace(dev, traj.obs_dim)
    st = torch.empty(20, dtype=torch.float64, device=dev)
    # envs that log forward progress name the observation component to difference over each path
    inner = getattr(algo, "env", None)
    while hasattr(inner, "wrapped_env"):
        inner = inner.wrapped_env
    prog_idx = getattr(inner, "progress_obs_index", None)
    prog = traj.obs[prog_idx % traj.obs_dim] if prog_idx is not None else None
    _lib.check(_lib.lib.rl_sample_stats(B, _lib.ptr(ret), _lib.ptr(base_c), _lib.ptr(adv), _lib.ptr(und),
                                        _lib.ptr(tin), _lib.ptr(valid_u8), _SHIFT["ret"], _SHIFT["und"],
                                        _lib.ptr(prog), N, _lib.ptr(ws), ws.numel(), _lib.ptr(st),
                                        _lib.stream_ptr()), "rl_sample_stats")
    # the iteration's one blocking host read: batch statistics and, riding along, the recorded log_std row
    # (Entropy, AveragePolicyStd)
    dense_fit = hasattr(baseline, "fit_dense")
    if dense_fit and log:
        logger.log("fitting baseline...")
    if dense_fit and D.is_distributed() and hasattr(baseline, "normal_eq_dense"):
        # sharded: the statistics row and the baseline's normal equations cross the ranks in ONE all-gather (every rank
        # then adds the rows in rank order: identical sums everywhere) instead of a gather and an all-reduce
        packed = baseline.normal_eq_dense(traj)
        rows = D.all_gather_rows(torch.cat([st, packed]))
        stats_read = read_async(rows[:, :st.numel()].contiguous())
        baseline.fit_from_packed(rows[:, st.numel():].sum(dim=0), 2 * traj.obs_dim + 4)
        dense_fit_done = True
    else:
        stats_read = read_async(D.all_gather_rows(st))      # [world, 20]: one collective, folded on the host
        dense_fit_done = False
    ls_read = read_async(traj.log_std) if (traj.log_std is not None and traj.log_std_planes is None) else None
    # LinearFeatureBaseline's normal equations need nothing the host is about to compute (returns, path index,
    # validity are on the device already): queue them behind the statistics so the device works through the wait
    if dense_fit and not dense_fit_done:
        baseline.fit_dense(traj, all_reduce=D.all_reduce_sum_ if D.is_distributed() else None)
    s = fold_stats(stats_read.get())
    traj.log_std_host = ls_read.get().astype(np.float64) if ls_read is not None else None
    cnt, n_paths = s[_COUNT], s[_NPATH]
    traj.count = float(cnt)                   # global number of valid samples (npo_inputs: 1 / W)

    def moments(i_sum, i_sq, n):
        m = s[i_sum] / n
        return m, max(s[i_sq] / n - m * m, 0.0)
    m_ret, var_y = moments(_RET, _RET2, cnt)
    _, var_pred = moments(_BASE, _BASE2, cnt)
    _, var_res = moments(_RES, _RES2, cnt)
    # explained variance of the baseline (special.explained_variance_1d, reference :68-71)
    if np.isclose(var_y, 0):
        ev = 0 if var_pred > 0 else 1
    else:
        ev = 1 - var_res / (var_y + 1e-8)

    # advantage centring / shifting (algos/util.py:7-12), statistics over valid samples
    mean_a, denom, shift = 0.0, 1.0, 0.0
    if algo.center_adv:
        mean_a, var_a = moments(_ADV, _ADV2, cnt)
        denom = float(np.sqrt(var_a)) + 1e-8
    if algo.positive_adv:
        shift = -((s[_ADVMIN] - mean_a) / denom) + 1e-8
    adv_out = torch.empty((T, N), dtype=torch.float32, device=dev)
    _lib.check(_lib.lib.rl_adv_finish(B, _lib.ptr(adv), _lib.ptr(valid_u8), float(mean_a), float(denom),
                                      float(shift), _lib.ptr(adv_out), _lib.stream_ptr()), "rl_adv_finish")
    traj.advantages = adv_out
    # everything the policy update reads is on the device now: let it start (algos/npo.py::prefetch_update) BEFORE the
    # host turns to the remaining statistics, a host-side baseline fit and the env / policy diagnostics -- the device
    # idled ~35 us here while the host computed log lines (profiles/r05_timeline.csv)
    paths = PathList(traj)
    samples_data = SamplesData(_traj=traj, paths=paths)
    if hasattr(algo, "prefetch_update") and traj.obs.is_cuda and getattr(algo, "_update_follows", False):
        algo.prefetch_update(samples_data)

    if prog is not None and n_paths > 0:
        m_prog, var_prog = moments(_PROG, _PROG2, n_paths)
        traj.progress_stats = (float(m_prog), float(s[_PROGMAX]), float(s[_PROGMIN]), float(np.sqrt(var_prog)))
    m_und_s, var_und = moments(_UND, _UND2, n_paths) if n_paths > 0 else (np.nan, np.nan)
    mean_und = m_und_s + _SHIFT["und"]
    _SHIFT["ret"], _SHIFT["und"] = float(m_ret + _SHIFT["ret"]), (float(mean_und) if n_paths > 0 else 0.0)

    # Entropy = mean over samples of the policy entropy (reference :93)
    pdist = algo.policy.distribution
    if traj.log_std_planes is not None and hasattr(pdist, "entropy_sym"):
        e = pdist.entropy_sym(dict(log_std=traj.log_std_planes.to(torch.float64)), axis=0)
        (es,) = D.sums((e * valid.to(torch.float64)).sum())
        ent = float(es) / cnt
    elif traj.log_std_host is not None and hasattr(pdist, "entropy"):
        ent = float(np.asarray(pdist.entropy(dict(log_std=traj.log_std_host[None, :]))).reshape(-1)[0])
    elif traj.log_std is not None and hasattr(pdist, "entropy_sym"):
        ent = float(pdist.entropy_sym(dict(log_std=traj.log_std.to(torch.float64)), axis=0))
    else:
        ent = float("nan")

    if not dense_fit:
        if log:
            logger.log("fitting baseline...")
        if hasattr(baseline, 'fit_with_samples'):
            baseline.fit_with_samples(paths, samples_data)
        else:
            baseline.fit(paths)
    if log:
        logger.log("fitted")
        logger.record_tabular('Iteration', itr)
        logger.record_tabular('AverageDiscountedReturn', float(s[_DISC] / n_paths) if n_paths > 0 else np.nan)
        logger.record_tabular('AverageReturn', float(mean_und))
        logger.record_tabular('ExplainedVariance', ev)
        logger.record_tabular('NumTrajs', int(round(float(n_paths))))
        logger.record_tabular('Entropy', ent)
        logger.record_tabular('Perplexity', float(np.exp(ent)))
        logger.record_tabular('StdReturn', float(np.sqrt(var_und)) if n_paths > 0 else np.nan)
        logger.record_tabular('MaxReturn', float(s[_UNDMAX]))
        logger.record_tabular('MinReturn', float(s[_UNDMIN]))
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
