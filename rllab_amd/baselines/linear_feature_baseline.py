"""LinearFeatureBaseline (mirrors rllab/baselines/linear_feature_baseline.py:6-43).

Ridge regression of returns on phi = [clip(o,-10,10), clip(o)^2, t/100,
(t/100)^2, (t/100)^3, 1] with t the step index inside the path.  The per-path
``fit`` / ``predict`` keep the reference's numpy semantics for API users; the
sampler uses the dense forms: features are built as float64 device planes, the
normal equations Phi^T Phi and Phi^T y are accumulated on the device in float64
(all-reduced across ranks when sharded -- every rank then solves the same
(2*Do+4)^2 system with the reference's lstsq + regularisation-retry rule).
"""
import os

import numpy as np
import torch

from rllab_amd.baselines.base import Baseline


class LinearFeatureBaseline(Baseline):
    def __init__(self, env_spec, reg_coeff=1e-5):
        self._coeffs_value = None
        self._pending = None       # (HostRead of the packed normal equations, F) of a dense fit not solved yet
        self._reg_coeff = reg_coeff

    # The dense fit only LAUNCHES the normal-equation kernels and starts reading their (F + 1) x F result back; the
    # small host solve runs when the coefficients are first asked for -- in the training loop that is the next
    # iteration's prediction, i.e. while the next rollout occupies the device -- so neither the read nor the
    # lstsq sits between process_samples and the policy update.
    @property
    def _coeffs(self):
        if self._pending is not None:
            read, F = self._pending
            self._pending = None
            host = read.get()
            self._coeffs_value = self._solve(host[:F * F].reshape(F, F), host[F * F:])
        return self._coeffs_value

    @_coeffs.setter
    def _coeffs(self, value):
        self._pending = None
        self._coeffs_value = value

    def __getstate__(self):
        d = dict(self.__dict__)
        d["_coeffs_value"] = self._coeffs      # solve what is pending; handles do not pickle
        d["_pending"] = None
        return d

    def __setstate__(self, d):
        d = dict(d)
        if "_coeffs" in d:                     # snapshots written before the fit became asynchronous
            d["_coeffs_value"] = d.pop("_coeffs")
        d.setdefault("_pending", None)
        self.__dict__.update(d)

    def get_param_values(self, **tags):
        return self._coeffs

    def set_param_values(self, val, **tags):
        self._coeffs = val

    # -- per-path numpy API (reference semantics) -----------------------------------
    def _features(self, path):
        o = np.clip(path["observations"], -10, 10)
        l = len(path["rewards"])
        al = np.arange(l).reshape(-1, 1) / 100.0
        return np.concatenate([o, o ** 2, al, al ** 2, al ** 3, np.ones((l, 1))], axis=1)

    def _solve(self, gram, rhs):
        reg_coeff = self._reg_coeff
        coeffs = None
        for _ in range(5):
            coeffs = np.linalg.lstsq(gram + reg_coeff * np.identity(gram.shape[0]), rhs, rcond=None)[0]
            if not np.any(np.isnan(coeffs)):
                break
            reg_coeff *= 10
        return coeffs

    def fit(self, paths):
        if hasattr(paths, "traj"):  # lazy PathList of a dense batch
            return self.fit_dense(paths.traj)
        featmat = np.concatenate([self._features(path) for path in paths])
        returns = np.concatenate([path["returns"] for path in paths])
        self._coeffs = self._solve(featmat.T.dot(featmat), featmat.T.dot(returns))

    def predict(self, path):
        if self._coeffs is None:
            return np.zeros(len(path["rewards"]))
        return self._features(path).dot(self._coeffs)

    # -- dense device forms ------------------------------------------------------------
    def dense_coeffs(self):
        """Coefficients for the fused prediction inside rl_path_scan (None before the first fit:
        the reference predicts zeros, linear_feature_baseline.py:39-40)."""
        return self._coeffs

    @staticmethod
    def _features_dense(traj):
        """[F, T*N] float64 feature planes, F = 2*Do + 4 (torch form, for obs_dim > 20)."""
        o = traj.obs.reshape(traj.obs_dim, -1).to(torch.float64).clamp(-10, 10)
        al = (traj.time_in_path().reshape(1, -1).to(torch.float64)) / 100.0
        return torch.cat([o, o ** 2, al, al ** 2, al ** 3, torch.ones_like(al)], dim=0)

    def predict_dense(self, traj):
        """[T, N] float64 baseline plane, or None before the first fit (== zeros)."""
        if self._coeffs is None:
            return None
        if traj.obs_dim <= 21 and traj.device.type == "cuda":
            from rllab_amd.sampler.base import path_scan
            return path_scan(traj, True, self._coeffs)[2]
        w = torch.as_tensor(self._coeffs, dtype=torch.float64, device=traj.device)
        return (w @ self._features_dense(traj)).reshape(traj.T, traj.N)

    def fit_dense(self, traj, all_reduce=None):
        """Normal equations of the batch (one launch), their sum over the ranks, the solve started asynchronously."""
        packed = self.normal_eq_dense(traj)
        if all_reduce is not None:
            all_reduce(packed)
        self.fit_from_packed(packed, 2 * traj.obs_dim + 4)

    def fit_from_packed(self, packed, F):
        """``packed`` = [Phi^T W Phi | Phi^T W y] summed over all ranks (float64 device vector): start reading it; the
        solve happens where the coefficients are first needed."""
        from rllab_amd.misc.device_io import read_async
        self._coeffs_value = None
        self._pending = (read_async(packed), F)

    def normal_eq_dense(self, traj):
        """This rank's [Phi^T W Phi | Phi^T W y] as a float64 device vector of (F + 1) F entries (linear_feature_baseline.py:
        25-36 builds the same sums on the host)."""
        F = 2 * traj.obs_dim + 4
        if traj.obs_dim <= 21 and traj.device.type == "cuda":
            # Phi^T W Phi and Phi^T W y by rl_lfb_normal_eq: one pass, features rebuilt in LDS
            from rllab_amd import _lib
            from rllab_amd.sampler.base import _workspace, path_scan
            tin = getattr(traj, "tin", None)
            if tin is None:
                tin = path_scan(traj, True, None, want_values=False)[0]
            valid = traj.valid if traj.valid is not None else torch.ones((traj.T, traj.N), dtype=torch.bool,
                                                                         device=traj.device)
            valid_u8 = valid.contiguous().view(torch.uint8) if valid.dtype == torch.bool else valid.to(torch.uint8).contiguous()
            ws = _workspace(traj.device, traj.obs_dim)
            packed = torch.empty((F + 1) * F, dtype=torch.float64, device=traj.device)
            _lib.check(_lib.lib.rl_lfb_normal_eq(traj.B, traj.obs_dim, _lib.ptr(traj.obs), _lib.ptr(tin),
                                                 _lib.ptr(traj.returns), _lib.ptr(valid_u8), _lib.ptr(ws),
                                                 ws.numel(), _lib.ptr(packed),
                                                 1 if os.environ.get("RLLAB_LFB_VALU") is not None else 0,
                                                 _lib.stream_ptr()),
                       "rl_lfb_normal_eq")
        else:
            phi = self._features_dense(traj)
            w = traj.valid.reshape(1, -1).to(torch.float64) if traj.valid is not None else None
            y = traj.returns.reshape(-1).to(torch.float64)
            phi_w = phi * w if w is not None else phi
            packed = torch.cat([(phi_w @ phi.t()).reshape(-1), phi_w @ y])
        return packed
