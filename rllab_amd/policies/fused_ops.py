"""HIP-kernel implementations of the update-time functions of a GaussianMLPPolicy
(csrc/policy_kernels.hip through the C ABI): surrogate loss + mean KL, flat
gradient, Fisher-vector product.  Each result is a SUM of per-rank terms already
normalised by the global sample count, all-reduced here (RCCL) so callers see
global values.  Inputs are the tuples built by ``rllab_amd.algos.npo.npo_inputs``.
"""
import ctypes
import math
import os

import torch

from rllab_amd import _lib
from rllab_amd.core.serializable import Serializable
from rllab_amd.misc.device_io import read_async
from rllab_amd.sampler import dist as D


class FusedGaussianMLPOps(object):
    device_line_search = True    # line_search_device / line_search_resolve are available (ConjugateGradientOptimizer asks)

    def __init__(self, policy):
        self.policy = policy
        self.layout = policy.kernel_layout()
        assert self.layout is not None
        # what the kernels are told: the padded hidden widths (policies/kernel_layout.py); P_pad parameters
        self.dims = (policy.obs_dim, policy.action_dim) + self.layout.hidden3
        self.n_kernel = self.layout.P_pad
        # which kernel family the library picks for these dimensions (csrc/policy_kernels.hip::dispatch_net): the
        # cooperative kernels for wide / deep nets AND for equal-width nets on (obs_dim, action_dim) pairs the
        # one-wavefront-per-tile kernels are not instantiated for (they take the dimensions at run time)
        self.wide_kernels = self.layout.wide or (policy.obs_dim, policy.action_dim) not in self.NARROW_PAIRS
        self._ws = None
        self._loss_cache = None
        self._bound = {}     # key -> (PolicyBatch, tensors kept alive, inv_count float); <= 2 entries
        self._acts = None    # hidden-activation cache the gradient pass fills for the FVP passes
        self._absmax = None  # ... and max |obs| of the batch (rl_policy_batch.obs_absmax)
        self._acts_tag = None

    # parameter updates written by our own kernels straight into the parameter vector (see _eval_point): kept on the
    # policy, because the rollout's padded parameter copy must notice them too
    @property
    def _epoch(self):
        return getattr(self.policy, "_raw_writes", 0)

    @_epoch.setter
    def _epoch(self, value):
        self.policy._raw_writes = value

    # (obs_dim, action_dim) pairs the equal-width two-layer kernels are instantiated for (the HIP-native envs');
    # the cooperative kernels of the wide / deep nets take the widths at run time
    NARROW_PAIRS = ((4, 1), (6, 1), (11, 1), (13, 2), (20, 3), (20, 6), (21, 6))

    @staticmethod
    def supported(policy):
        """Two or three tanh hidden layers of at most 128 units each (zero-padded to the kernels' tiles),
        state-independent std (learned or, ``learn_std=False``, frozen: see ``masks_frozen``), parameters on the device,
        obs_dim <= 30, action_dim <= 8 (policy.kernel_layout())."""
        layout = policy.kernel_layout() if hasattr(policy, "kernel_layout") else None
        return layout is not None

    def accepts(self, inputs):
        """The kernels take ONE old log_std row (state-independent std)."""
        return inputs[4].numel() == self.dims[1] and inputs[0].is_cuda

    def _workspace(self, device):
        if self._ws is None or self._ws.device != device:
            n = _lib.lib.rl_policy_workspace_bytes(*self.dims)
            self._ws = torch.empty(n, dtype=torch.uint8, device=device)
        return self._ws

    def _batch(self, inputs):
        """The C-ABI batch descriptor of an input tuple.  Built once per tuple (one update uses
        the same tuple for ~30 passes): contiguity fix-ups, pointer extraction and the one
        host read of 1/W happen here, not per pass."""
        self.layout.theta()                 # bring the kernels' copy of the parameters up to date (no-op if current)
        key = tuple(id(t) for t in inputs)
        hit = self._bound.get(key)
        if hit is not None:
            return hit
        obs, act, adv, old_mean, old_log_std, w, inv_count = inputs
        theta = self.layout.theta()          # kernel layout; a persistent buffer, refreshed in place (_refresh)
        keep = [t.contiguous() for t in (obs, act, adv, old_mean, old_log_std.reshape(-1).float(), w, theta)]
        obs, act, adv, old_mean, old_ls, w, theta = keep
        assert theta.data_ptr() == self.layout.theta().data_ptr()       # updates are in place
        pol = self.policy
        inv = float(inv_count)
        b = _lib.PolicyBatch(
            n_samples=obs.shape[-1], obs_dim=self.dims[0], act_dim=self.dims[1], hidden0=self.dims[2],
            hidden1=self.dims[3], hidden2=self.dims[4], inv_count=inv,
            log_min_std=math.log(pol.min_std) if pol.min_std is not None else -1e30,
            theta=theta.data_ptr(), obs=obs.data_ptr(), actions=act.data_ptr(), advantages=adv.data_ptr(),
            old_means=old_mean.data_ptr(), old_log_std=old_ls.data_ptr(), weights=w.data_ptr(),
            layer_activations=self.layout.layer_activations,
            opts=_lib.launch_opts())      # (one process-wide struct, refreshed from the RLLAB_* switches before the calls
                                          # whose kernel choice it steers: _fvp_into, fvp_variant)
        if len(self._bound) >= 2:   # full batch + (optionally) its FVP subsample
            self._bound.clear()
        self._bound[key] = (b, keep + list(inputs), inv)
        return self._bound[key]

    def release(self):
        """Drop the cached descriptor (and the batch tensors it keeps alive)."""
        self._bound.clear()
        self._loss_cache = None
        self._acts = None
        self._acts_tag = None

    def _loss_record(self, tag, out, inv, rows=None):
        """Cache entry of one loss / KL evaluation whose four per-rank sums are in ``out`` (device); starts the
        host read.  Sharded: ONE all-gather of the four numbers (``rows``: already gathered, with the gradient), folded
        on the host (sum of three, max of one)."""
        if rows is None:
            rows = D.all_gather_rows(out)        # [world, 4], kept: the device-side line search compares against them
        c = dict(tag=tag, out=out, rows=rows, inv=inv, dev=None, read=read_async(rows), host=None)
        self._loss_cache = c
        return c

    def _loss_eval(self, inputs):
        """Launch the loss / KL pass at the current parameters (unless this batch was already evaluated at
        them) and start reading the four sums back; nothing here waits for the device."""
        # NPO / VPG ask for loss and KL before and after the step through separate calls
        # (npo.py:100-111): same batch, same parameters -> same pass.
        tag = self._eval_point(inputs)
        c = self._loss_cache
        if c is not None and c["tag"] == tag:
            return c
        b, keep, inv = self._batch(inputs)
        ws = self._workspace(keep[0].device)
        out = torch.empty(4, dtype=torch.float64, device=keep[0].device)
        _lib.check(_lib.lib.rl_policy_loss_kl(ctypes.byref(b), _lib.ptr(ws), ws.numel(), _lib.ptr(out),
                                              _lib.stream_ptr()), "rl_policy_loss_kl")
        return self._loss_record(tag, out, inv)

    @staticmethod
    def _resolve(c):
        """Host values of a ``_loss_eval`` record: (sum w lr adv, sum w KL, sum w logp adv) / W, max KL."""
        if c["host"] is None:
            h = c["read"].get().reshape(-1, 4)            # one row per rank
            sums = h[:, :3].sum(axis=0) * c["inv"]
            c["host"] = (float(sums[0]), float(sums[1]), float(sums[2]), float(h[:, 3].max()))
        return c["host"]

    def loss_stats(self, inputs):
        """[sum w lr adv, sum w KL, sum w logp adv] * inv_count (global) and max KL, as a
        float64 device tensor of 4."""
        c = self._loss_eval(inputs)
        if c["dev"] is None:
            c["dev"] = torch.as_tensor(self._resolve(c), dtype=torch.float64, device=c["out"].device)
        return c["dev"]

    def loss_stats_host(self, inputs):
        """The same four numbers as Python floats: one device read per evaluation point."""
        return self._resolve(self._loss_eval(inputs))

    def loss_and_kl(self, inputs):
        s = self.loss_stats_host(inputs)
        return -s[0], s[1]

    def loss_stats_deferred(self, inputs):
        """Like ``loss_stats_host`` but read later: returns ``f`` with ``f() -> (sum lr adv, KL, logp adv) / W, max KL``
        of the evaluation at the parameters as they are NOW, valid after they move on."""
        c = self._loss_eval(inputs)
        return lambda: self._resolve(c)

    def loss_and_kl_deferred(self, inputs):
        """Launch the pass now, read later: returns ``f`` with ``f() -> (loss, mean KL)``.  The record stays
        valid after the parameters move on (the optimizer needs loss_before only at its first comparison)."""
        c = self._loss_eval(inputs)

        def get():
            s = self._resolve(c)
            return -s[0], s[1]
        get.record = c          # (the sums still on the device: line_search_device compares candidates with them there)
        return get

    def _eval_point(self, inputs):
        """(batch, parameter version): flat_params._version counts torch's in-place updates, _epoch the
        ones written by rl_line_search_point straight into the parameter vector."""
        return (tuple(id(t) for t in inputs), self.policy.flat_params._version, self._epoch)

    def loss_grad(self, inputs, vpg=False, keep_activations=False, with_loss=False):
        """Flat gradient of the surrogate loss.  ``keep_activations``: also leave the hidden activations of
        the batch in device memory for the Fisher-vector products that follow at the same parameters (TRPO:
        one gradient, then cg_iters + 1 products), which then skip the forward pass.  ``with_loss``: the same
        pass also produces the loss / KL sums at these parameters (rl_policy_grad_loss), so the loss evaluation
        that belongs to this point costs no pass of its own."""
        b, keep, inv = self._batch(inputs)
        ws = self._workspace(keep[0].device)
        out = torch.empty(self.n_kernel, dtype=torch.float64, device=keep[0].device)
        self._acts_tag = None
        b.activations = None
        need = _lib.lib.rl_policy_activation_bytes(b.n_samples, self.dims[2], self.dims[3], self.dims[4]) \
            if keep_activations and not vpg else 0
        if need:
            if self._acts is None or self._acts.numel() < need or self._acts.device != keep[0].device:
                self._acts = torch.empty(need, dtype=torch.uint8, device=keep[0].device)
            b.activations = self._acts.data_ptr()
            # max |obs| of the batch, left by the same pass: the bound the two-way f16 split products scale by
            if self._absmax is None or self._absmax.device != keep[0].device:
                self._absmax = torch.zeros(1, dtype=torch.float32, device=keep[0].device)
            b.obs_absmax = self._absmax.data_ptr()
        tag = self._eval_point(inputs)
        have_loss = self._loss_cache is not None and self._loss_cache["tag"] == tag
        try:
            if with_loss and not have_loss:
                # gradient and loss sums in one buffer: sharded (host-issued collectives), both cross the ranks in ONE
                # all-gather, the gradient rows are then added in rank order on every rank (identical sums everywhere)
                both = torch.empty(self.n_kernel + 4, dtype=torch.float64, device=keep[0].device)
                out, out4 = both[:self.n_kernel], both[self.n_kernel:]
                _lib.check(_lib.lib.rl_policy_grad_loss(ctypes.byref(b), int(vpg), _lib.ptr(ws), ws.numel(),
                                                        _lib.ptr(out), _lib.ptr(out4), _lib.stream_ptr()),
                           "rl_policy_grad_loss")
                if D.is_distributed() and D.peer_reducer() is None:
                    rows = D.all_gather_rows(both)
                    self._loss_record(tag, out4, inv, rows=rows[:, self.n_kernel:].contiguous())
                    if b.activations:
                        self._acts_tag = tag
                    return self._mask_frozen(self.layout.unpack(rows[:, :self.n_kernel].sum(dim=0)))
                self._loss_record(tag, out4, inv)
            else:
                _lib.check(_lib.lib.rl_policy_grad(ctypes.byref(b), int(vpg), _lib.ptr(ws), ws.numel(),
                                                   _lib.ptr(out), _lib.stream_ptr()), "rl_policy_grad")
            if b.activations:
                self._acts_tag = tag
        finally:
            b.activations = None
            b.obs_absmax = None
        return self._mask_frozen(self.layout.unpack(D.update_sum_(out)))

    # ``GaussianMLPPolicy(learn_std=False)``: the log_std row is a parameter that is not trainable.  The passes compute
    # its gradient like any other; it is zeroed here.  That is all a frozen row needs: at theta_old the Fisher matrix is
    # block diagonal between the network and the log_std row (d2 KL / d mu d sigma = 0), so conjugate gradient started
    # from a gradient with zeros there keeps zeros there exactly, the step leaves the row alone, and the optimizers
    # may keep working on the full vector (device CG, line search kernels) -- ``masks_frozen`` tells them so.
    @property
    def masks_frozen(self):
        return self._frozen_index() is not None

    def _frozen_index(self):
        if not hasattr(self, "_frozen_idx"):
            pol = self.policy
            tr = pol._flat_index(trainable=True)
            if tr is None:
                self._frozen_idx = None
            else:
                keep = torch.ones(pol.flat_params.numel(), dtype=torch.bool, device=pol.flat_params.device)
                keep[tr] = False
                self._frozen_idx = torch.nonzero(keep).reshape(-1)
        return self._frozen_idx

    def _mask_frozen(self, g):
        idx = self._frozen_index()
        if idx is not None:
            g.index_fill_(0, idx, 0.0)
        return g

    def value_and_grad(self, inputs, penalty=0.0):
        """float64 (value, flat gradient over the trainable parameters) of  surrogate loss + penalty * mean KL  in ONE
        pass (rl_policy_grad_loss with rl_policy_batch.kl_penalty): PenaltyLbfgsOptimizer's objective when PPO / NPO
        run it on a GaussianMLPPolicy (rllab/algos/ppo.py:8-22, penalty_lbfgs_optimizer.py:66-79)."""
        b, keep, inv = self._batch(inputs)
        dev = keep[0].device
        ws = self._workspace(dev)
        grad = torch.empty(self.n_kernel, dtype=torch.float64, device=dev)
        out4 = torch.empty(4, dtype=torch.float64, device=dev)
        self._acts_tag = None
        b.activations = None
        b.kl_penalty = float(penalty)
        try:
            _lib.check(_lib.lib.rl_policy_grad_loss(ctypes.byref(b), 0, _lib.ptr(ws), ws.numel(), _lib.ptr(grad),
                                                    _lib.ptr(out4), _lib.stream_ptr()), "rl_policy_grad_loss")
        finally:
            b.kl_penalty = 0.0
        packed = torch.cat([out4[:3], self.layout.unpack(grad)])
        D.all_reduce_sum_(packed)
        idx = self.policy._flat_index(trainable=True)
        host = packed.cpu().numpy()
        g = host[3:] if idx is None else host[3:][idx.cpu().numpy()]
        return float((-host[0] + penalty * host[1]) * inv), g.copy()

    def _fvp_into(self, b, ws, vec32, out, inputs=None):
        cached = inputs is not None and self._acts_tag is not None and self._acts_tag == self._eval_point(inputs)
        b.activations = self._acts.data_ptr() if cached else None
        b.obs_absmax = self._absmax.data_ptr() if cached and self._absmax is not None else None
        b.opts = _lib.launch_opts()
        try:
            _lib.check(_lib.lib.rl_policy_fvp(ctypes.byref(b), _lib.ptr(vec32), _lib.ptr(ws), ws.numel(),
                                              _lib.ptr(out), _lib.stream_ptr()), "rl_policy_fvp")
        finally:
            b.activations = None
            b.obs_absmax = None
        out = D.update_sum_(out)
        if getattr(self.layout, "identity_layer", False):
            # the identity second layer of a one-hidden-layer policy is a CONSTANT of the kernel copy: without this its
            # (non-zero) Fisher rows would pull conjugate gradient out of the real parameters' subspace
            out.mul_(self.layout.real_mask)
        return out

    def fvp_variant(self, inputs):
        """Which arithmetic ``rl_policy_fvp`` would run the next product of this batch in (rl_policy_fvp_variant): 0 = f32
        matrix instructions, 1 = bf16 matrix instructions on three-way split f32 operands (csrc/policy_split_kernels.hip;
        cached activations, two 32-unit layers, whole 32-sample tiles).  A host query: launches nothing."""
        b, _, _ = self._batch(inputs)
        cached = self._acts_tag is not None and self._acts_tag == self._eval_point(inputs)
        b.activations = self._acts.data_ptr() if cached else None
        b.obs_absmax = self._absmax.data_ptr() if cached and self._absmax is not None else None
        b.opts = _lib.launch_opts()
        try:
            return int(_lib.lib.rl_policy_fvp_variant(ctypes.byref(b)))
        finally:
            b.activations = None
            b.obs_absmax = None

    def fvp(self, inputs, vec):
        b, keep, _ = self._batch(inputs)
        ws = self._workspace(keep[0].device)
        v = self.layout.pack(vec.to(torch.float32)).contiguous()
        out = torch.empty(self.n_kernel, dtype=torch.float64, device=keep[0].device)
        return self.layout.unpack(self._fvp_into(b, ws, v, out, inputs))

    def _cg_loop(self, b, ws, inputs, cg_iters, reg_coeff, residual_tol, x, r, p, p32, z, scal, st):
        """cg_iters x (Fisher-vector product of the direction p32, one krylov.cg iteration): product (+ row
        reduction), all-reduce when sharded, rl_cg_step -- three launches per iteration.
        ``fuse_cg`` (or RLLAB_FUSE_CG=1) selects rl_policy_fvp_cg_step instead, where the row reduction's last
        workgroup runs the CG algebra (two launches, bit-identical results).  MEASURED SLOWER on MI355X (update
        3.43 -> 3.52 ms at the bench size, profiles/r02_notes.md): the hand-over inside one launch needs a device-scope
        release / acquire, which on an 8-XCD part writes back and invalidates the per-XCD L2s -- more than the
        launch boundary it saves.  Kept as a tested alternative, off by default."""
        n = self.n_kernel
        if D.is_distributed() or self.wide_kernels or getattr(self.layout, "identity_layer", False) or \
                not getattr(self, "fuse_cg", bool(os.environ.get("RLLAB_FUSE_CG"))):
            for _ in range(cg_iters):
                self._fvp_into(b, ws, p32, z, inputs)
                _lib.check(_lib.lib.rl_cg_step(n, _lib.ptr(z), float(reg_coeff), float(residual_tol), _lib.ptr(x),
                                               _lib.ptr(r), _lib.ptr(p), _lib.ptr(p32), _lib.ptr(scal), st),
                           "rl_cg_step")
            return
        if getattr(self, "_ticket", None) is None or self._ticket.device != x.device:
            self._ticket = torch.zeros(1, dtype=torch.int32, device=x.device)
        cached = self._acts_tag is not None and self._acts_tag == self._eval_point(inputs)
        b.activations = self._acts.data_ptr() if cached else None
        try:
            for _ in range(cg_iters):
                _lib.check(_lib.lib.rl_policy_fvp_cg_step(
                    ctypes.byref(b), _lib.ptr(ws), ws.numel(), float(reg_coeff), float(residual_tol), _lib.ptr(x),
                    _lib.ptr(r), _lib.ptr(p), _lib.ptr(p32), _lib.ptr(scal), _lib.ptr(z), _lib.ptr(self._ticket), st),
                    "rl_policy_fvp_cg_step")
        finally:
            b.activations = None

    def cg(self, inputs, g, cg_iters, reg_coeff, residual_tol=1e-10):
        """krylov.cg (rllab/misc/krylov.py:7-39) on Hx = F x + reg_coeff x with the vector algebra of
        each iteration in ONE launch (rl_cg_step) between the Fisher-vector-product passes: two
        kernels + one all-reduce per iteration, no host synchronisation.
        Returns (x, x^T H x) as float64 device tensors.  Runs in the kernels' (zero-padded) parameter space:
        padded entries of g are 0 and stay 0 in every iterate."""
        b, keep, _ = self._batch(inputs)
        dev = keep[0].device
        ws = self._workspace(dev)
        n = self.n_kernel
        f64 = dict(dtype=torch.float64, device=dev)
        g = self.layout.pack(g.to(torch.float64)).contiguous()
        x, r, p, z = (torch.empty(n, **f64) for _ in range(4))
        p32 = torch.empty(n, dtype=torch.float32, device=dev)
        scal = torch.empty(4, **f64)
        st = _lib.stream_ptr()
        _lib.check(_lib.lib.rl_cg_init(n, _lib.ptr(g), _lib.ptr(x), _lib.ptr(r), _lib.ptr(p), _lib.ptr(p32),
                                       _lib.ptr(scal), st), "rl_cg_init")
        self._cg_loop(b, ws, inputs, cg_iters, reg_coeff, residual_tol, x, r, p, p32, z, scal, st)
        # F x for the initial step size (conjugate_gradient_optimizer.py:258-260)
        x32 = x.to(torch.float32)
        self._fvp_into(b, ws, x32, z, inputs)
        xHx = x.dot(z + float(reg_coeff) * x)
        return self.layout.unpack(x), xHx

    def cg_step_vector(self, inputs, g, cg_iters, reg_coeff, max_constraint, residual_tol=1e-10,
                       reuse_cg_residual=True):
        """CG as in ``cg`` followed by rl_trpo_step: returns (step, stats) with step = beta x as a float64
        device vector, stats = {x^T H x, beta} on the device -- no torch arithmetic in between.
        ``reuse_cg_residual``: take H x from CG's invariant r = g - H x instead of one more Fisher-vector
        product (the reference evaluates Hx(x) afresh, conjugate_gradient_optimizer.py:258-260; both are the
        same quantity to the rounding of an f32 product, tests/test_gpu_update_parity.py)."""
        b, keep, _ = self._batch(inputs)
        dev = keep[0].device
        ws = self._workspace(dev)
        n = self.n_kernel
        f64 = dict(dtype=torch.float64, device=dev)
        g = self.layout.pack(g.to(torch.float64)).contiguous()
        x, r, p, z, step = (torch.empty(n, **f64) for _ in range(5))
        p32 = torch.empty(n, dtype=torch.float32, device=dev)
        scal = torch.empty(4, **f64)
        stats = torch.empty(2, **f64)
        st = _lib.stream_ptr()
        _lib.check(_lib.lib.rl_cg_init(n, _lib.ptr(g), _lib.ptr(x), _lib.ptr(r), _lib.ptr(p), _lib.ptr(p32),
                                       _lib.ptr(scal), st), "rl_cg_init")
        self._cg_loop(b, ws, inputs, cg_iters, reg_coeff, residual_tol, x, r, p, p32, z, scal, st)
        if reuse_cg_residual:
            _lib.check(_lib.lib.rl_trpo_step(n, _lib.ptr(x), _lib.ptr(g), _lib.ptr(r), 0.0, float(max_constraint),
                                             _lib.ptr(step), _lib.ptr(stats), st), "rl_trpo_step")
        else:
            x32 = x.to(torch.float32)
            self._fvp_into(b, ws, x32, z, inputs)
            _lib.check(_lib.lib.rl_trpo_step(n, _lib.ptr(x), _lib.ptr(z), None, float(reg_coeff),
                                             float(max_constraint), _lib.ptr(step), _lib.ptr(stats), st),
                       "rl_trpo_step")
        return self.layout.unpack(step), stats

    def line_search_point(self, prev32, step, ratio):
        """theta <- (float)(prev - ratio * step), written in place into the policy's parameter vector."""
        theta = self.policy.flat_params
        assert prev32.dtype == torch.float32 and step.dtype == torch.float64 and theta.is_contiguous()
        _lib.check(_lib.lib.rl_line_search_point(theta.numel(), _lib.ptr(prev32), _lib.ptr(step), float(ratio),
                                                 _lib.ptr(theta.detach()), _lib.stream_ptr()),
                   "rl_line_search_point")
        self._epoch += 1

    # -- the backtracking line search decided on the device (conjugate_gradient_optimizer.py:262-274) -----------------
    def line_search_device(self, inputs, prev32, step, ratios, max_constraint, before_record):
        """Enqueue the first ``len(ratios)`` candidates of the line search WITHOUT a host read in between:
        candidate k's parameters, its loss / KL pass, and rl_line_search_decide (accept test
        ``loss < loss_before and kl <= max_constraint`` against ``before_record``'s sums; writes candidate k + 1's
        parameters only if nothing has been accepted yet).  Once a candidate is accepted the later loss passes return
        at once (rl_policy_batch.gate) and the parameters stay at the accepted point, so whatever the caller enqueues
        next -- the next rollout -- already runs on the updated policy.  Returns a record for ``line_search_resolve``;
        nothing here waits for the device.  Sharded: the candidate's sums are all-gathered like every loss evaluation,
        each rank folds the same rows in the same order and takes the same decision."""
        b, keep, inv = self._batch(inputs)
        dev = keep[0].device
        ws = self._workspace(dev)
        K = len(ratios)
        theta = self.policy.flat_params.detach()
        assert prev32.dtype == torch.float32 and step.dtype == torch.float64 and theta.is_contiguous() and K >= 1
        buf = getattr(self, "_ls_buf", None)
        if buf is None or buf.numel() != 3 + 4 * K or buf.device != dev:
            buf = self._ls_buf = torch.empty(3 + 4 * K, dtype=torch.float64, device=dev)
        buf.zero_()
        state, gate = buf[:2 + 4 * K], buf[2 + 4 * K:].view(torch.int32)      # {accepted, index, K x 4 sums} | gate word
        outs = torch.empty((K, 4), dtype=torch.float64, device=dev)
        n, st = theta.numel(), _lib.stream_ptr()
        brows = before_record["rows"]
        _lib.check(_lib.lib.rl_line_search_point(n, _lib.ptr(prev32), _lib.ptr(step), float(ratios[0]), _lib.ptr(theta),
                                                 st), "rl_line_search_point")
        self._epoch += 1
        b.gate = gate.data_ptr()
        try:
            for k in range(K):
                self.layout.theta()              # padded layouts: the kernels' copy follows the candidate
                _lib.check(_lib.lib.rl_policy_loss_kl(ctypes.byref(b), _lib.ptr(ws), ws.numel(), _lib.ptr(outs[k]),
                                                      st), "rl_policy_loss_kl")
                rows = D.all_gather_rows(outs[k])
                nxt = float(ratios[k + 1]) if k + 1 < K else 0.0
                _lib.check(_lib.lib.rl_line_search_decide(
                    rows.shape[0], _lib.ptr(rows), _lib.ptr(brows), inv, float(max_constraint), k, _lib.ptr(state),
                    _lib.ptr(gate), n, _lib.ptr(prev32), _lib.ptr(step), nxt, _lib.ptr(theta), st),
                    "rl_line_search_decide")
                self._epoch += 1                 # (the parameters MAY have moved: every cache keyed on them steps on)
        finally:
            b.gate = None
        return dict(read=read_async(state), K=K, inv=inv, state=state, world=brows.shape[0])

    def line_search_resolve(self, rec, inputs):
        """Host side of ``line_search_device``: ONE read.  Returns (accepted candidate or None, [(loss, kl)] of the
        candidates that were evaluated).  The accepted candidate's sums become the loss record of the current
        parameters, so the LossAfter / MeanKL the algorithm logs next cost no pass."""
        h = rec["read"].get()
        K, inv = rec["K"], rec["inv"]
        accepted = int(h[1]) if h[0] != 0.0 else None
        n_eval = K if accepted is None else accepted + 1
        sums = h[2:2 + 4 * K].reshape(K, 4)
        evals = [(-(sums[k, 0] * inv), sums[k, 1] * inv) for k in range(n_eval)]
        last = n_eval - 1                        # the parameters are at this candidate now
        host = (float(sums[last, 0] * inv), float(sums[last, 1] * inv), float(sums[last, 2] * inv), float(sums[last, 3]))
        # the record of the current parameters: the candidate's sums (already folded over the ranks by
        # rl_line_search_decide) as row 0 of a [world, 4] device plane of zeros -- the same shape every other record has,
        # so loss_stats() and a further device line search from this point read it like any other
        rows = torch.zeros((rec["world"], 4), dtype=torch.float64, device=rec["state"].device)
        rows[0].copy_(rec["state"][2 + 4 * last:6 + 4 * last])
        self._loss_cache = dict(tag=self._eval_point(inputs), out=rows[0], rows=rows, inv=inv, dev=None, read=None,
                                host=host)
        return accepted, [(float(l), float(c)) for l, c in evals]

    def hvp_approach(self):
        return FusedFisherHvp(self)


class FusedFisherHvp(Serializable):
    """HVP plug-in (interface of PerlmutterHvp, rllab/optimizers/
    conjugate_gradient_optimizer.py:13-55) backed by the fused Fisher-vector-product
    kernel.  Exact for the mean-KL Hessian at theta_new == theta_old, the only point
    where TRPO evaluates it."""

    def __init__(self, ops=None):
        self.ops = ops
        self.target = None
        self.reg_coeff = None

    def update_opt(self, f, target, inputs, reg_coeff):
        self.target, self.reg_coeff = target, reg_coeff

    def build_eval(self, inputs, trainable_index=None):
        ops, reg = self.ops, self.reg_coeff
        n = self.target.flat_params.numel()

        def eval(x):
            if trainable_index is None:
                full = x
            else:
                full = torch.zeros(n, dtype=x.dtype, device=x.device)
                full[trainable_index] = x
            hx = ops.fvp(inputs, full)
            if trainable_index is not None:
                hx = hx[trainable_index]
            return hx + reg * x
        return eval
