"""ConjugateGradientOptimizer + Hessian-vector-product plug-ins
(API and numerics of rllab/optimizers/conjugate_gradient_optimizer.py:13-296).

Where the reference hands Theano *expressions* to ``update_opt`` and compiles
``f_loss / f_grad / f_constraint / f_loss_constraint / f_Hx_plain``, this version
takes torch *closures* ``loss(flat_params, *inputs) -> 0-d tensor`` and
differentiates them with autograd.  The control flow of ``optimize`` is the
reference's (:229-296): loss_before -> flat gradient -> CG(Hx, g, cg_iters) ->
initial step sqrt(2*delta / (d^T H d + 1e-8)) -> backtracking line search over
``backtrack_ratio**k`` with parameter update ``prev - ratio*step`` -> accept /
reject predicate, including its ``<=`` / ``>=`` asymmetry.

Differences, all below the numeric contract:
  * every vector stays on the device; CG runs in float64 without host syncs;
  * each evaluated quantity is a SUM of per-rank terms already normalised by the
    global sample count, so one sum all-reduce (RCCL) per evaluation makes the
    sharded run identical to the single-process one.
"""
import os

import numpy as np
import torch

from rllab_amd.core.serializable import Serializable
from rllab_amd.misc import krylov, logger
from rllab_amd.sampler import dist as D


def _flat_for_grad(target):
    return target.flat_params.detach().clone().requires_grad_(True)


class PerlmutterHvp(Serializable):
    """Hx = grad(grad(f) . x) + reg*x by double back-propagation (reference :13-55)."""

    def __init__(self, num_slices=1):
        Serializable.quick_init(self, locals())
        self.target = None
        self.reg_coeff = None
        self._f = None
        self._num_slices = num_slices

    def update_opt(self, f, target, inputs, reg_coeff):
        self.target, self.reg_coeff, self._f = target, reg_coeff, f

    def build_eval(self, inputs, trainable_index=None):
        target, f, reg = self.target, self._f, self.reg_coeff

        def eval(x):
            flat = _flat_for_grad(target)
            g = torch.autograd.grad(f(flat, *inputs), flat, create_graph=True)[0]
            xf = torch.zeros_like(flat)
            if trainable_index is None:
                xf = x.to(flat.dtype)
            else:
                xf[trainable_index] = x.to(flat.dtype)
            hx = torch.autograd.grad((g * xf).sum(), flat)[0]
            if trainable_index is not None:
                hx = hx[trainable_index]
            hx = D.all_reduce_sum_(hx.to(torch.float64))
            return hx + reg * x
        return eval


class FiniteDifferenceHvp(Serializable):
    """Hx ~ (g(theta + eps x) - g(theta - eps x)) / (2 eps), eps = base_eps / ||theta||
    (reference :58-115)."""

    def __init__(self, base_eps=1e-8, symmetric=True, grad_clip=None, num_slices=1):
        Serializable.quick_init(self, locals())
        self.base_eps = base_eps
        self.symmetric = symmetric
        self.grad_clip = grad_clip
        self._num_slices = num_slices

    def update_opt(self, f, target, inputs, reg_coeff):
        self.target, self.reg_coeff, self._f = target, reg_coeff, f

    def build_eval(self, inputs, trainable_index=None):
        target, f, reg = self.target, self._f, self.reg_coeff

        def grad_at(theta):
            flat = theta.detach().clone().requires_grad_(True)
            g = torch.autograd.grad(f(flat, *inputs), flat)[0]
            if trainable_index is not None:
                g = g[trainable_index]
            return D.all_reduce_sum_(g.to(torch.float64))

        def eval(x):
            theta = target.flat_params.detach().to(torch.float64)
            pv = theta if trainable_index is None else theta[trainable_index]
            eps = float(np.float32(self.base_eps / (float(pv.norm()) + 1e-8)))

            def shifted(sign):
                th = theta.clone()
                if trainable_index is None:
                    th = th + sign * eps * x
                else:
                    th[trainable_index] += sign * eps * x
                return th.to(target.flat_params.dtype)
            gp = grad_at(shifted(+1.0))
            if self.symmetric:
                gm = grad_at(shifted(-1.0))
                hx = (gp - gm) / (2 * eps)
            else:
                hx = (gp - grad_at(theta.to(target.flat_params.dtype))) / eps
            return hx + reg * x
        return eval


class ConjugateGradientOptimizer(Serializable):
    # optimize() evaluates loss and constraint at the starting point anyway and keeps them in ``last_before``:
    # NPO logs LossBefore / MeanKLBefore from there instead of paying two extra blocking reads up front
    reports_before_values = True

    def __init__(self, cg_iters=10, reg_coeff=1e-5, subsample_factor=1., backtrack_ratio=0.8,
                 max_backtracks=15, accept_violation=False, hvp_approach=None, num_slices=1,
                 reuse_cg_residual=True):
        """``reuse_cg_residual`` (fused device CG only): d^T H d for the initial step size comes from CG's own
        residual, H d = g - r, instead of the reference's extra Hx(d) evaluation (:258-260) -- the same number
        to the rounding of one f32 Fisher-vector product, one pass over the batch less.  False = evaluate afresh."""
        Serializable.quick_init(self, locals())
        self._reuse_cg_residual = reuse_cg_residual
        self._cg_iters = cg_iters
        self._reg_coeff = reg_coeff
        self._subsample_factor = subsample_factor
        self._backtrack_ratio = backtrack_ratio
        self._max_backtracks = max_backtracks
        self._num_slices = num_slices
        self._loss = None
        self._constraint = None
        self._target = None
        self._max_constraint_val = None
        self._constraint_name = None
        self._accept_violation = accept_violation
        self._hvp_given = hvp_approach is not None
        if hvp_approach is None:
            hvp_approach = PerlmutterHvp(num_slices)
        self._hvp_approach = hvp_approach
        self._fused = None
        # fused device CG only: how many line-search candidates are decided on the device before the host looks
        # (0 = the host decides after every candidate); RLLAB_DEVICE_LINE_SEARCH overrides (A/B timing)
        try:
            self._device_line_search = max(0, int(os.environ.get("RLLAB_DEVICE_LINE_SEARCH", "3")))
        except ValueError:
            self._device_line_search = 3
        self._after_enqueue = None   # called once the whole update is enqueued and before its outcome is read
        self.last_backtrack_iters = None
        self.last_before = None      # (loss, constraint value) at the parameters optimize() started from

    def update_opt(self, loss, target, leq_constraint, inputs=None, extra_inputs=None,
                   constraint_name="constraint", fused=None, *args, **kwargs):
        """``loss`` and ``leq_constraint[0]`` are closures ``f(flat_params, *inputs)``;
        ``leq_constraint[1]`` is the bound.  ``fused`` optionally supplies HIP-kernel
        implementations of the same functions (see policies/fused_ops.py)."""
        constraint_term, constraint_value = leq_constraint
        self._loss = loss
        self._constraint = constraint_term
        self._target = target
        self._max_constraint_val = constraint_value
        self._constraint_name = constraint_name
        self._fused = fused
        if fused is not None and not self._hvp_given:
            self._hvp_approach = fused.hvp_approach()
        self._hvp_approach.update_opt(f=constraint_term, target=target, inputs=inputs,
                                      reg_coeff=self._reg_coeff)

    # -- evaluations ------------------------------------------------------------
    def _trainable_index(self, inputs=None):
        idx = self._target._flat_index(trainable=True)
        if (idx is not None and inputs is not None and not self._hvp_given
                and getattr(self._fused_for(inputs), "masks_frozen", False)):
            # the fused passes keep the gradient of the frozen entries (learn_std=False: the log_std row) at zero and
            # the Fisher matrix does not couple them to the rest: the device CG / line search work on the full vector,
            # they never move.  Only that path is full-length: a caller-given ``hvp_approach`` (FiniteDifferenceHvp,
            # a hand-built PerlmutterHvp) runs krylov.cg + set_param_values(trainable=True) in the trainable
            # subspace like the reference, so it keeps the index.
            return None
        return idx

    def _eval_scalar(self, fn, inputs):
        with torch.no_grad():
            v = fn(self._target.flat_params, *inputs).to(torch.float64)
        return D.all_reduce_sum_(v)

    def _fused_for(self, inputs):
        f = self._fused
        return f if (f is not None and f.accepts(inputs)) else None

    def loss(self, inputs, extra_inputs=None):
        inputs = tuple(inputs) + tuple(extra_inputs or ())
        if self._fused_for(inputs) is not None:
            return float(self._fused.loss_and_kl(inputs)[0])
        return float(self._eval_scalar(self._loss, inputs))

    def constraint_val(self, inputs, extra_inputs=None):
        inputs = tuple(inputs) + tuple(extra_inputs or ())
        if self._fused_for(inputs) is not None:
            return float(self._fused.loss_and_kl(inputs)[1])
        return float(self._eval_scalar(self._constraint, inputs))

    def _loss_constraint(self, inputs):
        if self._fused_for(inputs) is not None:
            return self._fused.loss_and_kl(inputs)
        return (self._eval_scalar(self._loss, inputs), self._eval_scalar(self._constraint, inputs))

    def _flat_grad(self, inputs, keep_activations=False, with_loss=False):
        if self._fused_for(inputs) is not None:
            g = self._fused.loss_grad(inputs, keep_activations=keep_activations, with_loss=with_loss)
        else:
            flat = _flat_for_grad(self._target)
            g = torch.autograd.grad(self._loss(flat, *inputs), flat)[0]
            g = D.all_reduce_sum_(g.to(torch.float64))
        idx = self._trainable_index(inputs)
        return g if idx is None else g[idx]

    # -- the update ---------------------------------------------------------------
    def prefetch(self, inputs, extra_inputs=None):
        """Launch the first pass of ``optimize(inputs)`` -- the flat gradient, which also produces the loss / KL sums
        of the starting point and leaves the activations for the Fisher-vector products -- NOW, asynchronously.
        ``process_samples`` calls this as soon as the advantages are on the device, so that the pass runs while the
        host writes the iteration's sample statistics and diagnostics (the device idled ~0.15 ms there).  The record is
        used by the next ``optimize`` only for the same input tensors at the same parameter version (torch's in-place
        counter + the kernels' own write epoch); anything else recomputes."""
        self._pre = None
        inputs = tuple(inputs) + tuple(extra_inputs or ())
        fused = getattr(self, "_fused", None)
        if fused is None or self._subsample_factor < 1 or self._fused_for(inputs) is None:
            return
        tag = fused._eval_point(inputs)
        flat_g = self._flat_grad(inputs, keep_activations=True, with_loss=True)
        before = fused.loss_and_kl_deferred(inputs)
        self._pre = (tag, flat_g, before, inputs)      # (the tensors themselves: their ids in ``tag`` cannot be recycled)

    def optimize(self, inputs, extra_inputs=None, subsample_grouped_inputs=None):
        inputs = tuple(inputs) + tuple(extra_inputs or ())
        target = self._target
        idx = self._trainable_index(inputs)
        pre, self._pre = getattr(self, "_pre", None), None

        if self._subsample_factor < 1:
            n_samples = inputs[0].shape[-1]
            inds = torch.as_tensor(
                np.random.choice(n_samples, int(n_samples * self._subsample_factor), replace=False),
                device=inputs[0].device)
            subsample_inputs = list(
                x.index_select(-1, inds) if torch.is_tensor(x) and x.dim() > 0 and x.shape[-1] == n_samples
                else x for x in inputs)
            # input convention (algos/npo.py): [..., weights, inv_count]; renormalise the mean
            cnt = D.all_reduce_sum_(subsample_inputs[-2].to(torch.float64).sum())
            subsample_inputs[-1] = (1.0 / cnt).to(inputs[-1].dtype)
            subsample_inputs = tuple(subsample_inputs)
        else:
            subsample_inputs = inputs

        logger.log("computing loss before")
        # launched now, read at the first line-search comparison: the host goes on queueing the gradient and
        # CG launches instead of waiting for this pass
        fused_here = self._fused_for(inputs) is not None
        if not fused_here:
            l_b, c_b = self._loss_constraint(inputs)
            before = lambda: (float(l_b), float(c_b))
        logger.log("performing update")
        logger.log("computing descent direction")
        # the Fisher-vector products below run on the same batch at the same parameters: let the gradient
        # pass leave its hidden activations for them (not when the products use a subsample); the same pass
        # hands back the loss / KL sums of this point (f_loss and f_grad share their forward pass)
        if pre is not None and fused_here and subsample_inputs is inputs and pre[0] == self._fused._eval_point(inputs):
            flat_g, before = pre[1], pre[2]            # launched by prefetch() at this very point
        else:
            flat_g = self._flat_grad(inputs, keep_activations=subsample_inputs is inputs, with_loss=True)
            if fused_here:
                before = self._fused.loss_and_kl_deferred(inputs)
        hvp = self._hvp_approach
        if self._fused is not None and not self._hvp_given and self._fused_for(inputs) is None:
            hvp = PerlmutterHvp(self._num_slices)   # batch the fused kernels cannot take
            hvp.update_opt(f=self._constraint, target=target, inputs=None, reg_coeff=self._reg_coeff)
        from rllab_amd.policies.fused_ops import FusedFisherHvp
        fused_cg = (self._fused_for(subsample_inputs) is not None and isinstance(hvp, FusedFisherHvp)
                    and idx is None)
        full_prev = target.flat_params.detach().clone()
        step_vec = None
        if fused_cg and self._fused_for(inputs) is not None:
            # device-side CG: FVP kernel + one vector-algebra kernel per iteration, then the initial step
            # sqrt(2 delta / (d^T H d + 1e-8)) d by rl_trpo_step (policies/fused_ops.py)
            step_vec, _ = self._fused.cg_step_vector(subsample_inputs, flat_g, self._cg_iters, self._reg_coeff,
                                                     self._max_constraint_val,
                                                     reuse_cg_residual=getattr(self, "_reuse_cg_residual", True))
        else:
            if fused_cg:
                descent_direction, dHd = self._fused.cg(subsample_inputs, flat_g, self._cg_iters, self._reg_coeff)
            else:
                Hx = hvp.build_eval(subsample_inputs, idx)
                descent_direction = krylov.cg(Hx, flat_g, cg_iters=self._cg_iters)
                dHd = descent_direction.dot(Hx(descent_direction))
            initial_step_size = torch.sqrt(2.0 * self._max_constraint_val * (1. / (dHd + 1e-8)))
            initial_step_size = torch.where(torch.isnan(initial_step_size),
                                            torch.ones_like(initial_step_size), initial_step_size)
            flat_descent_step = initial_step_size * descent_direction
            prev_param = (full_prev if idx is None else full_prev[idx]).to(torch.float64)
        logger.log("descent direction computed")

        n_iter = 0
        loss = constraint_val = float("nan")
        loss_before = None
        ratios = self._backtrack_ratio ** np.arange(self._max_backtracks)
        first = 0
        accepted = False
        n_spec = min(int(getattr(self, "_device_line_search", 3)), self._max_backtracks)
        if D.is_distributed():
            # every candidate's sums cross the ranks (a host-issued all-gather) whether its pass was gated off or not, so
            # the speculation is kept to the two candidates an iteration typically needs
            n_spec = min(n_spec, 2)
        if (step_vec is not None and n_spec > 0 and getattr(self._fused, "device_line_search", False)
                and getattr(before, "record", None) is not None and before.record.get("rows") is not None):
            # the first candidates decided on the device: nothing between the CG launches and the end of the search
            # waits for the host, and ``_after_enqueue`` (the next rollout, set by the algorithm) is queued behind the
            # search before its outcome is read.  In the rare case that none of them is accepted the loop below
            # goes on from candidate n_spec exactly as the reference does (and the speculative rollout is discarded by
            # the sampler: the parameter version it was launched at is gone).
            rec = self._fused.line_search_device(inputs, full_prev, step_vec, ratios[:n_spec],
                                                 self._max_constraint_val, before.record)
            hook = getattr(self, "_after_enqueue", None)
            if hook is not None:
                hook()
            loss_before, constraint_before = before()
            k_acc, evals = self._fused.line_search_resolve(rec, inputs)
            accepted = k_acc is not None
            n_iter = k_acc if accepted else n_spec - 1
            loss, constraint_val = evals[n_iter]
            first = n_spec
        for n_iter, ratio in (() if accepted else list(enumerate(ratios))[first:]):
            if step_vec is not None:
                self._fused.line_search_point(full_prev, step_vec, float(ratio))   # prev - ratio * step, in place
            else:
                cur_param = prev_param - float(ratio) * flat_descent_step
                target.set_param_values(cur_param, trainable=True)
            l_t, c_t = self._loss_constraint(inputs)
            loss, constraint_val = float(l_t), float(c_t)
            if loss_before is None:
                loss_before, constraint_before = before()
            if loss < loss_before and constraint_val <= self._max_constraint_val:
                break
        if loss_before is None:
            loss_before, constraint_before = before()
        self.last_before = (loss_before, constraint_before)
        if (np.isnan(loss) or np.isnan(constraint_val) or loss >= loss_before or
                constraint_val >= self._max_constraint_val) and not self._accept_violation:
            logger.log("Line search condition violated. Rejecting the step!")
            if np.isnan(loss):
                logger.log("Violated because loss is NaN")
            if np.isnan(constraint_val):
                logger.log("Violated because constraint %s is NaN" % self._constraint_name)
            if loss >= loss_before:
                logger.log("Violated because loss not improving")
            if constraint_val >= self._max_constraint_val:
                logger.log("Violated because constraint %s is violated" % self._constraint_name)
            with torch.no_grad():
                target.flat_params.copy_(full_prev)
        self.last_backtrack_iters = n_iter
        logger.log("backtrack iters: %d" % n_iter)
        logger.log("computing loss after")
        logger.log("optimization finished")
