"""Conjugate gradient (mirrors rllab/misc/krylov.py:7-39, Demmel p.312).

The iteration is the reference's, including the ``rdotr < residual_tol`` early
exit, but all vectors stay on the device in float64 and there is NO host
synchronisation inside the loop: the early exit is realised by an ``active``
flag that freezes x / r / p once the residual test fires (identical result,
at most cg_iters Hx evaluations).
"""
import torch


def cg(f_Ax, b, cg_iters=10, callback=None, verbose=False, residual_tol=1e-10):
    b = torch.as_tensor(b).to(torch.float64)
    p = b.clone()
    r = b.clone()
    x = torch.zeros_like(b)
    rdotr = r.dot(r)
    active = torch.ones((), dtype=torch.bool, device=b.device)
    tol = torch.as_tensor(residual_tol, dtype=torch.float64, device=b.device)
    for i in range(cg_iters):
        if callback is not None:
            callback(x)
        if verbose:
            print("%10i %10.3g %10.3g" % (i, float(rdotr), float(x.norm())))
        z = f_Ax(p).to(torch.float64)
        v = rdotr / p.dot(z)
        x_new = x + v * p
        r_new = r - v * z
        newrdotr = r_new.dot(r_new)
        mu = newrdotr / rdotr
        p_new = r_new + mu * p
        x = torch.where(active, x_new, x)
        r = torch.where(active, r_new, r)
        p = torch.where(active, p_new, p)
        rdotr = torch.where(active, newrdotr, rdotr)
        active = active & (rdotr >= tol)
    if callback is not None:
        callback(x)
    return x


def preconditioned_cg(f_Ax, f_Minvx, b, cg_iters=10, callback=None, verbose=False, residual_tol=1e-10):
    """Preconditioned CG (rllab/misc/krylov.py:40-75, Demmel p.318): the direction is built from y = M^-1 r and the
    stopping test is on y.r.  Same device-resident, sync-free form as ``cg``: the early exit freezes the iterates."""
    b = torch.as_tensor(b).to(torch.float64)
    x = torch.zeros_like(b)
    r = b.clone()
    p = f_Minvx(b).to(torch.float64)
    ydotr = p.dot(r)
    active = torch.ones((), dtype=torch.bool, device=b.device)
    tol = torch.as_tensor(residual_tol, dtype=torch.float64, device=b.device)
    for i in range(cg_iters):
        if callback is not None:
            callback(x, f_Ax)
        if verbose:
            print("%10i %10.3g %10.3g" % (i, float(ydotr), float(x.norm())))
        z = f_Ax(p).to(torch.float64)
        v = ydotr / p.dot(z)
        x_new = x + v * p
        r_new = r - v * z
        y = f_Minvx(r_new).to(torch.float64)
        newydotr = y.dot(r_new)
        p_new = y + (newydotr / ydotr) * p
        x = torch.where(active, x_new, x)
        r = torch.where(active, r_new, r)
        p = torch.where(active, p_new, p)
        ydotr = torch.where(active, newydotr, ydotr)
        active = active & (ydotr >= tol)
    return x
