"""Independent float64 restatement of the Swimmer-style env (TEST INFRASTRUCTURE).

Nothing here shares code or formulation with rllab_amd/csrc/dyn_planar.h: the
equations of motion are derived by automatic differentiation of the Lagrangian
written directly from vendor/mujoco_models/swimmer.xml (reference), i.e.

    T(q, qd) = sum_i 1/2 m_i |d p_i/dt|^2 + 1/2 I_i (d phi_i/dt)^2
    M = d^2 T / d qd^2,   c = d(M qd)/dq qd - dT/dq,   Q = sum_i J_i^T f_i + ...
    qacc = M^-1 (Q - c);  qd += h qacc;  q += h qd          (MuJoCo "Euler")

while the product uses composite-body inertia + a recursive bias/force pass.  Agreement
of the two to ~1e-9 after a full env step (50 substeps) is what pins the product's
physics; what neither can pin is MuJoCo 1.31 itself (proprietary, absent): the
reference's arithmetic for this env is "parity unpinned" (SURVEY.md 8c).

Follows: rllab/envs/mujoco/swimmer_env.py:25-45 (obs, reward),
rllab/envs/mujoco/mujoco_env.py:109-116,184-191 (reset, frame_skip loop),
rllab/mujoco_py/mjcore.py:58-81 (comvel = subtree momentum / subtree mass).
"""
import numpy as np
import torch

R_CAP, L_CAP, RHO_BODY = 0.1, 1.0, 1000.0
RHO_FLUID, MU_FLUID = 4000.0, 0.1
DT, FRAME_SKIP = 0.001, 50
LIMIT = np.deg2rad(100.0)
LIMIT_K, LIMIT_B = 1.0e4, 5.0e2   # engine's penalty limit model


def capsule_constants():
    r, L, rho = R_CAP, L_CAP, RHO_BODY
    m_c = rho * np.pi * r * r * L
    m_s = rho * 4.0 / 3.0 * np.pi * r ** 3
    m = m_c + m_s
    i_zz = m_c * (L * L / 12 + r * r / 4) + m_s * (83.0 / 320 * r * r + (L / 2 + 3 * r / 8) ** 2)
    i_xx = m_c * r * r / 2 + 0.4 * m_s * r * r
    bx = np.sqrt(6 * (2 * i_zz - i_xx) / m)
    by = np.sqrt(6 * i_xx / m)
    return m, i_zz, bx, by


MASS, INERTIA, BX, BY = capsule_constants()


def body_poses(q):
    """COM positions [3,2] and absolute angles [3] of torso, mid, back (torch, differentiable)."""
    x, y, t0, t1, t2 = q
    phi = torch.stack([t0, t0 + t1, t0 + t1 + t2])
    u = torch.stack([torch.cos(phi), torch.sin(phi)], dim=1)  # body x-axes
    origin = torch.stack([x, y])
    p0 = origin + 1.0 * u[0]            # torso capsule spans local x in [0.5, 1.5]
    j1 = origin + 0.5 * u[0]            # mid hinge at torso-local (0.5, 0)
    p1 = j1 - 0.5 * u[1]                # mid capsule spans local x in [-1, 0]
    j2 = j1 - 1.0 * u[1]                # back hinge at mid-local (-1, 0)
    p2 = j2 - 0.5 * u[2]
    return torch.stack([p0, p1, p2]), phi


def _flat_pose(q):
    p, phi = body_poses(q)
    return torch.cat([p.reshape(-1), phi])


def qacc(q, qd, ctrl):
    q = torch.as_tensor(q, dtype=torch.float64)
    qd = torch.as_tensor(qd, dtype=torch.float64)
    J = torch.autograd.functional.jacobian(_flat_pose, q)          # [9, 5]
    Jp, Jphi = J[:6].reshape(3, 2, 5), J[6:]

    def kinetic(qq, qqd):
        Jl = torch.autograd.functional.jacobian(_flat_pose, qq, create_graph=True)
        v = Jl @ qqd
        return 0.5 * MASS * (v[:6] ** 2).sum() + 0.5 * INERTIA * (v[6:] ** 2).sum()

    M = torch.autograd.functional.hessian(lambda v: kinetic(q, v), qd)

    def momentum(qq):
        return torch.autograd.functional.jacobian(lambda v: kinetic(qq, v), qd, create_graph=True)
    dp_dq = torch.autograd.functional.jacobian(momentum, q)        # d(M qd)/dq  [5,5]
    dT_dq = torch.autograd.functional.jacobian(lambda qq: kinetic(qq, qd), q)
    c = dp_dq @ qd - dT_dq
    # fluid forces (MuJoCo inertia-box model), per body in its own frame
    _, phi = body_poses(q)
    v = (Jp @ qd)                                                  # [3,2] COM velocities
    w = Jphi @ qd
    diam = (BX + 2 * BY) / 3
    Q = torch.zeros(5, dtype=torch.float64)
    for i in range(3):
        cs, sn = torch.cos(phi[i]), torch.sin(phi[i])
        vl = cs * v[i, 0] + sn * v[i, 1]
        vt = -sn * v[i, 0] + cs * v[i, 1]
        fl = -3 * np.pi * diam * MU_FLUID * vl - 0.5 * RHO_FLUID * BY * BY * vl.abs() * vl
        ft = -3 * np.pi * diam * MU_FLUID * vt - 0.5 * RHO_FLUID * BX * BY * vt.abs() * vt
        f = torch.stack([cs * fl - sn * ft, sn * fl + cs * ft])
        tz = -np.pi * diam ** 3 * MU_FLUID * w[i] - RHO_FLUID * BY * (BX ** 4 + BY ** 4) * w[i].abs() * w[i] / 64
        Q = Q + Jp[i].t() @ f + Jphi[i] * tz
    # actuators (gear 1, ctrl clamped to +-50) and penalty joint limits on the two hinges
    for k, j in enumerate((3, 4)):
        Q[j] = Q[j] + float(np.clip(ctrl[k], -50.0, 50.0))
        if q[j] < -LIMIT:
            Q[j] = Q[j] - LIMIT_K * (q[j] + LIMIT) - LIMIT_B * qd[j]
        if q[j] > LIMIT:
            Q[j] = Q[j] - LIMIT_K * (q[j] - LIMIT) - LIMIT_B * qd[j]
    return torch.linalg.solve(M, Q - c)


def com_and_vel(q, qd):
    q = torch.as_tensor(q, dtype=torch.float64)
    qd = torch.as_tensor(qd, dtype=torch.float64)
    p, _ = body_poses(q)
    J = torch.autograd.functional.jacobian(lambda qq: body_poses(qq)[0].reshape(-1), q).reshape(3, 2, 5)
    v = J @ qd
    return p.mean(0).numpy(), v.mean(0).numpy()   # equal masses


def observe(state):
    com, _ = com_and_vel(state[:5], state[5:])
    return np.concatenate([state, com, [0.0]])


def step(state, action, normalize=True):
    """One SwimmerEnv.step (behind NormalizedEnv when ``normalize``).  Returns
    (next_state, obs, reward, done)."""
    a = np.asarray(action, dtype=np.float64)
    lb, ub = -50.0, 50.0
    if normalize:
        a = np.clip(lb + (a + 1.0) * 0.5 * (ub - lb), lb, ub)
    q = torch.as_tensor(state[:5], dtype=torch.float64).clone()
    qd = torch.as_tensor(state[5:], dtype=torch.float64).clone()
    for _ in range(FRAME_SKIP):
        acc = qacc(q, qd, a)
        qd = qd + DT * acc
        q = q + DT * qd
    state = np.concatenate([q.numpy(), qd.numpy()])
    com, comvel = com_and_vel(q, qd)
    scaling = (ub - lb) * 0.5
    reward = comvel[0] - 0.5 * 1e-2 * np.sum(np.square(a / scaling))
    return state, np.concatenate([state, com, [0.0]]), reward, False


def reset(draws):
    z = np.asarray(draws, dtype=np.float64)
    return np.concatenate([0.01 * z[:5], 0.1 * z[5:]])


# ---- joint limits by MuJoCo's documented soft-constraint model (SwimmerEnv(limit_model="mujoco")) ---------------------
# Independent restatement in MuJoCo's OWN coordinates (qpos = x, y, torso angle, rot2, rot3; the product works in absolute
# body angles with the translations eliminated): the full 5 x 5 inertia from the Lagrangian above, the limit rows as unit
# vectors on the hinge coordinates, the constraint forces by non-negative least squares of the quadratic
#     1/2 f' (A + R) f + f' (a0 - a_ref),   A = J M^-1 J',  a0 = J qacc_unconstrained,  R = diag((1 - d) / d A_ii),
# with the reference acceleration a_ref = -b (J v) - k dist, b = 2 / (dmax timeconst), k = d / (dmax^2 timeconst^2 dampratio^2)
# and the impedance d(dist) of solimplimit = (0, .8, .03); solreflimit = (.02, 1) (vendor/mujoco_models/swimmer.xml:31,34).
MJ_TIMECONST, MJ_DAMPRATIO, MJ_DMIN, MJ_DMAX, MJ_WIDTH = 0.02, 1.0, 0.0, 0.8, 0.03


def mj_impedance(dist):
    x = min(abs(dist) / MJ_WIDTH, 1.0)
    y = 2.0 * x * x if x < 0.5 else 1.0 - 2.0 * (1.0 - x) ** 2
    return float(np.clip(MJ_DMIN + y * (MJ_DMAX - MJ_DMIN), 1e-4, 0.9999))


def _mass_bias_forces(q, qd, ctrl):
    """M, and Q - c WITHOUT any joint-limit term (same derivation as qacc above)."""
    q = torch.as_tensor(q, dtype=torch.float64)
    qd = torch.as_tensor(qd, dtype=torch.float64)
    J = torch.autograd.functional.jacobian(_flat_pose, q)
    Jp, Jphi = J[:6].reshape(3, 2, 5), J[6:]

    def kinetic(qq, qqd):
        Jl = torch.autograd.functional.jacobian(_flat_pose, qq, create_graph=True)
        v = Jl @ qqd
        return 0.5 * MASS * (v[:6] ** 2).sum() + 0.5 * INERTIA * (v[6:] ** 2).sum()

    M = torch.autograd.functional.hessian(lambda v: kinetic(q, v), qd)

    def momentum(qq):
        return torch.autograd.functional.jacobian(lambda v: kinetic(qq, v), qd, create_graph=True)
    dp_dq = torch.autograd.functional.jacobian(momentum, q)
    dT_dq = torch.autograd.functional.jacobian(lambda qq: kinetic(qq, qd), q)
    c = dp_dq @ qd - dT_dq
    _, phi = body_poses(q)
    v = (Jp @ qd)
    w = Jphi @ qd
    diam = (BX + 2 * BY) / 3
    Q = torch.zeros(5, dtype=torch.float64)
    for i in range(3):
        cs, sn = torch.cos(phi[i]), torch.sin(phi[i])
        vl = cs * v[i, 0] + sn * v[i, 1]
        vt = -sn * v[i, 0] + cs * v[i, 1]
        fl = -3 * np.pi * diam * MU_FLUID * vl - 0.5 * RHO_FLUID * BY * BY * vl.abs() * vl
        ft = -3 * np.pi * diam * MU_FLUID * vt - 0.5 * RHO_FLUID * BX * BY * vt.abs() * vt
        f = torch.stack([cs * fl - sn * ft, sn * fl + cs * ft])
        tz = -np.pi * diam ** 3 * MU_FLUID * w[i] - RHO_FLUID * BY * (BX ** 4 + BY ** 4) * w[i].abs() * w[i] / 64
        Q = Q + Jp[i].t() @ f + Jphi[i] * tz
    for k, j in enumerate((3, 4)):
        Q[j] = Q[j] + float(np.clip(ctrl[k], -50.0, 50.0))
    return M.numpy(), (Q - c).numpy()


def qacc_mujoco_limits(q, qd, ctrl):
    from scipy.optimize import nnls
    q = np.asarray(q, dtype=np.float64)
    qd = np.asarray(qd, dtype=np.float64)
    M, rhs = _mass_bias_forces(q, qd, ctrl)
    a_unc = np.linalg.solve(M, rhs)
    rows, dists = [], []
    for j in (3, 4):
        lo, hi = q[j] + LIMIT, LIMIT - q[j]                # dist to the lower / upper limit
        if lo < 0:
            e = np.zeros(5); e[j] = 1.0
            rows.append(e); dists.append(lo)
        elif hi < 0:
            e = np.zeros(5); e[j] = -1.0
            rows.append(e); dists.append(hi)
    if not rows:
        return a_unc
    J = np.stack(rows)
    Minv_Jt = np.linalg.solve(M, J.T)
    A = J @ Minv_Jt
    d = np.array([mj_impedance(x) for x in dists])
    b = 2.0 / (MJ_DMAX * MJ_TIMECONST)
    k = d / (MJ_DMAX ** 2 * MJ_TIMECONST ** 2 * MJ_DAMPRATIO ** 2)
    a_ref = -b * (J @ qd) - k * np.array(dists)
    H = A + np.diag((1.0 - d) / d * np.diag(A))
    g = a_ref - J @ a_unc
    # min 1/2 f'H f - f'g, f >= 0  ==  min |L' f - L^-1 g|^2, H = L L'
    L = np.linalg.cholesky(H)
    f, _ = nnls(L.T, np.linalg.solve(L, g), maxiter=200)
    return a_unc + Minv_Jt @ f


def step_mujoco_limits(state, action, normalize=True):
    """One SwimmerEnv(limit_model="mujoco").step; returns (next_state, obs, reward, done)."""
    a = np.asarray(action, dtype=np.float64)
    lb, ub = -50.0, 50.0
    if normalize:
        a = np.clip(lb + (a + 1.0) * 0.5 * (ub - lb), lb, ub)
    q, qd = np.array(state[:5], dtype=np.float64), np.array(state[5:], dtype=np.float64)
    for _ in range(FRAME_SKIP):
        acc = qacc_mujoco_limits(q, qd, a)
        qd = qd + DT * acc
        q = q + DT * qd
    state = np.concatenate([q, qd])
    com, comvel = com_and_vel(q, qd)
    scaling = (ub - lb) * 0.5
    reward = comvel[0] - 0.5 * 1e-2 * np.sum(np.square(a / scaling))
    return state, np.concatenate([state, com, [0.0]]), reward, False
