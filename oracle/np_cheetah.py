"""Independent float64 restatement of the HalfCheetah-style env (TEST INFRASTRUCTURE).

Nothing here shares code, constants or formulation with rllab_amd/csrc/dyn_planar.h /
dyn_cheetah.h / cheetah_constants.h: bodies, joints and geoms are typed in again from
vendor/mujoco_models/half_cheetah.xml (reference, lines 36-93) in MuJoCo's own (x, z)
coordinates with hinge angles about +y, capsule mass properties are integrated numerically
(quadrature over the solid of revolution) instead of using closed forms, and the equations
of motion come from automatic differentiation of the Lagrangian

    T = sum_i 1/2 m_i |d p_i/dt|^2 + 1/2 I_i (d phi_i/dt)^2 + sum_j 1/2 armature_j qd_j^2
    M = d^2T/dqd^2,  c = d(M qd)/dq qd - dT/dq,  Q = -dV/dq + springs + dampers + motors
                                                     + limit penalties + J_c^T f_contact
    qacc = M^-1 (Q - c);  qd += h qacc;  q += h qd    (semi-implicit Euler, MuJoCo "Euler")

while the product uses composite-body inertias + a recursive bias/force pass in (z, x)
plane coordinates.  Agreement of the two (~1e-10 per env step, including ground contact
and beyond-range joints) pins the product's rigid-body physics.  What neither can pin is
MuJoCo 1.31 itself (proprietary, absent): joint limits and foot contacts are the engine's
documented penalty model, restated here from that description -- "parity
unpinned" against the reference for this env; BASELINE config C5 asks for an
fp32 tolerance check vs a CPU rollout, which tests/ run against this file.

Follows: rllab/envs/mujoco/half_cheetah_env.py:22-46 (obs, reward),
rllab/envs/mujoco/mujoco_env.py:109-116,184-191 (reset, step),
rllab/mujoco_py/mjcore.py:58-81 (comvel = subtree momentum / subtree mass).
"""
import numpy as np
import torch

R_GEOM = 0.046
TOTAL_MASS = 14.0
GRAVITY = 9.81
DT, SUBSTEPS = 0.01, 4                       # engine: one 0.01 s step = 4 sub-steps of 0.0025 s
LIMIT_K, LIMIT_B = 2.0e3, 15.0               # engine's penalty joint-limit model
CONTACT_K, CONTACT_B, FRICTION_C, MU = 2.0e4, 3.0e2, 3.0e2, 0.4

# name, parent, body pos in parent frame (x, z), geoms: (centre x, centre z, axis angle about y, half length)
# fromto='-.5 0 0 .5 0 0' is a capsule along x: axis angle pi/2 about y, centre 0, half length 0.5
BODIES = [
    ("torso", None, (0.0, 0.7), [(0.0, 0.0, np.pi / 2, 0.5), (0.6, 0.1, 0.87, 0.15)]),
    ("bthigh", "torso", (-0.5, 0.0), [(0.1, -0.13, -3.8, 0.145)]),
    ("bshin", "bthigh", (0.16, -0.25), [(-0.14, -0.07, -2.03, 0.15)]),
    ("bfoot", "bshin", (-0.28, -0.14), [(0.03, -0.097, -0.27, 0.094)]),
    ("fthigh", "torso", (0.5, 0.0), [(-0.07, -0.12, 0.52, 0.133)]),
    ("fshin", "fthigh", (-0.14, -0.24), [(0.065, -0.09, -0.6, 0.106)]),
    ("ffoot", "fshin", (0.13, -0.18), [(0.045, -0.07, -0.6, 0.07)]),
]
# hinge: range lo, hi, stiffness, damping, armature (default class), motor gear
JOINTS = {
    "bthigh": (-0.52, 1.05, 240.0, 6.0, 0.1, 120.0),
    "bshin": (-0.785, 0.785, 180.0, 4.5, 0.1, 90.0),
    "bfoot": (-0.4, 0.785, 120.0, 3.0, 0.1, 60.0),
    "fthigh": (-1.0, 0.7, 180.0, 4.5, 0.1, 120.0),
    "fshin": (-1.2, 0.87, 120.0, 3.0, 0.1, 60.0),
    "ffoot": (-0.5, 0.5, 60.0, 1.5, 0.1, 30.0),
}
NAMES = [b[0] for b in BODIES]


def capsule_mass_inertia(half_len, r=R_GEOM, rho=1.0, n=200001):
    """Mass and transverse moment of inertia (about the centre, axis normal to the capsule
    axis) of a solid capsule, by quadrature over discs of radius rad(s) along the axis:
    dm = rho pi rad^2 ds, dI = dm (rad^2/4 + s^2)."""
    s = np.linspace(-(half_len + r), half_len + r, n)
    over = np.clip(np.abs(s) - half_len, 0.0, None)
    rad2 = np.clip(r * r - over * over, 0.0, None)
    dm = rho * np.pi * rad2
    f_i = dm * (rad2 / 4.0 + s * s)
    trap = getattr(np, "trapezoid", None) or np.trapz
    return trap(dm, s), trap(f_i, s)


def _body_constants():
    raw = []
    for name, parent, pos, geoms in BODIES:
        parts = []
        for gx, gz, ang, hl in geoms:
            m, i = capsule_mass_inertia(hl)
            parts.append((m, i, gx, gz))
        m = sum(p[0] for p in parts)
        cx = sum(p[0] * p[2] for p in parts) / m
        cz = sum(p[0] * p[3] for p in parts) / m
        inertia = sum(p[1] + p[0] * ((p[2] - cx) ** 2 + (p[3] - cz) ** 2) for p in parts)
        raw.append([m, inertia, cx, cz])
    scale = TOTAL_MASS / sum(b[0] for b in raw)       # settotalmass rescales masses and inertias
    return [(m * scale, i * scale, cx, cz) for m, i, cx, cz in raw]


BODY_CONST = _body_constants()   # per body: mass, inertia about COM, COM (x, z) in body frame


def _rot(theta, lx, lz):
    """Rotate the local (x, z) vector by `theta` about +y (right-handed: z turns towards x)."""
    c, s = torch.cos(theta), torch.sin(theta)
    return lx * c + lz * s, -lx * s + lz * c


def frames(q):
    """World frame origin (x, z) and absolute pitch of every body.  q = MuJoCo qpos
    [rootx, rootz, rooty, bthigh, bshin, bfoot, fthigh, fshin, ffoot]."""
    out = {}
    for k, (name, parent, pos, _) in enumerate(BODIES):
        if parent is None:
            out[name] = (q[0] + pos[0], q[1] + pos[1], q[2])
        else:
            px, pz, pth = out[parent]
            dx, dz = _rot(pth, pos[0], pos[1])
            out[name] = (px + dx, pz + dz, pth + q[2 + k])
    return out


def _pose_vector(q):
    """[com x, com z] * 7 followed by the 7 absolute pitches."""
    fr = frames(q)
    xs, th = [], []
    for (name, _, _, _), (m, i, cx, cz) in zip(BODIES, BODY_CONST):
        ox, oz, t = fr[name]
        dx, dz = _rot(t, cx, cz)
        xs += [ox + dx, oz + dz]
        th.append(t)
    return torch.stack(xs + th)


def _contact_points(q):
    """World (x, z) of every capsule end-sphere centre, and the owning body index."""
    fr = frames(q)
    pts, owner = [], []
    for k, (name, _, _, geoms) in enumerate(BODIES):
        ox, oz, t = fr[name]
        for gx, gz, ang, hl in geoms:
            ax, az = np.sin(ang), np.cos(ang)       # capsule axis = local z rotated by ang about y
            for sgn in (1.0, -1.0):
                dx, dz = _rot(t, gx + sgn * hl * ax, gz + sgn * hl * az)
                pts += [ox + dx, oz + dz]
                owner.append(k)
    return torch.stack(pts), owner


MASSES = torch.tensor([b[0] for b in BODY_CONST], dtype=torch.float64)
INERTIAS = torch.tensor([b[1] for b in BODY_CONST], dtype=torch.float64)
ARMATURE = torch.tensor([0.0, 0.0, 0.0] + [JOINTS[n][4] for n in NAMES[1:]], dtype=torch.float64)


def qacc(q, qd, ctrl):
    q = torch.as_tensor(q, dtype=torch.float64)
    qd = torch.as_tensor(qd, dtype=torch.float64)
    nb = len(BODIES)

    def kinetic(qq, v):
        J = torch.autograd.functional.jacobian(_pose_vector, qq, create_graph=True)
        w = J @ v
        lin = w[:2 * nb].reshape(nb, 2)
        return (0.5 * (MASSES * (lin ** 2).sum(1)).sum() + 0.5 * (INERTIAS * w[2 * nb:] ** 2).sum()
                + 0.5 * (ARMATURE * v ** 2).sum())

    def potential(qq):
        pv = _pose_vector(qq)
        return GRAVITY * (MASSES * pv[:2 * nb].reshape(nb, 2)[:, 1]).sum()

    M = torch.autograd.functional.hessian(lambda v: kinetic(q, v), qd)
    momentum = lambda qq: torch.autograd.functional.jacobian(lambda v: kinetic(qq, v), qd, create_graph=True)
    c = torch.autograd.functional.jacobian(momentum, q) @ qd - torch.autograd.functional.jacobian(lambda qq: kinetic(qq, qd), q)
    Q = -torch.autograd.functional.jacobian(potential, q)
    # hinges: spring (ref 0), damper, geared motor with ctrl clamped to +-1, penalty range limits
    for k, name in enumerate(NAMES[1:]):
        j = 3 + k
        lo, hi, stiff, damp, _, gear = JOINTS[name]
        t = -stiff * q[j] - damp * qd[j] + gear * float(np.clip(ctrl[k], -1.0, 1.0))
        if q[j] < lo:
            t = t - LIMIT_K * (q[j] - lo) - LIMIT_B * qd[j]
        if q[j] > hi:
            t = t - LIMIT_K * (q[j] - hi) - LIMIT_B * qd[j]
        Q[j] = Q[j] + t
    # floor contacts: penalty normal force + viscous-regularised Coulomb friction at the lowest
    # point of each capsule end sphere
    pts, owner = _contact_points(q)
    Jc = torch.autograd.functional.jacobian(lambda qq: _contact_points(qq)[0], q)      # [2*nc, 9]
    Jth = torch.autograd.functional.jacobian(lambda qq: _pose_vector(qq)[2 * nb:], q)  # [nb, 9]
    vel = Jc @ qd
    for cidx, b in enumerate(owner):
        z = pts[2 * cidx + 1]
        depth = R_GEOM - z
        if depth > 0:
            vx, vz = vel[2 * cidx], vel[2 * cidx + 1]
            fn = torch.clamp(CONTACT_K * depth - CONTACT_B * vz, min=0.0)
            ft = -torch.clamp(FRICTION_C * vx, -MU * fn, MU * fn)
            # force (ft, fn) acts at centre + (0, -r): equivalent wrench at the centre adds the
            # pitch torque of the lever: tau_y = lever_z * F_x - lever_x * F_z = -r * ft
            Q = Q + Jc[2 * cidx] * ft + Jc[2 * cidx + 1] * fn + Jth[b] * (-R_GEOM * ft)
    return torch.linalg.solve(M, Q - c)


def com_and_vel(q, qd):
    q = torch.as_tensor(q, dtype=torch.float64)
    qd = torch.as_tensor(qd, dtype=torch.float64)
    nb = len(BODIES)
    lin = lambda qq: _pose_vector(qq)[:2 * nb]
    p = lin(q).reshape(nb, 2)
    v = (torch.autograd.functional.jacobian(lin, q) @ qd).reshape(nb, 2)
    w = MASSES / MASSES.sum()
    return (w[:, None] * p).sum(0).numpy(), (w[:, None] * v).sum(0).numpy()


def observe(qpos, qvel):
    com, _ = com_and_vel(qpos, qvel)
    return np.concatenate([qpos[1:], qvel, [com[0], 0.0, com[1]]])


def step(qpos, qvel, action, normalize=True):
    """One HalfCheetahEnv.step (behind NormalizedEnv when ``normalize``): returns
    (qpos, qvel, obs, reward, done) in MuJoCo's coordinate order."""
    a = np.asarray(action, dtype=np.float64)
    if normalize:
        a = np.clip(-1.0 + (a + 1.0) * 0.5 * 2.0, -1.0, 1.0)
    q = torch.as_tensor(qpos, dtype=torch.float64).clone()
    qd = torch.as_tensor(qvel, dtype=torch.float64).clone()
    h = DT / SUBSTEPS
    for _ in range(SUBSTEPS):
        acc = qacc(q, qd, a)
        qd = qd + h * acc
        q = q + h * qd
    com, comvel = com_and_vel(q, qd)
    reward = comvel[0] - 1e-1 * 0.5 * np.sum(np.square(np.clip(a, -1.0, 1.0)))
    qn, qdn = q.numpy(), qd.numpy()
    return qn, qdn, np.concatenate([qn[1:], qdn, [com[0], 0.0, com[1]]]), reward, False


def reset(draws):
    z = np.asarray(draws, dtype=np.float64)
    return 0.01 * z[:9], 0.1 * z[9:]


# --- mapping to / from the engine's state vector -------------------------------------------
def to_engine_state(qpos, qvel):
    """Engine state = [z_abs, x, rooty, joints(6), zdot, xdot, rooty_dot, joint vel(6)]."""
    s = np.zeros(18)
    s[0], s[1], s[2:9] = qpos[1] + 0.7, qpos[0], qpos[2:]
    s[9], s[10], s[11:] = qvel[1], qvel[0], qvel[2:]
    return s


def from_engine_state(s):
    qpos = np.concatenate([[s[1], s[0] - 0.7], s[2:9]])
    qvel = np.concatenate([[s[10], s[9]], s[11:]])
    return qpos, qvel
