"""Independent float64 restatement of the MuJoCo-style planar tree envs (TEST INFRASTRUCTURE):
a model-driven version of oracle/np_cheetah.py's derivation (automatic differentiation of the Lagrangian
written in MuJoCo's own (x, z) coordinates, capsule mass properties by numerical quadrature, penalty joint
limits and floor contacts restated from DESIGN.md), used for the Walker2D- and Hopper-style envs.  Nothing here shares
code, constants or formulation with rllab_amd/csrc/dyn_planar.h / dyn_walker.h / walker_constants.h: bodies,
joints and geoms are typed in again from vendor/mujoco_models/walker2d.xml (reference, lines 3-59) and
hopper.xml (lines 3-49).

Follows: rllab/envs/mujoco/walker2d_env.py:28-49, hopper_env.py:38-62 (obs, reward, done),
rllab/envs/mujoco/mujoco_env.py:109-116,184-191 (reset, step), rllab/mujoco_py/mjcore.py:58-81 (comvel).
"""
import numpy as np
import torch

GRAVITY = 9.81
LIMIT_K, LIMIT_B = 2.0e3, 15.0               # engine's penalty joint-limit model
CONTACT_K, CONTACT_B, FRICTION_C = 2.0e4, 3.0e2, 3.0e2


def capsule_mass_inertia(half_len, r, rho=1000.0, n=200001):
    """Mass and transverse inertia of a solid capsule by quadrature over discs along the axis."""
    s = np.linspace(-(half_len + r), half_len + r, n)
    over = np.clip(np.abs(s) - half_len, 0.0, None)
    rad2 = np.clip(r * r - over * over, 0.0, None)
    dm = rho * np.pi * rad2
    trap = getattr(np, "trapezoid", None) or np.trapz
    return trap(dm, s), trap(dm * (rad2 / 4.0 + s * s), s)


def _rot(theta, lx, lz):
    """Rotate the local (x, z) vector by `theta` about +y (right-handed: z turns towards x)."""
    c, s = torch.cos(theta), torch.sin(theta)
    return lx * c + lz * s, -lx * s + lz * c


class PlanarModel(object):
    """bodies: (name, parent name | None, hinge position in the parent frame (x, z), geoms) with
    geoms = (centre x, centre z, axis angle about +y, half length, radius, friction) in the body frame whose
    origin is the hinge; joints: name -> (lo, hi, stiffness, damping, armature, gear, axis sign) in MuJoCo's
    joint convention (axis sign -1 = hinge about -y)."""

    def __init__(self, bodies, joints, root_height, dt, substeps, total_mass=None):
        self.bodies, self.joints = bodies, joints
        self.names = [b[0] for b in bodies]
        self.root_height, self.dt, self.substeps = root_height, dt, substeps
        raw = []
        for name, parent, pos, geoms in bodies:
            parts = []
            for gx, gz, ang, hl, rad, fr in geoms:
                m, i = capsule_mass_inertia(hl, rad)
                parts.append((m, i, gx, gz))
            m = sum(p[0] for p in parts)
            cx = sum(p[0] * p[2] for p in parts) / m
            cz = sum(p[0] * p[3] for p in parts) / m
            inertia = sum(p[1] + p[0] * ((p[2] - cx) ** 2 + (p[3] - cz) ** 2) for p in parts)
            raw.append([m, inertia, cx, cz])
        scale = 1.0 if total_mass is None else total_mass / sum(b[0] for b in raw)
        self.const = [(m * scale, i * scale, cx, cz) for m, i, cx, cz in raw]
        self.masses = torch.tensor([b[0] for b in self.const], dtype=torch.float64)
        self.inertias = torch.tensor([b[1] for b in self.const], dtype=torch.float64)
        self.armature = torch.tensor([0.0, 0.0, 0.0] + [joints[n][4] for n in self.names[1:]], dtype=torch.float64)
        self.sign = np.array([joints[n][6] for n in self.names[1:]], dtype=np.float64)

    # q = MuJoCo qpos [rootz (absolute), rootx, rooty, joints...]  (walker2d.xml: rootz declared first)
    def frames(self, q):
        out = {}
        for k, (name, parent, pos, _) in enumerate(self.bodies):
            if parent is None:
                out[name] = (q[1] + pos[0], q[0] + pos[1], q[2])
            else:
                px, pz, pth = out[parent]
                dx, dz = _rot(pth, pos[0], pos[1])
                out[name] = (px + dx, pz + dz, pth + float(self.sign[k - 1]) * q[2 + k])
        return out

    def pose_vector(self, q):
        fr = self.frames(q)
        xs, th = [], []
        for (name, _, _, _), (m, i, cx, cz) in zip(self.bodies, self.const):
            ox, oz, t = fr[name]
            dx, dz = _rot(t, cx, cz)
            xs += [ox + dx, oz + dz]
            th.append(t)
        return torch.stack(xs + th)

    def contact_points(self, q):
        fr = self.frames(q)
        pts, owner, rads, mus = [], [], [], []
        for k, (name, _, _, geoms) in enumerate(self.bodies):
            ox, oz, t = fr[name]
            for gx, gz, ang, hl, rad, fric in geoms:
                ax, az = np.sin(ang), np.cos(ang)
                for sgn in (1.0, -1.0):
                    dx, dz = _rot(t, gx + sgn * hl * ax, gz + sgn * hl * az)
                    pts += [ox + dx, oz + dz]
                    owner.append(k); rads.append(rad); mus.append(fric)
        return torch.stack(pts), owner, rads, mus

    def qacc(self, q, qd, ctrl):
        q = torch.as_tensor(q, dtype=torch.float64)
        qd = torch.as_tensor(qd, dtype=torch.float64)
        nb = len(self.bodies)
        M_, I_, A_ = self.masses, self.inertias, self.armature

        def kinetic(qq, v):
            J = torch.autograd.functional.jacobian(self.pose_vector, qq, create_graph=True)
            w = J @ v
            lin = w[:2 * nb].reshape(nb, 2)
            return 0.5 * (M_ * (lin ** 2).sum(1)).sum() + 0.5 * (I_ * w[2 * nb:] ** 2).sum() + 0.5 * (A_ * v ** 2).sum()

        def potential(qq):
            return GRAVITY * (M_ * self.pose_vector(qq)[:2 * nb].reshape(nb, 2)[:, 1]).sum()
        M = torch.autograd.functional.hessian(lambda v: kinetic(q, v), qd)
        mom = lambda qq: torch.autograd.functional.jacobian(lambda v: kinetic(qq, v), qd, create_graph=True)
        c = torch.autograd.functional.jacobian(mom, q) @ qd - torch.autograd.functional.jacobian(lambda qq: kinetic(qq, qd), q)
        Q = -torch.autograd.functional.jacobian(potential, q)
        for k, name in enumerate(self.names[1:]):
            j = 3 + k
            lo, hi, stiff, damp, _, gear, _ = self.joints[name]
            t = -stiff * q[j] - damp * qd[j] + gear * float(ctrl[k])
            if q[j] < lo:
                t = t - LIMIT_K * (q[j] - lo) - LIMIT_B * qd[j]
            if q[j] > hi:
                t = t - LIMIT_K * (q[j] - hi) - LIMIT_B * qd[j]
            Q[j] = Q[j] + t
        pts, owner, rads, mus = self.contact_points(q)
        Jc = torch.autograd.functional.jacobian(lambda qq: self.contact_points(qq)[0], q)
        Jth = torch.autograd.functional.jacobian(lambda qq: self.pose_vector(qq)[2 * nb:], q)
        vel = Jc @ qd
        for cidx, b in enumerate(owner):
            depth = rads[cidx] - pts[2 * cidx + 1]
            if depth > 0:
                vx, vz = vel[2 * cidx], vel[2 * cidx + 1]
                fn = torch.clamp(CONTACT_K * depth - CONTACT_B * vz, min=0.0)
                ft = -torch.clamp(FRICTION_C * vx, -mus[cidx] * fn, mus[cidx] * fn)
                # (ft, fn) acts at centre + (0, -r): pitch torque of the lever = -r * ft
                Q = Q + Jc[2 * cidx] * ft + Jc[2 * cidx + 1] * fn + Jth[b] * (-rads[cidx] * ft)
        return torch.linalg.solve(M, Q - c)

    def constraint_qfrc(self, q, qd):
        """Generalised force (MuJoCo joint coordinates) of the penalty joint limits and floor contacts at (q, qd):
        the quantity the engine reports where MuJoCo reports data.qfrc_constraint."""
        q = torch.as_tensor(q, dtype=torch.float64)
        qd = torch.as_tensor(qd, dtype=torch.float64)
        nb = len(self.bodies)
        Q = torch.zeros_like(q)
        for k, name in enumerate(self.names[1:]):
            j = 3 + k
            lo, hi = self.joints[name][0], self.joints[name][1]
            if q[j] < lo:
                Q[j] = Q[j] - LIMIT_K * (q[j] - lo) - LIMIT_B * qd[j]
            if q[j] > hi:
                Q[j] = Q[j] - LIMIT_K * (q[j] - hi) - LIMIT_B * qd[j]
        pts, owner, rads, mus = self.contact_points(q)
        Jc = torch.autograd.functional.jacobian(lambda qq: self.contact_points(qq)[0], q)
        Jth = torch.autograd.functional.jacobian(lambda qq: self.pose_vector(qq)[2 * nb:], q)
        vel = Jc @ qd
        for cidx, b in enumerate(owner):
            depth = rads[cidx] - pts[2 * cidx + 1]
            if depth > 0:
                vx, vz = vel[2 * cidx], vel[2 * cidx + 1]
                fn = torch.clamp(CONTACT_K * depth - CONTACT_B * vz, min=0.0)
                ft = -torch.clamp(FRICTION_C * vx, -mus[cidx] * fn, mus[cidx] * fn)
                Q = Q + Jc[2 * cidx] * ft + Jc[2 * cidx + 1] * fn + Jth[b] * (-rads[cidx] * ft)
        return Q.numpy()

    def com_and_vel(self, q, qd):
        q = torch.as_tensor(q, dtype=torch.float64)
        qd = torch.as_tensor(qd, dtype=torch.float64)
        nb = len(self.bodies)
        lin = lambda qq: self.pose_vector(qq)[:2 * nb]
        p = lin(q).reshape(nb, 2)
        v = (torch.autograd.functional.jacobian(lin, q) @ qd).reshape(nb, 2)
        w = self.masses / self.masses.sum()
        return (w[:, None] * p).sum(0).numpy(), (w[:, None] * v).sum(0).numpy()

    def advance(self, qpos, qvel, ctrl):
        q = torch.as_tensor(qpos, dtype=torch.float64).clone()
        qd = torch.as_tensor(qvel, dtype=torch.float64).clone()
        h = self.dt / self.substeps
        for _ in range(self.substeps):
            acc = self.qacc(q, qd, ctrl)
            qd = qd + h * acc
            q = q + h * qd
        return q.numpy(), qd.numpy()


_D = np.pi / 180.0
# walker2d.xml:19-50 (coordinate="global"): offsets are differences of the file's absolute positions
WALKER = PlanarModel(
    bodies=[
        ("torso", None, (0.0, 0.0), [(0.0, 0.0, 0.0, 0.2, 0.05, 0.9)]),
        ("thigh", "torso", (0.0, 1.05 - 1.25), [(0.0, (1.05 + 0.6) / 2 - 1.05, 0.0, (1.05 - 0.6) / 2, 0.05, 0.9)]),
        ("leg", "thigh", (0.0, 0.6 - 1.05), [(0.0, (0.6 + 0.1) / 2 - 0.6, 0.0, (0.6 - 0.1) / 2, 0.04, 0.9)]),
        ("foot", "leg", (0.0, 0.1 - 0.6), [(0.1, 0.0, np.pi / 2, 0.1, 0.06, 0.9)]),
        ("thigh_left", "torso", (0.0, 1.05 - 1.25), [(0.0, (1.05 + 0.6) / 2 - 1.05, 0.0, (1.05 - 0.6) / 2, 0.05, 0.9)]),
        ("leg_left", "thigh_left", (0.0, 0.6 - 1.05), [(0.0, (0.6 + 0.1) / 2 - 0.6, 0.0, (0.6 - 0.1) / 2, 0.04, 0.9)]),
        ("foot_left", "leg_left", (0.0, 0.1 - 0.6), [(0.1, 0.0, np.pi / 2, 0.1, 0.06, 1.9)]),
    ],
    # lo, hi (MuJoCo joint coordinate, radians), stiffness, damping, armature, gear, axis sign
    joints={n: (lo * _D, hi * _D, 0.0, 0.1, 0.01, 1.0, -1.0) for n, lo, hi in
            [("thigh", -150, 0), ("leg", -150, 0), ("foot", -45, 45), ("thigh_left", -150, 0), ("leg_left", -150, 0),
             ("foot_left", -45, 45)]},
    root_height=1.25, dt=0.005, substeps=2)
WALKER_CTRL = np.array([150.0, 100.0, 100.0, 150.0, 100.0, 100.0])


def walker_reset(draws):
    z = np.asarray(draws, dtype=np.float64)
    qpos = 0.01 * z[:9]
    qpos[0] += 1.25
    return qpos, 0.1 * z[9:]


def walker_observe(qpos, qvel):
    com, _ = WALKER.com_and_vel(qpos, qvel)
    return np.concatenate([qpos, qvel, [com[0], 0.0, com[1]]])


def walker_step(qpos, qvel, action, normalize=True):
    """One Walker2DEnv.step (behind NormalizedEnv when ``normalize``) in MuJoCo's conventions."""
    a = np.asarray(action, dtype=np.float64)
    lb, ub = -WALKER_CTRL, WALKER_CTRL
    if normalize:
        a = np.clip(lb + (a + 1.0) * 0.5 * (ub - lb), lb, ub)
    a = np.clip(a, lb, ub)
    q, qd = WALKER.advance(qpos, qvel, a)
    com, comvel = WALKER.com_and_vel(q, qd)
    reward = comvel[0] - 0.5 * 1e-2 * np.sum(np.square(a / ((ub - lb) * 0.5)))
    done = not (0.8 < q[0] < 2.0 and -1.0 < q[2] < 1.0)
    return q, qd, np.concatenate([q, qd, [com[0], 0.0, com[1]]]), reward, done


def walker_to_engine_state(qpos, qvel):
    """Engine state (tree convention): hinges about -y carry the opposite sign."""
    sgn = np.concatenate([[1.0, 1.0, 1.0], WALKER.sign])
    return np.concatenate([qpos * sgn, qvel * sgn])


# hopper.xml:21-40 (coordinate="global"): offsets are differences of the file's absolute positions; every joint has
# damping 1 and armature 1 (defaults, :4), hinges about -y, motors gear 1 with ctrlrange +-200 (:44-46); timestep 0.02
HOPPER = PlanarModel(
    bodies=[
        ("torso", None, (0.0, 0.0), [(0.0, 0.0, 0.0, 0.2, 0.05, 0.9)]),
        ("thigh", "torso", (0.0, 1.05 - 1.25), [(0.0, (1.05 + 0.6) / 2 - 1.05, 0.0, (1.05 - 0.6) / 2, 0.05, 0.9)]),
        ("leg", "thigh", (0.0, 0.6 - 1.05), [(0.0, (0.6 + 0.1) / 2 - 0.6, 0.0, (0.6 - 0.1) / 2, 0.04, 0.9)]),
        ("foot", "leg", (0.0, 0.1 - 0.6), [((0.26 - 0.13) / 2, 0.0, np.pi / 2, (0.26 + 0.13) / 2, 0.06, 2.0)]),
    ],
    joints={n: (lo * _D, hi * _D, 0.0, 1.0, 1.0, 1.0, -1.0) for n, lo, hi in
            [("thigh", -150, 0), ("leg", -150, 0), ("foot", -45, 45)]},
    root_height=1.25, dt=0.02, substeps=8)
HOPPER_CTRL = np.array([200.0, 200.0, 200.0])


def hopper_reset(draws):
    z = np.asarray(draws, dtype=np.float64)
    qpos = 0.01 * z[:6]
    qpos[0] += 1.25
    return qpos, 0.1 * z[6:]


def hopper_observe(qpos, qvel):
    com, _ = HOPPER.com_and_vel(qpos, qvel)
    qf = HOPPER.constraint_qfrc(qpos, qvel)
    return np.concatenate([qpos[0:1], qpos[2:], np.clip(qvel, -10, 10), np.clip(qf, -10, 10), [com[0], 0.0, com[1]]])


def hopper_step(qpos, qvel, action, normalize=True):
    """One HopperEnv.step (behind NormalizedEnv when ``normalize``) in MuJoCo's conventions."""
    a = np.asarray(action, dtype=np.float64)
    lb, ub = -HOPPER_CTRL, HOPPER_CTRL
    if normalize:
        a = np.clip(lb + (a + 1.0) * 0.5 * (ub - lb), lb, ub)
    a = np.clip(a, lb, ub)
    q, qd = HOPPER.advance(qpos, qvel, a)
    _, comvel = HOPPER.com_and_vel(q, qd)
    reward = comvel[0] + 1.0 - 0.5 * 0.01 * np.sum(np.square(a / ((ub - lb) * 0.5)))
    state = np.concatenate([q, qd])
    notdone = np.isfinite(state).all() and (np.abs(state[3:]) < 100).all() and state[0] > 0.7 and abs(state[2]) < 0.2
    return q, qd, hopper_observe(q, qd), reward, not notdone


def hopper_to_engine_state(qpos, qvel):
    sgn = np.concatenate([[1.0, 1.0, 1.0], HOPPER.sign])
    return np.concatenate([qpos * sgn, qvel * sgn])
