"""Position correction of Box2D 2.3's revolute and prismatic joints, restated in numpy float64 from the PUBLISHED
algorithm -- TEST INFRASTRUCTURE, never imported by the product, and written without looking at csrc/dyn_cartpole.h so
that it can referee it.

The reference's CartpoleEnv runs on pybox2d (Box2D 2.3.x; third party, absent from /root/reference and from this image).
What is restated (Box2D 2.3.1 sources as published by Erin Catto; function names are Box2D's):
  * b2RevoluteJoint::SolvePositionConstraints  (Dynamics/Joints/b2RevoluteJoint.cpp, no limit, no motor):
        rA = R(aA) (localAnchorA - localCenterA),  rB likewise,  C = cB + rB - cA - rA
        K  = [[mA + mB + iA rA.y^2 + iB rB.y^2,  -iA rA.x rA.y - iB rB.x rB.y], [sym,  mA + mB + iA rA.x^2 + iB rB.x^2]]
        impulse = -K^-1 C;   cA -= mA impulse, aA -= iA (rA x impulse);   cB += mB impulse, aB += iB (rB x impulse)
    (the FULL error is removed per visit: no Baumgarte factor, no clamp, for a point-to-point constraint)
  * b2PrismaticJoint::SolvePositionConstraints (Dynamics/Joints/b2PrismaticJoint.cpp, no limit):
        d = cB + rB - cA - rA;  perp = R(aA) localYAxisA;  s1 = (d + rA) x perp,  s2 = rB x perp
        C1 = (perp . d,  aB - aA - referenceAngle);  K = [[mA + mB + iA s1^2 + iB s2^2, iA s1 + iB s2], [sym, iA + iB]]
        (imp.x, imp.y) = K^-1 (-C1);  P = imp.x perp;  LA = imp.x s1 + imp.y,  LB = imp.x s2 + imp.y
        cA -= mA P, aA -= iA LA;  cB += mB P, aB += iB LB
  * b2Island::Solve's position phase: up to `positionIterations` sweeps over the island's joints, stopping early when
    every joint reports an error within b2_linearSlop (0.005 m) / b2_angularSlop (2 degrees).
m, i are INVERSE masses / inertias (Box2D's m_invMassA ...); a static body has both zero.

Used by tests/test_oracle_physics.py to decide what `CartpoleEnv.reset` (cartpole_env.py:28-43: the cart is moved by up
to 0.12 m, the pole body is not) does to the pole on the first step.  "Parity unpinned" against pybox2d itself."""
import numpy as np

LINEAR_SLOP = 0.005
ANGULAR_SLOP = 2.0 / 180.0 * np.pi


def rot(a, v):
    c, s = np.cos(a), np.sin(a)
    return np.array([c * v[0] - s * v[1], s * v[0] + c * v[1]])


def cross(a, b):
    return a[0] * b[1] - a[1] * b[0]


class Body(object):
    """c = world position of the centre of mass, a = angle, local_center = centre of mass in body coordinates."""

    def __init__(self, c, a, inv_mass, inv_inertia, local_center=(0.0, 0.0)):
        self.c = np.array(c, dtype=np.float64)
        self.a = float(a)
        self.m, self.i = float(inv_mass), float(inv_inertia)
        self.local_center = np.array(local_center, dtype=np.float64)


def revolute_solve_position(A, B, local_anchor_a, local_anchor_b):
    rA = rot(A.a, np.asarray(local_anchor_a) - A.local_center)
    rB = rot(B.a, np.asarray(local_anchor_b) - B.local_center)
    C = B.c + rB - A.c - rA
    err = float(np.hypot(*C))
    K = np.array([[A.m + B.m + A.i * rA[1] ** 2 + B.i * rB[1] ** 2, -A.i * rA[0] * rA[1] - B.i * rB[0] * rB[1]],
                  [-A.i * rA[0] * rA[1] - B.i * rB[0] * rB[1], A.m + B.m + A.i * rA[0] ** 2 + B.i * rB[0] ** 2]])
    imp = -np.linalg.solve(K, C)
    A.c -= A.m * imp
    A.a -= A.i * cross(rA, imp)
    B.c += B.m * imp
    B.a += B.i * cross(rB, imp)
    return err <= LINEAR_SLOP


def prismatic_solve_position(A, B, local_anchor_a, local_anchor_b, local_y_axis_a=(0.0, 1.0), reference_angle=0.0):
    rA = rot(A.a, np.asarray(local_anchor_a) - A.local_center)
    rB = rot(B.a, np.asarray(local_anchor_b) - B.local_center)
    d = B.c + rB - A.c - rA
    perp = rot(A.a, np.asarray(local_y_axis_a, dtype=np.float64))
    s1, s2 = cross(d + rA, perp), cross(rB, perp)
    C1 = np.array([perp.dot(d), B.a - A.a - reference_angle])
    k22 = A.i + B.i
    K = np.array([[A.m + B.m + A.i * s1 * s1 + B.i * s2 * s2, A.i * s1 + B.i * s2],
                  [A.i * s1 + B.i * s2, k22 if k22 != 0.0 else 1.0]])
    imp = np.linalg.solve(K, -C1)
    P = imp[0] * perp
    A.c -= A.m * P
    A.a -= A.i * (imp[0] * s1 + imp[1])
    B.c += B.m * P
    B.a += B.i * (imp[0] * s2 + imp[1])
    return abs(C1[0]) <= LINEAR_SLOP and abs(C1[1]) <= ANGULAR_SLOP


def cartpole_first_position_solve(cart_x, pole_angle=0.0, position_iterations=3):
    """The world of rllab/envs/box2d/models/cartpole.xml.mako right after CartpoleEnv.reset put the cart at ``cart_x``
    and left the pole body where the XML has it: density-1 boxes (cart 4/sqrt(12) x 3/sqrt(12) -> mass 1, inertia
    25/144; pole 0.1 x 1 hinged at its lower end -> mass 0.1, inertia about the centre 0.1 (0.01 + 1) / 12), the revolute
    joint anchored at (0, cart_h) of the XML pose, the prismatic joint track -> cart along x.  Joint order of the island:
    revolute, then prismatic (the depth-first island build starts at the pole's joint).  Returns (pole angle, cart x,
    hinge gap) after the position phase of ONE b2World::Step with every velocity zero."""
    cart_w, cart_h = 4.0 / np.sqrt(12.0), 3.0 / np.sqrt(12.0)
    cart = Body((cart_x, cart_h / 2), 0.0, 1.0 / (cart_w * cart_h), 1.0 / (cart_w * cart_h * (cart_w ** 2 + cart_h ** 2) / 12.0))
    m_pole = 0.1 * 1.0
    # pole: body origin at the hinge (0, cart_h), fixture from y = 0 to y = 1 -> local centre (0, 0.5)
    pole = Body(np.array([0.0, cart_h]) + rot(pole_angle, (0.0, 0.5)), pole_angle, 1.0 / m_pole,
                1.0 / (m_pole * (0.1 ** 2 + 1.0 ** 2) / 12.0), local_center=(0.0, 0.5))
    track = Body((0.0, cart_h / 2), 0.0, 0.0, 0.0)
    # anchors in body coordinates, from the XML pose: revolute at world (0, cart_h); prismatic at bodyB's origin
    rev = dict(local_anchor_a=(0.0, cart_h / 2), local_anchor_b=(0.0, 0.0))
    pri = dict(local_anchor_a=(0.0, 0.0), local_anchor_b=(0.0, 0.0))
    for _ in range(position_iterations):
        ok_r = revolute_solve_position(cart, pole, **rev)
        ok_p = prismatic_solve_position(track, cart, **pri)
        if ok_r and ok_p:
            break
    hinge_cart = cart.c + rot(cart.a, np.array(rev["local_anchor_a"]))
    hinge_pole = pole.c + rot(pole.a, -pole.local_center)
    return pole.a, cart.c[0], float(np.hypot(*(hinge_pole - hinge_cart)))
