#!/usr/bin/env python
"""Derive the rigid-body constants of the planar MuJoCo-style models from the numbers of their MJCF files
(reference: vendor/mujoco_models/half_cheetah.xml, walker2d.xml, hopper.xml) and write <name>_constants.h.

Plane mapping: the models move in MuJoCo's x-z plane.  The planar tree of dyn_planar.h uses CCW-positive
angles, so plane coordinates are (P1, P2) = (z, x): a +y rotation takes z towards x, i.e. P1 towards
P2 = CCW.  Gravity is -9.81 along P1.  Hinges declared about -y (walker2d.xml) therefore carry SIGN = -1:
MuJoCo's joint coordinate, range and motor torque are the negatives of the tree's.

Each body: name, parent index, hinge anchor in the PARENT frame (x, z), geoms.  A geom is a capsule
(centre x, centre z, axis angle about +y | "x" for an axis along +x, half length, radius, friction) in the
body frame, whose origin is the body's hinge (for coordinate="global" files the offsets below are the
differences of the file's absolute positions).
"""
import math
import os

HALF_CHEETAH = dict(
    name="cheetah", total_mass=14.0, density=1000.0,
    # half_cheetah.xml:36-93 (coordinate="local", every capsule radius 0.046, friction 0.4)
    bodies=[
        ("torso", -1, (0.0, 0.0), [(0.0, 0.0, "x", 0.5, 0.046, 0.4), (0.6, 0.1, 0.87, 0.15, 0.046, 0.4)]),
        ("bthigh", 0, (-0.5, 0.0), [(0.1, -0.13, -3.8, 0.145, 0.046, 0.4)]),
        ("bshin", 1, (0.16, -0.25), [(-0.14, -0.07, -2.03, 0.15, 0.046, 0.4)]),
        ("bfoot", 2, (-0.28, -0.14), [(0.03, -0.097, -0.27, 0.094, 0.046, 0.4)]),
        ("fthigh", 0, (0.5, 0.0), [(-0.07, -0.12, 0.52, 0.133, 0.046, 0.4)]),
        ("fshin", 4, (-0.14, -0.24), [(0.065, -0.09, -0.6, 0.106, 0.046, 0.4)]),
        ("ffoot", 5, (0.13, -0.18), [(0.045, -0.07, -0.6, 0.07, 0.046, 0.4)]),
    ],
    # stiffness, damping, lo, hi, gear, armature, sign
    joints={
        "bthigh": (240, 6, -0.52, 1.05, 120, 0.1, 1), "bshin": (180, 4.5, -0.785, 0.785, 90, 0.1, 1),
        "bfoot": (120, 3, -0.4, 0.785, 60, 0.1, 1), "fthigh": (180, 4.5, -1.0, 0.7, 120, 0.1, 1),
        "fshin": (120, 3, -1.2, 0.87, 60, 0.1, 1), "ffoot": (60, 1.5, -0.5, 0.5, 30, 0.1, 1),
    },
    legacy_layout=True,     # keep the committed cheetah_constants.h byte for byte
)

_D = math.pi / 180.0
WALKER2D = dict(
    name="walker", total_mass=None, density=1000.0,
    # walker2d.xml:3-7,19-50 (coordinate="global": torso origin (0, 1.25); hinges thigh (0, 1.05), leg (0, 0.6),
    # foot (0, 0.1); capsules torso (0,1.45)-(0,1.05) r .05, thigh (0,1.05)-(0,.6) r .05, leg (0,.6)-(0,.1) r .04,
    # foot (0,.1)-(.2,.1) r .06; friction .9, left foot 1.9)
    bodies=[
        ("torso", -1, (0.0, 0.0), [(0.0, 0.0, 0.0, 0.2, 0.05, 0.9)]),
        ("thigh", 0, (0.0, -0.2), [(0.0, -0.225, 0.0, 0.225, 0.05, 0.9)]),
        ("leg", 1, (0.0, -0.45), [(0.0, -0.25, 0.0, 0.25, 0.04, 0.9)]),
        ("foot", 2, (0.0, -0.5), [(0.1, 0.0, "x", 0.1, 0.06, 0.9)]),
        ("thigh_left", 0, (0.0, -0.2), [(0.0, -0.225, 0.0, 0.225, 0.05, 0.9)]),
        ("leg_left", 4, (0.0, -0.45), [(0.0, -0.25, 0.0, 0.25, 0.04, 0.9)]),
        ("foot_left", 5, (0.0, -0.5), [(0.1, 0.0, "x", 0.1, 0.06, 1.9)]),
    ],
    # hinges about -y: range -150..0 / -45..45 degrees and ctrlrange +-150 / +-100 are MuJoCo's; the tree's
    # coordinate is the negative (SIGN = -1), so lo/hi below are already mapped: [-hi_mj, -lo_mj]
    joints={
        "thigh": (0, 0.1, 0.0, 150 * _D, 150, 0.01, -1), "leg": (0, 0.1, 0.0, 150 * _D, 100, 0.01, -1),
        "foot": (0, 0.1, -45 * _D, 45 * _D, 100, 0.01, -1),
        "thigh_left": (0, 0.1, 0.0, 150 * _D, 150, 0.01, -1), "leg_left": (0, 0.1, 0.0, 150 * _D, 100, 0.01, -1),
        "foot_left": (0, 0.1, -45 * _D, 45 * _D, 100, 0.01, -1),
    },
    legacy_layout=False,
)

HOPPER = dict(
    name="hopper", total_mass=None, density=1000.0, xml="hopper.xml",
    # hopper.xml:3-7,21-40 (coordinate="global": torso origin (0, 1.25); hinges thigh (0, 1.05), leg (0, 0.6),
    # foot (0, 0.1); capsules torso (0,1.45)-(0,1.05) r .05, thigh (0,1.05)-(0,.6) r .05, leg (0,.6)-(0,.1) r .04,
    # foot (-.13,.1)-(.26,.1) r .06; friction .9, foot 2.0); joints: damping 1, armature 1, hinges about -y
    bodies=[
        ("torso", -1, (0.0, 0.0), [(0.0, 0.0, 0.0, 0.2, 0.05, 0.9)]),
        ("thigh", 0, (0.0, -0.2), [(0.0, -0.225, 0.0, 0.225, 0.05, 0.9)]),
        ("leg", 1, (0.0, -0.45), [(0.0, -0.25, 0.0, 0.25, 0.04, 0.9)]),
        ("foot", 2, (0.0, -0.5), [(0.065, 0.0, "x", 0.195, 0.06, 2.0)]),
    ],
    # ranges -150..0 / -150..0 / -45..45 degrees and ctrlrange +-200 are MuJoCo's; the tree's coordinate is the
    # negative (SIGN = -1), so lo/hi below are already mapped: [-hi_mj, -lo_mj]
    joints={
        "thigh": (0, 1.0, 0.0, 150 * _D, 200, 1.0, -1), "leg": (0, 1.0, 0.0, 150 * _D, 200, 1.0, -1),
        "foot": (0, 1.0, -45 * _D, 45 * _D, 200, 1.0, -1),
    },
    legacy_layout=False,
)


def capsule(half_len, r, rho):
    L = 2 * half_len
    m_c = rho * math.pi * r * r * L
    m_s = rho * 4.0 / 3.0 * math.pi * r ** 3
    i_perp = m_c * (L * L / 12 + r * r / 4) + m_s * (83.0 / 320 * r * r + (L / 2 + 3 * r / 8) ** 2)
    return m_c + m_s, i_perp


def derive(model):
    out, masses = [], []
    for name, parent, anchor, geoms in model["bodies"]:
        m_tot, mx, mz, parts, ends = 0.0, 0.0, 0.0, [], []
        for cx, cz, ang, hl, rad, fric in geoms:
            m, i = capsule(hl, rad, model["density"])
            ax = (1.0, 0.0) if ang == "x" else (math.sin(ang), math.cos(ang))   # capsule axis (x, z)
            parts.append((m, i, cx, cz))
            ends += [(cx + hl * ax[0], cz + hl * ax[1], rad, fric), (cx - hl * ax[0], cz - hl * ax[1], rad, fric)]
            m_tot += m
            mx += m * cx
            mz += m * cz
        comx, comz = mx / m_tot, mz / m_tot
        inertia = sum(i + m * ((cx - comx) ** 2 + (cz - comz) ** 2) for m, i, cx, cz in parts)
        out.append(dict(name=name, parent=parent, anchor=anchor, mass=m_tot, com=(comx, comz), inertia=inertia,
                        ends=ends))
        masses.append(m_tot)
    if model["total_mass"] is not None:          # compiler settotalmass
        scale = model["total_mass"] / sum(masses)
        for b in out:
            b["mass"] *= scale
            b["inertia"] *= scale
    return out


def emit(model):
    out = derive(model)
    nb = len(out)
    ns = model["name"]
    joints = model["joints"]
    names = [b["name"] for b in out]

    def arr(fn):
        return ", ".join("%.17g" % fn(b) for b in out)

    def j(k, d):
        return ", ".join("%.17g" % (joints[n][k] if n in joints else d) for n in names)
    src = "gen_cheetah_constants.py" if model["legacy_layout"] else "gen_planar_constants.py"
    xml = model.get("xml") or ("half_cheetah.xml" if ns == "cheetah" else "walker2d.xml")
    lines = [
        "// GENERATED by %s from the numbers of vendor/mujoco_models/%s." % (src, xml),
        "// Plane coordinates (P1, P2) = (z, x): +y hinge rotation = CCW.  Do not edit by hand.",
        "#pragma once",
        "namespace rl { namespace %s {" % ns,
        "constexpr int NB = %d;" % nb,
        "constexpr int PARENT[NB] = {%s};" % ", ".join(str(b["parent"]) for b in out),
        "constexpr double JX[NB] = {%s};  // anchor in parent frame, P1 (= z)" % arr(lambda b: b["anchor"][1]),
        "constexpr double JY[NB] = {%s};  // P2 (= x)" % arr(lambda b: b["anchor"][0]),
        "constexpr double CX[NB] = {%s};  // body COM in own frame, P1" % arr(lambda b: b["com"][1]),
        "constexpr double CY[NB] = {%s};  // P2" % arr(lambda b: b["com"][0]),
        "constexpr double MASS[NB] = {%s};" % arr(lambda b: b["mass"]),
        "constexpr double INERTIA[NB] = {%s};" % arr(lambda b: b["inertia"]),
    ]
    if model["legacy_layout"]:
        lines.append("constexpr double ARMATURE[NB] = {%s};" % ", ".join("0.1" if n in joints else "0" for n in names))
    else:
        lines.append("constexpr double ARMATURE[NB] = {%s};" % j(5, 0))
    lines += [
        "constexpr double STIFFNESS[NB] = {%s};" % j(0, 0),
        "constexpr double DAMPING[NB] = {%s};" % j(1, 0),
        "constexpr double LO[NB] = {%s};" % j(2, 0),
        "constexpr double HI[NB] = {%s};" % j(3, 0),
        "constexpr double GEAR[NB] = {%s};" % j(4, 0),
    ]
    if not model["legacy_layout"]:
        lines.append("constexpr double SIGN[NB] = {%s};  // MuJoCo joint coordinate = SIGN * tree coordinate" % j(6, 1))
    pts = [(i, e[1], e[0], e[2], e[3]) for i, b in enumerate(out) for e in b["ends"]]
    lines += [
        "constexpr int NC = %d;  // capsule end spheres that can touch the floor" % len(pts),
        "constexpr int CBODY[NC] = {%s};" % ", ".join(str(p[0]) for p in pts),
        "constexpr double CPX[NC] = {%s};" % ", ".join("%.17g" % p[1] for p in pts),
        "constexpr double CPY[NC] = {%s};" % ", ".join("%.17g" % p[2] for p in pts),
    ]
    if model["legacy_layout"]:
        lines.append("constexpr double CRAD = %.17g;" % pts[0][3])
    else:
        lines += ["constexpr double CRADS[NC] = {%s};" % ", ".join("%.17g" % p[3] for p in pts),
                  "constexpr double CMU[NC] = {%s};" % ", ".join("%.17g" % p[4] for p in pts)]
    lines.append("}}  // namespace rl::%s" % ns)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "%s_constants.h" % ns)
    text = "\n".join(lines) + "\n"
    # (an unchanged header is left alone: rewriting it would make every translation unit that includes it stale for
    #  __graft_entry__.build() -- tests/test_oracle_physics.py runs this generator on every CPU test run)
    if not (os.path.exists(path) and open(path).read() == text):
        open(path, "w").write(text)
    return path


def main():
    for model in (HALF_CHEETAH, WALKER2D, HOPPER):
        print(emit(model))


if __name__ == "__main__":
    main()
