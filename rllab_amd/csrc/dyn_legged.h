// dyn_legged.h -- the model half shared by the legged MuJoCo-style envs (Walker2D, Hopper): rigid-body constants
// from a generated <name>_constants.h namespace, penalty joint limits, gravity and capsule-floor contacts.
//
// condim-3 capsule / plane contacts are a spring-damper penalty model on the capsules' end spheres (per-geom radius
// and friction), the same model as dyn_cheetah.h.  Plane coordinates (P1, P2) = (z, x); gravity along -P1.
#pragma once
#include "dyn_planar.h"

// accessor struct over the constexpr arrays of namespace NS (namespaces cannot be template arguments)
#define RL_LEGGED_CONSTANTS(NAME, NS)                                                      \
    struct NAME {                                                                          \
        static constexpr int NB = NS::NB, NC = NS::NC;                                     \
        RL_HD static constexpr int parent(int i) { return NS::PARENT[i]; }                 \
        RL_HD static constexpr double jx(int i) { return NS::JX[i]; }                      \
        RL_HD static constexpr double jy(int i) { return NS::JY[i]; }                      \
        RL_HD static constexpr double cx(int i) { return NS::CX[i]; }                      \
        RL_HD static constexpr double cy(int i) { return NS::CY[i]; }                      \
        RL_HD static constexpr double mass(int i) { return NS::MASS[i]; }                  \
        RL_HD static constexpr double inertia(int i) { return NS::INERTIA[i]; }            \
        RL_HD static constexpr double armature(int i) { return NS::ARMATURE[i]; }          \
        RL_HD static constexpr double damping(int i) { return NS::DAMPING[i]; }            \
        RL_HD static constexpr double stiffness(int i) { return NS::STIFFNESS[i]; }        \
        RL_HD static constexpr double lo(int i) { return NS::LO[i]; }                      \
        RL_HD static constexpr double hi(int i) { return NS::HI[i]; }                      \
        RL_HD static constexpr double gear(int i) { return NS::GEAR[i]; }                  \
        RL_HD static constexpr double sign(int i) { return NS::SIGN[i]; }                  \
        RL_HD static constexpr int cbody(int c) { return NS::CBODY[c]; }                   \
        RL_HD static constexpr double cpx(int c) { return NS::CPX[c]; }                    \
        RL_HD static constexpr double cpy(int c) { return NS::CPY[c]; }                    \
        RL_HD static constexpr double crad(int c) { return NS::CRADS[c]; }                 \
        RL_HD static constexpr double cmu(int c) { return NS::CMU[c]; }                    \
    }

namespace rl {

template <class K>
struct LeggedModel {
    static constexpr int NB = K::NB;
    RL_HD static constexpr int parent(int i) { return K::parent(i); }
    RL_HD static constexpr double jx(int i) { return K::jx(i); }
    RL_HD static constexpr double jy(int i) { return K::jy(i); }
    RL_HD static constexpr double cx(int i) { return K::cx(i); }
    RL_HD static constexpr double cy(int i) { return K::cy(i); }
    RL_HD static constexpr double mass(int i) { return K::mass(i); }
    RL_HD static constexpr double inertia(int i) { return K::inertia(i); }
    RL_HD static constexpr double armature(int i) { return K::armature(i); }
    RL_HD static constexpr double damping(int i) { return K::damping(i); }
    RL_HD static constexpr double stiffness(int i) { return K::stiffness(i); }
    RL_HD static constexpr bool limited(int i) { return i >= 1; }
    RL_HD static constexpr double lo(int i) { return K::lo(i); }
    RL_HD static constexpr double hi(int i) { return K::hi(i); }
    RL_HD static constexpr double limit_k() { return 2.0e3; }
    RL_HD static constexpr double limit_b() { return 15.0; }
    RL_HD static constexpr double gx() { return -9.81; }  // gravity along -z = -P1
    RL_HD static constexpr double gy() { return 0.0; }

    static constexpr double CONTACT_K = 2.0e4;   // N/m per end sphere
    static constexpr double CONTACT_B = 3.0e2;   // N s/m while penetrating
    static constexpr double FRICTION_C = 3.0e2;  // N s/m tangential, clamped to mu * f_n

    // contact table (dyn_two_legs.h)
    static constexpr int NC = K::NC;
    RL_HD static constexpr int cbody(int c) { return K::cbody(c); }
    RL_HD static constexpr double cpx(int c) { return K::cpx(c); }
    RL_HD static constexpr double cpy(int c) { return K::cpy(c); }
    RL_HD static constexpr double crad(int c) { return K::crad(c); }
    RL_HD static constexpr double cmu(int c) { return K::cmu(c); }

    // capsule end spheres against the floor z = 0
    template <typename R>
    RL_HD static void external(const R* q, const PlanarKin<R, NB>& k, R* fx, R* fy, R* tz) {
        RL_UNROLL
        for (int i = 0; i < NB; ++i) { fx[i] = (R)0; fy[i] = (R)0; tz[i] = (R)0; }
        RL_UNROLL
        for (int c = 0; c < K::NC; ++c) {
            const int b = K::cbody(c);
            const R lx = (R)K::cpx(c), ly = (R)K::cpy(c), rad = (R)K::crad(c);
            const R rx = k.cs[b] * lx - k.sn[b] * ly;   // sphere centre relative to the body anchor
            const R ry = k.sn[b] * lx + k.cs[b] * ly;
            const R depth = rad - (q[0] + k.ax[b] + rx);
            if (depth > (R)0) {
                const R vn = k.vax[b] - k.om[b] * ry;   // velocity of the sphere centre
                const R vt = k.vay[b] + k.om[b] * rx;
                R fn = (R)CONTACT_K * depth - (R)CONTACT_B * vn;
                fn = rl_max(fn, (R)0);
                const R mu = (R)K::cmu(c);
                const R ft = -rl_clamp((R)FRICTION_C * vt, -mu * fn, mu * fn);
                // applied at the lowest point of the sphere; lever arm from the body COM
                const R ax_ = (k.ax[b] + rx - rad) - k.px[b];
                const R ay_ = (k.ay[b] + ry) - k.py[b];
                fx[b] = fx[b] + fn;
                fy[b] = fy[b] + ft;
                tz[b] = tz[b] + (ax_ * ft - ay_ * fn);
            }
        }
    }
};

}  // namespace rl
