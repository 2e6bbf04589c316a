// dyn_two_legs.h -- one sub-step of a planar torso carrying two three-link legs (HalfCheetah, Walker2D), written ONCE
// over a value type V and instantiated two ways with identical arithmetic per leg:
//
//   V = V2<R> (rl_math.h): both legs side by side in the two components of every value.  One env per lane / host
//                 thread -- the host oracle build, the per-step VecEnv kernels and the env-per-lane rollouts.  On gfx950
//                 every operation is one packed-f32 instruction for the two legs.
//   V = R         one leg per lane, the other leg in the neighbouring lane (x.other(v) = its value of v).  Used by
//                 the lane-group rollout, where a lone wavefront per SIMD pays per issued instruction: each lane walks a
//                 four-body chain (torso + its leg) instead of seven bodies.
//
// The legs meet only in the torso: every whole-tree quantity is  torso part + (own leg's part + other leg's part) --
// a sum that is bitwise symmetric in the two legs, so both lanes (both components) hold the same root block, the same
// root accelerations and the same torso state, and the two instantiations agree bit for bit.
//
// Formulation: composite rigid bodies in joint coordinates (dyn_planar.h forward_dynamics_crb) on the local chain
// 0 = torso, 1..3 = own leg; the joint-space matrix is block-arrow,
//     [ R    G_b   G_f ] [ u    ]   [ r_u ]      u = (a_P1, a_P2, torso angular acceleration),
//     [ G_b' K_b   0   ] [ th_b ] = [ r_b ]      K_L = the leg's 3x3 hinge block, G_L its coupling to the root,
//     [ G_f' 0     K_f ] [ th_f ]   [ r_f ]
// solved leaves first: each leg eliminates its own block in closed form (one reciprocal) and hands the root its Schur
// complement  G_L K_L^-1 G_L' (6 numbers) and  G_L K_L^-1 r_L (3 numbers); the 3x3 root system is solved by both.
// Exchanges per sub-step: 3 (subtree wrench) + 3 (composite first moment / inertia) + 9 (Schur) = 15 values.
//
// Model traits: dyn_planar.h's, plus the contact table  NC, cbody(c), cpx(c), cpy(c), crad(c), cmu(c)  ordered as
// [torso spheres][two per leg body, back leg][two per leg body, front leg], and CONTACT_K / CONTACT_B / FRICTION_C.
#pragma once
#include "dyn_planar.h"

namespace rl {

// ---- value-type helpers: the scalar forms live in rl_math.h; these are their two-component twins ------------------
RL_HD V2<float> rl_abs(V2<float> v) { return V2<float>{rl_abs(v.x), rl_abs(v.y)}; }
RL_HD V2<double> rl_abs(V2<double> v) { return V2<double>{rl_abs(v.x), rl_abs(v.y)}; }
RL_HD V2<float> rl_max(V2<float> a, V2<float> b) { return V2<float>{rl_max(a.x, b.x), rl_max(a.y, b.y)}; }
RL_HD V2<double> rl_max(V2<double> a, V2<double> b) { return V2<double>{rl_max(a.x, b.x), rl_max(a.y, b.y)}; }
RL_HD V2<float> rl_min(V2<float> a, V2<float> b) { return V2<float>{rl_min(a.x, b.x), rl_min(a.y, b.y)}; }
RL_HD V2<double> rl_min(V2<double> a, V2<double> b) { return V2<double>{rl_min(a.x, b.x), rl_min(a.y, b.y)}; }
RL_HD V2<float> rl_recip_normal(V2<float> d) { return V2<float>{rl_recip_normal(d.x), rl_recip_normal(d.y)}; }
RL_HD V2<double> rl_recip_normal(V2<double> d) { return V2<double>{rl_recip_normal(d.x), rl_recip_normal(d.y)}; }
// c > 0 ? a : b   and   c != 0 ? a : b,   per component
RL_HD float rl_if_pos(float c, float a, float b) { return c > 0.0f ? a : b; }
RL_HD double rl_if_pos(double c, double a, double b) { return c > 0.0 ? a : b; }
RL_HD V2<float> rl_if_pos(V2<float> c, V2<float> a, V2<float> b) { return V2<float>{c.x > 0.0f ? a.x : b.x, c.y > 0.0f ? a.y : b.y}; }
RL_HD V2<double> rl_if_pos(V2<double> c, V2<double> a, V2<double> b) { return V2<double>{c.x > 0.0 ? a.x : b.x, c.y > 0.0 ? a.y : b.y}; }
RL_HD float rl_if_nonzero(float c, float a, float b) { return c != 0.0f ? a : b; }
RL_HD double rl_if_nonzero(double c, double a, double b) { return c != 0.0 ? a : b; }
RL_HD V2<float> rl_if_nonzero(V2<float> c, V2<float> a, V2<float> b) { return V2<float>{c.x != 0.0f ? a.x : b.x, c.y != 0.0f ? a.y : b.y}; }
RL_HD V2<double> rl_if_nonzero(V2<double> c, V2<double> a, V2<double> b) { return V2<double>{c.x != 0.0 ? a.x : b.x, c.y != 0.0 ? a.y : b.y}; }
RL_HD bool rl_any_pos(float c) { return c > 0.0f; }
RL_HD bool rl_any_pos(double c) { return c > 0.0; }
RL_HD bool rl_any_pos(V2<float> c) { return c.x > 0.0f || c.y > 0.0f; }
RL_HD bool rl_any_pos(V2<double> c) { return c.x > 0.0 || c.y > 0.0; }
// a scalar as a value of type V
template <typename V> struct Lanes;
template <> struct Lanes<float> { RL_HD static float splat(float r) { return r; } };
template <> struct Lanes<double> { RL_HD static double splat(double r) { return r; } };
template <> struct Lanes<V2<float>> { RL_HD static V2<float> splat(float r) { return V2<float>{r, r}; } };
template <> struct Lanes<V2<double>> { RL_HD static V2<double> splat(double r) { return V2<double>{r, r}; } };
// rl_rotate_small (rl_math.h) over a value type: the series is chosen by the SCALAR type R
template <typename R, typename V>
RL_HD void rl_rotate_small_v(V& sn, V& cs, V d) {
    const V d2 = d * d;
    V sd, cd;
    if constexpr (sizeof(R) == 4) {
        sd = d * ((R)1 + d2 * ((R)(-1.0 / 6) + d2 * (R)(1.0 / 120)));
        cd = (R)1 + d2 * ((R)-0.5 + d2 * ((R)(1.0 / 24) + d2 * (R)(-1.0 / 720)));
    } else {
        sd = d * ((R)1 + d2 * ((R)(-1.0 / 6) + d2 * ((R)(1.0 / 120) + d2 * ((R)(-1.0 / 5040) +
             d2 * ((R)(1.0 / 362880) + d2 * (R)(-1.0 / 39916800))))));
        cd = (R)1 + d2 * ((R)-0.5 + d2 * ((R)(1.0 / 24) + d2 * ((R)(-1.0 / 720) + d2 * ((R)(1.0 / 40320) +
             d2 * ((R)(-1.0 / 3628800) + d2 * (R)(1.0 / 479001600))))));
    }
    const V s = sn * cd + cs * sd;
    const V c = cs * cd - sn * sd;
    sn = s;
    cs = c;
}
// both legs in one value: the other leg's value is the swapped pair
struct BothLegs {
    template <typename R> RL_HD V2<R> other(V2<R> v) const { return v.yx; }
};

template <class Mdl>
struct TwoLegs {
    static constexpr int NB = Mdl::NB;
    static_assert(NB == 7, "torso + two three-link legs");
    static constexpr bool topology_ok() {
        constexpr int want[7] = {-1, 0, 1, 2, 0, 4, 5};
        for (int i = 1; i < 7; ++i)
            if (Mdl::parent(i) != want[i]) return false;
        return true;
    }
    static_assert(topology_ok(), "bodies 1-2-3 and 4-5-6 are chains off body 0");
    static constexpr int NCT = Mdl::NC - 12;          // torso spheres; then two per leg body
    static constexpr bool contacts_ok() {
        if (NCT < 0) return false;
        for (int c = 0; c < NCT; ++c)
            if (Mdl::cbody(c) != 0) return false;
        for (int c = 0; c < 12; ++c)
            if (Mdl::cbody(NCT + c) != 1 + c / 2) return false;
        return true;
    }
    static_assert(contacts_ok(), "contact table: torso spheres, then two per leg body");
    static constexpr double total_mass() {
        double m = 0.0;
        for (int i = 0; i < NB; ++i) m += Mdl::mass(i);
        return m;
    }
    static constexpr double subtree_mass(int i) {      // i = a leg body: itself and the bodies below it in its leg
        const int last = i <= 3 ? 3 : 6;
        double m = 0.0;
        for (int b = i; b <= last; ++b) m += Mdl::mass(b);
        return m;
    }

    // the constants of one leg (local bodies 1..3 -> index 0..2; spheres two per body), as values of type V
    template <typename V>
    struct LegK {
        V jx[3], jy[3], cx[3], cy[3], mass[3], inertia[3], arm[3], stiff[3], damp[3], lo[3], hi[3], mc[3];
        V cpx[6], cpy[6], crad[6], cmu[6];
    };
    // F(body index of the leg's j-th body / sphere index) -> constant, for leg 0 (bodies 1..3) or 1 (bodies 4..6)
    template <typename R>
    RL_HD static LegK<R> leg_constants(int leg) {
        LegK<R> k;
        RL_UNROLL
        for (int j = 0; j < 3; ++j) {
#define RL_LEGSEL(f) (R)(leg == 0 ? Mdl::f(1 + j) : Mdl::f(4 + j))
            k.jx[j] = RL_LEGSEL(jx); k.jy[j] = RL_LEGSEL(jy); k.cx[j] = RL_LEGSEL(cx); k.cy[j] = RL_LEGSEL(cy);
            k.mass[j] = RL_LEGSEL(mass); k.inertia[j] = RL_LEGSEL(inertia); k.arm[j] = RL_LEGSEL(armature);
            k.stiff[j] = RL_LEGSEL(stiffness); k.damp[j] = RL_LEGSEL(damping); k.lo[j] = RL_LEGSEL(lo);
            k.hi[j] = RL_LEGSEL(hi);
#undef RL_LEGSEL
            k.mc[j] = (R)(leg == 0 ? subtree_mass(1 + j) : subtree_mass(4 + j));
        }
        RL_UNROLL
        for (int c = 0; c < 6; ++c) {
            k.cpx[c] = (R)(leg == 0 ? Mdl::cpx(NCT + c) : Mdl::cpx(NCT + 6 + c));
            k.cpy[c] = (R)(leg == 0 ? Mdl::cpy(NCT + c) : Mdl::cpy(NCT + 6 + c));
            k.crad[c] = (R)(leg == 0 ? Mdl::crad(NCT + c) : Mdl::crad(NCT + 6 + c));
            k.cmu[c] = (R)(leg == 0 ? Mdl::cmu(NCT + c) : Mdl::cmu(NCT + 6 + c));
        }
        return k;
    }
    template <typename R>
    RL_HD static LegK<V2<R>> both_leg_constants() {
        const LegK<R> a = leg_constants<R>(0), b = leg_constants<R>(1);
        LegK<V2<R>> k;
        RL_UNROLL
        for (int j = 0; j < 3; ++j) {
#define RL_PAIR(f) k.f[j] = V2<R>{a.f[j], b.f[j]}
            RL_PAIR(jx); RL_PAIR(jy); RL_PAIR(cx); RL_PAIR(cy); RL_PAIR(mass); RL_PAIR(inertia); RL_PAIR(arm);
            RL_PAIR(stiff); RL_PAIR(damp); RL_PAIR(lo); RL_PAIR(hi); RL_PAIR(mc);
#undef RL_PAIR
        }
        RL_UNROLL
        for (int c = 0; c < 6; ++c) {
            k.cpx[c] = V2<R>{a.cpx[c], b.cpx[c]}; k.cpy[c] = V2<R>{a.cpy[c], b.cpy[c]};
            k.crad[c] = V2<R>{a.crad[c], b.crad[c]}; k.cmu[c] = V2<R>{a.cmu[c], b.cmu[c]};
        }
        return k;
    }

    // state of one lane (or, for V = V2, of the env): the torso's coordinates are replicated in every lane / component
    template <typename V>
    struct State {
        V qr[3], qdr[3];     // root: P1, P2, torso angle and their rates
        V q[3], qd[3];       // the leg's hinge angles / rates
        V sn[4], cs[4];      // carried sine / cosine of the absolute angles: torso, then the leg's bodies
    };

    // x = K^-1 r pieces for a symmetric positive definite 3x3 (a, b, c, d, e, f) = (K00, K10, K20, K11, K21, K22)
    template <typename V>
    RL_HD static void spd3_inverse(V a, V b, V c, V d, V e, V f, V* adj) {
        const V A = d * f - e * e, B = c * e - b * f, C = b * e - c * d;
        const V D = a * f - c * c, E = b * c - a * e, F = a * d - b * b;
        const V det = a * A + (b * B + c * C);
        const V inv = rl_recip_normal(det);
        adj[0] = A * inv; adj[1] = B * inv; adj[2] = C * inv; adj[3] = D * inv; adj[4] = E * inv; adj[5] = F * inv;
    }
    template <typename V>
    RL_HD static void spd3_apply(const V* adj, V r0, V r1, V r2, V& x0, V& x1, V& x2) {
        x0 = adj[0] * r0 + (adj[1] * r1 + adj[2] * r2);
        x1 = adj[1] * r0 + (adj[3] * r1 + adj[4] * r2);
        x2 = adj[2] * r0 + (adj[4] * r1 + adj[5] * r2);
    }
    template <typename V>
    RL_HD static V dot3(V a0, V a1, V a2, V b0, V b1, V b2) { return a0 * b0 + (a1 * b1 + a2 * b2); }

    // one capsule-end sphere of local body b against the floor P1 = 0 (the penalty model of dyn_cheetah.h /
    // dyn_legged.h), in two steps: where it is (sphere_pose), what it pushes with (sphere_force).  A component (leg)
    // that does not penetrate adds exact (+)zeros to sums that are never -0, so WHETHER sphere_force runs for a sphere
    // above the floor changes no bit: the sub-step tests the spheres of a body together and skips the body's force
    // block when none of them touches -- one branch per body instead of one per sphere.  A branch on a vector compare
    // costs a lone wavefront ~60 cycles when taken (tools/ubench/branch_cost.hip); ten of them per sub-step were a sixth
    // of the one-env-per-wavefront rollout (profiles/r04_notes.md).  Here the compares also sit far ahead of the branches.
    template <typename R, typename V>
    RL_HD static void sphere_pose(V lx, V ly, V rad, V csb, V snb, V axb, V root_p1, V& rx, V& ry, V& depth) {
        rx = csb * lx - snb * ly;                   // sphere centre relative to the body anchor
        ry = snb * lx + csb * ly;
        depth = rad - ((root_p1 + axb) + rx);
    }
    template <typename R, typename V>
    RL_HD static void sphere_force(V rx, V ry, V depth, V rad, V mu, V axb, V ayb, V pxb, V pyb, V vaxb, V vayb, V omb,
                                   V& fx, V& fy, V& tz) {
        const V vn = vaxb - omb * ry;               // velocity of the sphere centre
        const V vt = vayb + omb * rx;
        const V zero = Lanes<V>::splat((R)0);
        V fn = (R)Mdl::CONTACT_K * depth - (R)Mdl::CONTACT_B * vn;
        fn = rl_max(fn, zero);
        const V lim = mu * fn;
        const V ft = -rl_max(-lim, rl_min((R)Mdl::FRICTION_C * vt, lim));
        // applied at the lowest point of the sphere; lever arm from the body COM
        const V ax_ = ((axb + rx) - rad) - pxb;
        const V ay_ = (ayb + ry) - pyb;
        const V mom = ax_ * ft - ay_ * fn;
        fx = fx + rl_if_pos(depth, fn, zero);
        fy = fy + rl_if_pos(depth, ft, zero);
        tz = tz + rl_if_pos(depth, mom, zero);
    }

    // positions / velocities along the chain torso -> leg, relative to the root origin (dyn_planar.h kinematics_sc)
    template <typename V>
    struct Kin {
        V om[4], ax[4], ay[4], lx[4], ly[4], ex[4], ey[4], px[4], py[4], vax[4], vay[4], vpx[4], vpy[4];
    };
    template <typename R, typename V>
    RL_HD static void kinematics(const LegK<V>& c, const State<V>& s, Kin<V>& k) {
        const V zero = Lanes<V>::splat((R)0);
        k.om[0] = s.qdr[2];
        RL_UNROLL
        for (int i = 1; i < 4; ++i) k.om[i] = k.om[i - 1] + s.qd[i - 1];
        k.vax[0] = s.qdr[0]; k.vay[0] = s.qdr[1];
        k.ex[0] = s.cs[0] * (R)Mdl::cx(0) - s.sn[0] * (R)Mdl::cy(0);
        k.ey[0] = s.sn[0] * (R)Mdl::cx(0) + s.cs[0] * (R)Mdl::cy(0);
        k.px[0] = k.ex[0]; k.py[0] = k.ey[0];
        k.ax[0] = zero; k.ay[0] = zero; k.lx[0] = zero; k.ly[0] = zero;      // the root anchor is the origin
        RL_UNROLL
        for (int i = 1; i < 4; ++i) {
            const int p = i - 1, j = i - 1;
            k.lx[i] = s.cs[p] * c.jx[j] - s.sn[p] * c.jy[j];                 // R(phi_parent) * joint offset
            k.ly[i] = s.sn[p] * c.jx[j] + s.cs[p] * c.jy[j];
            if (i == 1) { k.ax[i] = k.lx[i]; k.ay[i] = k.ly[i]; }
            else { k.ax[i] = k.ax[p] + k.lx[i]; k.ay[i] = k.ay[p] + k.ly[i]; }
            k.vax[i] = k.vax[p] - k.om[p] * k.ly[i];                         // + Omega_p x d
            k.vay[i] = k.vay[p] + k.om[p] * k.lx[i];
            k.ex[i] = s.cs[i] * c.cx[j] - s.sn[i] * c.cy[j];                 // R(phi_i) * com offset
            k.ey[i] = s.sn[i] * c.cx[j] + s.cs[i] * c.cy[j];
            k.px[i] = k.ax[i] + k.ex[i];
            k.py[i] = k.ay[i] + k.ey[i];
        }
        RL_UNROLL
        for (int i = 0; i < 4; ++i) {
            k.vpx[i] = k.vax[i] - k.om[i] * k.ey[i];
            k.vpy[i] = k.vay[i] + k.om[i] * k.ex[i];
        }
    }
    // centre of mass of the whole tree (world) and its velocity: torso + (own leg + other leg), scaled by 1 / total mass
    template <typename R, typename V, class X>
    RL_HD static void com(const X& x, const LegK<V>& c, const State<V>& s, V& cx, V& cy, V& vx, V& vy) {
        Kin<V> k;
        kinematics<R, V>(c, s, k);
        V lsx = c.mass[0] * k.px[1], lsy = c.mass[0] * k.py[1];
        V lvx = c.mass[0] * k.vpx[1], lvy = c.mass[0] * k.vpy[1];
        RL_UNROLL
        for (int i = 2; i < 4; ++i) {
            lsx = lsx + c.mass[i - 1] * k.px[i];
            lsy = lsy + c.mass[i - 1] * k.py[i];
            lvx = lvx + c.mass[i - 1] * k.vpx[i];
            lvy = lvy + c.mass[i - 1] * k.vpy[i];
        }
        const R m0 = (R)Mdl::mass(0), im = (R)(1.0 / total_mass());
        const V sx = m0 * k.px[0] + (lsx + x.other(lsx));
        const V sy = m0 * k.py[0] + (lsy + x.other(lsy));
        const V mvx = m0 * k.vpx[0] + (lvx + x.other(lvx));
        const V mvy = m0 * k.vpy[0] + (lvy + x.other(lvy));
        cx = s.qr[0] + sx * im;
        cy = s.qr[1] + sy * im;
        vx = mvx * im;
        vy = mvy * im;
    }

    // One sub-step of length h.  act[j]: motor torque on the leg's j-th hinge.  X: x.other(v) = the other leg's v.
    template <typename R, typename V, class X>
    RL_HD static void substep(const X& x, const LegK<V>& c, State<V>& s, const V* act, R h) {
        constexpr double GX = Mdl::gx(), GY = Mdl::gy();
        const V zero = Lanes<V>::splat((R)0);
        Kin<V> kn;
        kinematics<R, V>(c, s, kn);
        V (&om)[4] = kn.om; V (&ax)[4] = kn.ax; V (&ay)[4] = kn.ay; V (&lx)[4] = kn.lx; V (&ly)[4] = kn.ly;
        V (&ex)[4] = kn.ex; V (&ey)[4] = kn.ey; V (&px)[4] = kn.px; V (&py)[4] = kn.py;
        V (&vax)[4] = kn.vax; V (&vay)[4] = kn.vay;
        // ---- hinge torques: spring (ref 0), damper, soft range limits, motor ---------------------------------------------
        V tau[4];
        RL_UNROLL
        for (int j = 0; j < 3; ++j) {
            const V xq = s.q[j], v = s.qd[j];
            V t = -(c.stiff[j] * xq) - c.damp[j] * v;
            const V viol = xq - rl_max(c.lo[j], rl_min(xq, c.hi[j]));
            const V bv = (R)Mdl::limit_b() * v;
            const V damp = rl_if_nonzero(viol, bv, zero);
            t = (t - (R)Mdl::limit_k() * viol) - damp;
            tau[1 + j] = t + act[j];
        }
        // ---- contacts ----------------------------------------------------------------------------------------------------------
        V fx[4], fy[4], tz[4];
        RL_UNROLL
        for (int i = 0; i < 4; ++i) { fx[i] = zero; fy[i] = zero; tz[i] = zero; }
        // where the spheres are (all of them first: the tests below then wait for nothing) ...
        V trx[NCT > 0 ? NCT : 1], try_[NCT > 0 ? NCT : 1], tdp[NCT > 0 ? NCT : 1], lrx[6], lry[6], ldp[6];
        static_for<0, NCT>([&](auto Cc) {
            constexpr int cc = decltype(Cc)::value;
            sphere_pose<R, V>(Lanes<V>::splat((R)Mdl::cpx(cc)), Lanes<V>::splat((R)Mdl::cpy(cc)),
                              Lanes<V>::splat((R)Mdl::crad(cc)), s.cs[0], s.sn[0], ax[0], s.qr[0], trx[cc], try_[cc], tdp[cc]);
        });
        RL_UNROLL
        for (int cc = 0; cc < 6; ++cc) {
            const int b = 1 + cc / 2;
            sphere_pose<R, V>(c.cpx[cc], c.cpy[cc], c.crad[cc], s.cs[b], s.sn[b], ax[b], s.qr[0], lrx[cc], lry[cc], ldp[cc]);
        }
        // ... and what the touching ones push with, body by body (one group for all leg spheres but the feet's was
        // measured as well: no faster)
        if constexpr (NCT > 0) {
            bool any = false;
            static_for<0, NCT>([&](auto Cc) { any = any || rl_any_pos(tdp[decltype(Cc)::value]); });
            if (any) {
                static_for<0, NCT>([&](auto Cc) {
                    constexpr int cc = decltype(Cc)::value;
                    sphere_force<R, V>(trx[cc], try_[cc], tdp[cc], Lanes<V>::splat((R)Mdl::crad(cc)),
                                       Lanes<V>::splat((R)Mdl::cmu(cc)), ax[0], ay[0], px[0], py[0], vax[0], vay[0], om[0],
                                       fx[0], fy[0], tz[0]);
                });
            }
        }
        RL_UNROLL
        for (int j = 0; j < 3; ++j) {
            const int b = 1 + j, c0 = 2 * j, c1 = 2 * j + 1;
            if (rl_any_pos(ldp[c0]) || rl_any_pos(ldp[c1])) {
                sphere_force<R, V>(lrx[c0], lry[c0], ldp[c0], c.crad[c0], c.cmu[c0], ax[b], ay[b], px[b], py[b], vax[b],
                                   vay[b], om[b], fx[b], fy[b], tz[b]);
                sphere_force<R, V>(lrx[c1], lry[c1], ldp[c1], c.crad[c1], c.cmu[c1], ax[b], ay[b], px[b], py[b], vax[b],
                                   vay[b], om[b], fx[b], fy[b], tz[b]);
            }
        }
        // ---- velocity-product accelerations moved to the right-hand side, body wrenches about the own anchors ----------------
        V aax[4], aay[4], Fx[4], Fy[4], Nz[4];
        RL_UNROLL
        for (int i = 0; i < 4; ++i) {
            const V w2 = om[i] * om[i];
            V acx, acy;
            if (i == 0) {
                acx = -(w2 * ex[0]);
                acy = -(w2 * ey[0]);
            } else {
                const int p = i - 1;
                const V wp2 = om[p] * om[p];
                if (p == 0) { aax[i] = -(wp2 * lx[i]); aay[i] = -(wp2 * ly[i]); }
                else { aax[i] = aax[p] - wp2 * lx[i]; aay[i] = aay[p] - wp2 * ly[i]; }
                acx = aax[i] - w2 * ex[i];
                acy = aay[i] - w2 * ey[i];
            }
            const V m = (i == 0) ? Lanes<V>::splat((R)Mdl::mass(0)) : c.mass[i - 1];
            Fx[i] = fx[i] + m * ((R)GX - acx);
            Fy[i] = fy[i] + m * ((R)GY - acy);
            Nz[i] = (ex[i] * Fy[i] - ey[i] * Fx[i]) + tz[i];
        }
        // leaves -> root along the leg
        RL_UNROLL
        for (int i = 3; i >= 2; --i) {
            const int p = i - 1;
            Nz[p] = Nz[p] + (Nz[i] + (lx[i] * Fy[i] - ly[i] * Fx[i]));
            Fx[p] = Fx[p] + Fx[i];
            Fy[p] = Fy[p] + Fy[i];
        }
        // ... and both legs onto the torso (symmetric sums)
        const V legN = Nz[1] + (lx[1] * Fy[1] - ly[1] * Fx[1]);
        const V Nz0 = Nz[0] + (legN + x.other(legN));
        const V Fx0 = Fx[0] + (Fx[1] + x.other(Fx[1]));
        const V Fy0 = Fy[0] + (Fy[1] + x.other(Fy[1]));
        // ---- composite bodies: first moment and inertia about the ROOT origin ----------------------------------------------------
        V hx[4], hy[4], J[4];
        RL_UNROLL
        for (int i = 0; i < 4; ++i) {
            const V m = (i == 0) ? Lanes<V>::splat((R)Mdl::mass(0)) : c.mass[i - 1];
            const V in = (i == 0) ? Lanes<V>::splat((R)Mdl::inertia(0)) : c.inertia[i - 1];
            hx[i] = m * px[i];
            hy[i] = m * py[i];
            J[i] = in + m * (px[i] * px[i] + py[i] * py[i]);
        }
        RL_UNROLL
        for (int i = 3; i >= 2; --i) {
            hx[i - 1] = hx[i - 1] + hx[i];
            hy[i - 1] = hy[i - 1] + hy[i];
            J[i - 1] = J[i - 1] + J[i];
        }
        const V hx0 = hx[0] + (hx[1] + x.other(hx[1]));
        const V hy0 = hy[0] + (hy[1] + x.other(hy[1]));
        const V J0 = J[0] + (J[1] + x.other(J[1]));
        // ---- joint-space inertia of the leg and its coupling to the root (u = a_P1, a_P2, torso angular acceleration) --------------
        V gx[3], gy[3], gt[3], K[3][3];
        RL_UNROLL
        for (int kk = 1; kk < 4; ++kk) {
            const int j0 = kk - 1;
            gx[j0] = -(hy[kk] - c.mc[j0] * ay[kk]);
            gy[j0] = hx[kk] - c.mc[j0] * ax[kk];
            gt[j0] = J[kk] - (ax[kk] * hx[kk] + ay[kk] * hy[kk]);
            RL_UNROLL
            for (int j = 1; j <= kk; ++j) {
                V v = (J[kk] - ((ax[j] + ax[kk]) * hx[kk] + (ay[j] + ay[kk]) * hy[kk])) +
                      c.mc[j0] * (ax[j] * ax[kk] + ay[j] * ay[kk]);
                if (j == kk) v = v + c.arm[j0];
                K[j0][j - 1] = v;
            }
        }
        // ---- eliminate the leg: y = K^-1 r, Y = K^-1 G, Schur complement onto the root ----------------------------------------------
        V adj[6];
        spd3_inverse<V>(K[0][0], K[1][0], K[2][0], K[1][1], K[2][1], K[2][2], adj);
        V y[3], Yx[3], Yy[3], Yt[3];
        spd3_apply<V>(adj, Nz[1] + tau[1], Nz[2] + tau[2], Nz[3] + tau[3], y[0], y[1], y[2]);
        spd3_apply<V>(adj, gx[0], gx[1], gx[2], Yx[0], Yx[1], Yx[2]);
        spd3_apply<V>(adj, gy[0], gy[1], gy[2], Yy[0], Yy[1], Yy[2]);
        spd3_apply<V>(adj, gt[0], gt[1], gt[2], Yt[0], Yt[1], Yt[2]);
        const V sxx = dot3(gx[0], gx[1], gx[2], Yx[0], Yx[1], Yx[2]);
        const V syx = dot3(gy[0], gy[1], gy[2], Yx[0], Yx[1], Yx[2]);
        const V syy = dot3(gy[0], gy[1], gy[2], Yy[0], Yy[1], Yy[2]);
        const V stx = dot3(gt[0], gt[1], gt[2], Yx[0], Yx[1], Yx[2]);
        const V sty = dot3(gt[0], gt[1], gt[2], Yy[0], Yy[1], Yy[2]);
        const V stt = dot3(gt[0], gt[1], gt[2], Yt[0], Yt[1], Yt[2]);
        const V sux = dot3(gx[0], gx[1], gx[2], y[0], y[1], y[2]);
        const V suy = dot3(gy[0], gy[1], gy[2], y[0], y[1], y[2]);
        const V sut = dot3(gt[0], gt[1], gt[2], y[0], y[1], y[2]);
        // root block [[m, 0, -hy0], [0, m, hx0], [., ., J0]] minus both legs' complements (symmetric sums)
        const V mt = Lanes<V>::splat((R)total_mass());
        const V Rxx = mt - (sxx + x.other(sxx));
        const V Ryx = -(syx + x.other(syx));
        const V Ryy = mt - (syy + x.other(syy));
        const V Rtx = -hy0 - (stx + x.other(stx));
        const V Rty = hx0 - (sty + x.other(sty));
        const V Rtt = J0 - (stt + x.other(stt));
        const V rux = Fx0 - (sux + x.other(sux));
        const V ruy = Fy0 - (suy + x.other(suy));
        const V rut = Nz0 - (sut + x.other(sut));
        V adr[6], u0, u1, u2;
        spd3_inverse<V>(Rxx, Ryx, Rtx, Ryy, Rty, Rtt, adr);
        spd3_apply<V>(adr, rux, ruy, rut, u0, u1, u2);
        // ---- back-substitute the leg, integrate (semi-implicit Euler), carry the sines -------------------------------------------
        s.qdr[0] = s.qdr[0] + h * u0; s.qdr[1] = s.qdr[1] + h * u1; s.qdr[2] = s.qdr[2] + h * u2;
        RL_UNROLL
        for (int r = 0; r < 3; ++r) s.qr[r] = s.qr[r] + h * s.qdr[r];
        RL_UNROLL
        for (int j = 0; j < 3; ++j) {
            const V th = y[j] - (Yx[j] * u0 + (Yy[j] * u1 + Yt[j] * u2));
            s.qd[j] = s.qd[j] + h * th;
            s.q[j] = s.q[j] + h * s.qd[j];
        }
        V omn = s.qdr[2];
        RL_UNROLL
        for (int i = 0; i < 4; ++i) {
            if (i > 0) omn = omn + s.qd[i - 1];
            rl_rotate_small_v<R, V>(s.sn[i], s.cs[i], h * omn);
        }
    }

    // ---- one env per lane / host thread: (q, qd)[9] <-> both legs side by side -----------------------------------------------------
    // q = [P1, P2, torso angle, back leg hinges 1..3, front leg hinges 4..6]; sn / cs: the seven absolute angles
    template <typename R>
    RL_HD static void load(const R* q, const R* qd, const R* sn, const R* cs, State<V2<R>>& s) {
        RL_UNROLL
        for (int r = 0; r < 3; ++r) { s.qr[r] = V2<R>{q[r], q[r]}; s.qdr[r] = V2<R>{qd[r], qd[r]}; }
        RL_UNROLL
        for (int j = 0; j < 3; ++j) {
            s.q[j] = V2<R>{q[3 + j], q[6 + j]};
            s.qd[j] = V2<R>{qd[3 + j], qd[6 + j]};
            s.sn[1 + j] = V2<R>{sn[1 + j], sn[4 + j]};
            s.cs[1 + j] = V2<R>{cs[1 + j], cs[4 + j]};
        }
        s.sn[0] = V2<R>{sn[0], sn[0]};
        s.cs[0] = V2<R>{cs[0], cs[0]};
    }
    template <typename R>
    RL_HD static void store(const State<V2<R>>& s, R* q, R* qd) {
        RL_UNROLL
        for (int r = 0; r < 3; ++r) { q[r] = s.qr[r].x; qd[r] = s.qdr[r].x; }
        RL_UNROLL
        for (int j = 0; j < 3; ++j) {
            q[3 + j] = s.q[j].x; q[6 + j] = s.q[j].y;
            qd[3 + j] = s.qd[j].x; qd[6 + j] = s.qd[j].y;
        }
    }
    // centre of mass of (q, qd) and its velocity, env-per-lane form (exact sines of the seven absolute angles)
    template <typename R>
    RL_HD static void com_of(const R* q, const R* qd, R& cx, R& cy, R& vx, R& vy) {
        R sn[NB], cs[NB];
        PlanarTree<Mdl>::template angles<R>(q, sn, cs);
        State<V2<R>> s;
        load(q, qd, sn, cs, s);
        const LegK<V2<R>> k = both_leg_constants<R>();
        V2<R> c0, c1, v0, v1;
        com<R, V2<R>, BothLegs>(BothLegs(), k, s, c0, c1, v0, v1);
        cx = c0.x; cy = c1.x; vx = v0.x; vy = v1.x;
    }
    // n sub-steps from (q, qd) with hinge torques tau[1..6] (tau[0] unused), exact sines at the start
    template <typename R>
    RL_HD static void advance(R* q, R* qd, const R* tau, R h, int n) {
        R sn[NB], cs[NB];
        PlanarTree<Mdl>::template angles<R>(q, sn, cs);
        State<V2<R>> s;
        load(q, qd, sn, cs, s);
        const LegK<V2<R>> k = both_leg_constants<R>();
        V2<R> act[3];
        RL_UNROLL
        for (int j = 0; j < 3; ++j) act[j] = V2<R>{tau[1 + j], tau[4 + j]};
        const BothLegs x;
        for (int it = 0; it < n; ++it) substep<R, V2<R>, BothLegs>(x, k, s, act, h);
        store(s, q, qd);
    }
};

}  // namespace rl
