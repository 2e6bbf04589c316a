// dyn_two_legs.h -- one sub-step of a planar torso carrying two three-link legs (HalfCheetah, Walker2D) as ONE program
// over a value type V with one BODY per lane: lane (leg L, role r) holds body r of the chain torso (r = 0) -> hip link
// (1) -> shin (2) -> foot (3) of leg L; the torso is held by both legs' role-0 lanes with identical values.  The program is
// the definition of the env's arithmetic; it is instantiated three ways and every instantiation evaluates the same
// expressions in the same association (the front-end's per-expression a * b + c fusion sites included):
//
//   V = V8<R> (8-component vector, component 2 r + L = (role r, leg L): the two legs' copies of a role side by side)
//                 one env per lane / host thread -- the host oracle build, the per-step VecEnv kernels and the
//                 env-per-lane rollouts.  The arithmetic is four packed-f32 instructions per operation on gfx950, one
//                 per ROLE: every role move (up / down / nxt / prv / root / first, sel_root / sel_leaf) carries whole
//                 aligned register pairs -- a rename, no instruction -- other() is the packed instructions' own
//                 half-swap, and work that is meaningless on a role (the leg block on role 0) is dead as a whole
//                 instruction.  (Leg-major components, rounds 5's order, split every role move over two pairs:
//                 v_mov traffic and 308 live registers in the per-step kernel.)
//   V = V2<float> (both legs side by side, the roles on the four lanes of a quad)   rollout_two_leg_quad_kernel, 16 envs
//                 per wavefront; role moves are quad-permute DPP, the other leg is the swapped pair.
//   V = float     (one body per lane: roles on the lanes of a quad, the other leg eight lanes further in the row)
//                 rollout_two_leg_wave_kernel, one env per wavefront, where a lone wavefront pays per issued instruction:
//                 every lane runs ~1/2 the instruction stream of the former one-leg-per-lane chain walk (the spheres,
//                 link kinematics, wrenches and composite bodies of four bodies side by side; the 3 x 3 leg block solved
//                 one ROW per lane by cofactors in the cyclic order of the three leg roles).
//
// Exchange context X (template parameter; x.f(v) = the value of v in another lane):
//   up / down     parent / child role of the same leg (role 0's parent and role 3's child: the lane itself)
//   nxt / prv     cyclic successor / predecessor among the leg roles 1 -> 2 -> 3 -> 1 (role 0: itself)
//   root / first  role 0's / role 1's value, on every lane of the leg
//   other         the same role of the other leg
//   sel_root(a, b) / sel_leaf(a, b)   a on role 0 / role 3, b elsewhere
//
// What is valid where: kinematics, contacts, wrench and composite quantities on every lane; the leg block (rows of K,
// cofactors, y, Y) on roles 1..3 only -- role 0 runs the same instructions on meaningless operands (possibly NaN) and
// nothing it computes there is selected into a value that is used; the root block on role 0, broadcast by root().
// Adding the exact +0 a chain start contributes changes no bit of a value that is not -0 (sums here never are).
//
// Formulation: composite rigid bodies in joint coordinates, everything about the ROOT origin; the joint-space matrix is
// block-arrow,
//     [ R    G_b   G_f ] [ u    ]   [ r_u ]      u = (a_P1, a_P2, torso angular acceleration),
//     [ G_b' K_b   0   ] [ th_b ] = [ r_b ]      K_L = the leg's 3x3 hinge block, G_L its coupling to the root,
//     [ G_f' 0     K_f ] [ th_f ]   [ r_f ]
// K_L[k][j] (k the deeper hinge) = Jc_k + (a_k - a_j) . hc_k  with  Jc_k / hc_k = inertia / first moment of the subtree
// of hinge k about its own anchor a_k;  G_L = (-hc_y, hc_x, Jc + a . hc).  Each leg eliminates its own block (one
// reciprocal per lane) and the root system is  R - sum_L G_L K_L^-1 G_L'.
//
// Model traits: dyn_planar.h's, plus the contact table  NC, cbody(c), cpx(c), cpy(c), crad(c), cmu(c)  ordered as
// [torso spheres (2 or 4)][two per leg body, back leg][two per leg body, front leg], and CONTACT_K / CONTACT_B /
// FRICTION_C.  The torso's spheres are split between its two lanes (leg 0's lane takes the first half).
#pragma once
#include "dyn_planar.h"

namespace rl {

template <typename R> using V8 = R __attribute__((ext_vector_type(8)));

// ---- value-type helpers: the scalar forms live in rl_math.h; these are their vector twins ------------------------------
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
// eight components (the selects are the scalar forms' "x > y ? x : y", component by component)
#define RL_V8_FORMS(R)                                                                                                  \
    RL_HD V8<R> rl_abs(V8<R> v) { return __builtin_elementwise_abs(v); }                                                \
    RL_HD V8<R> rl_max(V8<R> a, V8<R> b) { return a > b ? a : b; }                                                      \
    RL_HD V8<R> rl_min(V8<R> a, V8<R> b) { return a < b ? a : b; }                                                      \
    RL_HD V8<R> rl_if_pos(V8<R> c, V8<R> a, V8<R> b) { return c > (R)0 ? a : b; }                                       \
    RL_HD V8<R> rl_if_nonzero(V8<R> c, V8<R> a, V8<R> b) { return c != (R)0 ? a : b; }                                  \
    RL_HD V8<R> rl_recip_normal(V8<R> d) {                                                                              \
        V8<R> r;                                                                                                        \
        RL_UNROLL                                                                                                       \
        for (int i = 0; i < 8; ++i) r[i] = rl_recip_normal((R)d[i]);                                                    \
        return r;                                                                                                       \
    }
RL_V8_FORMS(float)
RL_V8_FORMS(double)
#undef RL_V8_FORMS
// a scalar as a value of type V
template <typename V> struct Lanes;
template <> struct Lanes<float> { RL_HD static float splat(float r) { return r; } };
template <> struct Lanes<double> { RL_HD static double splat(double r) { return r; } };
template <> struct Lanes<V2<float>> { RL_HD static V2<float> splat(float r) { return V2<float>{r, r}; } };
template <> struct Lanes<V2<double>> { RL_HD static V2<double> splat(double r) { return V2<double>{r, r}; } };
template <> struct Lanes<V8<float>> { RL_HD static V8<float> splat(float r) { return (V8<float>)(r); } };
template <> struct Lanes<V8<double>> { RL_HD static V8<double> splat(double r) { return (V8<double>)(r); } };
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

// all eight lanes of an env in one value: component 2 r + L = (role r, leg L)
struct EightLanes {
    static constexpr int leg_of(int c) { return c & 1; }
    static constexpr int role_of(int c) { return c >> 1; }
#define RL_SHUF(...) __builtin_shufflevector(v, v, __VA_ARGS__)
    template <typename V> RL_HD V up(V v) const { return RL_SHUF(0, 1, 0, 1, 2, 3, 4, 5); }
    template <typename V> RL_HD V down(V v) const { return RL_SHUF(2, 3, 4, 5, 6, 7, 6, 7); }
    template <typename V> RL_HD V nxt(V v) const { return RL_SHUF(0, 1, 4, 5, 6, 7, 2, 3); }
    template <typename V> RL_HD V prv(V v) const { return RL_SHUF(0, 1, 6, 7, 2, 3, 4, 5); }
    template <typename V> RL_HD V root(V v) const { return RL_SHUF(0, 1, 0, 1, 0, 1, 0, 1); }
    template <typename V> RL_HD V first(V v) const { return RL_SHUF(2, 3, 2, 3, 2, 3, 2, 3); }
    template <typename V> RL_HD V other(V v) const { return RL_SHUF(1, 0, 3, 2, 5, 4, 7, 6); }
#undef RL_SHUF
    template <typename V> RL_HD V sel_root(V a, V b) const { return __builtin_shufflevector(a, b, 0, 1, 10, 11, 12, 13, 14, 15); }
    template <typename V> RL_HD V sel_leaf(V a, V b) const { return __builtin_shufflevector(a, b, 8, 9, 10, 11, 12, 13, 6, 7); }
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
        if (NCT != 2 && NCT != 4) return false;        // split evenly between the torso's two lanes
        for (int c = 0; c < NCT; ++c)
            if (Mdl::cbody(c) != 0) return false;
        for (int c = 0; c < 12; ++c)
            if (Mdl::cbody(NCT + c) != 1 + c / 2) return false;
        return true;
    }
    static_assert(contacts_ok(), "contact table: 2 or 4 torso spheres, then two per leg body");
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
    // a sphere slot nothing ever touches (radius -1e30: depth stays hugely negative, every product finite)
    static constexpr double NO_SPHERE_RAD = -1.0e30;
    static constexpr double NO_LIMIT = 3.0e38;

    // the constants of one lane = one body (and the hinge that carries it), as values of type V
    template <typename V>
    struct LaneK {
        V jx, jy;            // own hinge anchor in the parent's frame (role 0: 0 -- the root anchor is the origin)
        V cx, cy;            // centre of mass in the own frame
        V mass, inertia;
        V arm, stiff, damp, lo, hi;      // own hinge (role 0: none; limits at +-NO_LIMIT keep its "violation" at zero)
        V mc;                // mass of the subtree the own hinge carries (role 0: unused)
        V cpx[2], cpy[2], crad[2], cmu[2];   // two contact spheres
    };
    // scalar constants of lane (leg, role); both may be run-time values (a chain of selects over compile-time numbers)
    template <typename R>
    RL_HD static LaneK<R> lane_constants(int leg, int role) {
        LaneK<R> k;
#define RL_BODYSEL(f) (R)(role == 0 ? Mdl::f(0) : role == 1 ? (leg == 0 ? Mdl::f(1) : Mdl::f(4))                         \
                                    : role == 2 ? (leg == 0 ? Mdl::f(2) : Mdl::f(5)) : (leg == 0 ? Mdl::f(3) : Mdl::f(6)))
#define RL_HINGESEL(f, AT0) (R)(role == 0 ? (AT0) : role == 1 ? (leg == 0 ? Mdl::f(1) : Mdl::f(4))                      \
                                          : role == 2 ? (leg == 0 ? Mdl::f(2) : Mdl::f(5)) : (leg == 0 ? Mdl::f(3) : Mdl::f(6)))
        k.jx = RL_HINGESEL(jx, 0.0); k.jy = RL_HINGESEL(jy, 0.0);
        k.cx = RL_BODYSEL(cx); k.cy = RL_BODYSEL(cy);
        k.mass = RL_BODYSEL(mass); k.inertia = RL_BODYSEL(inertia);
        k.arm = RL_HINGESEL(armature, 0.0); k.stiff = RL_HINGESEL(stiffness, 0.0); k.damp = RL_HINGESEL(damping, 0.0);
        k.lo = RL_HINGESEL(lo, -NO_LIMIT); k.hi = RL_HINGESEL(hi, NO_LIMIT);
#undef RL_BODYSEL
#undef RL_HINGESEL
        k.mc = (R)(role == 0 ? 0.0 : role == 1 ? (leg == 0 ? subtree_mass(1) : subtree_mass(4))
                             : role == 2 ? (leg == 0 ? subtree_mass(2) : subtree_mass(5))
                                         : (leg == 0 ? subtree_mass(3) : subtree_mass(6)));
        RL_UNROLL
        for (int s = 0; s < 2; ++s) {
            // torso: NCT / 2 spheres per lane (leg 0's lane the first half), a slot beyond them is empty;
            // leg body b = 3 leg + role: its two spheres
#define RL_SPHSEL(f, EMPTY)                                                                                             \
    (R)(role == 0 ? (NCT == 4 ? (leg == 0 ? Mdl::f(s) : Mdl::f(2 + s)) : (s == 0 ? (leg == 0 ? Mdl::f(0) : Mdl::f(1)) : (EMPTY))) \
        : role == 1 ? (leg == 0 ? Mdl::f(NCT + s) : Mdl::f(NCT + 6 + s))                                                \
        : role == 2 ? (leg == 0 ? Mdl::f(NCT + 2 + s) : Mdl::f(NCT + 8 + s))                                            \
                    : (leg == 0 ? Mdl::f(NCT + 4 + s) : Mdl::f(NCT + 10 + s)))
            k.cpx[s] = RL_SPHSEL(cpx, 0.0); k.cpy[s] = RL_SPHSEL(cpy, 0.0);
            k.crad[s] = RL_SPHSEL(crad, NO_SPHERE_RAD); k.cmu[s] = RL_SPHSEL(cmu, 0.0);
#undef RL_SPHSEL
        }
        return k;
    }
    // all eight lanes side by side
    template <typename R>
    RL_HD static LaneK<V8<R>> all_lane_constants() {
        LaneK<V8<R>> k;
        RL_UNROLL
        for (int i = 0; i < 8; ++i) {
            const LaneK<R> a = lane_constants<R>(EightLanes::leg_of(i), EightLanes::role_of(i));
#define RL_PUT(f) k.f[i] = a.f
            RL_PUT(jx); RL_PUT(jy); RL_PUT(cx); RL_PUT(cy); RL_PUT(mass); RL_PUT(inertia); RL_PUT(arm); RL_PUT(stiff);
            RL_PUT(damp); RL_PUT(lo); RL_PUT(hi); RL_PUT(mc);
#undef RL_PUT
            RL_UNROLL
            for (int s = 0; s < 2; ++s) { k.cpx[s][i] = a.cpx[s]; k.cpy[s][i] = a.cpy[s]; k.crad[s][i] = a.crad[s]; k.cmu[s][i] = a.cmu[s]; }
        }
        return k;
    }
    // one role of both legs (the quad form): x = leg 0, y = leg 1
    template <typename R>
    RL_HD static LaneK<V2<R>> role_constants(int role) {
        const LaneK<R> a = lane_constants<R>(0, role), b = lane_constants<R>(1, role);
        LaneK<V2<R>> k;
#define RL_PAIR(f) k.f = V2<R>{a.f, b.f}
        RL_PAIR(jx); RL_PAIR(jy); RL_PAIR(cx); RL_PAIR(cy); RL_PAIR(mass); RL_PAIR(inertia); RL_PAIR(arm); RL_PAIR(stiff);
        RL_PAIR(damp); RL_PAIR(lo); RL_PAIR(hi); RL_PAIR(mc);
        RL_UNROLL
        for (int s = 0; s < 2; ++s) { RL_PAIR(cpx[s]); RL_PAIR(cpy[s]); RL_PAIR(crad[s]); RL_PAIR(cmu[s]); }
#undef RL_PAIR
        return k;
    }

    // state of one lane
    template <typename V>
    struct State {
        V q, w;              // own coordinate and its rate: hinge angle (roles 1..3), torso angle (role 0)
        V om;                // absolute angular rate of the own body = w summed root -> own (abs_rates)
        V sn, cs;            // carried sine / cosine of the own body's absolute angle
        V p1, p2, v1, v2;    // root translation and its rate, the same on every lane
    };

    // sum of v over the chain root -> own role, in the association ((v_0 + v_1) + v_2) + v_3
    template <typename R, typename V, class X>
    RL_HD static V chain_sum(const X& x, V v) {
        const V add = x.sel_root(Lanes<V>::splat((R)0), v);      // role 0 reads itself: it must not add itself again
        V s = x.up(v) + add;
        s = x.up(s) + add;
        s = x.up(s) + add;
        return s;
    }
    template <typename R, typename V, class X>
    RL_HD static void abs_rates(const X& x, State<V>& s) { s.om = chain_sum<R, V, X>(x, s.w); }
    // sum over the chain root -> own role of a quantity that is zero on role 0: no select, and two rounds (role 1's
    // sum is its own value plus the root's exact zero)
    template <typename V, class X>
    RL_HD static V chain_sum0(const X& x, V l) {
        V a = x.up(l) + l;
        a = x.up(a) + l;
        return a;
    }
    // own + the roles below in the leg, association  own + (child + (grandchild ...)); valid on roles 1..3
    template <typename R, typename V, class X>
    RL_HD static V subtree_sum(const X& x, V v) {
        const V add = x.sel_leaf(Lanes<V>::splat((R)0), v);      // role 3 reads itself
        V s = x.down(v) + add;
        s = x.down(s) + add;
        return s;
    }
    // sum over the six leg bodies, read at role 1:  (j_1 + j_2) + j_3  with  j_r = own leg's + other leg's
    template <typename V, class X>
    RL_HD static V leg_sum(const X& x, V p) {
        const V j = p + x.other(p);
        return (j + x.nxt(j)) + x.prv(j);
    }

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

    // positions / velocities of the own body relative to the root origin
    template <typename V>
    struct Kin {
        V omp;               // the parent's absolute rate
        V lx, ly;            // world vector parent anchor -> own anchor  (role 0: 0)
        V ax, ay;            // own anchor
        V rvx, rvy;          // velocity of the own anchor relative to the root's
        V ex, ey;            // own anchor -> own centre of mass
        V px, py;            // centre of mass
    };
    template <typename V, class X>
    RL_HD static void kinematics(const X& x, const LaneK<V>& c, const State<V>& s, Kin<V>& k) {
        const V csp = x.up(s.cs), snp = x.up(s.sn);
        k.omp = x.up(s.om);
        k.lx = csp * c.jx - snp * c.jy;                          // R(phi_parent) * joint offset
        k.ly = snp * c.jx + csp * c.jy;
        k.ax = chain_sum0(x, k.lx);
        k.ay = chain_sum0(x, k.ly);
        const V mx = -(k.omp * k.ly);                            // Omega_parent x l
        const V my = k.omp * k.lx;
        k.rvx = chain_sum0(x, mx);
        k.rvy = chain_sum0(x, my);
        k.ex = s.cs * c.cx - s.sn * c.cy;                        // R(phi) * com offset
        k.ey = s.sn * c.cx + s.cs * c.cy;
        k.px = k.ax + k.ex;
        k.py = k.ay + k.ey;
    }
    // centre of mass of the whole tree (world) and its velocity; valid on role 0 (both legs' role-0 lanes alike)
    template <typename R, typename V, class X>
    RL_HD static void com(const X& x, const LaneK<V>& c, const State<V>& s, V& cx, V& cy, V& vx, V& vy) {
        Kin<V> k;
        kinematics<V, X>(x, c, s, k);
        const V vpx = (s.v1 + k.rvx) - s.om * k.ey;              // velocity of the centre of mass
        const V vpy = (s.v2 + k.rvy) + s.om * k.ex;
        const V hx = c.mass * k.px, hy = c.mass * k.py, gx = c.mass * vpx, gy = c.mass * vpy;
        const R im = (R)(1.0 / total_mass());
        const V sx = hx + x.first(leg_sum(x, hx));
        const V sy = hy + x.first(leg_sum(x, hy));
        const V mvx = gx + x.first(leg_sum(x, gx));
        const V mvy = gy + x.first(leg_sum(x, gy));
        cx = s.p1 + sx * im;
        cy = s.p2 + sy * im;
        vx = mvx * im;
        vy = mvy * im;
    }

    // One sub-step of length h.  act: motor torque on the own hinge (role 0: 0).
    template <typename R, typename V, class X>
    RL_HD static void substep(const X& x, const LaneK<V>& c, State<V>& s, V act, R h) {
        constexpr double GX = Mdl::gx(), GY = Mdl::gy();
        const V zero = Lanes<V>::splat((R)0);
        // ---- own hinge: spring (ref 0), damper, soft range limits, motor ---------------------------------------------------------
        V tau;
        {
            const V xq = s.q, v = s.w;
            V t = -(c.stiff * xq) - c.damp * v;
            const V viol = xq - rl_max(c.lo, rl_min(xq, c.hi));
            const V bv = (R)Mdl::limit_b() * v;
            const V damp = rl_if_nonzero(viol, bv, zero);
            t = (t - (R)Mdl::limit_k() * viol) - damp;
            tau = t + act;
        }
        // ---- kinematics ------------------------------------------------------------------------------------------------------------
        Kin<V> kn;
        kinematics<V, X>(x, c, s, kn);
        const V om = s.om, ax = kn.ax, ay = kn.ay, ex = kn.ex, ey = kn.ey, px = kn.px, py = kn.py;
        const V vax = s.v1 + kn.rvx, vay = s.v2 + kn.rvy;
        // velocity-product acceleration of the own anchor:  aa = aa_parent - Omega_parent^2 l
        const V wp2 = kn.omp * kn.omp;
        const V tx = -(wp2 * kn.lx), ty = -(wp2 * kn.ly);
        const V aax = chain_sum0(x, tx), aay = chain_sum0(x, ty);
        // ---- contacts: two capsule-end spheres against the floor P1 = 0 (the penalty model of dyn_cheetah.h / dyn_legged.h).  A sphere
        // above the floor adds exact zeros; forces and their moments are taken about the root origin -------------------------------
        V fx = zero, fy = zero, tz = zero;
        RL_UNROLL
        for (int k = 0; k < 2; ++k) {
            const V rx = s.cs * c.cpx[k] - s.sn * c.cpy[k];      // sphere centre relative to the own anchor
            const V ry = s.sn * c.cpx[k] + s.cs * c.cpy[k];
            const V qx = ax + rx, qy = ay + ry;                  // ... relative to the root origin
            const V depth = c.crad[k] - (s.p1 + qx);
            const V vn = vax - om * ry;                          // velocity of the sphere centre
            const V vt = vay + om * rx;
            V fn = (R)Mdl::CONTACT_K * depth - (R)Mdl::CONTACT_B * vn;
            fn = rl_max(fn, zero);
            const V lim = c.cmu[k] * fn;
            const V ft = -rl_max(-lim, rl_min((R)Mdl::FRICTION_C * vt, lim));
            const V lx_ = qx - c.crad[k];                        // applied at the lowest point of the sphere
            const V mom = lx_ * ft - qy * fn;
            fx = fx + rl_if_pos(depth, fn, zero);
            fy = fy + rl_if_pos(depth, ft, zero);
            tz = tz + rl_if_pos(depth, mom, zero);
        }
        // ---- wrench on the own body about the root origin: inertial + gravity part W at the centre of mass, contacts ----------------
        const V w2 = om * om;
        const V acx = aax - w2 * ex, acy = aay - w2 * ey;
        const V Wx = c.mass * ((R)GX - acx), Wy = c.mass * ((R)GY - acy);
        const V Mw = px * Wy - py * Wx;
        const V Fx = fx + Wx, Fy = fy + Wy, Mz = Mw + tz;        // (role 0: with its own half of the torso's spheres)
        // ... summed over the subtree of the own hinge; its moment about the own anchor + the hinge torque drives the hinge
        const V SFx = subtree_sum<R, V, X>(x, Fx), SFy = subtree_sum<R, V, X>(x, Fy), SMz = subtree_sum<R, V, X>(x, Mz);
        const V rhs = (SMz - (ax * SFy - ay * SFx)) + tau;
        // ---- composite bodies: first moment and inertia of the subtree about the ROOT origin, then about the own anchor ---------------
        const V hx = c.mass * px, hy = c.mass * py;
        const V J = c.inertia + c.mass * (px * px + py * py);
        const V Shx = subtree_sum<R, V, X>(x, hx), Shy = subtree_sum<R, V, X>(x, hy), SJ = subtree_sum<R, V, X>(x, J);
        const V hcx = Shx - c.mc * ax, hcy = Shy - c.mc * ay;    // coupling to the root's translation: G = (-hcy, hcx, gt)
        const V gt = SJ - (ax * Shx + ay * Shy);                 // coupling to the torso's rotation
        const V Jc = gt - (ax * hcx + ay * hcy);
        // ---- the leg's 3x3 block, one row per lane in the cyclic order (own b, p = nxt, q = prv) ---------------------------------------
        const V dxa = x.nxt(ax) - ax, dya = x.nxt(ay) - ay;
        const V Kdeep = x.nxt(Jc) + (dxa * x.nxt(hcx) + dya * x.nxt(hcy));   // roles 1, 2: the partner p is the deeper hinge
        const V Kown = Jc - (dxa * hcx + dya * hcy);                        // role 3: its partner is the hip, itself the deeper
        const V Sbp = x.sel_leaf(Kown, Kdeep);
        const V d = Jc + c.arm;
        const V dp = x.nxt(d), dq = x.prv(d), Sbq = x.prv(Sbp), Spq = x.nxt(Sbp);
        // cofactors of the own row = cross product of the partners' rows; every lane expands the determinant along its row
        const V c0 = dp * dq - Spq * Spq;
        const V c1 = Spq * Sbq - Sbp * dq;
        const V c2 = Sbp * Spq - dp * Sbq;
        const V det = d * c0 + (Sbp * c1 + Sbq * c2);
        const V inv = rl_recip_normal(det);
#define RL_LEG_SOLVE(r) ((c0 * (r) + (c1 * x.nxt(r) + c2 * x.prv(r))) * inv)
        const V gxv = -hcy;
        const V y = RL_LEG_SOLVE(rhs), Yx = RL_LEG_SOLVE(gxv), Yy = RL_LEG_SOLVE(hcx), Yt = RL_LEG_SOLVE(gt);
#undef RL_LEG_SOLVE
        // ---- root block [[m, 0, -hy0], [0, m, hx0], [., ., J0]] minus both legs' Schur complements; the legs' own first moments,
        // inertias and wrenches ride in the same sums (valid on role 0) -------------------------------------------------------------
        const V mt = Lanes<V>::splat((R)total_mass());
        const V Rxx = mt - x.first(leg_sum(x, gxv * Yx));
        const V Ryx = -x.first(leg_sum(x, hcx * Yx));
        const V Ryy = mt - x.first(leg_sum(x, hcx * Yy));
        const V Rtx = -hy - x.first(leg_sum(x, gt * Yx + hy));
        const V Rty = hx - x.first(leg_sum(x, gt * Yy - hx));
        const V Rtt = J - x.first(leg_sum(x, gt * Yt - J));
        const V rux = (Wx + (fx + x.other(fx))) - x.first(leg_sum(x, gxv * y - Fx));
        const V ruy = (Wy + (fy + x.other(fy))) - x.first(leg_sum(x, hcx * y - Fy));
        const V rut = (Mw + (tz + x.other(tz))) - x.first(leg_sum(x, gt * y - Mz));
        V adr[6], u0, u1, u2;
        spd3_inverse<V>(Rxx, Ryx, Rtx, Ryy, Rty, Rtt, adr);
        spd3_apply<V>(adr, rux, ruy, rut, u0, u1, u2);
        // ---- back-substitute the hinge, integrate (semi-implicit Euler), carry the sines ----------------------------------------------
        const V U0 = x.root(u0), U1 = x.root(u1), U2 = x.root(u2);
        const V th = y - (Yx * U0 + (Yy * U1 + Yt * U2));
        const V acc = x.sel_root(u2, th);
        s.w = s.w + h * acc;
        s.q = s.q + h * s.w;
        s.v1 = s.v1 + h * U0;
        s.v2 = s.v2 + h * U1;
        s.p1 = s.p1 + h * s.v1;
        s.p2 = s.p2 + h * s.v2;
        abs_rates<R, V, X>(x, s);
        rl_rotate_small_v<R, V>(s.sn, s.cs, h * s.om);
    }

    // ---- one env per lane / host thread: (q, qd)[9] <-> eight lanes side by side ----------------------------------------------------
    // q = [P1, P2, torso angle, back leg hinges 1..3, front leg hinges 4..6]
    template <typename R>
    RL_HD static void load(const R* q, const R* qd, State<V8<R>>& s) {
        const EightLanes x;
        RL_UNROLL
        for (int i = 0; i < 8; ++i) {
            const int role = EightLanes::role_of(i), leg = EightLanes::leg_of(i);
            const int j = role == 0 ? 2 : 2 + 3 * leg + role;
            s.q[i] = q[j];
            s.w[i] = qd[j];
        }
        s.p1 = (V8<R>)(q[0]); s.p2 = (V8<R>)(q[1]); s.v1 = (V8<R>)(qd[0]); s.v2 = (V8<R>)(qd[1]);
        abs_rates<R, V8<R>, EightLanes>(x, s);
        exact_directions<R, V8<R>, EightLanes>(x, s);
    }
    // exact sine / cosine of every body's absolute angle (PlanarTree::angles: phi_child = phi_parent + hinge)
    template <typename R, typename V, class X>
    RL_HD static void exact_directions(const X& x, State<V>& s) {
        const V phi = chain_sum<R, V, X>(x, s.q);
        sincos_lanes(phi, s.sn, s.cs);
    }
    RL_HD static void sincos_lanes(float phi, float& sn, float& cs) { rl_sincos(phi, sn, cs); }
    RL_HD static void sincos_lanes(double phi, double& sn, double& cs) { rl_sincos(phi, sn, cs); }
    template <typename R>
    RL_HD static void sincos_lanes(V2<R> phi, V2<R>& sn, V2<R>& cs) {
        R s0, c0, s1, c1;
        rl_sincos((R)phi.x, s0, c0);
        rl_sincos((R)phi.y, s1, c1);
        sn = V2<R>{s0, s1};
        cs = V2<R>{c0, c1};
    }
    template <typename R>
    RL_HD static void sincos_lanes(V8<R> phi, V8<R>& sn, V8<R>& cs) {
        // (the torso's angle sits in components 0 and 1: evaluated once)
        RL_UNROLL
        for (int i = 0; i < 8; ++i) {
            if (i == 1) { sn[1] = sn[0]; cs[1] = cs[0]; continue; }
            R a, b;
            rl_sincos((R)phi[i], a, b);
            sn[i] = a;
            cs[i] = b;
        }
    }
    template <typename R>
    RL_HD static void store(const State<V8<R>>& s, R* q, R* qd) {
        q[0] = s.p1[0]; q[1] = s.p2[0]; qd[0] = s.v1[0]; qd[1] = s.v2[0];
        q[2] = s.q[0]; qd[2] = s.w[0];
        RL_UNROLL
        for (int i = 2; i < 8; ++i) {
            const int j = 2 + 3 * EightLanes::leg_of(i) + EightLanes::role_of(i);
            q[j] = s.q[i];
            qd[j] = s.w[i];
        }
    }
    // centre of mass of (q, qd) and its velocity, env-per-lane form (exact sines of the seven absolute angles)
    template <typename R>
    RL_HD static void com_of(const R* q, const R* qd, R& cx, R& cy, R& vx, R& vy) {
        State<V8<R>> s;
        load(q, qd, s);
        const LaneK<V8<R>> k = all_lane_constants<R>();
        V8<R> c0, c1, v0, v1;
        com<R, V8<R>, EightLanes>(EightLanes(), k, s, c0, c1, v0, v1);
        cx = c0[0]; cy = c1[0]; vx = v0[0]; vy = v1[0];
    }
    // n sub-steps from (q, qd) with hinge torques tau[1..6] (tau[0] unused), exact sines at the start
    template <typename R>
    RL_HD static void advance(R* q, R* qd, const R* tau, R h, int n) {
        State<V8<R>> s;
        load(q, qd, s);
        const LaneK<V8<R>> k = all_lane_constants<R>();
        // (tried in round 6: the most-used constants pinned in scalar register tuples for the env step, so that the packed
        //  instructions read (role, both legs) as one scalar pair instead of the compiler's two literal multiplies -- the
        //  per-step kernel got SLOWER, 0.77 -> 0.84 ms at 4 M envs: profiles/r06_notes.md)
        V8<R> act;
        RL_UNROLL
        for (int i = 0; i < 8; ++i)
            act[i] = EightLanes::role_of(i) == 0 ? (R)0 : tau[3 * EightLanes::leg_of(i) + EightLanes::role_of(i)];
        const EightLanes x;
        for (int it = 0; it < n; ++it) substep<R, V8<R>, EightLanes>(x, k, s, act, h);
        store(s, q, qd);
    }
};

}  // namespace rl
