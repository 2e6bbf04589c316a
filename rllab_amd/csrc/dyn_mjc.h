// dyn_mjc.h -- joint limits and floor contacts of the legged planar trees (HalfCheetah, Walker2D, Hopper) by MuJoCo's
// documented SOFT-CONSTRAINT model instead of the spring-damper penalties of dyn_two_legs.h / dyn_legged.h; single
// source for the gfx950 kernels and the host oracle build.  Selected per env by the rl_env_cfg flags
// RL_CFG_LIMIT_MUJOCO / RL_CFG_CONTACT_MUJOCO (HalfCheetahEnv(limit_model="mujoco", contact_model="mujoco"), ...);
// the penalty models stay the default.  An option, not a hot path: one env per thread, run-time loops over small
// per-thread arrays, the generic env-per-lane kernels only.
//
// What is restated (MuJoCo's "Computation" chapter: Constraint model / Solver parameters / Contact; the 1.31 binary
// the reference drives -- rllab/mujoco_py/mjlib.py:10 -- is absent, so this follows the published text, as
// dyn_swimmer_chain.h does for the Swimmer's two limit rows):
//   * unconstrained acceleration  M qacc_u = Q  with the joint-space inertia M(q) (armature on the hinge diagonal),
//     Q = passive joint forces (spring, damper) + actuation + gravity - velocity-product terms;
//   * scalar constraint rows i with Jacobian J_i and signed distance r_i, ACTIVE when r_i < margin:
//       joint limit of a hinge      r = q - lo (J = +e_hinge)  or  hi - q (J = -e_hinge);
//       floor contact of a capsule end sphere   r = height of the centre - radius; frame: normal +z, tangent +x, taken
//         at the lowest point of the sphere.  condim 3 (the floor's, half_cheetah.xml:53, hopper.xml:18, walker2d.xml:17)
//         with the pyramidal friction cone: in the plane the cone has two edges, rows  J_n + mu J_t  and  J_n - mu J_t
//         (the two edges across the plane have no motion to act on and are left out); both carry the contact's r;
//   * per row the reference acceleration  a_ref = -b (J v) - k (r - margin),
//       b = 2 / (dmax timeconst),  k = d(r) / (dmax^2 timeconst^2 dampratio^2)      from solref = (timeconst, dampratio),
//       impedance d(r) in [dmin, dmax] from solimp = (dmin, dmax, width): x = min(|r - margin| / width, 1),
//       y = 2 x^2 (x < 1/2), 1 - 2 (1 - x)^2 otherwise, d = dmin + y (dmax - dmin), kept inside [1e-4, 0.9999];
//   * regulariser  R_ii = (1 - d_i) / d_i * A_ii  with  A = J M^-1 J^T;
//   * forces  f >= 0  minimising  1/2 f^T (A + R) f + f^T (J qacc_u - a_ref)  by projected Gauss-Seidel from f = 0,
//     a fixed number of sweeps (MuJoCo's default solver and iteration count, <option iterations> unset in the three
//     files = 100);  qacc = qacc_u + M^-1 J^T f;  semi-implicit Euler as everywhere else in this engine.
// Solver parameters are the MJCFs' own: half_cheetah.xml:38-39 (solreflimit .02 1, solimplimit 0 .8 .03; geoms solref
// .02 1, solimp 0 .8 .01), hopper.xml:5 (geoms solref .02 1, solimp .8 .8 .01, margin .001; limits: the defaults),
// walker2d.xml:6 (everything default: solref .02 1, solimp .9 .95 .001).  Friction per sphere: the body geom's, as the
// penalty model takes it.  At most MAXC contacts are solved at once (the first MAXC active spheres in table order).
// PARITY UNPINNED like every env leg; oracle/np_mjc.py restates the same model independently
// (Lagrangian by automatic differentiation, the quadratic programme by scipy's NNLS).
#pragma once
#include "dyn_planar.h"

namespace rl {

struct MjcSol {
    double timeconst, dampratio, dmin, dmax, width, margin;
};

// Mdl: the PlanarTree traits + the contact table of dyn_legged.h / dyn_cheetah.h; Par: limit() / contact() solver parameters
template <class Mdl, class Par>
struct MjcTree {
    using Tree = PlanarTree<Mdl>;
    static constexpr int NB = Mdl::NB, NV = NB + 2, NC = Mdl::NC;
    static constexpr int MAXC = NC < 8 ? NC : 8;
    static constexpr int KMAX = (NB - 1) + 2 * MAXC;
    static constexpr int SWEEPS = 100;

    template <typename R>
    RL_HD static R impedance(R dist, const MjcSol& p) {
        const R x0 = rl_abs(dist) * (R)(1.0 / p.width);
        const R x = x0 < (R)1 ? x0 : (R)1;
        const R omx = (R)1 - x;
        const R y = x < (R)0.5 ? (R)2 * (x * x) : (R)1 - (R)2 * (omx * omx);
        const R d = (R)p.dmin + y * (R)(p.dmax - p.dmin);
        return rl_clamp(d, (R)1e-4, (R)0.9999);
    }

    // d(point attached to body b at (x1, x2) relative to the root origin) / d(coordinate): rows j1 (P1), j2 (P2)
    template <typename R>
    RL_HD static void point_jac(const PlanarKin<R, NB>& k, int b, R x1, R x2, R* j1, R* j2) {
        j1[0] = (R)1; j2[0] = (R)0;
        j1[1] = (R)0; j2[1] = (R)1;
        for (int j = 0; j < NB; ++j) {
            const bool on = Tree::is_ancestor(j, b);
            j1[2 + j] = on ? -(x2 - k.ay[j]) : (R)0;
            j2[2 + j] = on ? (x1 - k.ax[j]) : (R)0;
        }
    }

    // M = L D L^T in place (lower triangle of M -> strict lower triangle L; D and 1 / D returned)
    template <typename R>
    RL_HD static void factor(R (*M)[NV], R* Dg, R* Di) {
        for (int c = 0; c < NV; ++c) {
            R d = M[c][c];
            for (int t = 0; t < c; ++t) d = d - M[c][t] * M[c][t] * Dg[t];
            Dg[c] = d;
            const R inv = rl_recip_normal(d);
            Di[c] = inv;
            for (int r = c + 1; r < NV; ++r) {
                R v = M[r][c];
                for (int t = 0; t < c; ++t) v = v - M[r][t] * M[c][t] * Dg[t];
                M[r][c] = v * inv;
            }
        }
    }
    template <typename R>
    RL_HD static void solve(const R (*L)[NV], const R* Di, const R* b, R* x) {
        for (int r = 0; r < NV; ++r) {
            R v = b[r];
            for (int t = 0; t < r; ++t) v = v - L[r][t] * x[t];
            x[r] = v;
        }
        for (int r = 0; r < NV; ++r) x[r] = x[r] * Di[r];
        for (int r = NV - 1; r >= 0; --r) {
            R v = x[r];
            for (int t = r + 1; t < NV; ++t) v = v - L[t][r] * x[t];
            x[r] = v;
        }
    }

    // One sub-step of length h.  act_j[NB]: motor torque on hinge j (act_j[0] unused).  limit_mj / contact_mj: which of
    // the two constraint classes goes through the solver (the other keeps its penalty form inside Q).  qfrc (optional,
    // [NV]): J^T f of this sub-step -- MuJoCo's data.qfrc_constraint.
    template <typename R>
#if defined(__HIPCC__)
    __host__ __device__ __attribute__((noinline))       // one copy per env kind, not one per rollout instantiation
#else
    inline
#endif
    static void substep(R* q, R* qd, const R* act_j, R h, bool limit_mj, bool contact_mj, R* qfrc) {
        PlanarKin<R, NB> k;
        Tree::template kinematics<R>(q, qd, k);
        // ---- joint-space inertia (lower triangle) and generalised force ----------------------------------------------
        R M[NV][NV], Q[NV];
        for (int i = 0; i < NV; ++i) {
            Q[i] = (R)0;
            for (int j = 0; j <= i; ++j) M[i][j] = (R)0;
        }
        // velocity-product acceleration of every anchor / centre of mass (coordinates' second derivatives zero)
        R aa1[NB], aa2[NB];
        aa1[0] = (R)0; aa2[0] = (R)0;
        for (int b = 1; b < NB; ++b) {
            const int p = Mdl::parent(b);
            const R w2 = k.om[p] * k.om[p];
            aa1[b] = aa1[p] - w2 * k.lx[b];
            aa2[b] = aa2[p] - w2 * k.ly[b];
        }
        // penalty contacts (when that class is not solved): per-body force at the centre of mass + torque
        R pf1[NB], pf2[NB], ptz[NB];
        for (int b = 0; b < NB; ++b) { pf1[b] = (R)0; pf2[b] = (R)0; ptz[b] = (R)0; }
        if (!contact_mj) {
            for (int c = 0; c < NC; ++c) {
                const int b = Mdl::cbody(c);
                const R lx = (R)Mdl::cpx(c), ly = (R)Mdl::cpy(c), rad = (R)Mdl::crad(c);
                const R rx = k.cs[b] * lx - k.sn[b] * ly;
                const R ry = k.sn[b] * lx + k.cs[b] * ly;
                const R depth = rad - (q[0] + k.ax[b] + rx);
                if (depth > (R)0) {
                    const R vn = k.vax[b] - k.om[b] * ry;
                    const R vt = k.vay[b] + k.om[b] * rx;
                    R fn = (R)Mdl::CONTACT_K * depth - (R)Mdl::CONTACT_B * vn;
                    fn = rl_max(fn, (R)0);
                    const R mu = (R)Mdl::cmu(c);
                    const R ft = -rl_clamp((R)Mdl::FRICTION_C * vt, -mu * fn, mu * fn);
                    const R ax_ = (k.ax[b] + rx - rad) - k.px[b];
                    const R ay_ = (k.ay[b] + ry) - k.py[b];
                    pf1[b] = pf1[b] + fn;
                    pf2[b] = pf2[b] + ft;
                    ptz[b] = ptz[b] + (ax_ * ft - ay_ * fn);
                }
            }
        }
        for (int b = 0; b < NB; ++b) {
            R j1[NV], j2[NV], jw[NV];
            point_jac<R>(k, b, k.px[b], k.py[b], j1, j2);
            jw[0] = (R)0; jw[1] = (R)0;
            for (int j = 0; j < NB; ++j) jw[2 + j] = Tree::is_ancestor(j, b) ? (R)1 : (R)0;
            const R m = (R)Mdl::mass(b), I = (R)Mdl::inertia(b);
            for (int i = 0; i < NV; ++i)
                for (int j = 0; j <= i; ++j)
                    M[i][j] = M[i][j] + (m * (j1[i] * j1[j] + j2[i] * j2[j]) + I * (jw[i] * jw[j]));
            const R w2 = k.om[b] * k.om[b];
            const R ac1 = aa1[b] - w2 * k.ex[b], ac2 = aa2[b] - w2 * k.ey[b];
            const R F1 = m * ((R)Mdl::gx() - ac1) + pf1[b], F2 = m * ((R)Mdl::gy() - ac2) + pf2[b];
            for (int i = 0; i < NV; ++i) Q[i] = Q[i] + ((j1[i] * F1 + j2[i] * F2) + jw[i] * ptz[b]);
        }
        for (int j = 1; j < NB; ++j) {
            M[2 + j][2 + j] = M[2 + j][2 + j] + (R)Mdl::armature(j);
            const R x = q[2 + j], v = qd[2 + j];
            R t = -((R)Mdl::stiffness(j) * x) - (R)Mdl::damping(j) * v;
            if (!limit_mj) {
                const R viol = x - rl_clamp(x, (R)Mdl::lo(j), (R)Mdl::hi(j));
                const R damp = (viol != (R)0) ? (R)Mdl::limit_b() * v : (R)0;
                t = (t - (R)Mdl::limit_k() * viol) - damp;
            }
            Q[2 + j] = Q[2 + j] + (t + act_j[j]);
        }
        R Dg[NV], Di[NV], acc[NV];
        factor<R>(M, Dg, Di);
        solve<R>(M, Di, Q, acc);
        // ---- active constraint rows ------------------------------------------------------------------------------------
        R J[KMAX][NV], pos[KMAX];
        int cls[KMAX];                       // 0 = limit, 1 = contact
        int K = 0;
        if (limit_mj) {
            for (int j = 1; j < NB; ++j) {
                const R dlo = q[2 + j] - (R)Mdl::lo(j), dhi = (R)Mdl::hi(j) - q[2 + j];
                const R mg = (R)Par::limit().margin;
                if (dlo < mg || dhi < mg) {
                    const R sg = dlo < mg ? (R)1 : (R)-1;
                    for (int i = 0; i < NV; ++i) J[K][i] = (R)0;
                    J[K][2 + j] = sg;
                    pos[K] = dlo < mg ? dlo : dhi;
                    cls[K] = 0;
                    ++K;
                }
            }
        }
        if (contact_mj) {
            int nc = 0;
            for (int c = 0; c < NC && nc < MAXC; ++c) {
                const int b = Mdl::cbody(c);
                const R lx = (R)Mdl::cpx(c), ly = (R)Mdl::cpy(c), rad = (R)Mdl::crad(c);
                const R rx = k.cs[b] * lx - k.sn[b] * ly;
                const R ry = k.sn[b] * lx + k.cs[b] * ly;
                const R dist = (q[0] + k.ax[b] + rx) - rad;
                if (dist < (R)Par::contact().margin) {
                    R jn[NV], jt[NV];
                    point_jac<R>(k, b, k.ax[b] + rx - rad, k.ay[b] + ry, jn, jt);     // the lowest point of the sphere
                    const R mu = (R)Mdl::cmu(c);
                    for (int i = 0; i < NV; ++i) {
                        J[K][i] = jn[i] + mu * jt[i];
                        J[K + 1][i] = jn[i] - mu * jt[i];
                    }
                    pos[K] = dist; pos[K + 1] = dist;
                    cls[K] = 1; cls[K + 1] = 1;
                    K += 2;
                    ++nc;
                }
            }
        }
        R f[KMAX];
        if (K > 0) {
            R W[KMAX][NV];                   // rows of M^-1 J^T
            R A[KMAX][KMAX], g[KMAX], Rg[KMAX];
            for (int r = 0; r < K; ++r) solve<R>(M, Di, J[r], W[r]);
            for (int r = 0; r < K; ++r) {
                for (int c = 0; c < K; ++c) {
                    R s = (R)0;
                    for (int i = 0; i < NV; ++i) s = s + J[r][i] * W[c][i];
                    A[r][c] = s;
                }
                R jv = (R)0, ja = (R)0;
                for (int i = 0; i < NV; ++i) {
                    jv = jv + J[r][i] * qd[i];
                    ja = ja + J[r][i] * acc[i];
                }
                const MjcSol p = cls[r] ? Par::contact() : Par::limit();
                const R rr = pos[r] - (R)p.margin;
                const R d = impedance<R>(rr, p);
                const R bb = (R)(2.0 / (p.dmax * p.timeconst));
                const R kk = d * (R)(1.0 / (p.dmax * p.dmax * p.timeconst * p.timeconst * p.dampratio * p.dampratio));
                const R aref = -(bb * jv) - kk * rr;
                Rg[r] = ((R)1 - d) * rl_recip_normal(d) * A[r][r];
                g[r] = ja - aref;
                f[r] = (R)0;
            }
            for (int sweep = 0; sweep < SWEEPS; ++sweep) {
                for (int r = 0; r < K; ++r) {
                    R res = g[r] + Rg[r] * f[r];
                    for (int c = 0; c < K; ++c) res = res + A[r][c] * f[c];
                    const R next = f[r] - res * rl_recip_normal(A[r][r] + Rg[r]);
                    f[r] = rl_max(next, (R)0);
                }
            }
            for (int r = 0; r < K; ++r)
                for (int i = 0; i < NV; ++i) acc[i] = acc[i] + f[r] * W[r][i];
        }
        if (qfrc) {
            for (int i = 0; i < NV; ++i) {
                R s = (R)0;
                for (int r = 0; r < K; ++r) s = s + f[r] * J[r][i];
                qfrc[i] = s;
            }
        }
        for (int i = 0; i < NV; ++i) {
            qd[i] = qd[i] + h * acc[i];
            q[i] = q[i] + h * qd[i];
        }
    }

    template <typename R>
    RL_HD static void advance(R* q, R* qd, const R* act_j, R h, int n, int flags, R* qfrc = nullptr) {
        const bool lm = (flags & CFG_LIMIT_MUJOCO) != 0, cm = (flags & CFG_CONTACT_MUJOCO) != 0;
        for (int it = 0; it < n; ++it) substep<R>(q, qd, act_j, h, lm, cm, it + 1 == n ? qfrc : nullptr);
    }
};

// An env whose step ALWAYS runs the soft-constraint sub-steps: the kernels are instantiated for MjcEnv<HalfCheetah> etc. and
// launched when one of the two flags is set (env_kernels.hip), so that the default instantiations contain no trace of this
// path -- a call to the out-of-line sub-step inside vecenv_step_kernel<HalfCheetah> cost the penalty path a fifth of its
// speed (registers reserved for the callee).  The host build branches at run time inside Env::step instead.
template <class Base>
struct MjcEnv : Base {
    static constexpr bool IS_MJC = true;
    template <typename R>
    RL_HD static void step(R* s, const R* a, int normalize, R* obs, R& reward, bool& done,
                           const StepOpts<R>& o = Base::template default_opts<R>()) {
        Base::template step_model<R, true>(s, a, normalize, obs, reward, done, o);
    }
};
template <class E, class = void> struct has_mjc : std::false_type {};
template <class E> struct has_mjc<E, std::enable_if_t<E::HAS_MJC>> : std::true_type {};
template <class E, class = void> struct is_mjc_env : std::false_type {};
template <class E> struct is_mjc_env<E, std::enable_if_t<E::IS_MJC>> : std::true_type {};

}  // namespace rl
