// dyn_swimmer_chain.h -- the sub-step of the 3-link swimmer chain, written twice with IDENTICAL arithmetic (the same
// products, sums and fused multiply-adds in the same association; the lane-group program evaluates x / y pairs as
// two-component vectors):
//
//   swim_substep_scalar : one env per lane / host thread; the three bodies are unrolled in one instruction stream.
//                         Used by the host oracle build, by the per-step VecEnv kernels and as the definition of the
//                         env's arithmetic.
//   swim_substep_quad   : four lanes per env (lane role b = body 0, 1, 2; lane 3 idles with zero constants).  Every
//                         lane runs the same instruction stream on its own body, values cross lanes by
//                         quad-permute moves, the 3x3 solve is replicated.  Used by the fused rollout, where a lone
//                         wavefront per SIMD pays 4 cycles per issued instruction: the per-env instruction stream
//                         shrinks from ~260 to 89 per sub-step and four times as many wavefronts share the work.
//
// Bit-exactness contract: every value the quad program computes is computed by the scalar program with the same
// expression (same operands, same association, same statement boundaries -- the front-end contracts a*b+c only inside
// one expression), apart from terms that are exactly +-0 (lane 3's contributions, masked exchanges), which the scalar
// program omits; x + (+-0) == x for every non-zero x.  tests/ replay the quad program on the host (a lock-step emulator
// of the four lanes) against the scalar program bit for bit, and the GPU rollout against the host build.
//
// Physics: dyn_planar.h's absolute-angle formulation (forward_dynamics_abs) for the collinear chain -- all As / Ss
// coefficients vanish, the Schur complement of the translations is Sc_kl * cos(phi_l - phi_k) -- with the fluid model of
// dyn_swimmer.h.  Integrated variables: root position / velocity, absolute body rates om_b, joint angles th_b
// (th_0 = root angle) and the carried (sin, cos) of the absolute body angles.
#pragma once
#include "dyn_planar.h"

namespace rl {

template <class Mdl>
struct SwimChain {
    using Tree = PlanarTree<Mdl>;
    static_assert(Mdl::NB == 3, "three-link chain");
    static constexpr double INV_M = 1.0 / Tree::total_mass();
    static constexpr double jxo(int b) { return b < 2 ? Mdl::jx(b + 1) : 0.0; }      // child joint offset along own x
    static constexpr double cxb(int b) { return b < 3 ? Mdl::cx(b) : 0.0; }
    static constexpr double db(int b) { return b < 3 ? Tree::Dvec(b).x : 0.0; }
    // every body b couples to its two cyclic partners p = b + 1, q = b + 2 (mod 3); pair constants are symmetric
    static constexpr int nxt(int b) { return (b + 1) % 3; }
    static constexpr int nx2(int b) { return (b + 2) % 3; }
    static constexpr double scc(int k, int l) { return k < l ? Tree::Sc(k, l) : Tree::Sc(l, k); }
    static constexpr double acc(int k, int l) { return k < l ? Tree::Ac(k, l) : Tree::Ac(l, k); }
    static constexpr double sdiag(int b) { return Tree::Sc(b, b) + Tree::Kc(b, b); }
    static constexpr double dpq(int b) { return sdiag(nxt(b)) * sdiag(nx2(b)); }

    // ---- leaf expressions shared verbatim by both programs -----------------------------------------------------
    template <typename R>
    RL_HD static void fluid(R cs, R sn, R vpx, R vpy, R om, R visc_lin, R drag_ax, R drag_perp, R visc_ang, R drag_ang,
                            R& Fx, R& Fy, R& tz, R& ft_out) {
        const R vl = cs * vpx + sn * vpy;
        const R vt = -sn * vpx + cs * vpy;              // (the lane-group program rotates with the pair (-sn, cs))
        const R fl = -(vl * (visc_lin + drag_ax * rl_abs(vl)));
        const R ft = -(vt * (visc_lin + drag_perp * rl_abs(vt)));
        Fx = cs * fl - sn * ft;
        Fy = sn * fl + cs * ft;
        tz = -(om * (visc_ang + drag_ang * rl_abs(om)));
        ft_out = ft;      // moment of the fluid force about the body origin = cx_b * ft (the axial part has no arm)
    }
    // penalty joint-limit torque + actuation of one hinge
    template <typename R>
    RL_HD static R joint_torque(R th, R thd, R act, R lim_k, R lim_b) {
        const R viol = th - rl_clamp_finite(th, (R)Mdl::lo(1), (R)Mdl::hi(1));
        const R b_on = (viol != (R)0) ? lim_b : (R)0;         // the damper acts only beyond the range
        const R t = act - lim_k * viol;                       // one fused multiply-add
        return t - b_on * thd;                                // and another
    }
    // coupling of body b (cb, sb) to a partner x (cx, sx):  S = Sc_bx cos(phi_x - phi_b)  (bitwise symmetric in b <-> x:
    // the product and the fused product commute),  t = Ac_bx sin(phi_x - phi_b)  (rhs_b += w_x^2 t)
    template <typename R>
    RL_HD static void couple(R cb, R sb, R cx, R sx, R sc, R ac, R& S, R& t) {
        const R cd = cb * cx + sb * sx;
        const R sd = -sb * cx + cb * sx;
        S = sc * cd;
        t = ac * sd;
    }
    // row b of the symmetric 3x3 solve in the cyclic order (b, p, q): cofactors of row b are the cross product of rows
    // p and q, every body uses its own expansion of the determinant (equal up to rounding), so the four-lane
    // program needs no replicated adjugate:  x_b = cof . (r_b, r_p, r_q) * (1 / (row_b . cof))
    template <typename R>
    RL_HD static R solve_row(R d_b, R d_pq, R d_p, R d_q, R Sbp, R Sbq, R Spq, R rb, R rp, R rq) {
        const R c0 = d_pq - Spq * Spq;
        const R c1 = Spq * Sbq - Sbp * d_q;
        const R c2 = Sbp * Spq - d_p * Sbq;
        const R det = d_b * c0 + (Sbp * c1 + Sbq * c2);
        const R num = c0 * rb + (c1 * rp + c2 * rq);
        return num * rl_recip_normal(det);
    }

    // ---- joint limits by MuJoCo's soft-constraint model (SwimmerEnv(limit_model="mujoco"), scalar program only) -------
    // vendor/mujoco_models/swimmer.xml:31,34 put solreflimit = (timeconst 0.02, dampratio 1) and solimplimit = (dmin 0,
    // dmax 0.8, width 0.03) on the two hinges.  MuJoCo's documented model (Computation chapter: "Constraint model",
    // "Solver parameters"; the binary itself, version 1.31, is absent -- this follows the published text):
    //   * a limit is active when dist = q - lo (lower) or hi - q (upper) is negative; its Jacobian row J is +-1 on the hinge;
    //   * reference acceleration  a_ref = -b (J v) - k dist,  b = 2 / (dmax timeconst),
    //     k = d(dist) / (dmax^2 timeconst^2 dampratio^2);
    //   * impedance d(dist) in [dmin, dmax]: x = min(|dist| / width, 1), y = 2 x^2 (x < 1/2), 1 - 2 (1 - x)^2 otherwise
    //     (the fixed sigmoid of the three-number solimp; MuJoCo 2's five-number form names it midpoint 0.5, power 2),
    //     d = dmin + y (dmax - dmin), kept inside [mjMINIMP, mjMAXIMP] = [1e-4, 0.9999];
    //   * constraint forces f >= 0 minimise  1/2 f^T (A + R) f + f^T (a0 - a_ref),  A = J M^-1 J^T, a0 = J qacc_unconstrained,
    //     R = diag((1 - d_i) / d_i A_ii);  qacc = qacc_unconstrained + M^-1 J^T f.
    // MuJoCo reaches the minimum by projected Gauss-Seidel (swimmer.xml: 1000 iterations); with at most two rows it is
    // found exactly by enumerating the active sets.  In the absolute-angle coordinates of this file the translations are
    // already eliminated, so M^-1 restricted to the hinges is the inverse of the 3 x 3 system solved below and
    // J = e_j - e_{j-1}.
    static constexpr double MJ_TIMECONST = 0.02, MJ_DAMPRATIO = 1.0, MJ_DMIN = 0.0, MJ_DMAX = 0.8, MJ_WIDTH = 0.03;
    static constexpr double MJ_B = 2.0 / (MJ_DMAX * MJ_TIMECONST);
    static constexpr double MJ_KD = 1.0 / (MJ_DMAX * MJ_DMAX * MJ_TIMECONST * MJ_TIMECONST * MJ_DAMPRATIO * MJ_DAMPRATIO);
    template <typename R>
    RL_HD static R mj_impedance(R dist) {
        const R x0 = rl_abs(dist) * (R)(1.0 / MJ_WIDTH);
        const R x = x0 < (R)1 ? x0 : (R)1;
        const R omx = (R)1 - x;
        const R y = x < (R)0.5 ? (R)2 * (x * x) : (R)1 - (R)2 * (omx * omx);
        const R d = (R)MJ_DMIN + y * (R)(MJ_DMAX - MJ_DMIN);
        return rl_clamp(d, (R)1e-4, (R)0.9999);
    }
    // one hinge: sign (+1 lower limit active, -1 upper, 0 none), dist (< 0 when active)
    template <typename R>
    RL_HD static void mj_limit_state(R th, R& sign, R& dist) {
        const R dlo = th - (R)Mdl::lo(1), dhi = (R)Mdl::hi(1) - th;
        sign = dlo < (R)0 ? (R)1 : (dhi < (R)0 ? (R)-1 : (R)0);
        dist = dlo < (R)0 ? dlo : (dhi < (R)0 ? dhi : (R)0);
    }

    // ---- scalar program ---------------------------------------------------------------------------------------------
    // r = [rx, ry, vx, vy]; per body b: cs, sn, om (absolute rate), th (th[0] = root angle, th[1..2] = hinge angles);
    // act[b] = motor torque of hinge b (act[0] unused)
    // MJ: joint limits by the soft-constraint model above instead of the penalty torque (everything else unchanged)
    template <typename R, bool MJ = false>
    RL_HD static void substep_scalar(R* r, R* cs, R* sn, R* om, R* th, const R* act, R h) {
        const R VL = (R)Mdl::VISC_LIN, DAX = (R)Mdl::DRAG_AX, DPERP = (R)Mdl::DRAG_PERP, VA = (R)Mdl::VISC_ANG,
                DANG = (R)Mdl::DRAG_ANG;
        const R LK = (R)Mdl::limit_k(), LB = (R)Mdl::limit_b();
        R osn[3], ocs[3], wlx[3], wly[3];
        RL_UNROLL
        for (int b = 0; b < 3; ++b) {
            osn[b] = om[b] * sn[b];
            ocs[b] = om[b] * cs[b];
            wlx[b] = -(osn[b] * (R)jxo(b));
            wly[b] = ocs[b] * (R)jxo(b);
        }
        R vax[3], vay[3];
        vax[0] = r[2];                       vay[0] = r[3];
        vax[1] = r[2] + wlx[0];              vay[1] = r[3] + wly[0];
        vax[2] = (r[2] + wlx[1]) + wlx[0];   vay[2] = (r[3] + wly[1]) + wly[0];
        R Fx[3], Fy[3], tz[3], ft[3];
        RL_UNROLL
        for (int b = 0; b < 3; ++b) {
            const R vpx = vax[b] - osn[b] * (R)cxb(b);
            const R vpy = vay[b] + ocs[b] * (R)cxb(b);
            fluid(cs[b], sn[b], vpx, vpy, om[b], VL, DAX, DPERP, VA, DANG, Fx[b], Fy[b], tz[b], ft[b]);
        }
        R tau[3];
        tau[0] = (R)0;
        if constexpr (MJ) {
            tau[1] = act[1];                                   // the limits act as constraint forces after the solve
            tau[2] = act[2];
        } else {
            tau[1] = joint_torque(th[1], om[1] - om[0], act[1], LK, LB);
            tau[2] = joint_torque(th[2], om[2] - om[1], act[2], LK, LB);
        }
        // subtree force sums and generalised forces on the absolute angles
        R Fsx[3], Fsy[3], Q[3];
        Fsx[2] = Fx[2];                      Fsy[2] = Fy[2];
        Fsx[1] = Fx[1] + Fx[2];              Fsy[1] = Fy[1] + Fy[2];
        Fsx[0] = (Fx[0] + Fx[1]) + Fx[2];    Fsy[0] = (Fy[0] + Fy[1]) + Fy[2];
        Q[0] = (((R)cxb(0) * ft[0] + tz[0]) + (R)jxo(0) * (cs[0] * Fsy[1] - sn[0] * Fsx[1])) - tau[1];
        Q[1] = (((R)cxb(1) * ft[1] + tz[1]) + (R)jxo(1) * (cs[1] * Fsy[2] - sn[1] * Fsx[2])) + (tau[1] - tau[2]);
        Q[2] = ((R)cxb(2) * ft[2] + tz[2]) + tau[2];
        // translation coupling and centripetal terms
        R Gx[3], Gy[3], w2[3], fwx[3], fwy[3];
        RL_UNROLL
        for (int b = 0; b < 3; ++b) {
            const R DB = (R)db(b);
            Gx[b] = cs[b] * DB;
            Gy[b] = sn[b] * DB;
            w2[b] = om[b] * om[b];
            fwx[b] = Fx[b] + w2[b] * Gx[b];       // force + centripetal term of body b, one fused multiply-add
            fwy[b] = Fy[b] + w2[b] * Gy[b];
        }
        // their sum over the bodies, UNSCALED: the 1 / M of the translation block is applied where the sum is used
        // (the lane-group program folds this sum with one butterfly; its fourth lane adds an exact zero)
        const R sfx = (fwx[0] + fwx[1]) + fwx[2];
        const R sfy = (fwy[0] + fwy[1]) + fwy[2];
        R bq[3];
        RL_UNROLL
        for (int b = 0; b < 3; ++b) {
            const R cross = Gx[b] * sfy - Gy[b] * sfx;
            bq[b] = Q[b] - (R)INV_M * cross;
        }
        // coupling to the cyclic partners and the rows of the 3x3 solve
        // one coupling per pair: body b evaluates (b, p); its coupling to q is pair (q, b) seen from the other side --
        // S is symmetric bit for bit, the sine term changes sign
        R Sbp[3], tp[3], rb[3];
        RL_UNROLL
        for (int b = 0; b < 3; ++b) {
            const int p = nxt(b);
            couple(cs[b], sn[b], cs[p], sn[p], (R)scc(b, p), (R)acc(b, p), Sbp[b], tp[b]);
        }
        // the centripetal term of pair (q, b) is formed by its owner q (own rate^2 x own pair: one product, one rounding)
        R zc[3];
        RL_UNROLL
        for (int b = 0; b < 3; ++b) zc[b] = w2[b] * tp[b];
        RL_UNROLL
        for (int b = 0; b < 3; ++b) {
            const int p = nxt(b), q = nx2(b);
            rb[b] = (bq[b] + w2[p] * tp[b]) - zc[q];
        }
        R thb[3];
        RL_UNROLL
        for (int b = 0; b < 3; ++b) {
            const int p = nxt(b), q = nx2(b);
            thb[b] = solve_row((R)sdiag(b), (R)dpq(b), (R)sdiag(p), (R)sdiag(q), Sbp[b], Sbp[q], Sbp[p], rb[b], rb[p],
                               rb[q]);
        }
        if constexpr (MJ) {
            // thb = the unconstrained accelerations of the absolute angles.  Limit rows of hinge j (= 1, 2): J = sg_j (e_j - e_{j-1})
            R sg[2], dist[2];
            mj_limit_state(th[1], sg[0], dist[0]);
            mj_limit_state(th[2], sg[1], dist[1]);
            if (sg[0] != (R)0 || sg[1] != (R)0) {
                // X_c = S^-1 J_c^T (three more right-hand sides per row through the same cofactor solve)
                R X[2][3];
                RL_UNROLL
                for (int c = 0; c < 2; ++c) {
                    R e[3];
                    e[0] = c == 0 ? -sg[0] : (R)0;
                    e[1] = c == 0 ? sg[0] : -sg[1];
                    e[2] = c == 0 ? (R)0 : sg[1];
                    RL_UNROLL
                    for (int b = 0; b < 3; ++b) {
                        const int p = nxt(b), q = nx2(b);
                        X[c][b] = solve_row((R)sdiag(b), (R)dpq(b), (R)sdiag(p), (R)sdiag(q), Sbp[b], Sbp[q], Sbp[p], e[b], e[p],
                                            e[q]);
                    }
                }
                // A = J S^-1 J^T, a0 = J thb, a_ref, impedances
                const R A00 = sg[0] * (X[0][1] - X[0][0]);
                const R A01 = sg[0] * (X[1][1] - X[1][0]);
                const R A11 = sg[1] * (X[1][2] - X[1][1]);
                R c_[2], D[2];
                RL_UNROLL
                for (int c = 0; c < 2; ++c) {
                    const R a0 = sg[c] * (thb[c + 1] - thb[c]);
                    const R v = sg[c] * (om[c + 1] - om[c]);
                    const R d = mj_impedance(dist[c]);
                    const R aref = -((R)MJ_B * v) - ((R)MJ_KD * d) * dist[c];
                    c_[c] = aref - a0;                          // the solver minimises 1/2 f (A + R) f - f . c_
                    D[c] = (c == 0 ? A00 : A11) / d;            // A_ii + R_ii = A_ii / d_i
                }
                // exact minimum over f >= 0 of the (strictly convex) two-row problem, by active set
                R f0 = (R)0, f1 = (R)0;
                const bool on0 = sg[0] != (R)0, on1 = sg[1] != (R)0;
                if (on0 && on1) {
                    const R det = D[0] * D[1] - A01 * A01;
                    const R g0 = (c_[0] * D[1] - A01 * c_[1]) / det;
                    const R g1 = (D[0] * c_[1] - A01 * c_[0]) / det;
                    if (g0 >= (R)0 && g1 >= (R)0) {
                        f0 = g0; f1 = g1;
                    } else {
                        const R s0 = c_[0] / D[0], s1 = c_[1] / D[1];
                        if (s0 > (R)0 && A01 * s0 - c_[1] >= (R)0) f0 = s0;              // row 1 inactive: its gradient >= 0
                        else if (s1 > (R)0 && A01 * s1 - c_[0] >= (R)0) f1 = s1;
                    }
                } else if (on0) {
                    const R s0 = c_[0] / D[0];
                    f0 = s0 > (R)0 ? s0 : (R)0;
                } else {
                    const R s1 = c_[1] / D[1];
                    f1 = s1 > (R)0 ? s1 : (R)0;
                }
                RL_UNROLL
                for (int b = 0; b < 3; ++b) thb[b] = thb[b] + (X[0][b] * f0 + X[1][b] * f1);
            }
        }
        // translations
        R cxp[3], cyp[3];
        RL_UNROLL
        for (int b = 0; b < 3; ++b) {
            cxp[b] = -(Gy[b] * thb[b]);
            cyp[b] = Gx[b] * thb[b];
        }
        const R sx = (cxp[0] + cxp[1]) + cxp[2];
        const R sy = (cyp[0] + cyp[1]) + cyp[2];
        // root acceleration (sf - s) / M, integrated with the loop-invariant h / M
        const R hm = h * (R)INV_M;
        const R dx_ = sfx - sx;
        const R dy_ = sfy - sy;
        r[2] = r[2] + hm * dx_;
        r[3] = r[3] + hm * dy_;
        r[0] = r[0] + h * r[2];
        r[1] = r[1] + h * r[3];
        RL_UNROLL
        for (int b = 0; b < 3; ++b) om[b] = om[b] + h * thb[b];
        th[0] = th[0] + h * om[0];
        th[1] = th[1] + h * (om[1] - om[0]);
        th[2] = th[2] + h * (om[2] - om[1]);
        RL_UNROLL
        for (int b = 0; b < 3; ++b) rl_rotate_tiny(sn[b], cs[b], h * om[b]);
    }

    // ---- quad program ------------------------------------------------------------------------------------------------
    // The x / y pairs of the planar dynamics travel as two-component vectors (rl_math.h V2): one v_pk_{mul,fma,add}_f32
    // per pair on gfx950 (5 cycles for a lone wavefront against 2 x 4 for the scalar forms), operand swizzles and
    // broadcasts ride in op_sel.  Rotations use the body direction A = (cs, sn) and its perpendicular Ap = (-sn, cs), so
    // that no component needs a separate sign flip:  R(a, b) = A a + Ap b,  R^-1(a, b) = Ap.yx a + A.yx b.
    // per-lane constants, selected by the lane's role b = lane & 3 (role 3: all zero)
    template <typename R>
    struct LaneConst {
        V2<R> kj;                    // (-jxo, jxo): child joint offset, signed for (wlx, wly)
        V2<R> kc;                    // (-cxb, cxb): centre offset, signed for (vpx, vpy)
        V2<R> ksa;                   // (Sc, Ac) coupling constants to the partner p = b + 1 (mod 3)
        V2<R> kdqp;                  // (d_q, d_p)
        V2<R> kdrag;                 // (drag_ax, drag_perp)
        R jxo, cxb, db, visc_lin, visc_ang, drag_ang;
        R lim_k, lim_b;              // joint-limit penalty of the hinge that carries body b (roles 0 and 3: none)
        R d_b, d_pq;                 // diagonal of the 3x3 system in the cyclic order (role 3: identity)
        int b;
    };
    template <typename R>
    RL_HD static LaneConst<R> lane_const(int b) {
        LaneConst<R> c;
        c.b = b;
        c.jxo = (R)(b == 0 ? jxo(0) : b == 1 ? jxo(1) : 0.0);
        c.cxb = (R)(b == 0 ? cxb(0) : b == 1 ? cxb(1) : b == 2 ? cxb(2) : 0.0);
        c.db = (R)(b == 0 ? db(0) : b == 1 ? db(1) : b == 2 ? db(2) : 0.0);
        c.kj = V2<R>{-c.jxo, c.jxo};
        c.kc = V2<R>{-c.cxb, c.cxb};
        const bool body = b < 3;
        c.visc_lin = body ? (R)Mdl::VISC_LIN : (R)0;
        c.kdrag = V2<R>{body ? (R)Mdl::DRAG_AX : (R)0, body ? (R)Mdl::DRAG_PERP : (R)0};
        c.visc_ang = body ? (R)Mdl::VISC_ANG : (R)0;
        c.drag_ang = body ? (R)Mdl::DRAG_ANG : (R)0;
        const bool hinge = (b == 1 || b == 2);
        c.lim_k = hinge ? (R)Mdl::limit_k() : (R)0;
        c.lim_b = hinge ? (R)Mdl::limit_b() : (R)0;
        c.ksa = V2<R>{(R)(b == 0 ? scc(0, 1) : b == 1 ? scc(1, 2) : b == 2 ? scc(2, 0) : 0.0),
                      (R)(b == 0 ? acc(0, 1) : b == 1 ? acc(1, 2) : b == 2 ? acc(2, 0) : 0.0)};
        c.d_b = (R)(b == 0 ? sdiag(0) : b == 1 ? sdiag(1) : b == 2 ? sdiag(2) : 1.0);
        c.d_pq = (R)(b == 0 ? dpq(0) : b == 1 ? dpq(1) : b == 2 ? dpq(2) : 1.0);
        c.kdqp = V2<R>{(R)(b == 0 ? sdiag(2) : b == 1 ? sdiag(0) : b == 2 ? sdiag(1) : 0.0),
                       (R)(b == 0 ? sdiag(1) : b == 1 ? sdiag(2) : b == 2 ? sdiag(0) : 0.0)};
        return c;
    }
    template <typename R>
    struct Lane {
        V2<R> A, Ap;         // own body: (cs, sn) and (-sn, cs)   (role 3: A = (1, 0))
        V2<R> v, r;          // root velocity / position, replicated on the four lanes
        R om, th;            // absolute rate (role 3: 0, and stays 0 -- the other lanes read it as their zero), joint angle
        R qd;                // om - parent's om: the joint rate, carried from the end of the previous sub-step
        RL_HD void set_direction(R cs, R sn) { A = V2<R>{cs, sn}; Ap = V2<R>{-sn, cs}; }
    };

    // quad_perm controls: lane i of the quad reads lane P[i]
    static constexpr int QP(int a, int b, int c, int d) { return a | (b << 2) | (c << 4) | (d << 6); }
    // role 3 holds exact zeros for everything a neighbour may fetch, so "the parent of body 0" and "nothing" are lane 3
    static constexpr int PAR1 = QP(3, 0, 1, 2), PAR2 = QP(3, 3, 0, 3);       // parent, grandparent (or the zero lane)
    static constexpr int SHL1 = QP(1, 2, 3, 3), SHL2 = QP(2, 3, 3, 3);       // child, grandchild (or the zero lane)
    static constexpr int NX1 = QP(1, 2, 0, 3), NX2 = QP(2, 0, 1, 3);         // cyclic partners p, q of body = lane
    static constexpr int SW1 = QP(1, 0, 3, 2), SW2 = QP(2, 3, 0, 1);         // butterfly

    template <typename R, class X>
    RL_HD static R quad_sum(X& x, R v) {
        const R v1 = v + x.template qp<SW1>(v);
        return v1 + x.template qp<SW2>(v1);
    }
    template <int CTRL, typename R, class X>
    RL_HD static V2<R> qp2(X& x, V2<R> v) {
        return V2<R>{x.template qp<CTRL>(v.x), x.template qp<CTRL>(v.y)};
    }

    // the joint rate a lane carries: own absolute rate - parent's
    template <typename R, class X>
    RL_HD static R joint_rate(X& x, R om) { return om - x.template qp<PAR1>(om); }

    // X: exchange context, x.template qp<CTRL>(v) = value of v in the lane selected by CTRL
    template <typename R, class X>
    RL_HD static void substep_quad(X& x, const LaneConst<R>& c, Lane<R>& s, R act, R h) {
        using P = V2<R>;
        const P A = s.A, Ap = s.Ap;
        const P oc = s.om * A;                                   // (ocs, osn)
        const P wl = oc.yx * c.kj;                               // (wlx, wly) = (-(osn jxo), ocs jxo)
        // cross-lane sums stay per component: each is one v_add_f32 with the quad-permute fused in
        const P va = P{(s.v.x + x.template qp<PAR1>(wl.x)) + x.template qp<PAR2>(wl.x),
                       (s.v.y + x.template qp<PAR1>(wl.y)) + x.template qp<PAR2>(wl.y)};
        const P vp = va + oc.yx * c.kc;                          // (vax - osn cxb, vay + ocs cxb)
        // body-frame velocity (vl, vt) = R^-1 vp, fluid force (fl, ft) along / across the body, world force F = R f
        const P lt = Ap.yx * vp.xx + A.yx * vp.yy;
        const P m = P{c.visc_lin + c.kdrag.x * rl_abs(lt.x), c.visc_lin + c.kdrag.y * rl_abs(lt.y)};
        const P f = -(lt * m);
        const P F = A * f.xx + Ap * f.yy;
        const R tz = -(s.om * (c.visc_ang + c.drag_ang * rl_abs(s.om)));
        const R tau = joint_torque(s.th, s.qd, act, c.lim_k, c.lim_b);
        const R taun = x.template qp<SHL1>(tau);
        // force on the subtree hanging off this body's child joint = child's + grandchild's (the zero lane beyond)
        const P fn = P{x.template qp<SHL1>(F.x) + x.template qp<SHL2>(F.x),
                       x.template qp<SHL1>(F.y) + x.template qp<SHL2>(F.y)};
        const R Q = ((c.cxb * f.y + tz) + c.jxo * (A.x * fn.y - A.y * fn.x)) + (tau - taun);
        const P G = A * c.db;                                    // (Gx, Gy)
        const P Gp = Ap * c.db;                                  // (-Gy, Gx)
        const R w2 = s.om * s.om;
        const P fw = F + w2 * G;
        const P sf = P{quad_sum(x, fw.x), quad_sum(x, fw.y)};
        const R cross = G.x * sf.y - G.y * sf.x;
        const R bq = Q - (R)INV_M * cross;
        // coupling to the cyclic partner p; the pair (b, q) is partner q's own pair seen from the other side
        const P Pn = qp2<NX1>(x, A);
        const R w2p = x.template qp<NX1>(w2);
        const P cds = Ap.yx * Pn.xx + A.yx * Pn.yy;              // (cos, sin)(phi_p - phi_b)
        const R Sbp = c.ksa.x * cds.x, tp = c.ksa.y * cds.y;     // (S_bp, t_bp)
        const P SS = P{Sbp, x.template qp<NX2>(Sbp)};            // (S_bp, S_bq)
        const R zc = w2 * tp;                                    // own rate^2 x own pair, fetched by the pair's other end
        const R rb = (bq + w2p * tp) - x.template qp<NX2>(zc);
        const R Spq = x.template qp<NX1>(Sbp);
        const R rp = x.template qp<NX1>(rb), rq = x.template qp<NX2>(rb);
        // row b of the 3x3 solve (solve_row): the two cofactors and (det, num) evaluated side by side
        const R c0 = c.d_pq - Spq * Spq;
        const P c12 = Spq * SS.yx - SS * c.kdqp;                 // (c1, c2)
        // the determinant and its reciprocal do not wait for the right-hand sides: they fill the exchange gaps of rb
        const R det = c.d_b * c0 + (SS.x * c12.x + SS.y * c12.y);
        const R num = c0 * rb + (c12.x * rp + c12.y * rq);
        const R thb = num * rl_recip_normal(det);
        const P cp = Gp * thb;                                   // (cxp, cyp)
        const P s2 = P{quad_sum(x, cp.x), quad_sum(x, cp.y)};
        const R hm = h * (R)INV_M;
        const P d = sf - s2;
        s.v = s.v + hm * d;
        s.r = s.r + h * s.v;
        s.om = s.om + h * thb;
        s.qd = joint_rate(x, s.om);
        s.th = s.th + h * s.qd;
        // rl_rotate_tiny on the pair
        const R dth = h * s.om;
        R sd, cd;
        rl_tiny_sincos(dth, sd, cd);
        s.A = A * cd + Ap * sd;
        s.Ap = P{-s.A.y, s.A.x};
    }
};

}  // namespace rl
