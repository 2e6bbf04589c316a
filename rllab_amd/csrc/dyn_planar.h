// dyn_planar.h -- forward dynamics of a planar articulated tree with a free-floating
// root, single source for the gfx950 kernels and the host oracle build.
//
// This is the engine's stand-in for the part of MuJoCo 1.31 (proprietary binary,
// absent from the reference tree: rllab/mujoco_py/mjlib.py:10, .gitignore:21) that
// the reference reaches through MjModel.step / forward
// (rllab/mujoco_py/mjcore.py:46-84, rllab/envs/mujoco/mujoco_env.py:184-191) for
// the planar models named by BASELINE.json: vendor/mujoco_models/swimmer.xml and
// half_cheetah.xml.  Structure follows MuJoCo's documented pipeline: joint-space
// inertia M(q) (composite bodies), bias forces, passive forces (spring, damper,
// fluid), actuation, soft limit forces, qacc = M^-1 * tau, semi-implicit Euler
// (qvel += h*qacc; qpos += h*qvel).  "-style": constants come from the MJCF files,
// the constraint model (joint limits, contacts) is a documented spring-damper
// penalty instead of MuJoCo's convex solver -- see DESIGN.md.
//
// Coordinates: q = [root x, root y, root angle, hinge_1 .. hinge_{NB-1}] (nv = NB+2);
// body i >= 1 hangs from body parent(i) by a hinge at its own frame origin, which
// sits at joint(i) in the parent's frame.  All angles are about the plane normal.
//
// Model traits `Mdl` (all static constexpr functions, i = body index):
//   NB, parent(i), jx(i), jy(i)   hinge anchor in parent frame (root: unused)
//   cx(i), cy(i)                  centre of mass in own frame
//   mass(i), inertia(i)           about the COM
//   armature(i), damping(i), stiffness(i)   hinge i (i >= 1; root entries unused)
//   limited(i), lo(i), hi(i)      joint range
//   gx(), gy()                    gravity in the plane
// plus a static `external(...)` hook adding per-body world forces / torques
// (fluid drag, ground contact).
#pragma once
#include <type_traits>
#include "rl_math.h"

namespace rl {

// Compile-time loops: the body sees the index as a constant expression, so model constants
// (joint offsets, COM offsets, gravity, armature ...) can be tested with `if constexpr` and
// structurally-zero terms vanish from the instruction stream.  IEEE arithmetic forbids the
// optimiser from folding x*0 or x+0 itself, and one env is one dependent instruction chain
// per wavefront, so every such term costs issue slots on the critical path.
template <int I, int N, class F>
RL_HD void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
template <int I, int N, class F>   // I-1, I-2, ..., N
RL_HD void static_for_down(F&& f) {
    if constexpr (I > N) {
        f(std::integral_constant<int, I - 1>{});
        static_for_down<I - 1, N>(f);
    }
}

// (ox, oy) = R(cs, sn) * (CX, CY) for compile-time constants CX, CY
#define RL_ROTC(cs_, sn_, CX, CY, ox, oy)                 \
    do {                                                  \
        if constexpr ((CY) == 0.0) {                      \
            ox = (cs_) * (R)(CX);                         \
            oy = (sn_) * (R)(CX);                         \
        } else if constexpr ((CX) == 0.0) {               \
            ox = -((sn_) * (R)(CY));                      \
            oy = (cs_) * (R)(CY);                         \
        } else {                                          \
            ox = (cs_) * (R)(CX) - (sn_) * (R)(CY);       \
            oy = (sn_) * (R)(CX) + (cs_) * (R)(CY);       \
        }                                                 \
    } while (0)

template <typename R, int NB>
struct PlanarKin {
    R sn[NB], cs[NB];        // sin / cos of absolute body angle
    R ax[NB], ay[NB];        // hinge anchor (body frame origin) relative to the root origin
    R lx[NB], ly[NB];        // world vector parent anchor -> own anchor  (R(phi_parent) * joint offset)
    R ex[NB], ey[NB];        // world vector own anchor -> own COM         (R(phi_i) * com offset)
    R px[NB], py[NB];        // COM relative to the root origin
    R om[NB];                // absolute angular velocity
    R vax[NB], vay[NB];      // anchor velocity (world)
    R vpx[NB], vpy[NB];      // COM velocity (world)
};

template <class Mdl>
struct PlanarTree {
    static constexpr int NB = Mdl::NB;
    static constexpr int NV = NB + 2;
    static constexpr bool HAS_GRAVITY = (Mdl::gx() != 0.0) || (Mdl::gy() != 0.0);

    // is body A an ancestor of (or equal to) body B?
    static constexpr bool is_ancestor(int a, int b) {
        while (b > a) b = Mdl::parent(b);
        return a == b;
    }
    static constexpr double total_mass() {
        double m = 0.0;
        for (int i = 0; i < NB; ++i) m += Mdl::mass(i);
        return m;
    }

    // exact sin / cos of every absolute body angle
    template <typename R>
    RL_HD static void angles(const R* q, R* sn, R* cs) {
        R phi[NB];
        phi[0] = q[2];
        static_for<1, NB>([&](auto I) {
            constexpr int i = decltype(I)::value, p = Mdl::parent(i);
            phi[i] = phi[p] + q[2 + i];
        });
        static_for<0, NB>([&](auto I) {
            constexpr int i = decltype(I)::value;
            rl_sincos(phi[i], sn[i], cs[i]);
        });
    }

    // positions / velocities of every body from (q, qd)
    template <typename R>
    RL_HD static void kinematics(const R* q, const R* qd, PlanarKin<R, NB>& k) {
        angles(q, k.sn, k.cs);
        kinematics_sc(qd, k);
    }

    // the same with k.sn / k.cs already filled in
    template <typename R>
    RL_HD static void kinematics_sc(const R* qd, PlanarKin<R, NB>& k) {
        k.om[0] = qd[2];
        static_for<1, NB>([&](auto I) {
            constexpr int i = decltype(I)::value, p = Mdl::parent(i);
            k.om[i] = k.om[p] + qd[2 + i];
        });
        k.ax[0] = (R)0; k.ay[0] = (R)0;
        k.vax[0] = qd[0]; k.vay[0] = qd[1];
        static_for<0, NB>([&](auto I) {
            constexpr int i = decltype(I)::value;
            if constexpr (i > 0) {
                constexpr int p = Mdl::parent(i);
                constexpr double JX = Mdl::jx(i), JY = Mdl::jy(i);
                R dx, dy;                                   // R(phi_p) * joint offset
                RL_ROTC(k.cs[p], k.sn[p], JX, JY, dx, dy);
                k.lx[i] = dx;
                k.ly[i] = dy;
                if constexpr (p == 0) {                     // the root anchor is the origin
                    k.ax[i] = dx;
                    k.ay[i] = dy;
                } else {
                    k.ax[i] = k.ax[p] + dx;
                    k.ay[i] = k.ay[p] + dy;
                }
                k.vax[i] = k.vax[p] - k.om[p] * dy;         // + Omega_p x d
                k.vay[i] = k.vay[p] + k.om[p] * dx;
            }
            constexpr double CX = Mdl::cx(i), CY = Mdl::cy(i);
            R ex, ey;                                       // R(phi_i) * com offset
            RL_ROTC(k.cs[i], k.sn[i], CX, CY, ex, ey);
            k.ex[i] = ex;
            k.ey[i] = ey;
            if constexpr (i == 0) {
                k.px[i] = ex;
                k.py[i] = ey;
            } else {
                k.px[i] = k.ax[i] + ex;
                k.py[i] = k.ay[i] + ey;
            }
            k.vpx[i] = k.vax[i] - k.om[i] * ey;
            k.vpy[i] = k.vay[i] + k.om[i] * ex;
        });
    }

    // Solve the symmetric positive definite system S x = b (S given by its lower triangle,
    // destroyed).  N == 3 (a 3-link chain): closed-form adjugate solve -- ONE division on the
    // critical path instead of three sequential pivots; otherwise LDL^T.
    template <typename R, int N>
    RL_HD static void solve_spd(R (&S)[N][N], const R* b, R* x) {
        if constexpr (N == 3) {
            const R a = S[0][0], bb = S[1][0], c = S[2][0], d = S[1][1], e = S[2][1], f = S[2][2];
            const R A = d * f - e * e, B = c * e - bb * f, C = bb * e - c * d;
            const R D = a * f - c * c, E = bb * c - a * e, F = a * d - bb * bb;
            const R det = a * A + (bb * B + c * C);
            const R inv = rl_recip_normal(det);
            x[0] = (A * b[0] + (B * b[1] + C * b[2])) * inv;
            x[1] = (B * b[0] + (D * b[1] + E * b[2])) * inv;
            x[2] = (C * b[0] + (E * b[1] + F * b[2])) * inv;
        } else {
            R Dg[N], Di[N];   // D and 1/D of S = L D L^T
            static_for<0, N>([&](auto Cc) {
                constexpr int c = decltype(Cc)::value;
                R d = S[c][c];
                static_for<0, c>([&](auto T) {
                    constexpr int t = decltype(T)::value;
                    d = d - S[c][t] * S[c][t] * Dg[t];
                });
                Dg[c] = d;
                const R inv = rl_recip_normal(d);
                Di[c] = inv;
                static_for<c + 1, N>([&](auto Rr) {
                    constexpr int r = decltype(Rr)::value;
                    R v = S[r][c];
                    static_for<0, c>([&](auto T) {
                        constexpr int t = decltype(T)::value;
                        v = v - S[r][t] * S[c][t] * Dg[t];
                    });
                    S[r][c] = v * inv;
                });
            });
            static_for<0, N>([&](auto Rr) {
                constexpr int r = decltype(Rr)::value;
                R v = b[r];
                static_for<0, r>([&](auto T) {
                    constexpr int t = decltype(T)::value;
                    v = v - S[r][t] * x[t];
                });
                x[r] = v;
            });
            static_for<0, N>([&](auto Rr) {
                constexpr int r = decltype(Rr)::value;
                x[r] = x[r] * Di[r];
            });
            static_for_down<N, 0>([&](auto Rr) {
                constexpr int r = decltype(Rr)::value;
                R v = x[r];
                static_for<r + 1, N>([&](auto T) {
                    constexpr int t = decltype(T)::value;
                    v = v - S[t][r] * x[t];
                });
                x[r] = v;
            });
        }
    }

    // ---- compile-time geometry of the tree in ABSOLUTE body angles --------------------------------
    // COM_i = r + sum_{k in path(i)} R(phi_k) d_ik with constant vectors d_ik: the joint offset of the
    // child of k on the way to i (k a strict ancestor of i), or the COM offset (k == i).
    struct V2 { double x, y; };
    static constexpr V2 dvec(int i, int k) {
        if (k == i) return V2{Mdl::cx(i), Mdl::cy(i)};
        if (!is_ancestor(k, i)) return V2{0.0, 0.0};
        int n = i;
        while (Mdl::parent(n) != k) n = Mdl::parent(n);
        return V2{Mdl::jx(n), Mdl::jy(n)};
    }
    // D_k = sum_i m_i d_ik: first moment of the subtree hanging on phi_k
    static constexpr V2 Dvec(int k) {
        V2 s{0.0, 0.0};
        for (int i = 0; i < NB; ++i) { const V2 d = dvec(i, k); s.x += Mdl::mass(i) * d.x; s.y += Mdl::mass(i) * d.y; }
        return s;
    }
    // M_{phi_k phi_l} = Ac_kl cos(phi_l - phi_k) - As_kl sin(phi_l - phi_k) + Kc_kl
    static constexpr double Ac(int k, int l) {
        double s = 0.0;
        for (int i = 0; i < NB; ++i) { const V2 a = dvec(i, k), b = dvec(i, l); s += Mdl::mass(i) * (a.x * b.x + a.y * b.y); }
        return s;
    }
    static constexpr double As(int k, int l) {
        double s = 0.0;
        for (int i = 0; i < NB; ++i) { const V2 a = dvec(i, k), b = dvec(i, l); s += Mdl::mass(i) * (a.x * b.y - a.y * b.x); }
        return s;
    }
    // configuration-independent part: body inertia on the diagonal; armature acts on the RELATIVE
    // joint rate (phi_j' - phi_p'), i.e. +arm_j on (j,j) and (p,p), -arm_j on (j,p)
    static constexpr double Kc(int k, int l) {
        double s = (k == l) ? Mdl::inertia(k) : 0.0;
        for (int j = 1; j < NB; ++j) {
            const int p = Mdl::parent(j);
            if (k == l && (k == j || k == p)) s += Mdl::armature(j);
            if ((k == j && l == p) || (k == p && l == j)) s -= Mdl::armature(j);
        }
        return s;
    }
    // Schur complement with the translation block (m_total I_2): S_kl = Sc_kl cos - Ss_kl sin + Kc_kl
    static constexpr double Sc(int k, int l) {
        const V2 a = Dvec(k), b = Dvec(l);
        return Ac(k, l) - (a.x * b.x + a.y * b.y) / total_mass();
    }
    static constexpr double Ss(int k, int l) {
        const V2 a = Dvec(k), b = Dvec(l);
        return As(k, l) - (a.x * b.y - a.y * b.x) / total_mass();
    }

    // qacc from (kinematics, hinge torques tau_j[NB] incl. actuation, per-body external force (fx, fy) at
    // the COM and torque tz, world frame).
    //
    // Formulated in absolute body angles phi_k: every configuration dependence of the joint-space
    // inertia, of its Schur complement with the translations and of the velocity-product terms is a
    // compile-time coefficient times cos / sin of an angle DIFFERENCE,
    //     S_kl   = Sc_kl cos(phi_l - phi_k) - Ss_kl sin(phi_l - phi_k) + Kc_kl
    //     bias_k = - sum_l w_l^2 [As_kl cos(phi_l - phi_k) + Ac_kl sin(phi_l - phi_k)]
    // so there is no composite-inertia pass, no matrix assembly and no B^T B product at run time
    // (for the collinear swimmer chain all As / Ss vanish at compile time as well).
    template <typename R>
    RL_HD static void forward_dynamics_abs(const PlanarKin<R, NB>& k, const R* tau_j, const R* fx,
                                           const R* fy, const R* tz, R* qacc) {
        constexpr double INV_M = 1.0 / total_mass();
        // --- subtree force sums (leaves -> root) and generalised forces on the absolute angles -------
        R Fsx[NB], Fsy[NB], Q[NB];
        static_for<0, NB>([&](auto I) {
            constexpr int i = decltype(I)::value;
            Fsx[i] = fx[i];
            Fsy[i] = fy[i];
            Q[i] = (k.ex[i] * fy[i] - k.ey[i] * fx[i]) + tz[i];       // own force at the own COM + pure torque
        });
        static_for_down<NB, 1>([&](auto I) {
            constexpr int i = decltype(I)::value, p = Mdl::parent(i);
            // everything beyond joint i translates with anchor i when phi_p alone varies
            Q[p] = Q[p] + (k.lx[i] * Fsy[i] - k.ly[i] * Fsx[i]);
            Fsx[p] = Fsx[p] + Fsx[i];
            Fsy[p] = Fsy[p] + Fsy[i];
        });
        static_for<1, NB>([&](auto I) {
            constexpr int i = decltype(I)::value, p = Mdl::parent(i);
            Q[i] = Q[i] + tau_j[i];                                    // hinge torque: +tau on the child,
            Q[p] = Q[p] - tau_j[i];                                    // -tau on the parent
        });
        // --- rotated first moments G_k = R(phi_k) D_k (gravity, translation coupling, centripetal) ------
        R Gx[NB], Gy[NB], w2[NB];
        static_for<0, NB>([&](auto I) {
            constexpr int i = decltype(I)::value;
            constexpr V2 D = Dvec(i);
            RL_ROTC(k.cs[i], k.sn[i], D.x, D.y, Gx[i], Gy[i]);
            w2[i] = k.om[i] * k.om[i];
            if constexpr (HAS_GRAVITY) Q[i] = Q[i] + (Gx[i] * (R)Mdl::gy() - Gy[i] * (R)Mdl::gx());
        });
        // translation rows: m r'' + sum_k G_k^perp phi_k'' - sum_l w_l^2 G_l = Q_r
        R qrx = Fsx[0], qry = Fsy[0];
        if constexpr (HAS_GRAVITY) {
            qrx = qrx + (R)(total_mass() * Mdl::gx());
            qry = qry + (R)(total_mass() * Mdl::gy());
        }
        static_for<0, NB>([&](auto I) {
            constexpr int i = decltype(I)::value;
            qrx = qrx + w2[i] * Gx[i];
            qry = qry + w2[i] * Gy[i];
        });
        const R grx = qrx * (R)INV_M, gry = qry * (R)INV_M;           // (Q_r - c_r) / m
        // --- pairwise angle differences: Schur matrix and velocity-product terms -----------------------
        R S[NB][NB], b[NB], th[NB];
        static_for<0, NB>([&](auto Kk) {
            constexpr int kk = decltype(Kk)::value;
            S[kk][kk] = (R)(Sc(kk, kk) + Kc(kk, kk));
            // rhs: Q_k - G_k^perp . (Q_r - c_r)/m     with G^perp = (-Gy, Gx)
            b[kk] = Q[kk] - (Gx[kk] * gry - Gy[kk] * grx);
        });
        static_for<0, NB>([&](auto Kk) {
            constexpr int kk = decltype(Kk)::value;
            static_for<kk + 1, NB>([&](auto Ll) {
                constexpr int l = decltype(Ll)::value;
                constexpr double AC = Ac(kk, l), AS = As(kk, l), SC = Sc(kk, l), SS = Ss(kk, l), KC = Kc(kk, l);
                constexpr bool NEED_C = (SC != 0.0) || (AS != 0.0), NEED_S = (SS != 0.0) || (AC != 0.0);
                R cd = (R)0, sd = (R)0;                                // cos / sin of (phi_l - phi_k)
                if constexpr (NEED_C) cd = k.cs[kk] * k.cs[l] + k.sn[kk] * k.sn[l];
                if constexpr (NEED_S) sd = k.sn[l] * k.cs[kk] - k.cs[l] * k.sn[kk];
                // S[l][kk] (lower triangle)
                if constexpr (SC != 0.0 && SS != 0.0) S[l][kk] = (R)SC * cd - (R)SS * sd + (R)KC;
                else if constexpr (SC != 0.0 && KC != 0.0) S[l][kk] = (R)SC * cd + (R)KC;
                else if constexpr (SC != 0.0) S[l][kk] = (R)SC * cd;
                else if constexpr (SS != 0.0) S[l][kk] = (R)KC - (R)SS * sd;
                else S[l][kk] = (R)KC;
                // t = As cos + Ac sin:  bias_k -= w_l^2 t,  bias_l += w_k^2 t   (bias moves to the rhs)
                if constexpr (AC != 0.0 || AS != 0.0) {
                    R t;
                    if constexpr (AC != 0.0 && AS != 0.0) t = (R)AS * cd + (R)AC * sd;
                    else if constexpr (AC != 0.0) t = (R)AC * sd;
                    else t = (R)AS * cd;
                    b[kk] = b[kk] + w2[l] * t;
                    b[l] = b[l] - w2[kk] * t;
                }
            });
        });
        solve_spd<R, NB>(S, b, th);
        // translations: r'' = (Q_r - c_r)/m - sum_k G_k^perp phi_k'' / m
        R sx = (R)0, sy = (R)0;
        static_for<0, NB>([&](auto Rr) {
            constexpr int r = decltype(Rr)::value;
            if constexpr (r == 0) { sx = -(Gy[0] * th[0]); sy = Gx[0] * th[0]; }
            else { sx = sx - Gy[r] * th[r]; sy = sy + Gx[r] * th[r]; }
        });
        qacc[0] = grx - sx * (R)INV_M;
        qacc[1] = gry - sy * (R)INV_M;
        // back to the joint coordinates: root angle, then hinge = child angle - parent angle
        qacc[2] = th[0];
        static_for<1, NB>([&](auto Rr) {
            constexpr int r = decltype(Rr)::value;
            qacc[2 + r] = th[r] - th[Mdl::parent(r)];
        });
    }

    // The pairwise absolute-angle form serves the small trees (Swimmer's scalar program, Hopper, the pendula); the
    // two-legged seven-body trees (HalfCheetah, Walker2D) have their own sub-step, dyn_two_legs.h.
    template <typename R>
    RL_HD static void forward_dynamics(const PlanarKin<R, NB>& k, const R* tau_j, const R* fx,
                                       const R* fy, const R* tz, R* qacc) {
        static_assert(NB <= 4, "trees beyond four bodies: dyn_two_legs.h");
        forward_dynamics_abs(k, tau_j, fx, fy, tz, qacc);
    }

    // passive joint torques: spring (ref 0), damper, soft range limits
    template <typename R>
    RL_HD static void joint_passive(const R* q, const R* qd, R* tau_j) {
        static_for<1, NB>([&](auto I) {
            constexpr int i = decltype(I)::value;
            const R x = q[2 + i], v = qd[2 + i];
            R t;
            if constexpr (Mdl::stiffness(i) != 0.0 && Mdl::damping(i) != 0.0)
                t = -(R)Mdl::stiffness(i) * x - (R)Mdl::damping(i) * v;
            else if constexpr (Mdl::stiffness(i) != 0.0)
                t = -(R)Mdl::stiffness(i) * x;
            else if constexpr (Mdl::damping(i) != 0.0)
                t = -(R)Mdl::damping(i) * v;
            else
                t = (R)0;
            if constexpr (Mdl::limited(i)) {
                // penalty beyond the range: -K * (x - clamp(x)) and, only while beyond, -B * v
                const R viol = x - rl_clamp(x, (R)Mdl::lo(i), (R)Mdl::hi(i));
                const R damp = (viol != (R)0) ? (R)Mdl::limit_b() * v : (R)0;
                t = t - (R)Mdl::limit_k() * viol - damp;
            }
            tau_j[i] = t;
        });
        tau_j[0] = (R)0;
    }

    // One mj_step-style substep: q, qd advanced by h with hinge actuation act_j[NB].
    // (sn, cs) carry sin / cos of the absolute body angles from sub-step to sub-step: instead of
    // three range reductions + two polynomials per body and sub-step (a quarter of the sub-step's
    // instructions, at the head of its dependency chain) they are rotated by the small angle
    // h * omega_i -- exactly the increment the integrator applies to the angle itself.  The caller
    // seeds them with angles() at the start of every env step, so the state between env steps is
    // (q, qd) alone and rounding drift is bounded by one env step's worth of sub-steps.
    template <typename R>
    RL_HD static void substep(R* q, R* qd, const R* act_j, R h, R* sn, R* cs) {
        PlanarKin<R, NB> k;
        RL_UNROLL
        for (int i = 0; i < NB; ++i) { k.sn[i] = sn[i]; k.cs[i] = cs[i]; }
        kinematics_sc(qd, k);
        R tau_j[NB], fx[NB], fy[NB], tz[NB];
        joint_passive(q, qd, tau_j);
        static_for<1, NB>([&](auto I) {
            constexpr int i = decltype(I)::value;
            tau_j[i] = tau_j[i] + act_j[i];
        });
        Mdl::template external<R>(q, k, fx, fy, tz);   // SETS fx, fy, tz of every body
        R qacc[NV];
        forward_dynamics(k, tau_j, fx, fy, tz, qacc);
        RL_UNROLL
        for (int r = 0; r < NV; ++r) {
            qd[r] = qd[r] + h * qacc[r];
            q[r] = q[r] + h * qd[r];
        }
        R om[NB];
        om[0] = qd[2];
        static_for<1, NB>([&](auto I) {
            constexpr int i = decltype(I)::value, p = Mdl::parent(i);
            om[i] = om[p] + qd[2 + i];
        });
        RL_UNROLL
        for (int i = 0; i < NB; ++i) rl_rotate_small(sn[i], cs[i], h * om[i]);
    }

    // Generalised force of the model's CONSTRAINT-like terms (external contact wrenches + joint-limit penalties) on the
    // tree's coordinates [P1, P2, root hinge, hinges 1..NB-1] at (q, qd): the analogue of MuJoCo's data.qfrc_constraint
    // that some envs observe (hopper_env.py:44).  Hinge i collects the moment about its anchor of every wrench in its
    // subtree; the translations collect the force sums.
    template <typename R>
    RL_HD static void constraint_forces(const R* q, const R* qd, R* qf) {
        PlanarKin<R, NB> k;
        kinematics(q, qd, k);
        R fx[NB], fy[NB], tz[NB];
        Mdl::template external<R>(q, k, fx, fy, tz);
        R sx = (R)0, sy = (R)0;
        RL_UNROLL
        for (int b = 0; b < NB; ++b) { sx = sx + fx[b]; sy = sy + fy[b]; }
        qf[0] = sx;
        qf[1] = sy;
        static_for<0, NB>([&](auto I) {
            constexpr int i = decltype(I)::value;
            R m = (R)0;
            static_for<i, NB>([&](auto B) {
                constexpr int b = decltype(B)::value;
                if constexpr (is_ancestor(i, b))
                    m = m + (tz[b] + ((k.px[b] - k.ax[i]) * fy[b] - (k.py[b] - k.ay[i]) * fx[b]));
            });
            if constexpr (i >= 1 && Mdl::limited(i)) {
                const R x = q[2 + i], v = qd[2 + i];
                const R viol = x - rl_clamp(x, (R)Mdl::lo(i), (R)Mdl::hi(i));
                const R damp = (viol != (R)0) ? (R)Mdl::limit_b() * v : (R)0;
                m = m - (R)Mdl::limit_k() * viol - damp;
            }
            qf[2 + i] = m;
        });
    }

    // subtree(root) centre of mass (world) and its velocity
    template <typename R>
    RL_HD static void com(const R* q, const R* qd, R& cx, R& cy, R& vx, R& vy) {
        PlanarKin<R, NB> k;
        angles(q, k.sn, k.cs);
        com_sc(q, qd, k, cx, cy, vx, vy);
    }
    // the same with k.sn / k.cs already filled in (the fused Swimmer rollout evaluates the sines once, lane-parallel).
    // The mass-weighted sums are scaled by the reciprocal of the (compile-time) total mass: one multiply each instead
    // of a correctly rounded division (12 instructions on gfx950); both builds do the same.
    template <typename R>
    RL_HD static void com_sc(const R* q, const R* qd, PlanarKin<R, NB>& k, R& cx, R& cy, R& vx, R& vy) {
        kinematics_sc(qd, k);
        R sx = (R)0, sy = (R)0, mvx = (R)0, mvy = (R)0;
        RL_UNROLL
        for (int i = 0; i < NB; ++i) {
            const R mi = (R)Mdl::mass(i);
            sx = sx + mi * k.px[i];
            sy = sy + mi * k.py[i];
            mvx = mvx + mi * k.vpx[i];
            mvy = mvy + mi * k.vpy[i];
        }
        const R im = (R)(1.0 / total_mass());
        cx = q[0] + sx * im;
        cy = q[1] + sy * im;
        vx = mvx * im;
        vy = mvy * im;
    }
};

}  // namespace rl
