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
#include "rl_math.h"

#if defined(__HIPCC__)
#define RL_UNROLL _Pragma("unroll")
#else
#define RL_UNROLL
#endif

namespace rl {

template <typename R, int NB>
struct PlanarKin {
    R sn[NB], cs[NB];        // sin / cos of absolute body angle
    R ax[NB], ay[NB];        // hinge anchor (body frame origin) relative to the root origin
    R px[NB], py[NB];        // COM relative to the root origin
    R om[NB];                // absolute angular velocity
    R vax[NB], vay[NB];      // anchor velocity (world)
    R vpx[NB], vpy[NB];      // COM velocity (world)
};

template <class Mdl>
struct PlanarTree {
    static constexpr int NB = Mdl::NB;
    static constexpr int NV = NB + 2;

    // positions / velocities of every body from (q, qd)
    template <typename R>
    RL_HD static void kinematics(const R* q, const R* qd, PlanarKin<R, NB>& k) {
        R phi[NB];
        phi[0] = q[2];
        k.om[0] = qd[2];
        RL_UNROLL
        for (int i = 1; i < NB; ++i) {
            phi[i] = phi[Mdl::parent(i)] + q[2 + i];
            k.om[i] = k.om[Mdl::parent(i)] + qd[2 + i];
        }
        RL_UNROLL
        for (int i = 0; i < NB; ++i) rl_sincos(phi[i], k.sn[i], k.cs[i]);
        k.ax[0] = (R)0; k.ay[0] = (R)0;
        k.vax[0] = qd[0]; k.vay[0] = qd[1];
        RL_UNROLL
        for (int i = 0; i < NB; ++i) {
            if (i > 0) {
                const int p = Mdl::parent(i);
                const R jx = (R)Mdl::jx(i), jy = (R)Mdl::jy(i);
                const R dx = k.cs[p] * jx - k.sn[p] * jy;   // R(phi_p) * joint offset
                const R dy = k.sn[p] * jx + k.cs[p] * jy;
                k.ax[i] = k.ax[p] + dx;
                k.ay[i] = k.ay[p] + dy;
                k.vax[i] = k.vax[p] - k.om[p] * dy;         // + Omega_p x d
                k.vay[i] = k.vay[p] + k.om[p] * dx;
            }
            const R cx = (R)Mdl::cx(i), cy = (R)Mdl::cy(i);
            const R ex = k.cs[i] * cx - k.sn[i] * cy;       // R(phi_i) * com offset
            const R ey = k.sn[i] * cx + k.cs[i] * cy;
            k.px[i] = k.ax[i] + ex;
            k.py[i] = k.ay[i] + ey;
            k.vpx[i] = k.vax[i] - k.om[i] * ey;
            k.vpy[i] = k.vay[i] + k.om[i] * ex;
        }
    }

    // qacc from (q, qd, hinge torques tau_j[NB] incl. actuation, per-body external
    // force (fx, fy) at the COM and torque tz, world frame)
    template <typename R>
    RL_HD static void forward_dynamics(const PlanarKin<R, NB>& k, const R* tau_j, const R* fx,
                                       const R* fy, const R* tz, R* qacc) {
        // --- velocity-product (bias) accelerations with qacc = 0 ---------------------
        R aax[NB], aay[NB];  // anchor acceleration
        R Fx[NB], Fy[NB], Nz[NB];
        aax[0] = (R)0; aay[0] = (R)0;
        RL_UNROLL
        for (int i = 0; i < NB; ++i) {
            if (i > 0) {
                const int p = Mdl::parent(i);
                const R w2 = k.om[p] * k.om[p];
                aax[i] = aax[p] - w2 * (k.ax[i] - k.ax[p]);
                aay[i] = aay[p] - w2 * (k.ay[i] - k.ay[p]);
            }
            const R w2 = k.om[i] * k.om[i];
            const R ex = k.px[i] - k.ax[i], ey = k.py[i] - k.ay[i];
            const R acx = aax[i] - w2 * ex, acy = aay[i] - w2 * ey;
            const R m = (R)Mdl::mass(i);
            // net force on body i after moving m*a_bias to the right-hand side
            Fx[i] = fx[i] + m * ((R)Mdl::gx() - acx);
            Fy[i] = fy[i] + m * ((R)Mdl::gy() - acy);
            // moment of that force about the body's own anchor, plus pure torque
            Nz[i] = (ex * Fy[i] - ey * Fx[i]) + tz[i];
        }
        // --- accumulate subtree wrenches (leaves -> root) ----------------------------
        RL_UNROLL
        for (int i = NB - 1; i > 0; --i) {
            const int p = Mdl::parent(i);
            const R dx = k.ax[i] - k.ax[p], dy = k.ay[i] - k.ay[p];
            Nz[p] = Nz[p] + Nz[i] + (dx * Fy[i] - dy * Fx[i]);
            Fx[p] = Fx[p] + Fx[i];
            Fy[p] = Fy[p] + Fy[i];
        }
        R rhs[NV];
        rhs[0] = Fx[0];
        rhs[1] = Fy[0];
        rhs[2] = Nz[0];
        RL_UNROLL
        for (int i = 1; i < NB; ++i) rhs[2 + i] = Nz[i] + tau_j[i];

        // --- composite bodies: mass, first moment, inertia about the ROOT origin ------
        R mc[NB], hx[NB], hy[NB], J[NB];
        RL_UNROLL
        for (int i = 0; i < NB; ++i) {
            const R m = (R)Mdl::mass(i);
            mc[i] = m;
            hx[i] = m * k.px[i];
            hy[i] = m * k.py[i];
            J[i] = (R)Mdl::inertia(i) + m * (k.px[i] * k.px[i] + k.py[i] * k.py[i]);
        }
        RL_UNROLL
        for (int i = NB - 1; i > 0; --i) {
            const int p = Mdl::parent(i);
            mc[p] = mc[p] + mc[i];
            hx[p] = hx[p] + hx[i];
            hy[p] = hy[p] + hy[i];
            J[p] = J[p] + J[i];
        }
        // --- joint-space inertia (symmetric, lower triangle used) --------------------
        R Mm[NV][NV];
        RL_UNROLL
        for (int r = 0; r < NV; ++r)
            RL_UNROLL
            for (int c = 0; c < NV; ++c) Mm[r][c] = (R)0;
        Mm[0][0] = mc[0];
        Mm[1][1] = mc[0];
        RL_UNROLL
        for (int kk = 0; kk < NB; ++kk) {          // hinge of body kk (kk = 0: root rotation)
            const int col = 2 + kk;
            // translation rows: e_x . perp(h_k - mc_k a_k), e_y . perp(...)
            Mm[col][0] = -(hy[kk] - mc[kk] * k.ay[kk]);
            Mm[col][1] = (hx[kk] - mc[kk] * k.ax[kk]);
            // ancestors-or-self j of kk
            int j = kk;
            RL_UNROLL
            for (int depth = 0; depth < NB; ++depth) {
                if (j < 0) break;
                const R v = J[kk] - ((k.ax[j] + k.ax[kk]) * hx[kk] + (k.ay[j] + k.ay[kk]) * hy[kk]) +
                            mc[kk] * (k.ax[j] * k.ax[kk] + k.ay[j] * k.ay[kk]);
                Mm[col][2 + j] = v;
                j = (j == 0) ? -1 : Mdl::parent(j);
            }
            if (kk > 0) Mm[col][col] = Mm[col][col] + (R)Mdl::armature(kk);
        }
        // --- solve M qacc = rhs by LDL^T on the lower triangle ------------------------
        // (entries Mm[r][c], r >= c; unrelated hinge pairs stay 0)
        R Dg[NV], Di[NV];   // D and 1/D of M = L D L^T
        RL_UNROLL
        for (int c = 0; c < NV; ++c) {
            R d = Mm[c][c];
            RL_UNROLL
            for (int t = 0; t < c; ++t) d = d - Mm[c][t] * Mm[c][t] * Dg[t];
            Dg[c] = d;
            const R inv = (R)1 / d;
            Di[c] = inv;
            RL_UNROLL
            for (int r = c + 1; r < NV; ++r) {
                R v = Mm[r][c];
                RL_UNROLL
                for (int t = 0; t < c; ++t) v = v - Mm[r][t] * Mm[c][t] * Dg[t];
                Mm[r][c] = v * inv;
            }
        }
        RL_UNROLL
        for (int r = 0; r < NV; ++r) {
            R v = rhs[r];
            RL_UNROLL
            for (int t = 0; t < r; ++t) v = v - Mm[r][t] * qacc[t];
            qacc[r] = v;
        }
        RL_UNROLL
        for (int r = 0; r < NV; ++r) qacc[r] = qacc[r] * Di[r];
        RL_UNROLL
        for (int r = NV - 1; r >= 0; --r) {
            R v = qacc[r];
            RL_UNROLL
            for (int t = r + 1; t < NV; ++t) v = v - Mm[t][r] * qacc[t];
            qacc[r] = v;
        }
    }

    // passive joint torques: spring (ref 0), damper, soft range limits
    template <typename R>
    RL_HD static void joint_passive(const R* q, const R* qd, R* tau_j) {
        RL_UNROLL
        for (int i = 1; i < NB; ++i) {
            const R x = q[2 + i], v = qd[2 + i];
            R t = -(R)Mdl::stiffness(i) * x - (R)Mdl::damping(i) * v;
            if (Mdl::limited(i)) {
                const R lo = (R)Mdl::lo(i), hi = (R)Mdl::hi(i);
                if (x < lo) t = t - (R)Mdl::limit_k() * (x - lo) - (R)Mdl::limit_b() * v;
                if (x > hi) t = t - (R)Mdl::limit_k() * (x - hi) - (R)Mdl::limit_b() * v;
            }
            tau_j[i] = t;
        }
        tau_j[0] = (R)0;
    }

    // one mj_step-style substep: q, qd advanced by h with hinge actuation act_j[NB]
    template <typename R>
    RL_HD static void substep(R* q, R* qd, const R* act_j, R h) {
        PlanarKin<R, NB> k;
        kinematics(q, qd, k);
        R tau_j[NB], fx[NB], fy[NB], tz[NB];
        joint_passive(q, qd, tau_j);
        RL_UNROLL
        for (int i = 0; i < NB; ++i) {
            tau_j[i] = tau_j[i] + act_j[i];
            fx[i] = (R)0; fy[i] = (R)0; tz[i] = (R)0;
        }
        Mdl::template external<R>(q, k, fx, fy, tz);
        R qacc[NV];
        forward_dynamics(k, tau_j, fx, fy, tz, qacc);
        RL_UNROLL
        for (int r = 0; r < NV; ++r) {
            qd[r] = qd[r] + h * qacc[r];
            q[r] = q[r] + h * qd[r];
        }
    }

    // subtree(root) centre of mass (world) and its velocity
    template <typename R>
    RL_HD static void com(const R* q, const R* qd, R& cx, R& cy, R& vx, R& vy) {
        PlanarKin<R, NB> k;
        kinematics(q, qd, k);
        R m = (R)0, sx = (R)0, sy = (R)0, mvx = (R)0, mvy = (R)0;
        RL_UNROLL
        for (int i = 0; i < NB; ++i) {
            const R mi = (R)Mdl::mass(i);
            m = m + mi;
            sx = sx + mi * k.px[i];
            sy = sy + mi * k.py[i];
            mvx = mvx + mi * k.vpx[i];
            mvy = mvy + mi * k.vpy[i];
        }
        cx = q[0] + sx / m;
        cy = q[1] + sy / m;
        vx = mvx / m;
        vy = mvy / m;
    }
};

}  // namespace rl
