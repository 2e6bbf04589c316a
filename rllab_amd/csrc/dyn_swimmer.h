// dyn_swimmer.h -- SwimmerEnv-style env: planar 3-link chain in a viscous / inertial
// fluid, single source for the gfx950 kernels and the host oracle build.
//
// Replaces, for one env copy:
//   SwimmerEnv.step / get_current_obs      rllab/envs/mujoco/swimmer_env.py:25-45
//   MujocoEnv.reset_mujoco / forward_dynamics  rllab/envs/mujoco/mujoco_env.py:109-116,184-191
//   MjModel.step x frame_skip, forward, _compute_subtree (comvel)
//                                           rllab/mujoco_py/mjcore.py:46-84
//   model constants                         vendor/mujoco_models/swimmer.xml:3-42
//   NormalizedEnv.step                      rllab/envs/normalized_env.py:78-92
// Constants from the MJCF: three capsules (radius 0.1, cylinder length 1,
// density 1000), torso = 2 slides + hinge at the world origin with its capsule
// spanning local x in [0.5, 1.5]; mid hinged at torso-local (0.5, 0), back hinged at
// mid-local (-1, 0), both capsules spanning local x in [-1, 0]; hinges limited to
// +-100 deg; two motors, gear 1, ctrlrange +-50; timestep 0.001, frame_skip 50,
// Euler; fluid density 4000, viscosity 0.1; no contacts (collision="predefined").
// Fluid forces follow MuJoCo's documented inertia-box model (per body, in the body
// frame at the COM):  viscous  f = -3*pi*d*mu*v, t = -pi*d^3*mu*w  and inertial
// drag  f_i = -0.5*rho*b_j*b_k*|v_i|*v_i, t_z = -rho*b_z*(b_x^4+b_y^4)*|w|*w/64,
// with b the equivalent inertia box of the capsule and d its mean edge.
// Joint limits: spring-damper penalty by default; SwimmerEnv(limit_model="mujoco") = rl_env_cfg flag
// RL_CFG_LIMIT_MUJOCO: MuJoCo's documented soft-constraint model from the MJCF's own solreflimit / solimplimit
// (dyn_swimmer_chain.h), on the scalar program (VecEnv kernels, generic rollout shapes, host build).
//
// State (10 reals per env): qpos[5] = x, y, torso angle, rot2, rot3; qvel[5].
#pragma once
#include "dyn_planar.h"
#include "dyn_swimmer_chain.h"

namespace rl {

struct SwimmerModel {
    static constexpr int NB = 3;
    RL_HD static constexpr int parent(int i) { return i - 1; }
    // hinge anchor in the parent frame
    RL_HD static constexpr double jx(int i) { return i == 1 ? 0.5 : (i == 2 ? -1.0 : 0.0); }
    RL_HD static constexpr double jy(int) { return 0.0; }
    // capsule centre in the own frame
    RL_HD static constexpr double cx(int i) { return i == 0 ? 1.0 : -0.5; }
    RL_HD static constexpr double cy(int) { return 0.0; }
    // capsule r = 0.1, L = 1, rho = 1000: m = rho*(pi r^2 L + 4/3 pi r^3)
    RL_HD static constexpr double mass(int) { return 35.604716740684324; }
    // about the COM, axis normal to the plane: cylinder m_c (L^2/12 + r^2/4) + two
    // hemispheres 2*m_h*(83/320 r^2 + (L/2 + 3r/8)^2)
    RL_HD static constexpr double inertia(int) { return 3.917566039026472; }
    RL_HD static constexpr double armature(int) { return 0.0; }
    RL_HD static constexpr double damping(int) { return 0.0; }
    RL_HD static constexpr double stiffness(int) { return 0.0; }
    RL_HD static constexpr bool limited(int i) { return i >= 1; }
    RL_HD static constexpr double lo(int) { return -1.7453292519943295; }  // -100 deg
    RL_HD static constexpr double hi(int) { return 1.7453292519943295; }
    RL_HD static constexpr double limit_k() { return 1.0e4; }  // N m / rad beyond the range
    RL_HD static constexpr double limit_b() { return 5.0e2; }  // N m s / rad while beyond
    RL_HD static constexpr double gx() { return 0.0; }
    RL_HD static constexpr double gy() { return 0.0; }  // gravity is normal to the plane

    // equivalent inertia box of the capsule: bx = sqrt(6 (Iyy + Izz - Ixx) / m) along the
    // axis, by = bz = sqrt(6 Ixx / m) across, with Ixx = m_c r^2/2 + 2/5 m_s r^2
    static constexpr double BX = 1.1362476946200648, BY = 0.17115524428733953;
    static constexpr double RHO = 4000.0, MU = 0.1;
    static constexpr double DIAM = (BX + 2.0 * BY) / 3.0;
    static constexpr double PI = 3.14159265358979323846;
    static constexpr double VISC_LIN = 3.0 * PI * DIAM * MU;
    static constexpr double VISC_ANG = PI * DIAM * DIAM * DIAM * MU;
    static constexpr double DRAG_AX = 0.5 * RHO * BY * BY;        // along the capsule axis
    static constexpr double DRAG_PERP = 0.5 * RHO * BX * BY;      // across it
    static constexpr double DRAG_ANG = RHO * BY * (BX * BX * BX * BX + BY * BY * BY * BY) / 64.0;

    template <typename R>
    RL_HD static void external(const R* /*q*/, const PlanarKin<R, NB>& k, R* fx, R* fy, R* tz) {
        RL_UNROLL
        for (int i = 0; i < NB; ++i) {
            // COM velocity in the body frame
            const R vl = k.cs[i] * k.vpx[i] + k.sn[i] * k.vpy[i];
            const R vt = -k.sn[i] * k.vpx[i] + k.cs[i] * k.vpy[i];
            const R w = k.om[i];
            const R fl = -(vl * ((R)VISC_LIN + (R)DRAG_AX * rl_abs(vl)));
            const R ft = -(vt * ((R)VISC_LIN + (R)DRAG_PERP * rl_abs(vt)));
            fx[i] = k.cs[i] * fl - k.sn[i] * ft;
            fy[i] = k.sn[i] * fl + k.cs[i] * ft;
            tz[i] = -(w * ((R)VISC_ANG + (R)DRAG_ANG * rl_abs(w)));
        }
    }
};

struct Swimmer {
    static constexpr int OBS = 13;
    static constexpr int ACT = 2;
    static constexpr int STATE = 10;
    static constexpr int RESET_DRAWS = 10;  // N(0,1): 5 for qpos, 5 for qvel
    static constexpr bool RESET_NORMAL = true;
    static constexpr int KIND = 2;
    static constexpr bool TERMINATES = false;   // done is always False (swimmer_env.py:44)
    static constexpr int FRAME_SKIP = 50;
    using Tree = PlanarTree<SwimmerModel>;
    using Chain = SwimChain<SwimmerModel>;

    // (qpos, qvel) -> chain variables: absolute rates by prefix sums along the chain, exact sin / cos
    template <typename R>
    RL_HD static void to_chain(const R* q, const R* qd, R* r4, R* th, R* om, R* sn, R* cs) {
        r4[0] = q[0]; r4[1] = q[1]; r4[2] = qd[0]; r4[3] = qd[1];
        th[0] = q[2]; th[1] = q[3]; th[2] = q[4];
        om[0] = qd[2];
        om[1] = om[0] + qd[3];
        om[2] = om[1] + qd[4];
        Tree::template angles<R>(q, sn, cs);
    }
    template <typename R>
    RL_HD static void from_chain(const R* r4, const R* th, const R* om, R* q, R* qd) {
        q[0] = r4[0]; q[1] = r4[1]; qd[0] = r4[2]; qd[1] = r4[3];
        q[2] = th[0]; q[3] = th[1]; q[4] = th[2];
        qd[2] = om[0];
        qd[3] = om[1] - om[0];
        qd[4] = om[2] - om[1];
    }

    template <typename R> RL_HD static void action_bounds(R* lb, R* ub) {
        lb[0] = (R)-50; lb[1] = (R)-50;
        ub[0] = (R)50; ub[1] = (R)50;
    }

    template <typename R> RL_HD static StepOpts<R> default_opts() { return make_opts<R>(1e-2, 0.0, 1); }

    // MujocoEnv.reset_mujoco: qpos = init_qpos + 0.01*N(0,1), qvel = init_qvel + 0.1*N(0,1),
    // init_qpos = init_qvel = 0 (mujoco_env.py:109-116)
    template <typename R> RL_HD static void reset(R* s, const R* z, int /*flags*/ = 0, R /*link_len*/ = (R)1) {
        RL_UNROLL
        for (int i = 0; i < 5; ++i) {
            s[i] = z[i] * (R)0.01;
            s[5 + i] = z[5 + i] * (R)0.1;
        }
    }

    // obs = [qpos, qvel, com_subtree(torso)] (swimmer_env.py:25-30); the plane is z = 0
    template <typename R> RL_HD static void observe(const R* s, R* o) {
        R cx, cy, vx, vy;
        Tree::template com<R>(s, s + 5, cx, cy, vx, vy);
        RL_UNROLL
        for (int i = 0; i < 10; ++i) o[i] = s[i];
        o[10] = cx; o[11] = cy; o[12] = (R)0;
    }

    // Env.step in three parts, so that the fused rollout can run the sub-step loop on a lane group per env:
    //   step_begin : NormalizedEnv action map + ctrl clamp           (normalized_env.py:78-92, ctrllimited motors)
    //   sub-steps  : FRAME_SKIP x Chain::substep_*                   (mjcore.py:46-49 x frame_skip)
    //   step_end   : observation, reward, done                       (swimmer_env.py:25-45)
    template <typename R>
    RL_HD static void step_begin(const R* a, int normalize, R* act, R* ctrl, const R* dact = nullptr) {
        const R lb = (R)-50, ub = (R)50;
        ctrl[0] = (R)0;
        RL_UNROLL
        for (int k = 0; k < 2; ++k) {
            R v = a[k];
            if (normalize) {
                v = lb + (v + (R)1) * (R)0.5 * (ub - lb);
                v = rl_clamp(v, lb, ub);
            }
            act[k] = v;
            R applied = v;
            if (dact) applied = v + dact[k];           // ctrl = inject_action_noise(action) (mujoco_env.py:175-187)
            ctrl[1 + k] = rl_clamp(applied, lb, ub);   // ctrllimited: MuJoCo clamps ctrl to ctrlrange
        }
    }

    template <typename R>
    RL_HD static void step_end(const R* s, const R* act, R* obs, R& reward, bool& done,
                               R ctrl_cost_coeff = (R)1e-2) {
        PlanarKin<R, Tree::NB> k;
        Tree::template angles<R>(s, k.sn, k.cs);
        step_end_sc(s, act, k, obs, reward, done, ctrl_cost_coeff);
    }
    // the same with the sines / cosines of the absolute body angles already in k.sn / k.cs (rl_sincos of the same
    // angles, evaluated one body per lane by the fused rollout)
    template <typename R>
    RL_HD static void step_end_sc(const R* s, const R* act, PlanarKin<R, Tree::NB>& k, R* obs, R& reward, bool& done,
                                  R ctrl_cost_coeff = (R)1e-2) {
        R cx, cy, vx, vy;
        Tree::template com_sc<R>(s, s + 5, k, cx, cy, vx, vy);
        RL_UNROLL
        for (int i = 0; i < 10; ++i) obs[i] = s[i];
        obs[10] = cx; obs[11] = cy; obs[12] = (R)0;
        // reward = comvel_x - 0.5 * ctrl_cost_coeff * sum((action / scaling)^2), scaling = (ub - lb)/2 = 50 (as a
        // multiplication by 1/50); coefficient 1e-2 unless SwimmerEnv(ctrl_cost_coeff=..) says otherwise (0.5 * c is
        // exact, so the default keeps its bits)
        const R inv_scaling = (R)(1.0 / 50.0);
        const R a0 = act[0] * inv_scaling, a1 = act[1] * inv_scaling;
        const R ctrl_cost = (R)0.5 * ctrl_cost_coeff * (a0 * a0 + a1 * a1);
        reward = vx - ctrl_cost;
        done = false;
    }

    template <typename R>
    RL_HD static void step(R* s, const R* a, int normalize, R* obs, R& reward, bool& done,
                           const StepOpts<R>& o = default_opts<R>()) {
        R act[2], ctrl[3];
        step_begin(a, normalize, act, ctrl, o.dact);
        // 50 sub-steps in the chain program's variables (dyn_swimmer_chain.h): root translation, absolute body
        // rates, joint angles, carried sin / cos of the absolute body angles
        R r4[4], th[3], om[3], sn[3], cs[3];
        to_chain(s, s + 5, r4, th, om, sn, cs);
        if (o.flags & CFG_LIMIT_MUJOCO) {
            for (int it = 0; it < FRAME_SKIP; ++it) Chain::template substep_scalar<R, true>(r4, cs, sn, om, th, ctrl, (R)0.001);
        } else {
            for (int it = 0; it < FRAME_SKIP; ++it) Chain::template substep_scalar<R>(r4, cs, sn, om, th, ctrl, (R)0.001);
        }
        from_chain(r4, th, om, s, s + 5);
        step_end(s, act, obs, reward, done, o.ctrl_cost_coeff);
    }

    // subtree COM of the torso = of the whole chain: position and velocity (get_body_com / get_body_comvel)
    template <typename R> RL_HD static void com(const R* s, R* c4) {
        Tree::template com<R>(s, s + 5, c4[0], c4[1], c4[2], c4[3]);
    }
    static constexpr bool HAS_COM = true;
};

}  // namespace rl
