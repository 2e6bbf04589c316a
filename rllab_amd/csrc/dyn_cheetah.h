// dyn_cheetah.h -- HalfCheetahEnv-style env: planar 7-body / 9-DoF articulated tree with
// joint springs, dampers, armature, geared motors, gravity and capsule-floor contacts; single
// source for the gfx950 kernels and the host oracle build.
//
// Replaces, for one env copy:
//   HalfCheetahEnv.step / get_current_obs   rllab/envs/mujoco/half_cheetah_env.py:22-46
//   MujocoEnv.reset_mujoco / forward_dynamics   rllab/envs/mujoco/mujoco_env.py:109-116,184-191
//   MjModel.step / forward / _compute_subtree   rllab/mujoco_py/mjcore.py:46-84
//   model constants                         vendor/mujoco_models/half_cheetah.xml:36-93
//                                           (through gen_cheetah_constants.py -> cheetah_constants.h)
//   NormalizedEnv.step                      rllab/envs/normalized_env.py:78-92
// "-style" (BASELINE.json): rigid-body tree, joint passive forces and actuation follow the
// MJCF; what MuJoCo 1.31 (proprietary, absent) solves with its soft-constraint solver --
// joint limits and the condim-3 capsule/plane contacts (friction 0.4, solref .02 1) -- is a
// spring-damper penalty here: normal force k*depth - b*v_n (>= 0) per capsule end sphere,
// viscous-regularised Coulomb friction |f_t| <= 0.4 f_n, limit torque k_l*viol + b_l*qd.
// Explicit penalties on ~1 kg feet are stiff, so one 0.01 s MuJoCo step is integrated as
// 4 semi-implicit Euler sub-steps of 0.0025 s (documented deviation).
//
// Plane coordinates (P1, P2) = (z, x) so that MuJoCo's +y hinge angle is CCW-positive.
// State (18 reals): q[9] = [z (absolute height of the torso frame), x, rooty, bthigh, bshin,
// bfoot, fthigh, fshin, ffoot], qd[9].  MuJoCo's qpos = [x, z - 0.7, rooty, ...].
#pragma once
#include "cheetah_constants.h"
#include "dyn_two_legs.h"
#include "dyn_mjc.h"

namespace rl {

struct CheetahModel {
    static constexpr int NB = cheetah::NB;
    RL_HD static constexpr int parent(int i) { return cheetah::PARENT[i]; }
    RL_HD static constexpr double jx(int i) { return cheetah::JX[i]; }
    RL_HD static constexpr double jy(int i) { return cheetah::JY[i]; }
    RL_HD static constexpr double cx(int i) { return cheetah::CX[i]; }
    RL_HD static constexpr double cy(int i) { return cheetah::CY[i]; }
    RL_HD static constexpr double mass(int i) { return cheetah::MASS[i]; }
    RL_HD static constexpr double inertia(int i) { return cheetah::INERTIA[i]; }
    RL_HD static constexpr double armature(int i) { return cheetah::ARMATURE[i]; }
    RL_HD static constexpr double damping(int i) { return cheetah::DAMPING[i]; }
    RL_HD static constexpr double stiffness(int i) { return cheetah::STIFFNESS[i]; }
    RL_HD static constexpr bool limited(int i) { return i >= 1; }
    RL_HD static constexpr double lo(int i) { return cheetah::LO[i]; }
    RL_HD static constexpr double hi(int i) { return cheetah::HI[i]; }
    RL_HD static constexpr double limit_k() { return 2.0e3; }
    RL_HD static constexpr double limit_b() { return 15.0; }
    RL_HD static constexpr double gx() { return -9.81; }  // gravity along -z = -P1
    RL_HD static constexpr double gy() { return 0.0; }

    // contact table (dyn_two_legs.h): torso spheres first, then two per leg body
    static constexpr int NC = cheetah::NC;
    RL_HD static constexpr int cbody(int c) { return cheetah::CBODY[c]; }
    RL_HD static constexpr double cpx(int c) { return cheetah::CPX[c]; }
    RL_HD static constexpr double cpy(int c) { return cheetah::CPY[c]; }
    RL_HD static constexpr double crad(int) { return cheetah::CRAD; }
    RL_HD static constexpr double cmu(int) { return MU; }

    static constexpr double CONTACT_K = 2.0e4;   // N/m per end sphere
    static constexpr double CONTACT_B = 3.0e2;   // N s/m while penetrating
    static constexpr double FRICTION_C = 3.0e2;  // N s/m tangential, clamped to mu * f_n
    static constexpr double MU = 0.4;
};

// half_cheetah.xml:38-39: joints solreflimit = ".02 1", solimplimit = "0 .8 .03"; geoms solref = "0.02 1", solimp = "0.0 0.8 0.01"
struct CheetahMjcPar {
    RL_HD static constexpr MjcSol limit() { return MjcSol{0.02, 1.0, 0.0, 0.8, 0.03, 0.0}; }
    RL_HD static constexpr MjcSol contact() { return MjcSol{0.02, 1.0, 0.0, 0.8, 0.01, 0.0}; }
};

struct HalfCheetah {
    static constexpr int OBS = 20;
    static constexpr int ACT = 6;
    static constexpr int STATE = 18;
    static constexpr int ACT_BUF = ACT;   // step_begin's hand-over to step_end
    static constexpr int RESET_DRAWS = 18;  // N(0,1): 9 for qpos, 9 for qvel (MuJoCo order)
    static constexpr bool RESET_NORMAL = true;
    static constexpr int KIND = 3;
    static constexpr bool TERMINATES = false;   // done is always False (half_cheetah_env.py:45)
    static constexpr int SUBSTEPS = 4;      // 4 x 0.0025 s = one 0.01 s MuJoCo step, frame_skip 1
    using Tree = PlanarTree<CheetahModel>;
    using Legs = TwoLegs<CheetahModel>;
    using Mjc = MjcTree<CheetahModel, CheetahMjcPar>;     // limit_model / contact_model = "mujoco" (dyn_mjc.h)

    template <typename R> RL_HD static void action_bounds(R* lb, R* ub) {
        RL_UNROLL
        for (int k = 0; k < ACT; ++k) { lb[k] = (R)-1; ub[k] = (R)1; }
    }

    // qpos = init + 0.01 N(0,1), qvel = 0.1 N(0,1) in MuJoCo order [x, z, rooty, joints]
    template <typename R> RL_HD static StepOpts<R> default_opts() { return make_opts<R>(0.0, 0.0, 1); }

    template <typename R> RL_HD static void reset(R* s, const R* z, int /*flags*/ = 0, R /*link_len*/ = (R)1) {
        s[0] = (R)0.7 + z[1] * (R)0.01;   // absolute torso height
        s[1] = z[0] * (R)0.01;            // x
        s[9] = z[10] * (R)0.1;            // zdot
        s[10] = z[9] * (R)0.1;            // xdot
        RL_UNROLL
        for (int i = 2; i < 9; ++i) {
            s[i] = z[i] * (R)0.01;
            s[9 + i] = z[9 + i] * (R)0.1;
        }
    }

    // obs = [qpos[1:], qvel, com_subtree(torso)] (half_cheetah_env.py:22-27)
    template <typename R> RL_HD static void observe(const R* s, R* o) {
        R cz, cx, vz, vx;
        Legs::template com_of<R>(s, s + 9, cz, cx, vz, vx);
        write_obs(s, cx, cz, o);
    }

    template <typename R> RL_HD static void write_obs(const R* s, R cx, R cz, R* o) {
        o[0] = s[0] - (R)0.7;             // rootz (slide displacement)
        RL_UNROLL
        for (int i = 2; i < 9; ++i) o[i - 1] = s[i];
        o[8] = s[10];                     // xdot
        o[9] = s[9];                      // zdot
        RL_UNROLL
        for (int i = 2; i < 9; ++i) o[8 + i] = s[9 + i];
        o[17] = cx; o[18] = (R)0; o[19] = cz;
    }

    // Env.step in three parts (the lane-group rollouts run the sub-steps one body per lane):
    //   step_begin : NormalizedEnv action map, ctrl clamp, geared motor torques        (normalized_env.py:78-92)
    //   sub-steps  : SUBSTEPS x TwoLegs::substep
    //   step_end   : observation, reward, done                                         (half_cheetah_env.py:22-46)
    template <typename R>
    RL_HD static void step_begin(const R* a, int normalize, const StepOpts<R>& o, R* act, R* tau) {
        tau[0] = (R)0;
        RL_UNROLL
        for (int k = 0; k < ACT; ++k) {
            R v = a[k];
            if (normalize) v = rl_clamp((R)-1 + (v + (R)1) * (R)0.5 * (R)2, (R)-1, (R)1);
            act[k] = v;
            R applied = v;
            if (o.dact) applied = v + o.dact[k];       // ctrl = inject_action_noise(action) (mujoco_env.py:175-187)
            tau[1 + k] = (R)cheetah::GEAR[1 + k] * rl_clamp(applied, (R)-1, (R)1);  // ctrllimited motor
        }
    }
    // (cz, cx, vz, vx): centre of mass of the state s and its velocity (TwoLegs::com)
    template <typename R>
    RL_HD static void step_end_com(const R* s, const R* act, R cz, R cx, R /*vz*/, R vx, R* obs, R& reward, bool& done,
                                   const StepOpts<R>& /*o*/) {
        write_obs(s, cx, cz, obs);
        // reward = comvel_x - 0.1 * 0.5 * sum(clip(action)^2)   (half_cheetah_env.py:37-46)
        R ctrl = (R)0;
        RL_UNROLL
        for (int kk = 0; kk < ACT; ++kk) {
            const R c = rl_clamp(act[kk], (R)-1, (R)1);
            ctrl = ctrl + c * c;
        }
        reward = vx - (R)0.1 * (R)0.5 * ctrl;
        done = false;
    }
    template <typename R>
    RL_HD static void step(R* s, const R* a, int normalize, R* obs, R& reward, bool& done,
                           const StepOpts<R>& o = default_opts<R>()) {
#if !defined(__HIP_DEVICE_COMPILE__)
        // host build: the constraint model is a run-time option of the one step (device: MjcEnv<> instantiations)
        if (o.flags & (CFG_LIMIT_MUJOCO | CFG_CONTACT_MUJOCO)) {
            step_model<R, true>(s, a, normalize, obs, reward, done, o);
            return;
        }
#endif
        step_model<R, false>(s, a, normalize, obs, reward, done, o);
    }
    static constexpr bool HAS_MJC = true;
    template <typename R, bool MJC>
    RL_HD static void step_model(R* s, const R* a, int normalize, R* obs, R& reward, bool& done, const StepOpts<R>& o) {
        R act[ACT], tau[CheetahModel::NB];
        step_begin(a, normalize, o, act, tau);
        if constexpr (MJC)
            Mjc::template advance<R>(s, s + 9, tau, (R)0.0025, SUBSTEPS, o.flags);       // MuJoCo's soft constraints
        else   // all eight body lanes in one value (dyn_two_legs.h): exact sines at the start, SUBSTEPS sub-steps
            Legs::template advance<R>(s, s + 9, tau, (R)0.0025, SUBSTEPS);
        R cz, cx, vz, vx;
        Legs::template com_of<R>(s, s + 9, cz, cx, vz, vx);
        step_end_com(s, act, cz, cx, vz, vx, obs, reward, done, o);
    }

    // (x, z) of the torso subtree COM and its velocity, in the order get_body_com / get_body_comvel report them
    template <typename R> RL_HD static void com(const R* s, R* c4) {
        R cz, cx, vz, vx;
        Legs::template com_of<R>(s, s + 9, cz, cx, vz, vx);
        c4[0] = cx; c4[1] = cz; c4[2] = vx; c4[3] = vz;
    }
    static constexpr bool HAS_COM = true;
};

}  // namespace rl
