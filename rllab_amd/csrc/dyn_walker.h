// dyn_walker.h -- Walker2DEnv-style env: planar 7-body / 9-DoF biped (torso + two 3-link legs) with joint
// dampers, armature, torque motors, gravity and capsule-floor contacts; single source for the gfx950
// kernels and the host oracle build.
//
// Replaces, for one env copy:
//   Walker2DEnv.step / get_current_obs      rllab/envs/mujoco/walker2d_env.py:28-49
//   MujocoEnv.reset_mujoco / forward_dynamics   rllab/envs/mujoco/mujoco_env.py:109-116,184-191
//   MjModel.step / forward / _compute_subtree   rllab/mujoco_py/mjcore.py:46-84
//   model constants                         vendor/mujoco_models/walker2d.xml:3-59
//                                           (through gen_planar_constants.py -> walker_constants.h)
//   NormalizedEnv.step                      rllab/envs/normalized_env.py:78-92
// "-style": rigid-body tree, joint passive forces and actuation follow the MJCF; joint limits and the
// condim-3 capsule/plane contacts are the same spring-damper penalty model as dyn_cheetah.h (per-geom
// radius and friction), and one 0.005 s MuJoCo step (frame_skip 1) is integrated as 2 semi-implicit
// Euler sub-steps of 0.0025 s.
//
// Plane coordinates (P1, P2) = (z, x).  The leg hinges are declared about -y in the MJCF: MuJoCo's joint
// coordinate, velocity and motor torque are the negatives of the tree's (walker::SIGN).
// State (18 reals, tree convention): q[9] = [z (absolute torso height = MuJoCo's rootz, ref 1.25), x,
// rooty, thigh, leg, foot, thigh_left, leg_left, foot_left], qd[9].
#pragma once
#include "dyn_legged.h"
#include "dyn_two_legs.h"
#include "dyn_mjc.h"
#include "walker_constants.h"

namespace rl {

RL_LEGGED_CONSTANTS(WalkerK, walker);
using WalkerModel = LeggedModel<WalkerK>;   // joint limits + capsule-floor contacts: dyn_legged.h

// walker2d.xml:3-7 sets no solver parameter: MuJoCo's defaults, solref = "0.02 1", solimp = "0.9 0.95 0.001", margin 0
struct WalkerMjcPar {
    RL_HD static constexpr MjcSol limit() { return MjcSol{0.02, 1.0, 0.9, 0.95, 0.001, 0.0}; }
    RL_HD static constexpr MjcSol contact() { return MjcSol{0.02, 1.0, 0.9, 0.95, 0.001, 0.0}; }
};

struct Walker2D {
    static constexpr int OBS = 21;
    static constexpr int ACT = 6;
    static constexpr int STATE = 18;
    static constexpr int ACT_BUF = ACT + 1;   // step_begin's hand-over to step_end
    static constexpr int RESET_DRAWS = 18;  // N(0,1): 9 for qpos, 9 for qvel (MuJoCo order)
    static constexpr bool RESET_NORMAL = true;
    static constexpr int KIND = 5;
    static constexpr bool TERMINATES = true;   // a path can end before max_path_length (walker2d_env.py:47-49)
    static constexpr int SUBSTEPS = 2;      // 2 x 0.0025 s = one 0.005 s MuJoCo step, frame_skip 1
    using Tree = PlanarTree<WalkerModel>;
    using Legs = TwoLegs<WalkerModel>;
    using Mjc = MjcTree<WalkerModel, WalkerMjcPar>;       // limit_model / contact_model = "mujoco" (dyn_mjc.h)

    template <typename R> RL_HD static void action_bounds(R* lb, R* ub) {
        RL_UNROLL
        for (int k = 0; k < ACT; ++k) { lb[k] = -(R)walker::GEAR[1 + k]; ub[k] = (R)walker::GEAR[1 + k]; }
    }

    // qpos = init + 0.01 N(0,1) with init_qpos = [1.25, 0, ...], qvel = 0.1 N(0,1), MuJoCo order
    // [rootz, rootx, rooty, joints] and MuJoCo sign convention for the joints
    template <typename R> RL_HD static StepOpts<R> default_opts() { return make_opts<R>(1e-2, 0.0, 1); }

    template <typename R> RL_HD static void reset(R* s, const R* z, int /*flags*/ = 0, R /*link_len*/ = (R)1) {
        s[0] = (R)1.25 + z[0] * (R)0.01;
        s[1] = z[1] * (R)0.01;
        s[2] = z[2] * (R)0.01;
        s[9] = z[9] * (R)0.1;
        s[10] = z[10] * (R)0.1;
        s[11] = z[11] * (R)0.1;
        RL_UNROLL
        for (int i = 3; i < 9; ++i) {
            s[i] = (R)walker::SIGN[i - 2] * (z[i] * (R)0.01);
            s[9 + i] = (R)walker::SIGN[i - 2] * (z[9 + i] * (R)0.1);
        }
    }

    // obs = [qpos, qvel, com_subtree(torso)] in MuJoCo's convention (walker2d_env.py:28-33)
    template <typename R> RL_HD static void observe(const R* s, R* o) {
        R cz, cx, vz, vx;
        Legs::template com_of<R>(s, s + 9, cz, cx, vz, vx);
        write_obs(s, cx, cz, o);
    }

    template <typename R> RL_HD static void write_obs(const R* s, R cx, R cz, R* o) {
        o[0] = s[0]; o[1] = s[1]; o[2] = s[2];
        o[9] = s[9]; o[10] = s[10]; o[11] = s[11];
        RL_UNROLL
        for (int i = 3; i < 9; ++i) {
            o[i] = (R)walker::SIGN[i - 2] * s[i];
            o[9 + i] = (R)walker::SIGN[i - 2] * s[9 + i];
        }
        o[18] = cx; o[19] = (R)0; o[20] = cz;
    }

    // Env.step in three parts (the lane-group rollouts run the sub-steps one body per lane); act[ACT] carries the
    // clipped action, act[ACT] the control cost accumulated in step_begin
    template <typename R>
    RL_HD static void step_begin(const R* a, int normalize, const StepOpts<R>& o, R* act, R* tau) {
        tau[0] = (R)0;
        R ctrl_cost = (R)0;
        RL_UNROLL
        for (int k = 0; k < ACT; ++k) {
            const R ub = (R)walker::GEAR[1 + k], lb = -ub;
            R v = a[k];
            if (normalize) v = rl_clamp(lb + (v + (R)1) * (R)0.5 * (ub - lb), lb, ub);
            act[k] = rl_clamp(v, lb, ub);                        // action = clip(action, *bounds); ctrllimited motor
            R applied = act[k];
            if (o.dact) applied = rl_clamp(act[k] + o.dact[k], lb, ub);   // ctrl = action + noise, ctrllimited
            tau[1 + k] = (R)walker::SIGN[1 + k] * applied;       // gear 1: torque = ctrl, about the MJCF axis
            const R sc = act[k] / ((ub - lb) * (R)0.5);
            ctrl_cost = ctrl_cost + sc * sc;
        }
        act[ACT] = ctrl_cost;
    }
    template <typename R>
    RL_HD static void step_end_com(const R* s, const R* act, R cz, R cx, R /*vz*/, R vx, R* obs, R& reward, bool& done,
                                   const StepOpts<R>& o) {
        write_obs(s, cx, cz, obs);
        // reward = comvel_x - 0.5 * 1e-2 * sum((action / scaling)^2)      (walker2d_env.py:35-44)
        reward = vx - (R)0.5 * o.ctrl_cost_coeff * act[ACT];
        // done = not (0.8 < qpos[0] < 2.0 and -1 < qpos[2] < 1)            (:46-48)
        done = !(s[0] > (R)0.8 && s[0] < (R)2.0 && s[2] > (R)-1.0 && s[2] < (R)1.0);
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
        R act[ACT + 1], tau[WalkerModel::NB];
        step_begin(a, normalize, o, act, tau);
        if constexpr (MJC)
            Mjc::template advance<R>(s, s + 9, tau, (R)0.0025, SUBSTEPS, o.flags);       // MuJoCo's soft constraints
        else   // all eight body lanes in one value (dyn_two_legs.h): exact sines at the start, SUBSTEPS sub-steps
            Legs::template advance<R>(s, s + 9, tau, (R)0.0025, SUBSTEPS);
        R cz, cx, vz, vx;
        Legs::template com_of<R>(s, s + 9, cz, cx, vz, vx);
        step_end_com(s, act, cz, cx, vz, vx, obs, reward, done, o);
    }

    template <typename R> RL_HD static void com(const R* s, R* c4) {
        R cz, cx, vz, vx;
        Legs::template com_of<R>(s, s + 9, cz, cx, vz, vx);
        c4[0] = cx; c4[1] = cz; c4[2] = vx; c4[3] = vz;
    }
    static constexpr bool HAS_COM = true;
};

}  // namespace rl
