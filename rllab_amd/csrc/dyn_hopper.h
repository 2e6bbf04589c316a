// dyn_hopper.h -- HopperEnv-style env: planar 4-body / 6-DoF one-legged hopper (torso, thigh, leg, foot) with joint
// dampers, armature, torque motors, gravity and capsule-floor contacts; single source for the gfx950 kernels and
// the host oracle build.
//
// Replaces, for one env copy:
//   HopperEnv.step / get_current_obs          rllab/envs/mujoco/hopper_env.py:38-62
//   MujocoEnv.reset_mujoco / forward_dynamics rllab/envs/mujoco/mujoco_env.py:109-116,184-191
//   MjModel.step / forward / _compute_subtree rllab/mujoco_py/mjcore.py:46-84
//   model constants                           vendor/mujoco_models/hopper.xml:3-49
//                                             (through gen_planar_constants.py -> hopper_constants.h)
//   NormalizedEnv.step                        rllab/envs/normalized_env.py:78-92
// "-style": rigid-body tree, joint passive forces and actuation follow the MJCF; joint limits and the capsule / plane
// contacts are the penalty model of dyn_legged.h, and one 0.02 s MuJoCo step (RK4 in the MJCF, frame_skip 1) is
// integrated as 8 semi-implicit Euler sub-steps of 0.0025 s.  The observed qfrc_constraint is the generalised force
// of exactly those penalty terms (PlanarTree::constraint_forces) at the observed state, in MuJoCo's joint convention.
//
// Plane coordinates (P1, P2) = (z, x); the leg hinges are declared about -y (hopper::SIGN = -1).
// State (12 reals, tree convention): q[6] = [z (absolute torso height = MuJoCo's rootz, ref 1.25), x, rooty,
// thigh, leg, foot], qd[6].
#pragma once
#include "dyn_legged.h"
#include "dyn_mjc.h"
#include "hopper_constants.h"

namespace rl {

RL_LEGGED_CONSTANTS(HopperK, hopper);
using HopperModel = LeggedModel<HopperK>;

// hopper.xml:4-5: geoms margin 0.001, solref ".02 1", solimp ".8 .8 .01"; the joints set none (MuJoCo's defaults)
struct HopperMjcPar {
    RL_HD static constexpr MjcSol limit() { return MjcSol{0.02, 1.0, 0.9, 0.95, 0.001, 0.0}; }
    RL_HD static constexpr MjcSol contact() { return MjcSol{0.02, 1.0, 0.8, 0.8, 0.01, 0.001}; }
};

struct Hopper {
    static constexpr int OBS = 20;
    static constexpr int ACT = 3;
    static constexpr int STATE = 12;
    static constexpr int RESET_DRAWS = 12;  // N(0,1): 6 for qpos, 6 for qvel (MuJoCo order)
    static constexpr bool RESET_NORMAL = true;
    static constexpr int KIND = 6;
    static constexpr bool TERMINATES = true;   // a path can end before max_path_length (hopper_env.py:57-61)
    static constexpr int SUBSTEPS = 8;      // 8 x 0.0025 s = one 0.02 s MuJoCo step, frame_skip 1
    static constexpr int NQ = 6;
    using Tree = PlanarTree<HopperModel>;
    using Mjc = MjcTree<HopperModel, HopperMjcPar>;       // limit_model / contact_model = "mujoco" (dyn_mjc.h)

    template <typename R> RL_HD static void action_bounds(R* lb, R* ub) {
        RL_UNROLL
        for (int k = 0; k < ACT; ++k) { lb[k] = -(R)hopper::GEAR[1 + k]; ub[k] = (R)hopper::GEAR[1 + k]; }
    }

    // qpos = init + 0.01 N(0,1) with init_qpos = [1.25, 0, ...], qvel = 0.1 N(0,1), MuJoCo order
    // [rootz, rootx, rooty, joints] and MuJoCo sign convention for the joints
    template <typename R> RL_HD static StepOpts<R> default_opts() { return make_opts<R>(0.01, 1.0, 1); }

    template <typename R> RL_HD static void reset(R* s, const R* z, int /*flags*/ = 0, R /*link_len*/ = (R)1) {
        s[0] = (R)1.25 + z[0] * (R)0.01;
        s[1] = z[1] * (R)0.01;
        s[2] = z[2] * (R)0.01;
        s[6] = z[6] * (R)0.1;
        s[7] = z[7] * (R)0.1;
        s[8] = z[8] * (R)0.1;
        RL_UNROLL
        for (int i = 3; i < NQ; ++i) {
            s[i] = (R)hopper::SIGN[i - 2] * (z[i] * (R)0.01);
            s[NQ + i] = (R)hopper::SIGN[i - 2] * (z[NQ + i] * (R)0.1);
        }
    }

    // obs = [qpos[0:1], qpos[2:], clip(qvel, +-10), clip(qfrc_constraint, +-10), com_subtree(torso)] in MuJoCo's
    // convention (hopper_env.py:38-46)
    template <typename R> RL_HD static void observe(const R* s, R* o) {
        R cz, cx, vz, vx;
        Tree::template com<R>(s, s + NQ, cz, cx, vz, vx);
        write_obs(s, cx, cz, o);
    }

    // qfrc: data.qfrc_constraint as the constraint solve of the step's last sub-step left it (limit_model / contact_model =
    // "mujoco"); null: the penalty models' generalised force at the state (also what reset() / get_current_obs report)
    template <typename R> RL_HD static void write_obs(const R* s, R cx, R cz, R* o, const R* qfrc = nullptr) {
        R qf[NQ];
        if (qfrc) {
            RL_UNROLL
            for (int i = 0; i < NQ; ++i) qf[i] = qfrc[i];
        } else {
            Tree::template constraint_forces<R>(s, s + NQ, qf);
        }
        o[0] = s[0];
        o[1] = s[2];
        RL_UNROLL
        for (int i = 3; i < NQ; ++i) o[i - 1] = (R)hopper::SIGN[i - 2] * s[i];
        RL_UNROLL
        for (int i = 0; i < NQ; ++i) {
            const R sg = (i >= 3) ? (R)hopper::SIGN[i - 2] : (R)1;
            o[5 + i] = rl_clamp(sg * s[NQ + i], (R)-10, (R)10);
            o[11 + i] = rl_clamp(sg * qf[i], (R)-10, (R)10);
        }
        o[17] = cx; o[18] = (R)0; o[19] = cz;
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
        R act[ACT], tau[HopperModel::NB];
        tau[0] = (R)0;
        R ctrl_cost = (R)0;
        RL_UNROLL
        for (int k = 0; k < ACT; ++k) {
            const R ub = (R)hopper::GEAR[1 + k], lb = -ub;
            R v = a[k];
            if (normalize) v = rl_clamp(lb + (v + (R)1) * (R)0.5 * (ub - lb), lb, ub);
            act[k] = rl_clamp(v, lb, ub);                        // action = clip(action, *bounds); ctrllimited motor
            R applied = act[k];
            if (o.dact) applied = rl_clamp(act[k] + o.dact[k], lb, ub);   // ctrl = action + noise, ctrllimited
            tau[1 + k] = (R)hopper::SIGN[1 + k] * applied;       // gear 1: torque = ctrl, about the MJCF axis
            const R sc = act[k] / ((ub - lb) * (R)0.5);
            ctrl_cost = ctrl_cost + sc * sc;
        }
        R q[NQ], qd[NQ];
        RL_UNROLL
        for (int i = 0; i < NQ; ++i) { q[i] = s[i]; qd[i] = s[NQ + i]; }
        constexpr bool mjc = MJC;
        R qfrc[NQ];
        if constexpr (MJC) {
            Mjc::template advance<R>(q, qd, tau, (R)0.0025, SUBSTEPS, o.flags, qfrc);    // MuJoCo's soft constraints
        } else {
            R sn[HopperModel::NB], cs[HopperModel::NB];
            Tree::template angles<R>(q, sn, cs);
            for (int it = 0; it < SUBSTEPS; ++it) Tree::template substep<R>(q, qd, tau, (R)0.0025, sn, cs);
        }
        RL_UNROLL
        for (int i = 0; i < NQ; ++i) { s[i] = q[i]; s[NQ + i] = qd[i]; }
        R cz, cx, vz, vx;
        Tree::template com<R>(q, qd, cz, cx, vz, vx);
        write_obs(s, cx, cz, obs, mjc ? qfrc : nullptr);
        // reward = comvel_x + alive_coeff - 0.5 * ctrl_cost_coeff * sum((action / scaling)^2), alive_coeff 1,
        // ctrl_cost_coeff 0.01                                                    (hopper_env.py:27-28,53-55)
        reward = vx + o.alive_coeff - (R)0.5 * o.ctrl_cost_coeff * ctrl_cost;
        // notdone = isfinite(state).all() and (|state[3:]| < 100).all() and state[0] > .7 and |state[2]| < .2,
        // state = [qpos, qvel]                                                    (:56-60)
        bool ok = (s[0] > (R)0.7) && (rl_abs(s[2]) < (R)0.2) && (rl_abs(s[0]) < (R)1e30) &&
                  (rl_abs(s[1]) < (R)1e30);
        RL_UNROLL
        for (int i = 3; i < 2 * NQ; ++i) ok = ok && (rl_abs(s[i]) < (R)100);
        done = !ok;
    }

    template <typename R> RL_HD static void com(const R* s, R* c4) {
        R cz, cx, vz, vx;
        Tree::template com<R>(s, s + NQ, cz, cx, vz, vx);
        c4[0] = cx; c4[1] = cz; c4[2] = vx; c4[3] = vz;
    }
    static constexpr bool HAS_COM = true;
};

}  // namespace rl
