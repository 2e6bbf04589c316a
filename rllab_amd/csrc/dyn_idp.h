// dyn_idp.h -- InvertedDoublePendulumEnv-style env: a cart on a rail (slide joint, force motor) carrying two hinged
// capsule poles; minimal coordinates, single source for the gfx950 kernels and the host oracle build.
//
// Replaces, for one env copy:
//   InvertedDoublePendulumEnv.step / get_current_obs / reset_mujoco
//                                             rllab/envs/mujoco/inverted_double_pendulum_env.py:24-58
//   MujocoEnv.forward_dynamics                rllab/envs/mujoco/mujoco_env.py:184-191
//   MjModel.step / forward                    rllab/mujoco_py/mjcore.py:46-49
//   model constants                           vendor/mujoco_models/inverted_double_pendulum.xml.mako:19-118
//                                             (noise = False: both poles 0.6 long)
//   NormalizedEnv.step                        rllab/envs/normalized_env.py:78-92
// "-style": the MJCF's bodies (capsule cart r 0.1 x 0.2, two capsule poles r 0.045 x 0.6, density 1000), joint
// damping 0.05, gravity (1e-5, 0, -9.81), motor gear 500 with ctrlrange +-1 and the +-10 m slider range (as the
// engine's penalty limit, whose force is the observed qfrc_constraint); one env step = frame_skip 2 x 0.01 s
// (RK4 in the MJCF) integrated as 8 semi-implicit Euler sub-steps of 0.0025 s.
//
// Equations of motion in (x, phi1, phi2), phi = absolute pole angles from the vertical towards +x:
//   [ mt        A c1      B c2   ] [xdd ]   [ F - d xd + mt gx + A s1 w1^2 + B s2 w2^2 + f_limit ]
//   [ A c1      J1        D cd   ] [w1d ] = [ A (gx c1 - gz s1) - D sd w2^2 - d th1d + d th2d    ]
//   [ B c2      D cd      J2     ] [w2d ]   [ B (gx c2 - gz s2) + D sd w1^2 - d th2d              ]
// A = m1 lc + m2 l1, B = m2 lc, D = m2 l1 lc, J1 = m1 lc^2 + m2 l1^2 + I, J2 = m2 lc^2 + I, cd / sd = cos / sin(phi1 - phi2).
// State (6 reals, MuJoCo order): qpos = [x, hinge, hinge2 (relative)], qvel.
#pragma once
#include "rl_math.h"

namespace rl {

namespace idp {
constexpr double PI = 3.14159265358979323846;
constexpr double RHO = 1000.0;
constexpr double cap_mass(double hl, double r) { return RHO * (PI * r * r * 2 * hl + 4.0 / 3.0 * PI * r * r * r); }
// transverse inertia of a solid capsule about its centre
constexpr double cap_inertia(double hl, double r) {
    const double L = 2 * hl, mc = RHO * PI * r * r * L, ms = RHO * 4.0 / 3.0 * PI * r * r * r;
    return mc * (L * L / 12 + r * r / 4) + ms * (83.0 / 320 * r * r + (L / 2 + 3 * r / 8) * (L / 2 + 3 * r / 8));
}
constexpr double L1 = 0.6, LC = 0.3, L2 = 0.6;
constexpr double M0 = cap_mass(0.1, 0.1), M1 = cap_mass(0.3, 0.045), M2 = M1;
constexpr double IP = cap_inertia(0.3, 0.045);
constexpr double MT = M0 + M1 + M2;
constexpr double A_ = M1 * LC + M2 * L1, B_ = M2 * LC, D_ = M2 * L1 * LC;
constexpr double J1 = M1 * LC * LC + M2 * L1 * L1 + IP, J2 = M2 * LC * LC + IP;
}  // namespace idp

struct InvertedDoublePendulum {
    static constexpr int OBS = 11;
    static constexpr int ACT = 1;
    static constexpr int STATE = 6;
    static constexpr int RESET_DRAWS = 1;      // one uniform [0,1) draw: the first pole's start angle
    static constexpr bool RESET_NORMAL = false;
    static constexpr int KIND = 7;
    static constexpr bool TERMINATES = true;   // a path can end before max_path_length: tip height <= 1 (inverted_double_pendulum_env.py:44)
    static constexpr bool HAS_COM = false;   // no subtree-COM export (get_body_com is a MujocoEnv method)
    static constexpr int SUBSTEPS = 8;         // 8 x 0.0025 s = frame_skip 2 x timestep 0.01

    static constexpr double PI = idp::PI, L1 = idp::L1, LC = idp::LC, L2 = idp::L2, MT = idp::MT, A_ = idp::A_,
                            B_ = idp::B_, D_ = idp::D_, J1 = idp::J1, J2 = idp::J2;
    static constexpr double DAMP = 0.05, GX = 1e-5, GZ = -9.81, GEAR = 500.0, RANGE = 10.0;
    static constexpr double LIMIT_K = 2.0e3, LIMIT_B = 15.0;

    template <typename R> RL_HD static void action_bounds(R* lb, R* ub) { lb[0] = (R)-1; ub[0] = (R)1; }

    // random_start: qpos[1] = (U[0,1) - 0.5) * 40 / 180 * pi, everything else 0   (:47-58)
    template <typename R> RL_HD static StepOpts<R> default_opts() { return make_opts<R>(0.0, 0.0, 1); }

    template <typename R> RL_HD static void reset(R* s, const R* u, int flags = 0, R /*link_len*/ = (R)1) {
        RL_UNROLL
        for (int i = 0; i < STATE; ++i) s[i] = (R)0;
        // random_start=False keeps the model's initial pose (inverted_double_pendulum_env.py:20,47-58)
        if (!(flags & CFG_FIXED_START)) s[1] = (u[0] - (R)0.5) * (R)(40.0 / 180.0 * PI);
    }

    template <typename R> RL_HD static R limit_force(R x, R xd) {
        const R viol = x - rl_clamp(x, (R)-RANGE, (R)RANGE);
        const R damp = (viol != (R)0) ? (R)LIMIT_B * xd : (R)0;
        return -((R)LIMIT_K * viol) - damp;
    }

    // obs = [x, sin(hinges), cos(hinges), clip(qvel, +-10), clip(qfrc_constraint, +-10)]   (:24-32)
    template <typename R> RL_HD static void observe(const R* s, R* o) {
        R s1, c1, s2, c2;
        rl_sincos(s[1], s1, c1);
        rl_sincos(s[2], s2, c2);
        o[0] = s[0];
        o[1] = s1; o[2] = s2; o[3] = c1; o[4] = c2;
        RL_UNROLL
        for (int i = 0; i < 3; ++i) o[5 + i] = rl_clamp(s[3 + i], (R)-10, (R)10);
        o[8] = rl_clamp(limit_force(s[0], s[3]), (R)-10, (R)10);
        o[9] = (R)0; o[10] = (R)0;
    }

    template <typename R> RL_HD static void substep(R* s, R force, R h) {
        const R w1 = s[4], w2 = s[4] + s[5];
        R s1, c1, s2, c2;
        rl_sincos(s[1], s1, c1);
        rl_sincos(s[1] + s[2], s2, c2);
        const R cd = c1 * c2 + s1 * s2, sd = s1 * c2 - c1 * s2;
        const R a = (R)MT, b = (R)A_ * c1, c = (R)B_ * c2, d = (R)J1, e = (R)D_ * cd, f = (R)J2;
        const R r0 = (((force - (R)DAMP * s[3]) + (R)(MT * GX)) + (R)A_ * s1 * (w1 * w1)) + (R)B_ * s2 * (w2 * w2) +
                     limit_force(s[0], s[3]);
        const R r1 = ((R)A_ * ((R)GX * c1 - (R)GZ * s1) - (R)D_ * sd * (w2 * w2)) + ((R)DAMP * s[5] - (R)DAMP * s[4]);
        const R r2 = ((R)B_ * ((R)GX * c2 - (R)GZ * s2) + (R)D_ * sd * (w1 * w1)) - (R)DAMP * s[5];
        // symmetric 3x3 solve by the adjugate
        const R Aa = d * f - e * e, Bb = c * e - b * f, Cc = b * e - c * d;
        const R Dd = a * f - c * c, Ee = b * c - a * e, Ff = a * d - b * b;
        const R inv = (R)1 / (a * Aa + (b * Bb + c * Cc));
        const R xdd = (Aa * r0 + (Bb * r1 + Cc * r2)) * inv;
        const R w1d = (Bb * r0 + (Dd * r1 + Ee * r2)) * inv;
        const R w2d = (Cc * r0 + (Ee * r1 + Ff * r2)) * inv;
        s[3] = s[3] + h * xdd;
        s[4] = s[4] + h * w1d;
        s[5] = s[5] + h * (w2d - w1d);
        s[0] = s[0] + h * s[3];
        s[1] = s[1] + h * s[4];
        s[2] = s[2] + h * s[5];
    }

    template <typename R>
    RL_HD static void step(R* s, const R* a, int normalize, R* obs, R& reward, bool& done,
                           const StepOpts<R>& o = default_opts<R>()) {
        R v = a[0];
        if (normalize) v = rl_clamp((R)-1 + (v + (R)1) * (R)0.5 * (R)2, (R)-1, (R)1);   // lb + (a + 1) * 0.5 * (ub - lb)
        R applied = rl_clamp(v, (R)-1, (R)1);           // action = clip(action, *bounds)
        if (o.dact) applied = applied + o.dact[0];      // ctrl = inject_action_noise(action)
        const R ctrl = rl_clamp(applied, (R)-1, (R)1);  // ctrllimited motor
        const R force = (R)GEAR * ctrl;
        for (int it = 0; it < SUBSTEPS; ++it) substep<R>(s, force, (R)0.0025);
        observe<R>(s, obs);
        // tip site: x, _, y = site_xpos[0]                                                  (:38)
        R s1, c1, s2, c2;
        rl_sincos(s[1], s1, c1);
        rl_sincos(s[1] + s[2], s2, c2);
        const R tx = (s[0] + (R)L1 * s1) + (R)L2 * s2;
        const R ty = (R)L1 * c1 + (R)L2 * c2;
        const R dist_penalty = (R)0.01 * (tx * tx) + (ty - (R)2) * (ty - (R)2);
        const R vel_penalty = (R)1e-3 * (s[4] * s[4]) + (R)5e-3 * (s[5] * s[5]);
        reward = ((R)10 - dist_penalty) - vel_penalty;   // alive_bonus 10                     (:39-43)
        done = ty <= (R)1;                                 // (:44)
    }
};

}  // namespace rl
