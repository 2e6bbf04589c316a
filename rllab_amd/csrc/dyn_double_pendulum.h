// dyn_double_pendulum.h -- DoublePendulumEnv dynamics, single source for the gfx950
// kernels and the host oracle build.
//
// Replaces, for one env copy:
//   NormalizedEnv.step                 rllab/envs/normalized_env.py:78-92
//   Box2DEnv.step / forward_dynamics   rllab/envs/box2d/box2d_env.py:119-183 (torque control
//                                      = joint motor with speed +-1e5 and maxMotorTorque |a|,
//                                      :134-143)
//   DoublePendulumEnv.reset / compute_reward / get_tip_pos / is_current_done
//                                      rllab/envs/box2d/double_pendulum_env.py:32-61
//   world description                  rllab/envs/box2d/models/double_pendulum.xml.mako:1-40
//                                      (timestep 0.01, velitr 20, positr 20), frame_skip 2
//                                      (double_pendulum_env.py:16)
// World: two 0.1 x 1.0 links of density 5 (mass 0.5 each, COM at local (0,-0.5)) hanging
// from a static track; link_joint_1 = revolute(track, link1) at (0,0), link_joint_2 =
// revolute(link1, link2) at (0,-1) carrying the motor; gravity (0,-10).  The b2World::Step
// arithmetic (pybox2d, third party, absent) is restated from the published algorithm as in
// dyn_cartpole.h.  Island joint order [link_joint_2, link_joint_1] (DFS from link2).
//
// State (17 reals per env): link1 centre x,y,angle,vx,vy,w [0..5]; link2 [6..11];
// joint2 impulse x,y + motor impulse [12..14]; joint1 impulse x,y [15..16].  Impulses
// persist across reset() like the reference's long-lived b2World.
#pragma once
#include "rl_math.h"

namespace rl {

struct DoublePendulum {
    static constexpr int OBS = 6;
    static constexpr int ACT = 1;
    static constexpr int STATE = 17;
    static constexpr int RESET_DRAWS = 4;  // N(0,1): angle1, angle2, w1, w2
    static constexpr bool RESET_NORMAL = true;
    static constexpr int KIND = 1;
    static constexpr bool TERMINATES = false;   // is_current_done returns False (double_pendulum_env.py:60-61)
    static constexpr bool HAS_COM = false;   // no subtree-COM export (get_body_com is a MujocoEnv method)
    static constexpr int VEL_ITERS = 20;
    static constexpr int POS_ITERS = 20;
    static constexpr int FRAME_SKIP = 2;

    template <typename R> struct C {
        static constexpr R dt = (R)0.01;
        static constexpr R grav = (R)-10.0;
        static constexpr R inv_m = (R)2.0;                       // 1 / 0.5
        static constexpr R inv_i = (R)23.762376237623762;        // 1 / (0.5 * 1.01 / 12)
        static constexpr R half = (R)0.5;                        // |local centre| = link_len / 2
        static constexpr R max_translation = (R)2.0;
        static constexpr R max_rotation = (R)1.57079632679489661923;
        static constexpr R linear_slop = (R)0.005;
        static constexpr R act_lb = (R)-50.0;
        static constexpr R act_ub = (R)50.0;
        static constexpr R motor_speed = (R)1e5;
    };

    template <typename R> RL_HD static void action_bounds(R* lb, R* ub) {
        lb[0] = C<R>::act_lb;
        ub[0] = C<R>::act_ub;
    }

    // Link geometry.  Both links are boxes 0.1 x link_len of density 5 hinged at one end (double_pendulum.xml.mako:9-29);
    // link_len is 1 unless DoublePendulumEnv(template_args=dict(noise=True)) drew another (double_pendulum_env.py:17-21).
    // len == 1 takes the compile-time constants, so the default keeps its bits.
    template <typename R> struct Geo { R len, half, inv_m, inv_i; };
    template <typename R> RL_HD static Geo<R> geo(R len) {
        Geo<R> g;
        if (len == (R)1) {
            g.len = (R)1; g.half = C<R>::half; g.inv_m = C<R>::inv_m; g.inv_i = C<R>::inv_i;
        } else {
            const R mass = (R)5 * (R)0.1 * len;                         // density * width * length
            g.len = len;
            g.half = (R)0.5 * len;
            g.inv_m = (R)1 / mass;
            g.inv_i = (R)12 / (mass * ((R)0.01 + len * len));           // box about its centre: m (w^2 + l^2) / 12
        }
        return g;
    }

    template <typename R> RL_HD static void solve22(R k11, R k12, R k22, R bx, R by, R& x, R& y) {
        R det = k11 * k22 - k12 * k12;
        if (det != (R)0) det = (R)1 / det;
        x = det * (k22 * bx - k12 * by);
        y = det * (k11 * by - k12 * bx);
    }

    // reset: bodies back to the XML pose, then angles / angular velocities overwritten with
    // N(0, [0.1, 0.1, 0.01, 0.01]) (double_pendulum_env.py:32-41).  Body origins stay at
    // (0,0) and (0,-1): joint 2 starts violated and is pulled together by the position solver.
    template <typename R> RL_HD static StepOpts<R> default_opts() { return make_opts<R>(0.0, 0.0, FRAME_SKIP); }

    template <typename R> RL_HD static void reset(R* s, const R* z, int /*flags*/ = 0, R link_len = (R)1) {
        const Geo<R> g = geo<R>(link_len);
        const R a1 = z[0] * (R)0.1, a2 = z[1] * (R)0.1, w1 = z[2] * (R)0.01, w2 = z[3] * (R)0.01;
        R s1, c1, s2, c2;
        rl_sincos(a1, s1, c1);
        rl_sincos(a2, s2, c2);
        // centre = origin + R(a) * (0, -len / 2); link2's origin sits at (0, -len)
        s[0] = s1 * g.half;            s[1] = -c1 * g.half;
        s[2] = a1; s[3] = (R)0; s[4] = (R)0; s[5] = w1;
        s[6] = s2 * g.half;            s[7] = -g.len - c2 * g.half;
        s[8] = a2; s[9] = (R)0; s[10] = (R)0; s[11] = w2;
    }

    // xml <state> list: sin/cos(link1 angle), link1 avel, sin/cos(link2 angle), link2 avel
    template <typename R> RL_HD static void observe(const R* s, R* o) {
        R s1, c1, s2, c2;
        rl_sincos(s[2], s1, c1);
        rl_sincos(s[8], s2, c2);
        o[0] = s1; o[1] = c1; o[2] = s[5];
        o[3] = s2; o[4] = c2; o[5] = s[11];
    }

    template <typename R> RL_HD static void clamp_motion(R h, R& vx, R& vy, R& w) {
        using K = C<R>;
        R tx = h * vx, ty = h * vy;
        R tt = tx * tx + ty * ty;
        if (tt > K::max_translation * K::max_translation) {
            R ratio = K::max_translation / rl_sqrt(tt);
            vx = vx * ratio; vy = vy * ratio;
        }
        R rot = h * w;
        if (rot * rot > K::max_rotation * K::max_rotation) {
            R ratio = K::max_rotation / rl_abs(rot);
            w = w * ratio;
        }
    }

    // one b2World::Step(0.01, 20, 20) with the joint-2 motor set from `torque`
    template <typename R> RL_HD static void world_step(R* s, R torque, const Geo<R>& g = geo<R>((R)1)) {
        using K = C<R>;
        const R h = K::dt, m = g.inv_m, ii = g.inv_i;
        R x1 = s[0], y1 = s[1], a1 = s[2], vx1 = s[3], vy1 = s[4], w1 = s[5];
        R x2 = s[6], y2 = s[7], a2 = s[8], vx2 = s[9], vy2 = s[10], w2 = s[11];
        R j2x = s[12], j2y = s[13], jm = s[14], j1x = s[15], j1y = s[16];
        const R motor_speed = (torque > (R)0) ? K::motor_speed : -K::motor_speed;
        const R max_impulse = h * rl_abs(torque);

        // integrate velocities (gravity only; the motor acts through its joint)
        vy1 = vy1 + h * K::grav;
        vy2 = vy2 + h * K::grav;

        R s1, c1, s2, c2;
        rl_sincos(a1, s1, c1);
        rl_sincos(a2, s2, c2);
        // joint 2 (A = link1, B = link2): rA = q1*(0,-0.5), rB = q2*(0,0.5)
        R rAx = s1 * g.half, rAy = -c1 * g.half;
        R rBx = -s2 * g.half, rBy = c2 * g.half;
        R k11 = m + m + rAy * rAy * ii + rBy * rBy * ii;
        R k12 = -rAy * rAx * ii - rBy * rBx * ii;
        R k22 = m + m + rAx * rAx * ii + rBx * rBx * ii;
        const R motor_mass = (R)1 / (ii + ii);
        // warm start joint 2 (point impulse + motor impulse)
        vx1 = vx1 - m * j2x; vy1 = vy1 - m * j2y;
        w1 = w1 - ii * ((rAx * j2y - rAy * j2x) + jm);
        vx2 = vx2 + m * j2x; vy2 = vy2 + m * j2y;
        w2 = w2 + ii * ((rBx * j2y - rBy * j2x) + jm);
        // joint 1 (A = static track, B = link1): rB = q1*(0,0.5)
        R tBx = -s1 * g.half, tBy = c1 * g.half;
        R t11 = m + tBy * tBy * ii;
        R t12 = -tBy * tBx * ii;
        R t22 = m + tBx * tBx * ii;
        vx1 = vx1 + m * j1x; vy1 = vy1 + m * j1y;
        w1 = w1 + ii * (tBx * j1y - tBy * j1x);

        for (int it = 0; it < VEL_ITERS; ++it) {
            // joint 2 motor
            {
                R cdot = w2 - w1 - motor_speed;
                R imp = -motor_mass * cdot;
                R old = jm;
                jm = rl_clamp(old + imp, -max_impulse, max_impulse);
                imp = jm - old;
                w1 = w1 - ii * imp;
                w2 = w2 + ii * imp;
            }
            // joint 2 point constraint
            {
                R cdx = vx2 + (-w2 * rBy) - vx1 - (-w1 * rAy);
                R cdy = vy2 + (w2 * rBx) - vy1 - (w1 * rAx);
                R ix, iy;
                solve22(k11, k12, k22, -cdx, -cdy, ix, iy);
                j2x = j2x + ix; j2y = j2y + iy;
                vx1 = vx1 - m * ix; vy1 = vy1 - m * iy;
                w1 = w1 - ii * (rAx * iy - rAy * ix);
                vx2 = vx2 + m * ix; vy2 = vy2 + m * iy;
                w2 = w2 + ii * (rBx * iy - rBy * ix);
            }
            // joint 1 point constraint (vA = wA = 0)
            {
                R cdx = vx1 + (-w1 * tBy);
                R cdy = vy1 + (w1 * tBx);
                R ix, iy;
                solve22(t11, t12, t22, -cdx, -cdy, ix, iy);
                j1x = j1x + ix; j1y = j1y + iy;
                vx1 = vx1 + m * ix; vy1 = vy1 + m * iy;
                w1 = w1 + ii * (tBx * iy - tBy * ix);
            }
        }

        // integrate positions (island body order: link2, link1)
        clamp_motion(h, vx2, vy2, w2);
        x2 = x2 + h * vx2; y2 = y2 + h * vy2; a2 = a2 + h * w2;
        clamp_motion(h, vx1, vy1, w1);
        x1 = x1 + h * vx1; y1 = y1 + h * vy1; a1 = a1 + h * w1;

        for (int it = 0; it < POS_ITERS; ++it) {
            rl_sincos(a1, s1, c1);
            rl_sincos(a2, s2, c2);
            bool ok2, ok1;
            {   // joint 2
                R ax = s1 * g.half, ay = -c1 * g.half;
                R bx = -s2 * g.half, by = c2 * g.half;
                R Cx = x2 + bx - x1 - ax, Cy = y2 + by - y1 - ay;
                R err = rl_sqrt(Cx * Cx + Cy * Cy);
                R p11 = m + m + ii * ay * ay + ii * by * by;
                R p12 = -ii * ax * ay - ii * bx * by;
                R p22 = m + m + ii * ax * ax + ii * bx * bx;
                R ix, iy;
                solve22(p11, p12, p22, Cx, Cy, ix, iy);
                ix = -ix; iy = -iy;
                x1 = x1 - m * ix; y1 = y1 - m * iy;
                a1 = a1 - ii * (ax * iy - ay * ix);
                x2 = x2 + m * ix; y2 = y2 + m * iy;
                a2 = a2 + ii * (bx * iy - by * ix);
                ok2 = err <= K::linear_slop;
            }
            {   // joint 1: anchor of the static track is the world origin; uses link1's pose
                // as updated by joint 2 in this iteration
                R sn, cs;
                rl_sincos(a1, sn, cs);
                R bx = -sn * g.half, by = cs * g.half;
                R Cx = x1 + bx, Cy = y1 + by;
                R err = rl_sqrt(Cx * Cx + Cy * Cy);
                R p11 = m + ii * by * by;
                R p12 = -ii * bx * by;
                R p22 = m + ii * bx * bx;
                R ix, iy;
                solve22(p11, p12, p22, Cx, Cy, ix, iy);
                ix = -ix; iy = -iy;
                x1 = x1 + m * ix; y1 = y1 + m * iy;
                a1 = a1 + ii * (bx * iy - by * ix);
                ok1 = err <= K::linear_slop;
            }
            if (ok2 && ok1) break;
        }

        s[0] = x1; s[1] = y1; s[2] = a1; s[3] = vx1; s[4] = vy1; s[5] = w1;
        s[6] = x2; s[7] = y2; s[8] = a2; s[9] = vx2; s[10] = vy2; s[11] = w2;
        s[12] = j2x; s[13] = j2y; s[14] = jm; s[15] = j1x; s[16] = j1y;
    }

    template <typename R>
    RL_HD static void step(R* s, const R* a, int normalize, R* obs, R& reward, bool& done,
                           const StepOpts<R>& o = default_opts<R>()) {
        using K = C<R>;
        R act = a[0];
        if (normalize) {
            act = K::act_lb + (act + (R)1) * (R)0.5 * (K::act_ub - K::act_lb);
            act = rl_clamp(act, K::act_lb, K::act_ub);
        }
        R applied = act;
        if (o.dact) applied = act + o.dact[0];                      // _inject_action_noise (box2d_env.py:219-226)
        const R torque = rl_clamp(applied, K::act_lb, K::act_ub);   // forward_dynamics clips (box2d_env.py:123-124)
        const Geo<R> g = geo<R>(o.link_len);
        for (int f = 0; f < o.frame_skip; ++f) world_step(s, torque, g);
        // reward = -|tip - (0, 2 link_len)|, tip = link2.position - link_len*(sin a2, cos a2)
        // (double_pendulum_env.py:43-58); link2.position = centre - R(a2)*(0,-link_len/2)
        R s2, c2;
        rl_sincos(s[8], s2, c2);
        const R ox = s[6] - s2 * g.half, oy = s[7] + c2 * g.half;
        R tx, ty, goal;
        if (g.len == (R)1) { tx = ox - s2; ty = oy - c2; goal = (R)2; }
        else { tx = ox - g.len * s2; ty = oy - g.len * c2; goal = g.len * (R)2; }
        const R dx = tx, dy = ty - goal;
        reward = -rl_sqrt(dx * dx + dy * dy);
        done = false;
        observe(s, obs);
    }
};

}  // namespace rl
