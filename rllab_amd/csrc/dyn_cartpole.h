// dyn_cartpole.h -- CartpoleEnv dynamics, single source for the gfx950 kernels
// and the host oracle build.
//
// Replaces, for one env copy:
//   NormalizedEnv.step            rllab/envs/normalized_env.py:78-92
//   Box2DEnv.step/forward_dynamics rllab/envs/box2d/box2d_env.py:119-183
//   CartpoleEnv.reset/compute_reward/is_current_done
//                                  rllab/envs/box2d/cartpole_env.py:28-56
//   world description             rllab/envs/box2d/models/cartpole.xml.mako:1-47
//   parser defaults (8 velocity / 3 position iterations, warm starting on,
//   sleeping off)                 rllab/envs/box2d/parser/xml_box2d.py:45-60
// The arithmetic of b2World::Step lives in pybox2d (Box2D 2.3.x, third party,
// not vendored in the reference).  It is restated here from the published
// algorithm: integrate velocities -> init joints + warm start -> N velocity
// iterations -> integrate positions (2 m / pi/2 per-step clamps) -> up to M
// position iterations -> clear forces.  Island joint order is
// [pole_joint (revolute), track_cart (prismatic)] (DFS from the pole, see
// DESIGN.md).  No contacts exist in this world (jointed bodies do not collide,
// pole/track share group -1).
//
// State (16 reals per env, SoA in HBM):
//   [0..5]   cart  centre x, y, angle, vx, vy, w      (centre == body origin)
//   [6..11]  pole  centre-of-mass x, y, angle, vx, vy, w
//   [12..13] revolute joint accumulated impulse (warm start)
//   [14..15] prismatic joint accumulated impulse (perp, angular)
// The impulses persist across reset(), exactly as the reference's long-lived
// b2World keeps them (CartpoleEnv.reset only rewrites body state).
#pragma once
#include "rl_math.h"

namespace rl {

struct Cartpole {
    static constexpr int OBS = 4;
    static constexpr int ACT = 1;
    static constexpr int STATE = 16;
    static constexpr int RESET_DRAWS = 4;     // uniform [0,1) draws per reset
    static constexpr bool RESET_NORMAL = false;
    static constexpr int KIND = 0;
    static constexpr bool TERMINATES = true;   // a path can end before max_path_length: is_current_done (cartpole_env.py:53-56)
    static constexpr bool HAS_COM = false;   // no subtree-COM export (get_body_com is a MujocoEnv method)

    template <typename R> struct C {
        static constexpr R cart_h = (R)0.86602540378443864676;   // 3/sqrt(12)
        static constexpr R cart_cy = (R)0.43301270189221932338;  // cart_h/2
        static constexpr R dt = (R)0.05;
        static constexpr R grav = (R)-10.0;
        static constexpr R inv_m_cart = (R)1.0;
        static constexpr R inv_i_cart = (R)5.76;                 // 1 / (25/144)
        static constexpr R inv_m_pole = (R)10.0;
        static constexpr R inv_i_pole = (R)118.81188118811880772;  // 1 / (0.1*1.01/12)
        static constexpr R pole_lc = (R)0.5;                     // pole local centre (0, 0.5)
        static constexpr R max_translation = (R)2.0;
        static constexpr R max_rotation = (R)1.57079632679489661923;  // 0.5*pi
        static constexpr R linear_slop = (R)0.005;
        static constexpr R angular_slop = (R)0.03490658503988659;  // 2/180*pi
        static constexpr R act_lb = (R)-10.0;
        static constexpr R act_ub = (R)10.0;
    };
    static constexpr int VEL_ITERS = 8;
    static constexpr int POS_ITERS = 3;

    template <typename R> RL_HD static void action_bounds(R* lb, R* ub) {
        lb[0] = C<R>::act_lb;
        ub[0] = C<R>::act_ub;
    }

    // b2Mat22::Solve / b2Mat33::Solve22 with a symmetric matrix [[k11,k12],[k12,k22]]
    template <typename R> RL_HD static void solve22(R k11, R k12, R k22, R bx, R by, R& x, R& y) {
        R det = k11 * k22 - k12 * k12;
        if (det != (R)0) det = (R)1 / det;
        x = det * (k22 * bx - k12 * by);
        y = det * (k11 * by - k12 * bx);
    }

    // CartpoleEnv.reset (cartpole_env.py:28-43): bodies back to the XML pose,
    // then cart x/vx and pole angle/w overwritten with U(-0.05*b, 0.05*b),
    // b = [2.4, 4, 0.2, 4].  The pole body origin stays at (0, cart_h): the
    // joint starts violated by |x| and is pulled together by the position solver.
    template <typename R> RL_HD static StepOpts<R> default_opts() { return make_opts<R>(0.0, 0.0, 1); }

    template <typename R> RL_HD static void reset(R* s, const R* u, int flags = 0, R /*link_len*/ = (R)1) {
        const R b0 = (R)2.4, b1 = (R)4.0, b2 = (R)0.2, b3 = (R)4.0, rr = (R)0.05;
        R lo0 = -rr * b0, lo1 = -rr * b1, lo2 = -rr * b2, lo3 = -rr * b3;
        R xpos = lo0 + u[0] * (rr * b0 - lo0);
        R xvel = lo1 + u[1] * (rr * b1 - lo1);
        R apos = lo2 + u[2] * (rr * b2 - lo2);
        R avel = lo3 + u[3] * (rr * b3 - lo3);
        s[0] = xpos; s[1] = C<R>::cart_cy; s[2] = (R)0; s[3] = xvel; s[4] = (R)0; s[5] = (R)0;
        R sn, cs;
        rl_sincos(apos, sn, cs);
        // centre = origin + R(apos) * (0, 0.5)
        s[6] = -sn * C<R>::pole_lc;
        s[7] = C<R>::cart_h + cs * C<R>::pole_lc;
        s[8] = apos; s[9] = (R)0; s[10] = (R)0; s[11] = avel;
        // engine option (not the reference's code path): the pole origin follows the cart,
        // so the hinge starts closed instead of being pulled together by the first position solve
        if (flags & CFG_POLE_FOLLOWS_CART) s[6] = s[6] + xpos;
    }

    // xml <state> list: cart xpos, cart xvel, pole apos, pole avel (cartpole.xml.mako:41-44)
    template <typename R> RL_HD static void observe(const R* s, R* o) {
        o[0] = s[0]; o[1] = s[3]; o[2] = s[8]; o[3] = s[11];
    }

    template <typename R> RL_HD static bool is_done(const R* s) {
        return rl_abs(s[0]) > (R)2.4 || rl_abs(s[8]) > (R)0.2;
    }

    // One b2World::Step(dt, 8, 3) with `force` applied on the cart along its
    // local +x at its origin (box2d_env.py:126-133, xml control anchor 0,0).
    template <typename R> RL_HD static void world_step(R* s, R force) {
        using K = C<R>;
        const R h = K::dt;
        const R mA = K::inv_m_cart, iA = K::inv_i_cart;   // revolute body A = cart
        const R mB = K::inv_m_pole, iB = K::inv_i_pole;   // revolute body B = pole
        R cx = s[0], cy = s[1], ca = s[2], cvx = s[3], cvy = s[4], cw = s[5];
        R px = s[6], py = s[7], pa = s[8], pvx = s[9], pvy = s[10], pw = s[11];
        R rix = s[12], riy = s[13], pix = s[14], piy = s[15];

        R sA, cA, sB, cB;
        rl_sincos(ca, sA, cA);
        rl_sincos(pa, sB, cB);

        // integrate velocities: v += h*(g + invMass*F).  World force = R(ca)*(force, 0).
        R fx = cA * force, fy = sA * force;
        cvx = cvx + h * (mA * fx);
        cvy = cvy + h * (K::grav + mA * fy);
        pvy = pvy + h * K::grav;

        // revolute init: rA = qA*(0, cart_h/2), rB = qB*(0, -0.5)
        R rAx = -sA * K::cart_cy, rAy = cA * K::cart_cy;
        R rBx = sB * K::pole_lc, rBy = -cB * K::pole_lc;
        R k11 = mA + mB + rAy * rAy * iA + rBy * rBy * iB;
        R k12 = -rAy * rAx * iA - rBy * rBx * iB;
        R k22 = mA + mB + rAx * rAx * iA + rBx * rBx * iB;
        // warm start (dtRatio == 1 for a fixed time step)
        cvx = cvx - mA * rix; cvy = cvy - mA * riy;
        cw = cw - iA * (rAx * riy - rAy * rix);
        pvx = pvx + mB * rix; pvy = pvy + mB * riy;
        pw = pw + iB * (rBx * riy - rBy * rix);
        // prismatic init (A = static track, B = cart): K = diag(mCart, iCart), s2 = a2 = 0
        const R q11 = mA, q12 = (R)0, q22 = iA;
        cvy = cvy + mA * pix;
        cw = cw + iA * piy;

        for (int it = 0; it < VEL_ITERS; ++it) {
            // revolute point constraint
            R cdx = pvx + (-pw * rBy) - cvx - (-cw * rAy);
            R cdy = pvy + (pw * rBx) - cvy - (cw * rAx);
            R ix, iy;
            solve22(k11, k12, k22, -cdx, -cdy, ix, iy);
            rix = rix + ix; riy = riy + iy;
            cvx = cvx - mA * ix; cvy = cvy - mA * iy;
            cw = cw - iA * (rAx * iy - rAy * ix);
            pvx = pvx + mB * ix; pvy = pvy + mB * iy;
            pw = pw + iB * (rBx * iy - rBy * ix);
            // prismatic: Cdot1 = (vB.y, wB)
            R dx, dy;
            solve22(q11, q12, q22, -cvy, -cw, dx, dy);
            pix = pix + dx; piy = piy + dy;
            cvy = cvy + mA * dx;
            cw = cw + iA * dy;
        }

        // integrate positions with Box2D's per-step motion clamps
        {
            R tx = h * cvx, ty = h * cvy;
            R tt = tx * tx + ty * ty;
            if (tt > K::max_translation * K::max_translation) {
                R ratio = K::max_translation / rl_sqrt(tt);
                cvx = cvx * ratio; cvy = cvy * ratio;
            }
            R rot = h * cw;
            if (rot * rot > K::max_rotation * K::max_rotation) {
                R ratio = K::max_rotation / rl_abs(rot);
                cw = cw * ratio;
            }
            cx = cx + h * cvx; cy = cy + h * cvy; ca = ca + h * cw;
        }
        {
            R tx = h * pvx, ty = h * pvy;
            R tt = tx * tx + ty * ty;
            if (tt > K::max_translation * K::max_translation) {
                R ratio = K::max_translation / rl_sqrt(tt);
                pvx = pvx * ratio; pvy = pvy * ratio;
            }
            R rot = h * pw;
            if (rot * rot > K::max_rotation * K::max_rotation) {
                R ratio = K::max_rotation / rl_abs(rot);
                pw = pw * ratio;
            }
            px = px + h * pvx; py = py + h * pvy; pa = pa + h * pw;
        }

        // position iterations (early exit when every joint is within slop)
        for (int it = 0; it < POS_ITERS; ++it) {
            rl_sincos(ca, sA, cA);
            rl_sincos(pa, sB, cB);
            R ax = -sA * K::cart_cy, ay = cA * K::cart_cy;
            R bx = sB * K::pole_lc, by = -cB * K::pole_lc;
            R Cx = px + bx - cx - ax;
            R Cy = py + by - cy - ay;
            R err = rl_sqrt(Cx * Cx + Cy * Cy);
            R p11 = mA + mB + iA * ay * ay + iB * by * by;
            R p12 = -iA * ax * ay - iB * bx * by;
            R p22 = mA + mB + iA * ax * ax + iB * bx * bx;
            R ix, iy;
            solve22(p11, p12, p22, Cx, Cy, ix, iy);
            ix = -ix; iy = -iy;
            cx = cx - mA * ix; cy = cy - mA * iy;
            ca = ca - iA * (ax * iy - ay * ix);
            px = px + mB * ix; py = py + mB * iy;
            pa = pa + iB * (bx * iy - by * ix);
            bool ok_rev = err <= K::linear_slop;
            // prismatic: C1 = (cart.y - track.y, cart.angle)
            R c1x = cy - K::cart_cy, c1y = ca;
            R lin_err = rl_abs(c1x), ang_err = rl_abs(c1y);
            R dx, dy;
            solve22(q11, q12, q22, -c1x, -c1y, dx, dy);
            cy = cy + mA * dx;
            ca = ca + iA * dy;
            bool ok_pri = (lin_err <= K::linear_slop) && (ang_err <= K::angular_slop);
            if (ok_rev && ok_pri) break;
        }

        s[0] = cx; s[1] = cy; s[2] = ca; s[3] = cvx; s[4] = cvy; s[5] = cw;
        s[6] = px; s[7] = py; s[8] = pa; s[9] = pvx; s[10] = pvy; s[11] = pw;
        s[12] = rix; s[13] = riy; s[14] = pix; s[15] = piy;
    }

    // Env.step for one env.  `a` is the policy action; with normalize != 0 it is
    // mapped to lb + (a+1)*0.5*(ub-lb) and clipped (normalized_env.py:81-83),
    // otherwise used as given.  Box2DEnv.forward_dynamics clips again
    // (box2d_env.py:123-124).  Reward is evaluated after the step with the action
    // handed to the inner env (cartpole_env.py:46-51).
    // `o.dact`: Box2DEnv._inject_action_noise adds its noise after the reward generator captured the action
    // (box2d_env.py:163-175); `o.frame_skip` world steps per env step, reward evaluated after the last (:171-179).
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
        if (o.dact) applied = act + o.dact[0];
        R force = rl_clamp(applied, K::act_lb, K::act_ub);
        for (int f = 0; f < o.frame_skip; ++f) world_step(s, force);
        done = is_done(s);
        R notdone = done ? (R)0 : (R)1;
        R sn, cs;
        rl_sincos(s[8], sn, cs);
        R ucost = (R)1e-5 * (act * act);
        R xcost = (R)1 - cs;
        reward = notdone * (R)10 - notdone * xcost - notdone * ucost;
        observe(s, obs);
    }
};

// CartpoleSwingupEnv (rllab/envs/box2d/cartpole_swingup_env.py:14-61): the same Box2D world
// (cartpole.xml.mako) and island solver, started hanging down.
//   reset   U([-1,-2,pi-1,-3], [1,2,pi+1,3]) on cart x / vx and pole angle / w, pole body origin left
//           at the XML pose (:29-41)
//   reward  -100 when done, else -1 beyond max_reward_cart_pos (= max_cart_pos, so unreachable),
//           else cos(pole angle)  (:43-52);   done = |cart x| > 3  (:54-56)
struct CartpoleSwingup : Cartpole {
    static constexpr int KIND = 4;

    template <typename R> RL_HD static void reset(R* s, const R* u, int flags = 0, R /*link_len*/ = (R)1) {
        const R PI_ = (R)3.14159265358979323846;
        const R lo0 = (R)-1, lo1 = (R)-2, lo2 = PI_ - (R)1, lo3 = (R)-3;
        const R hi0 = (R)1, hi1 = (R)2, hi2 = PI_ + (R)1, hi3 = (R)3;
        R xpos = lo0 + u[0] * (hi0 - lo0);
        R xvel = lo1 + u[1] * (hi1 - lo1);
        R apos = lo2 + u[2] * (hi2 - lo2);
        R avel = lo3 + u[3] * (hi3 - lo3);
        s[0] = xpos; s[1] = C<R>::cart_cy; s[2] = (R)0; s[3] = xvel; s[4] = (R)0; s[5] = (R)0;
        R sn, cs;
        rl_sincos(apos, sn, cs);
        s[6] = -sn * C<R>::pole_lc;
        s[7] = C<R>::cart_h + cs * C<R>::pole_lc;
        s[8] = apos; s[9] = (R)0; s[10] = (R)0; s[11] = avel;
        if (flags & CFG_POLE_FOLLOWS_CART) s[6] = s[6] + xpos;
    }

    template <typename R> RL_HD static bool is_done(const R* s) { return rl_abs(s[0]) > (R)3; }

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
        if (o.dact) applied = act + o.dact[0];
        const R force = rl_clamp(applied, K::act_lb, K::act_ub);
        for (int f = 0; f < o.frame_skip; ++f) world_step(s, force);
        done = is_done(s);
        R sn, cs;
        rl_sincos(s[8], sn, cs);
        reward = done ? (R)-100 : cs;
        observe(s, obs);
    }
};

}  // namespace rl
