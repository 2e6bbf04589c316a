// oracle/env_host.cpp -- TEST INFRASTRUCTURE (never imported by the product).
//
// Host build of the env dynamics headers (rllab_amd/csrc/dyn_*.h): the same
// source the gfx950 kernels are compiled from, instantiated
//   * in float  -- the bit-exact leg: GPU kernel output must equal this, bit for
//     bit, on identical states / actions / injected draws.  It pins the device
//     toolchain, the SoA plane indexing, the auto-reset / horizon logic and the
//     RNG plumbing, which is everything that differs between the two builds;
//   * in double -- the physics leg: compared against the INDEPENDENT float64
//     restatements in oracle/np_*.py (written from the reference's XML / MJCF
//     constants with a different formulation), which is what catches a wrong
//     equation that a shared source cannot.
// Parity status: the reference's own arithmetic for these envs lives in pybox2d /
// MuJoCo 1.31 (third party, absent) and its tests hold no golden vectors for it
//  => env dynamics are "parity unpinned" against the reference.
//
// Build: see oracle/Makefile (the ROCm clang++, -O2 -ffp-contract=on -mfma).
#include <stdint.h>
#include <string.h>

#include <array>
#include <vector>

#include "../rllab_amd/csrc/envs.h"

namespace {

template <class E, typename R>
int reset_t(R* s, const R* draws) { E::template reset<R>(s, draws); return 0; }

template <class E, typename R>
int observe_t(const R* s, R* o) { E::template observe<R>(s, o); return 0; }

template <class E, typename R>
int step_t(R* s, const R* a, int normalize, R* obs, R* reward, int* done) {
    bool d;
    E::template step<R>(s, a, normalize, obs, *reward, d);
    *done = d ? 1 : 0;
    return 0;
}

// ---- env options (rl_env_cfg of the C ABI; doubles here so that the float64 leg sees them unrounded) -------------
struct OracleCfg {
    double ctrl_cost_coeff, alive_coeff, action_noise, obs_noise;
    int frame_skip, flags;
    double link_len;
};

template <class E, typename R>
rl::EnvCfgT<R> cfg_of(const OracleCfg* c) {
    rl::EnvCfgT<R> o = rl::default_cfg<E, R>();
    if (c) {
        o.ctrl_cost_coeff = (R)c->ctrl_cost_coeff; o.alive_coeff = (R)c->alive_coeff;
        o.action_noise = (R)c->action_noise; o.obs_noise = (R)c->obs_noise;
        if (c->frame_skip > 0) o.frame_skip = c->frame_skip;
        o.flags = c->flags;
        if (c->link_len > 0.0) o.link_len = (R)c->link_len;
    }
    return o;
}

template <class E>
int default_cfg_t(OracleCfg* c) {
    const rl::EnvCfgT<double> d = rl::default_cfg<E, double>();
    c->ctrl_cost_coeff = d.ctrl_cost_coeff; c->alive_coeff = d.alive_coeff; c->action_noise = 0.0; c->obs_noise = 0.0;
    c->frame_skip = d.frame_skip; c->flags = 0; c->link_len = d.link_len;
    return 0;
}

template <class E, typename R>
int reset_cfg_t(R* s, const R* draws, const OracleCfg* c) {
    const rl::EnvCfgT<R> k = cfg_of<E, R>(c);
    E::template reset<R>(s, draws, k.flags, k.link_len);
    return 0;
}

// Env.step under options: zact = the N(0,1) draws of the action noise (read only when action_noise != 0)
template <class E, typename R>
int step_cfg_t(R* s, const R* a, int normalize, const OracleCfg* c, const R* zact, R* obs, R* reward, int* done) {
    bool d;
    rl::step_cfg<E, R>(s, a, normalize, cfg_of<E, R>(c), zact, obs, *reward, d);
    *done = d ? 1 : 0;
    return 0;
}

template <class E, typename R>
int obs_noise_t(const OracleCfg* c, const R* z, R* obs) {
    const rl::EnvCfgT<R> k = cfg_of<E, R>(c);
    if (k.obs_noise != (R)0) rl::add_obs_noise<E, R>(k, z, obs);
    return 0;
}

template <class E, typename R>
int com_t(const R* s, R* c4) {
    if constexpr (E::HAS_COM) { E::template com<R>(s, c4); return 0; }
    else return -2;
}

template <class E>
int bounds_t(double* lb, double* ub) { E::template action_bounds<double>(lb, ub); return 0; }

template <class E>
int query_t(int* obs_dim, int* act_dim, int* state_dim, int* reset_draws, int* reset_is_normal) {
    *obs_dim = E::OBS; *act_dim = E::ACT; *state_dim = E::STATE;
    *reset_draws = E::RESET_DRAWS; *reset_is_normal = E::RESET_NORMAL ? 1 : 0;
    return 0;
}

// Serial replay of the lock-step VecEnvExecutor contract over n envs with the
// GPU's plane layout (state [S][n], actions [A][n], obs [O][n]) -- one env after
// the other, the way the reference's VecEnvExecutor.step loops
// (sandbox/rocky/tf/envs/vec_env_executor.py:16-28).
template <class E>
int vec_step_t(int n, int normalize, float scale_reward, int max_path_length, int auto_reset, float* state,
               int32_t* ts, const float* actions, const float* reset_draws, float* obs, float* reward,
               uint8_t* done, const OracleCfg* c = nullptr, const float* act_z = nullptr,
               const float* obs_z = nullptr) {
    const rl::EnvCfgT<float> cfg = cfg_of<E, float>(c);
    for (int i = 0; i < n; ++i) {
        float s[E::STATE], a[E::ACT], o[E::OBS], d[E::RESET_DRAWS], r, za[E::ACT], zo[E::OBS];
        for (int k = 0; k < E::STATE; ++k) s[k] = state[(size_t)k * n + i];
        for (int k = 0; k < E::ACT; ++k) a[k] = actions[(size_t)k * n + i];
        if (cfg.action_noise != 0.0f)
            for (int k = 0; k < E::ACT; ++k) za[k] = act_z[(size_t)k * n + i];
        bool dn;
        rl::step_cfg<E, float>(s, a, normalize, cfg, za, o, r, dn);
        int t = ts[i] + 1;
        if (max_path_length > 0 && t >= max_path_length) dn = true;
        if (dn && auto_reset) {
            for (int k = 0; k < E::RESET_DRAWS; ++k) d[k] = reset_draws[(size_t)k * n + i];
            E::template reset<float>(s, d, cfg.flags, cfg.link_len);
            E::template observe<float>(s, o);
            t = 0;
        }
        if (cfg.obs_noise != 0.0f) {
            for (int k = 0; k < E::OBS; ++k) zo[k] = obs_z[(size_t)k * n + i];
            rl::add_obs_noise<E, float>(cfg, zo, o);
        }
        for (int k = 0; k < E::STATE; ++k) state[(size_t)k * n + i] = s[k];
        ts[i] = t;
        for (int k = 0; k < E::OBS; ++k) obs[(size_t)k * n + i] = o[k];
        reward[i] = r * scale_reward;
        done[i] = dn ? 1 : 0;
    }
    return 0;
}

template <class E>
int vec_reset_t(int n, float* state, int32_t* ts, const uint8_t* mask, const float* draws, float* obs,
                const OracleCfg* c = nullptr, const float* obs_z = nullptr) {
    const rl::EnvCfgT<float> cfg = cfg_of<E, float>(c);
    for (int i = 0; i < n; ++i) {
        if (mask && !mask[i]) continue;
        float s[E::STATE], o[E::OBS], d[E::RESET_DRAWS], zo[E::OBS];
        for (int k = 0; k < E::STATE; ++k) s[k] = state[(size_t)k * n + i];
        for (int k = 0; k < E::RESET_DRAWS; ++k) d[k] = draws[(size_t)k * n + i];
        E::template reset<float>(s, d, cfg.flags, cfg.link_len);
        E::template observe<float>(s, o);
        if (cfg.obs_noise != 0.0f) {
            for (int k = 0; k < E::OBS; ++k) zo[k] = obs_z[(size_t)k * n + i];
            rl::add_obs_noise<E, float>(cfg, zo, o);
        }
        for (int k = 0; k < E::STATE; ++k) state[(size_t)k * n + i] = s[k];
        ts[i] = 0;
        for (int k = 0; k < E::OBS; ++k) obs[(size_t)k * n + i] = o[k];
    }
    return 0;
}

}  // namespace

// `FN` is a function template name, `...` its call arguments.
#define ORACLE_DISPATCH(kind, FN, ...)                         \
    switch (kind) {                                            \
        case 0: return FN<rl::Cartpole>(__VA_ARGS__);          \
        ORACLE_EXTRA_ENV_CASES(FN, __VA_ARGS__)                \
        default: return -1;                                    \
    }
#define ORACLE_DISPATCH_R(kind, FN, R, ...)                    \
    switch (kind) {                                            \
        case 0: return FN<rl::Cartpole, R>(__VA_ARGS__);       \
        ORACLE_EXTRA_ENV_CASES_R(FN, R, __VA_ARGS__)           \
        default: return -1;                                    \
    }

#ifndef ORACLE_EXTRA_ENV_CASES
#define ORACLE_EXTRA_ENV_CASES(FN, ...)
#define ORACLE_EXTRA_ENV_CASES_R(FN, R, ...)
#endif

extern "C" {

int oracle_env_query(int kind, int* obs_dim, int* act_dim, int* state_dim, int* reset_draws, int* reset_is_normal) {
    ORACLE_DISPATCH(kind, query_t, obs_dim, act_dim, state_dim, reset_draws, reset_is_normal)
}

// raw action bounds of the env (what the reference reads from env.action_space.bounds)
int oracle_env_action_bounds(int kind, double* lb, double* ub) { ORACLE_DISPATCH(kind, bounds_t, lb, ub) }

// ---- the same calls under env options (OracleCfg; null = the env's defaults) ------------------------------------
int oracle_env_default_cfg(int kind, OracleCfg* c) { ORACLE_DISPATCH(kind, default_cfg_t, c) }
int oracle_env_reset_cfg_f32(int kind, float* s, const float* draws, const OracleCfg* c) {
    ORACLE_DISPATCH_R(kind, reset_cfg_t, float, s, draws, c)
}
int oracle_env_reset_cfg_f64(int kind, double* s, const double* draws, const OracleCfg* c) {
    ORACLE_DISPATCH_R(kind, reset_cfg_t, double, s, draws, c)
}
int oracle_env_step_cfg_f32(int kind, float* s, const float* a, int normalize, const OracleCfg* c, const float* zact,
                            float* obs, float* reward, int* done) {
    ORACLE_DISPATCH_R(kind, step_cfg_t, float, s, a, normalize, c, zact, obs, reward, done)
}
int oracle_env_step_cfg_f64(int kind, double* s, const double* a, int normalize, const OracleCfg* c, const double* zact,
                            double* obs, double* reward, int* done) {
    ORACLE_DISPATCH_R(kind, step_cfg_t, double, s, a, normalize, c, zact, obs, reward, done)
}
int oracle_env_obs_noise_f32(int kind, const OracleCfg* c, const float* z, float* obs) {
    ORACLE_DISPATCH_R(kind, obs_noise_t, float, c, z, obs)
}
int oracle_env_obs_noise_f64(int kind, const OracleCfg* c, const double* z, double* obs) {
    ORACLE_DISPATCH_R(kind, obs_noise_t, double, c, z, obs)
}
int oracle_env_com_f32(int kind, const float* s, float* c4) { ORACLE_DISPATCH_R(kind, com_t, float, s, c4) }
int oracle_env_com_f64(int kind, const double* s, double* c4) { ORACLE_DISPATCH_R(kind, com_t, double, s, c4) }
int oracle_vecenv_step_cfg_f32(int kind, int n, int normalize, float scale_reward, int max_path_length,
                               int auto_reset, float* state, int32_t* ts, const float* actions,
                               const float* reset_draws, float* obs, float* reward, uint8_t* done,
                               const OracleCfg* c, const float* act_z, const float* obs_z) {
    ORACLE_DISPATCH(kind, vec_step_t, n, normalize, scale_reward, max_path_length, auto_reset, state, ts,
                    actions, reset_draws, obs, reward, done, c, act_z, obs_z)
}
int oracle_vecenv_reset_cfg_f32(int kind, int n, float* state, int32_t* ts, const uint8_t* mask,
                                const float* draws, float* obs, const OracleCfg* c, const float* obs_z) {
    ORACLE_DISPATCH(kind, vec_reset_t, n, state, ts, mask, draws, obs, c, obs_z)
}

// single env, array-of-struct state (state_dim contiguous values)
int oracle_env_reset_f32(int kind, float* s, const float* draws) { ORACLE_DISPATCH_R(kind, reset_t, float, s, draws) }
int oracle_env_reset_f64(int kind, double* s, const double* draws) { ORACLE_DISPATCH_R(kind, reset_t, double, s, draws) }
int oracle_env_observe_f32(int kind, const float* s, float* o) { ORACLE_DISPATCH_R(kind, observe_t, float, s, o) }
int oracle_env_observe_f64(int kind, const double* s, double* o) { ORACLE_DISPATCH_R(kind, observe_t, double, s, o) }
int oracle_env_step_f32(int kind, float* s, const float* a, int normalize, float* obs, float* reward, int* done) {
    ORACLE_DISPATCH_R(kind, step_t, float, s, a, normalize, obs, reward, done)
}
int oracle_env_step_f64(int kind, double* s, const double* a, int normalize, double* obs, double* reward, int* done) {
    ORACLE_DISPATCH_R(kind, step_t, double, s, a, normalize, obs, reward, done)
}

int oracle_vecenv_step_f32(int kind, int n, int normalize, float scale_reward, int max_path_length,
                           int auto_reset, float* state, int32_t* ts, const float* actions,
                           const float* reset_draws, float* obs, float* reward, uint8_t* done) {
    ORACLE_DISPATCH(kind, vec_step_t, n, normalize, scale_reward, max_path_length, auto_reset, state, ts,
                    actions, reset_draws, obs, reward, done)
}

int oracle_vecenv_reset_f32(int kind, int n, float* state, int32_t* ts, const uint8_t* mask,
                            const float* draws, float* obs) {
    ORACLE_DISPATCH(kind, vec_reset_t, n, state, ts, mask, draws, obs)
}

// Philox4x32-10 blocks for counters (c0 + i, c1, c2, c3): the integer stream the
// device RNG must reproduce exactly.
int oracle_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, int count,
                  uint32_t* out) {
    for (int i = 0; i < count; ++i) {
        rl::Philox4 p = rl::philox4x32_10(c0 + (uint32_t)i, c1, c2, c3, k0, k1);
        for (int k = 0; k < 4; ++k) out[4 * i + k] = p.v[k];
    }
    return 0;
}

void oracle_sincos_f32(int n, const float* x, float* s, float* c) {
    for (int i = 0; i < n; ++i) rl::rl_sincos(x[i], s[i], c[i]);
}


// ---- lock-step emulation of the four-lane ("quad") swimmer sub-step (rllab_amd/csrc/dyn_swimmer_chain.h) --------------
// The quad program is a straight-line per-lane function whose only cross-lane operation is x.qp<CTRL>(v).  The
// emulator runs it by replay: pass k executes every lane from the start, answers the first k exchange points from
// the log of earlier passes and records point k; after as many passes as there are exchange points every lane's
// result is exact.  Returns, for `nsub` sub-steps from the same (qpos, qvel, ctrl), the state of the scalar program
// (out_scalar) and of the emulated quad program (out_quad): they must be bit-identical.
}  // extern "C"

namespace {
template <typename R>
struct ReplayCtx {
    int lane, filled, counter;
    std::vector<std::array<R, 4>>* log;
    template <int CTRL> R qp(R v) {
        const int k = counter++;
        if (k < filled) return (*log)[k][(CTRL >> (2 * lane)) & 3];
        if ((int)log->size() <= k) log->resize(k + 1);
        if (k == filled) (*log)[k][lane] = v;
        return v;   // beyond the recorded prefix: placeholder, this pass's result is discarded
    }
};

template <typename R>
void swim_compare(const R* state, const R* ctrl2, int nsub, R* out_scalar, R* out_quad) {
    using Env = rl::Swimmer;
    using Chain = Env::Chain;
    const R h = (R)0.001;
    R ctrl[3] = {(R)0, ctrl2[0], ctrl2[1]};
    // scalar
    {
        R r4[4], th[3], om[3], sn[3], cs[3], q[5], qd[5];
        Env::to_chain(state, state + 5, r4, th, om, sn, cs);
        for (int it = 0; it < nsub; ++it) Chain::template substep_scalar<R>(r4, cs, sn, om, th, ctrl, h);
        Env::from_chain(r4, th, om, q, qd);
        for (int i = 0; i < 5; ++i) { out_scalar[i] = q[i]; out_scalar[5 + i] = qd[i]; }
        for (int b = 0; b < 3; ++b) { out_scalar[10 + b] = sn[b]; out_scalar[13 + b] = cs[b]; }
    }
    // quad (emulated)
    {
        R r4[4], th[3], om[3], sn[3], cs[3];
        Env::to_chain(state, state + 5, r4, th, om, sn, cs);
        Chain::Lane<R> lanes[4];
        Chain::LaneConst<R> consts[4];
        for (int b = 0; b < 4; ++b) {
            consts[b] = Chain::template lane_const<R>(b);
            lanes[b].set_direction(b < 3 ? cs[b] : (R)1, b < 3 ? sn[b] : (R)0);
            lanes[b].om = b < 3 ? om[b] : (R)0; lanes[b].th = b < 3 ? th[b] : (R)0;
            lanes[b].r = rl::V2<R>{r4[0], r4[1]}; lanes[b].v = rl::V2<R>{r4[2], r4[3]};
        }
        // the carried joint rate: own absolute rate - parent's (quad_perm PAR1: lane 0 reads the zero lane 3)
        for (int b = 0; b < 4; ++b) lanes[b].qd = lanes[b].om - lanes[(b + 3) & 3].om;
        for (int it = 0; it < nsub; ++it) {
            std::vector<std::array<R, 4>> log;
            int n_points = -1;
            Chain::Lane<R> result[4];
            for (int pass = 0; n_points < 0 || pass <= n_points; ++pass) {
                for (int b = 0; b < 4; ++b) {
                    ReplayCtx<R> x{b, pass, 0, &log};
                    Chain::Lane<R> s = lanes[b];
                    Chain::template substep_quad<R>(x, consts[b], s, b < 3 ? ctrl[b] : (R)0, h);
                    n_points = x.counter;
                    result[b] = s;
                }
            }
            for (int b = 0; b < 4; ++b) lanes[b] = result[b];
        }
        // back to (qpos, qvel) exactly as the rollout kernel does it
        out_quad[0] = lanes[0].r.x; out_quad[1] = lanes[0].r.y; out_quad[5] = lanes[0].v.x; out_quad[6] = lanes[0].v.y;
        for (int b = 0; b < 3; ++b) out_quad[2 + b] = lanes[b].th;
        for (int b = 0; b < 3; ++b) out_quad[7 + b] = lanes[b].qd;
        for (int b = 0; b < 3; ++b) { out_quad[10 + b] = lanes[b].A.y; out_quad[13 + b] = lanes[b].A.x; }
        // the replicated root translation must agree on every lane of the quad
        for (int b = 1; b < 4; ++b)
            if (!(lanes[b].r.x == lanes[0].r.x && lanes[b].r.y == lanes[0].r.y && lanes[b].v.x == lanes[0].v.x &&
                  lanes[b].v.y == lanes[0].v.y)) out_quad[0] = out_quad[0] * (R)0 + (R)1e30;   // poison: caught by the test
        // role 3 is the zero lane the bodies read "no parent" / "no child" from
        if (!(lanes[3].om == (R)0)) out_quad[0] = out_quad[0] * (R)0 + (R)1e30;
    }
}
}  // namespace

// ---- lock-step emulation of the one-body-per-lane instantiation of the two-legged sub-step (rllab_amd/csrc/dyn_two_legs.h) ---
// The same replay idea: the scalar-lane program's only cross-lane operations are the context's lane moves; pass k answers
// the first k exchange points from the log and records point k.  Returns, for `nsub` sub-steps from the same (q, qd, tau),
// the state of the eight-component instantiation (what HostEnv / the per-step kernels / the env-per-lane rollouts run)
// and of eight emulated scalar lanes (what rollout_two_leg_wave_kernel runs with V = float and DPP lane moves): they must
// be bit-identical on the roles where each value is defined, and the replicated root coordinates must agree on all lanes.
namespace {
template <typename R>
struct LaneReplayCtx {
    int lane, filled;                  // lane = 4 * leg + role
    mutable int counter;
    std::vector<std::array<R, 8>>* log;
    R move(R v, const int (&from)[8]) const {
        const int k = counter++;
        if (k < filled) return (*log)[k][from[lane]];
        if ((int)log->size() <= k) log->resize(k + 1);
        if (k == filled) (*log)[k][lane] = v;
        return v;   // beyond the recorded prefix: placeholder, this pass's result is discarded
    }
    R up(R v) const { static const int f[8] = {0, 0, 1, 2, 4, 4, 5, 6}; return move(v, f); }
    R down(R v) const { static const int f[8] = {1, 2, 3, 3, 5, 6, 7, 7}; return move(v, f); }
    R nxt(R v) const { static const int f[8] = {0, 2, 3, 1, 4, 6, 7, 5}; return move(v, f); }
    R prv(R v) const { static const int f[8] = {0, 3, 1, 2, 4, 7, 5, 6}; return move(v, f); }
    R root(R v) const { static const int f[8] = {0, 0, 0, 0, 4, 4, 4, 4}; return move(v, f); }
    R first(R v) const { static const int f[8] = {1, 1, 1, 1, 5, 5, 5, 5}; return move(v, f); }
    R other(R v) const { static const int f[8] = {4, 5, 6, 7, 0, 1, 2, 3}; return move(v, f); }
    R sel_root(R a, R b) const { return (lane & 3) == 0 ? a : b; }
    R sel_leaf(R a, R b) const { return (lane & 3) == 3 ? a : b; }
};

// the quad form: four lanes (the roles), every value a pair (leg 0, leg 1) -- what rollout_two_leg_quad_kernel runs
template <typename R>
struct QuadReplayCtx {
    using P = rl::V2<R>;
    int lane, filled;                  // lane = role
    mutable int counter;
    std::vector<std::array<P, 4>>* log;
    P move(P v, const int (&from)[4]) const {
        const int k = counter++;
        if (k < filled) return (*log)[k][from[lane]];
        if ((int)log->size() <= k) log->resize(k + 1);
        if (k == filled) (*log)[k][lane] = v;
        return v;
    }
    P up(P v) const { static const int f[4] = {0, 0, 1, 2}; return move(v, f); }
    P down(P v) const { static const int f[4] = {1, 2, 3, 3}; return move(v, f); }
    P nxt(P v) const { static const int f[4] = {0, 2, 3, 1}; return move(v, f); }
    P prv(P v) const { static const int f[4] = {0, 3, 1, 2}; return move(v, f); }
    P root(P v) const { static const int f[4] = {0, 0, 0, 0}; return move(v, f); }
    P first(P v) const { static const int f[4] = {1, 1, 1, 1}; return move(v, f); }
    P other(P v) const { return v.yx; }
    P sel_root(P a, P b) const { return lane == 0 ? a : b; }
    P sel_leaf(P a, P b) const { return lane == 3 ? a : b; }
};
template <typename R, class F>
void lock_step4(F&& f) {
    std::vector<std::array<rl::V2<R>, 4>> log;
    int n_points = -1;
    for (int pass = 0; n_points < 0 || pass <= n_points; ++pass)
        for (int l = 0; l < 4; ++l) {
            QuadReplayCtx<R> x{l, pass, 0, &log};
            f(x, l);
            n_points = x.counter;
        }
}
// state, centre of mass after nsub sub-steps of the quad form, laid out like two_leg_compare's outputs
template <class Env, typename R>
void two_leg_quad_form(const R* state, const R* tau, int nsub, R* out) {
    using Legs = typename Env::Legs;
    using P = rl::V2<R>;
    using QState = typename Legs::template State<P>;
    const R h = (R)0.0025;
    QState lanes[4], result[4];
    typename Legs::template LaneK<P> kc[4];
    P act[4];
    for (int r = 0; r < 4; ++r) {
        kc[r] = Legs::template role_constants<R>(r);
        const int j0 = r == 0 ? 2 : 2 + r, j1 = r == 0 ? 2 : 5 + r;
        lanes[r].q = P{state[j0], state[j1]}; lanes[r].w = P{state[9 + j0], state[9 + j1]};
        lanes[r].p1 = P{state[0], state[0]}; lanes[r].p2 = P{state[1], state[1]};
        lanes[r].v1 = P{state[9], state[9]}; lanes[r].v2 = P{state[10], state[10]};
        act[r] = r == 0 ? P{(R)0, (R)0} : P{tau[r], tau[3 + r]};
    }
    auto run = [&](auto&& body) {
        lock_step4<R>([&](const QuadReplayCtx<R>& x, int l) { QState s = lanes[l]; body(x, l, s); result[l] = s; });
        for (int l = 0; l < 4; ++l) lanes[l] = result[l];
    };
    run([&](const QuadReplayCtx<R>& x, int, QState& s) { Legs::template abs_rates<R, P, QuadReplayCtx<R>>(x, s); });
    run([&](const QuadReplayCtx<R>& x, int, QState& s) { Legs::template exact_directions<R, P, QuadReplayCtx<R>>(x, s); });
    for (int it = 0; it < nsub; ++it)
        run([&](const QuadReplayCtx<R>& x, int l, QState& s) { Legs::template substep<R, P, QuadReplayCtx<R>>(x, kc[l], s, act[l], h); });
    out[0] = lanes[0].p1.x; out[1] = lanes[0].p2.x; out[9] = lanes[0].v1.x; out[10] = lanes[0].v2.x;
    out[2] = lanes[0].q.x; out[11] = lanes[0].w.x;
    for (int r = 1; r < 4; ++r) {
        out[2 + r] = lanes[r].q.x; out[5 + r] = lanes[r].q.y;
        out[11 + r] = lanes[r].w.x; out[14 + r] = lanes[r].w.y;
    }
    run([&](const QuadReplayCtx<R>& x, int, QState& s) { Legs::template exact_directions<R, P, QuadReplayCtx<R>>(x, s); });
    P c[4][4];
    lock_step4<R>([&](const QuadReplayCtx<R>& x, int l) {
        Legs::template com<R, P, QuadReplayCtx<R>>(x, kc[l], lanes[l], c[l][0], c[l][1], c[l][2], c[l][3]);
    });
    for (int k = 0; k < 4; ++k) out[18 + k] = c[0][k].x;
    bool same = lanes[0].q.x == lanes[0].q.y && lanes[0].w.x == lanes[0].w.y;       // the torso, held by both legs
    for (int k = 0; k < 4; ++k) same = same && c[0][k].x == c[0][k].y;
    for (int r = 0; r < 4; ++r) same = same && lanes[r].p1.x == lanes[0].p1.y && lanes[r].v2.y == lanes[0].v2.x;
    if (!same) out[0] = out[0] * (R)0 + (R)1e30;
}

// run `f(ctx, lane)` on the eight lanes in lock step
template <typename R, class F>
void lock_step8(F&& f) {
    std::vector<std::array<R, 8>> log;
    int n_points = -1;
    for (int pass = 0; n_points < 0 || pass <= n_points; ++pass)
        for (int l = 0; l < 8; ++l) {
            LaneReplayCtx<R> x{l, pass, 0, &log};
            f(x, l);
            n_points = x.counter;
        }
}

template <class Env, typename R>
void two_leg_compare(const R* state, const R* tau, int nsub, R* out_packed, R* out_lanes) {
    using Legs = typename Env::Legs;
    using LState = typename Legs::template State<R>;
    const R h = (R)0.0025;
    // eight components: exactly what Env::step runs between step_begin and step_end
    {
        R q[9], qd[9];
        for (int i = 0; i < 9; ++i) { q[i] = state[i]; qd[i] = state[9 + i]; }
        Legs::template advance<R>(q, qd, tau, h, nsub);
        for (int i = 0; i < 9; ++i) { out_packed[i] = q[i]; out_packed[9 + i] = qd[i]; }
        R c[4];
        Legs::template com_of<R>(q, qd, c[0], c[1], c[2], c[3]);
        for (int k = 0; k < 4; ++k) out_packed[18 + k] = c[k];
    }
    // eight scalar lanes, as the one-env-per-wavefront kernel hands the env over and takes it back
    {
        LState lanes[8];
        typename Legs::template LaneK<R> kc[8];
        R act[8];
        for (int l = 0; l < 8; ++l) {
            const int leg = l >> 2, role = l & 3, j = role == 0 ? 2 : 2 + 3 * leg + role;
            kc[l] = Legs::template lane_constants<R>(leg, role);
            lanes[l].q = state[j]; lanes[l].w = state[9 + j];
            lanes[l].p1 = state[0]; lanes[l].p2 = state[1]; lanes[l].v1 = state[9]; lanes[l].v2 = state[10];
            act[l] = role == 0 ? (R)0 : tau[3 * leg + role];
        }
        LState result[8];
        auto directions = [&]() {
            lock_step8<R>([&](const LaneReplayCtx<R>& x, int l) {
                LState s = lanes[l];
                Legs::template exact_directions<R, R, LaneReplayCtx<R>>(x, s);
                result[l] = s;
            });
            for (int l = 0; l < 8; ++l) lanes[l] = result[l];
        };
        lock_step8<R>([&](const LaneReplayCtx<R>& x, int l) {
            LState s = lanes[l];
            Legs::template abs_rates<R, R, LaneReplayCtx<R>>(x, s);
            result[l] = s;
        });
        for (int l = 0; l < 8; ++l) lanes[l] = result[l];
        directions();
        for (int it = 0; it < nsub; ++it) {
            lock_step8<R>([&](const LaneReplayCtx<R>& x, int l) {
                LState s = lanes[l];
                Legs::template substep<R, R, LaneReplayCtx<R>>(x, kc[l], s, act[l], h);
                result[l] = s;
            });
            for (int l = 0; l < 8; ++l) lanes[l] = result[l];
        }
        out_lanes[0] = lanes[0].p1; out_lanes[1] = lanes[0].p2; out_lanes[9] = lanes[0].v1; out_lanes[10] = lanes[0].v2;
        out_lanes[2] = lanes[0].q; out_lanes[11] = lanes[0].w;
        for (int l = 1; l < 8; ++l) {
            if ((l & 3) == 0) continue;
            const int j = 2 + 3 * (l >> 2) + (l & 3);
            out_lanes[j] = lanes[l].q; out_lanes[9 + j] = lanes[l].w;
        }
        // centre of mass as the kernel forms it: exact sines of the new angles, then Legs::com, read on role 0
        directions();
        R c[8][4];
        lock_step8<R>([&](const LaneReplayCtx<R>& x, int l) {
            Legs::template com<R, R, LaneReplayCtx<R>>(x, kc[l], lanes[l], c[l][0], c[l][1], c[l][2], c[l][3]);
        });
        for (int k = 0; k < 4; ++k) out_lanes[18 + k] = c[0][k];
        // the replicated root translation must agree on every lane, the torso and the centre of mass between the two role-0 lanes
        bool same = true;
        for (int l = 1; l < 8; ++l)
            same = same && lanes[l].p1 == lanes[0].p1 && lanes[l].p2 == lanes[0].p2 && lanes[l].v1 == lanes[0].v1 &&
                   lanes[l].v2 == lanes[0].v2;
        same = same && lanes[4].q == lanes[0].q && lanes[4].w == lanes[0].w && lanes[4].sn == lanes[0].sn && lanes[4].cs == lanes[0].cs;
        for (int k = 0; k < 4; ++k) same = same && c[0][k] == c[4][k];
        if (!same) out_lanes[0] = out_lanes[0] * (R)0 + (R)1e30;   // poison: caught by the test
    }
}
}  // namespace

// the per-lane constant table of the two-leg program (dyn_two_legs.h lane_constants): out[8][20] =
// jx, jy, cx, cy, mass, inertia, arm, stiff, damp, lo, hi, mc, cpx[2], cpy[2], crad[2], cmu[2] for lane 4 * leg + role
template <class Env>
static void two_leg_lane_table(double* out) {
    using Legs = typename Env::Legs;
    for (int l = 0; l < 8; ++l) {
        const typename Legs::template LaneK<double> k = Legs::template lane_constants<double>(l >> 2, l & 3);
        double* o = out + 20 * l;
        o[0] = k.jx; o[1] = k.jy; o[2] = k.cx; o[3] = k.cy; o[4] = k.mass; o[5] = k.inertia; o[6] = k.arm; o[7] = k.stiff;
        o[8] = k.damp; o[9] = k.lo; o[10] = k.hi; o[11] = k.mc;
        for (int s2 = 0; s2 < 2; ++s2) { o[12 + s2] = k.cpx[s2]; o[14 + s2] = k.cpy[s2]; o[16 + s2] = k.crad[s2]; o[18 + s2] = k.cmu[s2]; }
    }
}
extern "C" {
// kind: 3 = HalfCheetah, 5 = Walker2D.  state[18] = (q, qd), tau[7] (tau[0] unused).  out_*[22] = q, qd, centre of mass (4)
int oracle_two_leg_compare_f32(int kind, const float* state, const float* tau, int nsub, float* out_packed, float* out_lanes) {
    if (kind == 3) two_leg_compare<rl::HalfCheetah, float>(state, tau, nsub, out_packed, out_lanes);
    else if (kind == 5) two_leg_compare<rl::Walker2D, float>(state, tau, nsub, out_packed, out_lanes);
    else return -1;
    return 0;
}
int oracle_two_leg_compare_f64(int kind, const double* state, const double* tau, int nsub, double* out_packed, double* out_lanes) {
    if (kind == 3) two_leg_compare<rl::HalfCheetah, double>(state, tau, nsub, out_packed, out_lanes);
    else if (kind == 5) two_leg_compare<rl::Walker2D, double>(state, tau, nsub, out_packed, out_lanes);
    else return -1;
    return 0;
}
// dyn_two_legs.h's per-lane constant table: out[8][20] (see two_leg_lane_table above)
int oracle_two_leg_lane_table(int kind, double* out) {
    if (kind == 3) two_leg_lane_table<rl::HalfCheetah>(out);
    else if (kind == 5) two_leg_lane_table<rl::Walker2D>(out);
    else return -1;
    return 0;
}
// the same sub-steps in the quad form (four role lanes, both legs side by side in every value): out[22]
int oracle_two_leg_quad_form_f32(int kind, const float* state, const float* tau, int nsub, float* out) {
    if (kind == 3) two_leg_quad_form<rl::HalfCheetah, float>(state, tau, nsub, out);
    else if (kind == 5) two_leg_quad_form<rl::Walker2D, float>(state, tau, nsub, out);
    else return -1;
    return 0;
}
int oracle_two_leg_quad_form_f64(int kind, const double* state, const double* tau, int nsub, double* out) {
    if (kind == 3) two_leg_quad_form<rl::HalfCheetah, double>(state, tau, nsub, out);
    else if (kind == 5) two_leg_quad_form<rl::Walker2D, double>(state, tau, nsub, out);
    else return -1;
    return 0;
}
int oracle_swim_quad_compare_f32(const float* state, const float* ctrl2, int nsub, float* out_scalar, float* out_quad) {
    swim_compare<float>(state, ctrl2, nsub, out_scalar, out_quad);
    return 0;
}
int oracle_swim_quad_compare_f64(const double* state, const double* ctrl2, int nsub, double* out_scalar, double* out_quad) {
    swim_compare<double>(state, ctrl2, nsub, out_scalar, out_quad);
    return 0;
}
}  // extern "C"
