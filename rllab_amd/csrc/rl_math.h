// rl_math.h -- arithmetic primitives shared by every env dynamics header.
//
// Everything in the env step path must produce the same bits when the header is
// compiled by hipcc for gfx950 and by the same clang for the x86 host oracle build
// (oracle/env_host.cpp).  That rules out libm / ocml transcendentals (their
// last-ulp behaviour differs) and anything order-dependent.  Both builds use
// -ffp-contract=on: the clang FRONTEND fuses a*b+c inside one expression into
// llvm.fmuladd at the same source sites for both targets, and both back-ends lower
// it to a hardware FMA (v_fma_f32 / vfmadd, one rounding), so contraction is
// deterministic across the two builds -- unlike -ffp-contract=fast, where each
// back-end picks its own fusion sites.  What is left is IEEE-754 +,-,*,/ and sqrt
// (correctly rounded on both targets), fused multiply-add, float<->int conversion.
// The GPU parity tests replay >1e5 env-steps bit for bit to hold this in place.
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define RL_HD __host__ __device__ __forceinline__
#define RL_UNROLL _Pragma("unroll")
#else
#define RL_HD inline __attribute__((always_inline))
#define RL_UNROLL
#endif

namespace rl {

// explicit fused multiply-add: v_fma_f32 on gfx950, vfmadd on the host (-mfma).
RL_HD float rl_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
RL_HD double rl_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

RL_HD float rl_sqrt(float x) { return __builtin_sqrtf(x); }
RL_HD double rl_sqrt(double x) { return __builtin_sqrt(x); }
RL_HD float rl_abs(float x) { return __builtin_fabsf(x); }
RL_HD double rl_abs(double x) { return __builtin_fabs(x); }

template <typename R> RL_HD R rl_min(R a, R b) { return a < b ? a : b; }
template <typename R> RL_HD R rl_max(R a, R b) { return a > b ? a : b; }
template <typename R> RL_HD R rl_clamp(R x, R lo, R hi) { return rl_max(lo, rl_min(x, hi)); }
// clamp of a finite x to [lo, hi] with lo < hi: one v_med3_f32 on the device; returns one of its arguments, so the
// host form is bit-identical
RL_HD float rl_clamp_finite(float x, float lo, float hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fmed3f(x, lo, hi);
#else
    return x < lo ? lo : (x > hi ? hi : x);
#endif
}
// 1 / d, correctly rounded, for a normal d far from the exponent limits.  Device: v_rcp_f32 (within 1 ulp) + ONE
// Newton step in fused arithmetic -- tools/ubench/rcp_exactness.hip checks on the hardware that this equals RN(1 / d)
// for all 2^23 mantissas (the seed is wrong by an ulp for 10.7 % of them, the refined value for none), so the host
// (which divides) and the device agree bit for bit.  Three issue slots + the transcendental's second pass, against
// eight for a correctly rounded quotient: the dynamics multiply by this reciprocal instead of dividing.
RL_HD float rl_recip_normal(float d) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e0 = __builtin_fmaf(-d, r0, 1.0f);
    return __builtin_fmaf(e0, r0, r0);
#else
    return 1.0f / d;
#endif
}
RL_HD double rl_recip_normal(double d) { return 1.0 / d; }
RL_HD double rl_clamp_finite(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

// Deterministic single-precision sin/cos.  Cody-Waite three-term reduction by
// pi/4 followed by degree-7 / degree-8 minimax polynomials on [-pi/4, pi/4]
// (the classic single-precision kernel; max error < 1.5 ulp for |x| < 8192,
// which covers every joint angle these envs can reach).  Only mul/add/sub and
// float->int truncation, so host and device agree bit for bit.
RL_HD void rl_sincos(float x, float& s, float& c) {
    const float FOPI = 1.27323954473516f;  // 4/pi
    const float DP1 = 0.78515625f;
    const float DP2 = 2.4187564849853515625e-4f;
    const float DP3 = 3.77489497744594108e-8f;
    float ax = rl_abs(x);
    int j = (int)(ax * FOPI);
    j = (j + 1) & ~1;  // round up to even octant boundary
    float y = (float)j;
    float r = ((ax - y * DP1) - y * DP2) - y * DP3;
    float z = r * r;
    float ps = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float pc = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z
               - 0.5f * z + 1.0f;
    int q = (j >> 1) & 3;  // quadrant of the reduced argument
    float sv = (q & 1) ? pc : ps;
    float cv = (q & 1) ? ps : pc;
    if (q & 2) sv = -sv;
    if ((q == 1) || (q == 2)) cv = -cv;
    s = (x < 0.0f) ? -sv : sv;
    c = cv;
}

// (sn, cs) <- rotation of (sn, cs) by the small angle d (|d| < ~0.2): Taylor sin / cos of d,
// truncated below the rounding error of the type (float: d^7 / 5040 < 3e-9 * d, double: d^13).
// Used by the articulated-body sub-step to carry sin / cos of the body angles.
template <typename R>
RL_HD void rl_rotate_small(R& sn, R& cs, R d) {
    const R d2 = d * d;
    R sd, cd;
    if constexpr (sizeof(R) == 4) {
        sd = d * ((R)1 + d2 * ((R)(-1.0 / 6) + d2 * (R)(1.0 / 120)));
        cd = (R)1 + d2 * ((R)-0.5 + d2 * ((R)(1.0 / 24) + d2 * (R)(-1.0 / 720)));
    } else {
        sd = d * ((R)1 + d2 * ((R)(-1.0 / 6) + d2 * ((R)(1.0 / 120) + d2 * ((R)(-1.0 / 5040) +
             d2 * ((R)(1.0 / 362880) + d2 * (R)(-1.0 / 39916800))))));
        cd = (R)1 + d2 * ((R)-0.5 + d2 * ((R)(1.0 / 24) + d2 * ((R)(-1.0 / 720) + d2 * ((R)(1.0 / 40320) +
             d2 * ((R)(-1.0 / 3628800) + d2 * (R)(1.0 / 479001600))))));
    }
    const R s = sn * cd + cs * sd;
    const R c = cs * cd - sn * sd;
    sn = s;
    cs = c;
}

// Same rotation for |d| << 1 (one 1 ms sub-step of a swimmer link: |d| = 1e-3 |omega| < 0.02): in single precision the
// series are cut where the next term is below half an ulp of the result (sin: d^5/120 < 3e-11, cos: d^4/24 < 7e-9; the
// carried pair is re-seeded with exact values every env step, so nothing accumulates beyond 50 sub-steps).
template <typename R>
RL_HD void rl_tiny_sincos(R d, R& sd, R& cd) {
    const R d2 = d * d;
    if constexpr (sizeof(R) == 4) {
        const R t = d2 * (R)(-1.0 / 6);
        sd = d + d * t;
        cd = (R)1 + d2 * (R)-0.5;
    } else {
        sd = d * ((R)1 + d2 * ((R)(-1.0 / 6) + d2 * ((R)(1.0 / 120) + d2 * ((R)(-1.0 / 5040) +
             d2 * ((R)(1.0 / 362880) + d2 * (R)(-1.0 / 39916800))))));
        cd = (R)1 + d2 * ((R)-0.5 + d2 * ((R)(1.0 / 24) + d2 * ((R)(-1.0 / 720) + d2 * ((R)(1.0 / 40320) +
             d2 * ((R)(-1.0 / 3628800) + d2 * (R)(1.0 / 479001600))))));
    }
}
template <typename R>
RL_HD void rl_rotate_tiny(R& sn, R& cs, R d) {
    R sd, cd;
    rl_tiny_sincos(d, sd, cd);
    const R s = sn * cd + cs * sd;
    const R c = cs * cd - sn * sd;
    sn = s;
    cs = c;
}

// Two-component vector for the x / y pairs of the planar dynamics: elementwise +, -, *, scalar broadcast and
// .xy swizzles (clang ext_vector_type, the same front-end for both builds; -ffp-contract=on fuses a * b + c per
// component exactly as for scalars).  On gfx950 one v_pk_{mul,add,fma}_f32 per operation, swizzles in op_sel.
template <typename R> using V2 = R __attribute__((ext_vector_type(2)));

// The double instantiation is host-only (independent physics checks at 1e-10);
// it may use libm.
RL_HD void rl_sincos(double x, double& s, double& c) {
    s = sin(x);
    c = cos(x);
}

// ---------------------------------------------------------------------------
// Philox4x32-10 counter-based generator (Salmon et al., SC'11).  Pure 32/64-bit
// integer arithmetic: the stream is identical on host and device and is pinned
// against a numpy restatement in tests/.  Key = (seed_lo, seed_hi); counter =
// (env index, step, episode/iteration counter, purpose tag).
// ---------------------------------------------------------------------------
struct Philox4 {
    uint32_t v[4];
};

RL_HD Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    const uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int i = 0; i < 10; ++i) {
        uint64_t p0 = (uint64_t)M0 * c0;
        uint64_t p1 = (uint64_t)M1 * c2;
        uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t n0 = hi1 ^ c1 ^ k0;
        uint32_t n1 = lo1;
        uint32_t n2 = hi0 ^ c3 ^ k1;
        uint32_t n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    Philox4 r;
    r.v[0] = c0; r.v[1] = c1; r.v[2] = c2; r.v[3] = c3;
    return r;
}

// 24-bit uniform in [0, 1): exact in float, identical on host and device.
RL_HD float u32_to_unit(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }
// (0, 1] variant for Box-Muller's log.
RL_HD float u32_to_unit_open(uint32_t x) { return ((float)(x >> 8) + 1.0f) * (1.0f / 16777216.0f); }

enum RngPurpose : uint32_t {
    RNG_RESET = 0x52455345u,   // reset draws
    RNG_POLICY = 0x504f4c49u,  // policy action noise
    RNG_ACT_NOISE = 0x414e4f49u,  // env action noise (Box2DEnv / MujocoEnv action_noise)
    RNG_OBS_NOISE = 0x4f4e4f49u,  // env observation noise (Box2DEnv obs_noise)
};

// ---- env options ------------------------------------------------------------------------------------------------
// What the reference's env constructors take and the kernels honour at run time (include/rllab_amd.h: rl_env_cfg):
//   SwimmerEnv / Walker2DEnv / HopperEnv(ctrl_cost_coeff=..)   swimmer_env.py:15-21, walker2d_env.py:21-27, hopper_env.py:27-35
//   HopperEnv(alive_coeff=..)
//   Box2DEnv(frame_skip=.., obs_noise=.., action_noise=..)     box2d_env.py:30-58,194-230
//   MujocoEnv(action_noise=..)                                  mujoco_env.py:40-43,175-185
// EnvCfg travels by value into every env kernel; StepOpts is what one Env::step sees of it (plus the additive action
// perturbation the noise amounts to for this transition).  Defaults come from each env (Env::default_opts), so a
// launch without options runs the same arithmetic, bit for bit, as before the options existed.
template <typename R>
struct EnvCfgT {
    R ctrl_cost_coeff, alive_coeff, action_noise, obs_noise;
    int frame_skip, flags;
    R link_len;                  // DoublePendulumEnv: length of both links (1 unless template_args noise drew another)
};
using EnvCfg = EnvCfgT<float>;
enum EnvCfgFlags : int {
    CFG_POLE_FOLLOWS_CART = 1,   // CartpoleEnv.reset moves the pole with the cart (hinge starts closed)
    CFG_FIXED_START = 2,         // InvertedDoublePendulumEnv(random_start=False)
    CFG_LIMIT_MUJOCO = 4,        // SwimmerEnv(limit_model="mujoco"): joint limits by MuJoCo's documented soft-constraint
                                 // model from the MJCF's own solreflimit / solimplimit (dyn_swimmer_chain.h) instead
                                 // of the penalty spring-damper; HalfCheetahEnv / Walker2DEnv / HopperEnv(limit_model=
                                 // "mujoco"): the hinges' limits as rows of dyn_mjc.h's constraint solve
    CFG_CONTACT_MUJOCO = 8,      // HalfCheetahEnv / Walker2DEnv / HopperEnv(contact_model="mujoco"): the capsule end spheres'
                                 // floor contacts as pyramidal-cone rows of dyn_mjc.h's constraint solve (penalty otherwise)
};
template <typename R>
struct StepOpts {
    R ctrl_cost_coeff, alive_coeff;
    int frame_skip;
    const R* dact;               // per-dimension additive perturbation of the applied action, or null
    R link_len;
    int flags;                   // EnvCfgFlags
};
template <typename R>
RL_HD StepOpts<R> make_opts(double ctrl_cost_coeff, double alive_coeff, int frame_skip) {
    StepOpts<R> o;
    o.ctrl_cost_coeff = (R)ctrl_cost_coeff; o.alive_coeff = (R)alive_coeff; o.frame_skip = frame_skip; o.dact = nullptr;
    o.link_len = (R)1;
    o.flags = 0;
    return o;
}
// the options an env runs with when the caller gives none
template <class Env, typename R>
RL_HD EnvCfgT<R> default_cfg() {
    const StepOpts<R> o = Env::template default_opts<R>();
    EnvCfgT<R> c;
    c.ctrl_cost_coeff = o.ctrl_cost_coeff; c.alive_coeff = o.alive_coeff; c.action_noise = (R)0; c.obs_noise = (R)0;
    c.frame_skip = o.frame_skip; c.flags = 0; c.link_len = (R)1;
    return c;
}
template <typename R>
RL_HD StepOpts<R> opts_from_cfg(const EnvCfgT<R>& c) {
    StepOpts<R> o;
    o.ctrl_cost_coeff = c.ctrl_cost_coeff; o.alive_coeff = c.alive_coeff; o.frame_skip = c.frame_skip; o.dact = nullptr;
    o.link_len = c.link_len;
    o.flags = c.flags;
    return o;
}
// Box2DEnv._inject_action_noise / MujocoEnv.inject_action_noise (box2d_env.py:219-226, mujoco_env.py:175-182):
//   noise = action_noise * N(0,1);  noise = 0.5 * (ub - lb) * noise;  applied = action + noise.   z = the N(0,1) draws
template <class Env, typename R>
RL_HD void action_perturbation(const EnvCfgT<R>& c, const R* z, R* dact) {
    R lb[Env::ACT], ub[Env::ACT];
    Env::template action_bounds<R>(lb, ub);
    RL_UNROLL
    for (int k = 0; k < Env::ACT; ++k) {
        const R noise = c.action_noise * z[k];
        dact[k] = (R)0.5 * (ub[k] - lb[k]) * noise;
    }
}
// Box2DEnv._inject_obs_noise (box2d_env.py:194-201): obs + 1 * obs_noise * N(0,1), entry-wise
template <class Env, typename R>
RL_HD void add_obs_noise(const EnvCfgT<R>& c, const R* z, R* obs) {
    RL_UNROLL
    for (int k = 0; k < Env::OBS; ++k) {
        const R noise = c.obs_noise * z[k];
        obs[k] = obs[k] + noise;
    }
}
// Env.step under a cfg: options + (if action_noise != 0) the perturbation from the draws `zact`
template <class Env, typename R>
RL_HD void step_cfg(R* s, const R* a, int normalize, const EnvCfgT<R>& c, const R* zact, R* obs, R& reward, bool& done) {
    StepOpts<R> o = opts_from_cfg<R>(c);
    R dact[Env::ACT];
    if (c.action_noise != (R)0) {
        action_perturbation<Env, R>(c, zact, dact);
        o.dact = dact;
    }
    Env::template step<R>(s, a, normalize, obs, reward, done, o);
}

}  // namespace rl
