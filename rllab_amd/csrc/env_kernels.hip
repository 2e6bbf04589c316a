// env_kernels.hip -- lock-step vectorised env kernels and the fused rollout.
//
// One thread owns one env copy for the whole launch: state lives in VGPRs, HBM is
// touched only for the SoA state planes (coalesced: lane i <-> env i) and for the
// trajectory planes.  Envs never talk to each other, so there is no LDS traffic
// besides the policy weights and no inter-workgroup synchronisation at all.
// Build with -ffp-contract=on (front-end contraction): the env arithmetic must match
// the host oracle build bit for bit (rl_math.h); the policy MLP uses explicit FMAs.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <type_traits>
#include "../../include/rllab_amd.h"
#include "capi_util.h"
#include "device_rng.h"
#include "envs.h"
#include "policy_mfma.h"
#include "policy_wide.h"

namespace rl {

constexpr int BLOCK = 64;  // one wavefront per workgroup: 4096 envs -> 64 workgroups on 64 CUs

template <class Env>
__device__ __forceinline__ void load_state(const float* __restrict__ state, int n, int i, float* s) {
#pragma unroll
    for (int k = 0; k < Env::STATE; ++k) s[k] = state[(size_t)k * n + i];
}
template <class Env>
__device__ __forceinline__ void store_state(float* __restrict__ state, int n, int i, const float* s) {
#pragma unroll
    for (int k = 0; k < Env::STATE; ++k) state[(size_t)k * n + i] = s[k];
}

// Values just loaded from global memory inside a wave-uniform branch are made to ARRIVE inside that branch.  The memory
// counter (vmcnt) counts loads and stores alike and the compiler places the wait at the first use: behind the join that
// is a conservative vmcnt(0) on EVERY path -- in the rollout loops it stood behind the step's trajectory stores and
// exposed their full round trip every env-step (injected noise / reset planes are the parity runs' path; a training
// run takes the other side of these branches and must not wait at all).
template <int COUNT>
__device__ __forceinline__ void landed(float* v) {
#pragma unroll
    for (int k = 0; k < COUNT; ++k) asm volatile("" : "+v"(v[k]));
}

template <class Env>
__device__ __forceinline__ void reset_one(float* s, const float* __restrict__ draws, int n, int i,
                                          uint64_t seed, uint32_t env_global, uint64_t step, const EnvCfg& cfg) {
    float d[Env::RESET_DRAWS];
    if (draws) {
#pragma unroll
        for (int k = 0; k < Env::RESET_DRAWS; ++k) d[k] = draws[(size_t)k * n + i];
        landed<Env::RESET_DRAWS>(d);
    } else {
        philox_draws<Env::RESET_DRAWS, Env::RESET_NORMAL>(d, seed, env_global, step, RNG_RESET);
    }
    Env::template reset<float>(s, d, cfg.flags, cfg.link_len);
}

// N(0,1) draws of one env for one transition: slice `z` (a [COUNT][n] plane set injected by the caller -- parity
// runs) or the Philox stream under `purpose`
template <int COUNT>
__device__ __forceinline__ void noise_draws(float* d, const float* __restrict__ z, int n, int i, uint64_t seed,
                                            uint32_t env_global, uint64_t step, uint32_t purpose) {
    if (z) {
#pragma unroll
        for (int k = 0; k < COUNT; ++k) d[k] = z[(size_t)k * n + i];
        landed<COUNT>(d);
    } else {
        philox_draws<COUNT, true>(d, seed, env_global, step, purpose);
    }
}
// the observation a caller sees: raw observation + obs_noise * N(0,1) (Box2DEnv.get_current_obs, box2d_env.py:210-218).
// Wave-uniform branch: a launch without obs noise pays one scalar compare.
template <class Env>
__device__ __forceinline__ void observed(float* o, const EnvCfg& cfg, const float* __restrict__ z, int n, int i,
                                         uint64_t seed, uint32_t env_global, uint64_t step) {
    if (cfg.obs_noise != 0.0f) {
        float zn[Env::OBS];
        noise_draws<Env::OBS>(zn, z, n, i, seed, env_global, step, RNG_OBS_NOISE);
        add_obs_noise<Env, float>(cfg, zn, o);
    }
}
// Env.step with the launch's options; draws its action-noise variates only when the option is on
template <class Env>
__device__ __forceinline__ void step_one(float* s, const float* a, int normalize, const EnvCfg& cfg,
                                         const float* __restrict__ z, int n, int i, uint64_t seed, uint32_t env_global,
                                         uint64_t step, float* o, float& r, bool& d) {
    float zn[Env::ACT];
    if (cfg.action_noise != 0.0f) noise_draws<Env::ACT>(zn, z, n, i, seed, env_global, step, RNG_ACT_NOISE);
    step_cfg<Env, float>(s, a, normalize, cfg, zn, o, r, d);
}

// ---------------------------------------------------------------------------
// VecEnvExecutor.reset / masked Env.reset
// ---------------------------------------------------------------------------
template <class Env>
__global__ void __launch_bounds__(BLOCK)
vecenv_reset_kernel(int n, float* __restrict__ state, int32_t* __restrict__ ts,
                    const uint8_t* __restrict__ mask, const float* __restrict__ draws, uint64_t seed,
                    uint64_t step, int env_offset, EnvCfg cfg, const float* __restrict__ obs_z,
                    float* __restrict__ obs) {
    int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    if (mask && !mask[i]) return;
    float s[Env::STATE];
    load_state<Env>(state, n, i, s);  // persisted solver state survives reset
    reset_one<Env>(s, draws, n, i, seed, (uint32_t)(env_offset + i), step, cfg);
    store_state<Env>(state, n, i, s);
    ts[i] = 0;
    float o[Env::OBS];
    Env::template observe<float>(s, o);
    observed<Env>(o, cfg, obs_z, n, i, seed, (uint32_t)(env_offset + i), step);
#pragma unroll
    for (int k = 0; k < Env::OBS; ++k) obs[(size_t)k * n + i] = o[k];
}

// get_current_obs (box2d_env.py:210-218, mujoco_env.py:118-131): the observation of the state as it is, no
// transition -- after a set_state, or to look at an env between steps
template <class Env>
__global__ void __launch_bounds__(BLOCK)
vecenv_observe_kernel(int n, const float* __restrict__ state, float* __restrict__ obs) {
    int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    float s[Env::STATE];
    load_state<Env>(state, n, i, s);
    float o[Env::OBS];
    Env::template observe<float>(s, o);
#pragma unroll
    for (int k = 0; k < Env::OBS; ++k) obs[(size_t)k * n + i] = o[k];
}

// MujocoEnv.get_body_com / get_body_comvel of the torso subtree (mujoco_env.py:232-238, mjcore.py:58-81):
// com4 [4][n] = position (2) and velocity (2) of the subtree centre of mass in the env's (forward, up) axes
template <class Env>
__global__ void __launch_bounds__(BLOCK)
vecenv_com_kernel(int n, const float* __restrict__ state, float* __restrict__ com4) {
    int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    float s[Env::STATE];
    load_state<Env>(state, n, i, s);
    float c[4];
    Env::template com<float>(s, c);
#pragma unroll
    for (int k = 0; k < 4; ++k) com4[(size_t)k * n + i] = c[k];
}

// ---------------------------------------------------------------------------
// VecEnvExecutor.step
// ---------------------------------------------------------------------------
// (two wavefronts per SIMD asked for: the one-thread two-leg programs sit at 255 + 2 registers, one above the budget that
// lets a second wavefront hide the first one's waits -- STEP_WPS, round 6)
#ifndef RL_STEP_WPS
#define RL_STEP_WPS 2
#endif
template <class Env>
__global__ void __launch_bounds__(BLOCK, RL_STEP_WPS)
vecenv_step_kernel(int n, int normalize, float scale_reward, int max_path_length, int auto_reset,
                   float* __restrict__ state, int32_t* __restrict__ ts,
                   const float* __restrict__ actions, const float* __restrict__ reset_draws,
                   uint64_t seed, uint64_t step_value, const uint64_t* __restrict__ step_dev, int env_offset,
                   EnvCfg cfg, const float* __restrict__ act_z, const float* __restrict__ obs_z,
                   float* __restrict__ obs, float* __restrict__ reward, uint8_t* __restrict__ done) {
    int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    // the RNG counter of this transition: an argument, or -- for launches replayed from a hipGraph, whose arguments
    // are frozen at capture -- a device word that rl_counter_add advances between replays
    const uint64_t step = step_dev ? *step_dev : step_value;
    float s[Env::STATE];
    load_state<Env>(state, n, i, s);
    float a[Env::ACT];
#pragma unroll
    for (int k = 0; k < Env::ACT; ++k) a[k] = actions[(size_t)k * n + i];
    float o[Env::OBS];
    float r;
    bool d;
    const uint32_t env_global = (uint32_t)(env_offset + i);
    step_one<Env>(s, a, normalize, cfg, act_z, n, i, seed, env_global, step, o, r, d);
    int t = ts[i] + 1;
    if (max_path_length > 0 && t >= max_path_length) d = true;
    if (d && auto_reset) {
        reset_one<Env>(s, reset_draws, n, i, seed, env_global, step, cfg);
        Env::template observe<float>(s, o);
        t = 0;
    }
    observed<Env>(o, cfg, obs_z, n, i, seed, env_global, step);
    store_state<Env>(state, n, i, s);
    ts[i] = t;
#pragma unroll
    for (int k = 0; k < Env::OBS; ++k) obs[(size_t)k * n + i] = o[k];
    reward[i] = r * scale_reward;
    done[i] = d ? 1 : 0;
}

// ---------------------------------------------------------------------------
// Fused rollout: GaussianMLPPolicy.get_actions + env.step + record + auto-reset,
// T times, one launch.
// ---------------------------------------------------------------------------
// mean = Wout^T tanh(W1^T tanh(W0^T o + b0) + b1) + bout   (network.py:36-101)
//
// The policy GEMMs of the 64 envs of a wavefront run on the matrix cores (policy_mfma.h): the
// observations are handed over through a [input][lane] LDS tile, two blocks of 32 envs go
// through v_mfma_f32_32x32x2_f32 chains (layer 0: (DO+2)/2 steps, layer 1: 16*H/32 steps per
// 32-unit tile) with the activations staying in the output-fragment registers, the thin output
// layer (DA columns) is a per-lane dot product folded across the two lane halves, and lane l
// picks the mean of its own env from block l/32.  One env-step of policy costs 2*HT*(KS0+KS1)
// matrix instructions instead of ~(DO*H + H*H + H*DA) VALU FMAs + LDS weight reads per lane --
// with a single wavefront per SIMD the rollout is bound by instruction issue, so the policy's
// share of the step shrinks by the ratio of those counts.
template <class Env, int H, bool ACTS = false>
struct RolloutPolicy {
    using N = Net<Env::OBS, Env::ACT, H>;
    static constexpr int XROWS = 2 * N::KS0;                       // inputs + bias row + zero padding
    static constexpr int LDS_FLOATS = N::FA0 + N::FA1 + N::TAILP + XROWS * WV;
    static constexpr int T_B1 = 0, T_W2 = N::W2 - N::B1, T_B2 = N::B2 - N::B1, T_LS = N::LSTD - N::B1;

    float* fa0;
    float* fa1;
    float* tail;
    float* xbuf;
    int act0 = 0, act1 = 0;     // hidden activations (rl_activation codes, wave-uniform; RolloutDev.act0 / act1)

    __device__ __forceinline__ void init(float* smem, const float* __restrict__ theta) {
        fa0 = smem;
        fa1 = fa0 + N::FA0;
        tail = fa1 + N::FA1;
        xbuf = tail + N::TAILP;
        stage_fragments<N, BLOCK>(theta, fa0, fa1, nullptr);
        for (int k = threadIdx.x; k < N::TAIL; k += BLOCK) tail[k] = theta[N::B1 + k];
        for (int k = threadIdx.x; k < XROWS * WV; k += BLOCK) xbuf[k] = (k / WV == Env::OBS) ? 1.0f : 0.0f;
        __syncthreads();
    }

    __device__ __forceinline__ float log_std(int k) const { return tail[T_LS + k]; }

    // Lane-group rollouts keep 16 envs per wavefront, replicated on lanes l and l + 16k: one block of 32
    // "samples" (lanes 0..31) then covers every env and all four copies read the same result.
    __device__ __forceinline__ void forward16(const float* o, float* mean) const { run<1>(o, mean); }

    // o: this lane's observation; mean: this lane's action mean
    __device__ __forceinline__ void forward(const float* o, float* mean) const { run<2>(o, mean); }

    template <int NBLK>
    __device__ __forceinline__ void run(const float* o, float* mean) const {
        constexpr int DO = Env::OBS, DA = Env::ACT, HT = N::HT, KS0 = N::KS0, KS1 = N::KS1;
        const int lane = threadIdx.x, lj = lane & 31, lh = lane >> 5;
        wave_sync();                                          // the previous step's reads are done
#pragma unroll
        for (int d = 0; d < DO; ++d) xbuf[d * WV + lane] = o[d];
        wave_sync();
#pragma unroll
        for (int blk = 0; blk < NBLK; ++blk) {
            float xb[KS0];
#pragma unroll
            for (int m = 0; m < KS0; ++m) xb[m] = xbuf[(2 * m + lh) * WV + 32 * blk + lj];
            f32x16 h0[HT], h1[HT];
#pragma unroll
            for (int t = 0; t < HT; ++t) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
                for (int m = 0; m < KS0; ++m) acc = mfma(fa0[(t * KS0 + m) * WV + lane], xb[m], acc);
                if constexpr (ACTS) act_frag(h0[t], acc, act0);
                else act_frag(h0[t], acc, 0);
            }
#pragma unroll
            for (int t = 0; t < HT; ++t) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = tail[T_B1 + 32 * t + frag_unit(r, 0) + 4 * lh];
#pragma unroll
                for (int m = 0; m < KS1; ++m) acc = mfma(fa1[(t * KS1 + m) * WV + lane], h0[m / 16][m % 16], acc);
                if constexpr (ACTS) act_frag(h1[t], acc, act1);
                else act_frag(h1[t], acc, 0);
            }
#pragma unroll
            for (int k = 0; k < DA; ++k) {
                float pm = 0.0f;
#pragma unroll
                for (int t = 0; t < HT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        pm = __builtin_fmaf(h1[t][r], tail[T_W2 + (32 * t + frag_unit(r, 0) + 4 * lh) * DA + k], pm);
                const float mk = tail[T_B2 + k] + half_sum(pm);
                if (NBLK == 1 || lh == blk) mean[k] = mk;
            }
        }
    }
};

// The same network for the lane-group rollouts (16 envs per wavefront, env e replicated on lanes e + 16 g):
// v_mfma_f32_16x16x4_f32 tiles, Z^T[unit][env] = W^T[unit][k] X^T[k][env] with N = the 16 envs, so nothing is
// computed twice across the four replicas -- lane group g supplies input row 4 m + g at k-step m and receives
// units 16 t + 4 g + j (j = register) of output tile t, which after tanh are again one k-row per register of the
// next layer.  Half the matrix passes and half the tanh of the 32x32x2 form; all weight fragments, biases and
// output-layer columns are loop invariant and live in VGPRs (one wavefront per SIMD owns 512 registers), so the
// policy touches no LDS at all.  The DA output columns are per-lane dot products over the lane's 4 NT units,
// folded across the lane groups with v_permlane32_swap / v_permlane16_swap (two columns share one fold).
using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// v_permlane32_swap: lanes [32,64) of a <-> lanes [0,32) of b, i.e. a' = [a.lo | b.lo], b' = [a.hi | b.hi];
// v_permlane16_swap: odd rows of a <-> even rows of b, a' = [a.r0 b.r0 a.r2 b.r2], b' = [a.r1 b.r1 a.r3 b.r3].
// Inline asm: this compiler's __builtin_amdgcn_permlane{16,32}_swap returns element 0 for both results.  The
// s_nop covers the VALU-write -> permlane-read hazard the assembler does not see inside an asm block.
__device__ __forceinline__ void swap32(float a, float b, float& r0, float& r1) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    r0 = a;
    r1 = b;
}
__device__ __forceinline__ void swap16(float a, float b, float& r0, float& r1) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    r0 = a;
    r1 = b;
}
// sum over the four lanes e, e + 16, e + 32, e + 48, result on all of them
__device__ __forceinline__ float group_sum(float v) {
    float a, b;
    swap32(v, v, a, b);
    const float s = a + b;
    swap16(s, s, a, b);
    return a + b;
}
// the same for two values at the price of one: returns sum(v0) and sum(v1) on every lane
__device__ __forceinline__ void group_sum2(float v0, float v1, float& s0, float& s1) {
    float a, b;
    swap32(v0, v1, a, b);            // a = [v0.lo | v1.lo], b = [v0.hi | v1.hi]
    const float h = a + b;           // lanes < 32: v0 folded over the halves, lanes >= 32: v1
    swap16(h, h, a, b);
    const float q = a + b;           // folded over the rows of each half
    swap32(q, q, s0, s1);            // low half everywhere, high half everywhere
}

template <class Env, int H, bool ACTS = false>
struct RolloutPolicy16 {
    using N = Net<Env::OBS, Env::ACT, H>;
    static constexpr int DO = Env::OBS, DA = Env::ACT;
    static constexpr int NT = H / 16;                  // 16-unit tiles per hidden layer
    static constexpr int KS0 = (DO + 1 + 3) / 4;       // k-steps of layer 0 (inputs + the bias slot)
    static constexpr int KS1 = H / 4;                  // k-steps of layer 1: step 4 t + j feeds register j of tile t

    float a0[NT][KS0];      // W0[4 m + g][16 t + n]  (row DO = b0, rows beyond = 0)
    float a1[NT][KS1];      // W1[16 (s / 4) + 4 g + s % 4][16 t + n]
    float b1[NT][4];        // b1[16 t + 4 g + j]
    float w2[DA][NT][4];    // W2[16 t + 4 g + j][k]
    float b2[DA], lstd[DA];
    bool g0, g1, g2;
    int act0 = 0, act1 = 0;     // hidden activations (rl_activation codes, wave-uniform)

    __device__ __forceinline__ void init(const float* __restrict__ th) {
        const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15;
        g0 = g == 0; g1 = g == 1; g2 = g == 2;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int m = 0; m < KS0; ++m) {
                const int d = 4 * m + g, u = 16 * t + n;
                a0[t][m] = d < DO ? th[N::W0 + d * H + u] : (d == DO ? th[N::B0 + u] : 0.0f);
            }
#pragma unroll
            for (int s_ = 0; s_ < KS1; ++s_)
                a1[t][s_] = th[N::W1 + (16 * (s_ / 4) + 4 * g + s_ % 4) * H + 16 * t + n];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                b1[t][j] = th[N::B1 + 16 * t + 4 * g + j];
#pragma unroll
                for (int k = 0; k < DA; ++k) w2[k][t][j] = th[N::W2 + (16 * t + 4 * g + j) * DA + k];
            }
        }
#pragma unroll
        for (int k = 0; k < DA; ++k) {
            b2[k] = th[N::B2 + k];
            lstd[k] = th[N::LSTD + k];
        }
    }

    __device__ __forceinline__ float log_std(int k) const { return lstd[k]; }

    __device__ __forceinline__ float pick(float c0, float c1, float c2, float c3) const {
        return g0 ? c0 : g1 ? c1 : g2 ? c2 : c3;
    }
    static __device__ __forceinline__ float input(const float* o, int d) {
        return d < DO ? o[d] : (d == DO ? 1.0f : 0.0f);
    }

    // o: the observation of this lane's env (identical on its four replicas); mean: its action mean, on all four
    __device__ __forceinline__ void forward16(const float* o, float* mean) const {
        float xb[KS0];
#pragma unroll
        for (int m = 0; m < KS0; ++m)
            xb[m] = pick(input(o, 4 * m), input(o, 4 * m + 1), input(o, 4 * m + 2), input(o, 4 * m + 3));
        f32x4 h0[NT], h1[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int m = 0; m < KS0; ++m) acc = mfma16(a0[t][m], xb[m], acc);
#pragma unroll
            for (int j = 0; j < 4; ++j) h0[t][j] = ACTS ? act_one(acc[j], act0) : ftanh(acc[j]);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f32x4 acc;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = b1[t][j];
#pragma unroll
            for (int s_ = 0; s_ < KS1; ++s_) acc = mfma16(a1[t][s_], h0[s_ / 4][s_ % 4], acc);
#pragma unroll
            for (int j = 0; j < 4; ++j) h1[t][j] = ACTS ? act_one(acc[j], act1) : ftanh(acc[j]);
        }
        float pm[DA];
#pragma unroll
        for (int k = 0; k < DA; ++k) {
            pm[k] = 0.0f;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) pm[k] = __builtin_fmaf(h1[t][j], w2[k][t][j], pm[k]);
        }
#pragma unroll
        for (int k = 0; k + 1 < DA; k += 2) {
            float s0, s1;
            group_sum2(pm[k], pm[k + 1], s0, s1);
            mean[k] = b2[k] + s0;
            mean[k + 1] = b2[k + 1] + s1;
        }
        if (DA & 1) mean[DA - 1] = b2[DA - 1] + group_sum(pm[DA - 1]);
    }
};

// The mean network of the WIDE / DEEP policies (two or three tanh layers of 32 / 64 / 128 units after zero padding,
// policy_wide.h; GaussianMLPPolicy(hidden_sizes=...) is free-form, gaussian_mlp_policy.py:21-58): the chain of
// RolloutPolicy with the shape at run time.  Per layer the weight fragments sit in LDS as [row tile][k-step][lane]
// (k in the register order of the producing fragment, so activations stay in registers between layers); a wavefront
// carries up to 4 fragments per layer (row-tile loops are bounded by 4 and predicated on the wave-uniform tile count).
// One env-step costs sum_l HT_l * KS_l matrix instructions per block of 32 envs -- 284 for (13 -> 128 -> 128) -- so
// these rollouts are bound by the policy, not by the physics.
template <class Env>
struct RolloutPolicyWide {
    static constexpr int DO = Env::OBS, DA = Env::ACT;
    static constexpr int KS0 = (DO + 2) / 2;
    static constexpr int XROWS = 2 * KS0;
    WideShape s;
    float* f[WIDE_MAX_L];
    float* tail;
    float* xbuf;

    static size_t lds_floats(const WideShape& s, int threads) {
        size_t n = (size_t)s.HT[0] * KS0 * WV;
        for (int l = 1; l < s.L; ++l) n += (size_t)s.HT[l] * (s.H[l - 1] / 2) * WV;
        n += s.tail;
        n += (size_t)XROWS * threads;                       // one [input][lane] tile per wavefront
        return n;
    }

    __device__ __forceinline__ void init(float* smem, const float* __restrict__ th, const WideShape& shape) {
        s = shape;
        const int nt = blockDim.x;
        float* o = smem;
        f[0] = o; o += s.HT[0] * KS0 * WV;
        for (int l = 1; l < WIDE_MAX_L; ++l) { f[l] = o; o += (l < s.L) ? s.HT[l] * (s.H[l - 1] / 2) * WV : 0; }
        tail = o; o += s.tail;
        xbuf = o + (threadIdx.x >> 6) * XROWS * WV;
        for (int e = threadIdx.x; e < s.HT[0] * KS0 * WV; e += nt) {
            const int l_ = e % WV, m = (e / WV) % KS0, t = e / (WV * KS0);
            const int i = 32 * t + (l_ & 31), d = 2 * m + (l_ >> 5);
            f[0][e] = d < DO ? th[s.oW[0] + d * s.H[0] + i] : (d == DO ? th[s.ob[0] + i] : 0.0f);
        }
        // every index into the shape's arrays is a compile-time constant (a run-time index would move the whole
        // struct to scratch memory)
#pragma unroll
        for (int l = 1; l < WIDE_MAX_L; ++l)
            if (l < s.L) {
                const int ks = s.H[l - 1] / 2;
                for (int e = threadIdx.x; e < s.HT[l] * ks * WV; e += nt) {
                    const int l_ = e % WV, m = (e / WV) % ks, t = e / (WV * ks);
                    const int i = 32 * t + (l_ & 31), k = 32 * (m / 16) + frag_unit(m % 16, l_ >> 5);
                    f[l][e] = th[s.oW[l] + k * s.H[l] + i];
                }
            }
        for (int k = threadIdx.x; k < s.tail; k += nt) {
            int src;
            if (k < s.tWo) {
                if (s.L > 2 && k >= s.tb[2]) src = s.ob[2] + (k - s.tb[2]);
                else src = s.ob[1] + (k - s.tb[1]);
            } else if (k < s.tbo) src = s.oWo + (k - s.tWo);
            else if (k < s.tls) src = s.obo + (k - s.tbo);
            else src = (k - s.tls) < DA ? s.ols + (k - s.tls) : -1;
            tail[k] = src >= 0 ? th[src] : 0.0f;
        }
        for (int k = threadIdx.x & 63; k < XROWS * WV; k += WV) xbuf[k] = (k / WV == DO) ? 1.0f : 0.0f;
        __syncthreads();
    }

    __device__ __forceinline__ float log_std(int k) const { return tail[s.tls + k]; }
    __device__ __forceinline__ void forward16(const float* o, float* mean) const { run<1>(o, mean); }
    __device__ __forceinline__ void forward(const float* o, float* mean) const { run<2>(o, mean); }

    template <int NBLK>
    __device__ __forceinline__ void run(const float* o, float* mean) const {
        const int lane = threadIdx.x & 63, lj = lane & 31, lh = lane >> 5;
        wave_sync();
#pragma unroll
        for (int d = 0; d < DO; ++d) xbuf[d * WV + lane] = o[d];
        wave_sync();
        const int L = s.L;
#pragma unroll
        for (int blk = 0; blk < NBLK; ++blk) {
            float xb[KS0];
#pragma unroll
            for (int m = 0; m < KS0; ++m) xb[m] = xbuf[(2 * m + lh) * WV + 32 * blk + lj];
            f32x16 h[2][4];                                  // ping-pong: layer l reads h[(l + 1) & 1], writes h[l & 1]
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (t < s.HT[0]) {
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
                    for (int m = 0; m < KS0; ++m) acc = mfma(f[0][(t * KS0 + m) * WV + lane], xb[m], acc);
#pragma unroll
                    for (int r = 0; r < 16; ++r) h[0][t][r] = (s.ident & 1) ? acc[r] : ftanh(acc[r]);
                }
#pragma unroll
            for (int l = 1; l < WIDE_MAX_L; ++l)
                if (l < L) {
                    const int ks = s.H[l - 1] / 2;
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (t < s.HT[l]) {
                            f32x16 acc;
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[r] = tail[s.tb[l] + 32 * t + frag_unit(r, 0) + 4 * lh];
#pragma unroll
                            for (int tt = 0; tt < 4; ++tt)
                                if (tt < s.HT[l - 1]) {
#pragma unroll
                                    for (int mm = 0; mm < 16; ++mm)
                                        acc = mfma(f[l][(t * ks + 16 * tt + mm) * WV + lane], h[(l + 1) & 1][tt][mm], acc);
                                }
#pragma unroll
                            for (int r = 0; r < 16; ++r) h[l & 1][t][r] = ((s.ident >> l) & 1) ? acc[r] : ftanh(acc[r]);
                        }
                }
            // the last hidden layer sits in h[(L - 1) & 1]: both cases spelled out, a run-time index would send the
            // fragments to scratch memory
            if (L == 3) head<NBLK>(h[0], blk, mean);
            else head<NBLK>(h[1], blk, mean);
        }
    }

    template <int NBLK>
    __device__ __forceinline__ void head(const f32x16 (&hl)[4], int blk, float* mean) const {
        const int lane = threadIdx.x & 63, lh = lane >> 5;
        const int HTL = s.L == 3 ? s.HT[2] : s.HT[1];
#pragma unroll
        for (int k = 0; k < DA; ++k) {
            float pm = 0.0f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (t < HTL) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        pm = __builtin_fmaf(hl[t][r], tail[s.tWo + (32 * t + frag_unit(r, 0) + 4 * lh) * DA + k], pm);
                }
            const float mk = tail[s.tbo + k] + half_sum(pm);
            if (NBLK == 1 || lh == blk) mean[k] = mk;
        }
    }
};

// The same wide / deep mean network for the lane-group Swimmer, evaluated by FOUR wavefronts per group of 16 envs
// (rollout_swimmer_quad_coop_kernel).  A lone wavefront pays for every instruction it issues, so the 284 matrix
// instructions of a (13 -> 128 -> 128) forward pass on 32-column tiles -- half of them idle: the group holds 16 envs --
// cost more than the 50 physics sub-steps.  Here the workgroup is the group's four replicas-in-time: every wavefront
// carries the same 16 envs through the same instruction stream (physics, noise, resets: identical values, wavefront 0
// stores), and the network is split by OUTPUT UNITS: a layer of H units is H / 16 blocks of v_mfma_f32_16x16x4_f32
// (16 units x 16 envs x 4 inputs), wavefront w takes blocks w, w + 4; activations cross the wavefronts through LDS
// ([k / 16][env, k % 4][(k / 4) % 4]: a lane's operands of four k-steps are one 16-byte read), one barrier per layer.
// The output layer is split by inputs instead (each wavefront contracts a quarter of the last hidden layer), the partial
// means meet in LDS.  Two accumulators per block (even / odd k-steps), so dependent matrix instructions never queue.
// Per env-step and wavefront: (13 -> 128 -> 128 -> 2) 8 + 64 + 8 matrix instructions instead of 284 of twice the length.
constexpr int COOP_WAVES = 4;

template <class Env>
struct RolloutPolicyCoop {
    static constexpr int DO = Env::OBS, DA = Env::ACT;
    static constexpr int K0 = (DO + 3) & ~3;               // layer-0 inputs, padded to whole k-steps (b0 initialises the accumulator)
    static_assert(K0 == 16 || K0 == 32, "whole groups of four k-steps");
    WideShape s;
    float* wf[WIDE_MAX_L];      // layer l: [block][k-step group][lane][k-step % 4] = W_l[4 m + lane / 16][16 block + lane % 16]
    float* wo;                  // output layer: [k-step group][lane][k-step % 4] = Wo[4 m + lane / 16][lane % 16] (zero beyond DA)
    float* bias[WIDE_MAX_L];    // b_l
    float* tailo;               // bo [DA], log_std [DA]
    float* act[2];              // activations (ping-pong)
    float* part;                // partial means [action][env][wavefront]
    int wave;

    static size_t lds_floats(const WideShape& s) {
        size_t n = (size_t)K0 * s.H[0];
        int hmax = s.H[0];
        for (int l = 1; l < s.L; ++l) {
            n += (size_t)s.H[l - 1] * s.H[l];
            if (s.H[l] > hmax) hmax = s.H[l];
        }
        n += (size_t)16 * s.H[s.L - 1];                    // wo
        for (int l = 0; l < s.L; ++l) n += s.H[l];         // biases
        n += 4 * ((2 * DA + 3) / 4);
        n += 2 * (size_t)16 * hmax;                        // act
        n += (size_t)DA * 16 * COOP_WAVES;                 // part
        return n;
    }

    __device__ __forceinline__ void init(float* smem, const float* __restrict__ th, const WideShape& shape) {
        s = shape;
        wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int nt = blockDim.x;
        float* o = smem;
        int hmax = s.H[0];
#pragma unroll
        for (int l = 0; l < WIDE_MAX_L; ++l) {
            const int K = l == 0 ? K0 : s.H[l > 0 ? l - 1 : 0];
            wf[l] = o;
            o += (l < s.L) ? K * s.H[l] : 0;
            if (l < s.L && s.H[l] > hmax) hmax = s.H[l];
        }
        const int HL = s.L == 3 ? s.H[2] : s.H[1];
        wo = o; o += 16 * HL;
#pragma unroll
        for (int l = 0; l < WIDE_MAX_L; ++l) { bias[l] = o; o += (l < s.L) ? s.H[l] : 0; }
        tailo = o; o += 4 * ((2 * DA + 3) / 4);
        act[0] = o; o += 16 * hmax;
        act[1] = o; o += 16 * hmax;
        part = o;
#pragma unroll
        for (int l = 0; l < WIDE_MAX_L; ++l)
            if (l < s.L) {
                const int K = l == 0 ? K0 : s.H[l > 0 ? l - 1 : 0], G = K / 16, H = s.H[l];
                for (int e = threadIdx.x; e < K * H; e += nt) {
                    const int j = e & 3, ln = (e >> 2) & 63, mg = (e >> 8) % G, b = (e >> 8) / G;
                    const int k = 4 * (4 * mg + j) + (ln >> 4), i = 16 * b + (ln & 15);
                    wf[l][e] = (l > 0 || k < DO) ? th[s.oW[l] + k * H + i] : 0.0f;
                }
                for (int e = threadIdx.x; e < H; e += nt) bias[l][e] = th[s.ob[l] + e];
            }
        for (int e = threadIdx.x; e < 16 * HL; e += nt) {
            const int j = e & 3, ln = (e >> 2) & 63, mg = e >> 8, k = 4 * (4 * mg + j) + (ln >> 4), a = ln & 15;
            wo[e] = a < DA ? th[s.oWo + k * DA + a] : 0.0f;
        }
        for (int e = threadIdx.x; e < 2 * DA; e += nt) tailo[e] = e < DA ? th[s.obo + e] : th[s.ols + (e - DA)];
        __syncthreads();
    }

    __device__ __forceinline__ float log_std(int k) const { return tailo[DA + k]; }

    // one layer's blocks of this wavefront: NBW = 1 or 2 (blocks wave, wave + 4); B operands: xg[g] = the lane's four
    // k-steps of group g (registers: layer 0) or read from `src`
    template <int NBW, bool FROM_REGS>
    __device__ __forceinline__ void blocks(const float* wfl, const float* bl, int G, const f32x4* xg, const float* src,
                                           float* dst, bool ident) const {
        const int lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4;
        const int lp = n * 4 + q;                               // the lane's slot inside an activation group
        f32x4 acc[NBW][2];
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            acc[j][0] = *reinterpret_cast<const f32x4*>(bl + 16 * (wave + COOP_WAVES * j) + 4 * q);
            acc[j][1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
        const f32x4* wv = reinterpret_cast<const f32x4*>(wfl) + lane;
        const f32x4* sv = reinterpret_cast<const f32x4*>(src) + lp;
        const int groups = FROM_REGS ? K0 / 16 : G;            // (a constant for the register operands: xg stays in registers)
        // the operands of group g + 1 are read while the products of group g run (the last group reads itself again)
        f32x4 bv, av[NBW];
        if constexpr (FROM_REGS) bv = xg[0];
        else bv = sv[0];
#pragma unroll
        for (int j = 0; j < NBW; ++j) av[j] = wv[(wave + COOP_WAVES * j) * G * 64];
        for (int g = 0; g < groups; ++g) {
            const int gn = g + 1 < groups ? g + 1 : g;
            f32x4 bn, an[NBW];
            if constexpr (FROM_REGS) bn = xg[K0 / 16 > 1 ? 1 : 0];
            else bn = sv[gn * 64];
#pragma unroll
            for (int j = 0; j < NBW; ++j) an[j] = wv[((wave + COOP_WAVES * j) * G + gn) * 64];
            __builtin_amdgcn_sched_barrier(0);                  // (the reads stay in front of the products)
#pragma unroll
            for (int j = 0; j < NBW; ++j) {
                acc[j][0] = mfma16(av[j][0], bv[0], acc[j][0]);
                acc[j][1] = mfma16(av[j][1], bv[1], acc[j][1]);
                acc[j][0] = mfma16(av[j][2], bv[2], acc[j][0]);
                acc[j][1] = mfma16(av[j][3], bv[3], acc[j][1]);
            }
            bv = bn;
#pragma unroll
            for (int j = 0; j < NBW; ++j) av[j] = an[j];
        }
        // unit 16 b + 4 q + r of env n -> k-step m' = 4 b + q (group b, sub-step q), slot (n, r)
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            const int b = wave + COOP_WAVES * j;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float z = acc[j][0][r] + acc[j][1][r];
                dst[b * 256 + (n * 4 + r) * 4 + q] = ident ? z : ftanh(z);
            }
        }
    }
    template <bool FROM_REGS>
    __device__ __forceinline__ void layer(int l, int K, const f32x4* xg, const float* src, float* dst) const {
        const int nb = s.H[l] / 16;                            // 2, 4 or 8 blocks
        const bool ident = (s.ident >> l) & 1;                  // (WideShape.ident: an identity layer)
        if (wave + COOP_WAVES < nb) blocks<2, FROM_REGS>(wf[l], bias[l], K / 16, xg, src, dst, ident);
        else if (wave < nb) blocks<1, FROM_REGS>(wf[l], bias[l], K / 16, xg, src, dst, ident);
        __syncthreads();
    }

    // o: the observation of this lane's env (identical on its four replicas and on the four wavefronts); mean: its action
    // mean, on all of them
    __device__ __forceinline__ void forward16(const float* o, float* mean) const {
        const int lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4;
        f32x4 xg[K0 / 16];
#pragma unroll
        for (int g = 0; g < K0 / 16; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int d = 4 * (4 * g + j);                  // k = d + q
                const float c0 = d < DO ? o[d < DO ? d : 0] : 0.0f, c1 = d + 1 < DO ? o[d + 1 < DO ? d + 1 : 0] : 0.0f;
                const float c2 = d + 2 < DO ? o[d + 2 < DO ? d + 2 : 0] : 0.0f, c3 = d + 3 < DO ? o[d + 3 < DO ? d + 3 : 0] : 0.0f;
                xg[g][j] = q == 0 ? c0 : q == 1 ? c1 : q == 2 ? c2 : c3;
            }
        layer<true>(0, K0, xg, nullptr, act[0]);
        layer<false>(1, s.H[0], nullptr, act[0], act[1]);
        const float* last = act[1];
        int HL = s.H[1];
        if (s.L == 3) {
            layer<false>(2, s.H[1], nullptr, act[1], act[0]);
            last = act[0];
            HL = s.H[2];
        }
        // output layer: this wavefront's quarter of the k-steps (all operand reads in front of the products)
        {
            const int per = HL / 16;                            // k-steps per wavefront: 2, 4 or 8
            const f32x4* wov = reinterpret_cast<const f32x4*>(wo) + lane;
            const f32x4* lv = reinterpret_cast<const f32x4*>(last) + (n * 4 + q);
            f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
            if (per >= 4) {
                const int g0 = wave * (per / 4), g1 = per == 8 ? g0 + 1 : g0;
                const f32x4 a0 = wov[g0 * 64], b0 = lv[g0 * 64], a1 = wov[g1 * 64], b1 = lv[g1 * 64];
                __builtin_amdgcn_sched_barrier(0);
                acc0 = mfma16(a0[0], b0[0], acc0);
                acc1 = mfma16(a0[1], b0[1], acc1);
                acc0 = mfma16(a0[2], b0[2], acc0);
                acc1 = mfma16(a0[3], b0[3], acc1);
                if (per == 8) {
                    acc0 = mfma16(a1[0], b1[0], acc0);
                    acc1 = mfma16(a1[1], b1[1], acc1);
                    acc0 = mfma16(a1[2], b1[2], acc0);
                    acc1 = mfma16(a1[3], b1[3], acc1);
                }
            } else {                                            // half a group: k-steps 2 wave, 2 wave + 1
                const int g0 = wave >> 1;
                const bool hi = (wave & 1) != 0;
                const f32x4 a0 = wov[g0 * 64], b0 = lv[g0 * 64];
                acc0 = mfma16(hi ? a0[2] : a0[0], hi ? b0[2] : b0[0], acc0);
                acc1 = mfma16(hi ? a0[3] : a0[1], hi ? b0[3] : b0[1], acc1);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * q + r < DA) part[((4 * q + r) * 16 + n) * COOP_WAVES + wave] = acc0[r] + acc1[r];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < DA; ++k) {
            const f32x4 p = *reinterpret_cast<const f32x4*>(part + (k * 16 + n) * COOP_WAVES);
            mean[k] = tailo[k] + ((p[0] + p[1]) + (p[2] + p[3]));
        }
    }
};

struct RolloutDev {
    int n, T, max_path_length, normalize, reset_at_start, env_offset;
    float scale_reward, log_min_std;
    uint64_t seed, step_counter;
    float* state;
    int32_t* ts;
    const float* theta;
    const float* eps;
    const float* reset_draws;
    float* obs;
    float* actions;
    float* means;
    float* rewards;
    uint8_t* dones;
    float* last_obs;
    EnvCfg cfg;
    const float* act_noise_z;   // [T][Da][n] injected N(0,1) draws of the env's action noise, or null (Philox)
    const float* obs_noise_z;   // [T+1][Do][n]: slice 0 = the first observation, slice t + 1 = the one after step t
    float* log_stds;            // [Da][T][n] agent_info "log_std" of a policy with a log-std NETWORK (else unused)
    int act0, act1;             // hidden activations of the equal-width (32,32) / (64,64) policies (rl_activation codes)
    // rl_running_norm (NORM instantiations of the generic rollout): per-env running estimates, in place
    double* nobs_mean; double* nobs_var; double* nrew_mean; double* nrew_var;
    double obs_alpha, rew_alpha;
    int norm_obs, norm_rew;
};

// A policy whose log-std is a second network on the observation (GaussianMLPPolicy(adaptive_std=True) / std_network=...,
// gaussian_mlp_policy.py:60-98): two RolloutPolicyWide side by side in LDS, the same observation through both.
template <class Env>
struct RolloutPolicyDual {
    RolloutPolicyWide<Env> mean_net, std_net;
    static size_t lds_floats(const WideShape& ms, const WideShape& ss, int threads) {
        return RolloutPolicyWide<Env>::lds_floats(ms, threads) + RolloutPolicyWide<Env>::lds_floats(ss, threads);
    }
    __device__ __forceinline__ void init(float* smem, const float* __restrict__ th_mean, const WideShape& ms,
                                         const float* __restrict__ th_std, const WideShape& ss, size_t mean_floats) {
        mean_net.init(smem, th_mean, ms);
        std_net.init(smem + mean_floats, th_std, ss);
    }
    __device__ __forceinline__ float log_std(int) const { return 0.0f; }      // per step instead: log_std16 / log_std64
    __device__ __forceinline__ void forward16(const float* o, float* mean) const { mean_net.forward16(o, mean); }
    __device__ __forceinline__ void forward(const float* o, float* mean) const { mean_net.forward(o, mean); }
    __device__ __forceinline__ void log_std16(const float* o, float* ls) const { std_net.forward16(o, ls); }
    __device__ __forceinline__ void log_std64(const float* o, float* ls) const { std_net.forward(o, ls); }
};
template <class P> struct state_std : std::false_type {};
template <class Env> struct state_std<RolloutPolicyDual<Env>> : std::true_type {};


// EPW = envs per wavefront.  64: one env per lane, the throughput shape (every lane does useful physics).
// 16: env e lives on lanes e, e+16, e+32, e+48 -- the physics is replicated (free: a lone wavefront is bound by its
// instruction stream, not by lanes) and the policy runs on 16x16x4 tiles with nothing computed twice
// (RolloutPolicy16): a quarter of the matrix passes and half the tanh per env-step, on four times as many
// wavefronts.  The latency shape, chosen while n / 16 wavefronts still find a SIMD each.
// The lane-group shapes are compiled for workgroups of up to FOUR wavefronts, each with its own envs.  Up to 256
// wavefronts are launched as single-wavefront workgroups (one per CU); beyond that, four per workgroup, which puts
// one on each SIMD of a CU -- single-wavefront workgroups are not spread evenly over the SIMDs once a CU holds
// several (16 384 Swimmer envs = 1024 wavefronts took 1.9x the time of 256).
constexpr int LANE_TPB = 256;

// v[0..NP) -> NP planes of a [NP][T][n] array at (t, i): the scalar row pointer walks the planes (one s_add_u32 /
// s_addc_u32 pair per plane), every store is the scalar-base + 32-bit lane offset form of global_store.  The asm
// statements emit nothing: they keep the walk a walk (NP hoisted 64-bit plane bases do not fit the scalar register
// file next to the argument block and come back as v_readlane reloads + vector address adds) and the zero-extension
// of the lane offset next to its add, which is what the back-end's scalar-base addressing pattern needs.
template <int NP, typename V>
__device__ __forceinline__ void store_planes(V* row_ptr, size_t plane, uint32_t& lane_bytes, const V* v) {
    typedef __attribute__((address_space(1))) V* global_ptr;
    uintptr_t walk = reinterpret_cast<uintptr_t>(row_ptr);          // address of plane k's row, a scalar
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        asm volatile("" : "+s"(walk));
        asm volatile("" : "+v"(lane_bytes));
        *reinterpret_cast<global_ptr>(walk + lane_bytes) = v[k];
        walk += plane * sizeof(V);
    }
}

// NORM: NormalizedEnv(normalize_obs / normalize_reward) -- the env copy's running estimates live in registers for the
// rollout (float64: 2 Do + 2 values), are fed and applied in the reference's order (rl_running_norm in the header) and go
// back to their arrays at the end.  The four replicas of a 16-envs-per-wavefront env carry identical copies.
template <class Env, class Pol, int EPW, bool NORM = false>
__device__ __forceinline__ void rollout_body(const RolloutDev& a, const Pol& pol) {
    const int n = a.n, T = a.T;
    // every lane stays alive (the matrix instructions and the cross-lane exchanges need the whole
    // wavefront); lanes past the last env shadow env n-1 and only their stores are masked
    const int lane = threadIdx.x & 63;
    const int wave_global = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int i_raw = wave_global * EPW + (lane & (EPW - 1));
    const bool live = (i_raw < n) && (lane < EPW);              // one of the replicas stores
    const int i = (i_raw < n) ? i_raw : n - 1;
    const uint32_t env_global = (uint32_t)(a.env_offset + i);
    const size_t plane = (size_t)T * n;

    float std_[Env::ACT];
#pragma unroll
    for (int k = 0; k < Env::ACT; ++k) std_[k] = __expf(fmaxf(pol.log_std(k), a.log_min_std));

    float s[Env::STATE];
    load_state<Env>(a.state, n, i, s);
    int ts = a.ts[i];
    const size_t draws_slice = (size_t)Env::RESET_DRAWS * n;
    if (a.reset_at_start) {
        reset_one<Env>(s, a.reset_draws, n, i, a.seed, env_global, a.step_counter, a.cfg);
        ts = 0;
    }
    float o[Env::OBS];
    float zq[Env::ACT] = {};      // policy noise of four steps, one step per replica group (lane-group shapes)
    const size_t obs_z_slice = (size_t)Env::OBS * n;
    // a continuation (reset_at_start == 0) carries on from the observation the previous launch / rl_vecenv_step / reset
    // left in last_obs -- its observation noise and whitening included, nothing is drawn or fed twice
    const bool resume_obs = !a.reset_at_start && a.last_obs != nullptr;
    if (resume_obs) {
#pragma unroll
        for (int k = 0; k < Env::OBS; ++k) o[k] = a.last_obs[(size_t)k * n + i];
    } else {
        Env::template observe<float>(s, o);
        observed<Env>(o, a.cfg, a.obs_noise_z, n, i, a.seed, env_global, a.step_counter);
    }
    // running normalisation (normalized_env.py:33-49): feed the estimate with an observation, optionally whiten it
    double nm[NORM ? Env::OBS : 1], nv[NORM ? Env::OBS : 1], rm = 0.0, rv = 1.0;
    auto feed_obs = [&](float* ob, bool whiten) {
        if constexpr (NORM) {
            if (a.norm_obs) {
                const double al = a.obs_alpha;
#pragma unroll
                for (int k = 0; k < Env::OBS; ++k) {
                    const double x = (double)ob[k];
                    nm[k] = nm[k] * (1.0 - al) + al * x;
                    const double dlt = x - nm[k];
                    nv[k] = nv[k] * (1.0 - al) + al * (dlt * dlt);
                    if (whiten) ob[k] = (float)(dlt / (sqrt(nv[k]) + 1e-8));
                }
            }
        }
    };
    if constexpr (NORM) {
        if (a.norm_obs) {
#pragma unroll
            for (int k = 0; k < Env::OBS; ++k) { nm[k] = a.nobs_mean[(size_t)k * n + i]; nv[k] = a.nobs_var[(size_t)k * n + i]; }
        }
        if (a.norm_rew) { rm = a.nrew_mean[i]; rv = a.nrew_var[i]; }
        if (!resume_obs) feed_obs(o, true);      // reset() returns the whitened first observation
    }

    uint32_t lane_f32 = (uint32_t)i * 4, lane_u8 = (uint32_t)i;      // byte offset of env i inside a row
    for (int t = 0; t < T; ++t) {
        const size_t off = (size_t)t * n + i;
        const size_t row = (size_t)t * n;
        if (live) store_planes<Env::OBS>(a.obs + row, plane, lane_f32, o);
        float mean[Env::ACT], act[Env::ACT], z[Env::ACT];
        if constexpr (EPW == 16) pol.forward16(o, mean);
        else pol.forward(o, mean);
        if constexpr (state_std<Pol>::value) {
            // the log-std network on the same observation, floored (gaussian_mlp_policy.py:100-101,120-121), recorded
            float ls[Env::ACT];
            if constexpr (EPW == 16) pol.log_std16(o, ls);
            else pol.log_std64(o, ls);
#pragma unroll
            for (int k = 0; k < Env::ACT; ++k) {
                ls[k] = fmaxf(ls[k], a.log_min_std);
                std_[k] = __expf(ls[k]);
            }
            if (live) store_planes<Env::ACT>(a.log_stds + row, plane, lane_f32, ls);
        }
        if (a.eps) {
#pragma unroll
            for (int k = 0; k < Env::ACT; ++k) z[k] = a.eps[k * plane + off];
            landed<Env::ACT>(z);
        } else if constexpr (EPW == 16) {
            // the four replicas of an env draw the noise of FOUR consecutive steps at once -- replica g that of step
            // t + g, the same Philox block (seed; env, step, POLICY) the one-draw-per-step form evaluates -- and
            // every step fetches its row from the replica that holds it: a quarter of the integer multiplies and
            // transcendentals of the draw per env-step
            if ((t & 3) == 0)
                philox_draws<Env::ACT, true>(zq, a.seed, env_global, a.step_counter + (uint64_t)(t + (lane >> 4)),
                                             RNG_POLICY);
#pragma unroll
            for (int k = 0; k < Env::ACT; ++k) z[k] = __shfl(zq[k], (lane & 15) + 16 * (t & 3), 64);
        } else {
            philox_draws<Env::ACT, true>(z, a.seed, env_global, a.step_counter + (uint64_t)t, RNG_POLICY);
        }
#pragma unroll
        for (int k = 0; k < Env::ACT; ++k) act[k] = __builtin_fmaf(z[k], std_[k], mean[k]);  // rnd * exp(log_std) + mean
        if (live) {
            store_planes<Env::ACT>(a.actions + row, plane, lane_f32, act);
            store_planes<Env::ACT>(a.means + row, plane, lane_f32, mean);
        }

        float r;
        bool d;
        step_one<Env>(s, act, a.normalize, a.cfg,
                      a.act_noise_z ? a.act_noise_z + (size_t)t * Env::ACT * n : nullptr, n, i, a.seed, env_global,
                      a.step_counter + (uint64_t)t, o, r, d);
        ts += 1;
        if (a.max_path_length > 0 && ts >= a.max_path_length) d = true;
        if constexpr (NORM) {
            if (a.norm_rew) {             // reward / (sqrt(var) + 1e-8) after the update, then the scale (:85-92)
                const double al = a.rew_alpha, x = (double)r;
                rm = rm * (1.0 - al) + al * x;
                const double dlt = x - rm;
                rv = rv * (1.0 - al) + al * (dlt * dlt);
                r = (float)((x / (sqrt(rv) + 1e-8)) * (double)a.scale_reward);
            } else {
                r = r * a.scale_reward;
            }
        }
        if (live) {
            const float rs = NORM ? r : r * a.scale_reward;
            const uint8_t db = d ? 1 : 0;
            store_planes<1>(a.rewards + row, plane, lane_f32, &rs);
            store_planes<1>(a.dones + row, plane, lane_u8, &db);
        }
        if (d) {
            if constexpr (NORM) feed_obs(o, false);      // the terminal observation feeds the estimate, nobody sees it
            const float* dr = a.reset_draws ? a.reset_draws + (size_t)(t + 1) * draws_slice : nullptr;
            reset_one<Env>(s, dr, n, i, a.seed, env_global, a.step_counter + (uint64_t)t + 1, a.cfg);
            Env::template observe<float>(s, o);
            ts = 0;
        }
        observed<Env>(o, a.cfg, a.obs_noise_z ? a.obs_noise_z + (size_t)(t + 1) * obs_z_slice : nullptr, n, i, a.seed,
                      env_global, a.step_counter + (uint64_t)t + 1);
        if constexpr (NORM) feed_obs(o, true);
    }
    if constexpr (NORM) {
        if (live) {
            if (a.norm_obs) {
#pragma unroll
                for (int k = 0; k < Env::OBS; ++k) { a.nobs_mean[(size_t)k * n + i] = nm[k]; a.nobs_var[(size_t)k * n + i] = nv[k]; }
            }
            if (a.norm_rew) { a.nrew_mean[i] = rm; a.nrew_var[i] = rv; }
        }
    }
    if (live) {
        store_state<Env>(a.state, n, i, s);
        a.ts[i] = ts;
        if (a.last_obs) {
#pragma unroll
            for (int k = 0; k < Env::OBS; ++k) a.last_obs[(size_t)k * n + i] = o[k];
        }
    }
}

// ACTS = false: tanh layers, the instruction stream of every earlier round; true: the hidden activations are taken from
// RolloutDev.act0 / act1 at run time (rectify layers, the identity layer of a one-hidden-layer policy)
template <class Env, int H0, int H1, int EPW, bool ACTS = false, bool NORM = false>
__global__ void __launch_bounds__(EPW == 16 ? LANE_TPB : BLOCK) rollout_kernel(RolloutDev a) {
    static_assert(H0 == H1, "the fused rollout is built for equal hidden sizes");
    static_assert(EPW == 64 || EPW == 16, "64 (env per lane) or 16 (four replicas)");
    using Pol = typename std::conditional<EPW == 16, RolloutPolicy16<Env, H0, ACTS>, RolloutPolicy<Env, H0, ACTS>>::type;
    Pol pol;
    if constexpr (EPW == 16) {
        pol.init(a.theta);
        pol.act0 = a.act0; pol.act1 = a.act1;
    } else {
        __shared__ __attribute__((aligned(16))) float smem[RolloutPolicy<Env, H0, ACTS>::LDS_FLOATS];
        pol.init(smem, a.theta);
        pol.act0 = a.act0; pol.act1 = a.act1;
    }
    rollout_body<Env, Pol, EPW, NORM>(a, pol);
}

// the same rollout for the wide / deep policies (RolloutPolicyWide: shape at run time, weight fragments in LDS)
template <class Env, int EPW>
__global__ void __launch_bounds__(EPW == 16 ? LANE_TPB : BLOCK) rollout_wide_kernel(RolloutDev a, WideShape shape) {
    extern __shared__ __attribute__((aligned(16))) float wide_smem[];
    RolloutPolicyWide<Env> pol;
    pol.init(wide_smem, a.theta, shape);
    rollout_body<Env, RolloutPolicyWide<Env>, EPW>(a, pol);
}

// ... and for policies with a log-std network: both networks' weight fragments in LDS
template <class Env, int EPW>
__global__ void __launch_bounds__(EPW == 16 ? LANE_TPB : BLOCK) rollout_dual_kernel(RolloutDev a, WideShape mean_shape,
                                                                                    WideShape std_shape,
                                                                                    const float* __restrict__ theta_std,
                                                                                    int mean_floats) {
    extern __shared__ __attribute__((aligned(16))) float wide_smem[];
    RolloutPolicyDual<Env> pol;
    pol.init(wide_smem, a.theta, mean_shape, theta_std, std_shape, (size_t)mean_floats);
    rollout_body<Env, RolloutPolicyDual<Env>, EPW>(a, pol);
}

// ---------------------------------------------------------------------------
// Lane-group rollout of the Swimmer: 16 envs per wavefront.  Everything per env-step (policy on the matrix
// cores, noise, action map, observation, reward, record, reset) runs env-per-lane exactly as in rollout_kernel
// -- replicated on the four lanes l, l+16, l+32, l+48, which costs nothing: a lone wavefront is bound by its
// instruction stream, not by lanes -- and only the 50 physics sub-steps switch to FOUR LANES PER ENV
// (dyn_swimmer_chain.h: one body per lane, quad-permute DPP exchange, x / y pairs on packed f32, replicated 3x3
// solve): 89 instead of ~260 instructions per sub-step, on four times as many wavefronts.  The lane-group state stays
// resident across env-steps; only the motor torques go in and the observation / reward inputs come back.
// ---------------------------------------------------------------------------
struct DppQuad {
    template <int CTRL> __device__ __forceinline__ float qp(float v) const {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
    }
};

// prefix sums along the chain, one body per lane (role b): v_0, v_0 + v_1, (v_0 + v_1) + v_2; role 3: 0
__device__ __forceinline__ float prefix3(const DppQuad& x, float v, int b) {
    constexpr int PAR1 = 0x93;      // quad_perm [3,0,1,2]: every lane reads its parent
    const float p1 = x.qp<PAR1>(v) + v;
    const float p2 = x.qp<PAR1>(p1) + v;
    return (b == 0) ? v : (b == 1) ? p1 : (b == 2) ? p2 : 0.0f;
}

constexpr int QUAD_ENVS = 16;   // envs per wavefront
// Unroll factors of the sub-step loops: the Swimmer's lane-group kernels, the one-env-per-wavefront kernel of the two-legged
// envs (its four sub-steps as one block: -3.7 % per rollout, tools/exp/rollout_lines.py; the 16-envs-per-wavefront kernels
// keep the loop -- every unrolled copy of that 800-instruction body costs minutes of compile time per instantiation).
#ifndef RL_SWIMMER_SUBSTEP_UNROLL
#define RL_SWIMMER_SUBSTEP_UNROLL 5
#endif
#ifndef RL_TWO_LEG_SUBSTEP_UNROLL
#define RL_TWO_LEG_SUBSTEP_UNROLL 4
#endif

// COOP: the workgroup's wavefronts all carry the SAME group of envs (RolloutPolicyCoop); wavefront 0 stores
template <class Pol, bool COOP = false>
__device__ __forceinline__ void swimmer_quad_body(const RolloutDev& a, const Pol& pol) {
    using Env = Swimmer;
    using Chain = Env::Chain;

    const int n = a.n, T = a.T;
    const int lane = threadIdx.x & 63;
    const int wave_global = COOP ? (int)blockIdx.x : (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    const int el = lane & (QUAD_ENVS - 1);            // env slot of this lane in the env-per-lane phases
    const int i_raw = wave_global * QUAD_ENVS + el;
    const bool live = (i_raw < n) && (lane < QUAD_ENVS) && (!COOP || threadIdx.x < 64);   // one of the copies stores
    const int i = (i_raw < n) ? i_raw : n - 1;
    const uint32_t env_global = (uint32_t)(a.env_offset + i);
    const size_t plane = (size_t)T * n;
    // lane-group phase: quad q_env = lane / 4 works on the env held by lane q_env, role b = lane % 4
    const int q_src = lane >> 2, b = lane & 3;
    Chain::LaneConst<float> kc = Chain::lane_const<float>(b);
    // the role-dependent constants are values in registers, not selects the optimiser may re-derive inside the sub-step
    asm volatile("" : "+v"(kc.lim_k), "+v"(kc.lim_b));
    const DppQuad dpp;

    float std_[Env::ACT];
#pragma unroll
    for (int k = 0; k < Env::ACT; ++k) std_[k] = __expf(fmaxf(pol.log_std(k), a.log_min_std));

    float s[Env::STATE];
    load_state<Env>(a.state, n, i, s);
    int ts = a.ts[i];
    const size_t draws_slice = (size_t)Env::RESET_DRAWS * n;
    if (a.reset_at_start) {
        reset_one<Env>(s, a.reset_draws, n, i, a.seed, env_global, a.step_counter, a.cfg);
        ts = 0;
    }
    float o[Env::OBS];
    float zq[Env::ACT] = {};      // policy noise of four steps, one step per replica group (lane-group shapes)
    Env::template observe<float>(s, o);
    const size_t obs_z_slice = (size_t)Env::OBS * n;
    observed<Env>(o, a.cfg, a.obs_noise_z, n, i, a.seed, env_global, a.step_counter);

    // Output addressing: one uniform base per array, advanced by a row (n floats) per env-step, + a loop-invariant
    // 32-bit byte offset per lane and plane in a VGPR (the dispatcher guarantees OBS * T * n * 4 < 2^32) -- a store is
    // one instruction; the plane bases as 64-bit scalars would not fit the scalar register file next to the argument
    // block and came back as ~45 v_readlane + 13 address adds per env-step.
    uint32_t vo_obs[Env::OBS], vo_act[Env::ACT];
#pragma unroll
    for (int k = 0; k < Env::OBS; ++k) {
        vo_obs[k] = (uint32_t)(((size_t)k * plane + (size_t)i) * 4);
        asm volatile("" : "+v"(vo_obs[k]));          // a value in a register, not an expression to re-derive per store
    }
#pragma unroll
    for (int k = 0; k < Env::ACT; ++k) {
        vo_act[k] = (uint32_t)(((size_t)k * plane + (size_t)i) * 4);
        asm volatile("" : "+v"(vo_act[k]));
    }
    uint32_t vo_row = (uint32_t)i * 4, vo_done = (uint32_t)i;
    asm volatile("" : "+v"(vo_row), "+v"(vo_done));
    // (the in-place asm keeps the zero-extension of the offset next to the add, which is what the back-end needs to
    //  select the scalar-base + 32-bit-offset form of global_store; it emits no instruction)
    auto at = [](auto* base, size_t row_elems, uint32_t& byte_off) {
        using P = decltype(base);
        asm volatile("" : "+v"(byte_off));
        return reinterpret_cast<P>(reinterpret_cast<char*>(base + row_elems) + byte_off);
    };

    // lane-group state of the env, resident across env-steps (valid until a reset touches the wavefront)
    Chain::Lane<float> ls;
    PlanarKin<float, 3> kin;
    float sn_b = 0.0f, cs_b = 1.0f;
    bool chain_valid = false;

    for (int t = 0; t < T; ++t) {
        const size_t off = (size_t)t * n + i;
        const size_t row = (size_t)t * n;            // uniform
        if (live) {
#pragma unroll
            for (int k = 0; k < Env::OBS; ++k) *at(a.obs, row, vo_obs[k]) = o[k];
        }
        float mean[Env::ACT], act[Env::ACT], z[Env::ACT];
        // (the noise first: its lane moves travel while the policy runs)
        if (a.eps) {
#pragma unroll
            for (int k = 0; k < Env::ACT; ++k) z[k] = a.eps[k * plane + off];
            landed<Env::ACT>(z);
        } else {
            // four steps' noise at once, one step per replica group (see rollout_kernel)
            if ((t & 3) == 0)
                philox_draws<Env::ACT, true>(zq, a.seed, env_global, a.step_counter + (uint64_t)(t + (lane >> 4)),
                                             RNG_POLICY);
#pragma unroll
            for (int k = 0; k < Env::ACT; ++k) z[k] = __shfl(zq[k], el + 16 * (t & 3), 64);
        }
        pol.forward16(o, mean);
#pragma unroll
        for (int k = 0; k < Env::ACT; ++k) {
            act[k] = __builtin_fmaf(z[k], std_[k], mean[k]);  // rnd * exp(log_std) + mean
            if (live) {
                *at(a.actions, row, vo_act[k]) = act[k];
                *at(a.means, row, vo_act[k]) = mean[k];
            }
        }

        // ---- Env.step: begin (env per lane) -> 50 sub-steps (lane group per env) -> end (env per lane) ----
        float eact[2], ctrl[3];
        if (a.cfg.action_noise != 0.0f) {      // wave-uniform: MujocoEnv(action_noise=..) (mujoco_env.py:175-187)
            float zn[Env::ACT], dact[Env::ACT];
            noise_draws<Env::ACT>(zn, a.act_noise_z ? a.act_noise_z + (size_t)t * Env::ACT * n : nullptr, n, i, a.seed,
                                  env_global, a.step_counter + (uint64_t)t, RNG_ACT_NOISE);
            action_perturbation<Env, float>(a.cfg, zn, dact);
            Env::template step_begin<float>(act, a.normalize, eact, ctrl, dact);
        } else {
            Env::template step_begin<float>(act, a.normalize, eact, ctrl);
        }
        {
            const float c1 = __shfl(ctrl[1], q_src, 64), c2 = __shfl(ctrl[2], q_src, 64);
            const float lact = (b == 1) ? c1 : (b == 2) ? c2 : 0.0f;
            if (!chain_valid) {
                // hand the env of lane q_src to its quad: each lane derives the variables of ITS body with the same
                // expressions as Swimmer::to_chain (first step, and after a reset anywhere in the wavefront)
                float g[10];
#pragma unroll
                for (int k = 0; k < 10; ++k) g[k] = __shfl(s[k], q_src, 64);
                ls.r = V2<float>{g[0], g[1]};
                ls.v = V2<float>{g[5], g[6]};
                const float phi1 = g[2] + g[3], om1 = g[7] + g[8];
                const float phi2 = phi1 + g[4], om2 = om1 + g[9];
                const float phi = (b == 0) ? g[2] : (b == 1) ? phi1 : (b == 2) ? phi2 : 0.0f;
                ls.om = (b == 0) ? g[7] : (b == 1) ? om1 : (b == 2) ? om2 : 0.0f;
                ls.th = (b == 0) ? g[2] : (b == 1) ? g[3] : (b == 2) ? g[4] : 0.0f;
                rl_sincos(phi, sn_b, cs_b);
            } else {
                // the quad still holds the env from the previous step: root translation and joint angles as they are,
                // absolute rates re-derived from the joint rates exactly as to_chain(from_chain(.)) does
                // (om_1 = qd_0 + qd_1, om_2 = om_1 + qd_2), (sin, cos) = the pair evaluated after the last sub-step
                ls.om = prefix3(dpp, ls.qd, b);
                ls.th = (b < 3) ? ls.th : 0.0f;
            }
            ls.set_direction(cs_b, sn_b);
            ls.qd = Chain::joint_rate(dpp, ls.om);
#pragma unroll RL_SWIMMER_SUBSTEP_UNROLL
            for (int it = 0; it < Env::FRAME_SKIP; ++it)
                Chain::template substep_quad<float>(dpp, kc, ls, lact, 0.001f);
            // exact sines of the new absolute angles, one body per lane: Swimmer::step_end's centre of mass needs
            // them now (PlanarTree::angles: phi_1 = th_0 + th_1, phi_2 = phi_1 + th_2), the next step's sub-steps
            // start from them
            rl_sincos(prefix3(dpp, ls.th, b), sn_b, cs_b);
            chain_valid = true;
            // back: qpos / qvel as Swimmer::from_chain forms them (joint rate = own absolute rate - parent's,
            // carried by the lane program; role 3 keeps om = 0)
            const int base = 4 * el;
            s[0] = __shfl(ls.r.x, base, 64);
            s[1] = __shfl(ls.r.y, base, 64);
            s[5] = __shfl(ls.v.x, base, 64);
            s[6] = __shfl(ls.v.y, base, 64);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                s[2 + j] = __shfl(ls.th, base + j, 64);
                s[7 + j] = __shfl(ls.qd, base + j, 64);
                kin.sn[j] = __shfl(sn_b, base + j, 64);
                kin.cs[j] = __shfl(cs_b, base + j, 64);
            }
        }
        float r;
        bool d;
        Env::template step_end_sc<float>(s, eact, kin, o, r, d, a.cfg.ctrl_cost_coeff);

        ts += 1;
        if (a.max_path_length > 0 && ts >= a.max_path_length) d = true;
        if (live) {
            *at(a.rewards, row, vo_row) = r * a.scale_reward;
            *at(a.dones, row, vo_done) = d ? 1 : 0;
        }
        if (d) {
            const float* dr = a.reset_draws ? a.reset_draws + (size_t)(t + 1) * draws_slice : nullptr;
            reset_one<Env>(s, dr, n, i, a.seed, env_global, a.step_counter + (uint64_t)t + 1, a.cfg);
            Env::template observe<float>(s, o);
            ts = 0;
        }
        if (__builtin_amdgcn_ballot_w64(d) != 0) chain_valid = false;     // some env of this wavefront starts afresh
        observed<Env>(o, a.cfg, a.obs_noise_z ? a.obs_noise_z + (size_t)(t + 1) * obs_z_slice : nullptr, n, i, a.seed,
                      env_global, a.step_counter + (uint64_t)t + 1);
    }
    if (live) {
        store_state<Env>(a.state, n, i, s);
        a.ts[i] = ts;
        if (a.last_obs) {
#pragma unroll
            for (int k = 0; k < Env::OBS; ++k) a.last_obs[(size_t)k * n + i] = o[k];
        }
    }
}

template <int H, bool ACTS = false>
__global__ void __launch_bounds__(LANE_TPB) rollout_swimmer_quad_kernel(RolloutDev a) {
    RolloutPolicy16<Swimmer, H, ACTS> pol;
    pol.init(a.theta);
    pol.act0 = a.act0; pol.act1 = a.act1;
    swimmer_quad_body(a, pol);
}

// the lane-group Swimmer under a wide / deep policy (RolloutPolicyWide::forward16: weight fragments in LDS)
__global__ void __launch_bounds__(LANE_TPB) rollout_swimmer_quad_wide_kernel(RolloutDev a, WideShape shape) {
    extern __shared__ __attribute__((aligned(16))) float wide_smem[];
    RolloutPolicyWide<Swimmer> pol;
    pol.init(wide_smem, a.theta, shape);
    swimmer_quad_body(a, pol);
}

// ... with the network split over the four wavefronts of a workgroup (RolloutPolicyCoop)
__global__ void __launch_bounds__(64 * COOP_WAVES) rollout_swimmer_quad_coop_kernel(RolloutDev a, WideShape shape) {
    extern __shared__ __attribute__((aligned(16))) float wide_smem[];
    RolloutPolicyCoop<Swimmer> pol;
    pol.init(wide_smem, a.theta, shape);
    swimmer_quad_body<RolloutPolicyCoop<Swimmer>, true>(a, pol);
}

// ---------------------------------------------------------------------------
// Lane-group rollout of the two-legged envs (HalfCheetah, Walker2D): 16 envs per wavefront, everything per
// env-step env-per-lane on four replicas exactly as in rollout_kernel, and the physics sub-steps ONE BODY PER LANE of
// the env's quad (dyn_two_legs.h, V = V2<float>: lane 4e + r of env e's quad holds role r = torso / hip link / shin /
// foot of BOTH legs side by side in two-component values; role moves are quad-permute DPP per component, the other leg
// is the swapped pair).  The bodies' state stays in the lanes across env-steps; each step hands the six motor torques in
// and the new state and the centre of mass back out.
// ---------------------------------------------------------------------------
// lane moves of dyn_two_legs.h inside a quad: lane i of the quad reads lane P[i]
struct TwoLegQuadMoves {
    // quad_perm [a, b, c, d] = a | b << 2 | c << 4 | d << 6
    enum : int {
        UP = 0 | 0 << 2 | 1 << 4 | 2 << 6,        // parent role (role 0: itself)
        DOWN = 1 | 2 << 2 | 3 << 4 | 3 << 6,      // child role (role 3: itself)
        NXT = 0 | 2 << 2 | 3 << 4 | 1 << 6,       // 1 -> 2 -> 3 -> 1
        PRV = 0 | 3 << 2 | 1 << 4 | 2 << 6,
        ROOT = 0,                                 // [0, 0, 0, 0]
        FIRST = 1 | 1 << 2 | 1 << 4 | 1 << 6,
        OTHER_ROW_ROR8 = 0x128
    };
    template <int CTRL> __device__ __forceinline__ static float mv(float v) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
    }
};
// V = float: one body per lane, the other leg eight lanes further in the row of 16 (row_ror:8)
struct DppBodyLanes : TwoLegQuadMoves {
    bool is_root, is_leaf;
    __device__ __forceinline__ float up(float v) const { return mv<UP>(v); }
    __device__ __forceinline__ float down(float v) const { return mv<DOWN>(v); }
    __device__ __forceinline__ float nxt(float v) const { return mv<NXT>(v); }
    __device__ __forceinline__ float prv(float v) const { return mv<PRV>(v); }
    __device__ __forceinline__ float root(float v) const { return mv<ROOT>(v); }
    __device__ __forceinline__ float first(float v) const { return mv<FIRST>(v); }
    __device__ __forceinline__ float other(float v) const { return mv<OTHER_ROW_ROR8>(v); }
    __device__ __forceinline__ float sel_root(float a, float b) const { return is_root ? a : b; }
    __device__ __forceinline__ float sel_leaf(float a, float b) const { return is_leaf ? a : b; }
};
// V = V2<float>: both legs in the two components, the roles on the lanes of a quad
struct DppBodyPairs : TwoLegQuadMoves {
    using P = V2<float>;
    bool is_root, is_leaf;
    __device__ __forceinline__ P up(P v) const { return P{mv<UP>(v.x), mv<UP>(v.y)}; }
    __device__ __forceinline__ P down(P v) const { return P{mv<DOWN>(v.x), mv<DOWN>(v.y)}; }
    __device__ __forceinline__ P nxt(P v) const { return P{mv<NXT>(v.x), mv<NXT>(v.y)}; }
    __device__ __forceinline__ P prv(P v) const { return P{mv<PRV>(v.x), mv<PRV>(v.y)}; }
    __device__ __forceinline__ P root(P v) const { return P{mv<ROOT>(v.x), mv<ROOT>(v.y)}; }
    __device__ __forceinline__ P first(P v) const { return P{mv<FIRST>(v.x), mv<FIRST>(v.y)}; }
    __device__ __forceinline__ P other(P v) const { return v.yx; }
    __device__ __forceinline__ P sel_root(P a, P b) const { return is_root ? a : b; }
    __device__ __forceinline__ P sel_leaf(P a, P b) const { return is_leaf ? a : b; }
};

template <class Env, class Pol>
__device__ __forceinline__ void two_leg_quad_body(const RolloutDev& a, const Pol& pol) {
    using Legs = typename Env::Legs;
    using Tree = typename Env::Tree;

    const int n = a.n, T = a.T;
    const int lane = threadIdx.x & 63;
    const int wave_global = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int el = lane & (QUAD_ENVS - 1);            // env slot of this lane in the env-per-lane phases
    const int i_raw = wave_global * QUAD_ENVS + el;
    const bool live = (i_raw < n) && (lane < QUAD_ENVS);   // one of the four copies stores
    const int i = (i_raw < n) ? i_raw : n - 1;
    const uint32_t env_global = (uint32_t)(a.env_offset + i);
    const size_t plane = (size_t)T * n;
    // lane-group phase: quad lane / 4 works on the env held by lane q_src, this lane on role (lane & 3) of both legs
    using P2 = V2<float>;
    const int q_src = lane >> 2, role = lane & 3;
    const typename Legs::template LaneK<P2> kc = Legs::template role_constants<float>(role);
    const DppBodyPairs dpp{{}, role == 0, role == 3};

    float std_[Env::ACT];
#pragma unroll
    for (int k = 0; k < Env::ACT; ++k) std_[k] = __expf(fmaxf(pol.log_std(k), a.log_min_std));

    float s[Env::STATE];
    load_state<Env>(a.state, n, i, s);
    int ts = a.ts[i];
    const size_t draws_slice = (size_t)Env::RESET_DRAWS * n;
    if (a.reset_at_start) {
        reset_one<Env>(s, a.reset_draws, n, i, a.seed, env_global, a.step_counter, a.cfg);
        ts = 0;
    }
    float o[Env::OBS];
    float zq[Env::ACT] = {};      // policy noise of four steps, one step per replica group (lane-group shapes)
    Env::template observe<float>(s, o);
    const size_t obs_z_slice = (size_t)Env::OBS * n;
    observed<Env>(o, a.cfg, a.obs_noise_z, n, i, a.seed, env_global, a.step_counter);
    // output addressing as in rollout_swimmer_quad_kernel: uniform base per array + 32-bit byte offset per lane and plane
    uint32_t vo_obs[Env::OBS], vo_act[Env::ACT];
#pragma unroll
    for (int k = 0; k < Env::OBS; ++k) {
        vo_obs[k] = (uint32_t)(((size_t)k * plane + (size_t)i) * 4);
        asm volatile("" : "+v"(vo_obs[k]));
    }
#pragma unroll
    for (int k = 0; k < Env::ACT; ++k) {
        vo_act[k] = (uint32_t)(((size_t)k * plane + (size_t)i) * 4);
        asm volatile("" : "+v"(vo_act[k]));
    }
    uint32_t vo_row = (uint32_t)i * 4, vo_done = (uint32_t)i;
    asm volatile("" : "+v"(vo_row), "+v"(vo_done));
    auto at = [](auto* base, size_t row_elems, uint32_t& byte_off) {
        using P = decltype(base);
        asm volatile("" : "+v"(byte_off));
        return reinterpret_cast<P>(reinterpret_cast<char*>(base + row_elems) + byte_off);
    };

    typename Legs::template State<P2> ls;           // resident across env-steps (valid until a reset touches the wavefront)
    bool chain_valid = false;
    const StepOpts<float> base_opts = opts_from_cfg<float>(a.cfg);

    for (int t = 0; t < T; ++t) {
        const size_t off = (size_t)t * n + i;
        const size_t row = (size_t)t * n;
        if (live) {
#pragma unroll
            for (int k = 0; k < Env::OBS; ++k) *at(a.obs, row, vo_obs[k]) = o[k];
        }
        float mean[Env::ACT], act[Env::ACT], z[Env::ACT];
        // (the noise first: its lane moves travel while the policy runs)
        if (a.eps) {
#pragma unroll
            for (int k = 0; k < Env::ACT; ++k) z[k] = a.eps[k * plane + off];
            landed<Env::ACT>(z);
        } else {
            // four steps' noise at once, one step per replica group (see rollout_kernel)
            if ((t & 3) == 0)
                philox_draws<Env::ACT, true>(zq, a.seed, env_global, a.step_counter + (uint64_t)(t + (lane >> 4)),
                                             RNG_POLICY);
#pragma unroll
            for (int k = 0; k < Env::ACT; ++k) z[k] = __shfl(zq[k], el + 16 * (t & 3), 64);
        }
        pol.forward16(o, mean);
#pragma unroll
        for (int k = 0; k < Env::ACT; ++k) act[k] = __builtin_fmaf(z[k], std_[k], mean[k]);  // rnd * exp(log_std) + mean
        if (live) {
#pragma unroll
            for (int k = 0; k < Env::ACT; ++k) {
                *at(a.actions, row, vo_act[k]) = act[k];
                *at(a.means, row, vo_act[k]) = mean[k];
            }
        }

        // ---- Env.step: begin (env per lane) -> sub-steps (one body per lane) -> end (env per lane) ----
        float eact[Env::ACT_BUF], tau[7];
        StepOpts<float> opts = base_opts;
        float dact[Env::ACT];
        if (a.cfg.action_noise != 0.0f) {      // wave-uniform: MujocoEnv(action_noise=..) (mujoco_env.py:175-187)
            float zn[Env::ACT];
            noise_draws<Env::ACT>(zn, a.act_noise_z ? a.act_noise_z + (size_t)t * Env::ACT * n : nullptr, n, i, a.seed,
                                  env_global, a.step_counter + (uint64_t)t, RNG_ACT_NOISE);
            action_perturbation<Env, float>(a.cfg, zn, dact);
            opts.dact = dact;
        }
        Env::template step_begin<float>(act, a.normalize, opts, eact, tau);
        float com4[4];
        {
            // the own hinge's motor torque, both legs (role 0: none)
            P2 lact = P2{0.0f, 0.0f};
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float tb = __shfl(tau[1 + j], q_src, 64), tf = __shfl(tau[4 + j], q_src, 64);
                lact = (role == 1 + j) ? P2{tb, tf} : lact;
            }
            if (!chain_valid) {
                // hand the env of lane q_src to its quad (first step, and after a reset anywhere in the wavefront)
                const float p1 = __shfl(s[0], q_src, 64), p2 = __shfl(s[1], q_src, 64);
                const float v1 = __shfl(s[9], q_src, 64), v2 = __shfl(s[10], q_src, 64);
                ls.p1 = P2{p1, p1}; ls.p2 = P2{p2, p2}; ls.v1 = P2{v1, v1}; ls.v2 = P2{v2, v2};
                const float q0 = __shfl(s[2], q_src, 64), w0 = __shfl(s[11], q_src, 64);
                ls.q = P2{q0, q0};
                ls.w = P2{w0, w0};
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const float qb = __shfl(s[3 + j], q_src, 64), qf = __shfl(s[6 + j], q_src, 64);
                    const float vb = __shfl(s[12 + j], q_src, 64), vf = __shfl(s[15 + j], q_src, 64);
                    ls.q = (role == 1 + j) ? P2{qb, qf} : ls.q;
                    ls.w = (role == 1 + j) ? P2{vb, vf} : ls.w;
                }
                // absolute rates and the exact sines of the absolute angles (PlanarTree::angles: phi_child = phi_parent +
                // hinge) -- otherwise they are the ones evaluated after the previous step's sub-steps
                Legs::template abs_rates<float, P2, DppBodyPairs>(dpp, ls);
                Legs::template exact_directions<float, P2, DppBodyPairs>(dpp, ls);
            }
#pragma unroll 1
            for (int it = 0; it < Env::SUBSTEPS; ++it)
                Legs::template substep<float, P2, DppBodyPairs>(dpp, kc, ls, lact, 0.0025f);
            // the sines of the new angles: the centre of mass of step_end needs them now (TwoLegs::com), the next step's
            // sub-steps start from them
            Legs::template exact_directions<float, P2, DppBodyPairs>(dpp, ls);
            chain_valid = true;
            P2 lc[4];
            Legs::template com<float, P2, DppBodyPairs>(dpp, kc, ls, lc[0], lc[1], lc[2], lc[3]);
            // back to the env-per-lane copies: lane 4 el + r holds role r of both legs
            const int base = 4 * el;
            s[0] = __shfl(ls.p1.x, base, 64); s[1] = __shfl(ls.p2.x, base, 64);
            s[9] = __shfl(ls.v1.x, base, 64); s[10] = __shfl(ls.v2.x, base, 64);
            s[2] = __shfl(ls.q.x, base, 64); s[11] = __shfl(ls.w.x, base, 64);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                s[3 + j] = __shfl(ls.q.x, base + 1 + j, 64);
                s[6 + j] = __shfl(ls.q.y, base + 1 + j, 64);
                s[12 + j] = __shfl(ls.w.x, base + 1 + j, 64);
                s[15 + j] = __shfl(ls.w.y, base + 1 + j, 64);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) com4[k] = __shfl(lc[k].x, base, 64);
        }
        float r;
        bool d;
        Env::template step_end_com<float>(s, eact, com4[0], com4[1], com4[2], com4[3], o, r, d, opts);

        ts += 1;
        if (a.max_path_length > 0 && ts >= a.max_path_length) d = true;
        if (live) {
            *at(a.rewards, row, vo_row) = r * a.scale_reward;
            *at(a.dones, row, vo_done) = d ? 1 : 0;
        }
        if (d) {
            const float* dr = a.reset_draws ? a.reset_draws + (size_t)(t + 1) * draws_slice : nullptr;
            reset_one<Env>(s, dr, n, i, a.seed, env_global, a.step_counter + (uint64_t)t + 1, a.cfg);
            Env::template observe<float>(s, o);
            ts = 0;
        }
        if (__builtin_amdgcn_ballot_w64(d) != 0) chain_valid = false;     // some env of this wavefront starts afresh
        observed<Env>(o, a.cfg, a.obs_noise_z ? a.obs_noise_z + (size_t)(t + 1) * obs_z_slice : nullptr, n, i, a.seed,
                      env_global, a.step_counter + (uint64_t)t + 1);
    }
    if (live) {
        store_state<Env>(a.state, n, i, s);
        a.ts[i] = ts;
        if (a.last_obs) {
#pragma unroll
            for (int k = 0; k < Env::OBS; ++k) a.last_obs[(size_t)k * n + i] = o[k];
        }
    }
}

template <class Env, int H, bool ACTS = false>
__global__ void __launch_bounds__(LANE_TPB) rollout_two_leg_quad_kernel(RolloutDev a) {
    RolloutPolicy16<Env, H, ACTS> pol;
    pol.init(a.theta);
    pol.act0 = a.act0; pol.act1 = a.act1;
    two_leg_quad_body<Env>(a, pol);
}

template <class Env>
__global__ void __launch_bounds__(LANE_TPB) rollout_two_leg_quad_wide_kernel(RolloutDev a, WideShape shape) {
    extern __shared__ __attribute__((aligned(16))) float wide_smem[];
    RolloutPolicyWide<Env> pol;
    pol.init(wide_smem, a.theta, shape);
    two_leg_quad_body<Env>(a, pol);
}

// ---------------------------------------------------------------------------
// ONE ENV PER WAVEFRONT (HalfCheetah, Walker2D at small env counts: C5's per-GPU shard is 1024 envs, i.e. 64
// wavefronts of 16 envs on 1024 SIMDs).  A lone wavefront pays per issued instruction whatever its lanes do, so the
// 16-env shape spends its env-step on work that is per-ENV in one instruction stream (profiles/r04_notes.md: 3.2 k
// vector + 88 matrix instructions per step).  With a wavefront per env the chip's idle SIMDs take the envs and the
// lanes take the policy's UNITS:
//   * policy (RolloutPolicyLane): lane u holds column u of W0 / W1 (unit u of both hidden layers), lane a column a of the
//     output layer; a layer is H fused multiply-adds per lane on the previous layer's activations, which every lane reads
//     back from a 256-byte LDS row as broadcast 16-byte reads.  No matrix instruction (a matrix-vector product wastes
//     its columns), ~130 instructions per step instead of 88 matrix + ~370 vector;
//   * everything per env (state, observation, action map, reward, reset) is computed by all lanes alike (wave-uniform
//     values); the trajectory goes out lane-distributed: lane k stores observation row k, lane a action / mean row a --
//     one store instruction per array;
//   * policy noise: lane j draws the Philox block of step t + j every 64 steps (the same (seed; env, step, POLICY)
//     blocks as every other shape), a step reads its row with v_readlane;
//   * physics: dyn_two_legs.h with ONE BODY PER LANE (V = float): lane (leg = bit 3, role = bits 0..1) holds the torso /
//     hip link / shin / foot of its leg, role moves are quad-permute DPP, the other leg is row_ror:8 -- every exchange
//     fused into the add / multiply that consumes it; 312 vector instructions per sub-step against the 570 of the
//     former one-leg-per-lane chain walk (round 5).  State resident across env-steps, hand-over to the per-env arithmetic
//     by v_readlane; one rl_sincos per lane and env-step.
// Dynamics are bit-identical to every other shape (same leaf functions; replayed against the host build in the parity
// tests); the policy's means agree with the other shapes to rounding (a different summation order), as between any two
// of them.
// ---------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(2))) float rl_f32x2;
typedef __attribute__((ext_vector_type(4))) float rl_f32x4;

__device__ __forceinline__ float lane_bcast(float v, int src_lane) {          // src_lane wave-uniform
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src_lane));
}

// Sums over the 64 lanes of N independent values at once, each left wave-uniform: four rotations inside the rows of 16
// (every lane of a row then holds the row's sum), then the row sums travel up (row_bcast:15 into rows 1 and 3,
// row_bcast:31 into rows 2 and 3) and lane 63 is read -- seven issue slots per value with the lane move fused into the
// add, no LDS.  The row_ror stages are builtins (the hazard recogniser spaces them); the two row_bcast stages are
// inline asm, which it does not look into: each is one block that carries its own wait states (see below).
template <int CTRL>
__device__ __forceinline__ float dpp_ror_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int N>
__device__ __forceinline__ void wave_totals_uniform(float* v) {
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = dpp_ror_add<0x128>(v[k]);      // row_ror:8
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = dpp_ror_add<0x124>(v[k]);      // row_ror:4
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = dpp_ror_add<0x122>(v[k]);      // row_ror:2
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = dpp_ror_add<0x121>(v[k]);      // row_ror:1
    // rows outside the row mask keep their value: the fused form of  v + (row enabled ? moved : 0).  A DPP source
    // written by the previous vector instruction needs two wait states, and the hazard recogniser does not look inside
    // inline asm: each stage is ONE asm block that opens with its own s_nop 1, so its first add is safe whatever the
    // scheduler placed in front of the block, and the later adds of a block read registers written before the block
    // (or, in the second stage, at least N - 1 + 2 slots earlier).  Two issue slots per stage, independent of
    // compiler version and scheduling flags.
#define RL_BCAST_STAGE(CTL)                                                                                              \
    if constexpr (N == 1) asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 " CTL : "+v"(v[0]));                          \
    else if constexpr (N == 2) asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 " CTL "\n\tv_add_f32_dpp %1, %1, %1 " CTL \
                                            : "+v"(v[0]), "+v"(v[1]));                                                   \
    else if constexpr (N == 3) asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 " CTL "\n\tv_add_f32_dpp %1, %1, %1 " CTL \
                                            "\n\tv_add_f32_dpp %2, %2, %2 " CTL : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]));   \
    else if constexpr (N == 6) asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 " CTL "\n\tv_add_f32_dpp %1, %1, %1 " CTL \
                                            "\n\tv_add_f32_dpp %2, %2, %2 " CTL "\n\tv_add_f32_dpp %3, %3, %3 " CTL       \
                                            "\n\tv_add_f32_dpp %4, %4, %4 " CTL "\n\tv_add_f32_dpp %5, %5, %5 " CTL       \
                                            : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5])); \
    else {                                                                                                               \
        _Pragma("unroll") for (int k = 0; k < N; ++k)                                                                    \
            asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 " CTL : "+v"(v[k]));                                        \
    }
    RL_BCAST_STAGE("row_bcast:15 row_mask:0xa")
    RL_BCAST_STAGE("row_bcast:31 row_mask:0xc")
#undef RL_BCAST_STAGE
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = lane_bcast(v[k], 63);
}

template <class Env, int H, bool ACTS = false>
struct RolloutPolicyLane {
    using N = Net<Env::OBS, Env::ACT, H>;
    static constexpr int DO = Env::OBS, DA = Env::ACT;
    static constexpr int DOP = (DO + 1) & ~1;
    float w0[DOP], b0;         // column u of W0 (u = lane % H), b0[u]
    float w1[H], b1;           // column u of W1, b1[u]
    float w2[DA], b2;          // row u of W2 (zero on the lanes that repeat a unit: H < 64), b2[a] (a = min(lane, DA - 1))
    float lstd;                // log_std[a]
    float* hbuf;               // LDS: H floats of this wavefront
    int act0 = 0, act1 = 0;    // hidden activations (rl_activation codes, wave-uniform)

    __device__ __forceinline__ void init(const float* __restrict__ th, float* lds_row) {
        const int lane = threadIdx.x & 63, u = lane & (H - 1), a = lane < DA ? lane : DA - 1;
        hbuf = lds_row;
#pragma unroll
        for (int k = 0; k < DOP; ++k) w0[k] = k < DO ? th[N::W0 + k * H + u] : 0.0f;
        b0 = th[N::B0 + u];
#pragma unroll
        for (int k = 0; k < H; ++k) w1[k] = th[N::W1 + k * H + u];
#pragma unroll
        for (int k = 0; k < DA; ++k) w2[k] = lane < H ? th[N::W2 + u * DA + k] : 0.0f;
        b1 = th[N::B1 + u];
        b2 = th[N::B2 + a];
        lstd = th[N::LSTD + a];
    }
    // sum_k w[k] v[k] with v read back from the wavefront's LDS row (every lane the same addresses: broadcast reads),
    // two products per packed instruction.  All reads are issued before the first product waits for one (the wavefront
    // is alone on its SIMD: a read issued behind a product would expose a full LDS round trip each).
    __device__ __forceinline__ float dot_row(const float* w) const {
        const rl_f32x4* hv = reinterpret_cast<const rl_f32x4*>(hbuf);
        rl_f32x4 v[H / 4];
#pragma unroll
        for (int q = 0; q < H / 4; ++q) v[q] = hv[q];
#pragma unroll
        for (int q = 0; q < H / 4; ++q) asm volatile("" : "+v"(v[q]));
        rl_f32x2 acc0 = {0.0f, 0.0f}, acc1 = {0.0f, 0.0f};
#pragma unroll
        for (int q = 0; q < H / 4; ++q) {
            acc0 = __builtin_elementwise_fma((rl_f32x2){w[4 * q], w[4 * q + 1]}, (rl_f32x2){v[q][0], v[q][1]}, acc0);
            acc1 = __builtin_elementwise_fma((rl_f32x2){w[4 * q + 2], w[4 * q + 3]}, (rl_f32x2){v[q][2], v[q][3]}, acc1);
        }
        const rl_f32x2 acc = acc0 + acc1;
        return acc[0] + acc[1];
    }
    // o: the env's observation (the same on every lane); returns mean[a] on lane a < DA
    __device__ __forceinline__ float forward(const float* o) const {
        const int lane = threadIdx.x & 63;
        rl_f32x2 acc = {b0, 0.0f};
#pragma unroll
        for (int k = 0; k < DOP; k += 2)
            acc = __builtin_elementwise_fma((rl_f32x2){w0[k], w0[k + 1]}, (rl_f32x2){o[k], k + 1 < DO ? o[k + 1] : 0.0f}, acc);
        const float h0 = ACTS ? act_one(acc[0] + acc[1], act0) : ftanh(acc[0] + acc[1]);
        if (lane < H) hbuf[lane] = h0;
        wave_sync();
        const float h1 = ACTS ? act_one(b1 + dot_row(w1), act1) : ftanh(b1 + dot_row(w1));
        wave_sync();                                  // (the reads of h0 are done before the next step's write)
        // output layer: unit u's contribution to every action from its own lane, summed over the lanes
        float p[DA];
#pragma unroll
        for (int k = 0; k < DA; ++k) p[k] = h1 * w2[k];
        wave_totals_uniform<DA>(p);
        float m = p[0];
#pragma unroll
        for (int k = 1; k < DA; ++k) m = (lane == k) ? p[k] : m;
        return b2 + m;
    }
};

// lane k <- v[k] (k < N; every v[k] wave-uniform): a binary select tree over the bits of the lane number -- five masks
// for any N <= 32, where a chain of N selects keeps N masks alive (they spill out of the scalar file)
template <int N>
__device__ __forceinline__ float lane_pick(const float* v, int lane) {
    static_assert(N <= 32, "five lane bits");
    const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0, b2 = (lane & 4) != 0, b3 = (lane & 8) != 0, b4 = (lane & 16) != 0;
    // (both candidates of a select are read into values first: "bit ? a[1] : a[0]" would come back as ONE load at a
    //  per-lane index and send the whole array to scratch memory)
    const float* t = v;
    // level L entry j stands for the entries j 2^(L+1) .. of v; an upper half that starts past N is never a lane's pick
    float l0[16], l1[8], l2[4], l3[2];
#define RL_PICK_LEVEL(OUT, IN, CNT, BIT, SPAN)                                                      \
    static_for<0, CNT>([&](auto J) {                                                              \
        constexpr int j = decltype(J)::value;                                                     \
        if constexpr ((2 * j + 1) * SPAN < N) {                                                   \
            const float lo_ = IN[2 * j], hi_ = IN[2 * j + 1];      /* both read before the select */ \
            OUT[j] = BIT ? hi_ : lo_;                                                             \
        } else if constexpr (2 * j * SPAN < N) OUT[j] = IN[2 * j];                                \
        else OUT[j] = 0.0f;                                                                       \
    });
    RL_PICK_LEVEL(l0, t, 16, b0, 1)
    RL_PICK_LEVEL(l1, l0, 8, b1, 2)
    RL_PICK_LEVEL(l2, l1, 4, b2, 4)
    RL_PICK_LEVEL(l3, l2, 2, b3, 8)
#undef RL_PICK_LEVEL
    if constexpr (16 < N) return b4 ? l3[1] : l3[0];
    else return l3[0];
}

template <class Env, int H, bool ACTS = false>
__global__ void __launch_bounds__(LANE_TPB) rollout_two_leg_wave_kernel(RolloutDev a) {
    using Legs = typename Env::Legs;
    __shared__ __attribute__((aligned(16))) float hrows[LANE_TPB / 64][H];
    RolloutPolicyLane<Env, H, ACTS> pol;
    pol.init(a.theta, hrows[threadIdx.x >> 6]);
    pol.act0 = a.act0; pol.act1 = a.act1;

    const int n = a.n, T = a.T;
    const int lane = threadIdx.x & 63;
    const int i_raw = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));   // this wavefront's env
    const bool alive = i_raw < n;
    const int i = alive ? i_raw : n - 1;
    const uint32_t env_global = (uint32_t)(a.env_offset + i);
    const size_t plane = (size_t)T * n;
    // physics: lane = body (leg = bit 3 of the lane, role = its two low bits; the second quad of each leg and the other
    // three rows repeat the first)
    const int role = lane & 3, leg = (lane >> 3) & 1;
    const typename Legs::template LaneK<float> kc = Legs::template lane_constants<float>(leg, role);
    const DppBodyLanes dpp{{}, role == 0, role == 3};
    const bool obs_lane = alive && lane < Env::OBS, act_lane = alive && lane < Env::ACT, lane0 = alive && lane == 0;

    const float std_l = __expf(fmaxf(pol.lstd, a.log_min_std));          // lane a: exp(log_std[a])

    float s[Env::STATE];
    load_state<Env>(a.state, n, i, s);
    int ts = a.ts[i];
    const size_t draws_slice = (size_t)Env::RESET_DRAWS * n;
    if (a.reset_at_start) {
        reset_one<Env>(s, a.reset_draws, n, i, a.seed, env_global, a.step_counter, a.cfg);
        ts = 0;
    }
    float o[Env::OBS];
    float zq[Env::ACT] = {};      // policy noise of 64 steps, one step per lane
    Env::template observe<float>(s, o);
    const size_t obs_z_slice = (size_t)Env::OBS * n;
    observed<Env>(o, a.cfg, a.obs_noise_z, n, i, a.seed, env_global, a.step_counter);
    // lane-distributed output addressing: uniform base per array advanced by a row per env-step + the 32-bit byte offset
    // of this lane's plane (the dispatcher keeps OBS * T * n * 4 < 2^32 for this kernel)
    uint32_t vo_obs = (uint32_t)(((size_t)(lane < Env::OBS ? lane : 0) * plane + (size_t)i) * 4);
    uint32_t vo_act = (uint32_t)(((size_t)(lane < Env::ACT ? lane : 0) * plane + (size_t)i) * 4);
    uint32_t vo_row = (uint32_t)i * 4, vo_done = (uint32_t)i;
    auto at = [](auto* base, size_t row_elems, uint32_t& byte_off) {
        using P = decltype(base);
        asm volatile("" : "+v"(byte_off));
        return reinterpret_cast<P>(reinterpret_cast<char*>(base + row_elems) + byte_off);
    };

    typename Legs::template State<float> ls;        // resident across env-steps (valid until the env is reset)
    bool chain_valid = false;
    const StepOpts<float> base_opts = opts_from_cfg<float>(a.cfg);

    for (int t = 0; t < T; ++t) {
        const size_t row = (size_t)t * n;
        if (obs_lane) *at(a.obs, row, vo_obs) = lane_pick<Env::OBS>(o, lane);   // observation row k from lane k
        const float mean_l = pol.forward(o);
        float zs[Env::ACT];
        if (a.eps) {
#pragma unroll
            for (int k = 0; k < Env::ACT; ++k) zs[k] = a.eps[k * plane + row + i];
            landed<Env::ACT>(zs);
        } else {
            // lane j draws the block of step t + j once per 64 steps; every step reads its row
            if ((t & 63) == 0)
                philox_draws<Env::ACT, true>(zq, a.seed, env_global, a.step_counter + (uint64_t)(t + lane), RNG_POLICY);
#pragma unroll
            for (int k = 0; k < Env::ACT; ++k) zs[k] = lane_bcast(zq[k], t & 63);
        }
        float z_l = zs[0];
#pragma unroll
        for (int k = 1; k < Env::ACT; ++k) z_l = (lane == k) ? zs[k] : z_l;
        const float act_l = __builtin_fmaf(z_l, std_l, mean_l);            // rnd * exp(log_std) + mean, action a on lane a
        if (act_lane) {
            *at(a.actions, row, vo_act) = act_l;
            *at(a.means, row, vo_act) = mean_l;
        }
        float act[Env::ACT];
#pragma unroll
        for (int k = 0; k < Env::ACT; ++k) act[k] = lane_bcast(act_l, k);

        // ---- Env.step: begin (per env, all lanes alike) -> sub-steps (one body per lane) -> end (per env) ----
        float eact[Env::ACT_BUF], tau[7];
        StepOpts<float> opts = base_opts;
        float dact[Env::ACT];
        if (a.cfg.action_noise != 0.0f) {      // wave-uniform: MujocoEnv(action_noise=..) (mujoco_env.py:175-187)
            float zn[Env::ACT];
            noise_draws<Env::ACT>(zn, a.act_noise_z ? a.act_noise_z + (size_t)t * Env::ACT * n : nullptr, n, i, a.seed,
                                  env_global, a.step_counter + (uint64_t)t, RNG_ACT_NOISE);
            action_perturbation<Env, float>(a.cfg, zn, dact);
            opts.dact = dact;
        }
        Env::template step_begin<float>(act, a.normalize, opts, eact, tau);
        float com4[4];
        {
            // the own hinge's motor torque (role 0: none)
            float lact = 0.0f;
#pragma unroll
            for (int j = 0; j < 3; ++j) lact = (role == 1 + j) ? (leg ? tau[4 + j] : tau[1 + j]) : lact;
            if (!chain_valid) {
                // hand the env to the body lanes (first step, and after a reset)
                ls.p1 = s[0]; ls.p2 = s[1]; ls.v1 = s[9]; ls.v2 = s[10];
                ls.q = s[2];
                ls.w = s[11];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    ls.q = (role == 1 + j) ? (leg ? s[6 + j] : s[3 + j]) : ls.q;
                    ls.w = (role == 1 + j) ? (leg ? s[15 + j] : s[12 + j]) : ls.w;
                }
                // absolute rates, exact sines of the absolute angles (PlanarTree::angles: phi_child = phi_parent + hinge)
                Legs::template abs_rates<float, float, DppBodyLanes>(dpp, ls);
                Legs::template exact_directions<float, float, DppBodyLanes>(dpp, ls);
            }
#pragma unroll RL_TWO_LEG_SUBSTEP_UNROLL
            for (int it = 0; it < Env::SUBSTEPS; ++it)
                Legs::template substep<float, float, DppBodyLanes>(dpp, kc, ls, lact, 0.0025f);
            // the sines of the new angles: step_end's centre of mass needs them now, the next step's sub-steps start from them
            Legs::template exact_directions<float, float, DppBodyLanes>(dpp, ls);
            chain_valid = true;
            float lc[4];
            Legs::template com<float, float, DppBodyLanes>(dpp, kc, ls, lc[0], lc[1], lc[2], lc[3]);
            // back to the per-env copy: lane r holds role r of the back leg, lane 8 + r of the front leg
            s[0] = lane_bcast(ls.p1, 0); s[1] = lane_bcast(ls.p2, 0);
            s[9] = lane_bcast(ls.v1, 0); s[10] = lane_bcast(ls.v2, 0);
            s[2] = lane_bcast(ls.q, 0); s[11] = lane_bcast(ls.w, 0);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                s[3 + j] = lane_bcast(ls.q, 1 + j);
                s[6 + j] = lane_bcast(ls.q, 9 + j);
                s[12 + j] = lane_bcast(ls.w, 1 + j);
                s[15 + j] = lane_bcast(ls.w, 9 + j);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) com4[k] = lane_bcast(lc[k], 0);
        }
        float r;
        bool d;
        Env::template step_end_com<float>(s, eact, com4[0], com4[1], com4[2], com4[3], o, r, d, opts);

        ts += 1;
        if (a.max_path_length > 0 && ts >= a.max_path_length) d = true;
        if (lane0) {
            *at(a.rewards, row, vo_row) = r * a.scale_reward;
            *at(a.dones, row, vo_done) = d ? 1 : 0;
        }
        if (d) {
            const float* dr = a.reset_draws ? a.reset_draws + (size_t)(t + 1) * draws_slice : nullptr;
            reset_one<Env>(s, dr, n, i, a.seed, env_global, a.step_counter + (uint64_t)t + 1, a.cfg);
            Env::template observe<float>(s, o);
            ts = 0;
            chain_valid = false;
        }
        observed<Env>(o, a.cfg, a.obs_noise_z ? a.obs_noise_z + (size_t)(t + 1) * obs_z_slice : nullptr, n, i, a.seed,
                      env_global, a.step_counter + (uint64_t)t + 1);
    }
    if (lane0) {
        store_state<Env>(a.state, n, i, s);
        a.ts[i] = ts;
        if (a.last_obs) {
#pragma unroll
            for (int k = 0; k < Env::OBS; ++k) a.last_obs[(size_t)k * n + i] = o[k];
        }
    }
}

__global__ void philox_debug_kernel(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                    uint32_t k1, int count, uint32_t* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    Philox4 p = philox4x32_10(c0 + (uint32_t)i, c1, c2, c3, k0, k1);
#pragma unroll
    for (int k = 0; k < 4; ++k) out[4 * i + k] = p.v[k];
}

// ---------------------------------------------------------------------------
// host-side dispatch
// ---------------------------------------------------------------------------
// rl_env_cfg (host, may be null = the env's defaults) -> the by-value kernel argument.  frame_skip 0 = env default.
template <class Env>
static int device_cfg(const rl_env_cfg* cfg, EnvCfg& c) {
    c = default_cfg<Env, float>();
    if (!cfg) return 0;
    if (cfg->frame_skip < 0 || cfg->frame_skip > 64)
        return set_error(RL_ERR_ARG, "rl_env_cfg.frame_skip = %d (0 = env default, 1..64)", cfg->frame_skip);
    if (cfg->action_noise < 0.0f || cfg->obs_noise < 0.0f)
        return set_error(RL_ERR_ARG, "rl_env_cfg: negative noise scale");
    c.ctrl_cost_coeff = cfg->ctrl_cost_coeff; c.alive_coeff = cfg->alive_coeff;
    c.action_noise = cfg->action_noise; c.obs_noise = cfg->obs_noise;
    if (cfg->frame_skip > 0) c.frame_skip = cfg->frame_skip;
    c.flags = cfg->flags;
    constexpr bool legged = has_mjc<Env>::value;          // HalfCheetah, Walker2D, Hopper (csrc/dyn_mjc.h)
    if ((cfg->flags & RL_CFG_LIMIT_MUJOCO) && !(std::is_same<Env, Swimmer>::value || legged))
        return set_error(RL_ERR_UNSUPPORTED, "rl_env_cfg.flags: RL_CFG_LIMIT_MUJOCO (soft-constraint joint limits) is built "
                                             "for the Swimmer, HalfCheetah, Walker2D and Hopper");
    if ((cfg->flags & RL_CFG_CONTACT_MUJOCO) && !legged)
        return set_error(RL_ERR_UNSUPPORTED, "rl_env_cfg.flags: RL_CFG_CONTACT_MUJOCO (soft-constraint floor contacts) is "
                                             "built for HalfCheetah, Walker2D and Hopper");
    if (cfg->link_len < 0.0f || cfg->link_len > 8.0f)
        return set_error(RL_ERR_ARG, "rl_env_cfg.link_len = %g (0 = the model's, else (0, 8])", (double)cfg->link_len);
    if (cfg->link_len > 0.0f) c.link_len = cfg->link_len;
    return 0;
}

template <class Env>
static int launch_com(int n, const float* state, float* com4, hipStream_t st) {
    if constexpr (Env::HAS_COM) {
        hipLaunchKernelGGL(vecenv_com_kernel<Env>, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, n, state, com4);
        return check_launch("vecenv_com_kernel");
    } else {
        return set_error(RL_ERR_UNSUPPORTED, "rl_vecenv_com: env kind %d has no subtree centre of mass", Env::KIND);
    }
}

template <class Env>
static int launch_reset(int n, float* state, int32_t* ts, const uint8_t* mask, const float* draws,
                        uint64_t seed, uint64_t step, int env_offset, const rl_env_cfg* cfg, float* obs,
                        hipStream_t st) {
    dim3 grid((n + BLOCK - 1) / BLOCK);
    EnvCfg c;
    int rc = device_cfg<Env>(cfg, c);
    if (rc) return rc;
    hipLaunchKernelGGL(vecenv_reset_kernel<Env>, grid, dim3(BLOCK), 0, st, n, state, ts, mask, draws, seed,
                       step, env_offset, c, cfg ? cfg->obs_noise_z : nullptr, obs);
    return check_launch("vecenv_reset_kernel");
}

template <class Env>
static int launch_observe(int n, const float* state, float* obs, hipStream_t st) {
    hipLaunchKernelGGL(vecenv_observe_kernel<Env>, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, n, state, obs);
    return check_launch("vecenv_observe_kernel");
}

template <class Env>
static int launch_step(int n, int normalize, float scale_reward, int mpl, int auto_reset, float* state,
                       int32_t* ts,
                       const float* actions, const float* reset_draws, uint64_t seed, uint64_t step,
                       const uint64_t* step_dev, int env_offset, const rl_env_cfg* cfg, float* obs, float* reward,
                       uint8_t* done, hipStream_t st) {
    dim3 grid((n + BLOCK - 1) / BLOCK);
    EnvCfg c;
    int rc = device_cfg<Env>(cfg, c);
    if (rc) return rc;
    if constexpr (has_mjc<Env>::value && !is_mjc_env<Env>::value) {
        // limit_model / contact_model = "mujoco": the kernel instantiated for the soft-constraint step
        if (c.flags & (RL_CFG_LIMIT_MUJOCO | RL_CFG_CONTACT_MUJOCO))
            return launch_step<MjcEnv<Env>>(n, normalize, scale_reward, mpl, auto_reset, state, ts, actions, reset_draws, seed,
                                            step, step_dev, env_offset, cfg, obs, reward, done, st);
    }
    hipLaunchKernelGGL(vecenv_step_kernel<Env>, grid, dim3(BLOCK), 0, st, n, normalize, scale_reward, mpl,
                       auto_reset, state, ts, actions, reset_draws, seed, step, step_dev, env_offset, c,
                       cfg ? cfg->action_noise_z : nullptr, cfg ? cfg->obs_noise_z : nullptr, obs, reward, done);
    return check_launch("vecenv_step_kernel");
}

// ---- launch rules as data ----------------------------------------------------------------------------------------------
// plan_rollout<Env> decides kernel and shape from the arguments (and rl_launch_opts: 0 = these rules, anything else an
// explicit request of a test or an A/B run); launch_rollout<Env> executes the plan; rl_rollout_plan_query hands the plan to
// the caller.  Nothing here reads the environment.
static const rl_launch_opts NO_OPTS = {};
static const char* env_name(int kind) {
    static const char* const names[] = {"Cartpole", "DoublePendulum", "Swimmer", "HalfCheetah", "CartpoleSwingup", "Walker2D",
                                        "Hopper", "InvertedDoublePendulum"};
    return (kind >= 0 && kind < 8) ? names[kind] : "?";
}

// wavefronts per workgroup of the lane-group shapes
static int lane_group_wpb(int waves, const rl_launch_opts& o) {
    const int v = o.rollout_wpb;
    if (v == 1 || v == 2 || v == 4) return v;
    return waves <= 256 ? 1 : LANE_TPB / 64;
}

constexpr size_t LDS_LIMIT = 160 * 1024;
// The wide / dual rollouts keep their weight fragments in LDS next to one [input][lane] tile per wavefront: a workgroup
// of several wavefronts may not fit where a single-wavefront one does.  Largest wavefronts-per-workgroup <= wpb whose
// workgroup fits the 160 KB of a CU; 0 when not even one wavefront does (the plan is then UNSUPPORTED and the Python side
// samples such a policy through the per-transition loop).
template <class LdsFn>
static int fit_wpb(int wpb, LdsFn lds_bytes) {
    for (; wpb >= 1; wpb >>= 1)
        if (lds_bytes(64 * wpb) <= LDS_LIMIT) return wpb;
    return 0;
}

template <class Env>
static size_t rollout_lds_bytes(int h0, int h1, int h2, int s0, int s1, int s2) {
    WideShape ms, ss;
    if (!wide_shape(Env::OBS, Env::ACT, h0, h1, h2, ms)) return 0;
    if (s0 == 0) return RolloutPolicyWide<Env>::lds_floats(ms, 64) * sizeof(float);
    if (!wide_shape(Env::OBS, Env::ACT, s0, s1, s2, ss)) return 0;
    return RolloutPolicyDual<Env>::lds_floats(ms, ss, 64) * sizeof(float);
}

static void plan_fill(rl_rollout_plan* p, int kernel, int epw, int waves, int wpb, size_t lds, const char* name) {
    p->kernel = kernel;
    p->envs_per_wavefront = epw;
    p->wavefronts = waves;
    p->wavefronts_per_workgroup = wpb;
    p->workgroups = wpb > 0 ? (waves + wpb - 1) / wpb : 0;
    p->lds_bytes = (int32_t)lds;
    p->lds_limit = (int32_t)LDS_LIMIT;
    p->reserved = 0;
    snprintf(p->name, sizeof(p->name), "%s", name);
}

// returns RL_OK with plan->kernel != RL_ROLLOUT_UNSUPPORTED, or RL_ERR_UNSUPPORTED (plan->kernel = UNSUPPORTED, the error
// string says why), or RL_ERR_ARG
template <class Env>
static int plan_rollout(const rl_rollout_args* g, rl_rollout_plan* p) {
    const rl_launch_opts& o = g->opts ? *g->opts : NO_OPTS;
    const int n = g->n_envs, T = g->horizon;
    const int cfg_flags = g->cfg ? g->cfg->flags : 0;
    plan_fill(p, RL_ROLLOUT_UNSUPPORTED, 0, 0, 0, 0, "");
    char nm[96];
    const bool equal = g->hidden2 == 0 && g->hidden0 == g->hidden1 && (g->hidden0 == 32 || g->hidden0 == 64);
    WideShape act_probe;
    act_probe.L = g->hidden2 > 0 ? 3 : 2;
    const bool wide_acts_ok = wide_activations(g->layer_activations, act_probe);   // tanh / identity layers only
    WideShape std_probe;
    std_probe.L = g->std_hidden2 > 0 ? 3 : 2;
    const bool std_acts_ok = wide_activations(g->std_layer_activations, std_probe);
    if ((g->layer_activations != 0 && !wide_acts_ok && (!equal || g->theta_std != nullptr)) ||
        (g->theta_std != nullptr && !std_acts_ok))
        return set_error(RL_ERR_UNSUPPORTED, "rl_rollout_gaussian_mlp: rectify hidden layers run on the (32,32) / (64,64) "
                         "kernels only (and not next to a log-std network), identity layers on those and on the wide / two-"
                         "network kernels (hidden %d,%d,%d, layers 0x%x / 0x%x)", g->hidden0, g->hidden1, g->hidden2,
                         g->layer_activations, g->std_layer_activations);
    const bool small_offsets = (size_t)Env::OBS * (size_t)T * (size_t)n * 4 < ((size_t)1 << 32);   // 32-bit plane offsets
    const int epw_req = (o.rollout_epw == 16 || o.rollout_epw == 64) ? o.rollout_epw : 0;
    const int epw_generic = epw_req ? epw_req : (n <= 16 * 1024 ? 16 : 64);
    const bool norm = g->norm != nullptr && (g->norm->normalize_obs || g->norm->normalize_reward);
    if (norm) {
        // running normalisation: the generic kernels of the (32,32) / (64,64) policies (the estimates ride in registers
        // next to the env state), every env of the launch reset at its start (reset() feeds the estimate)
        if (!equal || g->theta_std != nullptr)
            return set_error(RL_ERR_UNSUPPORTED, "rl_rollout_gaussian_mlp: running observation / reward normalisation runs on "
                             "the (32,32) / (64,64) kernels only (hidden %d,%d,%d)", g->hidden0, g->hidden1, g->hidden2);
        if (!g->reset_at_start && !g->last_obs && g->state)
            return set_error(RL_ERR_ARG, "rl_rollout_gaussian_mlp: a continuation under running normalisation starts from "
                             "last_obs (the whitened observation the previous launch ended on)");
        if (g->cfg && g->cfg->obs_noise != 0.0f && g->norm->normalize_obs)
            return set_error(RL_ERR_UNSUPPORTED, "rl_rollout_gaussian_mlp: obs_noise together with normalize_obs is sampled "
                             "through rl_vecenv_step (the terminal observation's noise draw)");
        const int epw = epw_generic, waves = (n + epw - 1) / epw;
        snprintf(nm, sizeof(nm), "rollout_kernel<%s, %d, %d, %d, acts, norm>", env_name(Env::KIND), g->hidden0, g->hidden1, epw);
        plan_fill(p, RL_ROLLOUT_GENERIC, epw, waves, (epw == 16) ? lane_group_wpb(waves, o) : 1, 0, nm);
        return RL_OK;
    }
    if (g->theta_std != nullptr) {
        // a log-std NETWORK (adaptive_std / std_network): both networks in the kernel, log_std planes recorded
        if (!g->log_stds) return set_error(RL_ERR_ARG, "rl_rollout_gaussian_mlp: theta_std without log_stds");
        WideShape ms, ss;
        if (!wide_shape(Env::OBS, Env::ACT, g->hidden0, g->hidden1, g->hidden2, ms) ||
            !wide_shape(Env::OBS, Env::ACT, g->std_hidden0, g->std_hidden1, g->std_hidden2, ss))
            return set_error(RL_ERR_UNSUPPORTED,
                             "rl_rollout_gaussian_mlp: mean net (%d,%d,%d) / log-std net (%d,%d,%d): each two or three tanh "
                             "layers of 32 / 64 / 128 units", g->hidden0, g->hidden1, g->hidden2, g->std_hidden0,
                             g->std_hidden1, g->std_hidden2);
        const int epw = epw_generic, waves = (n + epw - 1) / epw;
        const int wpb = fit_wpb((epw == 16) ? lane_group_wpb(waves, o) : 1, [&](int threads) {
            return RolloutPolicyDual<Env>::lds_floats(ms, ss, threads) * sizeof(float); });
        if (wpb == 0)
            return set_error(RL_ERR_UNSUPPORTED, "rollout of the two networks needs %zu B of LDS (a CU has %zu)",
                             RolloutPolicyDual<Env>::lds_floats(ms, ss, 64) * sizeof(float), LDS_LIMIT);
        snprintf(nm, sizeof(nm), "rollout_dual_kernel<%s, %d>", env_name(Env::KIND), epw);
        plan_fill(p, RL_ROLLOUT_DUAL, epw, waves, wpb, RolloutPolicyDual<Env>::lds_floats(ms, ss, 64 * wpb) * sizeof(float), nm);
        return RL_OK;
    }
    WideShape shape;
    const bool wide_ok = wide_shape(Env::OBS, Env::ACT, g->hidden0, g->hidden1, g->hidden2, shape);
    const int quad_waves = (n + QUAD_ENVS - 1) / QUAD_ENVS;
    if constexpr (std::is_same<Env, Swimmer>::value) {
        // lane group per env (swimmer_lane_kernel = 1: the env-per-lane kernel; RL_CFG_LIMIT_MUJOCO: the soft-constraint
        // limits live in the scalar sub-step program only)
        const bool lane_kernel = o.swimmer_lane_kernel == 1 || (cfg_flags & RL_CFG_LIMIT_MUJOCO) != 0;
        if (!lane_kernel && small_offsets && equal) {
            snprintf(nm, sizeof(nm), "rollout_swimmer_quad_kernel<%d>", g->hidden0);
            plan_fill(p, RL_ROLLOUT_SWIMMER_QUAD, QUAD_ENVS, quad_waves, lane_group_wpb(quad_waves, o), 0, nm);
            return RL_OK;
        }
        // the same lane-group physics under a wide / deep policy (rollout_epw set: the generic shapes, for tests)
        if (!lane_kernel && small_offsets && !equal && epw_req == 0 && wide_ok) {
            // four wavefronts per group while every one of them still finds a SIMD of its own
            const size_t coop_lds = RolloutPolicyCoop<Env>::lds_floats(shape) * sizeof(float);
            const bool coop = o.swimmer_coop == 1 || (o.swimmer_coop != 2 && quad_waves * COOP_WAVES <= 1024);
            if (coop && coop_lds <= LDS_LIMIT) {
                plan_fill(p, RL_ROLLOUT_SWIMMER_QUAD_COOP, QUAD_ENVS, quad_waves * COOP_WAVES, COOP_WAVES, coop_lds,
                          "rollout_swimmer_quad_coop_kernel");
                return RL_OK;
            }
            const int wpb = fit_wpb(lane_group_wpb(quad_waves, o), [&](int threads) {
                return RolloutPolicyWide<Env>::lds_floats(shape, threads) * sizeof(float); });
            if (wpb > 0) {
                plan_fill(p, RL_ROLLOUT_SWIMMER_QUAD_WIDE, QUAD_ENVS, quad_waves, wpb,
                          RolloutPolicyWide<Env>::lds_floats(shape, 64 * wpb) * sizeof(float), "rollout_swimmer_quad_wide_kernel");
                return RL_OK;
            }
        }
    }
    if constexpr (std::is_same<Env, HalfCheetah>::value || std::is_same<Env, Walker2D>::value) {
        // one body per lane while every lane-group wavefront still gets a SIMD of its own (two_leg_lane_kernel = 2: the
        // generic kernel, for A/B timing and for the tests that run every shape)
        // (RL_CFG_LIMIT_MUJOCO / RL_CFG_CONTACT_MUJOCO: the constraint solve of dyn_mjc.h lives in the env-per-lane program only)
        const bool lanes_on = o.two_leg_lane_kernel != 2 && epw_req == 0 &&
                              (cfg_flags & (RL_CFG_LIMIT_MUJOCO | RL_CFG_CONTACT_MUJOCO)) == 0;
        // one env per wavefront while the wavefronts still find (about) a SIMD each
        const bool wave_shape = o.two_leg_wave_kernel == 1 || (o.two_leg_wave_kernel != 2 && n <= 2048);
        if (lanes_on && wave_shape && small_offsets && equal) {
            snprintf(nm, sizeof(nm), "rollout_two_leg_wave_kernel<%s, %d>", env_name(Env::KIND), g->hidden0);
            plan_fill(p, RL_ROLLOUT_TWO_LEG_WAVE, 1, n, lane_group_wpb(n, o), 0, nm);
            return RL_OK;
        }
        if (lanes_on && small_offsets && n <= 16 * 1024 && equal) {
            snprintf(nm, sizeof(nm), "rollout_two_leg_quad_kernel<%s, %d>", env_name(Env::KIND), g->hidden0);
            plan_fill(p, RL_ROLLOUT_TWO_LEG_QUAD, QUAD_ENVS, quad_waves, lane_group_wpb(quad_waves, o), 0, nm);
            return RL_OK;
        }
        if (lanes_on && small_offsets && n <= 16 * 1024 && !equal && wide_ok) {
            const int wpb = fit_wpb(lane_group_wpb(quad_waves, o), [&](int threads) {
                return RolloutPolicyWide<Env>::lds_floats(shape, threads) * sizeof(float); });
            if (wpb > 0) {
                snprintf(nm, sizeof(nm), "rollout_two_leg_quad_wide_kernel<%s>", env_name(Env::KIND));
                plan_fill(p, RL_ROLLOUT_TWO_LEG_QUAD_WIDE, QUAD_ENVS, quad_waves, wpb,
                          RolloutPolicyWide<Env>::lds_floats(shape, 64 * wpb) * sizeof(float), nm);
                return RL_OK;
            }
        }
    }
    // 16 envs per wavefront while that still leaves every wavefront a SIMD of its own (1024 SIMDs); beyond, the
    // replicated physics would cost throughput.
    const int epw = epw_generic, waves = (n + epw - 1) / epw;
    int wpb = (epw == 16) ? lane_group_wpb(waves, o) : 1;
    if (equal) {
        snprintf(nm, sizeof(nm), "rollout_kernel<%s, %d, %d, %d>", env_name(Env::KIND), g->hidden0, g->hidden1, epw);
        plan_fill(p, RL_ROLLOUT_GENERIC, epw, waves, wpb, 0, nm);
        return RL_OK;
    }
    // wide / deep policies: two or three layers of 32 / 64 / 128 units, weight fragments in (dynamic) LDS
    if (!wide_ok)
        return set_error(RL_ERR_UNSUPPORTED,
                         "rl_rollout_gaussian_mlp: hidden sizes (%d,%d,%d) have no fused kernel (two or three tanh "
                         "layers of 32 / 64 / 128 units each); use the per-step rl_vecenv_step path",
                         g->hidden0, g->hidden1, g->hidden2);
    wpb = fit_wpb(wpb, [&](int threads) { return RolloutPolicyWide<Env>::lds_floats(shape, threads) * sizeof(float); });
    if (wpb == 0)
        return set_error(RL_ERR_UNSUPPORTED, "wide rollout policy needs %zu B of LDS (a CU has %zu)",
                         RolloutPolicyWide<Env>::lds_floats(shape, 64) * sizeof(float), LDS_LIMIT);
    snprintf(nm, sizeof(nm), "rollout_wide_kernel<%s, %d>", env_name(Env::KIND), epw);
    plan_fill(p, RL_ROLLOUT_WIDE, epw, waves, wpb, RolloutPolicyWide<Env>::lds_floats(shape, 64 * wpb) * sizeof(float), nm);
    return RL_OK;
}

// hipFuncSetAttribute once per kernel (dynamic LDS beyond 64 KB)
template <class K>
static int allow_big_lds(K kern, bool& done) {
    if (done) return 0;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
    if (e != hipSuccess) return set_error(RL_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    done = true;
    return 0;
}

template <class Env>
static int launch_rollout(const rl_rollout_args* g, hipStream_t st) {
    if constexpr (has_mjc<Env>::value && !is_mjc_env<Env>::value) {
        // limit_model / contact_model = "mujoco": the env-per-lane kernels instantiated for the soft-constraint step
        if (g->cfg && (g->cfg->flags & (RL_CFG_LIMIT_MUJOCO | RL_CFG_CONTACT_MUJOCO))) return launch_rollout<MjcEnv<Env>>(g, st);
    }
    rl_rollout_plan pl;
    int rc = plan_rollout<Env>(g, &pl);
    if (rc) return rc;
    RolloutDev a;
    a.n = g->n_envs; a.T = g->horizon; a.max_path_length = g->max_path_length;
    a.normalize = g->normalize; a.reset_at_start = g->reset_at_start; a.env_offset = g->env_offset;
    a.scale_reward = g->scale_reward; a.log_min_std = g->log_min_std;
    a.seed = g->seed; a.step_counter = g->step_counter;
    a.state = g->state; a.ts = g->ts; a.theta = g->theta; a.eps = g->eps; a.reset_draws = g->reset_draws;
    a.obs = g->obs; a.actions = g->actions; a.means = g->means; a.rewards = g->rewards; a.dones = g->dones;
    a.last_obs = g->last_obs;
    rc = device_cfg<Env>(g->cfg, a.cfg);
    if (rc) return rc;
    a.act_noise_z = g->cfg ? g->cfg->action_noise_z : nullptr;
    a.obs_noise_z = g->cfg ? g->cfg->obs_noise_z : nullptr;
    a.log_stds = g->log_stds;
    a.act0 = layer_act(RL_ACT_TANH, g->layer_activations, 0);
    a.act1 = layer_act(RL_ACT_TANH, g->layer_activations, 1);
    const bool acts = a.act0 != RL_ACT_TANH || a.act1 != RL_ACT_TANH;     // the run-time-activation instantiations
    const bool norm = g->norm != nullptr && (g->norm->normalize_obs || g->norm->normalize_reward);
    a.norm_obs = a.norm_rew = 0;
    if (norm) {
        const rl_running_norm& q = *g->norm;
        if ((q.normalize_obs && (!q.obs_mean || !q.obs_var)) || (q.normalize_reward && (!q.reward_mean || !q.reward_var)))
            return set_error(RL_ERR_ARG, "rl_running_norm: null estimate array");
        a.nobs_mean = q.obs_mean; a.nobs_var = q.obs_var; a.nrew_mean = q.reward_mean; a.nrew_var = q.reward_var;
        a.obs_alpha = q.obs_alpha; a.rew_alpha = q.reward_alpha;
        a.norm_obs = q.normalize_obs ? 1 : 0; a.norm_rew = q.normalize_reward ? 1 : 0;
    }
    const dim3 grid(pl.workgroups), block(64 * pl.wavefronts_per_workgroup);
    const size_t lds = (size_t)pl.lds_bytes;
    const int H = g->hidden0, epw = pl.envs_per_wavefront;
    WideShape shape, sshape;
    if (wide_shape(Env::OBS, Env::ACT, g->hidden0, g->hidden1, g->hidden2, shape))
        wide_activations(g->layer_activations, shape);      // identity layers (plan_rollout refused anything else)
    switch (pl.kernel) {
        case RL_ROLLOUT_DUAL: {
            wide_shape(Env::OBS, Env::ACT, g->std_hidden0, g->std_hidden1, g->std_hidden2, sshape);
            wide_activations(g->std_layer_activations, sshape);
            const size_t mean_floats = RolloutPolicyWide<Env>::lds_floats(shape, 64 * pl.wavefronts_per_workgroup);
            static bool a16 = false, a64 = false;
            if (epw == 16) {
                if ((rc = allow_big_lds(rollout_dual_kernel<Env, 16>, a16))) return rc;
                hipLaunchKernelGGL((rollout_dual_kernel<Env, 16>), grid, block, lds, st, a, shape, sshape, g->theta_std, (int)mean_floats);
            } else {
                if ((rc = allow_big_lds(rollout_dual_kernel<Env, 64>, a64))) return rc;
                hipLaunchKernelGGL((rollout_dual_kernel<Env, 64>), grid, block, lds, st, a, shape, sshape, g->theta_std, (int)mean_floats);
            }
            break;
        }
        case RL_ROLLOUT_GENERIC:
#define RL_GEN(HH, EE) do { if (norm) hipLaunchKernelGGL((rollout_kernel<Env, HH, HH, EE, true, true>), grid, block, 0, st, a); \
                            else if (acts) hipLaunchKernelGGL((rollout_kernel<Env, HH, HH, EE, true>), grid, block, 0, st, a); \
                            else hipLaunchKernelGGL((rollout_kernel<Env, HH, HH, EE, false>), grid, block, 0, st, a); } while (0)
            if (H == 32 && epw == 16) RL_GEN(32, 16);
            else if (H == 32) RL_GEN(32, 64);
            else if (epw == 16) RL_GEN(64, 16);
            else RL_GEN(64, 64);
#undef RL_GEN
            break;
        case RL_ROLLOUT_WIDE: {
            static bool a16 = false, a64 = false;
            if (epw == 16) {
                if ((rc = allow_big_lds(rollout_wide_kernel<Env, 16>, a16))) return rc;
                hipLaunchKernelGGL((rollout_wide_kernel<Env, 16>), grid, block, lds, st, a, shape);
            } else {
                if ((rc = allow_big_lds(rollout_wide_kernel<Env, 64>, a64))) return rc;
                hipLaunchKernelGGL((rollout_wide_kernel<Env, 64>), grid, block, lds, st, a, shape);
            }
            break;
        }
        default:
            if constexpr (std::is_same<Env, Swimmer>::value) {
                if (pl.kernel == RL_ROLLOUT_SWIMMER_QUAD) {
                    if (H == 32 && !acts) hipLaunchKernelGGL((rollout_swimmer_quad_kernel<32, false>), grid, block, 0, st, a);
                    else if (H == 32) hipLaunchKernelGGL((rollout_swimmer_quad_kernel<32, true>), grid, block, 0, st, a);
                    else if (!acts) hipLaunchKernelGGL((rollout_swimmer_quad_kernel<64, false>), grid, block, 0, st, a);
                    else hipLaunchKernelGGL((rollout_swimmer_quad_kernel<64, true>), grid, block, 0, st, a);
                    break;
                }
                if (pl.kernel == RL_ROLLOUT_SWIMMER_QUAD_COOP) {
                    static bool ca = false;
                    if ((rc = allow_big_lds(rollout_swimmer_quad_coop_kernel, ca))) return rc;
                    hipLaunchKernelGGL(rollout_swimmer_quad_coop_kernel, grid, block, lds, st, a, shape);
                    break;
                }
                if (pl.kernel == RL_ROLLOUT_SWIMMER_QUAD_WIDE) {
                    static bool wa = false;
                    if ((rc = allow_big_lds(rollout_swimmer_quad_wide_kernel, wa))) return rc;
                    hipLaunchKernelGGL(rollout_swimmer_quad_wide_kernel, grid, block, lds, st, a, shape);
                    break;
                }
            }
            if constexpr (std::is_same<Env, HalfCheetah>::value || std::is_same<Env, Walker2D>::value) {
                if (pl.kernel == RL_ROLLOUT_TWO_LEG_WAVE) {
                    if (H == 32 && !acts) hipLaunchKernelGGL((rollout_two_leg_wave_kernel<Env, 32, false>), grid, block, 0, st, a);
                    else if (H == 32) hipLaunchKernelGGL((rollout_two_leg_wave_kernel<Env, 32, true>), grid, block, 0, st, a);
                    else if (!acts) hipLaunchKernelGGL((rollout_two_leg_wave_kernel<Env, 64, false>), grid, block, 0, st, a);
                    else hipLaunchKernelGGL((rollout_two_leg_wave_kernel<Env, 64, true>), grid, block, 0, st, a);
                    break;
                }
                if (pl.kernel == RL_ROLLOUT_TWO_LEG_QUAD) {
                    if (H == 32 && !acts) hipLaunchKernelGGL((rollout_two_leg_quad_kernel<Env, 32, false>), grid, block, 0, st, a);
                    else if (H == 32) hipLaunchKernelGGL((rollout_two_leg_quad_kernel<Env, 32, true>), grid, block, 0, st, a);
                    else if (!acts) hipLaunchKernelGGL((rollout_two_leg_quad_kernel<Env, 64, false>), grid, block, 0, st, a);
                    else hipLaunchKernelGGL((rollout_two_leg_quad_kernel<Env, 64, true>), grid, block, 0, st, a);
                    break;
                }
                if (pl.kernel == RL_ROLLOUT_TWO_LEG_QUAD_WIDE) {
                    static bool ta = false;
                    if ((rc = allow_big_lds(rollout_two_leg_quad_wide_kernel<Env>, ta))) return rc;
                    hipLaunchKernelGGL((rollout_two_leg_quad_wide_kernel<Env>), grid, block, lds, st, a, shape);
                    break;
                }
            }
            return set_error(RL_ERR_UNSUPPORTED, "rollout plan %d has no launch for this env", pl.kernel);
    }
    return check_launch(pl.name);
}

}  // namespace rl

using namespace rl;

template <class Env>
static void fill_default_cfg(rl_env_cfg* cfg) {
    const EnvCfg c = default_cfg<Env, float>();
    cfg->ctrl_cost_coeff = c.ctrl_cost_coeff; cfg->alive_coeff = c.alive_coeff;
    cfg->action_noise = 0.0f; cfg->obs_noise = 0.0f; cfg->frame_skip = c.frame_skip; cfg->flags = 0;
    cfg->link_len = c.link_len; cfg->reserved = 0.0f;
    cfg->action_noise_z = nullptr; cfg->obs_noise_z = nullptr;
}

#define RL_DISPATCH_ENV(kind, CALL)                                                         \
    switch (kind) {                                                                         \
        case RL_ENV_CARTPOLE: { using E = rl::Cartpole; return CALL; }                      \
        RL_EXTRA_ENV_CASES(CALL)                                                            \
        default: return set_error(RL_ERR_ARG, "unknown or unbuilt env kind %d", (int)(kind)); \
    }

extern "C" int rl_env_query(int kind, int* obs_dim, int* act_dim, int* state_dim, int* reset_draws,
                            int* reset_is_normal) {
#define Q (obs_dim && (*obs_dim = E::OBS), act_dim && (*act_dim = E::ACT),                         \
           state_dim && (*state_dim = E::STATE), reset_draws && (*reset_draws = E::RESET_DRAWS),   \
           reset_is_normal && (*reset_is_normal = E::RESET_NORMAL ? 1 : 0), (int)RL_OK)
    RL_DISPATCH_ENV(kind, Q)
#undef Q
}

extern "C" int rl_env_terminates(int kind) {
#define Q (E::TERMINATES ? 1 : 0)
    RL_DISPATCH_ENV(kind, Q)
#undef Q
}

extern "C" int rl_env_action_bounds(int kind, float* lb, float* ub) {
    if (!lb || !ub) return set_error(RL_ERR_ARG, "rl_env_action_bounds: null output");
#define Q (E::template action_bounds<float>(lb, ub), (int)RL_OK)
    RL_DISPATCH_ENV(kind, Q)
#undef Q
}

extern "C" int rl_env_default_cfg(int kind, rl_env_cfg* cfg) {
    if (!cfg) return set_error(RL_ERR_ARG, "rl_env_default_cfg: null output");
#define Q (fill_default_cfg<E>(cfg), (int)RL_OK)
    RL_DISPATCH_ENV(kind, Q)
#undef Q
}

extern "C" int rl_vecenv_com(int kind, int n, const float* state, float* com4, void* stream) {
    if (n <= 0 || !state || !com4) return set_error(RL_ERR_ARG, "rl_vecenv_com: bad argument");
    RL_DISPATCH_ENV(kind, launch_com<E>(n, state, com4, (hipStream_t)stream))
}

extern "C" int rl_vecenv_reset(int kind, int n, float* state, int32_t* ts, const uint8_t* mask,
                               const float* draws, uint64_t seed, uint64_t step_counter, int env_offset,
                               const rl_env_cfg* cfg, float* obs, void* stream) {
    if (n <= 0 || !state || !ts || !obs) return set_error(RL_ERR_ARG, "rl_vecenv_reset: bad argument");
    RL_DISPATCH_ENV(kind, launch_reset<E>(n, state, ts, mask, draws, seed, step_counter, env_offset, cfg, obs,
                                          (hipStream_t)stream))
}

extern "C" int rl_vecenv_observe(int kind, int n, const float* state, float* obs, void* stream) {
    if (n <= 0 || !state || !obs) return set_error(RL_ERR_ARG, "rl_vecenv_observe: bad argument");
    RL_DISPATCH_ENV(kind, launch_observe<E>(n, state, obs, (hipStream_t)stream))
}

extern "C" int rl_vecenv_step(int kind, int n, int normalize, float scale_reward, int max_path_length,
                              int auto_reset, float* state, int32_t* ts, const float* actions, const float* reset_draws,
                              uint64_t seed, uint64_t step_counter, int env_offset, const rl_env_cfg* cfg,
                              float* obs, float* reward, uint8_t* done, void* stream) {
    if (n <= 0 || !state || !ts || !actions || !obs || !reward || !done)
        return set_error(RL_ERR_ARG, "rl_vecenv_step: bad argument");
    RL_DISPATCH_ENV(kind, launch_step<E>(n, normalize, scale_reward, max_path_length, auto_reset, state, ts,
                                         actions,
                                         reset_draws, seed, step_counter, nullptr, env_offset, cfg, obs, reward, done,
                                         (hipStream_t)stream))
}

extern "C" int rl_vecenv_step_graph(int kind, int n, int normalize, float scale_reward, int max_path_length,
                                    int auto_reset, float* state, int32_t* ts, const float* actions, uint64_t seed,
                                    const uint64_t* step_counter_dev, int env_offset, const rl_env_cfg* cfg,
                                    float* obs, float* reward, uint8_t* done, void* stream) {
    if (n <= 0 || !state || !ts || !actions || !obs || !reward || !done || !step_counter_dev)
        return set_error(RL_ERR_ARG, "rl_vecenv_step_graph: bad argument");
    RL_DISPATCH_ENV(kind, launch_step<E>(n, normalize, scale_reward, max_path_length, auto_reset, state, ts,
                                         actions, nullptr, seed, 0, step_counter_dev, env_offset, cfg, obs, reward,
                                         done, (hipStream_t)stream))
}

namespace rl {
__global__ void counter_add_kernel(uint64_t* c, uint64_t inc) { *c += inc; }
}

extern "C" int rl_counter_add(uint64_t* counter_dev, uint64_t increment, void* stream) {
    if (!counter_dev) return set_error(RL_ERR_ARG, "rl_counter_add: null counter");
    hipLaunchKernelGGL(rl::counter_add_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, counter_dev, increment);
    return check_launch("counter_add_kernel");
}

extern "C" int rl_rollout_gaussian_mlp(const rl_rollout_args* g, void* stream) {
    if (!g) return set_error(RL_ERR_ARG, "rl_rollout_gaussian_mlp: null args");
    if (g->n_envs <= 0 || g->horizon <= 0 || !g->state || !g->ts || !g->theta || !g->obs || !g->actions ||
        !g->means || !g->rewards || !g->dones)
        return set_error(RL_ERR_ARG, "rl_rollout_gaussian_mlp: bad argument");
    RL_DISPATCH_ENV(g->kind, launch_rollout<E>(g, (hipStream_t)stream))
}

extern "C" int rl_rollout_plan_query(const rl_rollout_args* g, rl_rollout_plan* plan) {
    if (!g || !plan) return set_error(RL_ERR_ARG, "rl_rollout_plan_query: null argument");
    if (g->n_envs <= 0 || g->horizon <= 0) return set_error(RL_ERR_ARG, "rl_rollout_plan_query: bad argument");
    RL_DISPATCH_ENV(g->kind, plan_rollout<E>(g, plan))
}

extern "C" int rl_rollout_lds_bytes(int kind, int hidden0, int hidden1, int hidden2, int std_hidden0, int std_hidden1,
                                    int std_hidden2, size_t* bytes, size_t* limit) {
    if (!bytes) return set_error(RL_ERR_ARG, "rl_rollout_lds_bytes: null output");
    if (limit) *limit = LDS_LIMIT;
#define Q (*bytes = rollout_lds_bytes<E>(hidden0, hidden1, hidden2, std_hidden0, std_hidden1, std_hidden2), (int)RL_OK)
    RL_DISPATCH_ENV(kind, Q)
#undef Q
}

extern "C" int rl_debug_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                               uint32_t k1, int count, uint32_t* out, void* stream) {
    if (count <= 0 || !out) return set_error(RL_ERR_ARG, "rl_debug_philox: bad argument");
    hipLaunchKernelGGL(philox_debug_kernel, dim3((count + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       c0, c1, c2, c3, k0, k1, count, out);
    return check_launch("philox_debug_kernel");
}
