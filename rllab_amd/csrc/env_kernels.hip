// env_kernels.hip -- lock-step vectorised env kernels and the fused rollout.
//
// One thread owns one env copy for the whole launch: state lives in VGPRs, HBM is
// touched only for the SoA state planes (coalesced: lane i <-> env i) and for the
// trajectory planes.  Envs never talk to each other, so there is no LDS traffic
// besides the policy weights and no inter-workgroup synchronisation at all.
// Build with -ffp-contract=on (front-end contraction): the env arithmetic must match
// the host oracle build bit for bit (rl_math.h); the policy MLP uses explicit FMAs.
#include <hip/hip_runtime.h>
#include "../../include/rllab_amd.h"
#include "capi_util.h"
#include "device_rng.h"
#include "envs.h"

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

template <class Env>
__device__ __forceinline__ void reset_one(float* s, const float* __restrict__ draws, int n, int i,
                                          uint64_t seed, uint32_t env_global, uint64_t step) {
    float d[Env::RESET_DRAWS];
    if (draws) {
#pragma unroll
        for (int k = 0; k < Env::RESET_DRAWS; ++k) d[k] = draws[(size_t)k * n + i];
    } else {
        philox_draws<Env::RESET_DRAWS, Env::RESET_NORMAL>(d, seed, env_global, step, RNG_RESET);
    }
    Env::template reset<float>(s, d);
}

// ---------------------------------------------------------------------------
// VecEnvExecutor.reset / masked Env.reset
// ---------------------------------------------------------------------------
template <class Env>
__global__ void __launch_bounds__(BLOCK)
vecenv_reset_kernel(int n, float* __restrict__ state, int32_t* __restrict__ ts,
                    const uint8_t* __restrict__ mask, const float* __restrict__ draws, uint64_t seed,
                    uint64_t step, int env_offset, float* __restrict__ obs) {
    int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    if (mask && !mask[i]) return;
    float s[Env::STATE];
    load_state<Env>(state, n, i, s);  // persisted solver state survives reset
    reset_one<Env>(s, draws, n, i, seed, (uint32_t)(env_offset + i), step);
    store_state<Env>(state, n, i, s);
    ts[i] = 0;
    float o[Env::OBS];
    Env::template observe<float>(s, o);
#pragma unroll
    for (int k = 0; k < Env::OBS; ++k) obs[(size_t)k * n + i] = o[k];
}

// ---------------------------------------------------------------------------
// VecEnvExecutor.step
// ---------------------------------------------------------------------------
template <class Env>
__global__ void __launch_bounds__(BLOCK)
vecenv_step_kernel(int n, int normalize, float scale_reward, int max_path_length, int auto_reset,
                   float* __restrict__ state, int32_t* __restrict__ ts,
                   const float* __restrict__ actions, const float* __restrict__ reset_draws,
                   uint64_t seed, uint64_t step, int env_offset, float* __restrict__ obs,
                   float* __restrict__ reward, uint8_t* __restrict__ done) {
    int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    float s[Env::STATE];
    load_state<Env>(state, n, i, s);
    float a[Env::ACT];
#pragma unroll
    for (int k = 0; k < Env::ACT; ++k) a[k] = actions[(size_t)k * n + i];
    float o[Env::OBS];
    float r;
    bool d;
    Env::template step<float>(s, a, normalize, o, r, d);
    int t = ts[i] + 1;
    if (max_path_length > 0 && t >= max_path_length) d = true;
    if (d && auto_reset) {
        reset_one<Env>(s, reset_draws, n, i, seed, (uint32_t)(env_offset + i), step);
        Env::template observe<float>(s, o);
        t = 0;
    }
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
__device__ __forceinline__ float fast_tanh(float x) {
    // tanh(x) = (e^{2x} - 1) / (e^{2x} + 1); clamp keeps e^{2x} finite.  Absolute
    // error < 2e-7 on the whole range, far inside the 1e-5 policy tolerance.
    float xc = fminf(fmaxf(x, -10.0f), 10.0f);
    float e = __expf(2.0f * xc);
    return (e - 1.0f) * __builtin_amdgcn_rcpf(e + 1.0f);
}

template <class Env, int H0, int H1>
struct PolicyLayout {
    static constexpr int DO = Env::OBS, DA = Env::ACT;
    static constexpr int W0 = 0;
    static constexpr int B0 = W0 + DO * H0;
    static constexpr int W1 = B0 + H0;
    static constexpr int B1 = W1 + H0 * H1;
    static constexpr int W2 = B1 + H1;
    static constexpr int B2 = W2 + H1 * DA;
    static constexpr int LS = B2 + DA;
    static constexpr int P = LS + DA;
    static constexpr int P_PAD = (P + 3) & ~3;
    // H0, H1 multiples of 4 => W0, B0, W1, B1, W2 are all 16-byte aligned in LDS
    static_assert(H0 % 4 == 0 && H1 % 4 == 0, "hidden sizes must be multiples of 4");
};

// mean = Wout^T tanh(W1^T tanh(W0^T o + b0) + b1) + bout   (network.py:36-101)
//
// One thread evaluates the MLP of its own env.  Weight rows are read from LDS
// with wave-uniform addresses (broadcast, 16 B per read); the layer input lives
// in a per-thread LDS column x[d][lane] (conflict-free) so that the loop over
// input units can stay ROLLED: with a fully unrolled layer LLVM clusters all P
// weight reads ahead of the FMAs and spills thousands of bytes per lane.
template <int IN, int OUT>
__device__ __forceinline__ void dense_layer(const float* __restrict__ w_lds, int w_off, int b_off,
                                            const float* __restrict__ x_col, float* y) {
#pragma unroll
    for (int j = 0; j < OUT; ++j) y[j] = w_lds[b_off + j];
#pragma unroll 2
    for (int d = 0; d < IN; ++d) {
        const float xd = x_col[d * BLOCK];
        const float* __restrict__ row = w_lds + w_off + d * OUT;
#pragma unroll
        for (int j = 0; j < OUT; ++j) y[j] = __builtin_fmaf(xd, row[j], y[j]);
    }
}

// x_col: per-thread LDS column (stride BLOCK) with room for max(DO, H0, H1) values;
// on entry it holds the observation.
template <class Env, int H0, int H1>
__device__ __forceinline__ void policy_mean(const float* __restrict__ w, float* __restrict__ x_col,
                                            float* mean) {
    using L = PolicyLayout<Env, H0, H1>;
    float h0[H0];
    dense_layer<L::DO, H0>(w, L::W0, L::B0, x_col, h0);
#pragma unroll
    for (int j = 0; j < H0; ++j) x_col[j * BLOCK] = fast_tanh(h0[j]);
    float h1[H1];
    dense_layer<H0, H1>(w, L::W1, L::B1, x_col, h1);
#pragma unroll
    for (int j = 0; j < H1; ++j) x_col[j * BLOCK] = fast_tanh(h1[j]);
    dense_layer<H1, L::DA>(w, L::W2, L::B2, x_col, mean);
}

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
};

template <class Env, int H0, int H1>
__global__ void __launch_bounds__(BLOCK) rollout_kernel(RolloutDev a) {
    using L = PolicyLayout<Env, H0, H1>;
    constexpr int XMAX = (L::DO > H0 ? (L::DO > H1 ? L::DO : H1) : (H0 > H1 ? H0 : H1));
    __shared__ __attribute__((aligned(16))) float w[L::P_PAD];
    __shared__ float xbuf[XMAX * BLOCK];
    for (int k = threadIdx.x; k < L::P; k += BLOCK) w[k] = a.theta[k];
    __syncthreads();
    float* x_col = xbuf + threadIdx.x;

    const int n = a.n, T = a.T;
    int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint32_t env_global = (uint32_t)(a.env_offset + i);
    const size_t plane = (size_t)T * n;

    float std_[Env::ACT];
#pragma unroll
    for (int k = 0; k < Env::ACT; ++k) std_[k] = __expf(fmaxf(w[L::LS + k], a.log_min_std));

    float s[Env::STATE];
    load_state<Env>(a.state, n, i, s);
    int ts = a.ts[i];
    const size_t draws_slice = (size_t)Env::RESET_DRAWS * n;
    if (a.reset_at_start) {
        reset_one<Env>(s, a.reset_draws, n, i, a.seed, env_global, a.step_counter);
        ts = 0;
    }
    float o[Env::OBS];
    Env::template observe<float>(s, o);

    for (int t = 0; t < T; ++t) {
        const size_t off = (size_t)t * n + i;
#pragma unroll
        for (int k = 0; k < Env::OBS; ++k) a.obs[k * plane + off] = o[k];

        float mean[Env::ACT], act[Env::ACT], z[Env::ACT];
#pragma unroll
        for (int k = 0; k < Env::OBS; ++k) x_col[k * BLOCK] = o[k];
        policy_mean<Env, H0, H1>(w, x_col, mean);
        if (a.eps) {
#pragma unroll
            for (int k = 0; k < Env::ACT; ++k) z[k] = a.eps[k * plane + off];
        } else {
            philox_draws<Env::ACT, true>(z, a.seed, env_global, a.step_counter + (uint64_t)t, RNG_POLICY);
        }
#pragma unroll
        for (int k = 0; k < Env::ACT; ++k) {
            act[k] = __builtin_fmaf(z[k], std_[k], mean[k]);  // rnd * exp(log_std) + mean
            a.actions[k * plane + off] = act[k];
            a.means[k * plane + off] = mean[k];
        }

        float r;
        bool d;
        Env::template step<float>(s, act, a.normalize, o, r, d);
        ts += 1;
        if (a.max_path_length > 0 && ts >= a.max_path_length) d = true;
        a.rewards[off] = r * a.scale_reward;
        a.dones[off] = d ? 1 : 0;
        if (d) {
            const float* dr = a.reset_draws ? a.reset_draws + (size_t)(t + 1) * draws_slice : nullptr;
            reset_one<Env>(s, dr, n, i, a.seed, env_global, a.step_counter + (uint64_t)t + 1);
            Env::template observe<float>(s, o);
            ts = 0;
        }
    }
    store_state<Env>(a.state, n, i, s);
    a.ts[i] = ts;
    if (a.last_obs) {
#pragma unroll
        for (int k = 0; k < Env::OBS; ++k) a.last_obs[(size_t)k * n + i] = o[k];
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
template <class Env>
static int launch_reset(int n, float* state, int32_t* ts, const uint8_t* mask, const float* draws,
                        uint64_t seed, uint64_t step, int env_offset, float* obs, hipStream_t st) {
    dim3 grid((n + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(vecenv_reset_kernel<Env>, grid, dim3(BLOCK), 0, st, n, state, ts, mask, draws, seed,
                       step, env_offset, obs);
    return check_launch("vecenv_reset_kernel");
}

template <class Env>
static int launch_step(int n, int normalize, float scale_reward, int mpl, int auto_reset, float* state,
                       int32_t* ts,
                       const float* actions, const float* reset_draws, uint64_t seed, uint64_t step,
                       int env_offset, float* obs, float* reward, uint8_t* done, hipStream_t st) {
    dim3 grid((n + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(vecenv_step_kernel<Env>, grid, dim3(BLOCK), 0, st, n, normalize, scale_reward, mpl,
                       auto_reset, state, ts, actions, reset_draws, seed, step, env_offset, obs, reward, done);
    return check_launch("vecenv_step_kernel");
}

template <class Env>
static int launch_rollout(const rl_rollout_args* g, hipStream_t st) {
    RolloutDev a;
    a.n = g->n_envs; a.T = g->horizon; a.max_path_length = g->max_path_length;
    a.normalize = g->normalize; a.reset_at_start = g->reset_at_start; a.env_offset = g->env_offset;
    a.scale_reward = g->scale_reward; a.log_min_std = g->log_min_std;
    a.seed = g->seed; a.step_counter = g->step_counter;
    a.state = g->state; a.ts = g->ts; a.theta = g->theta; a.eps = g->eps; a.reset_draws = g->reset_draws;
    a.obs = g->obs; a.actions = g->actions; a.means = g->means; a.rewards = g->rewards; a.dones = g->dones;
    a.last_obs = g->last_obs;
    dim3 grid((a.n + BLOCK - 1) / BLOCK);
    if (g->hidden0 == 32 && g->hidden1 == 32) {
        hipLaunchKernelGGL((rollout_kernel<Env, 32, 32>), grid, dim3(BLOCK), 0, st, a);
    } else if (g->hidden0 == 64 && g->hidden1 == 64) {
        hipLaunchKernelGGL((rollout_kernel<Env, 64, 64>), grid, dim3(BLOCK), 0, st, a);
    } else {
        return set_error(RL_ERR_UNSUPPORTED,
                         "rl_rollout_gaussian_mlp: hidden sizes (%d,%d) have no fused kernel "
                         "(built: 32x32, 64x64); use the per-step rl_vecenv_step path",
                         g->hidden0, g->hidden1);
    }
    return check_launch("rollout_kernel");
}

}  // namespace rl

using namespace rl;

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

extern "C" int rl_env_action_bounds(int kind, float* lb, float* ub) {
    if (!lb || !ub) return set_error(RL_ERR_ARG, "rl_env_action_bounds: null output");
#define Q (E::template action_bounds<float>(lb, ub), (int)RL_OK)
    RL_DISPATCH_ENV(kind, Q)
#undef Q
}

extern "C" int rl_vecenv_reset(int kind, int n, float* state, int32_t* ts, const uint8_t* mask,
                               const float* draws, uint64_t seed, uint64_t step_counter, int env_offset,
                               float* obs, void* stream) {
    if (n <= 0 || !state || !ts || !obs) return set_error(RL_ERR_ARG, "rl_vecenv_reset: bad argument");
    RL_DISPATCH_ENV(kind, launch_reset<E>(n, state, ts, mask, draws, seed, step_counter, env_offset, obs,
                                          (hipStream_t)stream))
}

extern "C" int rl_vecenv_step(int kind, int n, int normalize, float scale_reward, int max_path_length,
                              int auto_reset, float* state, int32_t* ts, const float* actions, const float* reset_draws,
                              uint64_t seed, uint64_t step_counter, int env_offset, float* obs,
                              float* reward, uint8_t* done, void* stream) {
    if (n <= 0 || !state || !ts || !actions || !obs || !reward || !done)
        return set_error(RL_ERR_ARG, "rl_vecenv_step: bad argument");
    RL_DISPATCH_ENV(kind, launch_step<E>(n, normalize, scale_reward, max_path_length, auto_reset, state, ts,
                                         actions,
                                         reset_draws, seed, step_counter, env_offset, obs, reward, done,
                                         (hipStream_t)stream))
}

extern "C" int rl_rollout_gaussian_mlp(const rl_rollout_args* g, void* stream) {
    if (!g) return set_error(RL_ERR_ARG, "rl_rollout_gaussian_mlp: null args");
    if (g->n_envs <= 0 || g->horizon <= 0 || !g->state || !g->ts || !g->theta || !g->obs || !g->actions ||
        !g->means || !g->rewards || !g->dones)
        return set_error(RL_ERR_ARG, "rl_rollout_gaussian_mlp: bad argument");
    RL_DISPATCH_ENV(g->kind, launch_rollout<E>(g, (hipStream_t)stream))
}

extern "C" int rl_debug_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                               uint32_t k1, int count, uint32_t* out, void* stream) {
    if (count <= 0 || !out) return set_error(RL_ERR_ARG, "rl_debug_philox: bad argument");
    hipLaunchKernelGGL(philox_debug_kernel, dim3((count + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       c0, c1, c2, c3, k0, k1, count, out);
    return check_launch("philox_debug_kernel");
}
