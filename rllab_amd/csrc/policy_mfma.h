// policy_mfma.h -- GaussianMLPPolicy mean network on v_mfma_f32_32x32x2_f32: the pieces shared by
// the update kernels (policy_kernels.hip) and the fused rollout (env_kernels.hip).
//
// Every dense layer is evaluated TRANSPOSED,  Z^T[unit][sample] = W^T[unit][k] * X^T[k][sample],
// so the MFMA output fragment (lane = sample + 32*half, register r = unit
// u(r, half) = (r&3) + 8*(r>>2) + 4*half) is, after the element-wise tanh, directly the B operand
// of the next layer: k-step m of that layer multiplies register m of this one, and the weight
// fragment staged in LDS for step m holds rows u(m, half) of W.  Layer 0's bias rides in the
// spare input slot (x[DO] = 1), layer 1's bias initialises the accumulator.
#pragma once
#include <hip/hip_runtime.h>

namespace rl {

constexpr int WV = 64;        // wavefront
constexpr int RL_SPLIT_NOT_TAKEN = 1 << 20;   // split_fvp_dispatch: this launch belongs to policy_pass_kernel
constexpr int TS = 32;        // samples per MFMA tile

using f32x16 = __attribute__((ext_vector_type(16))) float;

// tanh(x) = 1 - 2 / (e^{2x} + 1): exp2, add, rcp, fma.  Saturates correctly without a clamp
// (e -> inf gives 1, e -> 0 gives -1); absolute error < 2e-7, inside the 1e-5 policy tolerance.
__device__ __forceinline__ float ftanh(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);   // e^{2x}
    return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}

// Hidden activation of a layer, chosen at RUN time by a wave-uniform code (rl_activation: 0 tanh, 1 rectify, 2 identity):
// a policy with a rectify hidden_nonlinearity, or with ONE hidden layer (its kernel copy carries an identity second layer,
// policies/kernel_layout.py), runs on the kernels built for the tanh nets -- rllab/policies/gaussian_mlp_policy.py:21-69 and
// rllab/core/network.py:36-101 are free-form in both.  The branch is uniform, the tanh path is the instruction stream it
// always was (same bits), the derivative is expressed through the activation itself.
__device__ __forceinline__ void act_frag(f32x16& h, const f32x16& z, int code) {
    if (code == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) h[r] = ftanh(z[r]);
    } else if (code == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) h[r] = fmaxf(z[r], 0.0f);
    } else {
        h = z;
    }
}
// out = v * d act / dz  at activation value h
__device__ __forceinline__ void act_bwd_frag(f32x16& out, const f32x16& v, const f32x16& h, int code) {
    if (code == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) out[r] = v[r] * (1.0f - h[r] * h[r]);
    } else if (code == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) out[r] = h[r] > 0.0f ? v[r] : 0.0f;
    } else {
        out = v;
    }
}
__device__ __forceinline__ float act_one(float z, int code) { return code == 0 ? ftanh(z) : code == 1 ? fmaxf(z, 0.0f) : z; }
// per-layer codes of a batch / rollout: 0 = every hidden layer uses `activation`; else 2 bits per layer holding code + 1
__host__ __device__ inline int layer_act(int activation, int layer_activations, int l) {
    const int f = (layer_activations >> (2 * l)) & 3;
    return (layer_activations == 0 || f == 0) ? activation : f - 1;
}

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// unit held by register r of the 32x32 output fragment in lane half `half`
__host__ __device__ constexpr int frag_unit(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// compiler-level ordering of wave-private LDS traffic between lanes (the LDS queue of one
// wavefront is in order, so no s_barrier is needed -- only the compiler must not move
// accesses across the hand-over)
__device__ __forceinline__ void wave_sync() {
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    __builtin_amdgcn_wave_barrier();
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
}

template <int DO_, int DA_, int H_>
struct Net {
    static constexpr int DO = DO_, DA = DA_, H = H_;
    static constexpr int HT = H / 32;                 // 32-unit tiles per hidden layer
    static constexpr int KS0 = (DO + 2) / 2;          // k-steps of layer 0 (inputs + the bias slot)
    static constexpr int KS1 = 16 * HT;               // k-steps of layer 1
    static constexpr int W0 = 0;
    static constexpr int B0 = W0 + DO * H;
    static constexpr int W1 = B0 + H;
    static constexpr int B1 = W1 + H * H;
    static constexpr int W2 = B1 + H;
    static constexpr int B2 = W2 + H * DA;
    static constexpr int LSTD = B2 + DA;
    static constexpr int P = LSTD + DA;
    static constexpr int TAIL = P - B1;               // b1, W2, b2, log_std: VALU-side parameters
    static constexpr int TAILP = (TAIL + 3) & ~3;
    static constexpr int TSTR = H + 1;                // transposition tile stride (odd)
    static constexpr int XS = (2 * KS0) | 1;          // x tile stride (odd)
    static constexpr int GS = DA | 1;                 // gmu tile stride (odd)
    static constexpr int FA0 = HT * KS0 * WV;         // floats per layer-0 weight fragment set
    static constexpr int FA1 = HT * KS1 * WV;         // floats per layer-1 weight fragment set
    static constexpr int WAVE_LDS = TS * TSTR + TS * XS + TS * GS;
    // wavefronts per SIMD the register budget is declared for (2 x 256 or 1 x 512 registers)
    // (-DRL_POLICY_FORCE_WPS1=1: one wavefront per SIMD for every net -- an A/B knob, profiles/r03_notes.md)
#ifndef RL_POLICY_FORCE_WPS1
#define RL_POLICY_FORCE_WPS1 0
#endif
    static constexpr int WPS = (HT == 1 && DO <= 13 && !RL_POLICY_FORCE_WPS1) ? 2 : 1;
    // wavefronts per workgroup of the update passes.  -DRL_POLICY_WAVES8=1: the nets that run two wavefronts per SIMD
    // take them from ONE workgroup of eight (one staging of the weight fragments and one partial row per CU instead of
    // two) -- measured neutral on MI355X (FVP 0.311 vs 0.3125 ms, gradient 0.220 vs 0.217 ms at 2.048 M samples,
    // profiles/r03_notes.md): staging and the row count are not what bounds these passes.  Off.
#ifndef RL_POLICY_WAVES8
#define RL_POLICY_WAVES8 0
#endif
    static constexpr int WAVES = (WPS == 2 && RL_POLICY_WAVES8) ? 8 : 4;
    // the update kernels can keep the hidden activations of a batch in HBM between the gradient and the FVP passes
    static constexpr bool ACT_CACHE = true;
    // how the cached fragments of the NEXT tile travel to the FVP pass: LDS-direct loads into a wave-private landing
    // zone (32-unit nets at two wavefronts per SIMD: no registers to spare), or plain loads into registers (64-unit
    // nets run one wavefront per SIMD with 512 registers, and their landing zones would not fit LDS)
    static constexpr bool ACT_LDS_PREFETCH = (HT == 1);
    static_assert(H % 32 == 0 && HT <= 2, "hidden size must be 32 or 64");
    static_assert(DO + 1 <= 32, "obs_dim + 1 must fit one 32-row tile");

    // k index (unit of the previous layer) that lane half `half` contributes at k-step m of layer 1
    __host__ __device__ static constexpr int k1(int m, int half) { return 32 * (m / 16) + frag_unit(m % 16, half); }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WV);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WV);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, WV));
    return v;
}
// value + the value held by the same sample / unit in the other lane half
__device__ __forceinline__ float half_sum(float v) { return v + __shfl_xor(v, 32, WV); }

// stage one parameter vector as MFMA A-operand fragments (see the file header)
template <class N, int NT>
__device__ __forceinline__ void stage_fragments(const float* __restrict__ th, float* fa0, float* fa1,
                                                float* fa1t) {
    constexpr int H = N::H;
    for (int e = threadIdx.x; e < N::FA0; e += NT) {
        const int l = e % WV, m = (e / WV) % N::KS0, t = e / (WV * N::KS0);
        const int i = 32 * t + (l & 31), d = 2 * m + (l >> 5);
        fa0[e] = d < N::DO ? th[N::W0 + d * H + i] : (d == N::DO ? th[N::B0 + i] : 0.0f);
    }
    for (int e = threadIdx.x; e < N::FA1; e += NT) {
        const int l = e % WV, m = (e / WV) % N::KS1, t = e / (WV * N::KS1);
        const int i = 32 * t + (l & 31), k = N::k1(m, l >> 5);
        fa1[e] = th[N::W1 + k * H + i];                     // A[i][k] = W1[k][i]   (forward: W1^T)
        if (fa1t) fa1t[e] = th[N::W1 + i * H + k];          // A[i][k] = W1[i][k]   (backward: W1)
    }
}

}  // namespace rl
