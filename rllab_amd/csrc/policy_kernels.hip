// policy_kernels.hip -- fused TRPO/VPG update kernels for GaussianMLPPolicy
// (tanh MLP mean with two equal hidden layers + state-independent log_std) on the
// gfx950 matrix cores.
//
// One pass over the dense batch per launch; what the reference evaluates as the
// compiled Theano functions f_loss / f_constraint / f_loss_constraint, f_grad and
// f_Hx_plain (rllab/optimizers/conjugate_gradient_optimizer.py:27-46,194-215) on the
// surrogate loss and mean KL of rllab/algos/npo.py:72-82 with
// rllab/distributions/diagonal_gaussian.py:14-69:
//   MODE_LOSS : sum_b w_b * lr_b * adv_b ,  sum_b w_b * KL_b ,  sum_b w_b * logp_b * adv_b ,
//               max_b KL_b                                (f_loss_constraint, vpg f_kl)
//   MODE_GRAD : d/dtheta of  -sum_b w_b lr_b adv_b * inv_count          (f_grad)
//   MODE_VPG  : d/dtheta of  -sum_b w_b logp_b adv_b * inv_count        (vpg.py:91)
//   MODE_FVP  : Fisher-vector product  F v = J^T diag(2/(2 sigma^2+1e-8)) J v * inv_count
//               (+ the log_std block), which equals the Hessian of the mean KL at
//               theta_new == theta_old -- the only point where TRPO evaluates it
//               (PerlmutterHvp, :27-55; reg_coeff * v is added by the caller).
//
// Mapping (v_mfma_f32_32x32x2_f32: exact f32, D[32x32] += A[32x2] B[2x32]).
// A wavefront owns tiles of 32 samples.  Every dense layer is evaluated TRANSPOSED,
//     Z^T[unit][sample] = W^T[unit][k] * X^T[k][sample],
// so that the MFMA output fragment (lane = sample + 32*half, register r = unit
// u(r, half) = (r&3) + 8*(r>>2) + 4*half) is, after the element-wise tanh, directly the B
// operand of the next layer: k-step m of that layer multiplies register m of this one, and
// the weight fragment staged in LDS for step m holds rows u(m, half) of W.  Activations
// therefore never leave registers between layers and there is no per-FMA weight fetch:
// each MFMA (2048 FMAs) costs one conflict-free ds_read_b32 of its weight fragment.
// The same holds for the tangent pass (J v) and the back-propagation W1 * gz1.
// The batch reduction of outer products gW1 = sum_s h0_s (x) gz1_s is an MFMA with the
// SAMPLE axis as K: both operands are transposed through a wave-private LDS tile
// (stride H+1, conflict-free both ways).  The thin products (gW0: DO+1 rows, gW2: DA
// columns) would waste most of a 32x32 tile and run on the VALU against LDS-broadcast
// operands instead, filling issue slots next to the matrix pipe.
// Persistent accumulators live in registers for the whole grid-stride loop; the waves of a
// workgroup then fold their partials in a fixed order through LDS, every workgroup writes
// ONE partial row, and reduce_rows_kernel sums the rows in float64 in a fixed order
// (deterministic, identical on all ranks).
//
// Roofline: MFMA-bound.  FVP = 94 MFMAs (H = 32, swimmer) per 32-sample tile = 6016 matrix
// cycles per SIMD; HBM sees each sample's 80 B once per pass.
#include <hip/hip_runtime.h>
#include "../../include/rllab_amd.h"
#include "capi_util.h"
#include "cg_device.h"
#include "policy_mfma.h"

namespace rl {


// MODE_OUT / MODE_OUT_TAN / MODE_BWD: the mean network as a plain function on planes -- forward (and tangent) values
// OUT to [DA][B] planes, and back-propagation of an externally supplied output cotangent -- for policies whose
// distribution head is not the one fused here (adaptive_std: mean and log_std come from two networks,
// rllab/policies/gaussian_mlp_policy.py:60-98; rl_mlp_forward / rl_mlp_backward + rl_gaussian_head).
enum { MODE_LOSS = 0, MODE_GRAD = 1, MODE_FVP = 2, MODE_VPG = 3, MODE_OUT = 4, MODE_OUT_TAN = 5, MODE_BWD = 6 };
constexpr int LOSS_COLS = 4;  // sum w*lr*adv, sum w*kl, sum w*logp*adv, max kl

template <class N, int MODE, bool CACHE = false>
struct Smem {
    static constexpr bool GRADLIKE = (MODE == MODE_GRAD || MODE == MODE_FVP || MODE == MODE_VPG || MODE == MODE_BWD);
    static constexpr bool FVP = (MODE == MODE_FVP);
    static constexpr bool TAN = (MODE == MODE_FVP || MODE == MODE_OUT_TAN);      // tangent fragments staged
    static constexpr int ACT_FLOATS = 2 * N::H * TS;              // h0 | h1 fragments of one 32-sample tile
    static constexpr int A0 = 0;
    static constexpr int A1 = A0 + N::FA0;
    static constexpr int A1T = A1 + N::FA1;                       // backward fragments (W1 untransposed)
    static constexpr int DA0 = A1T + (GRADLIKE ? N::FA1 : 0);     // tangent fragments
    static constexpr int DA1 = DA0 + (TAN ? N::FA0 : 0);
    static constexpr int TAIL = DA1 + (TAN ? N::FA1 : 0);
    static constexpr int DTAIL = TAIL + N::TAILP;
    static constexpr int WAVE0 = DTAIL + (TAN ? N::TAILP : 0);
    static constexpr int WAVES = N::WAVES;
    static constexpr int ACTQ = WAVE0 + WAVES * N::WAVE_LDS;      // [WAVES][ACT_FLOATS] landing zone (cached FVP)
    static constexpr int TOTAL = ACTQ + ((FVP && CACHE && N::ACT_LDS_PREFETCH) ? WAVES * ACT_FLOATS : 0);
    static constexpr int RED = 0;                                 // [P] cross-wave fold, aliases the fragments
    static_assert(TOTAL >= N::P, "LDS fold buffer must fit");
};

struct PolicyBatch {
    int B;                     // samples
    const float* theta;        // [P]
    const float* vec;          // [P] tangent (MODE_FVP) or null
    float* acts;               // hidden-activation cache (CACHE variants): written by MODE_GRAD, read by MODE_FVP
    const float* obs;          // [DO][B]
    const float* act;          // [DA][B]
    const float* adv;          // [B]
    const float* old_mean;     // [DA][B]
    const float* old_log_std;  // [DA]
    const float* weight;       // [B] 0/1
    float inv_count;
    float log_min_std;
    float kl_penalty;          // MODE_GRAD / MODE_VPG: the gradient gains  + kl_penalty * d(sum w KL(old || new)) / dtheta
                               // (PenaltyLbfgsOptimizer's objective: PPO on the surrogate, regressors on the log-likelihood)
    const float* cot;          // MODE_BWD: [DA][B] cotangent on the network output (weights / normalisation included)
    float* out_mean;           // MODE_OUT / MODE_OUT_TAN: [DA][B] network output
    float* out_dmean;          // MODE_OUT_TAN: [DA][B] tangent of the output in direction vec
    float* partial;            // [grid][P]          (grad-like modes)
    double* partial_loss;      // [grid][LOSS_COLS]  (MODE_LOSS; MODE_GRAD: optional, null = gradient only)
    const int* gate;           // MODE_LOSS: rl_policy_batch.gate -- non-zero word: the launch returns at once
    unsigned* obs_absmax;      // MODE_GRAD with the cache: max |obs| of the batch as the bits of a float (atomic max; zeroed by
                               // the launcher), for the f16 split product's scales (policy_splith_kernels.hip); null = not asked
    int act0, act1;            // activation codes of the two hidden layers (rl_activation; RELU instantiations ignore them)
};

// hidden nonlinearity: tanh (GaussianMLPPolicy, network.py:38-39 default) or rectify (GaussianMLPRegressor /
// GaussianMLPBaseline, gaussian_mlp_regressor.py:31); the derivative is expressed through the activation itself
template <bool RELU> __device__ __forceinline__ float act_fn(float z) { return RELU ? fmaxf(z, 0.0f) : ftanh(z); }
template <bool RELU> __device__ __forceinline__ float act_dz(float h) {
    return RELU ? (h > 0.0f ? 1.0f : 0.0f) : (1.0f - h * h);
}

// ACTS = false: tanh layers (RELU: the regressors' compile-time rectify) -- the instruction stream of every earlier round;
// ACTS = true: the hidden activations are PolicyBatch.act0 / act1 at run time (rectify policies, the identity layer of a
// one-hidden-layer policy)
// wavefronts per SIMD of a pass: the loss pass is a forward + the distribution head -- no accumulators, a third of the
// registers of the gradient / product passes (92 for the (32, 32) nets, 128 with a dozen spilled for (64, 64)) -- and hides
// its matrix-result and transcendental latencies behind more wavefronts: 157 -> 138 us per pass over 2.048 M samples of the
// headline net, 122 -> 104 us over C5's 512 k (RL_POLICY_LOSS_WPS; A/B of both libraries: tools/exp/r05_call19.sh)
#ifndef RL_POLICY_LOSS_WPS
#define RL_POLICY_LOSS_WPS 4
#endif
template <class N, int MODE>
constexpr int pass_wps() { return (MODE == MODE_LOSS && N::WPS < RL_POLICY_LOSS_WPS) ? RL_POLICY_LOSS_WPS : N::WPS; }

template <class N, int MODE, bool CACHE, bool RELU = false, bool ACTS = false>
__global__ void __launch_bounds__(N::WAVES * WV, (pass_wps<N, MODE>())) policy_pass_kernel(PolicyBatch a) {
    constexpr int WAVES = N::WAVES;
    static_assert(!CACHE || MODE == MODE_GRAD || MODE == MODE_FVP, "activation cache: grad writes, FVP reads");
    static_assert(!RELU || MODE == MODE_LOSS || MODE == MODE_VPG, "rectify nets: loss and log-likelihood gradient");
    using S = Smem<N, MODE, CACHE>;
    constexpr int DO = N::DO, DA = N::DA, H = N::H, HT = N::HT, KS0 = N::KS0, KS1 = N::KS1, P = N::P;
    constexpr bool GRADLIKE = S::GRADLIKE, FVP = S::FVP, TAN = S::TAN;
    constexpr bool OUTMODE = (MODE == MODE_OUT || MODE == MODE_OUT_TAN), BWD = (MODE == MODE_BWD);
    constexpr bool HEAD = (MODE == MODE_LOSS || MODE == MODE_GRAD || MODE == MODE_VPG);   // the fused distribution head
    constexpr bool LOAD_ACTS = CACHE && FVP, STORE_ACTS = CACHE && !FVP;
    constexpr int TSTR = N::TSTR, XS = N::XS, GS = N::GS;
    // activation cache: per 32-sample tile, [h0 | h1][HT][4] rows of 64 lanes x float4 (registers 4q .. 4q+3 of a
    // fragment), i.e. the fragments exactly as the matrix pipe produced them -- 1 KB per wavefront access
    constexpr int ACT_ROWS = 2 * HT * 4;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (MODE == MODE_LOSS) {
        // a line-search candidate enqueued behind an accepted one (rl_line_search_decide): nothing to evaluate
        if (a.gate != nullptr && *a.gate != 0) return;
    }
    const int wave = threadIdx.x / WV, lane = threadIdx.x % WV;
    const int lj = lane & 31, lh = lane >> 5;
    const int c0 = RELU ? 1 : (ACTS ? a.act0 : 0), c1 = RELU ? 1 : (ACTS ? a.act1 : 0);   // hidden activations (wave-uniform codes)
    float* const fa0 = smem + S::A0;
    float* const fa1 = smem + S::A1;
    float* const fa1t = smem + S::A1T;
    float* const fda0 = smem + S::DA0;
    float* const fda1 = smem + S::DA1;
    float* const tail = smem + S::TAIL;      // theta[B1 .. P)
    float* const dtail = smem + S::DTAIL;    // vec[B1 .. P)
    float* const tb = smem + S::WAVE0 + wave * N::WAVE_LDS;   // [32][TSTR] transposition tile
    float* const tbx = tb + TS * TSTR;                        // [32][XS]   x (+1) rows
    float* const tbg = tbx + TS * XS;                         // [32][GS]   gmu rows

    stage_fragments<N, WAVES * WV>(a.theta, fa0, fa1, GRADLIKE ? fa1t : nullptr);
    if (TAN) stage_fragments<N, WAVES * WV>(a.vec, fda0, fda1, nullptr);
    for (int k = threadIdx.x; k < N::TAIL; k += WAVES * WV) {
        tail[k] = a.theta[N::B1 + k];
        if (TAN) dtail[k] = a.vec[N::B1 + k];
    }
    for (int k = lane; k < N::WAVE_LDS; k += WV) tb[k] = 0.0f;
    __syncthreads();
    constexpr int T_B1 = 0, T_W2 = N::W2 - N::B1, T_B2 = N::B2 - N::B1, T_LS = N::LSTD - N::B1;

    // effective log_std / std (state independent)
    float lstd[DA], inv_std[DA], var_[DA];
    bool floored[DA];
#pragma unroll
    for (int k = 0; k < DA; ++k) {
        const float raw = tail[T_LS + k];
        floored[k] = raw < a.log_min_std;
        lstd[k] = fmaxf(raw, a.log_min_std);
        inv_std[k] = __expf(-lstd[k]);
        var_[k] = __expf(2.0f * lstd[k]);
    }

    // ---- accumulators (registers, whole launch) ------------------------------------------
    double acc_loss = 0.0, acc_kl = 0.0, acc_vpg = 0.0;
    float max_kl = -INFINITY;
    f32x16 gW1[HT][HT];                 // [row tile][col tile] fragments of sum h0 (x) gz1
    f32x16 gW0[HT];                     // fragments of sum x_ext (x) gz0: rows d <= DO (DO = bias row) used
    // narrow inputs (DO + 1 <= 16 rows): the same product on 16x16x4 tiles -- half the matrix passes, half the
    // accumulator registers (rows d = 4 (lane / 16) + j, column unit 16 nt + lane % 16)
    constexpr bool W0_NARROW = (DO + 1 <= 16);
    constexpr int NT16 = H / 16;
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    f32x4_t gW0n[NT16];
    float gW2[HT][DA];                  // lane = row (unit), half = sample parity
    float gb1[HT], gb2[DA], gls[DA];
    float wsum = 0.0f;
#pragma unroll
    for (int ti = 0; ti < HT; ++ti) {
#pragma unroll
        for (int tj = 0; tj < HT; ++tj)
#pragma unroll
            for (int r = 0; r < 16; ++r) gW1[ti][tj][r] = 0.0f;
        gb1[ti] = 0.0f;
#pragma unroll
        for (int k = 0; k < DA; ++k) gW2[ti][k] = 0.0f;
    }
#pragma unroll
    for (int tj = 0; tj < HT; ++tj)
#pragma unroll
        for (int r = 0; r < 16; ++r) gW0[tj][r] = 0.0f;
#pragma unroll
    for (int nt = 0; nt < NT16; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) gW0n[nt][j] = 0.0f;
#pragma unroll
    for (int k = 0; k < DA; ++k) { gb2[k] = 0.0f; gls[k] = 0.0f; }

    const int B = a.B;
    const int n_tiles = (B + TS - 1) / TS;
    const int wave_global = blockIdx.x * WAVES + wave;
    const int waves_total = gridDim.x * WAVES;

    // Per-sample inputs of a tile (observation slots, weight) are fetched one tile AHEAD: the HBM
    // round trip (~1 us) of tile t+1 overlaps the matrix chains of tile t instead of stalling the
    // head of every tile.
    auto fetch = [&](int tile, float* xq, float& wq) {
        const int b = tile * TS + lj;
        const bool live = b < B;
        const int bi = live ? b : (B - 1);
        wq = live ? a.weight[bi] : 0.0f;
#pragma unroll
        for (int m = 0; m < KS0; ++m) {
            const int d = 2 * m + lh;
            xq[m] = d < DO ? a.obs[(size_t)d * B + bi] : (d == DO ? 1.0f : 0.0f);
        }
    };
    float xb[KS0], xb_next[KS0];
    float wgt = 0.0f, wgt_next = 0.0f;
    float xabs = 0.0f;                          // (cache-writing gradient pass only) running max |x| of this lane
    if (wave_global < n_tiles) fetch(wave_global, xb_next, wgt_next);
    // cached activations travel one tile ahead as well (FVP only), by LDS-direct loads into a wave-private landing
    // zone: 8 KB per tile would not fit the register budget of two wavefronts per SIMD next to the accumulators
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    float* const hq = smem + S::ACTQ + wave * S::ACT_FLOATS;
    auto fetch_acts = [&](int tile) {
        const float* src = a.acts + ((size_t)tile * ACT_ROWS * WV + lane) * 4;
#pragma unroll
        for (int q = 0; q < ACT_ROWS; ++q)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + q * WV * 4), (lptr_t)(hq + q * WV * 4), 16, 0, 0);
    };
    // register form of the same prefetch (64-unit nets)
    constexpr bool ACTS_VIA_LDS = LOAD_ACTS && N::ACT_LDS_PREFETCH, ACTS_VIA_REGS = LOAD_ACTS && !N::ACT_LDS_PREFETCH;
    f32x4 act_next[ACTS_VIA_REGS ? ACT_ROWS : 1];
    auto fetch_acts_regs = [&](int tile) {
        const f32x4* src = reinterpret_cast<const f32x4*>(a.acts) + (size_t)tile * ACT_ROWS * WV + lane;
#pragma unroll
        for (int q = 0; q < ACT_ROWS; ++q) act_next[q] = __builtin_nontemporal_load(src + q * WV);
    };
    if (ACTS_VIA_LDS && wave_global < n_tiles) fetch_acts(wave_global);
    if constexpr (ACTS_VIA_REGS) {
        if (wave_global < n_tiles) fetch_acts_regs(wave_global);
    }

    for (int tile = wave_global; tile < n_tiles; tile += waves_total) {
        asm volatile("" ::: "memory");   // keep the weight-fragment reads inside the loop
        const int b = tile * TS + lj;
        const bool live = b < B;
        const int bi = live ? b : (B - 1);
        // ---- B operands of layer 0: x_ext[sample][2m + half] (slot DO = 1 carries the bias) -----
#pragma unroll
        for (int m = 0; m < KS0; ++m) xb[m] = xb_next[m];
        wgt = wgt_next;
        if (tile + waves_total < n_tiles) fetch(tile + waves_total, xb_next, wgt_next);

        // ---- forward (or the fragments the gradient pass left in HBM) ------------------------------
        f32x16 h0[HT], h1[HT];
        if constexpr (ACTS_VIA_REGS) {
#pragma unroll
            for (int t = 0; t < HT; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        h0[t][4 * q + e] = act_next[t * 4 + q][e];
                        h1[t][4 * q + e] = act_next[(HT + t) * 4 + q][e];
                    }
            if (tile + waves_total < n_tiles) fetch_acts_regs(tile + waves_total);
        } else if constexpr (ACTS_VIA_LDS) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the tile fetched during the previous tile has landed
#pragma unroll
            for (int t = 0; t < HT; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(hq + ((t * 4 + q) * WV + lane) * 4);
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(hq + (((HT + t) * 4 + q) * WV + lane) * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { h0[t][4 * q + e] = v0[e]; h1[t][4 * q + e] = v1[e]; }
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // ... and is in registers before the zone is refilled
            if (tile + waves_total < n_tiles) fetch_acts(tile + waves_total);
        } else {
#pragma unroll
            for (int t = 0; t < HT; ++t) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
                for (int m = 0; m < KS0; ++m) acc = mfma(fa0[(t * KS0 + m) * WV + lane], xb[m], acc);
                act_frag(h0[t], acc, c0);
            }
#pragma unroll
            for (int t = 0; t < HT; ++t) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = tail[T_B1 + 32 * t + frag_unit(r, 0) + 4 * lh];
#pragma unroll
                for (int m = 0; m < KS1; ++m) acc = mfma(fa1[(t * KS1 + m) * WV + lane], h0[m / 16][m % 16], acc);
                act_frag(h1[t], acc, c1);
            }
            if constexpr (STORE_ACTS) {
#pragma unroll
                for (int m = 0; m < KS0; ++m) xabs = fmaxf(xabs, 2 * m + lh < DO ? fabsf(xb[m]) : 0.0f);   // (not the bias slot)
                f32x4* dst = reinterpret_cast<f32x4*>(a.acts) + (size_t)tile * ACT_ROWS * WV + lane;
#pragma unroll
                for (int t = 0; t < HT; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v0, v1;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v0[e] = h0[t][4 * q + e]; v1[e] = h1[t][4 * q + e]; }
                        __builtin_nontemporal_store(v0, dst + (t * 4 + q) * WV);
                        __builtin_nontemporal_store(v1, dst + ((HT + t) * 4 + q) * WV);
                    }
            }
        }

        // ---- per-sample cotangent on the mean ----------------------------------------------------
        float gmu[DA];
#pragma unroll
        for (int k = 0; k < DA; ++k) gmu[k] = 0.0f;
        float mean[DA];
        if constexpr (HEAD || OUTMODE) {
#pragma unroll
            for (int k = 0; k < DA; ++k) {
                float pm = 0.0f;
#pragma unroll
                for (int t = 0; t < HT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        pm = __builtin_fmaf(h1[t][r], tail[T_W2 + (32 * t + frag_unit(r, 0) + 4 * lh) * DA + k], pm);
                mean[k] = tail[T_B2 + k] + half_sum(pm);
            }
        }
        if constexpr (HEAD) {
            const float advb = a.adv[bi];
            float zz_new = 0.0f, zz_old = 0.0f, sls_new = 0.0f, sls_old = 0.0f, kl = 0.0f;
            float znew[DA], dmv[DA], numv[DA];          // the last two: pieces of the KL the penalty gradient reuses
#pragma unroll
            for (int k = 0; k < DA; ++k) {
                const float ak = a.act[(size_t)k * B + bi];
                const float mo = a.old_mean[(size_t)k * B + bi];
                const float lo = a.old_log_std[k];
                const float so = __expf(lo);
                znew[k] = (ak - mean[k]) * inv_std[k];
                const float zo = (ak - mo) / so;
                zz_new = __builtin_fmaf(znew[k], znew[k], zz_new);
                zz_old = __builtin_fmaf(zo, zo, zz_old);
                sls_new += lstd[k];
                sls_old += lo;
                const float dm = mo - mean[k];
                const float num = dm * dm + so * so - var_[k];
                const float den = 2.0f * var_[k] + 1e-8f;
                kl += num / den + lstd[k] - lo;
                dmv[k] = dm;
                numv[k] = num;
            }
            // logli_new - logli_old (diagonal_gaussian.py:56-69); the 0.5*Da*log(2 pi) terms cancel
            const float logp_new = -sls_new - 0.5f * zz_new;
            const float dlog = logp_new - (-sls_old - 0.5f * zz_old);
            const float lr = __expf(dlog);
            const float w1 = (lh == 0) ? wgt : 0.0f;        // both halves hold the sample: count it once
            // the gradient pass can hand back the loss / KL sums of the same forward pass (rl_policy_grad_loss)
            if (MODE == MODE_LOSS || ((MODE == MODE_GRAD || MODE == MODE_VPG) && a.partial_loss != nullptr)) {
                acc_loss += (double)(w1 * lr * advb);
                acc_kl += (double)(w1 * kl);
                acc_vpg += (double)(w1 * (logp_new - 0.5f * (float)DA * 1.8378770664093453f) * advb);
                if (w1 > 0.0f) max_kl = fmaxf(max_kl, kl);
            }
            if (MODE != MODE_LOSS) {
                // d(-w adv lr)/dmu_k = -w adv lr z_k / sigma_k ; VPG: lr -> 1 (d logp)
                const float c = -wgt * advb * (MODE == MODE_GRAD ? lr : 1.0f) * a.inv_count;
                const float c1 = (lh == 0) ? c : 0.0f;
#pragma unroll
                for (int k = 0; k < DA; ++k) {
                    gmu[k] = c * znew[k] * inv_std[k];
                    if (!floored[k]) gls[k] += c1 * (znew[k] * znew[k] - 1.0f);
                }
                if (a.kl_penalty != 0.0f) {     // wave-uniform: + kl_penalty * d(sum w KL(old || new) / W) / dtheta
                    const float p = a.kl_penalty * wgt * a.inv_count;
                    const float p1 = (lh == 0) ? p : 0.0f;
#pragma unroll
                    for (int k = 0; k < DA; ++k) {
                        // KL_k = num / den + ls_new - ls_old, num = (mu_o - mu)^2 + s_o^2 - v, den = 2 v + 1e-8, v = e^{2 ls_new}
                        const float den = 2.0f * var_[k] + 1e-8f;
                        const float dkl_mu = -2.0f * dmv[k] / den;
                        const float dkl_ls = 1.0f - (2.0f * var_[k] * den + 4.0f * var_[k] * numv[k]) / (den * den);
                        gmu[k] = __builtin_fmaf(p, dkl_mu, gmu[k]);
                        if (!floored[k]) gls[k] += p1 * dkl_ls;
                    }
                }
            }
        }
        float dmu_[DA];
        if constexpr (TAN) {
            // tangent forward: dmu = J v
            f32x16 dh0[HT], dh1[HT];
#pragma unroll
            for (int t = 0; t < HT; ++t) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
                for (int m = 0; m < KS0; ++m) acc = mfma(fda0[(t * KS0 + m) * WV + lane], xb[m], acc);   // dW0^T x + db0
                act_bwd_frag(dh0[t], acc, h0[t], c0);
            }
#pragma unroll
            for (int t = 0; t < HT; ++t) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = dtail[T_B1 + 32 * t + frag_unit(r, 0) + 4 * lh];
#pragma unroll
                for (int m = 0; m < KS1; ++m) {
                    acc = mfma(fda1[(t * KS1 + m) * WV + lane], h0[m / 16][m % 16], acc);                // dW1^T h0
                    acc = mfma(fa1[(t * KS1 + m) * WV + lane], dh0[m / 16][m % 16], acc);                // W1^T dh0
                }
                act_bwd_frag(dh1[t], acc, h1[t], c1);
            }
            const float c = wgt * a.inv_count;
#pragma unroll
            for (int k = 0; k < DA; ++k) {
                float pd = 0.0f;
#pragma unroll
                for (int t = 0; t < HT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int u = (32 * t + frag_unit(r, 0) + 4 * lh) * DA + k;
                        pd = __builtin_fmaf(h1[t][r], dtail[T_W2 + u], pd);
                        pd = __builtin_fmaf(dh1[t][r], tail[T_W2 + u], pd);
                    }
                const float dmu = dtail[T_B2 + k] + half_sum(pd);
                dmu_[k] = dmu;
                if constexpr (FVP) gmu[k] = c * dmu * (2.0f / (2.0f * var_[k] + 1e-8f));
            }
            if (FVP && lh == 0) wsum += c;
        }
        if constexpr (OUTMODE) {
            // the network as a function on planes: every sample's output (and tangent), nothing else
            if (live && lh == 0) {
#pragma unroll
                for (int k = 0; k < DA; ++k) {
                    a.out_mean[(size_t)k * B + b] = mean[k];
                    if constexpr (MODE == MODE_OUT_TAN) a.out_dmean[(size_t)k * B + b] = dmu_[k];
                }
            }
        }
        if constexpr (BWD) {
#pragma unroll
            for (int k = 0; k < DA; ++k) gmu[k] = live ? a.cot[(size_t)k * B + bi] : 0.0f;
        }

        if (GRADLIKE) {
            // ---- back-propagation: gz1 = (W2 gmu) (1 - h1^2); gz0 = (W1 gz1) (1 - h0^2) ----------------
            f32x16 gz1[HT], gz0[HT];
#pragma unroll
            for (int t = 0; t < HT; ++t) {
                f32x16 gh;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float g = 0.0f;
#pragma unroll
                    for (int k = 0; k < DA; ++k)
                        g = __builtin_fmaf(tail[T_W2 + (32 * t + frag_unit(r, 0) + 4 * lh) * DA + k], gmu[k], g);
                    gh[r] = g;
                }
                act_bwd_frag(gz1[t], gh, h1[t], c1);
            }
#pragma unroll
            for (int t = 0; t < HT; ++t) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
                for (int m = 0; m < KS1; ++m) acc = mfma(fa1t[(t * KS1 + m) * WV + lane], gz1[m / 16][m % 16], acc);
                act_bwd_frag(gz0[t], acc, h0[t], c0);
            }

            // ---- gb2, gmu / x rows for the broadcast (VALU) products -------------------------------------
            if (lh == 0) {
#pragma unroll
                for (int k = 0; k < DA; ++k) {
                    gb2[k] += gmu[k];
                    tbg[lj * GS + k] = gmu[k];
                }
            }
#pragma unroll
            for (int m = 0; m < KS0; ++m) tbx[lj * XS + 2 * m + lh] = xb[m];

            // ---- layer 1: gW1 += h0^T gz1 (samples are K), gb1 += column sums of gz1 -------------------
            // in the transposed role lane (c, half) owns unit c of a 32-unit tile and the samples of
            // parity `half`: operand m is sample 2m + half
            wave_sync();
#pragma unroll
            for (int t = 0; t < HT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) tb[lj * TSTR + 32 * t + frag_unit(r, 0) + 4 * lh] = gz1[t][r];
            wave_sync();
            float bop[HT][16];
#pragma unroll
            for (int t = 0; t < HT; ++t) {
                float s = 0.0f;
#pragma unroll
                for (int m = 0; m < 16; ++m) {
                    bop[t][m] = tb[(2 * m + lh) * TSTR + 32 * t + lj];
                    s += bop[t][m];
                }
                gb1[t] += s;
            }
            wave_sync();
#pragma unroll
            for (int t = 0; t < HT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) tb[lj * TSTR + 32 * t + frag_unit(r, 0) + 4 * lh] = h0[t][r];
            wave_sync();
#pragma unroll
            for (int ti = 0; ti < HT; ++ti) {
                float aop[16];
#pragma unroll
                for (int m = 0; m < 16; ++m) aop[m] = tb[(2 * m + lh) * TSTR + 32 * ti + lj];
#pragma unroll
                for (int tj = 0; tj < HT; ++tj)
#pragma unroll
                    for (int m = 0; m < 16; ++m) gW1[ti][tj] = mfma(aop[m], bop[tj][m], gW1[ti][tj]);
            }

            // ---- layer 2: gW2[u][k] += sum_s h1[s][u] gmu[s][k]  (lane = unit, gmu broadcast) ------------
            wave_sync();
#pragma unroll
            for (int t = 0; t < HT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) tb[lj * TSTR + 32 * t + frag_unit(r, 0) + 4 * lh] = h1[t][r];
            wave_sync();
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                float g[DA];
#pragma unroll
                for (int k = 0; k < DA; ++k) g[k] = tbg[(2 * m + lh) * GS + k];
#pragma unroll
                for (int t = 0; t < HT; ++t) {
                    const float hv = tb[(2 * m + lh) * TSTR + 32 * t + lj];
#pragma unroll
                    for (int k = 0; k < DA; ++k) gW2[t][k] = __builtin_fmaf(hv, g[k], gW2[t][k]);
                }
            }

            // ---- layer 0: gW0 += x_ext^T gz0 (samples are K).  Only DO+1 of the 32 rows are real (the x
            // tile's other columns hold stale finite values whose rows are never stored), but the matrix
            // pipe has the slack and the VALU does not: as 16 MFMAs this product costs 32 LDS reads
            // instead of ~450 VALU + LDS instructions.
            wave_sync();
#pragma unroll
            for (int t = 0; t < HT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) tb[lj * TSTR + 32 * t + frag_unit(r, 0) + 4 * lh] = gz0[t][r];
            wave_sync();
            if constexpr (W0_NARROW) {
                // A[m = input row][k = sample 4 s + kq], B[k = sample][n = unit]: lane = (m or n) + 16 kq.  Row 15 of
                // the x tile does not exist for XS = 15 (the read lands on the next sample's first input: finite, and
                // output row 15 is never stored).
                const int lm = lane & 15, kq = lane >> 4;
#pragma unroll
                for (int sp = 0; sp < 8; ++sp) {
                    const float av = tbx[(4 * sp + kq) * XS + lm];
#pragma unroll
                    for (int nt = 0; nt < NT16; ++nt)
                        gW0n[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, tb[(4 * sp + kq) * TSTR + 16 * nt + lm],
                                                                        gW0n[nt], 0, 0, 0);
                }
            } else {
                float ax[16];
#pragma unroll
                for (int m = 0; m < 16; ++m) ax[m] = tbx[(2 * m + lh) * XS + (lj < XS ? lj : 0)];
#pragma unroll
                for (int t = 0; t < HT; ++t)
#pragma unroll
                    for (int m = 0; m < 16; ++m)
                        gW0[t] = mfma(ax[m], tb[(2 * m + lh) * TSTR + 32 * t + lj], gW0[t]);
            }
            wave_sync();
        }
    }

    if constexpr (STORE_ACTS) {
        // a maximum is order-independent, and non-negative floats order like their bit patterns
        if (a.obs_absmax != nullptr) {
            xabs = wave_max(xabs);
            if (lane == 0) atomicMax(a.obs_absmax, __float_as_uint(xabs));
        }
    }
    // ---- fold the wavefronts of this workgroup in a fixed order, write ONE partial row --------
    __syncthreads();   // every wave is done with the weight fragments: the fold buffer aliases them
    auto fold_loss = [&]() {
        double* red = reinterpret_cast<double*>(smem);
        const double l = wave_sum(acc_loss), k = wave_sum(acc_kl), v = wave_sum(acc_vpg);
        const float mk = wave_max(max_kl);
        if (lane == 0) {
            red[wave * LOSS_COLS + 0] = l; red[wave * LOSS_COLS + 1] = k;
            red[wave * LOSS_COLS + 2] = v; red[wave * LOSS_COLS + 3] = (double)mk;
        }
        __syncthreads();
        if (threadIdx.x < LOSS_COLS) {
            const int c = threadIdx.x;
            double s = red[c];
            for (int w = 1; w < WAVES; ++w) s = (c == 3) ? fmax(s, red[w * LOSS_COLS + c]) : s + red[w * LOSS_COLS + c];
            a.partial_loss[(size_t)blockIdx.x * LOSS_COLS + c] = s;
        }
    };
    if (OUTMODE) {
        // nothing to fold: the planes are the result
    } else if (MODE == MODE_LOSS) {
        fold_loss();
    } else {
        float* red = smem + S::RED;
        for (int k = threadIdx.x; k < P; k += WAVES * WV) red[k] = 0.0f;
        __syncthreads();
        // half-pair sums (both sample parities of the transposed-role accumulators)
        float w2s[HT][DA], b1s[HT];
#pragma unroll
        for (int t = 0; t < HT; ++t) {
            b1s[t] = half_sum(gb1[t]);
#pragma unroll
            for (int k = 0; k < DA; ++k) w2s[t][k] = half_sum(gW2[t][k]);
        }
        float b2s[DA], lss[DA];
#pragma unroll
        for (int k = 0; k < DA; ++k) { b2s[k] = wave_sum(gb2[k]); lss[k] = wave_sum(gls[k]); }
        const float ws = wave_sum(wsum);
        for (int w = 0; w < WAVES; ++w) {
            if (wave == w) {
#pragma unroll
                for (int ti = 0; ti < HT; ++ti)
#pragma unroll
                    for (int tj = 0; tj < HT; ++tj)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            red[N::W1 + (32 * ti + frag_unit(r, 0) + 4 * lh) * H + 32 * tj + lj] += gW1[ti][tj][r];
                if constexpr (W0_NARROW) {
                    const int lm = lane & 15, kq = lane >> 4;
#pragma unroll
                    for (int nt = 0; nt < NT16; ++nt)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int d = 4 * kq + j;                        // row of x_ext^T gz0
                            if (d < DO) red[N::W0 + d * H + 16 * nt + lm] += gW0n[nt][j];
                            else if (d == DO) red[N::B0 + 16 * nt + lm] += gW0n[nt][j];
                        }
                } else {
#pragma unroll
                    for (int t = 0; t < HT; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int d = frag_unit(r, 0) + 4 * lh;          // row of x_ext^T gz0
                            if (d < DO) red[N::W0 + d * H + 32 * t + lj] += gW0[t][r];
                            else if (d == DO) red[N::B0 + 32 * t + lj] += gW0[t][r];
                        }
                }
                if (lh == 0) {
#pragma unroll
                    for (int t = 0; t < HT; ++t) {
                        red[N::B1 + 32 * t + lj] += b1s[t];
#pragma unroll
                        for (int k = 0; k < DA; ++k) red[N::W2 + (32 * t + lj) * DA + k] += w2s[t][k];
                    }
                }
                if (lane == 0) {
#pragma unroll
                    for (int k = 0; k < DA; ++k) {
                        red[N::B2 + k] += b2s[k];
                        if (FVP) {
                            // log_std block of the Fisher: d2KL/ds2 = 4 v (2 v - eps) / (2 v + eps)^2, v = sigma^2
                            const float vv = var_[k], e = 1e-8f;
                            const float c = floored[k] ? 0.0f
                                                       : 4.0f * vv * (2.0f * vv - e) / ((2.0f * vv + e) * (2.0f * vv + e));
                            red[N::LSTD + k] += c * a.vec[N::LSTD + k] * ws;
                        } else {
                            red[N::LSTD + k] += lss[k];
                        }
                    }
                }
            }
            __syncthreads();
        }
        float* row = a.partial + (size_t)blockIdx.x * P;
        for (int k = threadIdx.x; k < P; k += WAVES * WV) row[k] = red[k];
        if ((MODE == MODE_GRAD || MODE == MODE_VPG) && a.partial_loss != nullptr) {
            __syncthreads();   // the fold buffer is read out
            fold_loss();
        }
    }
}

// out[c] = sum_r partial[r][c] in float64, fixed order (deterministic): a workgroup owns 64
// columns, 16 wavefronts take interleaved rows, their partials meet in LDS in wave order.
constexpr int RR_WAVES = 16;
__global__ void __launch_bounds__(RR_WAVES * WV) reduce_rows_kernel(const float* __restrict__ partial, int rows,
                                                                     int cols, double* __restrict__ out) {
    __shared__ double part[RR_WAVES][WV];
    const int wave = threadIdx.x / WV, lane = threadIdx.x % WV;
    const int c = blockIdx.x * WV + lane;
    double s = 0.0;
    if (c < cols) {
        // four independent partial sums: the loads of a wavefront's rows are in flight together instead of one
        // dependent load-add per row (fixed association, still deterministic)
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int r = wave;
        for (; r + 3 * RR_WAVES < rows; r += 4 * RR_WAVES) {
            s0 += (double)partial[(size_t)r * cols + c];
            s1 += (double)partial[(size_t)(r + RR_WAVES) * cols + c];
            s2 += (double)partial[(size_t)(r + 2 * RR_WAVES) * cols + c];
            s3 += (double)partial[(size_t)(r + 3 * RR_WAVES) * cols + c];
        }
        for (; r < rows; r += RR_WAVES) s0 += (double)partial[(size_t)r * cols + c];
        s = (s0 + s1) + (s2 + s3);
    }
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && c < cols) {
        double t = part[0][lane];
        for (int w = 1; w < RR_WAVES; ++w) t += part[w][lane];
        out[c] = t;
    }
}

// rl_policy_fvp_cg_step: the row reduction of a Fisher-vector product and the CG iteration that consumes it in ONE
// launch.  Every workgroup reduces its 64 columns exactly like reduce_rows_kernel and publishes them; the workgroup
// that takes the LAST ticket (all columns are then in memory: release fence before the ticket, acquire after) runs
// cg_step_body over the whole vector and rewinds the ticket counter for the next launch.  Same arithmetic, same
// order, as reduce_rows_kernel followed by cg_step_kernel.
struct CgArgs {
    double reg, tol;
    double* x;
    double* r;
    double* p;
    float* p32;
    double* scal;
    double* fp;              // [n] scratch: F p
    unsigned int* ticket;    // zero before the first launch; left zero by every launch
};
static_assert(RR_WAVES * WV == CG_THREADS, "the reducing workgroup doubles as the CG workgroup");
__global__ void __launch_bounds__(RR_WAVES * WV) reduce_rows_cg_kernel(const float* __restrict__ partial, int rows,
                                                                        int cols, CgArgs c) {
    __shared__ double part[RR_WAVES][WV];
    __shared__ double scratch[CG_THREADS / 64];
    __shared__ unsigned int my_ticket;
    const int wave = threadIdx.x / WV, lane = threadIdx.x % WV;
    const int col = blockIdx.x * WV + lane;
    double s = 0.0;
    if (col < cols) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int r = wave;
        for (; r + 3 * RR_WAVES < rows; r += 4 * RR_WAVES) {
            s0 += (double)partial[(size_t)r * cols + col];
            s1 += (double)partial[(size_t)(r + RR_WAVES) * cols + col];
            s2 += (double)partial[(size_t)(r + 2 * RR_WAVES) * cols + col];
            s3 += (double)partial[(size_t)(r + 3 * RR_WAVES) * cols + col];
        }
        for (; r < rows; r += RR_WAVES) s0 += (double)partial[(size_t)r * cols + col];
        s = (s0 + s1) + (s2 + s3);
    }
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && col < cols) {
        double t = part[0][lane];
        for (int w = 1; w < RR_WAVES; ++w) t += part[w][lane];
        c.fp[col] = t;
    }
    __threadfence();                       // this workgroup's columns are visible device-wide before its ticket
    __syncthreads();
    if (threadIdx.x == 0) my_ticket = atomicAdd(c.ticket, 1u);
    __syncthreads();
    if (my_ticket != gridDim.x - 1) return;
    __threadfence();                       // acquire: every other workgroup's columns
    cg_step_body(cols, c.fp, c.reg, c.tol, c.x, c.r, c.p, c.p32, c.scal, scratch);
    if (threadIdx.x == 0) *c.ticket = 0u;
}

// loss partials: columns 0..2 summed, column 3 maxed; one wavefront per column
__global__ void __launch_bounds__(LOSS_COLS * WV) reduce_loss_kernel(const double* __restrict__ partial, int rows,
                                                                     double* __restrict__ out) {
    const int c = threadIdx.x / WV, lane = threadIdx.x % WV;
    double s = (c == 3) ? -INFINITY : 0.0;
    for (int r = lane; r < rows; r += WV) {
        const double v = partial[(size_t)r * LOSS_COLS + c];
        s = (c == 3) ? fmax(s, v) : s + v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double v = __shfl_xor(s, o, WV);
        s = (c == 3) ? fmax(s, v) : s + v;
    }
    if (lane == 0) out[c] = s;
}

constexpr int MAX_GRID = 256 * 3;   // workgroups of a pass: <= 3 per CU

// the same fixed-order float64 reductions for the wide / deep nets' passes (policy_wide_kernels.hip)
int launch_reduce_rows(const float* partial, int rows, int cols, double* out, hipStream_t st) {
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((cols + WV - 1) / WV), dim3(RR_WAVES * WV), 0, st, partial, rows, cols,
                       out);
    return check_launch("reduce_rows_kernel");
}
int launch_reduce_loss(const double* partial, int rows, double* out, hipStream_t st) {
    hipLaunchKernelGGL(reduce_loss_kernel, dim3(1), dim3(LOSS_COLS * WV), 0, st, partial, rows, out);
    return check_launch("reduce_loss_kernel");
}
int wide_dispatch(int mode, const rl_policy_batch* g, const float* vec, void* ws, size_t ws_bytes, double* out,
                  hipStream_t st, double* loss_out, const float* cot = nullptr, float* out_mean = nullptr,
                  float* out_dmean = nullptr);                       // policy_wide_kernels.hip
struct WideShape;
// policy_split_kernels.hip: the cached Fisher-vector product of the 32-unit nets on the bf16 matrix pipe (three-way
// split operands, f32 accuracy); RL_SPLIT_NOT_TAKEN when the launch is not its to make
int split_fvp_dispatch(const rl_policy_batch* g, const float* vec, void* ws, size_t ws_bytes, double* out, hipStream_t st);
int splith_fvp_dispatch(const rl_policy_batch* g, const float* vec, void* ws, size_t ws_bytes, double* out, hipStream_t st);
int csplit_fvp_dispatch(const rl_policy_batch* g, const float* vec, void* ws, size_t ws_bytes, double* out, hipStream_t st);
size_t csplit_workspace_bytes_for(int obs_dim, int act_dim, int h0, int h1, int h2);   // 0 = not its shape
size_t wide_workspace_bytes_for(int obs_dim, int act_dim, int h0, int h1, int h2);   // 0 = not a wide shape

struct PlaneArgs {                 // MODE_OUT / MODE_OUT_TAN / MODE_BWD
    const float* cot = nullptr;
    float* out_mean = nullptr;
    float* out_dmean = nullptr;
};

template <class N, int MODE, bool CACHE = false, bool RELU = false, bool ACTS = false>
static int launch_pass(const rl_policy_batch* g, const float* vec, void* workspace, size_t workspace_bytes,
                       double* out, hipStream_t st, double* loss_out = nullptr, const CgArgs* cg = nullptr,
                       const PlaneArgs* planes = nullptr) {
    using S = Smem<N, MODE, CACHE>;
    PolicyBatch a;
    a.cot = planes ? planes->cot : nullptr;
    a.out_mean = planes ? planes->out_mean : nullptr;
    a.out_dmean = planes ? planes->out_dmean : nullptr;
    a.acts = g->activations;
    a.kl_penalty = g->kl_penalty;
    a.B = g->n_samples; a.theta = g->theta; a.vec = vec; a.obs = g->obs; a.act = g->actions; a.adv = g->advantages;
    a.old_mean = g->old_means; a.old_log_std = g->old_log_std; a.weight = g->weights;
    a.inv_count = g->inv_count; a.log_min_std = g->log_min_std;
    a.gate = (MODE == MODE_LOSS) ? g->gate : nullptr;
    a.obs_absmax = nullptr;
    if (MODE == MODE_GRAD && CACHE && g->activations && g->obs_absmax) {
        a.obs_absmax = reinterpret_cast<unsigned*>(g->obs_absmax);
        hipError_t e = hipMemsetAsync(g->obs_absmax, 0, sizeof(float), st);
        if (e != hipSuccess) return set_error(RL_ERR_HIP, "hipMemsetAsync(obs_absmax): %s", hipGetErrorString(e));
    }
    a.act0 = layer_act(g->activation, g->layer_activations, 0);
    a.act1 = layer_act(g->activation, g->layer_activations, 1);
    const int n_tiles = (a.B + TS - 1) / TS;
    const size_t lds = (size_t)S::TOTAL * sizeof(float);
    if (lds > 160 * 1024) return set_error(RL_ERR_UNSUPPORTED, "policy pass needs %zu B of LDS", lds);
    constexpr int WAVES = N::WAVES;
    int blocks_per_cu = (int)((160 * 1024) / lds);
    if (blocks_per_cu > pass_wps<N, MODE>() * 4 / WAVES) blocks_per_cu = pass_wps<N, MODE>() * 4 / WAVES;
    int grid = 256 * blocks_per_cu;
    const int need = (n_tiles + WAVES - 1) / WAVES;
    if (grid > need) grid = need;
    if (grid > MAX_GRID) grid = MAX_GRID;
    const bool with_loss = (MODE == MODE_GRAD || MODE == MODE_VPG) && loss_out != nullptr;
    const size_t row_bytes = (((size_t)grid * N::P * sizeof(float)) + 15) & ~(size_t)15;
    constexpr bool OUTMODE = (MODE == MODE_OUT || MODE == MODE_OUT_TAN);
    const size_t need_bytes = OUTMODE ? 0 : (MODE == MODE_LOSS)
        ? (size_t)grid * LOSS_COLS * sizeof(double)
        : row_bytes + (with_loss ? (size_t)grid * LOSS_COLS * sizeof(double) : 0);
    if (workspace_bytes < need_bytes)
        return set_error(RL_ERR_ARG, "policy pass workspace too small: %zu < %zu bytes", workspace_bytes,
                         need_bytes);
    a.partial = (float*)workspace;
    a.partial_loss = (MODE == MODE_LOSS) ? (double*)workspace
                                         : (with_loss ? (double*)((char*)workspace + row_bytes) : nullptr);
    auto kern = policy_pass_kernel<N, MODE, CACHE, RELU, ACTS>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return set_error(RL_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * WV), lds, st, a);
    int rc = check_launch("policy_pass_kernel");
    if (rc) return rc;
    if (OUTMODE) return 0;
    if (MODE == MODE_LOSS) {
        hipLaunchKernelGGL(reduce_loss_kernel, dim3(1), dim3(LOSS_COLS * WV), 0, st, a.partial_loss, grid, out);
    } else if (cg != nullptr) {
        hipLaunchKernelGGL(reduce_rows_cg_kernel, dim3((N::P + WV - 1) / WV), dim3(RR_WAVES * WV), 0, st, a.partial,
                           grid, N::P, *cg);
    } else {
        hipLaunchKernelGGL(reduce_rows_kernel, dim3((N::P + WV - 1) / WV), dim3(RR_WAVES * WV), 0, st, a.partial,
                           grid, N::P, out);
        if (with_loss)
            hipLaunchKernelGGL(reduce_loss_kernel, dim3(1), dim3(LOSS_COLS * WV), 0, st, a.partial_loss, grid, loss_out);
    }
    return check_launch("policy reduce kernel");
}

template <class N, bool ACTS = false>
static int dispatch_mode(int mode, const rl_policy_batch* g, const float* vec, void* ws, size_t ws_bytes,
                         double* out, hipStream_t st, double* loss_out, const CgArgs* cg = nullptr) {
    switch (mode) {
        case MODE_LOSS: return launch_pass<N, MODE_LOSS, false, false, ACTS>(g, vec, ws, ws_bytes, out, st);
        case MODE_GRAD:
            if constexpr (N::ACT_CACHE)
                if (g->activations) return launch_pass<N, MODE_GRAD, true, false, ACTS>(g, vec, ws, ws_bytes, out, st, loss_out);
            return launch_pass<N, MODE_GRAD, false, false, ACTS>(g, vec, ws, ws_bytes, out, st, loss_out);
        case MODE_FVP:
            if constexpr (N::ACT_CACHE)
                if (g->activations) return launch_pass<N, MODE_FVP, true, false, ACTS>(g, vec, ws, ws_bytes, out, st, nullptr, cg);
            return launch_pass<N, MODE_FVP, false, false, ACTS>(g, vec, ws, ws_bytes, out, st, nullptr, cg);
        case MODE_VPG: return launch_pass<N, MODE_VPG, false, false, ACTS>(g, vec, ws, ws_bytes, out, st, loss_out);
    }
    return set_error(RL_ERR_ARG, "unknown policy pass mode %d", mode);
}

// the network as a function on planes (rl_mlp_forward / rl_mlp_backward)
template <class N>
static int dispatch_planes(int mode, const rl_policy_batch* g, const float* vec, void* ws, size_t ws_bytes, double* out,
                           hipStream_t st, const PlaneArgs* pl) {
    switch (mode) {
        case MODE_OUT: return launch_pass<N, MODE_OUT>(g, vec, ws, ws_bytes, out, st, nullptr, nullptr, pl);
        case MODE_OUT_TAN: return launch_pass<N, MODE_OUT_TAN>(g, vec, ws, ws_bytes, out, st, nullptr, nullptr, pl);
        case MODE_BWD: return launch_pass<N, MODE_BWD>(g, vec, ws, ws_bytes, out, st, nullptr, nullptr, pl);
    }
    return set_error(RL_ERR_ARG, "unknown plane mode %d", mode);
}

// rectify nets (regressors): loss / log-likelihood gradient only, one output, 32 hidden units
template <class N>
static int dispatch_relu(int mode, const rl_policy_batch* g, void* ws, size_t ws_bytes, double* out, hipStream_t st,
                         double* loss_out) {
    switch (mode) {
        case MODE_LOSS: return launch_pass<N, MODE_LOSS, false, true>(g, nullptr, ws, ws_bytes, out, st);
        case MODE_VPG: return launch_pass<N, MODE_VPG, false, true>(g, nullptr, ws, ws_bytes, out, st, loss_out);
    }
    return set_error(RL_ERR_UNSUPPORTED, "rectify networks: only the loss and the log-likelihood gradient are built");
}

// (obs_dim, act_dim, H) of the one-wavefront-per-tile kernels (two equal tanh layers): every HIP-native env's pair at 32
// and 64 units + the one-output value nets.  Everything else runs on the cooperative kernels -- and the two families
// cache their activations in different layouts, which is what the split Fisher-vector products ask this for.
#define RL_NARROW_NETS(X) \
    X(4, 1, 32) X(6, 1, 32) X(11, 1, 32) X(13, 2, 32) X(20, 3, 32) X(20, 6, 32) X(21, 6, 32) \
    X(4, 1, 64) X(6, 1, 64) X(11, 1, 64) X(13, 2, 64) X(20, 3, 64) X(20, 6, 64) X(21, 6, 64) \
    X(13, 1, 32) X(20, 1, 32) X(21, 1, 32)
bool net_has_narrow_kernel(int d, int k, int h0, int h1, int h2) {
    if (h2 != 0) return false;
#define NARROWCASE(DO, DA, H) if (d == DO && k == DA && h0 == H && h1 == H) return true;
    RL_NARROW_NETS(NARROWCASE)
#undef NARROWCASE
    return false;
}

static int dispatch_net(int mode, const rl_policy_batch* g, const float* vec, void* ws, size_t ws_bytes,
                        double* out, hipStream_t st, double* loss_out = nullptr, const CgArgs* cg = nullptr,
                        const PlaneArgs* pl = nullptr) {
    const int d = g->obs_dim, k = g->act_dim, h0 = g->hidden0, h1 = g->hidden1;
    // hidden activations per layer (rl_policy_batch.activation / layer_activations): the equal-width two-layer kernels
    // take any of tanh / rectify / identity per layer at run time; every other family evaluates tanh layers only
    const int a0 = layer_act(g->activation, g->layer_activations, 0), a1 = layer_act(g->activation, g->layer_activations, 1),
              a2 = g->hidden2 > 0 ? layer_act(g->activation, g->layer_activations, 2) : RL_ACT_TANH;
    if (a0 < 0 || a0 > RL_ACT_IDENTITY || a1 < 0 || a1 > RL_ACT_IDENTITY || a2 < 0 || a2 > RL_ACT_IDENTITY)
        return set_error(RL_ERR_ARG, "unknown activation (%d, layers 0x%x)", g->activation, g->layer_activations);
    const bool all_tanh = a0 == RL_ACT_TANH && a1 == RL_ACT_TANH && a2 == RL_ACT_TANH;
    if (mode >= MODE_OUT) {
        if (a0 == RL_ACT_RECTIFY || a1 == RL_ACT_RECTIFY || a2 == RL_ACT_RECTIFY)
            return set_error(RL_ERR_UNSUPPORTED, "rl_mlp_forward / rl_mlp_backward: tanh (and identity) layers");
        // (an identity layer -- the second layer of a one-hidden-layer network -- runs on the cooperative kernels)
#define PLANECASE(DO, DA, H) \
        if (all_tanh && g->hidden2 == 0 && d == DO && k == DA && h0 == H && h1 == H) \
            return dispatch_planes<Net<DO, DA, H>>(mode, g, vec, ws, ws_bytes, out, st, pl);
        PLANECASE(4, 1, 32) PLANECASE(6, 1, 32) PLANECASE(11, 1, 32) PLANECASE(13, 2, 32) PLANECASE(20, 3, 32)
        PLANECASE(20, 6, 32) PLANECASE(21, 6, 32)
        PLANECASE(4, 1, 64) PLANECASE(6, 1, 64) PLANECASE(11, 1, 64) PLANECASE(13, 2, 64) PLANECASE(20, 3, 64)
        PLANECASE(20, 6, 64) PLANECASE(21, 6, 64)
#undef PLANECASE
        // anything else with two or three tanh layers of 32 / 64 / 128 units: the cooperative kernels, whose operand
        // images need the caller's workspace (rl_mlp_forward_ws; rl_mlp_backward always had one)
        if (ws == nullptr)
            return set_error(RL_ERR_ARG, "rl_mlp_forward: obs_dim=%d act_dim=%d hidden=(%d,%d,%d) runs on the cooperative "
                             "kernels, which need a workspace: call rl_mlp_forward_ws", d, k, h0, h1, g->hidden2);
        return wide_dispatch(mode, g, vec, ws, ws_bytes, out, st, nullptr, pl ? pl->cot : nullptr,
                             pl ? pl->out_mean : nullptr, pl ? pl->out_dmean : nullptr);
    }
    if (g->hidden2 < 0) return set_error(RL_ERR_ARG, "rl_policy_batch.hidden2 = %d", g->hidden2);
    if (g->hidden2 > 0) {              // three hidden layers: the cooperative kernels (policy_wide_kernels.hip)
        if (cg) return set_error(RL_ERR_UNSUPPORTED, "rl_policy_fvp_cg_step: two-layer 32 / 64-unit nets only");
        if (!all_tanh)
            return set_error(RL_ERR_UNSUPPORTED, "three hidden layers: tanh layers only (the cooperative kernels)");
        if (mode == MODE_FVP) {
            const int rc = csplit_fvp_dispatch(g, vec, ws, ws_bytes, out, st);    // split-operand arithmetic (cached products)
            if (rc != RL_SPLIT_NOT_TAKEN) return rc;
        }
        return wide_dispatch(mode, g, vec, ws, ws_bytes, out, st, loss_out);
    }
    if (g->activation == RL_ACT_RECTIFY && g->layer_activations == 0 && k == 1 && h0 == 32 && h1 == 32 &&
        (mode == MODE_LOSS || mode == MODE_VPG)) {
        // the regressors' nets (GaussianMLPRegressor's default hidden nonlinearity): compile-time rectify instantiations
#define RELUCASE(DO) \
        if (d == DO && k == 1 && h0 == 32 && h1 == 32) return dispatch_relu<Net<DO, 1, 32>>(mode, g, ws, ws_bytes, out, st, loss_out);
        RELUCASE(4) RELUCASE(6) RELUCASE(11) RELUCASE(13) RELUCASE(20) RELUCASE(21)
#undef RELUCASE
    }
    if (g->kl_penalty != 0.0f && mode != MODE_VPG && mode != MODE_GRAD)
        return set_error(RL_ERR_ARG, "rl_policy_batch.kl_penalty applies to the gradient passes only");
    if (mode == MODE_FVP && cg == nullptr && all_tanh) {
        int rc = splith_fvp_dispatch(g, vec, ws, ws_bytes, out, st);         // two-way f16 split (obs_absmax set)
        if (rc != RL_SPLIT_NOT_TAKEN) return rc;
        rc = split_fvp_dispatch(g, vec, ws, ws_bytes, out, st);              // (32, 32): one wavefront per tile
        if (rc != RL_SPLIT_NOT_TAKEN) return rc;
        rc = csplit_fvp_dispatch(g, vec, ws, ws_bytes, out, st);             // 64-unit and wide nets: cooperative
        if (rc != RL_SPLIT_NOT_TAKEN) return rc;
    }
#define NETCASE(DO, DA, H) \
    if (d == DO && k == DA && h0 == H && h1 == H) \
        return all_tanh ? dispatch_mode<Net<DO, DA, H>, false>(mode, g, vec, ws, ws_bytes, out, st, loss_out, cg) \
                        : dispatch_mode<Net<DO, DA, H>, true>(mode, g, vec, ws, ws_bytes, out, st, loss_out, cg);
    RL_NARROW_NETS(NETCASE)
#undef NETCASE
    // anything else with tanh layers of 32 / 64 / 128 units: the cooperative kernels
    if (cg) return set_error(RL_ERR_UNSUPPORTED, "rl_policy_fvp_cg_step: two-layer 32 / 64-unit nets only");
    if (a0 == RL_ACT_RECTIFY || a1 == RL_ACT_RECTIFY)
        return set_error(RL_ERR_UNSUPPORTED, "obs_dim=%d act_dim=%d hidden=(%d,%d): rectify layers run on the "
                         "equal-width two-layer kernels of the HIP-native (obs, action) pairs only", d, k, h0, h1);
    // (identity layers: wide_pass_kernel takes them -- the second layer of a one-hidden-layer net of 65 .. 128 units)
    return wide_dispatch(mode, g, vec, ws, ws_bytes, out, st, loss_out);
}

}  // namespace rl

using namespace rl;

static int check_batch(const rl_policy_batch* g, const char* who) {
    if (!g) return set_error(RL_ERR_ARG, "%s: null batch", who);
    if (g->n_samples <= 0 || !g->theta || !g->obs || !g->weights)
        return set_error(RL_ERR_ARG, "%s: bad batch", who);
    return 0;
}

extern "C" size_t rl_policy_workspace_bytes(int obs_dim, int act_dim, int hidden0, int hidden1, int hidden2) {
    const bool narrow = hidden2 == 0 && hidden0 == hidden1 && (hidden0 == 32 || hidden0 == 64);
    if (!narrow) {
        const size_t w = wide_workspace_bytes_for(obs_dim, act_dim, hidden0, hidden1, hidden2);
        const size_t c = csplit_workspace_bytes_for(obs_dim, act_dim, hidden0, hidden1, hidden2);
        return w > c ? w : c;
    }
    // one partial row per workgroup: P floats (or LOSS_COLS doubles)
    const size_t P = (size_t)obs_dim * hidden0 + hidden0 + (size_t)hidden0 * hidden1 + hidden1 +
                     (size_t)hidden1 * act_dim + 2 * (size_t)act_dim;
    const size_t rows = MAX_GRID;
    // rl_policy_grad_loss keeps both kinds of partial rows at once
    const size_t a = (rows * P * sizeof(float) + 15) & ~(size_t)15, b = rows * LOSS_COLS * sizeof(double);
    // a net of these widths on an (obs_dim, act_dim) pair the kernels above are not instantiated for runs on the
    // cooperative kernels (dispatch_net): the workspace covers both
    const size_t w = wide_workspace_bytes_for(obs_dim, act_dim, hidden0, hidden1, hidden2);
    const size_t c = csplit_workspace_bytes_for(obs_dim, act_dim, hidden0, hidden1, hidden2);
    const size_t m = a + b > w ? a + b : w;
    return m > c ? m : c;
}

extern "C" size_t rl_policy_activation_bytes(int n_samples, int hidden0, int hidden1, int hidden2) {
    // one float per sample and hidden unit, in 32-sample tiles -- the same size for both kernel families (two equal
    // layers of 32 / 64 units: 2 * (hidden / 32) fragments of 16 x 64 floats per tile)
    auto ok = [](int h) { return h == 32 || h == 64 || h == 128; };
    if (n_samples <= 0 || !ok(hidden0) || !ok(hidden1) || (hidden2 != 0 && !ok(hidden2))) return 0;
    const size_t n_tiles = ((size_t)n_samples + TS - 1) / TS;
    return n_tiles * (size_t)TS * (size_t)(hidden0 + hidden1 + hidden2) * sizeof(float);
}

extern "C" int rl_policy_loss_kl(const rl_policy_batch* g, void* workspace, size_t workspace_bytes,
                                 double* out4, void* stream) {
    int rc = check_batch(g, "rl_policy_loss_kl");
    if (rc) return rc;
    if (!g->actions || !g->advantages || !g->old_means || !g->old_log_std || !out4)
        return set_error(RL_ERR_ARG, "rl_policy_loss_kl: bad argument");
    return dispatch_net(MODE_LOSS, g, nullptr, workspace, workspace_bytes, out4, (hipStream_t)stream);
}

extern "C" int rl_policy_grad(const rl_policy_batch* g, int vpg, void* workspace, size_t workspace_bytes,
                              double* grad_out, void* stream) {
    int rc = check_batch(g, "rl_policy_grad");
    if (rc) return rc;
    if (!g->actions || !g->advantages || !g->old_means || !g->old_log_std || !grad_out)
        return set_error(RL_ERR_ARG, "rl_policy_grad: bad argument");
    return dispatch_net(vpg ? MODE_VPG : MODE_GRAD, g, nullptr, workspace, workspace_bytes, grad_out,
                        (hipStream_t)stream);
}

extern "C" int rl_policy_grad_loss(const rl_policy_batch* g, int vpg, void* workspace, size_t workspace_bytes,
                                   double* grad_out, double* out4, void* stream) {
    int rc = check_batch(g, "rl_policy_grad_loss");
    if (rc) return rc;
    if (!g->actions || !g->advantages || !g->old_means || !g->old_log_std || !grad_out || !out4)
        return set_error(RL_ERR_ARG, "rl_policy_grad_loss: bad argument");
    return dispatch_net(vpg ? MODE_VPG : MODE_GRAD, g, nullptr, workspace, workspace_bytes, grad_out,
                        (hipStream_t)stream, out4);
}

extern "C" int rl_policy_fvp_cg_step(const rl_policy_batch* g, void* workspace, size_t workspace_bytes, double reg_coeff,
                                     double residual_tol, double* x, double* r, double* p, float* p32, double* scal,
                                     double* fvp_scratch, unsigned int* ticket, void* stream) {
    int rc = check_batch(g, "rl_policy_fvp_cg_step");
    if (rc) return rc;
    if (!x || !r || !p || !p32 || !scal || !fvp_scratch || !ticket)
        return set_error(RL_ERR_ARG, "rl_policy_fvp_cg_step: bad argument");
    CgArgs c;
    c.reg = reg_coeff; c.tol = residual_tol; c.x = x; c.r = r; c.p = p; c.p32 = p32; c.scal = scal;
    c.fp = fvp_scratch; c.ticket = ticket;
    return dispatch_net(MODE_FVP, g, p32, workspace, workspace_bytes, fvp_scratch, (hipStream_t)stream, nullptr, &c);
}

extern "C" int rl_policy_fvp(const rl_policy_batch* g, const float* vec, void* workspace,
                             size_t workspace_bytes, double* fvp_out, void* stream) {
    int rc = check_batch(g, "rl_policy_fvp");
    if (rc) return rc;
    if (!vec || !fvp_out) return set_error(RL_ERR_ARG, "rl_policy_fvp: bad argument");
    return dispatch_net(MODE_FVP, g, vec, workspace, workspace_bytes, fvp_out, (hipStream_t)stream);
}

extern "C" int rl_mlp_forward(const rl_policy_batch* g, const float* vec, float* out, float* dout, void* stream) {
    int rc = check_batch(g, "rl_mlp_forward");
    if (rc) return rc;
    if (!out || ((vec == nullptr) != (dout == nullptr)))
        return set_error(RL_ERR_ARG, "rl_mlp_forward: out is required; vec and dout come together");
    PlaneArgs pl;
    pl.out_mean = out;
    pl.out_dmean = dout;
    return dispatch_net(vec ? MODE_OUT_TAN : MODE_OUT, g, vec, nullptr, 0, nullptr, (hipStream_t)stream, nullptr, nullptr,
                        &pl);
}

extern "C" int rl_mlp_forward_ws(const rl_policy_batch* g, const float* vec, void* workspace, size_t workspace_bytes,
                                 float* out, float* dout, void* stream) {
    int rc = check_batch(g, "rl_mlp_forward_ws");
    if (rc) return rc;
    if (!out || ((vec == nullptr) != (dout == nullptr)))
        return set_error(RL_ERR_ARG, "rl_mlp_forward_ws: out is required; vec and dout come together");
    PlaneArgs pl;
    pl.out_mean = out;
    pl.out_dmean = dout;
    return dispatch_net(vec ? MODE_OUT_TAN : MODE_OUT, g, vec, workspace, workspace_bytes, nullptr, (hipStream_t)stream,
                        nullptr, nullptr, &pl);
}

extern "C" int rl_mlp_backward(const rl_policy_batch* g, const float* cotangent, void* workspace, size_t workspace_bytes,
                               double* grad_out, void* stream) {
    int rc = check_batch(g, "rl_mlp_backward");
    if (rc) return rc;
    if (!cotangent || !grad_out) return set_error(RL_ERR_ARG, "rl_mlp_backward: bad argument");
    PlaneArgs pl;
    pl.cot = cotangent;
    return dispatch_net(MODE_BWD, g, nullptr, workspace, workspace_bytes, grad_out, (hipStream_t)stream, nullptr, nullptr,
                        &pl);
}
