// policy_wide_kernels.hip -- the TRPO / VPG update passes of policy_kernels.hip for WIDE and DEEP GaussianMLPPolicy
// mean networks: two or three tanh hidden layers of 32 / 64 / 128 units each (after zero padding), e.g. the
// (100, 50, 25) nets of rllab's MuJoCo experiments -> (128, 64, 32), or (128, 128).
// (GaussianMLPPolicy(hidden_sizes=...) is free-form: rllab/policies/gaussian_mlp_policy.py:21-58,
//  rllab/core/network.py:36-101; the functions evaluated are the f_loss / f_grad / f_Hx_plain of
//  rllab/optimizers/conjugate_gradient_optimizer.py:27-46,194-215 on rllab/algos/npo.py:72-82, exactly as in
//  policy_kernels.hip -- see its header for the four modes.)
//
// Why a second kernel family.  policy_kernels.hip gives every wavefront a whole 32-sample tile: all weight
// fragments in LDS, all activations and the persistent outer-product accumulators in its registers.  At 128 units
// neither fits (W1 + its tangent + W1 for back-propagation = 192 KB of LDS; the W1-gradient accumulator alone is 256
// registers per lane).  Here the four wavefronts of a workgroup COOPERATE on one 32-sample tile:
//   * wavefront w owns row tile w (units 32 w .. 32 w + 31) of every layer: it runs that tile's MFMA chain
//     (v_mfma_f32_32x32x2_f32, Z^T[unit][sample] = A[unit][k] B[k][sample]), applies tanh / the derivative, keeps its
//     own fragment in registers for later phases and publishes it as a [unit][33] tile in LDS -- the B operand of
//     the next layer for ALL wavefronts;
//   * A operands (weights, tangents, untransposed weights for back-propagation) are "fragment images" in global
//     memory, built once per pass by wide_stage_kernel from the flat parameters: one k-step = one coalesced 256-byte
//     load, software-prefetched eight k-steps ahead; a few hundred KB, L2-resident for the whole pass;
//   * the batch reductions gW_l = sum_s h_{l-1,s} (x) gz_{l,s} are MFMAs with the SAMPLE axis as K: both operands are
//     column reads of the same LDS tiles (stride 33: conflict-free both ways); wavefront w accumulates row tile w
//     (up to four 32x32 fragments per layer, persistent registers for the whole launch);
//   * thin products (bias gradients, the DA output columns) run on the vector ALU, one thread per unit.
// One workgroup per CU, grid-stride over tiles; every workgroup writes ONE partial row, reduce_rows_kernel sums
// the rows in float64 in a fixed order (deterministic, identical on all ranks) -- as in policy_kernels.hip.
//
// Roofline: MFMA-bound.  A Fisher-vector product of (DO -> 128 -> 128 -> DA) is ~1400 matrix instructions per
// 32-sample tile (forward 284 + tangent 540 + back-propagation 256 + outer products 272 + input layer), 350 per
// wavefront; HBM sees 4 (DO + 1) bytes per sample per pass.
#include <hip/hip_runtime.h>
#include "../../include/rllab_amd.h"
#include "capi_util.h"
#include "policy_mfma.h"
#include "policy_wide.h"

namespace rl {

int launch_reduce_rows(const float* partial, int rows, int cols, double* out, hipStream_t st);   // policy_kernels.hip
int launch_reduce_loss(const double* partial, int rows, double* out, hipStream_t st);

#ifndef WIDE_PREFETCH
#define WIDE_PREFETCH 1
#endif
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int WW = 4;                  // wavefronts per workgroup = row tiles of the widest layer
constexpr int WNT = WW * WV;
constexpr int BS = WIDE_BS;
constexpr int MAXDA_LIMIT = WIDE_MAX_DA;      // the kernels are instantiated for up to 2 or up to 8 action dims (MD)
// WMODE_OUT / WMODE_OUT_TAN / WMODE_BWD: the network as a plain function on planes (rl_mlp_forward / rl_mlp_backward, as in
// policy_kernels.hip): forward (and tangent) values of the output, back-propagation of a given output cotangent -- what
// a policy with a log-std NETWORK (gaussian_mlp_policy.py:60-98) composes with the Gaussian head kernels
enum { WMODE_LOSS = 0, WMODE_GRAD = 1, WMODE_FVP = 2, WMODE_VPG = 3, WMODE_OUT = 4, WMODE_OUT_TAN = 5, WMODE_BWD = 6 };
constexpr bool wmode_tan(int m) { return m == WMODE_FVP || m == WMODE_OUT_TAN; }          // tangent forward pass
constexpr bool wmode_out(int m) { return m == WMODE_OUT || m == WMODE_OUT_TAN; }          // planes out, nothing else
constexpr bool wmode_gradlike(int m) { return m == WMODE_GRAD || m == WMODE_FVP || m == WMODE_VPG || m == WMODE_BWD; }
constexpr int WLOSS_COLS = 4;

struct WideBatch {
    int B;
    const float* theta;
    const float* vec;
    const float* img;          // fragment images of theta: forward (+ backward for the gradient-like modes)
    const float* dimg;         // forward images of vec (FVP)
    const float* obs;
    const float* act;
    const float* adv;
    const float* old_mean;
    const float* old_log_std;
    const float* weight;
    float inv_count, log_min_std, kl_penalty;
    float* partial;            // [grid][P]
    double* partial_loss;      // [grid][4] or null
    const float* cot;          // WMODE_BWD: [DA][B] cotangent on the network output (weights / normalisation included)
    float* out_mean;           // WMODE_OUT / WMODE_OUT_TAN: [DA][B] network output
    float* out_dmean;          // WMODE_OUT_TAN: [DA][B] tangent of the output in direction vec
    float* cache;              // hidden activations of every tile (rl_policy_batch.activations): the gradient pass writes
                               // them, the Fisher-vector products of the same point read them instead of re-running the
                               // forward chains; null = none.  Per tile 32 (H0 + H1 + H2) floats: [unit, layer after
                               // layer][32 samples], the image of the LDS tiles Hb without their padding column
    const int* gate;           // WMODE_LOSS: rl_policy_batch.gate -- non-zero word: the launch returns at once
    WideShape s;
};

// ---- fragment images ------------------------------------------------------------------------------------------
// forward image of layer l:  e = (t * KS + m) * 64 + lane  ->  A[i = 32 t + lane % 32][k = 2 m + lane / 32] = W_l[k][i]
//   (layer 0: k = input slot; slot DO carries b0 -- the input tile holds a 1 there -- slots beyond are zero)
// backward image through layer l >= 1:  A[i][k] = W_l[i][k],  i = unit of layer l-1 (row tile), k = unit of layer l
__global__ void __launch_bounds__(256) wide_stage_kernel(WideShape s, const float* __restrict__ th,
                                                         float* __restrict__ img, int with_backward) {
    const int total = with_backward ? s.img_all : s.img_fwd;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        float v = 0.0f;
        if (e < s.img_fwd) {
            int l = 0;
            if (s.L > 1 && e >= s.oF[1]) l = 1;
            if (s.L > 2 && e >= s.oF[2]) l = 2;
            const int r = e - s.oF[l];
            const int lane = r & 63, m = (r >> 6) % s.KS[l], t = (r >> 6) / s.KS[l];
            const int i = 32 * t + (lane & 31), k = 2 * m + (lane >> 5);
            if (l == 0) v = k < s.DO ? th[s.oW[0] + k * s.H[0] + i] : (k == s.DO ? th[s.ob[0] + i] : 0.0f);
            else v = th[s.oW[l] + k * s.H[l] + i];
        } else {
            int l = 1;
            if (s.L > 2 && e >= s.oT[2]) l = 2;
            const int r = e - s.oT[l];
            const int lane = r & 63, m = (r >> 6) % s.KT[l], t = (r >> 6) / s.KT[l];
            const int i = 32 * t + (lane & 31), k = 2 * m + (lane >> 5);
            v = th[s.oW[l] + i * s.H[l] + k];
        }
        img[e] = v;
    }
}

// ---- LDS plan ---------------------------------------------------------------------------------------------------
struct WideLds {
    int X, Hb[WIDE_MAX_L], Db[WIDE_MAX_L], tail, dtail, part, part2, gmu, red, kred, total;
};
__host__ __device__ inline WideLds wide_lds(const WideShape& s, int mode) {
    WideLds p;
    int o = 0;
    p.X = o; o += 2 * s.KS[0] * BS;              // input slots (obs, the bias slot, zero slots up to the k-step count)
    for (int l = 0; l < WIDE_MAX_L; ++l) { p.Hb[l] = o; o += s.H[l] * BS; }
    for (int l = 0; l < WIDE_MAX_L; ++l) { p.Db[l] = o; o += (mode != WMODE_LOSS && mode != WMODE_OUT) ? s.H[l] * BS : 0; }
    o = (o + 3) & ~3;
    p.tail = o; o += s.tail;
    p.dtail = o; o += wmode_tan(mode) ? s.tail : 0;
    p.part = o;                                   // per-wavefront partial dot products of the output layer [WW][DA][32]
    p.red = o;                                    // ... whose space serves the cross-thread folds at the end of the launch
    o += (WW * s.DA * 32 > 256 + 64) ? WW * s.DA * 32 : 256 + 64;
    p.part2 = o; o += (mode == WMODE_OUT_TAN) ? WW * s.DA * 32 : 0;      // ... of the output's tangent next to the output
    p.gmu = o; o += s.DA * 32;                    // gmu[k][sample] for the thread-per-unit accumulation
    // partial accumulators of the k-split (layers with fewer than 4 row tiles): up to 3 x one fragment
    bool narrow = false;
    for (int l = 1; l < WIDE_MAX_L; ++l) narrow = narrow || (s.H[l] > 0 && s.HT[l] < 4);
    for (int l = 0; l + 1 < WIDE_MAX_L; ++l) narrow = narrow || (s.H[l + 1] > 0 && s.HT[l] < 4 && wmode_gradlike(mode));
    p.kred = o; o += narrow ? 3 * 16 * 64 : 0;
    p.total = o;
    return p;
}

// U k-steps per round, the A operands of the next round in flight while this one runs on the matrix pipe.  The chain
// does not stop at a call boundary: during its LAST round a call fetches the first four operands of the NEXT chain of
// this wavefront (`next`, may be null) into `pre`, and a call whose operands were prefetched (`pv`) starts on them --
// otherwise every chain would begin with an exposed L2 round trip (~10 chains per tile).
struct WidePre {
    float v[4];
    bool valid;
};
// RL_WIDE_PINGPONG=1: two operand buffers used in turn instead of one refilled by register moves.  Measured (MI355X,
// 2.048 M samples, on the activation cache): neutral for two layers ((128,128) FVP 3.76 vs 3.74 ms), worse for three
// ((128,64,32) 3.72 vs 3.16 ms: the duplicated chain bodies cost 45 spilled registers) -- off.
#ifndef RL_WIDE_PINGPONG
#define RL_WIDE_PINGPONG 0
#endif
#if RL_WIDE_PINGPONG
template <int U>
__device__ __forceinline__ f32x16 wide_gemm(const float* __restrict__ img, int ks, const float* bt, int lane,
                                            f32x16 acc, WidePre& pre, const float* __restrict__ next) {
    const int lj = lane & 31, lh = lane >> 5;
    const float* bp = bt + lh * BS + lj;
    // two operand buffers used in turn (no register rotation): while round n runs on one, the loads of round n + 1
    // fill the other
    float a0[U], a1[U];
    if (pre.valid) {
#pragma unroll
        for (int j = 0; j < 4; ++j) a0[j] = pre.v[j];
#pragma unroll
        for (int j = 4; j < U; ++j) a0[j] = img[j * WV + lane];
    } else {
#pragma unroll
        for (int j = 0; j < U; ++j) a0[j] = img[j * WV + lane];
    }
    pre.valid = false;
    int base = 0;
    while (true) {
        const bool more_a = base + U < ks;
        if (more_a) {
#pragma unroll
            for (int j = 0; j < U; ++j) a1[j] = img[(base + U + j) * WV + lane];
        } else if (next != nullptr) {
#pragma unroll
            for (int j = 0; j < 4; ++j) pre.v[j] = next[j * WV + lane];
            pre.valid = true;
        }
#pragma unroll
        for (int j = 0; j < U; ++j) acc = mfma(a0[j], bp[(2 * (base + j)) * BS], acc);
        if (!more_a) break;
        base += U;
        const bool more_b = base + U < ks;
        if (more_b) {
#pragma unroll
            for (int j = 0; j < U; ++j) a0[j] = img[(base + U + j) * WV + lane];
        } else if (next != nullptr) {
#pragma unroll
            for (int j = 0; j < 4; ++j) pre.v[j] = next[j * WV + lane];
            pre.valid = true;
        }
#pragma unroll
        for (int j = 0; j < U; ++j) acc = mfma(a1[j], bp[(2 * (base + j)) * BS], acc);
        if (!more_b) break;
        base += U;
    }
    return acc;
}
#else
// (A/B form: one operand buffer refilled by register moves from the prefetch buffer -- one v_mov per k-step)
template <int U>
__device__ __forceinline__ f32x16 wide_gemm(const float* __restrict__ img, int ks, const float* bt, int lane,
                                            f32x16 acc, WidePre& pre, const float* __restrict__ next) {
    const int lj = lane & 31, lh = lane >> 5;
    const float* bp = bt + lh * BS + lj;
    float a_cur[U], a_nxt[U];
    if (pre.valid) {
#pragma unroll
        for (int j = 0; j < 4; ++j) a_cur[j] = pre.v[j];
#pragma unroll
        for (int j = 4; j < U; ++j) a_cur[j] = img[j * WV + lane];
    } else {
#pragma unroll
        for (int j = 0; j < U; ++j) a_cur[j] = img[j * WV + lane];
    }
    pre.valid = false;
    for (int base = 0; base < ks; base += U) {
        const bool more = base + U < ks;
        if (more) {
#pragma unroll
            for (int j = 0; j < U; ++j) a_nxt[j] = img[(base + U + j) * WV + lane];
        } else if (next != nullptr) {
#pragma unroll
            for (int j = 0; j < 4; ++j) pre.v[j] = next[j * WV + lane];
            pre.valid = true;
        }
#pragma unroll
        for (int j = 0; j < U; ++j) acc = mfma(a_cur[j], bp[(2 * (base + j)) * BS], acc);
        if (more) {
#pragma unroll
            for (int j = 0; j < U; ++j) a_cur[j] = a_nxt[j];
        }
    }
    return acc;
}
#endif
// the chunk length picks the round size (wave-uniform)
__device__ __forceinline__ f32x16 wide_gemm_any(const float* __restrict__ img, int ks, const float* bt, int lane,
                                                f32x16 acc, WidePre& pre, const float* __restrict__ next) {
    if ((ks & 7) == 0) return wide_gemm<8>(img, ks, bt, lane, acc, pre, next);
    return wide_gemm<4>(img, ks, bt, lane, acc, pre, next);
}

// publish an output fragment (lane = sample lj + 32 half, register r = unit frag_unit(r, half) of row tile t)
__device__ __forceinline__ void wide_put(float* bt, int t, int lane, const f32x16& v) {
    const int lj = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) bt[(32 * t + frag_unit(r, 0) + 4 * lh) * BS + lj] = v[r];
}

// ... and read it back (the owner of a row tile re-reads its own publication where the tanh derivative is needed, instead
// of holding 16 registers per layer from the forward pass to the end of back-propagation; LDS operations of one
// wavefront complete in order, no barrier between its write and its read)
__device__ __forceinline__ f32x16 wide_get(const float* bt, int t, int lane) {
    const int lj = lane & 31, lh = lane >> 5;
    f32x16 v;
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = bt[(32 * t + frag_unit(r, 0) + 4 * lh) * BS + lj];
    return v;
}

// Workgroups per CU the register budget is declared for.  Two (two wavefronts per SIMD, 256 registers each): one
// workgroup's waits -- operand loads from L2, barriers -- are the other's matrix time (measured on (13 -> 128 -> 128 -> 2):
// matrix pipe busy 40 % with one workgroup per CU, waits 39 %).  The three-layer gradient-like passes carry two sets of
// outer-product accumulators (144 persistent registers) and keep one workgroup per CU with 512 registers.
#ifndef RL_WIDE_WPS_L2
#define RL_WIDE_WPS_L2 2
#endif
// KSPLIT: some layer has fewer than four row tiles and its k-steps are split over wavefronts (below); MT: the largest
// number of column tiles of a hidden-to-hidden weight matrix (2: no layer beyond the first is wider than 64), which
// sizes the persistent outer-product accumulators.
#ifndef RL_WIDE_WPS_L3M2
#define RL_WIDE_WPS_L3M2 2
#endif
template <int L, int MODE, bool KSPLIT, int MT>
constexpr int wide_wps() {   // (independent of MD: the same register budget is declared, narrow heads simply do not spill)
    return (L == 2 || MODE == WMODE_LOSS) ? RL_WIDE_WPS_L2 : (MT == 2 ? RL_WIDE_WPS_L3M2 : 1);
}

// MD: how many action dimensions the per-action register arrays of the head are sized for (2 or 8): at 8 they are the
// largest single source of register pressure (a (13 -> 128 -> 128 -> 2) gradient pass: 72 spilled registers with 8, none with 2).
template <int L, int MODE, bool KSPLIT, int MT, int MD>
__global__ void __launch_bounds__(WNT, (wide_wps<L, MODE, KSPLIT, MT>())) wide_pass_kernel(WideBatch a) {
    constexpr int MAXDA = MD;
    constexpr bool FVP = (MODE == WMODE_FVP), GRADLIKE = wmode_gradlike(MODE);
    constexpr bool TAN = wmode_tan(MODE), OUTMODE = wmode_out(MODE), BWD = (MODE == WMODE_BWD);
    const WideShape& s = a.s;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (MODE == WMODE_LOSS) {
        // a line-search candidate enqueued behind an accepted one (rl_line_search_decide): nothing to evaluate
        if (a.gate != nullptr && *a.gate != 0) return;
    }
    const WideLds p = wide_lds(s, MODE);
    // the wavefront index as a SCALAR (readfirstlane): everything derived from it -- tile ownership, operand image
    // pointers, k-ranges -- then lives in scalar registers and is computed by the scalar unit
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid / WV), lane = tid % WV, lj = lane & 31,
              lh = lane >> 5;
    const int DO = s.DO, DA = s.DA, HL = s.H[L - 1];
    float* const X = smem + p.X;
    float* const tail = smem + p.tail;
    float* const dtail = smem + p.dtail;
    float* const part = smem + p.part;
    float* const gmub = smem + p.gmu;

    // ---- once per launch: zero the tiles, stage the tail parameters ---------------------------------------------
    for (int k = tid; k < p.tail; k += WNT) smem[k] = 0.0f;
    for (int k = tid; k < s.tail; k += WNT) {
        int src;
        if (k < s.tWo) {                                   // biases of layers >= 1
            int l = 1;
            if (L > 2 && k >= s.tb[2]) l = 2;
            src = s.ob[l] + (k - s.tb[l]);
        } else if (k < s.tbo) src = s.oWo + (k - s.tWo);
        else if (k < s.tls) src = s.obo + (k - s.tbo);
        else src = (k - s.tls) < DA ? s.ols + (k - s.tls) : -1;
        tail[k] = src >= 0 ? a.theta[src] : 0.0f;
        if (TAN) dtail[k] = src >= 0 ? a.vec[src] : 0.0f;
    }
    __syncthreads();
    if (tid < 32) X[DO * BS + tid] = 1.0f;                 // the bias slot of the input tile
    // (the per-action constants of the head -- log_std, 1 / sigma, sigma^2 -- are re-derived from the staged log_std row
    //  where they are used: a handful of transcendentals per tile instead of 32 registers for the whole launch)
    // ---- persistent accumulators ---------------------------------------------------------------------------------
    f32x16 gW[L - 1][MT];         // row tile `wave` of dW_l, l = 1 .. L-1, column tiles 0 .. HT[l]-1 (<= MT)
    f32x16 gW0;                   // column tile `wave` of [dW0 ; db0] (rows = input slots)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        gW0[r] = 0.0f;
#pragma unroll
        for (int l = 0; l < L - 1; ++l)
#pragma unroll
            for (int j = 0; j < MT; ++j) gW[l][j][r] = 0.0f;
    }
    float gbv[L];                 // thread-per-unit partial sums of db_l (l >= 1; l = 0 rides in gW0's bias row)
    float gWo[MAXDA];             // thread-per-unit partial sums of dWo[unit][k]
#pragma unroll
    for (int l = 0; l < L; ++l) gbv[l] = 0.0f;
#pragma unroll
    for (int k = 0; k < MAXDA; ++k) gWo[k] = 0.0f;
    float gbo[MAXDA], gls[MAXDA];  // per-sample-lane sums (wavefront 0)
#pragma unroll
    for (int k = 0; k < MAXDA; ++k) { gbo[k] = 0.0f; gls[k] = 0.0f; }
    double acc_loss = 0.0, acc_kl = 0.0, acc_vpg = 0.0;
    float max_kl = -INFINITY, wsum = 0.0f;

    const int B = a.B, n_tiles = (B + 31) / 32;
    const bool want_loss = (MODE == WMODE_LOSS) || (a.partial_loss != nullptr);

    // what this wavefront does in the GEMM phase of each layer (tile-invariant): its row tile t, its part q of the
    // k-steps (k-split) and where its operand images start
    struct LayerJob {
        int HT, ksp, t, q, chunk, k0;
        bool busy;
        const float* im;       // forward image (theta), at this wavefront's first k-step
        const float* dim;      // the same of the tangent image (FVP)
    };
    LayerJob job[L];
#pragma unroll
    for (int l = 0; l < L; ++l) {
        LayerJob& J = job[l];
        J.HT = s.HT[l];
        const bool split = KSPLIT && l >= 1;
        J.ksp = split ? 4 / J.HT : 1;
        J.t = split ? (wave & (J.HT - 1)) : wave;
        J.q = split ? wave / J.HT : 0;
        J.busy = split ? true : (wave < J.HT);
        J.chunk = s.KS[l] / J.ksp;
        J.k0 = J.q * J.chunk;
        J.im = a.img + s.oF[l] + (J.t * s.KS[l] + J.k0) * WV;
        J.dim = TAN ? a.dimg + s.oF[l] + (J.t * s.KS[l] + J.k0) * WV : nullptr;
    }
    WidePre pre;
    pre.valid = false;
    // activation cache (wave-uniform kernel argument): GRAD writes, FVP reads
    const bool cache_rd = FVP && a.cache != nullptr, cache_wr = (MODE == WMODE_GRAD) && a.cache != nullptr;
    int coff[L];
    coff[0] = 0;
#pragma unroll
    for (int l = 1; l < L; ++l) coff[l] = coff[l - 1] + 32 * s.H[l - 1];
    const size_t ctile = (size_t)coff[L - 1] + 32 * s.H[L - 1];

    // One tile ahead (WIDE_PREFETCH): the observation slots of this thread, the sample weight and the distribution head's
    // inputs of the NEXT tile travel while this one is worked on.  A workgroup is alone on its CU (one wavefront per SIMD):
    // loaded where they are used, each of these HBM round trips was sat out in full, twice per tile.  The loads are issued
    // BEHIND the tile's last operand-image load -- the memory queue answers in order, a younger image load would wait for
    // them (policy_csplit_kernels.hip: CS_FETCH_LATE) -- and in front of work that reads LDS only.
    constexpr int XPT = (WIDE_MAX_DO * 32 + WNT - 1) / WNT;      // observation values per thread
    constexpr bool HEAD_IN = !FVP && !BWD && !OUTMODE;           // the head reads actions, advantages, the old distribution
    float nx_x[XPT], nx_wgt = 0.0f, nx_adv = 0.0f, nx_act[MAXDA], nx_om[MAXDA];
#pragma unroll
    for (int k = 0; k < MAXDA; ++k) { nx_act[k] = 0.0f; nx_om[k] = 0.0f; }
    auto prefetch = [&](int t) {
#pragma unroll
        for (int j = 0; j < XPT; ++j) {
            const int e = tid + j * WNT, d = e >> 5, bb = t * 32 + (e & 31);
            nx_x[j] = (e < DO * 32) ? __builtin_nontemporal_load(a.obs + (size_t)d * B + (bb < B ? bb : B - 1)) : 0.0f;
        }
        const int b_ = t * 32 + lj, bi_ = b_ < B ? b_ : B - 1;
        nx_wgt = b_ < B ? a.weight[bi_] : 0.0f;
        if constexpr (HEAD_IN) {
            nx_adv = a.adv[bi_];
#pragma unroll
            for (int k = 0; k < MAXDA; ++k)
                if (k < DA) { nx_act[k] = a.act[(size_t)k * B + bi_]; nx_om[k] = a.old_mean[(size_t)k * B + bi_]; }
        }
    };
    if (WIDE_PREFETCH && (int)blockIdx.x < n_tiles) prefetch(blockIdx.x);
    bool fetched = false;
    auto prefetch_next = [&](int tile) {                 // (the workgroup's last tile fetches itself again: no branch)
        if (WIDE_PREFETCH && !fetched) {
            prefetch(tile + (int)gridDim.x < n_tiles ? tile + (int)gridDim.x : tile);
            fetched = true;
        }
    };

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int b = tile * 32 + lj;
        const int bi = b < B ? b : B - 1;
        const float wgt = WIDE_PREFETCH ? nx_wgt : (b < B ? a.weight[bi] : 0.0f);
        float hd_adv = nx_adv, hd_act[MAXDA], hd_om[MAXDA];
#pragma unroll
        for (int k = 0; k < MAXDA; ++k) { hd_act[k] = nx_act[k]; hd_om[k] = nx_om[k]; }
        fetched = false;
        __syncthreads();                                   // everybody is done with the previous tile's buffers
        if (WIDE_PREFETCH) {
#pragma unroll
            for (int j = 0; j < XPT; ++j) {
                const int e = tid + j * WNT;
                if (e < DO * 32) X[(e >> 5) * BS + (e & 31)] = nx_x[j];
            }
        } else
        for (int e = tid; e < DO * 32; e += WNT) {
            const int d = e >> 5, sm = e & 31;
            const int bb = tile * 32 + sm;
            X[d * BS + sm] = a.obs[(size_t)d * B + (bb < B ? bb : B - 1)];
        }
        if (cache_rd) {
            // the forward pass of this tile as the gradient pass left it: every layer's activations straight into their
            // LDS tiles (consecutive: Hb[l] = Hb[0] + BS * (units before layer l)), in flight together with the input tile
            const f32x4* src = reinterpret_cast<const f32x4*>(a.cache + (size_t)tile * ctile);
            float* hb0 = smem + p.Hb[0];
            const int n4 = (int)(ctile / 4);
            for (int e0 = tid; e0 < n4; e0 += 4 * WNT) {     // four 16-byte loads in flight per thread and round
                f32x4 v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = src[e0 + j * WNT < n4 ? e0 + j * WNT : e0];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = e0 + j * WNT;
                    if (e < n4) {
                        float* dst = hb0 + (e >> 3) * BS + 4 * (e & 7);
                        dst[0] = v[j][0]; dst[1] = v[j][1]; dst[2] = v[j][2]; dst[3] = v[j][3];
                    }
                }
            }
        }
        __syncthreads();

        // ---- forward (+ tangent) through the hidden layers ---------------------------------------------------------
        // Layer l has HT row tiles.  HT = 4: wavefront w owns tile w.  HT < 4 (layers >= 1, KSPLIT): the k-steps of a
        // tile are SPLIT over 4 / HT wavefronts -- wavefront w works on tile t = w % HT, k-range q = w / HT -- the
        // partial accumulators of q >= 1 meet in LDS and the owner (q = 0) finishes the tile, so every SIMD's matrix
        // pipe works in the narrow layers of nets like (128, 64, 32) too.
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const float* bin = (l == 0) ? X : smem + p.Hb[l - 1];
            const LayerJob& J = job[l];
            // the chain this wavefront runs next (for the operand prefetch): the tangent chains of this layer, else the
            // first chain of the next layer, else (last layer, gradient-like modes) nothing -- the head sits in between
            const float* nxt_h = TAN ? J.dim : (l + 1 < L && job[l + 1].busy ? job[l + 1].im : nullptr);
            if (cache_rd) {
                // (nothing: the activations of every layer are in their tiles since the staging of this tile)
            } else {
                f32x16 acc;
                if (J.busy) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        acc[r] = (l >= 1 && J.q == 0) ? tail[s.tb[l] + 32 * J.t + frag_unit(r, 0) + 4 * lh] : 0.0f;
                    acc = wide_gemm_any(J.im, J.chunk, bin + 2 * J.k0 * BS, lane, acc, pre, nxt_h);
                }
                if (KSPLIT && J.ksp > 1) {                              // wave-uniform: partial accumulators -> the owner
                    float* kr = smem + p.kred;
                    if (J.q > 0) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) kr[((wave - J.HT) * 16 + r) * WV + lane] = acc[r];
                    }
                    __syncthreads();
                    if (J.q == 0) {
                        for (int qq = 1; qq < J.ksp; ++qq)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[r] += kr[((J.t + J.HT * qq - J.HT) * 16 + r) * WV + lane];
                    }
                }
                if (J.busy && J.q == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = ((s.ident >> l) & 1) ? acc[r] : ftanh(acc[r]);
                    wide_put(smem + p.Hb[l], J.t, lane, acc);
                    if (cache_wr) {
                        float* dst = a.cache + (size_t)tile * ctile + coff[l] + (32 * J.t + 4 * lh) * 32 + lj;
#pragma unroll
                        for (int r = 0; r < 16; ++r) dst[frag_unit(r, 0) * 32] = acc[r];
                    }
                }
            }
            if (TAN) {
                f32x16 dacc;
                if (J.busy) {
                    const float* nxt_d = (l + 1 < L && job[l + 1].busy) ? (cache_rd ? job[l + 1].dim : job[l + 1].im) : nullptr;
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        dacc[r] = (l >= 1 && J.q == 0) ? dtail[s.tb[l] + 32 * J.t + frag_unit(r, 0) + 4 * lh] : 0.0f;
                    if (l == 0) {
                        dacc = wide_gemm_any(J.dim, J.chunk, bin, lane, dacc, pre, nxt_d);                       // dW0^T x
                    } else {
                        dacc = wide_gemm_any(J.dim, J.chunk, bin + 2 * J.k0 * BS, lane, dacc, pre, J.im);        // dW^T h
                        dacc = wide_gemm_any(J.im, J.chunk, smem + p.Db[l - 1] + 2 * J.k0 * BS, lane, dacc, pre,
                                             nxt_d);                                                             // W^T dh
                    }
                }
                if (KSPLIT && J.ksp > 1) {
                    float* kr = smem + p.kred;
                    __syncthreads();                                // the owners are done reading the h partials
                    if (J.q > 0) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) kr[((wave - J.HT) * 16 + r) * WV + lane] = dacc[r];
                    }
                    __syncthreads();
                    if (J.q == 0) {
                        for (int qq = 1; qq < J.ksp; ++qq)
#pragma unroll
                            for (int r = 0; r < 16; ++r) dacc[r] += kr[((J.t + J.HT * qq - J.HT) * 16 + r) * WV + lane];
                    }
                }
                if (J.busy && J.q == 0) {
                    const f32x16 hl = wide_get(smem + p.Hb[l], J.t, lane);
#pragma unroll
                    for (int r = 0; r < 16; ++r) dacc[r] *= ((s.ident >> l) & 1) ? 1.0f : (1.0f - hl[r] * hl[r]);
                    wide_put(smem + p.Db[l], J.t, lane, dacc);
                }
            }
            __syncthreads();
        }

        if (!GRADLIKE) prefetch_next(tile);                 // forward-only passes: no operand image is loaded behind this point
        // ---- output layer: partial dot products over this wavefront's quarter of the last hidden layer -----------
        // lane (sample lj, half lh) covers units [q0, q0 + HL / 8) of the quarter, halves are folded by half_sum
        // (FVP: the tangent of the output only; OUT_TAN: output and tangent; BWD: nothing -- the cotangent is given)
        if (!BWD) {
            const int per = HL / 8, q0 = wave * (HL / 4) + lh * per;
            const float* hb = smem + p.Hb[L - 1];
            const float* db = smem + p.Db[L - 1];
            float pm[MAXDA], pd[MAXDA];
#pragma unroll
            for (int k = 0; k < MAXDA; ++k) { pm[k] = 0.0f; pd[k] = 0.0f; }
            for (int u = q0; u < q0 + per; ++u) {
                const float hv = hb[u * BS + lj];
                const float dv = TAN ? db[u * BS + lj] : 0.0f;
#pragma unroll
                for (int k = 0; k < MAXDA; ++k)
                    if (k < DA) {
                        if (TAN) {
                            pd[k] = __builtin_fmaf(hv, dtail[s.tWo + u * DA + k], pd[k]);
                            pd[k] = __builtin_fmaf(dv, tail[s.tWo + u * DA + k], pd[k]);
                        }
                        if (!FVP) pm[k] = __builtin_fmaf(hv, tail[s.tWo + u * DA + k], pm[k]);
                    }
            }
#pragma unroll
            for (int k = 0; k < MAXDA; ++k)
                if (k < DA) {
                    const float v = half_sum(FVP ? pd[k] : pm[k]);
                    if (lh == 0) part[(wave * DA + k) * 32 + lj] = v;
                    if (MODE == WMODE_OUT_TAN) {
                        const float v2 = half_sum(pd[k]);
                        if (lh == 0) smem[p.part2 + (wave * DA + k) * 32 + lj] = v2;
                    }
                }
        }
        __syncthreads();
        if (OUTMODE) {
            // the network as a function on planes: every sample's output (and tangent), nothing else
            if (wave == 0 && lh == 0 && b < B) {
#pragma unroll
                for (int k = 0; k < MAXDA; ++k)
                    if (k < DA) {
                        a.out_mean[(size_t)k * B + b] =
                            tail[s.tbo + k] + ((part[(0 * DA + k) * 32 + lj] + part[(1 * DA + k) * 32 + lj]) +
                                               (part[(2 * DA + k) * 32 + lj] + part[(3 * DA + k) * 32 + lj]));
                        if (MODE == WMODE_OUT_TAN) {
                            const float* p2 = smem + p.part2;
                            a.out_dmean[(size_t)k * B + b] =
                                dtail[s.tbo + k] + ((p2[(0 * DA + k) * 32 + lj] + p2[(1 * DA + k) * 32 + lj]) +
                                                    (p2[(2 * DA + k) * 32 + lj] + p2[(3 * DA + k) * 32 + lj]));
                        }
                    }
            }
            continue;                                       // (barrier at the loop head)
        }

        // ---- per-sample cotangent on the mean (every wavefront, redundantly: lane = sample) ------------------------
        float gmu[MAXDA];
#pragma unroll
        for (int k = 0; k < MAXDA; ++k) gmu[k] = 0.0f;
        const bool keeper = (wave == 0 && lh == 0);        // the one lane that counts this sample in the scalar sums
        float lstd[MAXDA], inv_std[MAXDA], var_[MAXDA];
        bool floored[MAXDA];
#pragma unroll
        for (int k = 0; k < MAXDA; ++k) {
            const float raw = k < DA ? tail[s.tls + k] : 0.0f;
            floored[k] = raw < a.log_min_std;
            lstd[k] = fmaxf(raw, a.log_min_std);
            inv_std[k] = __expf(-lstd[k]);
            var_[k] = __expf(2.0f * lstd[k]);
        }
        if (BWD) {
#pragma unroll
            for (int k = 0; k < MAXDA; ++k)
                if (k < DA) gmu[k] = b < B ? a.cot[(size_t)k * B + bi] : 0.0f;
        } else if (!FVP) {
            float mean[MAXDA];
#pragma unroll
            for (int k = 0; k < MAXDA; ++k)
                mean[k] = k < DA ? tail[s.tbo + k] + ((part[(0 * DA + k) * 32 + lj] + part[(1 * DA + k) * 32 + lj]) +
                                                      (part[(2 * DA + k) * 32 + lj] + part[(3 * DA + k) * 32 + lj]))
                                 : 0.0f;
            const float advb = WIDE_PREFETCH ? hd_adv : a.adv[bi];
            float zz_new = 0.0f, zz_old = 0.0f, sls_new = 0.0f, sls_old = 0.0f, kl = 0.0f;
            float znew[MAXDA], dmv[MAXDA], numv[MAXDA];
#pragma unroll
            for (int k = 0; k < MAXDA; ++k) {
                znew[k] = 0.0f; dmv[k] = 0.0f; numv[k] = 0.0f;
                if (k < DA) {
                    const float ak = WIDE_PREFETCH ? hd_act[k] : a.act[(size_t)k * B + bi];
                    const float mo = WIDE_PREFETCH ? hd_om[k] : a.old_mean[(size_t)k * B + bi];
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
            }
            const float logp_new = -sls_new - 0.5f * zz_new;
            const float dlog = logp_new - (-sls_old - 0.5f * zz_old);
            const float lr = __expf(dlog);
            const float w1 = keeper ? wgt : 0.0f;
            if (want_loss) {
                acc_loss += (double)(w1 * lr * advb);
                acc_kl += (double)(w1 * kl);
                acc_vpg += (double)(w1 * (logp_new - 0.5f * (float)DA * 1.8378770664093453f) * advb);
                if (w1 > 0.0f) max_kl = fmaxf(max_kl, kl);
            }
            if (MODE != WMODE_LOSS) {
                const float c = -wgt * advb * (MODE == WMODE_GRAD ? lr : 1.0f) * a.inv_count;
                const float c1 = keeper ? c : 0.0f;
#pragma unroll
                for (int k = 0; k < MAXDA; ++k)
                    if (k < DA) {
                        gmu[k] = c * znew[k] * inv_std[k];
                        if (!floored[k]) gls[k] += c1 * (znew[k] * znew[k] - 1.0f);
                    }
                if (a.kl_penalty != 0.0f) {
                    const float pp = a.kl_penalty * wgt * a.inv_count;
                    const float p1 = keeper ? pp : 0.0f;
#pragma unroll
                    for (int k = 0; k < MAXDA; ++k)
                        if (k < DA) {
                            const float den = 2.0f * var_[k] + 1e-8f;
                            const float dkl_mu = -2.0f * dmv[k] / den;
                            const float dkl_ls = 1.0f - (2.0f * var_[k] * den + 4.0f * var_[k] * numv[k]) / (den * den);
                            gmu[k] = __builtin_fmaf(pp, dkl_mu, gmu[k]);
                            if (!floored[k]) gls[k] += p1 * dkl_ls;
                        }
                }
            }
        } else {
            const float c = wgt * a.inv_count;
#pragma unroll
            for (int k = 0; k < MAXDA; ++k)
                if (k < DA) {
                    const float dmu = dtail[s.tbo + k] + ((part[(0 * DA + k) * 32 + lj] + part[(1 * DA + k) * 32 + lj]) +
                                                          (part[(2 * DA + k) * 32 + lj] + part[(3 * DA + k) * 32 + lj]));
                    gmu[k] = c * dmu * (2.0f / (2.0f * var_[k] + 1e-8f));
                }
            if (keeper) wsum += c;
        }
        if (!GRADLIKE) continue;                            // MODE_LOSS: forward only (barrier at the loop head)

        if (keeper) {
#pragma unroll
            for (int k = 0; k < MAXDA; ++k) gbo[k] += gmu[k];
        }
        if (wave == 0 && lh == 0) {
#pragma unroll
            for (int k = 0; k < MAXDA; ++k)
                if (k < DA) gmub[k * 32 + lj] = gmu[k];
        }

        // ---- back-propagation: gz_{L-1} = (Wo gmu) (1 - h^2), then gz_{l-1} = (W_l gz_l) (1 - h_{l-1}^2) -----------
        f32x16 gz;
        if (wave < s.HT[L - 1]) {
            const f32x16 hlast = wide_get(smem + p.Hb[L - 1], wave, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int u = 32 * wave + frag_unit(r, 0) + 4 * lh;
                float g = 0.0f;
#pragma unroll
                for (int k = 0; k < MAXDA; ++k)
                    if (k < DA) g = __builtin_fmaf(tail[s.tWo + u * DA + k], gmu[k], g);
                gz[r] = g * (((s.ident >> (L - 1)) & 1) ? 1.0f : (1.0f - hlast[r] * hlast[r]));
            }
            wide_put(smem + p.Db[L - 1], wave, lane, gz);      // the tangent of this layer is consumed: reuse its tile
        }
        __syncthreads();
        // thread-per-unit sums over the samples of the tile: dWo[u][k] += h_{L-1}[u][s] gmu[k][s], db_{L-1}[u] += gz[u][s]
        {
            const int u = tid % HL, prt = tid / HL, nprt = WNT / HL;
            const float* hb = smem + p.Hb[L - 1] + u * BS;
            const float* gb = smem + p.Db[L - 1] + u * BS;
            float sb = 0.0f;
            for (int sm = prt; sm < 32; sm += nprt) {
                const float hv = hb[sm];
                sb += gb[sm];
#pragma unroll
                for (int k = 0; k < MAXDA; ++k)
                    if (k < DA) gWo[k] = __builtin_fmaf(hv, gmub[k * 32 + sm], gWo[k]);
            }
            gbv[L - 1] += sb;
        }
#pragma unroll
        for (int l = L - 1; l >= 1; --l) {
            // gz_{l-1} = (W_l gz_l) (1 - h_{l-1}^2): HT[l-1] output tiles, the k-steps split over 4 / HT[l-1] wavefronts
            const int HTo = s.HT[l - 1], ksp = KSPLIT ? 4 / HTo : 1;
            const int t = KSPLIT ? (wave & (HTo - 1)) : wave, q = KSPLIT ? wave / HTo : 0;
            const bool bbusy = KSPLIT ? true : (wave < HTo);
            const int chunk = s.KT[l] / ksp, k0 = q * chunk;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            if (bbusy) {
                const float* im = a.img + s.oT[l] + (t * s.KT[l] + k0) * WV;
                const float* bk = smem + p.Db[l] + 2 * k0 * BS;
                acc = wide_gemm_any(im, chunk, bk, lane, acc, pre, nullptr);
            }
            if (KSPLIT && ksp > 1) {
                float* kr = smem + p.kred;
                if (q > 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) kr[((wave - HTo) * 16 + r) * WV + lane] = acc[r];
                }
                __syncthreads();
                if (q == 0) {
                    for (int qq = 1; qq < ksp; ++qq)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[r] += kr[((t + HTo * qq - HTo) * 16 + r) * WV + lane];
                }
            }
            if (bbusy && q == 0) {
                const f32x16 hb = wide_get(smem + p.Hb[l - 1], t, lane);
#pragma unroll
                for (int r = 0; r < 16; ++r) gz[r] = acc[r] * (((s.ident >> (l - 1)) & 1) ? 1.0f : (1.0f - hb[r] * hb[r]));
                wide_put(smem + p.Db[l - 1], t, lane, gz);
            }
            __syncthreads();
            if (l - 1 >= 1) {
                const int H = s.H[l - 1], u = tid % H, prt = tid / H, nprt = WNT / H;
                const float* gb = smem + p.Db[l - 1] + u * BS;
                float sb = 0.0f;
                for (int sm = prt; sm < 32; sm += nprt) sb += gb[sm];
                gbv[l - 1] += sb;
            }
        }

        prefetch_next(tile);
        // ---- outer products over the sample axis (K = 32 samples = 16 k-steps) ------------------------------------
        // operand m of lane (c, half): value of unit c (of the row / column tile) at sample 2 m + half
#pragma unroll
        for (int l = 1; l < L; ++l) {
            if (wave < s.HT[l - 1]) {
                const float* ap = smem + p.Hb[l - 1] + (32 * wave + lj) * BS + lh;
                float aop[16];
#pragma unroll
                for (int m = 0; m < 16; ++m) aop[m] = ap[2 * m];
#pragma unroll
                for (int tj = 0; tj < MT; ++tj)
                    if (tj < s.HT[l]) {
                        const float* bp = smem + p.Db[l] + (32 * tj + lj) * BS + lh;
#pragma unroll
                        for (int m = 0; m < 16; ++m) gW[l - 1][tj] = mfma(aop[m], bp[2 * m], gW[l - 1][tj]);
                    }
            }
        }
        if (wave < s.HT[0]) {
            // rows = input slots (slot DO = 1: the bias row); the tile holds 2 KS0 of them, lanes beyond read row 0
            // (finite; their output rows are never stored)
            const float* ap = X + (lj < 2 * s.KS[0] ? lj : 0) * BS + lh;
            const float* bp = smem + p.Db[0] + (32 * wave + lj) * BS + lh;
#pragma unroll
            for (int m = 0; m < 16; ++m) gW0 = mfma(ap[2 * m], bp[2 * m], gW0);
        }
    }

    // ---- one partial row per workgroup ------------------------------------------------------------------------------
    auto fold_loss = [&]() {
        __syncthreads();
        double* red = reinterpret_cast<double*>(smem);
        const double l = wave_sum(acc_loss), k = wave_sum(acc_kl), v = wave_sum(acc_vpg);
        const float mk = wave_max(max_kl);
        if (tid == 0) {                                    // only wavefront 0 holds counted samples
            double* o = a.partial_loss + (size_t)blockIdx.x * WLOSS_COLS;
            o[0] = l; o[1] = k; o[2] = v; o[3] = (double)mk;
        }
        (void)red;
    };
    if (MODE == WMODE_LOSS) {
        fold_loss();
        return;
    }
    float* row = a.partial + (size_t)blockIdx.x * s.P;
#pragma unroll
    for (int l = 1; l < L; ++l)
        if (wave < s.HT[l - 1]) {
#pragma unroll
            for (int tj = 0; tj < MT; ++tj)
                if (tj < s.HT[l]) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        row[s.oW[l] + (32 * wave + frag_unit(r, 0) + 4 * lh) * s.H[l] + 32 * tj + lj] = gW[l - 1][tj][r];
                }
        }
    if (wave < s.HT[0]) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = frag_unit(r, 0) + 4 * lh;
            if (d < DO) row[s.oW[0] + d * s.H[0] + 32 * wave + lj] = gW0[r];
            else if (d == DO) row[s.ob[0] + 32 * wave + lj] = gW0[r];
        }
    }
    // thread-per-unit partials: the `nprt` threads of a unit meet in LDS in a fixed order
    __syncthreads();
    float* red = smem + p.red;
#pragma unroll
    for (int l = 1; l < L; ++l) {
        const int H = s.H[l], u = tid % H, prt = tid / H, nprt = WNT / H;
        red[tid] = gbv[l];
        __syncthreads();
        if (prt == 0) {
            float t = red[u];
            for (int q = 1; q < nprt; ++q) t += red[q * H + u];
            row[s.ob[l] + u] = t;
        }
        __syncthreads();
    }
    {
        const int u = tid % HL, prt = tid / HL, nprt = WNT / HL;
#pragma unroll
        for (int k = 0; k < MAXDA; ++k) {
            if (k < DA) {                                   // wave-uniform
                red[tid] = gWo[k];
                __syncthreads();
                if (prt == 0) {
                    float t = red[u];
                    for (int q = 1; q < nprt; ++q) t += red[q * HL + u];
                    row[s.oWo + u * DA + k] = t;
                }
                __syncthreads();
            }
        }
    }
    if (wave == 0) {
        const float ws = wave_sum(wsum);
#pragma unroll
        for (int k = 0; k < MAXDA; ++k)
            if (k < DA) {
                const float b2 = wave_sum(gbo[k]), ls = wave_sum(gls[k]);
                if (lane == 0) {
                    row[s.obo + k] = b2;
                    if (FVP) {
                        // log_std block of the Fisher: d2KL/ds2 = 4 v (2 v - eps) / (2 v + eps)^2, v = sigma^2
                        const float raw = tail[s.tls + k];
                        const float vv = __expf(2.0f * fmaxf(raw, a.log_min_std)), e = 1e-8f;
                        const float c = raw < a.log_min_std ? 0.0f
                                                            : 4.0f * vv * (2.0f * vv - e) / ((2.0f * vv + e) * (2.0f * vv + e));
                        row[s.ols + k] = c * a.vec[s.ols + k] * ws;
                    } else {
                        row[s.ols + k] = ls;
                    }
                }
            }
    }
    if ((MODE == WMODE_GRAD || MODE == WMODE_VPG) && a.partial_loss != nullptr) fold_loss();
}

constexpr int WIDE_GRID = 512;         // up to two workgroups per CU (wide_wps)

size_t wide_workspace_bytes(const WideShape& s) {
    const size_t rows = ((size_t)WIDE_GRID * s.P * sizeof(float) + 15) & ~(size_t)15;
    const size_t loss = (size_t)WIDE_GRID * WLOSS_COLS * sizeof(double);
    const size_t imgs = ((size_t)s.img_all + (size_t)s.img_fwd) * sizeof(float);
    return rows + loss + imgs + 64;
}

size_t wide_workspace_bytes_for(int obs_dim, int act_dim, int h0, int h1, int h2) {
    WideShape s;
    if (!wide_shape(obs_dim, act_dim, h0, h1, h2, s)) return 0;
    return wide_workspace_bytes(s);
}

struct WidePlanes { const float* cot; float* out_mean; float* out_dmean; };

template <int L, int MODE, bool KSPLIT, int MT, int MD>
static int launch_wide(const WideShape& s, const rl_policy_batch* g, const float* vec, void* workspace,
                       size_t workspace_bytes, double* out, hipStream_t st, double* loss_out,
                       const WidePlanes* planes = nullptr) {
    if (workspace_bytes < wide_workspace_bytes(s))
        return set_error(RL_ERR_ARG, "policy pass workspace too small: %zu < %zu bytes", workspace_bytes,
                         wide_workspace_bytes(s));
    WideBatch a;
    a.s = s;
    a.B = g->n_samples; a.theta = g->theta; a.vec = vec; a.obs = g->obs; a.act = g->actions; a.adv = g->advantages;
    a.old_mean = g->old_means; a.old_log_std = g->old_log_std; a.weight = g->weights;
    a.inv_count = g->inv_count; a.log_min_std = g->log_min_std; a.kl_penalty = g->kl_penalty;
    a.cot = planes ? planes->cot : nullptr;
    a.out_mean = planes ? planes->out_mean : nullptr;
    a.out_dmean = planes ? planes->out_dmean : nullptr;
    a.cache = (MODE == WMODE_GRAD || MODE == WMODE_FVP) ? g->activations : nullptr;
    a.gate = (MODE == WMODE_LOSS) ? g->gate : nullptr;
    const int n_tiles = (a.B + 31) / 32;
    const WideLds p = wide_lds(s, MODE);
    const size_t lds = (size_t)p.total * sizeof(float);
    if (lds > 160 * 1024) return set_error(RL_ERR_UNSUPPORTED, "wide policy pass needs %zu B of LDS", lds);
    int per_cu = wide_wps<L, MODE, KSPLIT, MT>();
    if ((size_t)per_cu * lds > 160 * 1024) per_cu = 1;
    const int max_grid = 256 * per_cu;
    const int grid = n_tiles < max_grid ? n_tiles : max_grid;
    const size_t rows = ((size_t)WIDE_GRID * s.P * sizeof(float) + 15) & ~(size_t)15;
    const bool with_loss = (MODE == WMODE_GRAD || MODE == WMODE_VPG) && loss_out != nullptr;
    a.partial = (float*)workspace;
    double* lossp = (double*)((char*)workspace + rows);
    a.partial_loss = (MODE == WMODE_LOSS || with_loss) ? lossp : nullptr;
    float* img = (float*)((char*)workspace + rows + (size_t)WIDE_GRID * WLOSS_COLS * sizeof(double));
    float* dimg = img + s.img_all;
    a.img = img; a.dimg = dimg;
    const int with_bwd = wmode_gradlike(MODE) ? 1 : 0;
    const int n_img = with_bwd ? s.img_all : s.img_fwd;
    hipLaunchKernelGGL(wide_stage_kernel, dim3((n_img + 255) / 256 < 512 ? (n_img + 255) / 256 : 512), dim3(256), 0, st,
                       s, g->theta, img, with_bwd);
    if (wmode_tan(MODE))
        hipLaunchKernelGGL(wide_stage_kernel, dim3((s.img_fwd + 255) / 256 < 512 ? (s.img_fwd + 255) / 256 : 512),
                           dim3(256), 0, st, s, vec, dimg, 0);
    int rc = check_launch("wide_stage_kernel");
    if (rc) return rc;
    auto kern = wide_pass_kernel<L, MODE, KSPLIT, MT, MD>;
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
        if (e != hipSuccess) return set_error(RL_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_lds = 160 * 1024;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WNT), lds, st, a);
    rc = check_launch("wide_pass_kernel");
    if (rc) return rc;
    if (wmode_out(MODE)) return 0;                      // the planes are the result
    if (MODE == WMODE_LOSS) return launch_reduce_loss(a.partial_loss, grid, out, st);
    rc = launch_reduce_rows(a.partial, grid, s.P, out, st);
    if (rc) return rc;
    if (with_loss) return launch_reduce_loss(a.partial_loss, grid, loss_out, st);
    return 0;
}

template <int L, bool KSPLIT, int MT, int MD>
static int wide_mode_md(const WideShape& s, int mode, const rl_policy_batch* g, const float* vec, void* ws,
                        size_t ws_bytes, double* out, hipStream_t st, double* loss_out, const WidePlanes* pl) {
    switch (mode) {
        case WMODE_OUT: return launch_wide<L, WMODE_OUT, KSPLIT, MT, MD>(s, g, vec, ws, ws_bytes, out, st, nullptr, pl);
        case WMODE_OUT_TAN: return launch_wide<L, WMODE_OUT_TAN, KSPLIT, MT, MD>(s, g, vec, ws, ws_bytes, out, st, nullptr, pl);
        case WMODE_BWD: return launch_wide<L, WMODE_BWD, KSPLIT, MT, MD>(s, g, vec, ws, ws_bytes, out, st, nullptr, pl);
        case WMODE_LOSS: return launch_wide<L, WMODE_LOSS, KSPLIT, MT, MD>(s, g, vec, ws, ws_bytes, out, st, nullptr);
        case WMODE_GRAD: return launch_wide<L, WMODE_GRAD, KSPLIT, MT, MD>(s, g, vec, ws, ws_bytes, out, st, loss_out);
        case WMODE_FVP: return launch_wide<L, WMODE_FVP, KSPLIT, MT, MD>(s, g, vec, ws, ws_bytes, out, st, nullptr);
        case WMODE_VPG: return launch_wide<L, WMODE_VPG, KSPLIT, MT, MD>(s, g, vec, ws, ws_bytes, out, st, loss_out);
    }
    return set_error(RL_ERR_ARG, "unknown policy pass mode %d", mode);
}

template <int L, bool KSPLIT, int MT>
static int wide_mode(const WideShape& s, int mode, const rl_policy_batch* g, const float* vec, void* ws,
                     size_t ws_bytes, double* out, hipStream_t st, double* loss_out, const WidePlanes* pl) {
    return s.DA <= 2 ? wide_mode_md<L, KSPLIT, MT, 2>(s, mode, g, vec, ws, ws_bytes, out, st, loss_out, pl)
                     : wide_mode_md<L, KSPLIT, MT, 8>(s, mode, g, vec, ws, ws_bytes, out, st, loss_out, pl);
}

template <int L>
static int wide_shape_class(const WideShape& s, int mode, const rl_policy_batch* g, const float* vec, void* ws,
                            size_t ws_bytes, double* out, hipStream_t st, double* loss_out, const WidePlanes* pl) {
    bool ksplit = false;
    int mt = 1;
    for (int l = 0; l < s.L; ++l) {
        if (s.HT[l] < 4 && (l >= 1 || l + 1 < s.L)) ksplit = true;      // forward of layer l >= 1, backward INTO layer l
        if (l >= 1 && s.HT[l] > mt) mt = s.HT[l];
    }
    if (mt <= 2) return ksplit ? wide_mode<L, true, 2>(s, mode, g, vec, ws, ws_bytes, out, st, loss_out, pl)
                               : wide_mode<L, false, 2>(s, mode, g, vec, ws, ws_bytes, out, st, loss_out, pl);
    return ksplit ? wide_mode<L, true, 4>(s, mode, g, vec, ws, ws_bytes, out, st, loss_out, pl)
                  : wide_mode<L, false, 4>(s, mode, g, vec, ws, ws_bytes, out, st, loss_out, pl);
}

// entry point for policy_kernels.hip's dispatcher: nets that the one-wavefront-per-tile kernels are not built for
int wide_dispatch(int mode, const rl_policy_batch* g, const float* vec, void* ws, size_t ws_bytes, double* out,
                  hipStream_t st, double* loss_out, const float* cot, float* out_mean, float* out_dmean) {
    const WidePlanes planes = {cot, out_mean, out_dmean};
    const WidePlanes* pl = &planes;
    WideShape s;
    if (g->activation != RL_ACT_TANH || !wide_shape(g->obs_dim, g->act_dim, g->hidden0, g->hidden1, g->hidden2, s) ||
        !wide_activations(g->layer_activations, s))
        return set_error(RL_ERR_UNSUPPORTED,
                         "no fused policy kernel for obs_dim=%d act_dim=%d hidden=(%d,%d,%d), layers 0x%x: two or three tanh "
                         "(or identity) layers of 32 / 64 / 128 units, obs_dim <= %d, act_dim <= %d",
                         g->obs_dim, g->act_dim, g->hidden0, g->hidden1, g->hidden2, g->layer_activations, WIDE_MAX_DO,
                         WIDE_MAX_DA);
    return s.L == 2 ? wide_shape_class<2>(s, mode, g, vec, ws, ws_bytes, out, st, loss_out, pl)
                    : wide_shape_class<3>(s, mode, g, vec, ws, ws_bytes, out, st, loss_out, pl);
}

}  // namespace rl
