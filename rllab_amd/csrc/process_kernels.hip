// process_kernels.hip -- BaseSampler.process_samples (rllab/sampler/base.py:48-104) and
// LinearFeatureBaseline (rllab/baselines/linear_feature_baseline.py:6-43) on dense [T][n]
// planes: path indexing, baseline prediction, batch statistics and the ridge-regression
// normal equations, each as one pass over the batch.
//
//   rl_path_scan     : per column, forward max-scan of path starts and backward or-scan of done
//                      flags -> step index inside its path (the `arange(l)` of
//                      linear_feature_baseline.py:16-19), whole-path validity
//                      (batch_polopt.py:30-34 / vectorized_sampler.py:72-97) and, fused into
//                      the same pass, the baseline prediction  phi(o, t) . w  (:38-43).
//   rl_sample_stats  : every moment process_samples needs (explained variance :68-71,
//                      advantage centring algos/util.py:7-12, per-path return statistics
//                      :93-103) in ONE read of the batch, float64, deterministic.
//   rl_adv_finish    : (a - mean) / (std + 1e-8) [+ shift to positive], zero on invalid samples.
//   rl_lfb_normal_eq : Phi^T W Phi and Phi^T W y in float64 without materialising Phi
//                      (2*Do+4 features x B samples x 8 B = 0.5 GB at the headline config):
//                      features are rebuilt per 64-sample tile in LDS, every lane owns a
//                      register block of the Gram matrix, f64 FMAs run at the vector rate.
// All four are HBM-/issue-bound single passes; none survives in a profile next to the rollout.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "../../include/rllab_amd.h"
#include "capi_util.h"

namespace rl {

constexpr int PS_EW = 32;   // envs per workgroup
constexpr int PS_KB = 32;   // time chunks per workgroup
constexpr int MAX_DO = 21;  // observation sizes the feature kernels are built for (2*Do+5 <= 64)

// features of one sample, reference order: clip(o), clip(o)^2, al, al^2, al^3, 1 with al = t/100
template <class F>
__device__ __forceinline__ void lfb_features(int Do, const float* __restrict__ obs, size_t plane, size_t off,
                                             int tin, F&& emit) {
    for (int d = 0; d < Do; ++d) {
        double o = (double)obs[(size_t)d * plane + off];
        o = fmin(fmax(o, -10.0), 10.0);
        emit(d, o);
        emit(Do + d, o * o);
    }
    const double al = (double)tin / 100.0;
    emit(2 * Do, al);
    emit(2 * Do + 1, al * al);
    emit(2 * Do + 2, al * al * al);
    emit(2 * Do + 3, 1.0);
}

__global__ void __launch_bounds__(PS_EW* PS_KB)
path_scan_kernel(int T, int n, int Do, const uint8_t* __restrict__ done, const float* __restrict__ obs,
                 const double* __restrict__ coeffs, int whole_paths, int32_t* __restrict__ tin,
                 uint8_t* __restrict__ valid, double* __restrict__ values) {
    __shared__ int s_start[PS_KB][PS_EW];     // last path start inside the chunk, or -1
    __shared__ int s_done[PS_KB][PS_EW];      // any done flag inside the chunk
    __shared__ double s_w[2 * MAX_DO + 4];
    const int e = threadIdx.x, k = threadIdx.y;
    const int i = blockIdx.x * PS_EW + e;
    const int L = (T + PS_KB - 1) / PS_KB;
    const int t0 = k * L, t1 = min(T, t0 + L);
    const bool live = (i < n) && (t0 < T);
    const int F = 2 * Do + 4;
    if (coeffs)
        for (int f = threadIdx.y * PS_EW + threadIdx.x; f < F; f += PS_EW * PS_KB) s_w[f] = coeffs[f];
    int last = -1, any = 0;
    if (live) {
        for (int t = t0; t < t1; ++t) {
            const bool start = (t == 0) || done[(size_t)(t - 1) * n + i];
            if (start) last = t;
            any |= done[(size_t)t * n + i];
        }
    }
    s_start[k][e] = last;
    s_done[k][e] = any;
    __syncthreads();
    if (!live) return;
    int cur = -1;                      // last start before this chunk
    for (int j = 0; j < k; ++j) cur = max(cur, s_start[j][e]);
    int later = 0;                     // any done after this chunk
    for (int j = k + 1; j < PS_KB; ++j) later |= s_done[j][e];
    // suffix-or inside the chunk needs a backward walk; do it first into a bit mask (L <= 32 * 8)
    // -- chunks are short (T / 32 steps), so a second forward walk with a running count suffices:
    int remaining = 0;                 // number of done flags at or after t inside the chunk
    for (int t = t0; t < t1; ++t) remaining += done[(size_t)t * n + i] ? 1 : 0;
    const size_t plane = (size_t)T * n;
    for (int t = t0; t < t1; ++t) {
        const size_t off = (size_t)t * n + i;
        if ((t == 0) || done[(size_t)(t - 1) * n + i]) cur = t;
        const int ti = t - cur;
        tin[off] = ti;
        valid[off] = (!whole_paths || remaining > 0 || later) ? 1 : 0;
        if (done[off]) remaining -= 1;
        if (values) {
            double acc = 0.0;
            if (coeffs) lfb_features(Do, obs, plane, off, ti, [&](int f, double v) { acc = fma(s_w[f], v, acc); });
            values[off] = acc;
        }
    }
}

// The same scan with the chunk's done flags in registers (read once instead of four times) and the baseline
// prediction restructured so that a thread has its chunk's L <= LMAX observation loads of one component in flight at a
// time (d outer, steps inner: the loop above issues one dependent load per feature); EW = 16 envs per workgroup fill
// the chip at 4096 envs, neighbouring env groups on one XCD (they share the rows' 128-B lines) -- scan_kernels.hip's
// gae_reg_kernel has the reasoning.  Features are accumulated in the reference's order per sample: bit-identical values.
template <int LMAX, int EW>
__global__ void __launch_bounds__(1024)
path_scan_reg_kernel(int T, int n, int L, int Do, const uint8_t* __restrict__ done, const float* __restrict__ obs,
                     const double* __restrict__ coeffs, int whole_paths, int32_t* __restrict__ tin,
                     uint8_t* __restrict__ valid, double* __restrict__ values) {
    extern __shared__ int s_scan[];            // [KB][EW] last path start inside the chunk or -1 | [KB][EW] any done inside
    __shared__ double s_w[2 * MAX_DO + 4];
    const int e = threadIdx.x, k = threadIdx.y, KB = blockDim.y;
    const int G = gridDim.x, w = blockIdx.x;
    const int g = (G % 8 == 0) ? (w % 8) * (G / 8) + w / 8 : w;
    const int i = g * EW + e;
    const int t0 = k * L, t1 = min(T, t0 + L);
    const bool live = (i < n) && (t0 < T);
    const int F = 2 * Do + 4;
    if (coeffs)
        for (int f = threadIdx.y * EW + threadIdx.x; f < F; f += EW * KB) s_w[f] = coeffs[f];
    uint32_t dn = 0;                           // bit j: done[t0 + j]
    bool prev_done = true;                     // done[t0 - 1] (a path starts at t = 0)
    if (live) {
        uint8_t dd[LMAX];
#pragma unroll
        for (int j = 0; j < LMAX; ++j) dd[j] = done[(size_t)(t0 + j < t1 ? t0 + j : t0) * n + i];
        if (t0 > 0) prev_done = done[(size_t)(t0 - 1) * n + i] != 0;
#pragma unroll
        for (int j = 0; j < LMAX; ++j) dn |= (t0 + j < t1 && dd[j]) ? (1u << j) : 0u;
    }
    int last = -1;
    if (live) {
        // path starts inside the chunk: t0 if prev_done, t0 + j + 1 for every done bit j with t0 + j + 1 < t1
        const uint32_t starts = ((dn << 1) | (prev_done ? 1u : 0u)) & ((L >= 32) ? 0xffffffffu : ((1u << (t1 - t0)) - 1u));
        if (starts) last = t0 + (31 - __clz(starts));
    }
    s_scan[k * EW + e] = last;
    s_scan[(KB + k) * EW + e] = dn != 0;
    __syncthreads();
    if (!live) return;
    int cur = -1;                      // last start before this chunk
    for (int j = 0; j < k; ++j) cur = max(cur, s_scan[j * EW + e]);
    int later = 0;                     // any done after this chunk
    for (int j = k + 1; j < KB; ++j) later |= s_scan[(KB + j) * EW + e];
    int ti[LMAX];
    bool start = prev_done;
#pragma unroll
    for (int j = 0; j < LMAX; ++j) {
        if (start) cur = t0 + j;
        ti[j] = t0 + j - cur;
        start = (dn >> j) & 1u;
    }
#pragma unroll
    for (int j = 0; j < LMAX; ++j) {
        if (t0 + j < t1) {
            const size_t off = (size_t)(t0 + j) * n + i;
            tin[off] = ti[j];
            valid[off] = (!whole_paths || (dn >> j) != 0u || later) ? 1 : 0;     // a done flag at or after this step
        }
    }
    if (!values) return;
    double acc[LMAX];
#pragma unroll
    for (int j = 0; j < LMAX; ++j) acc[j] = 0.0;
    if (coeffs) {
        const size_t plane = (size_t)T * n;
        for (int d = 0; d < Do; ++d) {
            float ob[LMAX];
#pragma unroll
            for (int j = 0; j < LMAX; ++j) ob[j] = obs[(size_t)d * plane + (size_t)(t0 + j < t1 ? t0 + j : t0) * n + i];
            const double w0 = s_w[d], w1 = s_w[Do + d];
#pragma unroll
            for (int j = 0; j < LMAX; ++j) {
                double o = (double)ob[j];
                o = fmin(fmax(o, -10.0), 10.0);
                acc[j] = fma(w0, o, acc[j]);
                acc[j] = fma(w1, o * o, acc[j]);
            }
        }
        const double wa = s_w[2 * Do], wb = s_w[2 * Do + 1], wc = s_w[2 * Do + 2], wd = s_w[2 * Do + 3];
#pragma unroll
        for (int j = 0; j < LMAX; ++j) {
            const double al = (double)ti[j] / 100.0;
            acc[j] = fma(wa, al, acc[j]);
            acc[j] = fma(wb, al * al, acc[j]);
            acc[j] = fma(wc, al * al * al, acc[j]);
            acc[j] = fma(wd, 1.0, acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < LMAX; ++j)
        if (t0 + j < t1) values[(size_t)(t0 + j) * n + i] = acc[j];
}

// ---------------------------------------------------------------------------------------------
// one-pass batch statistics
// ---------------------------------------------------------------------------------------------
enum {
    ST_COUNT = 0,      // valid samples
    ST_RET, ST_RET2,   // sum / sum of squares of returns (shifted by ret_shift)
    ST_BASE, ST_BASE2, // baseline predictions (shifted by ret_shift)
    ST_RES, ST_RES2,   // returns - baseline
    ST_ADV, ST_ADV2,   // advantages
    ST_NPATH,          // valid paths (starts)
    ST_UND, ST_UND2,   // undiscounted path return (shifted by und_shift) at path starts
    ST_DISC,           // discounted return at path starts
    ST_PROG, ST_PROG2, // per-path progress x[last step] - x[first step] of one observation component
    ST_NSUM,           // number of summed columns
    ST_ADVMIN = ST_NSUM, ST_UNDMAX, ST_UNDMIN, ST_PROGMAX, ST_PROGMIN,
    ST_NCOLS
};
constexpr int ST_BLOCK = 256;

struct StatAcc {
    double s[ST_NSUM];
    double adv_min, und_max, und_min, prog_max, prog_min;
};

__device__ __forceinline__ bool stat_is_max(int c) { return c == ST_UNDMAX || c == ST_PROGMAX; }

__global__ void __launch_bounds__(ST_BLOCK)
sample_stats_kernel(size_t B, const float* __restrict__ ret, const double* __restrict__ base,
                    const float* __restrict__ adv, const float* __restrict__ undisc,
                    const int32_t* __restrict__ tin, const uint8_t* __restrict__ valid, double ret_shift,
                    double und_shift, const float* __restrict__ prog, int n_cols, double* __restrict__ partial) {
    StatAcc a;
#pragma unroll
    for (int c = 0; c < ST_NSUM; ++c) a.s[c] = 0.0;
    a.adv_min = INFINITY; a.und_max = -INFINITY; a.und_min = INFINITY;
    a.prog_max = -INFINITY; a.prog_min = INFINITY;
    for (size_t b = (size_t)blockIdx.x * ST_BLOCK + threadIdx.x; b < B; b += (size_t)gridDim.x * ST_BLOCK) {
        if (!valid[b]) continue;
        const double r = (double)ret[b] - ret_shift;
        const double v = (base ? base[b] : 0.0) - ret_shift;
        const double res = r - v;
        const double ad = (double)adv[b];
        a.s[ST_COUNT] += 1.0;
        a.s[ST_RET] += r; a.s[ST_RET2] += r * r;
        a.s[ST_BASE] += v; a.s[ST_BASE2] += v * v;
        a.s[ST_RES] += res; a.s[ST_RES2] += res * res;
        a.s[ST_ADV] += ad; a.s[ST_ADV2] += ad * ad;
        a.adv_min = fmin(a.adv_min, ad);
        if (tin[b] == 0) {
            const double u = (double)undisc[b];
            a.s[ST_NPATH] += 1.0;
            a.s[ST_UND] += u - und_shift; a.s[ST_UND2] += (u - und_shift) * (u - und_shift);
            a.s[ST_DISC] += (double)ret[b];
            a.und_max = fmax(a.und_max, u);
            a.und_min = fmin(a.und_min, u);
        }
        if (prog) {
            // last step of a path: the next step of this column starts a new path (or there is none)
            const bool last = (b + (size_t)n_cols >= B) || (tin[b + (size_t)n_cols] == 0);
            if (last) {
                const double p = (double)prog[b] - (double)prog[b - (size_t)tin[b] * (size_t)n_cols];
                a.s[ST_PROG] += p; a.s[ST_PROG2] += p * p;
                a.prog_max = fmax(a.prog_max, p);
                a.prog_min = fmin(a.prog_min, p);
            }
        }
    }
    __shared__ double sm[ST_BLOCK / 64][ST_NCOLS];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double cols[ST_NCOLS];
#pragma unroll
    for (int c = 0; c < ST_NSUM; ++c) cols[c] = a.s[c];
    cols[ST_ADVMIN] = a.adv_min; cols[ST_UNDMAX] = a.und_max; cols[ST_UNDMIN] = a.und_min;
    cols[ST_PROGMAX] = a.prog_max; cols[ST_PROGMIN] = a.prog_min;
#pragma unroll
    for (int c = 0; c < ST_NCOLS; ++c) {
        double v = cols[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double w = __shfl_xor(v, o, 64);
            v = (c < ST_NSUM) ? v + w : (stat_is_max(c) ? fmax(v, w) : fmin(v, w));
        }
        if (lane == 0) sm[wave][c] = v;
    }
    __syncthreads();
    if (threadIdx.x < ST_NCOLS) {
        const int c = threadIdx.x;
        double v = sm[0][c];
        for (int w = 1; w < ST_BLOCK / 64; ++w)
            v = (c < ST_NSUM) ? v + sm[w][c] : (stat_is_max(c) ? fmax(v, sm[w][c]) : fmin(v, sm[w][c]));
        partial[(size_t)blockIdx.x * ST_NCOLS + c] = v;
    }
}

// fold the per-workgroup rows (one wavefront per column, fixed order)
__global__ void __launch_bounds__(64) stats_reduce_kernel(const double* __restrict__ partial, int rows,
                                                          double* __restrict__ out) {
    const int c = blockIdx.x, lane = threadIdx.x;
    const bool is_sum = c < ST_NSUM, is_max = stat_is_max(c);
    double v = is_sum ? 0.0 : (is_max ? -INFINITY : INFINITY);
    for (int r = lane; r < rows; r += 64) {
        const double w = partial[(size_t)r * ST_NCOLS + c];
        v = is_sum ? v + w : (is_max ? fmax(v, w) : fmin(v, w));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double w = __shfl_xor(v, o, 64);
        v = is_sum ? v + w : (is_max ? fmax(v, w) : fmin(v, w));
    }
    if (lane == 0) out[c] = v;
}

__global__ void __launch_bounds__(256)
adv_finish_kernel(size_t B, const float* __restrict__ adv_in, const uint8_t* __restrict__ valid, double mean,
                  double denom, double shift, float* __restrict__ adv_out) {
    const size_t b = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const double a = ((double)adv_in[b] - mean) / denom + shift;
    adv_out[b] = valid[b] ? (float)a : 0.0f;
}

// ---------------------------------------------------------------------------------------------
// normal equations of the linear feature baseline
// ---------------------------------------------------------------------------------------------
constexpr int NE_TILE = 64;   // samples per tile (one per lane)
constexpr int NE_WAVES = 4;   // wavefronts per workgroup

// FB = side of the register block per lane; FE = 8 * FB extended features (phi, y, zero padding)
template <int FB>
__global__ void __launch_bounds__(NE_WAVES * 64)
lfb_normal_eq_kernel(size_t B, int Do, const float* __restrict__ obs, const int32_t* __restrict__ tin,
                     const float* __restrict__ ret, const uint8_t* __restrict__ valid,
                     double* __restrict__ partial) {
    constexpr int FE = 8 * FB;
    constexpr int STR = FE + 2;                       // row stride (f64), keeps 16-byte alignment
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double* tile = sm + (size_t)wave * NE_TILE * STR;  // [sample][feature]
    const int bi = lane >> 3, bj = lane & 7;           // block row / column of this lane
    const int F = 2 * Do + 4;
    double acc[FB][FB];
#pragma unroll
    for (int r = 0; r < FB; ++r)
#pragma unroll
        for (int c = 0; c < FB; ++c) acc[r][c] = 0.0;
    const size_t n_tiles = (B + NE_TILE - 1) / NE_TILE;
    for (size_t tl = (size_t)blockIdx.x * NE_WAVES + wave; tl < n_tiles; tl += (size_t)gridDim.x * NE_WAVES) {
        const size_t b = tl * NE_TILE + lane;
        const bool use = (b < B) && valid[b];
        double* row = tile + (size_t)lane * STR;
        if (use) {
            lfb_features(Do, obs, B, b, tin[b], [&](int f, double v) { row[f] = v; });
            row[F] = (double)ret[b];
            for (int f = F + 1; f < FE; ++f) row[f] = 0.0;
        } else {
            for (int f = 0; f < FE; ++f) row[f] = 0.0;
        }
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        __builtin_amdgcn_wave_barrier();
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
#pragma unroll 4
        for (int s = 0; s < NE_TILE; ++s) {
            const double* rs = tile + (size_t)s * STR;
            double a[FB], bb[FB];
#pragma unroll
            for (int r = 0; r < FB; ++r) { a[r] = rs[bi * FB + r]; bb[r] = rs[bj * FB + r]; }
#pragma unroll
            for (int r = 0; r < FB; ++r)
#pragma unroll
                for (int c = 0; c < FB; ++c) acc[r][c] = fma(a[r], bb[c], acc[r][c]);
        }
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        __builtin_amdgcn_wave_barrier();
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
    }
    // fold the wavefronts of the workgroup in order, one partial [FE][FE] per workgroup
    __syncthreads();
    double* red = sm;   // [FE][FE], aliases the tiles
    for (int w = 0; w < NE_WAVES; ++w) {
        if (wave == w) {
#pragma unroll
            for (int r = 0; r < FB; ++r)
#pragma unroll
                for (int c = 0; c < FB; ++c) {
                    double* p = red + (size_t)(bi * FB + r) * FE + bj * FB + c;
                    *p = (w == 0) ? acc[r][c] : *p + acc[r][c];
                }
        }
        __syncthreads();
    }
    double* out = partial + (size_t)blockIdx.x * FE * FE;
    for (int k = threadIdx.x; k < FE * FE; k += NE_WAVES * 64) out[k] = red[k];
}

// The same normal equations on the f64 matrix pipe: Gram = Phi^T Phi is D[16x16] += A[16x4] B[4x16] with the
// SAMPLE axis as K, A[m = feature i][k = sample] and B[k = sample][n = feature j] being the same numbers, so a lane
// reads ONE double per 16-feature tile and k-step (4 samples) from the [sample][feature] LDS tile instead of the
// 2 FB per sample of the register-blocked form above -- that kernel is bound by those LDS reads, not by its FMAs.
// Only the upper triangle of 16x16 tiles is accumulated; the mirror is filled when the partial is written.
// lane l: operand index (feature) l % 16, k = l / 16; accumulator register j holds row 4 j + l / 16, column l % 16
// (the f64 instruction interleaves the rows of the lane groups; tools/ubench/mfma_f64_layout.hip prints the map).
using f64x4 = __attribute__((ext_vector_type(4))) double;

template <int FE>
__global__ void __launch_bounds__(NE_WAVES * 64)
lfb_normal_eq_mfma_kernel(size_t B, int Do, const float* __restrict__ obs, const int32_t* __restrict__ tin,
                          const float* __restrict__ ret, const uint8_t* __restrict__ valid,
                          double* __restrict__ partial) {
    constexpr int NT = FE / 16;
    constexpr int STR = FE + 2;
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double* tile = sm + (size_t)wave * NE_TILE * STR;  // [sample][feature]
    const int lm = lane & 15, kq = lane >> 4;
    const int F = 2 * Do + 4;
    f64x4 acc[NT][NT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
    const size_t n_tiles = (B + NE_TILE - 1) / NE_TILE;
    // A sample's inputs travel ONE TILE AHEAD in registers, every load of a tile issued together (round 6: the feature
    // loop used to load an observation component, use it, load the next -- Do dependent round trips per tile in front of
    // the matrix instructions).  Same features, same sums in the same order: bit-identical output.
    float xo[MAX_DO], rt = 0.0f;
    int tn = 0;
    bool use_n = false;
    auto fetch = [&](size_t t_) {
        const size_t b = t_ * NE_TILE + lane;
        const bool in = (t_ < n_tiles) && (b < B);
        const size_t bc = in ? b : 0;
        use_n = in && valid[bc];
        tn = tin[bc];
        rt = ret[bc];
#pragma unroll
        for (int d = 0; d < MAX_DO; ++d) xo[d] = obs[(size_t)(d < Do ? d : 0) * B + bc];
    };
    const size_t stride = (size_t)gridDim.x * NE_WAVES;
    size_t tl = (size_t)blockIdx.x * NE_WAVES + wave;
    fetch(tl);
    for (; tl < n_tiles; tl += stride) {
        const bool use = use_n;
        double* row = tile + (size_t)lane * STR;
        if (use) {
#pragma unroll
            for (int d = 0; d < MAX_DO; ++d)
                if (d < Do) {
                    double o = (double)xo[d];
                    o = fmin(fmax(o, -10.0), 10.0);
                    row[d] = o;
                    row[Do + d] = o * o;
                }
            const double al = (double)tn / 100.0;
            row[2 * Do] = al;
            row[2 * Do + 1] = al * al;
            row[2 * Do + 2] = al * al * al;
            row[2 * Do + 3] = 1.0;
            row[F] = (double)rt;
            for (int f = F + 1; f < FE; ++f) row[f] = 0.0;
        } else {
            for (int f = 0; f < FE; ++f) row[f] = 0.0;
        }
        fetch(tl + stride);          // (clamped to sample 0 beyond the batch: no branch around the loads)
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        __builtin_amdgcn_wave_barrier();
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
#pragma unroll 4
        for (int s4 = 0; s4 < NE_TILE / 4; ++s4) {
            const double* rs = tile + (size_t)(4 * s4 + kq) * STR + lm;
            double op[NT];
#pragma unroll
            for (int a = 0; a < NT; ++a) op[a] = rs[16 * a];
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int c = a; c < NT; ++c)
                    acc[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[a], op[c], acc[a][c], 0, 0, 0);
        }
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        __builtin_amdgcn_wave_barrier();
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
    }
    // fold the wavefronts of the workgroup in order, one partial [FE][FE] per workgroup (both triangles)
    __syncthreads();
    double* red = sm;   // [FE][FE], aliases the tiles
    for (int w = 0; w < NE_WAVES; ++w) {
        if (wave == w) {
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int c = a; c < NT; ++c)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        double* p = red + (size_t)(16 * a + 4 * j + kq) * FE + 16 * c + lm;
                        *p = (w == 0) ? acc[a][c][j] : *p + acc[a][c][j];
                    }
        }
        __syncthreads();
    }
    double* out = partial + (size_t)blockIdx.x * FE * FE;
    for (int k = threadIdx.x; k < FE * FE; k += NE_WAVES * 64) {
        const int i = k / FE, j = k % FE;
        out[k] = (i / 16 <= j / 16) ? red[k] : red[(size_t)j * FE + i];     // lower tiles: mirror of the upper ones
    }
}

// out = gram (F*F, row-major) followed by rhs (F): sum of the per-workgroup partials, one wavefront per
// output entry (lanes take interleaved rows, butterfly at the end: fixed order, deterministic)
__global__ void __launch_bounds__(256)
lfb_reduce_kernel(const double* __restrict__ partial, int rows, int FE, int F, double* __restrict__ out) {
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (k >= (F + 1) * F) return;
    const int i = (k < F * F) ? k / F : k - F * F;   // feature row
    const int j = (k < F * F) ? k % F : F;           // feature column (F = the y column)
    double s = 0.0;
    for (int r = lane; r < rows; r += 64) s += partial[(size_t)r * FE * FE + (size_t)i * FE + j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) out[k] = s;
}

}  // namespace rl

using namespace rl;

extern "C" int rl_path_scan(int T, int n, int obs_dim, const uint8_t* dones, const float* obs,
                            const double* coeffs, int whole_paths, int32_t* tin, uint8_t* valid,
                            double* values, void* stream) {
    if (T <= 0 || n <= 0 || !dones || !tin || !valid || obs_dim < 0 || obs_dim > MAX_DO ||
        (values && coeffs && !obs))
        return set_error(RL_ERR_ARG, "rl_path_scan: bad argument (obs_dim <= %d)", MAX_DO);
    constexpr int LMAX = 16;
#ifndef RL_PS_NARROW_BELOW
#define RL_PS_NARROW_BELOW 1024
#endif
#ifndef RL_PS_KB_NARROW
#define RL_PS_KB_NARROW 32           // as rl_gae: 32 chunks of 16 steps per 16-env workgroup (27.7 against 30.1 us at 4096 x 500)
#endif
    const bool narrow = (n + 31) / 32 < RL_PS_NARROW_BELOW;
    const int ew = narrow ? 16 : 32, kb_max = narrow ? RL_PS_KB_NARROW : 32;
    if (T <= kb_max * LMAX) {
        const int L = (T + kb_max - 1) / kb_max, KB = (T + L - 1) / L;
        dim3 grid((n + ew - 1) / ew), block(ew, KB);
        const size_t lds = (size_t)2 * KB * ew * sizeof(int);
        if (narrow)
            hipLaunchKernelGGL((path_scan_reg_kernel<LMAX, 16>), grid, block, lds, (hipStream_t)stream, T, n, L, obs_dim, dones,
                               obs, coeffs, whole_paths, tin, valid, values);
        else
            hipLaunchKernelGGL((path_scan_reg_kernel<LMAX, 32>), grid, block, lds, (hipStream_t)stream, T, n, L, obs_dim, dones,
                               obs, coeffs, whole_paths, tin, valid, values);
        return check_launch("path_scan_reg_kernel");
    }
    dim3 grid((n + PS_EW - 1) / PS_EW), block(PS_EW, PS_KB);
    hipLaunchKernelGGL(path_scan_kernel, grid, block, 0, (hipStream_t)stream, T, n, obs_dim, dones, obs, coeffs,
                       whole_paths, tin, valid, values);
    return check_launch("path_scan_kernel");
}

extern "C" int rl_sample_stats_cols(void) { return ST_NCOLS; }

extern "C" size_t rl_process_workspace_bytes(int obs_dim) {
    const int FE = (2 * obs_dim + 5 <= 32) ? 32 : 64;
    const size_t a = (size_t)1024 * ST_NCOLS * sizeof(double);
    const size_t b = (size_t)512 * FE * FE * sizeof(double);
    return a > b ? a : b;
}

extern "C" int rl_sample_stats(size_t n_samples, const float* returns, const double* baselines,
                               const float* advantages, const float* undiscounted, const int32_t* tin,
                               const uint8_t* valid, double ret_shift, double und_shift, const float* progress,
                               int n_cols, void* workspace, size_t workspace_bytes, double* out, void* stream) {
    if (n_samples == 0 || !returns || !advantages || !undiscounted || !tin || !valid || !workspace || !out ||
        (progress && n_cols <= 0))
        return set_error(RL_ERR_ARG, "rl_sample_stats: bad argument");
    int grid = (int)((n_samples + ST_BLOCK * 8 - 1) / (ST_BLOCK * 8));
    if (grid > 1024) grid = 1024;
    if (grid < 1) grid = 1;
    if (workspace_bytes < (size_t)grid * ST_NCOLS * sizeof(double))
        return set_error(RL_ERR_ARG, "rl_sample_stats: workspace too small");
    hipLaunchKernelGGL(sample_stats_kernel, dim3(grid), dim3(ST_BLOCK), 0, (hipStream_t)stream, n_samples, returns,
                       baselines, advantages, undiscounted, tin, valid, ret_shift, und_shift, progress, n_cols,
                       (double*)workspace);
    hipLaunchKernelGGL(stats_reduce_kernel, dim3(ST_NCOLS), dim3(64), 0, (hipStream_t)stream,
                       (const double*)workspace, grid, out);
    return check_launch("sample_stats_kernel");
}

extern "C" int rl_adv_finish(size_t n_samples, const float* adv_in, const uint8_t* valid, double mean,
                             double denom, double shift, float* adv_out, void* stream) {
    if (n_samples == 0 || !adv_in || !valid || !adv_out) return set_error(RL_ERR_ARG, "rl_adv_finish: bad argument");
    hipLaunchKernelGGL(adv_finish_kernel, dim3((unsigned)((n_samples + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, n_samples, adv_in, valid, mean, denom, shift, adv_out);
    return check_launch("adv_finish_kernel");
}

extern "C" int rl_lfb_normal_eq(size_t n_samples, int obs_dim, const float* obs, const int32_t* tin,
                                const float* returns, const uint8_t* valid, void* workspace,
                                size_t workspace_bytes, double* out, int variant, void* stream) {
    if (n_samples == 0 || obs_dim <= 0 || obs_dim > MAX_DO || !obs || !tin || !returns || !valid || !workspace || !out)
        return set_error(RL_ERR_ARG, "rl_lfb_normal_eq: bad argument (obs_dim <= %d)", MAX_DO);
    const int F = 2 * obs_dim + 4;
    const int FE = (F + 1 <= 32) ? 32 : 64;
    const size_t n_tiles = (n_samples + NE_TILE - 1) / NE_TILE;
    int grid = (int)((n_tiles + NE_WAVES - 1) / NE_WAVES);
    if (grid > 512) grid = 512;
    if (workspace_bytes < (size_t)grid * FE * FE * sizeof(double))
        return set_error(RL_ERR_ARG, "rl_lfb_normal_eq: workspace too small");
    const size_t lds = (size_t)NE_WAVES * NE_TILE * (FE + 2) * sizeof(double);
    hipError_t e = hipSuccess;
    // variant = 1 (rl_launch_opts.lfb_valu) selects the register-blocked vector kernel (A/B timing; same sums up to
    // association): an argument, so that the parity test can run both forms in one process
    const bool valu = variant == 1;
    if (FE == 32 && !valu) {
        static bool set = false;
        if (!set) { e = hipFuncSetAttribute(reinterpret_cast<const void*>(lfb_normal_eq_mfma_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); set = true; }
        hipLaunchKernelGGL(lfb_normal_eq_mfma_kernel<32>, dim3(grid), dim3(NE_WAVES * 64), lds, (hipStream_t)stream, n_samples,
                           obs_dim, obs, tin, returns, valid, (double*)workspace);
    } else if (!valu) {
        static bool set = false;
        if (!set) { e = hipFuncSetAttribute(reinterpret_cast<const void*>(lfb_normal_eq_mfma_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); set = true; }
        hipLaunchKernelGGL(lfb_normal_eq_mfma_kernel<64>, dim3(grid), dim3(NE_WAVES * 64), lds, (hipStream_t)stream, n_samples,
                           obs_dim, obs, tin, returns, valid, (double*)workspace);
    } else if (FE == 32) {
        static bool set32 = false;
        if (!set32) { e = hipFuncSetAttribute(reinterpret_cast<const void*>(lfb_normal_eq_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); set32 = true; }
        hipLaunchKernelGGL(lfb_normal_eq_kernel<4>, dim3(grid), dim3(NE_WAVES * 64), lds, (hipStream_t)stream, n_samples,
                           obs_dim, obs, tin, returns, valid, (double*)workspace);
    } else {
        static bool set64 = false;
        if (!set64) { e = hipFuncSetAttribute(reinterpret_cast<const void*>(lfb_normal_eq_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); set64 = true; }
        hipLaunchKernelGGL(lfb_normal_eq_kernel<8>, dim3(grid), dim3(NE_WAVES * 64), lds, (hipStream_t)stream, n_samples,
                           obs_dim, obs, tin, returns, valid, (double*)workspace);
    }
    if (e != hipSuccess) return set_error(RL_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(lfb_reduce_kernel, dim3(((F + 1) * F + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                       (const double*)workspace, grid, FE, F, out);
    return check_launch("lfb_normal_eq_kernel");
}
