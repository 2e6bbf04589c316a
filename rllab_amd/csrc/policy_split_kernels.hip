// policy_split_kernels.hip -- the Fisher-vector product of a (32, 32) tanh GaussianMLPPolicy on the bf16 matrix
// pipe at f32 accuracy: every f32 operand is split three ways, x = hi + mid + lo with each part a bf16 (exact: the
// parts are successive round-to-nearest residuals), and a product of two operands is the sum of the six cross terms
// hi hi + hi mid + mid hi + hi lo + lo hi + mid mid on v_mfma_f32_32x32x16_bf16 with f32 accumulation.  The dropped
// terms are below 2^-26 of |a b| (an f32 fused multiply-add rounds at 2^-24); measured against float64 the six-term
// dot product is three times closer than an f32 fma chain (tools/ubench/bf16_split_layout.hip).
//
// Why: v_mfma_f32_32x32x2_f32 runs at the f32 VECTOR rate and shares the vector datapath (policy_kernels.hip's
// product retires a 32-sample tile in 7.9 k cycles against a floor of 5.06 k matrix + 1.4 k vector cycles,
// profiles/r03_notes.md); the bf16 pipe is 16 times faster and runs beside the vector ALU, so the six terms cost
// 6/16 of the f32 matrix cycles and the tile becomes vector-bound (the splits).
//
// What it computes: rl_policy_fvp on cached activations (rllab/optimizers/conjugate_gradient_optimizer.py:27-55 at
// theta_new == theta_old, see policy_kernels.hip's header) -- same inputs, same partial-row / float64 row reduction,
// a result that differs from policy_pass_kernel<N, MODE_FVP, true> by rounding only.
//
// Mapping.  One wavefront per SIMD (512 registers) owns tiles of 32 samples; every loop-invariant operand (the split
// fragments of dW0^T, dW1^T, W1^T, W1 and the output layer's rows) lives in registers.  All f32 fragments are
// "sample-major": lane (s = lane & 31, half = lane >> 5) holds sample s and the 16 units frag_unit(r, half) -- the
// layout the matrix pipe produces and the layout the gradient pass left the activations in.  The products whose
// contraction runs over SAMPLES (gW1 += h0^T gz1, gW0 += x^T gz0) need their operands unit-major: the bf16 parts
// are written to a wave-private [32 samples][72 B] LDS image and come back through ds_read_b64_tr_b16, the
// transposing read (4 samples of one unit per lane and instruction).  The thin products of the output layer
// accumulate per lane and are reduced over the lanes once per launch.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "../../include/rllab_amd.h"
#include "capi_util.h"
#include "policy_mfma.h"

namespace rl {

int launch_reduce_rows(const float* partial, int rows, int cols, double* out, hipStream_t st);   // policy_kernels.hip

namespace split {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int H = 32;
constexpr int WAVES = 4;                       // one per SIMD
constexpr int LAND_BYTES = 2 * H * TS * 4;     // h0 | h1 fragments of one tile, as the gradient pass stored them
constexpr int TR_STRIDE = 72;                  // bytes per sample row of a transposition image (32 bf16 + 8 B: the
                                               // 8-byte writes of 16 consecutive samples fall on 16 different slots)
constexpr int TR_PART = TS * TR_STRIDE;        // one part (hi / mid / lo) of a 32 x 32 fragment
constexpr int TR_BYTES = 3 * TR_PART;
constexpr int N_TR = 4;                        // h0, x, gz1, gz0
constexpr int WAVE_BYTES = LAND_BYTES + N_TR * TR_BYTES;
constexpr int LDS_BYTES = WAVES * WAVE_BYTES;

struct Args {
    int B;
    const float* theta;
    const float* vec;
    const float* acts;
    const float* obs;
    const float* weight;
    float inv_count;
    float log_min_std;
    float* partial;            // [grid][P]
};

struct Parts { bf16x8 p[3]; };                 // hi, mid, lo of eight values

__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// c += A B to f32 accuracy: the six cross terms, smallest first
__device__ __forceinline__ f32x16 mm6(const Parts& A, const Parts& B, f32x16 c) {
    c = mfma16(A.p[1], B.p[1], c);
    c = mfma16(A.p[0], B.p[2], c);
    c = mfma16(A.p[2], B.p[0], c);
    c = mfma16(A.p[0], B.p[1], c);
    c = mfma16(A.p[1], B.p[0], c);
    c = mfma16(A.p[0], B.p[0], c);
    return c;
}

// x = hi + mid + lo, each a bf16: successive round-to-nearest residuals (every subtraction is exact)
__device__ __forceinline__ void split_pair(float a0, float a1, Parts& out, int j) {
    const f32x2 a = {a0, a1};
    const bf16x2 h = __builtin_convertvector(a, bf16x2);
    const f32x2 r = a - __builtin_convertvector(h, f32x2);
    const bf16x2 m = __builtin_convertvector(r, bf16x2);
    const f32x2 l = r - __builtin_convertvector(m, f32x2);
    const bf16x2 q = __builtin_convertvector(l, bf16x2);
    out.p[0][j] = h[0]; out.p[0][j + 1] = h[1];
    out.p[1][j] = m[0]; out.p[1][j + 1] = m[1];
    out.p[2][j] = q[0]; out.p[2][j + 1] = q[1];
}
__device__ __forceinline__ void split8(const float* v, Parts& out) {
#pragma unroll
    for (int j = 0; j < 8; j += 2) split_pair(v[j], v[j + 1], out, j);
}
// a 16-register fragment -> the operands of its two k-blocks (registers 8 kb .. 8 kb + 7)
__device__ __forceinline__ void split_frag(const f32x16& v, Parts (&out)[2]) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int j = 0; j < 8; j += 2) split_pair(v[8 * kb + j], v[8 * kb + j + 1], out[kb], j);
}

// sample-major parts -> the transposition image: row = sample, column = unit; a lane's eight units of k-block kb are
// the two runs 16 kb + 8 h2 + 4 half + (0 .. 3)
__device__ __forceinline__ void store_parts_units(char* img, int lj, int lh, const Parts (&f)[2]) {
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const bf16x4 v = {f[kb].p[p][4 * h2], f[kb].p[p][4 * h2 + 1], f[kb].p[p][4 * h2 + 2], f[kb].p[p][4 * h2 + 3]};
                *reinterpret_cast<bf16x4*>(img + p * TR_PART + lj * TR_STRIDE + 2 * (16 * kb + 8 * h2 + 4 * lh)) = v;
            }
}
// the x fragment: a lane's eight inputs of k-block kb are the run 16 kb + 8 half + (0 .. 7)
template <int KB0>
__device__ __forceinline__ void store_parts_inputs(char* img, int lj, int lh, const Parts (&f)[KB0]) {
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int kb = 0; kb < KB0; ++kb)
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const bf16x4 v = {f[kb].p[p][4 * h2], f[kb].p[p][4 * h2 + 1], f[kb].p[p][4 * h2 + 2], f[kb].p[p][4 * h2 + 3]};
                *reinterpret_cast<bf16x4*>(img + p * TR_PART + lj * TR_STRIDE + 2 * (16 * kb + 8 * lh + 4 * h2)) = v;
            }
}
// unit-major operand of sample k-block kb: lane (unit = lane & 31, half) receives samples frag_unit(8 kb + j, half).
// `lane_off` = (4 half + ((lane & 15) >> 2)) * TR_STRIDE + 32 ((lane >> 4) & 1) + 8 (lane & 3)   (tools/ubench/bf16_split_layout.hip)
__device__ __forceinline__ void load_parts_transposed(const char* img, int lane_off, int kb, Parts& out) {
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const char* a = img + p * TR_PART + lane_off + (2 * kb) * 8 * TR_STRIDE;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a + 8 * TR_STRIDE));
        const bf16x4 l4 = __builtin_bit_cast(bf16x4, lo), h4 = __builtin_bit_cast(bf16x4, hi);
        out.p[p] = __builtin_shufflevector(l4, h4, 0, 1, 2, 3, 4, 5, 6, 7);
    }
}

template <int DO, int DA>
__global__ void __launch_bounds__(WAVES * WV, 1) fvp_split_kernel(Args a) {
    using N = Net<DO, DA, H>;
    constexpr int P = N::P;
    constexpr int KB0 = (DO + 1 + 15) / 16;        // k-blocks of the input layer (inputs + the bias slot)
    static_assert(DO + 1 <= 32 && LDS_BYTES >= P * 4, "");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x / WV, lane = threadIdx.x % WV;
    const int lj = lane & 31, lh = lane >> 5;
    char* const land = smem + wave * WAVE_BYTES;
    char* const img_h0 = land + LAND_BYTES;
    char* const img_x = img_h0 + TR_BYTES;
    char* const img_g1 = img_x + TR_BYTES;
    char* const img_g0 = img_g1 + TR_BYTES;
    const int lane_off = (4 * lh + ((lane & 15) >> 2)) * TR_STRIDE + 32 * ((lane >> 4) & 1) + 8 * (lane & 3);

    // the images start as zeros: columns no fragment writes (inputs beyond the bias slot) stay zero operands
    for (int k = lane * 16; k < N_TR * TR_BYTES; k += WV * 16) *reinterpret_cast<f32x4*>(img_h0 + k) = f32x4{0, 0, 0, 0};
    wave_sync();

    // ---- loop-invariant operands, split once per launch, register resident ---------------------------------------
    const float* __restrict__ th = a.theta;
    const float* __restrict__ vc = a.vec;
    Parts A1[KB0], A2[2], A3[2], A4[2];
    {
        float t[8];
#pragma unroll
        for (int kb = 0; kb < KB0; ++kb) {            // A1[i = lj][d] = dW0[d][i] (d < DO), db0[i] (d == DO)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int d = 16 * kb + 8 * lh + j;
                t[j] = d < DO ? vc[N::W0 + d * H + lj] : (d == DO ? vc[N::B0 + lj] : 0.0f);
            }
            split8(t, A1[kb]);
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = vc[N::W1 + frag_unit(8 * kb + j, lh) * H + lj];     // dW1^T: A[i][k] = dW1[k][i]
            split8(t, A2[kb]);
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = th[N::W1 + frag_unit(8 * kb + j, lh) * H + lj];     // W1^T
            split8(t, A3[kb]);
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = th[N::W1 + lj * H + frag_unit(8 * kb + j, lh)];     // W1: A[k][i] = W1[k][i]
            split8(t, A4[kb]);
        }
    }
    float db1r[16], W2r[16][DA], dW2r[16][DA];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int u = frag_unit(r, lh);
        db1r[r] = vc[N::B1 + u];
#pragma unroll
        for (int k = 0; k < DA; ++k) {
            W2r[r][k] = th[N::W2 + u * DA + k];
            dW2r[r][k] = vc[N::W2 + u * DA + k];
        }
    }
    float db2[DA], fk[DA], var_[DA];
    bool floored[DA];
#pragma unroll
    for (int k = 0; k < DA; ++k) {
        const float raw = th[N::LSTD + k];
        floored[k] = raw < a.log_min_std;
        const float ls = fmaxf(raw, a.log_min_std);
        var_[k] = __expf(2.0f * ls);
        fk[k] = 2.0f / (2.0f * var_[k] + 1e-8f);
        db2[k] = vc[N::B2 + k];
    }

    // ---- accumulators ---------------------------------------------------------------------------------------------
    f32x16 gW1, gW0;
    float gW2l[16][DA], gb1l[16], gb2[DA], wsum = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        gW1[r] = 0.0f; gW0[r] = 0.0f; gb1l[r] = 0.0f;
#pragma unroll
        for (int k = 0; k < DA; ++k) gW2l[r][k] = 0.0f;
    }
#pragma unroll
    for (int k = 0; k < DA; ++k) gb2[k] = 0.0f;

    const int B = a.B;
    const int n_tiles = B / TS;
    const int wave_global = blockIdx.x * WAVES + wave;
    const int waves_total = gridDim.x * WAVES;

    // one tile ahead: the observation slots and the weight in registers, the cached activations by LDS-direct loads
    auto fetch = [&](int tile, float (&xq)[KB0][8], float& wq) {
        const int b = tile * TS + lj;
        wq = a.weight[b];
#pragma unroll
        for (int kb = 0; kb < KB0; ++kb)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int d = 16 * kb + 8 * lh + j;
                xq[kb][j] = d < DO ? a.obs[(size_t)d * B + b] : (d == DO ? 1.0f : 0.0f);
            }
    };
    auto fetch_acts = [&](int tile) {
        const float* src = a.acts + ((size_t)tile * 8 * WV + lane) * 4;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + q * WV * 4), (lptr_t)(land + q * WV * 16), 16, 0, 0);
    };
    float xb[KB0][8], xb_next[KB0][8];
    float wgt = 0.0f, wgt_next = 0.0f;
    if (wave_global < n_tiles) {
        fetch(wave_global, xb_next, wgt_next);
        fetch_acts(wave_global);
    }

    for (int tile = wave_global; tile < n_tiles; tile += waves_total) {
        // ---- this tile's inputs; the next tile's start travelling ------------------------------------------------
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        f32x16 h0, h1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(land + (q * WV + lane) * 16);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(land + ((4 + q) * WV + lane) * 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) { h0[4 * q + e] = v0[e]; h1[4 * q + e] = v1[e]; }
        }
#pragma unroll
        for (int kb = 0; kb < KB0; ++kb)
#pragma unroll
            for (int j = 0; j < 8; ++j) xb[kb][j] = xb_next[kb][j];
        wgt = wgt_next;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the landing zone is in registers before it is refilled
        if (tile + waves_total < n_tiles) {
            fetch(tile + waves_total, xb_next, wgt_next);
            fetch_acts(tile + waves_total);
        }

        // ---- operands of this tile: x, h0 (their unit-major forms start through LDS now) -----------------------------
        Parts Xs[KB0], H0s[2];
#pragma unroll
        for (int kb = 0; kb < KB0; ++kb) split8(xb[kb], Xs[kb]);
        split_frag(h0, H0s);
        store_parts_inputs<KB0>(img_x, lj, lh, Xs);
        store_parts_units(img_h0, lj, lh, H0s);
        f32x16 dz0, dz1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            dz0[r] = 1.0f - h0[r] * h0[r];
            dz1[r] = 1.0f - h1[r] * h1[r];
        }

        // ---- tangent forward: dmu = J v ---------------------------------------------------------------------------------
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int kb = 0; kb < KB0; ++kb) acc = mm6(A1[kb], Xs[kb], acc);          // dW0^T x + db0
        f32x16 dh0;
#pragma unroll
        for (int r = 0; r < 16; ++r) dh0[r] = acc[r] * dz0[r];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = db1r[r];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) acc = mm6(A2[kb], H0s[kb], acc);           // dW1^T h0
        {
            Parts D0s[2];
            split_frag(dh0, D0s);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) acc = mm6(A3[kb], D0s[kb], acc);       // W1^T dh0
        }
        f32x16 dh1;
#pragma unroll
        for (int r = 0; r < 16; ++r) dh1[r] = acc[r] * dz1[r];
        const float c = wgt * a.inv_count;
        float gmu[DA];
#pragma unroll
        for (int k = 0; k < DA; ++k) {
            float pd = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                pd = __builtin_fmaf(h1[r], dW2r[r][k], pd);
                pd = __builtin_fmaf(dh1[r], W2r[r][k], pd);
            }
            const float dmu = db2[k] + half_sum(pd);
            gmu[k] = c * dmu * fk[k];
        }
        if (lh == 0) {
            wsum += c;
#pragma unroll
            for (int k = 0; k < DA; ++k) gb2[k] += gmu[k];
        }

        // ---- back-propagation, sample-major ---------------------------------------------------------------------------------
        f32x16 gz1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float g = 0.0f;
#pragma unroll
            for (int k = 0; k < DA; ++k) {
                g = __builtin_fmaf(W2r[r][k], gmu[k], g);
                gW2l[r][k] = __builtin_fmaf(h1[r], gmu[k], gW2l[r][k]);
            }
            gz1[r] = g * dz1[r];
            gb1l[r] += gz1[r];
        }
        Parts G1s[2];
        split_frag(gz1, G1s);
        store_parts_units(img_g1, lj, lh, G1s);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) acc = mm6(A4[kb], G1s[kb], acc);           // W1 gz1
        f32x16 gz0;
#pragma unroll
        for (int r = 0; r < 16; ++r) gz0[r] = acc[r] * dz0[r];
        {
            Parts G0s[2];
            split_frag(gz0, G0s);
            store_parts_units(img_g0, lj, lh, G0s);
        }

        // ---- the batch reductions: gW1 += h0^T gz1, gW0 += x_ext^T gz0 (samples are K, operands unit-major) ---------------
        wave_sync();
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            Parts At, Bt;
            load_parts_transposed(img_h0, lane_off, kb, At);
            load_parts_transposed(img_g1, lane_off, kb, Bt);
            gW1 = mm6(At, Bt, gW1);
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            Parts At, Bt;
            load_parts_transposed(img_x, lane_off, kb, At);
            load_parts_transposed(img_g0, lane_off, kb, Bt);
            gW0 = mm6(At, Bt, gW0);
        }
        wave_sync();
    }

    // ---- fold the wavefronts of this workgroup in a fixed order, write ONE partial row ----------------------------------
    // per-lane accumulators of the thin products: sum over the 32 samples of a lane half
    float b1s[16], w2s[16][DA];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float v = gb1l[r];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, WV);
        b1s[r] = v;
#pragma unroll
        for (int k = 0; k < DA; ++k) {
            float w = gW2l[r][k];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) w += __shfl_xor(w, o, WV);
            w2s[r][k] = w;
        }
    }
    float b2s[DA];
#pragma unroll
    for (int k = 0; k < DA; ++k) b2s[k] = wave_sum(gb2[k]);
    const float ws = wave_sum(wsum);
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    for (int k = threadIdx.x; k < P; k += WAVES * WV) red[k] = 0.0f;
    __syncthreads();
    for (int w = 0; w < WAVES; ++w) {
        if (wave == w) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int u = frag_unit(r, lh);
                red[N::W1 + u * H + lj] += gW1[r];                    // row = unit of h0, column = unit of gz1
                if (u < DO) red[N::W0 + u * H + lj] += gW0[r];        // row = input (or the bias slot)
                else if (u == DO) red[N::B0 + lj] += gW0[r];
            }
            if (lj == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int u = frag_unit(r, lh);
                    red[N::B1 + u] += b1s[r];
#pragma unroll
                    for (int k = 0; k < DA; ++k) red[N::W2 + u * DA + k] += w2s[r][k];
                }
            }
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < DA; ++k) {
                    red[N::B2 + k] += b2s[k];
                    // log_std block of the Fisher: d2KL/ds2 = 4 v (2 v - eps) / (2 v + eps)^2, v = sigma^2
                    const float vv = var_[k], e = 1e-8f;
                    const float cc = floored[k] ? 0.0f : 4.0f * vv * (2.0f * vv - e) / ((2.0f * vv + e) * (2.0f * vv + e));
                    red[N::LSTD + k] += cc * vc[N::LSTD + k] * ws;
                }
            }
        }
        __syncthreads();
    }
    float* row = a.partial + (size_t)blockIdx.x * P;
    for (int k = threadIdx.x; k < P; k += WAVES * WV) row[k] = red[k];
}

template <int DO, int DA>
static int launch(const rl_policy_batch* g, const float* vec, void* ws, size_t ws_bytes, double* out, hipStream_t st) {
    using N = Net<DO, DA, H>;
    Args a;
    a.B = g->n_samples; a.theta = g->theta; a.vec = vec; a.acts = g->activations; a.obs = g->obs; a.weight = g->weights;
    a.inv_count = g->inv_count; a.log_min_std = g->log_min_std;
    const int n_tiles = a.B / TS;
    int grid = (n_tiles + WAVES - 1) / WAVES;
    if (grid > 256) grid = 256;                   // one workgroup per CU
    const size_t need = (size_t)grid * N::P * sizeof(float);
    if (ws_bytes < need) return set_error(RL_ERR_ARG, "policy pass workspace too small: %zu < %zu bytes", ws_bytes, need);
    a.partial = (float*)ws;
    auto kern = fvp_split_kernel<DO, DA>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           LDS_BYTES);
        if (e != hipSuccess) return set_error(RL_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * WV), LDS_BYTES, st, a);
    int rc = check_launch("fvp_split_kernel");
    if (rc) return rc;
    return launch_reduce_rows(a.partial, grid, N::P, out, st);
}

}  // namespace split

// The split product takes a cached Fisher-vector product of a two-layer 32-unit tanh net whose batch is a whole number
// of tiles; everything else stays on policy_pass_kernel.  RLLAB_FVP_SPLIT=0 switches it off (A/B runs, tests of the
// bit-identical cached / recomputed pair).  Returns RL_SPLIT_NOT_TAKEN when the launch is not its to make.
#define SPLIT_SHAPES(X) X(4, 1) X(6, 1) X(11, 1) X(13, 2) X(13, 1) X(20, 1) X(21, 1)
bool split_fvp_takes(const rl_policy_batch* g) {
    if (!g->activations || g->hidden2 != 0 || g->hidden0 != 32 || g->hidden1 != 32 || g->activation != RL_ACT_TANH ||
        g->n_samples <= 0 || g->n_samples % TS != 0)
        return false;
    const char* e = getenv("RLLAB_FVP_SPLIT");
    if (e && e[0] == '0') return false;
#define SPLITCASE(DO, DA) if (g->obs_dim == DO && g->act_dim == DA) return true;
    SPLIT_SHAPES(SPLITCASE)
#undef SPLITCASE
    return false;
}
int split_fvp_dispatch(const rl_policy_batch* g, const float* vec, void* ws, size_t ws_bytes, double* out, hipStream_t st) {
    if (!split_fvp_takes(g)) return RL_SPLIT_NOT_TAKEN;
#define SPLITCASE(DO, DA) if (g->obs_dim == DO && g->act_dim == DA) return split::launch<DO, DA>(g, vec, ws, ws_bytes, out, st);
    SPLIT_SHAPES(SPLITCASE)
#undef SPLITCASE
    return RL_SPLIT_NOT_TAKEN;
}

}  // namespace rl

extern "C" int rl_policy_fvp_variant(const rl_policy_batch* g) {
    if (!g) return rl::set_error(RL_ERR_ARG, "rl_policy_fvp_variant: null batch");
    return rl::split_fvp_takes(g) ? 1 : 0;
}
