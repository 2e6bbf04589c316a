// policy_split16_kernels.hip -- the split-operand Fisher-vector product of a (32, 32) tanh GaussianMLPPolicy
// (policy_split_kernels.hip: three-way bf16 split of every f32 operand, six cross terms, f32 accumulation) on
// 16-SAMPLE tiles and v_mfma_f32_16x16x32_bf16, three or four wavefronts per SIMD (round 6; the structure rounds 3 - 5
// left untested).  An A/B variant behind rl_launch_opts.fvp_split = 3: it ties with fvp_split_kernel at three wavefronts
// per SIMD and loses at four (profiles/r06_notes.md section 6) -- fvp_split_kernel stays the library's choice.  Same inputs, same partial-row / float64 row reduction, a result that differs from fvp_split_kernel by
// rounding only (rllab/optimizers/conjugate_gradient_optimizer.py:27-55 at theta_new == theta_old).
//
// Why another shape: fvp_split_kernel's tile is one dependency chain (split -> products -> element-wise -> split ...)
// and two wavefronts per SIMD leave its matrix pipe 36 % and its vector issue 35 % busy WITHOUT overlapping
// (profiles/r05_notes.md section 1).  A 16-sample tile halves every fragment (a 32-unit fragment is 8 registers), so a
// wavefront fits 128 registers and a SIMD holds four of them.
//
// Mapping.  Lane l = (s = l & 15, g = l >> 4).  A "sample-major" fragment holds, for sample s of the tile, the eight units
// U(g, i) = 16 (g >> 1) + 8 (i >> 2) + 4 (g & 1) + (i & 3), i = 0 .. 7:
//   * it is what two result tiles (row blocks t = 0, 1; register r) of a product hold when row m = 4 g + r of the A
//     operand's block t carries unit U(g, 4 t + r);
//   * packed to bf16 it IS the B operand of the next layer (k-slot (g, j) = register j), one instruction contracts over
//     all 32 units;
//   * it is what the gradient pass left in the activation cache: row q = 2 (g >> 1) + (i >> 2) of the 32-sample tile's
//     [h0 | h1][4] rows, lane slot (sample, half = g & 1), a float4 -- two 16-byte LDS-direct loads per layer and lane.
// Sample-contracted products (gW1 += h0^T gz1, gW0 += x^T gz0): a part is transposed on the matrix pipe (part as the A
// operand x a selector of 16 features), lane (feature n, g) receives samples 4 g .. 4 g + 3: FOUR of the eight k-slots of
// the outer product's operands.  The other four carry a second PART of the same samples, so one instruction adds two
// cross terms (A = [hi | mid], B = [hi | hi] gives hi hi + mid hi): three instructions per outer-product block instead
// of six, and the matrix time per sample is that of the 32-sample kernel (87 x 16 cycles per 16 samples).
// Vector code is PLAIN f32 / integer instructions throughout: packed-f32 and dot instructions do not run beside the
// matrix pipe (tools/ubench/mfma_bf16_valu_overlap.hip; MI355X_MICROARCH.md "price of one filler beside MFMAs").
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "../../include/rllab_amd.h"
#include "capi_util.h"
#include "policy_mfma.h"

namespace rl {

int launch_reduce_rows(const float* partial, int rows, int cols, double* out, hipStream_t st);   // policy_kernels.hip

namespace split16 {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int H = 32;
constexpr int T16 = 16;                 // samples per tile
// wavefronts per SIMD: 4 (128 registers) or 3 (168); one workgroup of 4 WPS wavefronts per CU

// build-time switch: the residual a - bf16(a) of a pair as two v_dot2c_f32_bf16 (policy_split_kernels.hip's
// RL_SPLIT_DOT2: 7 instead of 11 instructions per pair) -- off: dot instructions cost the matrix pipe of the SIMD
#ifndef RL_SPLIT16_DOT2
#define RL_SPLIT16_DOT2 0
#endif
// build-time switch: residual subtractions and element-wise stages on packed-f32 instructions (two values per issue slot).
// The kernel is bound by the SUM of its issue slots at three / four wavefronts per SIMD (profiles/r06_notes.md section 6),
// so fewer slots are worth more than the overlap packed instructions forgo.
#ifndef RL_SPLIT16_PK
#define RL_SPLIT16_PK 1
#endif

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
    int ablate;                // timing ablations (rl_launch_opts.reserved[0], WRONG results): 1 = the tile loop re-uses the first
                               // tile's cached fragments, 2 = ... its observations / weight (no LDS-direct pieces in the loop)
};

struct Parts { u32x4 p[3]; };                  // hi, mid, lo of eight values (bf16 pairs)
struct TParts { unsigned v[3][2]; };           // a transposed block: [part][pair of samples]

__host__ __device__ constexpr int unit_of(int g, int i) { return 16 * (g >> 1) + 8 * (i >> 2) + 4 * (g & 1) + (i & 3); }

__device__ __forceinline__ f32x4 mfma(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// c += A B to f32 accuracy: the six cross terms, smallest first
__device__ __forceinline__ f32x4 mm6(const Parts& A, const Parts& B, f32x4 c) {
    c = mfma(A.p[1], B.p[1], c);
    c = mfma(A.p[0], B.p[2], c);
    c = mfma(A.p[2], B.p[0], c);
    c = mfma(A.p[0], B.p[1], c);
    c = mfma(A.p[1], B.p[0], c);
    c = mfma(A.p[0], B.p[0], c);
    return c;
}
// the same six terms over FOUR samples per lane group: the upper four k-slots carry a second part
__device__ __forceinline__ f32x4 outer6(const TParts& A, const TParts& B, f32x4 c) {
    c = mfma(u32x4{A.v[2][0], A.v[2][1], A.v[1][0], A.v[1][1]}, u32x4{B.v[0][0], B.v[0][1], B.v[1][0], B.v[1][1]}, c);   // lo hi + mid mid
    c = mfma(u32x4{A.v[0][0], A.v[0][1], A.v[0][0], A.v[0][1]}, u32x4{B.v[1][0], B.v[1][1], B.v[2][0], B.v[2][1]}, c);   // hi mid + hi lo
    c = mfma(u32x4{A.v[0][0], A.v[0][1], A.v[1][0], A.v[1][1]}, u32x4{B.v[0][0], B.v[0][1], B.v[0][0], B.v[0][1]}, c);   // hi hi + mid hi
    return c;
}

__device__ __forceinline__ unsigned cvt_pk(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
#if RL_SPLIT16_DOT2
__device__ __forceinline__ bf16x2 dot2_selector(unsigned bits) {      // (in a register: policy_split_kernels.hip's finding)
    unsigned v;
    asm("v_mov_b32 %0, %1" : "=v"(v) : "s"(bits));
    return __builtin_bit_cast(bf16x2, v);
}
#endif
// (a0, a1) - their packed bf16 rounding h, exact
__device__ __forceinline__ void residual(float a0, float a1, unsigned h, float& r0, float& r1) {
#if RL_SPLIT16_DOT2
    const bf16x2 e0 = dot2_selector(0x0000bf80u), e1 = dot2_selector(0xbf800000u);
    r0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, h), e0, a0, false);
    r1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, h), e1, a1, false);
#elif RL_SPLIT16_PK
    const f32x2 r = f32x2{a0, a1} - f32x2{__uint_as_float(h << 16), __uint_as_float(h & 0xffff0000u)};
    r0 = r[0]; r1 = r[1];
#else
    r0 = a0 - __uint_as_float(h << 16);
    r1 = a1 - __uint_as_float(h & 0xffff0000u);
#endif
}
// element-wise helpers over register pairs
__device__ __forceinline__ void pk_dtanh(const float (&h)[8], float (&dz)[8]) {          // dz = 1 - h h
#if RL_SPLIT16_PK
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x2 hh = {h[2 * j], h[2 * j + 1]};
        const f32x2 d = __builtin_elementwise_fma(-hh, hh, f32x2{1.0f, 1.0f});
        dz[2 * j] = d[0]; dz[2 * j + 1] = d[1];
    }
#else
#pragma unroll
    for (int i = 0; i < 8; ++i) dz[i] = __builtin_fmaf(-h[i], h[i], 1.0f);
#endif
}
__device__ __forceinline__ void pk_mul(const f32x4 (&acc)[2], const float (&d)[8], float (&out)[8]) {   // out = acc d
#if RL_SPLIT16_PK
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x2 p = f32x2{acc[j >> 1][2 * (j & 1)], acc[j >> 1][2 * (j & 1) + 1]} * f32x2{d[2 * j], d[2 * j + 1]};
        out[2 * j] = p[0]; out[2 * j + 1] = p[1];
    }
#else
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = acc[i >> 2][i & 3] * d[i];
#endif
}
// x = hi + mid + lo, each a bf16: successive round-to-nearest residuals; written stage by stage over the four pairs
// of a fragment (independent instructions next to each other)
__device__ __forceinline__ void split8(const float (&v)[8], Parts& out) {
    float r[8], l[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) out.p[0][j] = cvt_pk(v[2 * j], v[2 * j + 1]);
#pragma unroll
    for (int j = 0; j < 4; ++j) residual(v[2 * j], v[2 * j + 1], out.p[0][j], r[2 * j], r[2 * j + 1]);
#pragma unroll
    for (int j = 0; j < 4; ++j) out.p[1][j] = cvt_pk(r[2 * j], r[2 * j + 1]);
#pragma unroll
    for (int j = 0; j < 4; ++j) residual(r[2 * j], r[2 * j + 1], out.p[1][j], l[2 * j], l[2 * j + 1]);
#pragma unroll
    for (int j = 0; j < 4; ++j) out.p[2][j] = cvt_pk(l[2 * j], l[2 * j + 1]);
}
// sample-major parts x selector of 16 features -> lane (feature n, g) holds samples 4 g .. 4 g + 3 of every part (exact:
// every output is one product with 1)
__device__ __forceinline__ void transpose16(const Parts& f, u32x4 sel, TParts& out) {
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        f32x4 d = {0.0f, 0.0f, 0.0f, 0.0f};
        d = mfma(f.p[p], sel, d);
        out.v[p][0] = cvt_pk(d[0], d[1]);
        out.v[p][1] = cvt_pk(d[2], d[3]);
    }
}
// sum over the four lane groups g of a sample (lanes s, s + 16, s + 32, s + 48), in every one of them
__device__ __forceinline__ float group_sum(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float t = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    const auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(t), __float_as_uint(t), false, false);
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}

template <int DO, int DA, int WPS>
struct Shape {
    using N = Net<DO, DA, H>;
    static constexpr int WAVES = 4 * WPS;
    static constexpr int KX = (DO + 1 + 15) / 16;            // 16-row blocks of x_ext on the transposed side
    static constexpr int N_OPS = 8;                          // dW0^T, dW1^T, W1^T, W1: two row blocks each
    static constexpr int OPS_BYTES = N_OPS * 3 * WV * 16;
    static constexpr int N_SEL = 2 + KX;                     // feature selectors: two unit blocks, KX input blocks
    static constexpr int SEL_BYTES = N_SEL * WV * 16;
    static constexpr int TAILV = 16 * DA + 8;                // floats per lane group: W2 | dW2 | db1 at its units
    static constexpr int TAIL_BYTES = 4 * TAILV * 4;
    static constexpr int LAND_BYTES = 4 * WV * 16;           // h0 (2 rows) | h1 (2 rows) of 64 lanes x 16 B
    static constexpr int XLAND_BYTES = WV * 16;              // 16 rows (inputs, then the weight) x 16 samples, ONE 1 KB piece
    static constexpr int PARK_BYTES = 2 * WV * 16;           // 1 - h0^2 waits here for the back-propagation (8 registers)
    static constexpr int WAVE_BYTES = LAND_BYTES + XLAND_BYTES + PARK_BYTES;
    static constexpr int LDS_TOTAL = WAVES * WAVE_BYTES + OPS_BYTES + SEL_BYTES + TAIL_BYTES;
    static_assert(DO + 1 <= 16, "inputs + the weight row travel as ONE 16-row piece");
    static_assert(DA <= 2, "per-lane sums of the output layer's outer product: 8 DA registers");
    static_assert(LDS_TOTAL >= WAVES * N::P * 4 && LDS_TOTAL <= 160 * 1024, "LDS budget; the fold rows alias everything");
};

template <int DO, int DA, int WPS>
__global__ void __launch_bounds__(4 * WPS * WV) fvp_split16_kernel(Args a) {
    using S = Shape<DO, DA, WPS>;
    using N = typename S::N;
    constexpr int P = N::P, KX = S::KX, WAVES = S::WAVES;
    constexpr int O_DW0 = 0, O_DW1 = 2, O_W1T = 4, O_W1 = 6;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x / WV, lane = threadIdx.x % WV;
    const int s = lane & 15, g = lane >> 4, half = g & 1, qh = g >> 1;
    char* const land = smem + wave * S::WAVE_BYTES;
    char* const xland = land + S::LAND_BYTES;
    char* const park = xland + S::XLAND_BYTES;
    char* const ops = smem + WAVES * S::WAVE_BYTES;                       // [N_OPS][3][64] x 16 B
    char* const sels = ops + S::OPS_BYTES;                                // [N_SEL][64] x 16 B
    float* const tailv = reinterpret_cast<float*>(sels + S::SEL_BYTES);   // [4][TAILV]

    const int B = a.B;
    const int n_tiles = B / T16;
    const int wave_global = blockIdx.x * WAVES + wave;
    const int waves_total = gridDim.x * WAVES;

    // ---- the next tile's inputs: LDS-direct loads, invisible to the compiler's wait counts (policy_split_kernels.hip) ----
    const unsigned land_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t)land);
    const unsigned xland_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t)xland);
    const int xrow = (lane >> 3) + 8 * ((lane >> 2) & 1);
    auto fetch = [&](int tile, int skip) {
        // cached fragments: 32-sample tile tile >> 1, row (layer, q = 2 qh + jj), lane slot (sample, half)
        const float* src = a.acts + (((size_t)(tile >> 1) * 8 + 2 * qh) * WV + (16 * (tile & 1) + s + 32 * half)) * 4;
        if (!(skip & 1))
        asm volatile("s_mov_b32 m0, %2\n\t"
                     "global_load_lds_dwordx4 %0, off\n\t"
                     "global_load_lds_dwordx4 %0, off offset:1024\n\t"
                     "s_mov_b32 m0, %3\n\t"
                     "global_load_lds_dwordx4 %1, off\n\t"
                     "global_load_lds_dwordx4 %1, off offset:1024"
                     :: "v"(src), "v"(src + 4 * WV * 4), "s"(land_lds), "s"(land_lds + 2048u) : "memory");
        // observations + weight: ONE piece.  Lane L carries four consecutive samples (quad L & 3) of row d = (L >> 3) +
        // 8 ((L >> 2) & 1): rows < DO are inputs, row DO is the weight (its slot in x_ext is the constant 1) -- an LDS-direct
        // piece costs the CU ~150 cycles whatever it carries (first build: nine 256-byte pieces here, 0.415 ms)
        const float* xp = (xrow < DO ? a.obs + (size_t)xrow * B : a.weight) + tile * T16 + 4 * (lane & 3);
        if (!(skip & 2))
        asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(xp), "s"(xland_lds) : "memory");
    };
    if (wave_global < n_tiles) fetch(wave_global, 0);
    asm volatile("" ::: "memory");   // (the first tile's inputs travel while the operands below are staged)

    const float* __restrict__ th = a.theta;
    const float* __restrict__ vc = a.vec;
    // ---- loop-invariant operands, split once per workgroup ---------------------------------------------------------
    // block o = 2 fam + t: A[m = s][k-slot (g, j)], row m of block t = unit U(m >> 2, 4 t + (m & 3))
    for (int o = wave; o < S::N_OPS; o += WAVES) {
        const int fam = o >> 1, t = o & 1;
        const int ua = unit_of(s >> 2, 4 * t + (s & 3));
        float tv8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int d = 8 * g + j, ub = unit_of(g, j);
            tv8[j] = fam == 0 ? (d < DO ? vc[N::W0 + d * H + ua] : (d == DO ? vc[N::B0 + ua] : 0.0f))   // dW0^T (+ db0)
                   : fam == 1 ? vc[N::W1 + ub * H + ua]          // dW1^T: A[a][b] = dW1[b][a]
                   : fam == 2 ? th[N::W1 + ub * H + ua]          // W1^T
                              : th[N::W1 + ua * H + ub];         // W1:    A[b][a] = W1[b][a]
        }
        Parts tp;
        split8(tv8, tp);
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(ops + ((o * 3 + p) * WV + lane) * 16) = tp.p[p];
    }
    for (int o = wave; o < S::N_SEL; o += WAVES) {             // B[k-slot (g, j)][n = s] = (feature of the slot == 16 block + n)
        u32x4 sel;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            unsigned w = 0;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int j = 2 * jj + e;
                const int f = o < 2 ? unit_of(g, j) - 16 * o : 8 * g + j - 16 * (o - 2);
                if (f == s) w |= 0x3f80u << (16 * e);           // bf16 1.0
            }
            sel[jj] = w;
        }
        *reinterpret_cast<u32x4*>(sels + (o * WV + lane) * 16) = sel;
    }
    for (int e = threadIdx.x; e < 4 * 8; e += WAVES * WV) {
        const int gg = e / 8, i = e % 8, u = unit_of(gg, i);
        float* tv = tailv + gg * S::TAILV;
#pragma unroll
        for (int k = 0; k < DA; ++k) {
            tv[k * 8 + i] = th[N::W2 + u * DA + k];
            tv[(DA + k) * 8 + i] = vc[N::W2 + u * DA + k];
        }
        tv[2 * DA * 8 + i] = vc[N::B1 + u];
    }
    __syncthreads();
    auto op = [&](int o) -> Parts {
        Parts t;
#pragma unroll
        for (int p = 0; p < 3; ++p) t.p[p] = *reinterpret_cast<const u32x4*>(ops + ((o * 3 + p) * WV + lane) * 16);
        return t;
    };
    auto sel_of = [&](int o) -> u32x4 { return *reinterpret_cast<const u32x4*>(sels + (o * WV + lane) * 16); };
    const float* const tv = tailv + g * S::TAILV;
    auto row8 = [&](int off, float (&out)[8]) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(tv + off), hi = *reinterpret_cast<const f32x4*>(tv + off + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { out[e] = lo[e]; out[4 + e] = hi[e]; }
    };
    auto stage = [&]() { asm volatile("" ::: "memory"); };      // keeps a stage's LDS reads inside it

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
    f32x4 gW1[2][2], gW0[KX][2];
    float gW2l[8][DA], gb1l[8], gb2[DA], wsum = 0.0f;
#pragma unroll
    for (int b1 = 0; b1 < 2; ++b1)
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2) gW1[b1][b2] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int bx = 0; bx < KX; ++bx)
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2) gW0[bx][b2] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        gb1l[i] = 0.0f;
#pragma unroll
        for (int k = 0; k < DA; ++k) gW2l[i][k] = 0.0f;
    }
#pragma unroll
    for (int k = 0; k < DA; ++k) gb2[k] = 0.0f;

    // every load the compiler knows of has landed before the loop (policy_split_kernels.hip: the loop header would
    // otherwise inherit "loads may be pending" and wait for the hidden loads of the next tile)
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
    for (int tile = wave_global; tile < n_tiles; tile += waves_total) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int nxt = tile + waves_total < n_tiles ? tile + waves_total : tile;     // (the last tile prefetches itself)
        float h0[8], xv[8];
        {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(land + lane * 16);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(land + (WV + lane) * 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) { h0[e] = v0[e]; h0[4 + e] = v1[e]; }
        }
        // x[d][sample] sits at byte 128 (d & 7) + 64 (d >> 3) + 4 sample of the piece (conflict-free: the lane groups
        // beyond the inputs read their neighbours' addresses -- a broadcast)
        const char* xl = xland + 64 * half + 4 * s;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int d = 8 * g + j;
            const float v = *reinterpret_cast<const float*>(xl + 128 * j);
            xv[j] = d < DO ? v : (d == DO ? 1.0f : 0.0f);
        }
        const float wgt = *reinterpret_cast<const float*>(xland + 128 * (DO & 7) + 64 * (DO >> 3) + 4 * s);
        stage();

        // ---- tangent forward: dmu = J v ---------------------------------------------------------------------------------
        Parts Xs, H0s;
        split8(xv, Xs);
        split8(h0, H0s);
        f32x4 acc[2];
        acc[0] = acc[1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        {
            acc[0] = mm6(op(O_DW0), Xs, acc[0]);                            // dW0^T x + db0
            acc[1] = mm6(op(O_DW0 + 1), Xs, acc[1]);
        }
        TParts Xt[KX];
#pragma unroll
        for (int bx = 0; bx < KX; ++bx) transpose16(Xs, sel_of(2 + bx), Xt[bx]);
        stage();
        float dh0[8];
        {
            float dz0[8];
            pk_dtanh(h0, dz0);
            pk_mul(acc, dz0, dh0);
            *reinterpret_cast<f32x4*>(park + lane * 16) = f32x4{dz0[0], dz0[1], dz0[2], dz0[3]};
            *reinterpret_cast<f32x4*>(park + (WV + lane) * 16) = f32x4{dz0[4], dz0[5], dz0[6], dz0[7]};
        }
        {
            float b1r[8];
            row8(2 * DA * 8, b1r);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i >> 2][i & 3] = b1r[i];
            acc[0] = mm6(op(O_DW1), H0s, acc[0]);                           // dW1^T h0
            acc[1] = mm6(op(O_DW1 + 1), H0s, acc[1]);
        }
        TParts H0t[2];
        transpose16(H0s, sel_of(0), H0t[0]);
        transpose16(H0s, sel_of(1), H0t[1]);
        stage();
        {
            Parts D0s;
            split8(dh0, D0s);
            acc[0] = mm6(op(O_W1T), D0s, acc[0]);                           // W1^T dh0
            acc[1] = mm6(op(O_W1T + 1), D0s, acc[1]);
        }
        stage();
        // ---- the second layer's activations; the landing zones are free again: the next tile starts travelling --------
        float h1[8];
        {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(land + (2 * WV + lane) * 16);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(land + (3 * WV + lane) * 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) { h1[e] = v0[e]; h1[4 + e] = v1[e]; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        fetch(nxt, a.ablate);
        const float c = wgt * a.inv_count;
        float dz1[8], dg[8], gmu[DA];                                       // dg: dh1, then gz1
        pk_dtanh(h1, dz1);
        pk_mul(acc, dz1, dg);
#pragma unroll
        for (int k = 0; k < DA; ++k) {
            float W2r[8], dW2r[8];
            row8(k * 8, W2r);
            row8((DA + k) * 8, dW2r);
#if RL_SPLIT16_PK
            f32x2 q0 = {0.0f, 0.0f}, q1 = {0.0f, 0.0f};                     // two chains of pairs
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                q0 = __builtin_elementwise_fma(f32x2{h1[2 * j], h1[2 * j + 1]}, f32x2{dW2r[2 * j], dW2r[2 * j + 1]}, q0);
                q1 = __builtin_elementwise_fma(f32x2{dg[2 * j], dg[2 * j + 1]}, f32x2{W2r[2 * j], W2r[2 * j + 1]}, q1);
            }
            const f32x2 q = q0 + q1;
            const float p0 = q[0], p1 = q[1];
#else
            float p0 = 0.0f, p1 = 0.0f;                                     // two chains
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                p0 = __builtin_fmaf(h1[i], dW2r[i], p0);
                p1 = __builtin_fmaf(dg[i], W2r[i], p1);
            }
#endif
            const float dmu = db2[k] + group_sum(p0 + p1);
            gmu[k] = c * dmu * fk[k];
        }
        if (g == 0) {
            wsum += c;
#pragma unroll
            for (int k = 0; k < DA; ++k) gb2[k] += gmu[k];
        }
        stage();
        // ---- back-propagation, sample-major ---------------------------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < 8; ++i) dg[i] = 0.0f;
#pragma unroll
        for (int k = 0; k < DA; ++k) {
            float W2r[8];
            row8(k * 8, W2r);
#if RL_SPLIT16_PK
            const f32x2 gk = {gmu[k], gmu[k]};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x2 d = __builtin_elementwise_fma(f32x2{W2r[2 * j], W2r[2 * j + 1]}, gk, f32x2{dg[2 * j], dg[2 * j + 1]});
                dg[2 * j] = d[0]; dg[2 * j + 1] = d[1];
                const f32x2 w = __builtin_elementwise_fma(f32x2{h1[2 * j], h1[2 * j + 1]}, gk, f32x2{gW2l[2 * j][k], gW2l[2 * j + 1][k]});
                gW2l[2 * j][k] = w[0]; gW2l[2 * j + 1][k] = w[1];
            }
#else
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                dg[i] = __builtin_fmaf(W2r[i], gmu[k], dg[i]);
                gW2l[i][k] = __builtin_fmaf(h1[i], gmu[k], gW2l[i][k]);
            }
#endif
        }
#if RL_SPLIT16_PK
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x2 d = f32x2{dg[2 * j], dg[2 * j + 1]} * f32x2{dz1[2 * j], dz1[2 * j + 1]};
            dg[2 * j] = d[0]; dg[2 * j + 1] = d[1];
            const f32x2 b = f32x2{gb1l[2 * j], gb1l[2 * j + 1]} + d;
            gb1l[2 * j] = b[0]; gb1l[2 * j + 1] = b[1];
        }
#else
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            dg[i] *= dz1[i];
            gb1l[i] += dg[i];
        }
#endif
        stage();
        Parts G1s;
        split8(dg, G1s);
        acc[0] = acc[1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        {
            acc[0] = mm6(op(O_W1), G1s, acc[0]);                            // W1 gz1
            acc[1] = mm6(op(O_W1 + 1), G1s, acc[1]);
        }
        {
            TParts G1t[2];
            transpose16(G1s, sel_of(0), G1t[0]);
            transpose16(G1s, sel_of(1), G1t[1]);
#pragma unroll
            for (int b1 = 0; b1 < 2; ++b1)
#pragma unroll
                for (int b2 = 0; b2 < 2; ++b2) gW1[b1][b2] = outer6(H0t[b1], G1t[b2], gW1[b1][b2]);   // gW1 += h0^T gz1
        }
        stage();
        {
            float gz0[8];
            const f32x4 z0 = *reinterpret_cast<const f32x4*>(park + lane * 16);
            const f32x4 z1 = *reinterpret_cast<const f32x4*>(park + (WV + lane) * 16);
            const float dz0[8] = {z0[0], z0[1], z0[2], z0[3], z1[0], z1[1], z1[2], z1[3]};
            pk_mul(acc, dz0, gz0);
            Parts G0s;
            split8(gz0, G0s);
            TParts G0t[2];
            transpose16(G0s, sel_of(0), G0t[0]);
            transpose16(G0s, sel_of(1), G0t[1]);
#pragma unroll
            for (int bx = 0; bx < KX; ++bx)
#pragma unroll
                for (int b2 = 0; b2 < 2; ++b2) gW0[bx][b2] = outer6(Xt[bx], G0t[b2], gW0[bx][b2]);    // gW0 += x_ext^T gz0
        }
    }
    // the last tile's (redundant) prefetch is still travelling into the landing zone, which the fold rows alias
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- fold the wavefronts of this workgroup in a fixed order, write ONE partial row ----------------------------------
    float b1s[8], w2s[8][DA];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float v = gb1l[i];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, WV);           // over the 16 samples of a lane group
        b1s[i] = v;
#pragma unroll
        for (int k = 0; k < DA; ++k) {
            float w = gW2l[i][k];
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) w += __shfl_xor(w, o, WV);
            w2s[i][k] = w;
        }
    }
    float b2s[DA];
#pragma unroll
    for (int k = 0; k < DA; ++k) b2s[k] = wave_sum(gb2[k]);
    const float ws = wave_sum(wsum);
    __syncthreads();
    float* const myrow = reinterpret_cast<float*>(smem) + wave * P;
#pragma unroll
    for (int b1 = 0; b1 < 2; ++b1)
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                myrow[N::W1 + (16 * b1 + 4 * g + r) * H + 16 * b2 + s] = gW1[b1][b2][r];   // row = unit of h0, column = unit of gz1
#pragma unroll
    for (int bx = 0; bx < KX; ++bx)
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int d = 16 * bx + 4 * g + r;                            // row = input (or the bias slot)
                if (d < DO) myrow[N::W0 + d * H + 16 * b2 + s] = gW0[bx][b2][r];
                else if (d == DO) myrow[N::B0 + 16 * b2 + s] = gW0[bx][b2][r];
            }
    if (s == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int u = unit_of(g, i);
            myrow[N::B1 + u] = b1s[i];
#pragma unroll
            for (int k = 0; k < DA; ++k) myrow[N::W2 + u * DA + k] = w2s[i][k];
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < DA; ++k) {
            myrow[N::B2 + k] = b2s[k];
            // log_std block of the Fisher: d2KL/ds2 = 4 v (2 v - eps) / (2 v + eps)^2, v = sigma^2
            const float vv = var_[k], e = 1e-8f;
            const float cc = floored[k] ? 0.0f : 4.0f * vv * (2.0f * vv - e) / ((2.0f * vv + e) * (2.0f * vv + e));
            myrow[N::LSTD + k] = cc * vc[N::LSTD + k] * ws;
        }
    }
    __syncthreads();
    float* row = a.partial + (size_t)blockIdx.x * P;
    for (int k = threadIdx.x; k < P; k += WAVES * WV) {
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) t += reinterpret_cast<const float*>(smem)[w * P + k];
        row[k] = t;
    }
}

template <int DO, int DA, int WPS>
static int launch(const rl_policy_batch* g, const float* vec, void* ws, size_t ws_bytes, double* out, hipStream_t st) {
    using S = Shape<DO, DA, WPS>;
    using N = typename S::N;
    constexpr int WAVES = S::WAVES;
    Args a;
    a.B = g->n_samples; a.theta = g->theta; a.vec = vec; a.acts = g->activations; a.obs = g->obs; a.weight = g->weights;
    a.inv_count = g->inv_count; a.log_min_std = g->log_min_std;
    a.ablate = g->opts ? g->opts->reserved[0] : 0;
    const int n_tiles = a.B / T16;
    int grid = (n_tiles + WAVES - 1) / WAVES;
    if (grid > 256) grid = 256;                   // one workgroup per CU, WPS wavefronts per SIMD
    const size_t need = (size_t)grid * N::P * sizeof(float);
    if (ws_bytes < need) return set_error(RL_ERR_ARG, "policy pass workspace too small: %zu < %zu bytes", ws_bytes, need);
    a.partial = (float*)ws;
    auto kern = fvp_split16_kernel<DO, DA, WPS>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           S::LDS_TOTAL);
        if (e != hipSuccess) return set_error(RL_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * WV), S::LDS_TOTAL, st, a);
    int rc = check_launch("fvp_split16_kernel");
    if (rc) return rc;
    return launch_reduce_rows(a.partial, grid, N::P, out, st);
}

}  // namespace split16

// (obs_dim, act_dim) pairs of the HIP-native envs whose (32, 32) policy has at most two actions (the per-lane sums of the
// output layer's outer product are 8 act_dim registers of the 128)
#define SPLIT16_SHAPES(X) X(4, 1) X(6, 1) X(11, 1) X(13, 2) X(13, 1)
bool split16_fvp_takes(const rl_policy_batch* g) {
    if (!g->activations || g->hidden2 != 0 || g->hidden0 != 32 || g->hidden1 != 32 || g->activation != RL_ACT_TANH ||
        g->layer_activations != 0 || g->n_samples <= 0 || g->n_samples % TS != 0)
        return false;
#define SPLITCASE(DO, DA) if (g->obs_dim == DO && g->act_dim == DA) return true;
    SPLIT16_SHAPES(SPLITCASE)
#undef SPLITCASE
    return false;
}
int split16_fvp_dispatch(const rl_policy_batch* g, const float* vec, void* ws, size_t ws_bytes, double* out, hipStream_t st) {
    if (!split16_fvp_takes(g)) return RL_SPLIT_NOT_TAKEN;
    // three wavefronts per SIMD (168 registers, nothing spilled) unless rl_launch_opts.fvp_split_wps = 4 asks for four (128
    // registers, 17 - 62 spilled: slower, profiles/r06_notes.md section 6)
    const bool three = !(g->opts && g->opts->fvp_split_wps == 4);
#define SPLITCASE(DO, DA) if (g->obs_dim == DO && g->act_dim == DA) \
        return three ? split16::launch<DO, DA, 3>(g, vec, ws, ws_bytes, out, st) : split16::launch<DO, DA, 4>(g, vec, ws, ws_bytes, out, st);
    SPLIT16_SHAPES(SPLITCASE)
#undef SPLITCASE
    return RL_SPLIT_NOT_TAKEN;
}

}  // namespace rl
