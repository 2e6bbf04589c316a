// policy_splith_kernels.hip -- the Fisher-vector product of policy_split_kernels.hip (two 32- or 64-unit tanh layers on
// cached activations) with every f32 operand split TWO ways into f16 parts and THREE cross terms per product on
// v_mfma_f32_32x32x16_f16 (round 6; the three-way bf16 split needs six): 47 matrix instructions per 32-sample tile instead
// of 87, five vector instructions per split pair instead of seven.
//
// Arithmetic (Ootomo & Yokota's error-corrected f16 product, the lo part scaled by 2^11 so that it stays a normal number):
//   per-tile operand b:      hi = f16(b),  lo' = f16(2^11 (b - hi))          b = hi + 2^-11 lo' to ~2^-23 |b|, |b| in [2^-13, 2^15]
//   loop-invariant image A:  hi = f16(A),  lo = f16(A - hi),  hs = f16(2^-11 hi)   (A scaled by a power of two to sit high)
//   A b  = A.hi b.hi + A.lo b.hi + A.hs b.lo'                                 one f32 accumulator, three instructions
//   a^T b over samples (both per tile):  c0 += a.hi b.hi,  c1 += a.hi b.lo' + a.lo' b.hi,  result c0 + 2^-11 c1
//                                        or (H = 64: registers) a as an image: lo = 2^-11 lo', hs = 2^-11 hi
// The matrix pipe honours f16 subnormal inputs and, measured against float64 over ten magnitude regimes, this product is
// closer than an f32 fma chain (9e-8 against 1.0e-7 normwise; the bf16 six-term form 6.9e-8; tools/ubench/f16_split.hip,
// profiles/r06_f16_split.txt).  What f16 lacks is RANGE, so every operand class carries a power-of-two scale chosen per launch from
// maxima the kernel reduces itself (|v| per block, |W1|, |W2|) and from max |obs| (rl_policy_batch.obs_absmax, left there
// by the gradient pass that filled the activation cache), with WORST-CASE bounds for the intermediate operands: no f16
// operand can overflow whatever the data, and the scales cancel exactly (powers of two) in the fold.  A per-tile operand
// keeps an absolute floor of 2^-36 below its bound of 2^15 -- 51 binades; an image 2^-25 below a maximum of 2^4 .. 2^14.
//
// Same inputs, same partial-row / float64 row reduction, same tiling and data movement as fvp_split_kernel /
// fvp_split64_kernel (see that file for the mapping); results differ from theirs by rounding only
// (tests/test_gpu_fvp_split.py runs every test of the bf16 kernels on these too).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "../../include/rllab_amd.h"
#include "capi_util.h"
#include "policy_mfma.h"

namespace rl {

int launch_reduce_rows(const float* partial, int rows, int cols, double* out, hipStream_t st);   // policy_kernels.hip

namespace splith {

typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct Args {
    int B;
    const float* theta;
    const float* vec;
    const float* acts;
    const float* obs;
    const float* weight;
    const float* xmax;         // device: max |obs| over the batch (>= what this launch reads; the bias slot not included)
    float inv_count;
    float log_min_std;
    float* partial;            // [grid][P]
};

struct PB { f16x8 hi, lo; };        // per-tile operand: hi, lo' = 2^11 (b - hi)
struct PA { f16x8 hi, lo, hs; };    // image: hi, lo = A - hi, hs = 2^-11 hi

// timing ablations (WRONG results): no per-tile global loads / no matrix instructions / no split arithmetic
#ifndef RL_ABL_FETCH
#define RL_ABL_FETCH 0
#endif
#ifndef RL_ABL_MFMA
#define RL_ABL_MFMA 0
#endif
#ifndef RL_ABL_SPLIT
#define RL_ABL_SPLIT 0
#endif
#ifndef RL_ABL_TRANSPOSE
#define RL_ABL_TRANSPOSE 0       // the transpositions left out (the parts are handed on as they are): what transposing LDS reads could buy at most
#endif

__device__ __forceinline__ f32x16 mf(f16x8 a, f16x8 b, f32x16 c) {
#if RL_ABL_MFMA
    asm volatile("" : "+v"(c) : "v"(a), "v"(b));
    return c;
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#endif
}
// c += A b, smallest terms first
__device__ __forceinline__ f32x16 mm_ab(const PA& A, const PB& b, f32x16 c) {
    c = mf(A.hs, b.lo, c);
    c = mf(A.lo, b.hi, c);
    c = mf(A.hi, b.hi, c);
    return c;
}
// both operands per tile: c0 += a.hi b.hi, c1 += the cross terms (result c0 + 2^-11 c1)
__device__ __forceinline__ void mm_bb(const PB& a, const PB& b, f32x16& c0, f32x16& c1) {
    c1 = mf(a.hi, b.lo, c1);
    c1 = mf(a.lo, b.hi, c1);
    c0 = mf(a.hi, b.hi, c0);
}
// a per-tile operand of known good magnitude as an image (exact wherever the scaled part stays a normal number)
__device__ __forceinline__ PA as_image(const PB& a) {
    const f16 k = (f16)0.00048828125f;      // 2^-11
    PA r;
    r.hi = a.hi; r.lo = a.lo * k; r.hs = a.hi * k;
    return r;
}

__device__ __forceinline__ float half_sum_swap(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// b = hi + 2^-11 lo'.  RL_SPLITH_MIX: five instructions per pair -- v_cvt_pk_f16_f32, the residuals as two v_fma_mix_f32
// (f16 half x -1 + f32, exact), the scaled parts as v_fma_mixlo_f16 / v_fma_mixhi_f16 (f32 x 2^11 rounded to f16 once) --
// where the compiler's own selection is seven (two unpacking conversions, two packed f32 operations, a packed conversion);
// tools/ubench/f16_split.hip checks the two forms bit for bit on the device.
#ifndef RL_SPLITH_MIX
#define RL_SPLITH_MIX 1
#endif
__device__ __forceinline__ void split_pair(float a0, float a1, PB& out, int j) {
#if RL_SPLITH_MIX
#if RL_ABL_SPLIT
    unsigned hb, lb;
    asm volatile("" : "=v"(hb), "=v"(lb) : "v"(a0), "v"(a1));
#else
    // ONE asm statement: between separate ones the compiler puts a conservative s_nop (59 per tile in the first build)
    unsigned hb, lb;
    float r0, r1;
    const float k = 2048.0f;
    asm("v_cvt_pk_f16_f32 %0, %4, %5\n\t"
        "v_fma_mix_f32 %2, %0, -1.0, %4 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %3, %0, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %1, %2, %6, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixhi_f16 %1, %3, %6, 0 op_sel_hi:[0,0,0]"
        : "=&v"(hb), "=&v"(lb), "=&v"(r0), "=&v"(r1) : "v"(a0), "v"(a1), "s"(k));
#endif
    const f16x2 h = __builtin_bit_cast(f16x2, hb), l = __builtin_bit_cast(f16x2, lb);
#else
    const f16x2 h = __builtin_convertvector(f32x2{a0, a1}, f16x2);
    const float r0 = a0 - (float)h[0], r1 = a1 - (float)h[1];                                                // exact
    const f16x2 l = __builtin_convertvector(f32x2{r0, r1} * f32x2{2048.0f, 2048.0f}, f16x2);
#endif
    out.hi[j] = h[0]; out.hi[j + 1] = h[1];
    out.lo[j] = l[0]; out.lo[j + 1] = l[1];
}
// eight values (a k-block of one lane) in ONE asm statement: the compiler puts a conservative s_nop between separate asm
// statements (59 per tile in the first build), and a stage's instructions are issued stage by stage so that no instruction
// waits on the one in front of it
__device__ __forceinline__ void split8v(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7, PB& out) {
#if RL_SPLITH_MIX && !RL_ABL_SPLIT
    unsigned h0, h1, h2, h3, l0, l1, l2, l3;
    float r0, r1, r2, r3, r4, r5, r6, r7;
    const float k = 2048.0f;
    asm("v_cvt_pk_f16_f32 %0, %16, %17\n\t"
        "v_cvt_pk_f16_f32 %1, %18, %19\n\t"
        "v_cvt_pk_f16_f32 %2, %20, %21\n\t"
        "v_cvt_pk_f16_f32 %3, %22, %23\n\t"
        "v_fma_mix_f32 %8, %0, -1.0, %16 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %9, %0, -1.0, %17 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %10, %1, -1.0, %18 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %11, %1, -1.0, %19 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %12, %2, -1.0, %20 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %13, %2, -1.0, %21 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %14, %3, -1.0, %22 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %15, %3, -1.0, %23 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %4, %8, %24, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixlo_f16 %5, %10, %24, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixlo_f16 %6, %12, %24, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixlo_f16 %7, %14, %24, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixhi_f16 %4, %9, %24, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixhi_f16 %5, %11, %24, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixhi_f16 %6, %13, %24, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixhi_f16 %7, %15, %24, 0 op_sel_hi:[0,0,0]"
        : "=&v"(h0), "=&v"(h1), "=&v"(h2), "=&v"(h3), "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3),
          "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
        : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "s"(k));
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    out.hi = __builtin_bit_cast(f16x8, u32x4{h0, h1, h2, h3});
    out.lo = __builtin_bit_cast(f16x8, u32x4{l0, l1, l2, l3});
#else
    split_pair(a0, a1, out, 0); split_pair(a2, a3, out, 2); split_pair(a4, a5, out, 4); split_pair(a6, a7, out, 6);
#endif
}
__device__ __forceinline__ void split8(const float* v, PB& out) { split8v(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], out); }
__device__ __forceinline__ void split_frag(const f32x16& v, PB (&out)[2]) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
        split8v(v[8 * kb], v[8 * kb + 1], v[8 * kb + 2], v[8 * kb + 3], v[8 * kb + 4], v[8 * kb + 5], v[8 * kb + 6], v[8 * kb + 7], out[kb]);
}
// eight values of an image (already multiplied by the image's scale)
__device__ __forceinline__ PA image8(const float* t) {
    PA r;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const f16 h = (f16)t[j];
        r.hi[j] = h;
        r.lo[j] = (f16)(t[j] - (float)h);
        r.hs[j] = (f16)((float)h * 0.00048828125f);
    }
    return r;
}
// an f32 fragment whose values ARE f16 numbers (a transposed part) -> one part of the operands of its two k-blocks
__device__ __forceinline__ void pack_exact(const f32x16& d, f16x8& o0, f16x8& o1) {
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const f16x2 h0 = __builtin_convertvector(f32x2{d[j], d[j + 1]}, f16x2);
        const f16x2 h1 = __builtin_convertvector(f32x2{d[8 + j], d[8 + j + 1]}, f16x2);
        o0[j] = h0[0]; o0[j + 1] = h0[1];
        o1[j] = h1[0]; o1[j + 1] = h1[1];
    }
}
// sample-major parts of a 32-unit fragment -> unit-major parts (part x identity on the matrix pipe, exact)
__device__ __forceinline__ void transpose_units(const PB (&f)[2], const f16x8 (&Id)[2], PB (&out)[2]) {
#if RL_ABL_TRANSPOSE
    out[0] = f[0]; out[1] = f[1];
    return;
#endif
    f32x16 d;
#pragma unroll
    for (int r = 0; r < 16; ++r) d[r] = 0.0f;
    d = mf(f[0].hi, Id[0], d);
    d = mf(f[1].hi, Id[1], d);
    pack_exact(d, out[0].hi, out[1].hi);
#pragma unroll
    for (int r = 0; r < 16; ++r) d[r] = 0.0f;
    d = mf(f[0].lo, Id[0], d);
    d = mf(f[1].lo, Id[1], d);
    pack_exact(d, out[0].lo, out[1].lo);
}
template <int KB0>
__device__ __forceinline__ void transpose_inputs(const PB (&f)[KB0], const f16x8 (&Idx)[KB0], PB (&out)[2]) {
#if RL_ABL_TRANSPOSE
    out[0] = f[0]; out[1] = f[KB0 - 1];
    return;
#endif
    f32x16 d;
#pragma unroll
    for (int r = 0; r < 16; ++r) d[r] = 0.0f;
#pragma unroll
    for (int kb = 0; kb < KB0; ++kb) d = mf(f[kb].hi, Idx[kb], d);
    pack_exact(d, out[0].hi, out[1].hi);
#pragma unroll
    for (int r = 0; r < 16; ++r) d[r] = 0.0f;
#pragma unroll
    for (int kb = 0; kb < KB0; ++kb) d = mf(f[kb].lo, Idx[kb], d);
    pack_exact(d, out[0].lo, out[1].lo);
}

__device__ __forceinline__ f32x2 pair_of(const f32x16& v, int j) { return f32x2{v[2 * j], v[2 * j + 1]}; }
__device__ __forceinline__ void set_pair(f32x16& v, int j, f32x2 p) { v[2 * j] = p[0]; v[2 * j + 1] = p[1]; }
// out = acc * (1 - h h) * k
__device__ __forceinline__ void times_dtanh(const f32x16& acc, const f32x16& h, f32x16& out, float k = 1.0f) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const f32x2 hh = pair_of(h, j);
        const f32x2 d = __builtin_elementwise_fma(-hh, hh, f32x2{1.0f, 1.0f});
        set_pair(out, j, k == 1.0f ? pair_of(acc, j) * d : (pair_of(acc, j) * f32x2{k, k}) * d);
    }
}

// ---- the scales of a launch ------------------------------------------------------------------------------------------
// E(x): x < 2^E(x) for finite x >= 0 (the exponent of frexp), at least -30 (zeros, tiny blocks)
__device__ __forceinline__ int exp_of(float x) {
    const int e = (int)((__float_as_uint(x) >> 23) & 255u) - 126;
    return e < -30 ? -30 : e;
}
__device__ __forceinline__ float pow2(int e) { return __builtin_ldexpf(1.0f, e < -126 ? -126 : (e > 127 ? 127 : e)); }
__device__ __forceinline__ float uniform(float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
struct Scales {
    float sx;      // x~ = sx x (|x~| < 2^6)
    float sb;      // the bias slot of x~_ext (<= 2^5)
    float sv0;     // image dW0^T                       (< 2^4)
    float sv0b;    // ... its db0 row: sv0 sx / sb      (< 2^15)
    float sw1;     // images W1^T and W1                (< 2^7)
    float s1;      // image dW1^T, db1, dW2, db2: the scale of dz1 / dh1 / dmu  (image dW1^T < 2^14)
    float eg2;     // 2^e_g: gmu~ = wgt dmu~ (fk inv_count 2^e_g)      (|gz1~| < 2^15)
    float kg;      // gz0~ = kg (1 - h0 h0) (W1~ gz1~)                 (|gz0~| < 2^15)
    float u1;      // unscale of gb2, gW2, gb1, gW1
    float u0;      // unscale of gW0 (XT_SCALE included)
    float u0b;     // ... of gb0
};
// HL: log2 of the hidden width (5 or 6); DAL: ceil(log2(act_dim)); every argument a maximum of absolute values
template <int HL, int DAL>
__device__ __forceinline__ Scales make_scales(float bx, float bv0, float bv1, float bv2, float bw1, float bw2, float fkmax) {
    const int Ex = exp_of(fmaxf(bx, 1.0f)), Ev0 = exp_of(bv0), Ev1 = exp_of(bv1), Ev2 = exp_of(bv2), EW1 = exp_of(bw1),
              EW2 = exp_of(bw2), Efk = exp_of(fkmax);
    // the observations are scaled by THEIR maximum (bx excludes the bias slot): x~ < 2^6 whatever their size.  The bias slot
    // carries 2^e_b, e_b = min(e_x, 5) (it is transposed with the observations: < 2^6 as well), and the db0 row of the image
    // makes up the difference, so that every term of dz0~ has the scale 2^(e_v0 + e_x)
    const int e_x = 6 - exp_of(bx), e_b = imin(e_x, 5), e_w1 = 7 - EW1;
    const int xt = e_x <= 5 ? 11 : 1 + imax(11, e_x);                   // sum_d |x~_d| + 2^e_x < 2^xt  (DO <= 31)
    const int e_v0 = imin(15 - xt - Ev0, 14 - Ev1 - e_w1 - e_x);        // |dz0~| < 2^15; image dW1~^T < 2^14
    const int s0 = e_v0 + e_x, s1 = s0 + e_w1;
    const int EZ0 = 5 + Ev0 + Ex;                                       // |dz0| < (DO + 1 <= 32) |dW0|max |x_ext|max
    const int EZ1 = 1 + imax(HL + 1 + Ev1, HL + EW1 + EZ0);             // |dz1| < (H + 1) |dW1|max + H |W1|max |dh0|max
    const int EM = 1 + imax(HL + 1 + Ev2, HL + EW2 + EZ1);              // |dmu|
    const int EG1 = DAL + EW2 + EM + Efk;                               // |W2 gmu|, weights 0 / 1
    int sg = 15 - EG1;                                                  // s1 + e_g
    sg = sg < -100 ? -100 : (sg > 100 ? 100 : sg);
    const int kg = -(HL + 7);                                           // |W1~ gz1~| < H 2^7 2^15
    Scales s;      // (every lane computed the same numbers: scalar registers)
    s.sx = uniform(pow2(e_x)); s.sb = uniform(pow2(e_b)); s.sv0 = uniform(pow2(e_v0)); s.sv0b = uniform(pow2(e_v0 + e_x - e_b));
    s.sw1 = uniform(pow2(e_w1)); s.s1 = uniform(pow2(s1));
    s.eg2 = uniform(pow2(sg - s1)); s.kg = uniform(pow2(kg));
    s.u1 = uniform(pow2(-sg)); s.u0 = uniform(pow2(-(sg + e_w1 + kg + e_x + 8)));       // (+ 8: XT_SCALE)
    s.u0b = uniform(pow2(-(sg + e_w1 + kg + e_b + 8)));
    return s;
}
// maxima of |vec| over the blocks of the net and of |theta| over W1, W2, by the whole workgroup (every thread returns
// them); `red` = NW x 5 floats of LDS
template <class N, int NT>
__device__ __forceinline__ void block_maxima(const float* __restrict__ th, const float* __restrict__ vc, float* red,
                                             float (&m)[5]) {
#pragma unroll
    for (int i = 0; i < 5; ++i) m[i] = 0.0f;
    // every load of a thread is issued before the first is used (a rolled loop would sit out one memory round trip per
    // element: 23 of them for a (64, 64) net -- measured as +21 us per launch)
    constexpr int PER = (N::LSTD + NT - 1) / NT;
    float vv[PER], ww[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int e = threadIdx.x + i * NT, ec = e < N::LSTD ? e : N::LSTD - 1;
        vv[i] = vc[ec];
        ww[i] = (e >= N::W1 && e < N::B2) ? th[ec] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int e = threadIdx.x + i * NT;
        const float v = e < N::LSTD ? fabsf(vv[i]) : 0.0f, w = fabsf(ww[i]);
        if (e < N::W1) m[0] = fmaxf(m[0], v);
        else if (e < N::W2) m[1] = fmaxf(m[1], v);
        else m[2] = fmaxf(m[2], v);
        if (e >= N::W1 && e < N::B1) m[3] = fmaxf(m[3], w);
        if (e >= N::W2 && e < N::B2) m[4] = fmaxf(m[4], w);
    }
    const int wave = threadIdx.x / WV, lane = threadIdx.x % WV;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        m[i] = wave_max(m[i]);
        if (lane == 0) red[wave * 5 + i] = m[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        float v = 0.0f;
#pragma unroll
        for (int w = 0; w < NT / WV; ++w) v = fmaxf(v, red[w * 5 + i]);
        m[i] = v;
    }
}
__host__ __device__ constexpr int ceil_log2(int n) { return n <= 1 ? 0 : n <= 2 ? 1 : n <= 4 ? 2 : 3; }

// =====================================================================================================================
// (32, 32), two wavefronts per SIMD: fvp_split_kernel<DO, DA, 2> on the f16 arithmetic
constexpr int H = 32;
constexpr int LAND_BYTES = 2 * H * TS * 4;
constexpr int ops_count(int kb0) { return kb0 + 6; }
constexpr int ops_bytes(int kb0) { return ops_count(kb0) * 3 * WV * 16; }
#ifndef RL_SPLITH_TWO_ACC
#define RL_SPLITH_TWO_ACC 0      // gW1 += h0^T gz1~: two accumulators (1: 47 spilled registers with both) or h0 as an image (0)
#endif
#ifndef RL_SPLITH_TWO_ACC_W0
#define RL_SPLITH_TWO_ACC_W0 0   // gW0 += x~^T gz0~: two accumulators (1: 31 spilled registers) or x~ as an image (0)
#endif
// The sample-side operand as an image multiplies its parts by 2^-11, which must not push them into the f16 subnormals:
// h0 lives in (-1, 1) (exact from 2^-3 up, an absolute 2^-25 below), and the observations are transposed with 2^8 x identity
// (XT_SCALE; x~ < 2^6, so 2^8 x~ < 2^14): an observation down to 2^-17 of the batch maximum keeps every bit.
constexpr float XT_SCALE = 256.0f;

template <int DO, int DA, int WPS>
__global__ void __launch_bounds__(4 * WPS * WV, 1) fvp_splith_kernel(Args a) {
    using N = Net<DO, DA, H>;
    constexpr int P = N::P;
    constexpr int WAVES = 4 * WPS;                 // two wavefronts per SIMD; wide heads (16 DA registers of thin sums): one
    constexpr int KB0 = (DO + 1 + 15) / 16;
    constexpr int N_OPS = ops_count(KB0), OPS_BYTES = ops_bytes(KB0);
    constexpr int TAILV = 16 * DA * 2 + 16;        // floats per lane half: W2 rows | dW2~ rows | db1~
    constexpr int SPILL_BYTES = 2 * 2 * WV * 16;   // a wavefront's h0 parts wait here for the back-propagation
    constexpr int XPIECES = (DO + 1 + 7) / 8;
    constexpr int XLAND_BYTES = XPIECES * WV * 16;
    constexpr int WAVE_BYTES = LAND_BYTES + SPILL_BYTES + XLAND_BYTES;
    constexpr int LDS_TOTAL = WAVES * WAVE_BYTES + OPS_BYTES + 2 * TAILV * 4 + WAVES * 5 * 4;
    static_assert(DO + 1 <= 32 && (DA <= 2 || WPS == 1), "two k-blocks of inputs + the bias slot; thin heads at two per SIMD");
    static_assert(LDS_TOTAL >= WAVES * P * 4 && LDS_TOTAL <= 160 * 1024, "LDS budget; the fold rows alias the landing zones");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x / WV, lane = threadIdx.x % WV;
    const int lj = lane & 31, lh = lane >> 5;
    char* const land = smem + wave * WAVE_BYTES;
    char* const spill = land + LAND_BYTES;
    char* const xland = spill + SPILL_BYTES;
    char* const ops = smem + WAVES * WAVE_BYTES;                       // [N_OPS][3][64] x 16 B
    float* const tailv = reinterpret_cast<float*>(ops + OPS_BYTES);    // [2][TAILV]
    float* const red = tailv + 2 * TAILV;

    const int B = a.B;
    const int n_tiles = B / TS;
    const int wave_global = blockIdx.x * WAVES + wave;
    const int waves_total = gridDim.x * WAVES;

    // the next tile's observations + weight and cached activations travel by LDS-direct loads the compiler does not see
    // (policy_split_kernels.hip: RL_SPLIT_ASM_DMA)
    const unsigned xland_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t)xland);
    auto fetch = [&](int tile) {
#pragma unroll
        for (int p = 0; p < XPIECES; ++p) {
            const int d = 8 * p + (lane >> 3);
            const float* g = (d < DO ? a.obs + (size_t)d * B : a.weight) + tile * TS + 4 * (lane & 7);
            asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(g), "s"(xland_lds + p * (WV * 16)) : "memory");
        }
    };
    const unsigned land_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t)land);
    auto fetch_acts = [&](int tile) {
        const float* src = a.acts + ((size_t)tile * 8 * WV + lane) * 4;
        asm volatile("s_mov_b32 m0, %2\n\t"
                     "global_load_lds_dwordx4 %0, off\n\t"
                     "global_load_lds_dwordx4 %0, off offset:1024\n\t"
                     "global_load_lds_dwordx4 %0, off offset:2048\n\t"
                     "global_load_lds_dwordx4 %0, off offset:3072\n\t"
                     "s_mov_b32 m0, %3\n\t"
                     "global_load_lds_dwordx4 %1, off\n\t"
                     "global_load_lds_dwordx4 %1, off offset:1024\n\t"
                     "global_load_lds_dwordx4 %1, off offset:2048\n\t"
                     "global_load_lds_dwordx4 %1, off offset:3072"
                     :: "v"(src), "v"(src + 4 * WV * 4), "s"(land_lds), "s"(land_lds + 4096u) : "memory");
    };
    if (wave_global < n_tiles) {
        fetch(wave_global);
        fetch_acts(wave_global);
    }
    asm volatile("" ::: "memory");

    const float* __restrict__ th = a.theta;
    const float* __restrict__ vc = a.vec;
    // ---- the scales of this launch ---------------------------------------------------------------------------------------
    float db2[DA], fkS[DA], var_[DA];
    bool floored[DA];
    float fkmax = 0.0f;
#pragma unroll
    for (int k = 0; k < DA; ++k) {
        const float raw = th[N::LSTD + k];
        floored[k] = raw < a.log_min_std;
        const float ls = fmaxf(raw, a.log_min_std);
        var_[k] = __expf(2.0f * ls);
        fkS[k] = 2.0f / (2.0f * var_[k] + 1e-8f) * a.inv_count;
        fkmax = fmaxf(fkmax, fkS[k]);
    }
    float mx[5];
    block_maxima<N, WAVES * WV>(th, vc, red, mx);
    const Scales S = make_scales<5, ceil_log2(DA)>(*a.xmax, mx[0], mx[1], mx[2], mx[3], mx[4], fkmax);
#pragma unroll
    for (int k = 0; k < DA; ++k) {
        fkS[k] *= S.eg2;
        db2[k] = vc[N::B2 + k] * S.s1;
    }

    // ---- loop-invariant images, built once per launch ----------------------------------------------------------------------
    // block o: 0 .. KB0 - 1 = dW0^T (+ db0 in the bias slot), KB0 + kb = dW1^T, KB0 + 2 + kb = W1^T, KB0 + 4 + kb = W1
    for (int o = wave; o < N_OPS; o += WAVES) {
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (o < KB0) {
                const int d = 16 * o + 8 * lh + j;
                t[j] = d < DO ? vc[N::W0 + d * H + lj] * S.sv0 : (d == DO ? vc[N::B0 + lj] * S.sv0b : 0.0f);
            } else {
                const int q = o - KB0, kb = q & 1, u = frag_unit(8 * kb + j, lh);
                t[j] = q < 2 ? vc[N::W1 + u * H + lj] * S.s1
                     : q < 4 ? th[N::W1 + u * H + lj] * S.sw1
                             : th[N::W1 + lj * H + u] * S.sw1;
            }
        }
        const PA im = image8(t);
        *reinterpret_cast<f16x8*>(ops + ((o * 3 + 0) * WV + lane) * 16) = im.hi;
        *reinterpret_cast<f16x8*>(ops + ((o * 3 + 1) * WV + lane) * 16) = im.lo;
        *reinterpret_cast<f16x8*>(ops + ((o * 3 + 2) * WV + lane) * 16) = im.hs;
    }
    for (int e = threadIdx.x; e < 2 * 16; e += WAVES * WV) {
        const int hh = e / 16, r = e % 16, u = frag_unit(r, hh);
        float* tv = tailv + hh * TAILV;
#pragma unroll
        for (int k = 0; k < DA; ++k) {
            tv[k * 16 + r] = th[N::W2 + u * DA + k];
            tv[(DA + k) * 16 + r] = vc[N::W2 + u * DA + k] * S.s1;
        }
        tv[2 * DA * 16 + r] = vc[N::B1 + u] * S.s1;
    }
    __syncthreads();
    auto op = [&](int o) -> PA {
        PA t;
        t.hi = *reinterpret_cast<const f16x8*>(ops + ((o * 3 + 0) * WV + lane) * 16);
        t.lo = *reinterpret_cast<const f16x8*>(ops + ((o * 3 + 1) * WV + lane) * 16);
        t.hs = *reinterpret_cast<const f16x8*>(ops + ((o * 3 + 2) * WV + lane) * 16);
        return t;
    };
    // acc += sum_kb op(base + kb) x Bk[kb] with the operand blocks read one AHEAD
    auto chain_ahead = [&](int base, int nb, const PB* Bk, f32x16 acc_, PA& cur, int nxt_o) -> f32x16 {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if (kb < nb) {
                PA nxtp = cur;
                const int o = kb + 1 < nb ? base + kb + 1 : nxt_o;
                if (o >= 0) nxtp = op(o);
                __builtin_amdgcn_sched_barrier(0);
                acc_ = mm_ab(cur, Bk[kb], acc_);
                cur = nxtp;
            }
        }
        return acc_;
    };
    f16x8 Id[2], Idx[KB0];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        Id[0][j] = (f16)(frag_unit(j, lh) == lj ? 1.0f : 0.0f);
        Id[1][j] = (f16)(frag_unit(8 + j, lh) == lj ? 1.0f : 0.0f);
#pragma unroll
        for (int kb = 0; kb < KB0; ++kb) Idx[kb][j] = (f16)(16 * kb + 8 * lh + j == lj ? XT_SCALE : 0.0f);
    }

    // ---- accumulators ---------------------------------------------------------------------------------------------
    f32x16 gW1, gW0, gW1c, gW0c;
    float gW2l[16][DA], gb1l[16], gb2[DA], wsum = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        gW1[r] = 0.0f; gW0[r] = 0.0f; gW1c[r] = 0.0f; gW0c[r] = 0.0f; gb1l[r] = 0.0f;
#pragma unroll
        for (int k = 0; k < DA; ++k) gW2l[r][k] = 0.0f;
    }
#pragma unroll
    for (int k = 0; k < DA; ++k) gb2[k] = 0.0f;

    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): every load the compiler knows of has landed (policy_split_kernels.hip)
    for (int tile = wave_global; tile < n_tiles; tile += waves_total) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        f32x16 h0, h1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(land + (q * WV + lane) * 16);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(land + ((4 + q) * WV + lane) * 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) { h0[4 * q + e] = v0[e]; h1[4 * q + e] = v1[e]; }
        }
        float xb[KB0][8], wgt;
        {
            const float* xl = reinterpret_cast<const float*>(xland) + lj;
#pragma unroll
            for (int kb = 0; kb < KB0; ++kb)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int d = 16 * kb + 8 * lh + j;
                    const float v = xl[32 * (d < DO ? d : DO)];
                    xb[kb][j] = d < DO ? v * S.sx : (d == DO ? S.sb : 0.0f);
                }
            wgt = xl[32 * DO];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the landing zone is in registers before it is refilled
        {
            const int nxt = tile + waves_total < n_tiles ? tile + waves_total : tile;
#if !RL_ABL_FETCH
            fetch(nxt);
            fetch_acts(nxt);
#else
            (void)nxt;
#endif
        }
        const float* tv = tailv + lh * TAILV;
        auto stage = [&]() { asm volatile("" ::: "memory"); };

        PA opcur = op(0);
        __builtin_amdgcn_sched_barrier(0);
        PB Xs[KB0], H0s[2];
#pragma unroll
        for (int kb = 0; kb < KB0; ++kb) split8(xb[kb], Xs[kb]);
        split_frag(h0, H0s);

        // ---- tangent forward ---------------------------------------------------------------------------------------------
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        acc = chain_ahead(0, KB0, Xs, acc, opcur, KB0);                  // dW0~^T x~ + db0~
        f32x16 dh0;
        times_dtanh(acc, h0, dh0);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = tv[2 * DA * 16 + r];       // db1~
        acc = chain_ahead(KB0, 2, H0s, acc, opcur, KB0 + 2);             // dW1~^T h0
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            *reinterpret_cast<f16x8*>(spill + ((kb * 2 + 0) * WV + lane) * 16) = H0s[kb].hi;
            *reinterpret_cast<f16x8*>(spill + ((kb * 2 + 1) * WV + lane) * 16) = H0s[kb].lo;
        }
        stage();
        {
            PB D0s[2];
            split_frag(dh0, D0s);
            acc = chain_ahead(KB0 + 2, 2, D0s, acc, opcur, -1);          // W1~^T dh0~
        }
        stage();
        f32x16 dz1, gz1;
        float gmu[DA];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f32x2 hh = pair_of(h1, j);
            const f32x2 d = __builtin_elementwise_fma(-hh, hh, f32x2{1.0f, 1.0f});
            set_pair(dz1, j, d);
            set_pair(acc, j, pair_of(acc, j) * d);                        // dh1~
        }
        auto rows_of = [&](int which, int k, f32x4 (&out_)[4]) {
#pragma unroll
            for (int q = 0; q < 4; ++q) out_[q] = *reinterpret_cast<const f32x4*>(tv + (which * DA + k) * 16 + 4 * q);
        };
#pragma unroll
        for (int k = 0; k < DA; ++k) {
            f32x4 wq[4], dq[4];
            rows_of(0, k, wq);
            rows_of(1, k, dq);
            f32x2 pa[2], pb[2];
            pa[0] = pa[1] = pb[0] = pb[1] = f32x2{0.0f, 0.0f};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int q = j >> 1, e = 2 * (j & 1);
                pa[j & 1] = __builtin_elementwise_fma(pair_of(h1, j), f32x2{dq[q][e], dq[q][e + 1]}, pa[j & 1]);
                pb[j & 1] = __builtin_elementwise_fma(pair_of(acc, j), f32x2{wq[q][e], wq[q][e + 1]}, pb[j & 1]);
            }
            const f32x2 pd = (pa[0] + pb[0]) + (pa[1] + pb[1]);
            const float dmu = db2[k] + half_sum_swap(pd[0] + pd[1]);
            gmu[k] = wgt * (dmu * fkS[k]);
        }
        if (lh == 0) {
            wsum += wgt * a.inv_count;
#pragma unroll
            for (int k = 0; k < DA; ++k) gb2[k] += gmu[k];
        }
        stage();
        // ---- back-propagation, sample-major ---------------------------------------------------------------------------------
        {
            f32x2 g[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] = f32x2{0.0f, 0.0f};
#pragma unroll
            for (int k = 0; k < DA; ++k) {
                f32x4 wq[4];
                rows_of(0, k, wq);
                const f32x2 gk = {gmu[k], gmu[k]};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int q = j >> 1, e = 2 * (j & 1);
                    g[j] = __builtin_elementwise_fma(f32x2{wq[q][e], wq[q][e + 1]}, gk, g[j]);
                    const f32x2 w2 = __builtin_elementwise_fma(pair_of(h1, j), gk, f32x2{gW2l[2 * j][k], gW2l[2 * j + 1][k]});
                    gW2l[2 * j][k] = w2[0]; gW2l[2 * j + 1][k] = w2[1];
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x2 gz = g[j] * pair_of(dz1, j);
                set_pair(gz1, j, gz);
                const f32x2 b1n = f32x2{gb1l[2 * j], gb1l[2 * j + 1]} + gz;
                gb1l[2 * j] = b1n[0]; gb1l[2 * j + 1] = b1n[1];
            }
        }
        stage();
        opcur = op(KB0 + 4);
        __builtin_amdgcn_sched_barrier(0);
        PB G1s[2];
        split_frag(gz1, G1s);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        acc = chain_ahead(KB0 + 4, 2, G1s, acc, opcur, -1);              // W1~ gz1~
        stage();
        {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                H0s[kb].hi = *reinterpret_cast<const f16x8*>(spill + ((kb * 2 + 0) * WV + lane) * 16);
                H0s[kb].lo = *reinterpret_cast<const f16x8*>(spill + ((kb * 2 + 1) * WV + lane) * 16);
            }
            PB H0t[2], G1t[2];
            transpose_units(H0s, Id, H0t);
            transpose_units(G1s, Id, G1t);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {                              // gW1 += h0^T gz1~ (samples are K)
#if RL_SPLITH_TWO_ACC
                mm_bb(H0t[kb], G1t[kb], gW1, gW1c);
#else
                gW1 = mm_ab(as_image(H0t[kb]), G1t[kb], gW1);
#endif
            }
        }
        f32x16 gz0;
        times_dtanh(acc, h0, gz0, S.kg);
        {
            PB G0s[2], G0t[2], Xt[2];
            split_frag(gz0, G0s);
            transpose_units(G0s, Id, G0t);
            transpose_inputs<KB0>(Xs, Idx, Xt);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {                              // gW0 += x~_ext^T gz0~
#if RL_SPLITH_TWO_ACC_W0
                mm_bb(Xt[kb], G0t[kb], gW0, gW0c);
#else
                gW0 = mm_ab(as_image(Xt[kb]), G0t[kb], gW0);
#endif
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- fold the wavefronts of this workgroup in a fixed order, write ONE partial row; the scales cancel here -----------
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
    float* const myrow = reinterpret_cast<float*>(smem) + wave * P;
    constexpr float CROSS = 0.00048828125f;     // 2^-11
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int u = frag_unit(r, lh);
        myrow[N::W1 + u * H + lj] = __builtin_fmaf(gW1c[r], CROSS, gW1[r]) * S.u1;
        const float g0 = __builtin_fmaf(gW0c[r], CROSS, gW0[r]);
        if (u < DO) myrow[N::W0 + u * H + lj] = g0 * S.u0;
        else if (u == DO) myrow[N::B0 + lj] = g0 * S.u0b;
    }
    if (lj == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int u = frag_unit(r, lh);
            myrow[N::B1 + u] = b1s[r] * S.u1;
#pragma unroll
            for (int k = 0; k < DA; ++k) myrow[N::W2 + u * DA + k] = w2s[r][k] * S.u1;
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < DA; ++k) {
            myrow[N::B2 + k] = b2s[k] * S.u1;
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

// =====================================================================================================================
// (64, 64), one wavefront per SIMD: fvp_split64_kernel<DO, DA> on the f16 arithmetic.  The sample-axis products take their
// sample-side operand (h0, x~: magnitudes known) as an image -- the second set of accumulators would be 96 registers.
template <int DO, int DA>
__global__ void __launch_bounds__(4 * WV, 1) fvp_splith64_kernel(Args a) {
    constexpr int HT = 2, HH = 64;
    using N = Net<DO, DA, HH>;
    constexpr int P = N::P;
    constexpr int WAVES = 4;
    constexpr int KB0 = (DO + 1 + 15) / 16;
    constexpr int KBH = 2 * HT;
    constexpr int O_DW0 = 0, O_DW1 = O_DW0 + HT * KB0, O_W1T = O_DW1 + HT * KBH, O_W1 = O_W1T + HT * KBH,
                  N_OPS = O_W1 + HT * KBH;
    constexpr int OPS_BYTES = N_OPS * 3 * WV * 16;
    constexpr int TAILV = HT * 16 * DA * 2 + HT * 16;
    constexpr int LAND64 = 2 * HH * TS * 4;
    constexpr int GMU_BYTES = TS * 8 * 4;
    constexpr int WAVE_BYTES = LAND64 + GMU_BYTES;
    constexpr int LDS_TOTAL = WAVES * WAVE_BYTES + OPS_BYTES + 2 * TAILV * 4 + WAVES * 5 * 4;
    static_assert(DO + 1 <= 32 && DA <= 8, "two k-blocks of inputs + the bias slot; at most eight actions");
    static_assert(LDS_TOTAL >= WAVES * P * 4 && LDS_TOTAL <= 160 * 1024, "LDS budget; the fold rows alias everything");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x / WV, lane = threadIdx.x % WV;
    const int lj = lane & 31, lh = lane >> 5;
    char* const land = smem + wave * WAVE_BYTES;
    float* const gmub = reinterpret_cast<float*>(land + LAND64);
    char* const ops = smem + WAVES * WAVE_BYTES;
    float* const tailv = reinterpret_cast<float*>(ops + OPS_BYTES);
    float* const red = tailv + 2 * TAILV;

    const int B = a.B;
    const int n_tiles = B / TS;
    const int wave_global = blockIdx.x * WAVES + wave;
    const int waves_total = gridDim.x * WAVES;

    auto fetch = [&](int tile, float (&xq)[KB0][8], float& wq) {
        const int b = tile * TS + lj;
        wq = a.weight[b];
#pragma unroll
        for (int kb = 0; kb < KB0; ++kb)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int d = 16 * kb + 8 * lh + j;
                xq[kb][j] = a.obs[(size_t)(d < DO ? d : DO - 1) * B + b];
            }
    };
    auto fetch_acts = [&](int tile) {
        const float* src = a.acts + ((size_t)tile * (4 * 2 * HT) * WV + lane) * 4;
#pragma unroll
        for (int q = 0; q < 4 * 2 * HT; ++q)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + q * WV * 4), (lptr_t)(land + q * WV * 16), 16, 0, 0);
    };
    float xb[KB0][8], xb_next[KB0][8];
    float wgt = 0.0f, wgt_next = 0.0f;
    if (wave_global < n_tiles) {
        fetch(wave_global, xb_next, wgt_next);
        fetch_acts(wave_global);
    }
    asm volatile("" ::: "memory");

    const float* __restrict__ th = a.theta;
    const float* __restrict__ vc = a.vec;
    float db2[DA], fkS[DA], var_[DA];
    bool floored[DA];
    float fkmax = 0.0f;
#pragma unroll
    for (int k = 0; k < DA; ++k) {
        const float raw = th[N::LSTD + k];
        floored[k] = raw < a.log_min_std;
        const float ls = fmaxf(raw, a.log_min_std);
        var_[k] = __expf(2.0f * ls);
        fkS[k] = 2.0f / (2.0f * var_[k] + 1e-8f) * a.inv_count;
        fkmax = fmaxf(fkmax, fkS[k]);
    }
    float mx[5];
    block_maxima<N, WAVES * WV>(th, vc, red, mx);
    const Scales S = make_scales<6, ceil_log2(DA)>(*a.xmax, mx[0], mx[1], mx[2], mx[3], mx[4], fkmax);
#pragma unroll
    for (int k = 0; k < DA; ++k) {
        fkS[k] *= S.eg2;
        db2[k] = vc[N::B2 + k] * S.s1;
    }

    // ---- loop-invariant images in LDS: block (row tile t, k-block): A[i = 32 t + lj][k-slot (lh, j)] ---------------------
    for (int o = wave; o < N_OPS; o += WAVES) {
        float tv8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (o < O_DW1) {
                const int t = o / KB0, kb0 = o % KB0, i = 32 * t + lj, d = 16 * kb0 + 8 * lh + j;
                tv8[j] = d < DO ? vc[N::W0 + d * HH + i] * S.sv0 : (d == DO ? vc[N::B0 + i] * S.sv0b : 0.0f);
            } else {
                const int q = o - O_DW1, fam = q / (HT * KBH), r_ = q % (HT * KBH), t = r_ / KBH, kbg = r_ % KBH;
                const int i = 32 * t + lj, u = 32 * (kbg >> 1) + frag_unit(8 * (kbg & 1) + j, lh);
                tv8[j] = fam == 0 ? vc[N::W1 + u * HH + i] * S.s1
                       : fam == 1 ? th[N::W1 + u * HH + i] * S.sw1
                                  : th[N::W1 + i * HH + u] * S.sw1;
            }
        }
        const PA im = image8(tv8);
        *reinterpret_cast<f16x8*>(ops + ((o * 3 + 0) * WV + lane) * 16) = im.hi;
        *reinterpret_cast<f16x8*>(ops + ((o * 3 + 1) * WV + lane) * 16) = im.lo;
        *reinterpret_cast<f16x8*>(ops + ((o * 3 + 2) * WV + lane) * 16) = im.hs;
    }
    for (int e = threadIdx.x; e < 2 * HT * 16; e += WAVES * WV) {
        const int hh = e / (HT * 16), t = (e / 16) % HT, r = e % 16, u = 32 * t + frag_unit(r, hh);
        float* tv = tailv + hh * TAILV;
#pragma unroll
        for (int k = 0; k < DA; ++k) {
            tv[(k * HT + t) * 16 + r] = th[N::W2 + u * DA + k];
            tv[((DA + k) * HT + t) * 16 + r] = vc[N::W2 + u * DA + k] * S.s1;
        }
        tv[2 * DA * HT * 16 + t * 16 + r] = vc[N::B1 + u] * S.s1;
    }
    __syncthreads();
    auto op = [&](int o) -> PA {
        PA t;
        t.hi = *reinterpret_cast<const f16x8*>(ops + ((o * 3 + 0) * WV + lane) * 16);
        t.lo = *reinterpret_cast<const f16x8*>(ops + ((o * 3 + 1) * WV + lane) * 16);
        t.hs = *reinterpret_cast<const f16x8*>(ops + ((o * 3 + 2) * WV + lane) * 16);
        return t;
    };
    auto chains = [&](int base, const auto& Bk, f32x16 (&acc_)[HT]) {
        constexpr int NKB = sizeof(Bk) / sizeof(PB);
        PA cur = op(base);
#pragma unroll
        for (int i = 0; i < NKB * HT; ++i) {
            const int kb = i / HT, t = i % HT;
            PA nxtp = cur;
            if (i + 1 < NKB * HT) nxtp = op(base + ((i + 1) % HT) * NKB + (i + 1) / HT);
            __builtin_amdgcn_sched_barrier(0);
            acc_[t] = mm_ab(cur, Bk[kb], acc_[t]);
            cur = nxtp;
        }
    };
    f16x8 Id[2], Idx[KB0];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        Id[0][j] = (f16)(frag_unit(j, lh) == lj ? 1.0f : 0.0f);
        Id[1][j] = (f16)(frag_unit(8 + j, lh) == lj ? 1.0f : 0.0f);
#pragma unroll
        for (int kb = 0; kb < KB0; ++kb) Idx[kb][j] = (f16)(16 * kb + 8 * lh + j == lj ? XT_SCALE : 0.0f);
    }
    const int h1_off = (((HT + (lane >> 5)) * 4 + ((lane & 31) >> 3)) * WV + 32 * ((lane & 7) >> 2)) * 16 + (lane & 3) * 4;

    f32x16 gW1[HT][HT], gW0[HT];
    float gb1l[HT][16], gW2u[DA], gb2[DA], wsum = 0.0f;
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            gW0[t][r] = 0.0f; gb1l[t][r] = 0.0f;
#pragma unroll
            for (int t2 = 0; t2 < HT; ++t2) gW1[t][t2][r] = 0.0f;
        }
#pragma unroll
    for (int k = 0; k < DA; ++k) { gb2[k] = 0.0f; gW2u[k] = 0.0f; }

    for (int tile = wave_global; tile < n_tiles; tile += waves_total) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        f32x16 h0[HT], h1[HT];
#pragma unroll
        for (int t = 0; t < HT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(land + ((t * 4 + q) * WV + lane) * 16);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(land + (((HT + t) * 4 + q) * WV + lane) * 16);
#pragma unroll
                for (int e = 0; e < 4; ++e) { h0[t][4 * q + e] = v0[e]; h1[t][4 * q + e] = v1[e]; }
            }
#pragma unroll
        for (int kb = 0; kb < KB0; ++kb)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int d = 16 * kb + 8 * lh + j;
                xb[kb][j] = d < DO ? xb_next[kb][j] * S.sx : (d == DO ? S.sb : 0.0f);
            }
        wgt = wgt_next;
        const int nxt = tile + waves_total < n_tiles ? tile + waves_total : tile;
        fetch(nxt, xb_next, wgt_next);
        const float* tv = tailv + lh * TAILV;
        auto stage = [&]() { asm volatile("" ::: "memory"); };

        PB Xs[KB0], H0s[KBH];
#pragma unroll
        for (int kb = 0; kb < KB0; ++kb) split8(xb[kb], Xs[kb]);
#pragma unroll
        for (int t = 0; t < HT; ++t) {
            PB tmp[2];
            split_frag(h0[t], tmp);
            H0s[2 * t] = tmp[0]; H0s[2 * t + 1] = tmp[1];
        }

        f32x16 acc[HT], dh0[HT];
#pragma unroll
        for (int t = 0; t < HT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
        chains(O_DW0, Xs, acc);                                                            // dW0~^T x~ + db0~
#pragma unroll
        for (int t = 0; t < HT; ++t) times_dtanh(acc[t], h0[t], dh0[t]);
        stage();
        {
            f32x4 bq[HT][4];
#pragma unroll
            for (int t = 0; t < HT; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) bq[t][q] = *reinterpret_cast<const f32x4*>(tv + 2 * DA * HT * 16 + t * 16 + 4 * q);
#pragma unroll
            for (int t = 0; t < HT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = bq[t][r >> 2][r & 3];
        }
        chains(O_DW1, H0s, acc);                                                           // dW1~^T h0
        stage();
        {
            PB D0s[KBH];
#pragma unroll
            for (int t = 0; t < HT; ++t) {
                PB tmp[2];
                split_frag(dh0[t], tmp);
                D0s[2 * t] = tmp[0]; D0s[2 * t + 1] = tmp[1];
            }
            chains(O_W1T, D0s, acc);                                                       // W1~^T dh0~
        }
        stage();
        f32x16 dz1[HT], gz1[HT];
        float gmu[DA];
#pragma unroll
        for (int t = 0; t < HT; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x2 hh = pair_of(h1[t], j);
                const f32x2 d = __builtin_elementwise_fma(-hh, hh, f32x2{1.0f, 1.0f});
                set_pair(dz1[t], j, d);
                set_pair(acc[t], j, pair_of(acc[t], j) * d);                          // dh1~
            }
        auto rows_of = [&](int which, int k, f32x4 (&out_)[HT][4]) {
#pragma unroll
            for (int t = 0; t < HT; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    out_[t][q] = *reinterpret_cast<const f32x4*>(tv + ((which * DA + k) * HT + t) * 16 + 4 * q);
        };
#pragma unroll
        for (int k = 0; k < DA; ++k) {
            f32x4 wq[HT][4], dq[HT][4];
            rows_of(0, k, wq);
            rows_of(1, k, dq);
            f32x2 pa[HT], pb[HT];
#pragma unroll
            for (int t = 0; t < HT; ++t) pa[t] = pb[t] = f32x2{0.0f, 0.0f};
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int t = 0; t < HT; ++t) {
                    const int q = j >> 1, e = 2 * (j & 1);
                    pa[t] = __builtin_elementwise_fma(pair_of(h1[t], j), f32x2{dq[t][q][e], dq[t][q][e + 1]}, pa[t]);
                    pb[t] = __builtin_elementwise_fma(pair_of(acc[t], j), f32x2{wq[t][q][e], wq[t][q][e + 1]}, pb[t]);
                }
            const f32x2 pd = (pa[0] + pb[0]) + (pa[1] + pb[1]);
            const float dmu = db2[k] + half_sum_swap(pd[0] + pd[1]);
            gmu[k] = wgt * (dmu * fkS[k]);
        }
        if (lh == 0) {
            wsum += wgt * a.inv_count;
#pragma unroll
            for (int k = 0; k < DA; ++k) { gb2[k] += gmu[k]; gmub[lj * 8 + k] = gmu[k]; }
        }
        {
            f32x2 g[HT][8];
#pragma unroll
            for (int t = 0; t < HT; ++t)
#pragma unroll
                for (int j = 0; j < 8; ++j) g[t][j] = f32x2{0.0f, 0.0f};
#pragma unroll
            for (int k = 0; k < DA; ++k) {
                f32x4 wq[HT][4];
                rows_of(0, k, wq);
                const f32x2 gk = {gmu[k], gmu[k]};
#pragma unroll
                for (int t = 0; t < HT; ++t)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int q = j >> 1, e = 2 * (j & 1);
                        g[t][j] = __builtin_elementwise_fma(f32x2{wq[t][q][e], wq[t][q][e + 1]}, gk, g[t][j]);
                    }
            }
#pragma unroll
            for (int t = 0; t < HT; ++t)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const f32x2 gz = g[t][j] * pair_of(dz1[t], j);
                    set_pair(gz1[t], j, gz);
                    const f32x2 b1n = f32x2{gb1l[t][2 * j], gb1l[t][2 * j + 1]} + gz;
                    gb1l[t][2 * j] = b1n[0]; gb1l[t][2 * j + 1] = b1n[1];
                }
        }
        // ---- gW2 += h1^T gmu~ with the UNITS on the lanes: lane u walks the tile's samples ---------------------------------
        wave_sync();
        {
            const char* hp = land + h1_off;
#pragma unroll 8
            for (int s_ = 0; s_ < TS; ++s_) {
                const float hv = *reinterpret_cast<const float*>(hp + s_ * 16);
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(gmub + s_ * 8);
                const f32x4 g1 = *reinterpret_cast<const f32x4*>(gmub + s_ * 8 + 4);
#pragma unroll
                for (int k = 0; k < DA; ++k) gW2u[k] = __builtin_fmaf(hv, k < 4 ? g0[k] : g1[k - 4], gW2u[k]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the landing zone has been read: refill it
        fetch_acts(nxt);
        stage();
        PB G1s[KBH];
#pragma unroll
        for (int t = 0; t < HT; ++t) {
            PB tmp[2];
            split_frag(gz1[t], tmp);
            G1s[2 * t] = tmp[0]; G1s[2 * t + 1] = tmp[1];
        }
#pragma unroll
        for (int t = 0; t < HT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
        chains(O_W1, G1s, acc);                                                            // W1~ gz1~
        stage();
        {
            PB H0t[HT][2], G1t[HT][2];
#pragma unroll
            for (int t = 0; t < HT; ++t) {
                const PB hs[2] = {H0s[2 * t], H0s[2 * t + 1]}, gs[2] = {G1s[2 * t], G1s[2 * t + 1]};
                transpose_units(hs, Id, H0t[t]);
                transpose_units(gs, Id, G1t[t]);
            }
#pragma unroll
            for (int ti = 0; ti < HT; ++ti)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    const PA him = as_image(H0t[ti][kb]);
#pragma unroll
                    for (int tj2 = 0; tj2 < HT; ++tj2) gW1[ti][tj2] = mm_ab(him, G1t[tj2][kb], gW1[ti][tj2]);   // gW1 += h0^T gz1~
                }
        }
        stage();
        {
            PB G0t[HT][2], Xt[2];
#pragma unroll
            for (int t = 0; t < HT; ++t) {
                f32x16 gz0;
                times_dtanh(acc[t], h0[t], gz0, S.kg);
                PB gs[2];
                split_frag(gz0, gs);
                transpose_units(gs, Id, G0t[t]);
            }
            transpose_inputs<KB0>(Xs, Idx, Xt);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const PA xim = as_image(Xt[kb]);
#pragma unroll
                for (int t = 0; t < HT; ++t) gW0[t] = mm_ab(xim, G0t[t][kb], gW0[t]);                           // gW0 += x~_ext^T gz0~
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    float b1s[HT][16];
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = gb1l[t][r];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, WV);
            b1s[t][r] = v;
        }
    float b2s[DA];
#pragma unroll
    for (int k = 0; k < DA; ++k) b2s[k] = wave_sum(gb2[k]);
    const float ws = wave_sum(wsum);
    __syncthreads();
    float* const myrow = reinterpret_cast<float*>(smem) + wave * P;
#pragma unroll
    for (int ti = 0; ti < HT; ++ti)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int u = 32 * ti + frag_unit(r, lh);
#pragma unroll
            for (int tj2 = 0; tj2 < HT; ++tj2) myrow[N::W1 + u * HH + 32 * tj2 + lj] = gW1[ti][tj2][r] * S.u1;
        }
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = frag_unit(r, lh);
            if (d < DO) myrow[N::W0 + d * HH + 32 * t + lj] = gW0[t][r] * S.u0;
            else if (d == DO) myrow[N::B0 + 32 * t + lj] = gW0[t][r] * S.u0b;
        }
    if (lj == 0) {
#pragma unroll
        for (int t = 0; t < HT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) myrow[N::B1 + 32 * t + frag_unit(r, lh)] = b1s[t][r] * S.u1;
    }
#pragma unroll
    for (int k = 0; k < DA; ++k) myrow[N::W2 + lane * DA + k] = gW2u[k] * S.u1;        // lane = unit
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < DA; ++k) {
            myrow[N::B2 + k] = b2s[k] * S.u1;
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

static Args make_args(const rl_policy_batch* g, const float* vec) {
    Args a;
    a.B = g->n_samples; a.theta = g->theta; a.vec = vec; a.acts = g->activations; a.obs = g->obs; a.weight = g->weights;
    a.xmax = g->obs_absmax; a.inv_count = g->inv_count; a.log_min_std = g->log_min_std; a.partial = nullptr;
    return a;
}

template <int DO, int DA>
static int launch32(const rl_policy_batch* g, const float* vec, void* ws, size_t ws_bytes, double* out, hipStream_t st) {
    using N = Net<DO, DA, H>;
    constexpr int WPS = DA <= 2 ? 2 : 1;
    constexpr int WAVES = 4 * WPS;
    constexpr int KB0 = (DO + 1 + 15) / 16;
    constexpr int LDS_BYTES = WAVES * (LAND_BYTES + 2 * 2 * WV * 16 + ((DO + 1 + 7) / 8) * WV * 16) + ops_bytes(KB0) +
                              2 * (16 * DA * 2 + 16) * 4 + WAVES * 5 * 4;
    Args a = make_args(g, vec);
    const int n_tiles = a.B / TS;
    int grid = (n_tiles + WAVES - 1) / WAVES;
    if (grid > 256) grid = 256;                   // one workgroup per CU
    const size_t need = (size_t)grid * N::P * sizeof(float);
    if (ws_bytes < need) return set_error(RL_ERR_ARG, "policy pass workspace too small: %zu < %zu bytes", ws_bytes, need);
    a.partial = (float*)ws;
    auto kern = fvp_splith_kernel<DO, DA, WPS>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           LDS_BYTES);
        if (e != hipSuccess) return set_error(RL_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * WV), LDS_BYTES, st, a);
    int rc = check_launch("fvp_splith_kernel");
    if (rc) return rc;
    return launch_reduce_rows(a.partial, grid, N::P, out, st);
}

template <int DO, int DA>
static int launch64(const rl_policy_batch* g, const float* vec, void* ws, size_t ws_bytes, double* out, hipStream_t st) {
    using N = Net<DO, DA, 64>;
    constexpr int HT = 2, WAVES = 4;
    constexpr int KB0 = (DO + 1 + 15) / 16;
    constexpr int N_OPS = HT * KB0 + 3 * HT * 2 * HT;
    constexpr int LDS_BYTES = WAVES * (2 * 64 * TS * 4 + TS * 8 * 4) + N_OPS * 3 * WV * 16 + 2 * (HT * 16 * DA * 2 + HT * 16) * 4 +
                              WAVES * 5 * 4;
    Args a = make_args(g, vec);
    const int n_tiles = a.B / TS;
    int grid = (n_tiles + WAVES - 1) / WAVES;
    if (grid > 256) grid = 256;                   // one workgroup per CU, one wavefront per SIMD
    const size_t need = (size_t)grid * N::P * sizeof(float);
    if (ws_bytes < need) return set_error(RL_ERR_ARG, "policy pass workspace too small: %zu < %zu bytes", ws_bytes, need);
    a.partial = (float*)ws;
    auto kern = fvp_splith64_kernel<DO, DA>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           LDS_BYTES);
        if (e != hipSuccess) return set_error(RL_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * WV), LDS_BYTES, st, a);
    int rc = check_launch("fvp_splith64_kernel");
    if (rc) return rc;
    return launch_reduce_rows(a.partial, grid, N::P, out, st);
}

}  // namespace splith

// the (obs_dim, act_dim) pairs of fvp_split_kernel and of fvp_split64_kernel
#define SPLITH_SHAPES(X) X(4, 1) X(6, 1) X(11, 1) X(13, 2) X(13, 1) X(20, 3) X(20, 6) X(21, 6)
#define SPLITH64_SHAPES(X) X(4, 1) X(6, 1) X(11, 1) X(13, 2) X(20, 3) X(20, 6) X(21, 6)
bool split_fvp_takes(const rl_policy_batch* g);                          // policy_split_kernels.hip
// The f16 form takes what the bf16 split kernels take, for the shapes above, when the caller provides max |obs|
// (rl_policy_batch.obs_absmax) and does not ask for another kernel (rl_launch_opts.fvp_split: 0 = the library's choice,
// 4 = this one).
bool splith_fvp_takes(const rl_policy_batch* g) {
    if (!split_fvp_takes(g) || !g->obs_absmax) return false;
    const int req = g->opts ? g->opts->fvp_split : 0;
    if (req != 0 && req != 4) return false;
    if (g->opts && g->opts->fvp_split_wps == 1) return false;
#define SPLITCASE(DO, DA) if (g->obs_dim == DO && g->act_dim == DA) return true;
    if (g->hidden0 == 64) { SPLITH64_SHAPES(SPLITCASE) }
    else { SPLITH_SHAPES(SPLITCASE) }
#undef SPLITCASE
    return false;
}
int splith_fvp_dispatch(const rl_policy_batch* g, const float* vec, void* ws, size_t ws_bytes, double* out, hipStream_t st) {
    if (!splith_fvp_takes(g)) return RL_SPLIT_NOT_TAKEN;
    if (g->hidden0 == 64) {
#define SPLITCASE(DO, DA) if (g->obs_dim == DO && g->act_dim == DA) return splith::launch64<DO, DA>(g, vec, ws, ws_bytes, out, st);
        SPLITH64_SHAPES(SPLITCASE)
#undef SPLITCASE
        return RL_SPLIT_NOT_TAKEN;
    }
#define SPLITCASE(DO, DA) if (g->obs_dim == DO && g->act_dim == DA) return splith::launch32<DO, DA>(g, vec, ws, ws_bytes, out, st);
    SPLITH_SHAPES(SPLITCASE)
#undef SPLITCASE
    return RL_SPLIT_NOT_TAKEN;
}

}  // namespace rl
