// Does a bf16 MFMA stream overlap with FP32 vector work on the same SIMD -- from another wavefront, and inside one
// wavefront?  (tools/ubench/mfma_valu_overlap.hip asked the same of the f32-input MFMA: it does not overlap at all.)
// Workgroup = 8 wavefronts = 2 per SIMD (w and w + 4 share a SIMD).  Per loop iteration a stream is worth 512 cycles:
// 16 x v_mfma_f32_32x32x16_bf16 (32 cycles each) or 128 x v_fma_f32 (4 cycles each) or the three-way split of 28 values
// (16 pairs x (9 + 2) instructions, four independent chains).
//   mode 0: MFMA + MFMA   1: FMA + FMA   2: MFMA + FMA   3: MFMA alone   4: FMA alone
//   mode 5: one wavefront alternating 1 MFMA / 8 FMA    6: MFMA + split   7: split alone   8: one wavefront alternating MFMA / split
//   mode 9 / 10: MFMA + 128 v_cvt_pk_bf16_f32 / those alone    11 / 12: MFMA + 128 and / shift / subtract / those alone
//   mode 13 / 14: MFMA + the split of 16 pairs written stage by stage over eight pairs (eight independent chains) / alone
// hipcc --offload-arch=gfx950 -O3 -o tools/ubench/mfma_bf16_valu_overlap tools/ubench/mfma_bf16_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __forceinline__ void split_pair(f32x2 a, bf16x2& h, bf16x2& m, bf16x2& q) {
    h = __builtin_convertvector(a, bf16x2);
#ifdef SCALAR_SUB       // the residuals by two v_sub_f32 instead of one v_pk_add_f32
    const f32x2 hf = __builtin_convertvector(h, f32x2);
    float r0 = a[0] - hf[0], r1 = a[1] - hf[1];
    asm volatile("" : "+v"(r0), "+v"(r1));
    const f32x2 r = {r0, r1};
    m = __builtin_convertvector(r, bf16x2);
    const f32x2 mf = __builtin_convertvector(m, f32x2);
    float l0 = r0 - mf[0], l1 = r1 - mf[1];
    asm volatile("" : "+v"(l0), "+v"(l1));
    const f32x2 l = {l0, l1};
#else
    const f32x2 r = a - __builtin_convertvector(h, f32x2);
    m = __builtin_convertvector(r, bf16x2);
    const f32x2 l = r - __builtin_convertvector(m, f32x2);
#endif
    q = __builtin_convertvector(l, bf16x2);
}

__global__ void __launch_bounds__(512) k(float* out, int iters, int mode) {
    const int wave = threadIdx.x >> 6;
    const int hi = wave >> 2;   // 0: first wave of its SIMD, 1: second
    int role;                   // 0 mfma, 1 fma, 2 idle, 3 alternate mfma / fma, 4 split, 5 alternate mfma / split
    switch (mode) {
        case 0: role = 0; break;
        case 1: role = 1; break;
        case 2: role = hi; break;
        case 3: role = hi ? 2 : 0; break;
        case 4: role = hi ? 2 : 1; break;
        case 5: role = hi ? 2 : 3; break;
        case 6: role = hi ? 4 : 0; break;
        case 7: role = hi ? 2 : 4; break;
        case 8: role = hi ? 2 : 5; break;
        case 9: role = hi ? 6 : 0; break;      // MFMA + conversions only
        case 10: role = hi ? 2 : 6; break;
        case 11: role = hi ? 7 : 0; break;     // MFMA + and / shift / subtract only
        case 12: role = hi ? 2 : 7; break;
        case 13: role = hi ? 8 : 0; break;     // MFMA + the split of eight pairs, stage by stage (eight independent chains)
        default: role = hi ? 2 : 8; break;
    }
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    bf16x8 A, Bv;
    for (int j = 0; j < 8; ++j) { A[j] = (__bf16)(a + j); Bv[j] = (__bf16)(b - j); }
    f32x16 acc0 = {}, acc1 = {};
    float v0 = a, v1 = b, v2 = a + b, v3 = a - b, v4 = a * b, v5 = 1.f, v6 = 2.f, v7 = 3.f;
    f32x2 s0 = {a, b}, s1 = {b, a}, s2 = {a + 1.f, b + 1.f}, s3 = {a - 1.f, b - 2.f};
    unsigned sink = 0, sink1 = 0, sink2 = 0, sink3 = 0;
    auto fma8 = [&]() {
        v0 = __builtin_fmaf(v0, b, a); v1 = __builtin_fmaf(v1, b, a); v2 = __builtin_fmaf(v2, b, a);
        v3 = __builtin_fmaf(v3, b, a); v4 = __builtin_fmaf(v4, b, a); v5 = __builtin_fmaf(v5, b, a);
        v6 = __builtin_fmaf(v6, b, a); v7 = __builtin_fmaf(v7, b, a);
    };
    // one pair: 9 vector instructions + 2 to keep its inputs changing; four independent pairs take turns, as the
    // fragments of the kernel offer (8 independent pairs each)
    auto split_of = [&](f32x2& s, unsigned& snk) {
        bf16x2 h, m, q;
        split_pair(s, h, m, q);
        snk += __builtin_bit_cast(unsigned, h) ^ __builtin_bit_cast(unsigned, m) ^ __builtin_bit_cast(unsigned, q);
#ifdef SCALAR_SUB
        s[0] = __builtin_fmaf(s[0], b, a); s[1] = __builtin_fmaf(s[1], b, a);
#else
        s = s * s1 + s1;       // (one more packed instruction)
#endif
    };
    int turn = 0;
    auto split1 = [&]() {
        switch (turn++ & 3) {
            case 0: split_of(s0, sink); break;
            case 1: split_of(s2, sink1); break;
            case 2: split_of(s3, sink2); break;
            default: split_of(s1, sink3); break;
        }
    };
    if (role == 0) {
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, Bv, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Bv, A, acc1, 0, 0, 0);
            }
    } else if (role == 1) {
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int u = 0; u < 16; ++u) fma8();
    } else if (role == 3) {
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, Bv, acc0, 0, 0, 0);
                fma8();
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Bv, A, acc1, 0, 0, 0);
                fma8();
            }
    } else if (role == 4) {
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int u = 0; u < 16; ++u) split1();
    } else if (role == 5) {
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, Bv, acc0, 0, 0, 0);
                split1();
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Bv, A, acc1, 0, 0, 0);
                split1();
            }
    }
    if (role == 6) {          // 128 v_cvt_pk_bf16_f32, eight independent chains
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                float* v[8] = {&v0, &v1, &v2, &v3, &v4, &v5, &v6, &v7};
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    unsigned r;
                    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(*v[c]), "v"(b));
                    *v[c] = __uint_as_float(r);
                }
            }
    } else if (role == 7) {   // 128 of v_and_b32 / v_lshlrev_b32 / v_sub_f32 in turn, eight independent chains
        unsigned w0 = __float_as_uint(v0), w1 = __float_as_uint(v1), w2 = __float_as_uint(v2), w3 = __float_as_uint(v3);
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                asm volatile("v_and_b32 %0, 0xffff0f0f, %0\n v_and_b32 %1, 0xffff0f0f, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3\n"
                             "v_sub_f32 %4, %4, %5\n v_sub_f32 %6, %6, %5\n v_sub_f32 %7, %7, %5\n v_sub_f32 %8, %8, %5"
                             : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(v4), "+v"(b), "+v"(v5), "+v"(v6), "+v"(v7));
            }
        v0 = __uint_as_float(w0 ^ w1 ^ w2 ^ w3);
    }
    if (role == 8) {
        f32x2 p[8];
        for (int c = 0; c < 8; ++c) p[c] = f32x2{a + c, b - c};
        unsigned snk = 0;
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                bf16x2 h[8], m[8], q[8];
                f32x2 r[8], l[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) h[c] = __builtin_convertvector(p[c], bf16x2);
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const f32x2 hf = __builtin_convertvector(h[c], f32x2);
                    r[c] = f32x2{p[c][0] - hf[0], p[c][1] - hf[1]};
                }
#pragma unroll
                for (int c = 0; c < 8; ++c) m[c] = __builtin_convertvector(r[c], bf16x2);
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const f32x2 mf = __builtin_convertvector(m[c], f32x2);
                    l[c] = f32x2{r[c][0] - mf[0], r[c][1] - mf[1]};
                }
#pragma unroll
                for (int c = 0; c < 8; ++c) q[c] = __builtin_convertvector(l[c], bf16x2);
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    snk += __builtin_bit_cast(unsigned, h[c]) ^ __builtin_bit_cast(unsigned, m[c]) ^ __builtin_bit_cast(unsigned, q[c]);
                    p[c][0] = __builtin_fmaf(p[c][0], b, a); p[c][1] = __builtin_fmaf(p[c][1], b, a);
                }
            }
        v0 = p[0][0] + p[7][1] + (float)snk;
    }
    float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + s0[0] + s0[1] + s1[0] + s2[0] + s3[1] + (float)(sink + sink1 + sink2 + sink3);
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    if (s == 123.456f) out[threadIdx.x] = s;
}

int main() {
    float* d;
    hipMalloc(&d, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    const char* names[] = {"bf16 MFMA + bf16 MFMA", "FMA + FMA", "bf16 MFMA + FMA", "bf16 MFMA alone", "FMA alone",
                           "one wave: 1 MFMA / 8 FMA", "bf16 MFMA + split", "split alone", "one wave: 1 MFMA / 1 split pair", "bf16 MFMA + cvt_pk_bf16 only", "cvt_pk_bf16 alone",
                           "bf16 MFMA + and/shift/sub only", "and/shift/sub alone",
                           "bf16 MFMA + staged split (8 chains)", "staged split alone"};
    for (int mode = 0; mode < 15; ++mode) {
        k<<<256, 512>>>(d, 10, mode);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<<<256, 512>>>(d, iters, mode);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("mode %d %-42s %8.3f ms = %6.0f ns per iteration (512 cycles of one stream = %.0f ns at 2.4 GHz)\n", mode,
               names[mode], ms, ms * 1e6 / iters, 512 / 2.4);
    }
    return 0;
}
