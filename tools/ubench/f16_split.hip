// f16_split.hip -- the facts a two-way f16 split product would rest on, checked on the device:
//   1. v_mfma_f32_32x32x16_f16 honours f16 SUBNORMAL inputs (a part below 2^-14 is not flushed);
//   2. x = hi + lo with hi = f16(x), lo = f16(x - hi): three cross terms hi hi + hi lo + lo hi (one f32 accumulator)
//      against float64, next to an f32 fma chain, the three-way bf16 split with six cross terms
//      (policy_split_kernels.hip), the scaled-lo form with two accumulators (lo' = f16(2^11 (x - hi)), c = c0 + 2^-11 c1)
//      and the four-term form (+ lo lo), over magnitude regimes;
//   3. the issue rate of the f16 instruction equals the bf16 one's.
// hipcc --offload-arch=gfx950 -O2 -o /tmp/f16_split tools/ubench/f16_split.hip && /tmp/f16_split
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__host__ __device__ constexpr int frag_unit(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// D[32][32] = A[32][K] B[K][32], K = 16 KB; mode: 1 bf16x3 six terms, 2 f16x2 three terms, 3 f16x2 scaled lo two
// accumulators, 4 f16x2 four terms, 5 f16 hi only (to see the size of what the lo parts carry)
template <int KB>
__global__ void mm(const float* A, const float* B, float* D, int mode) {
    const int l = threadIdx.x, n = l & 31, h = l >> 5;
    f32x16 c = {}, c1 = {};
    for (int kb = 0; kb < KB; ++kb) {
        float a[8], b[8];
        for (int j = 0; j < 8; ++j) {
            a[j] = A[n * 16 * KB + 16 * kb + 8 * h + j];
            b[j] = B[(16 * kb + 8 * h + j) * 32 + n];
        }
        if (mode == 1) {
            bf16x8 ap[3], bp[3];
            for (int j = 0; j < 8; ++j) {
                float r = a[j];
                for (int p = 0; p < 3; ++p) { ap[p][j] = (__bf16)r; r -= (float)ap[p][j]; }
                r = b[j];
                for (int p = 0; p < 3; ++p) { bp[p][j] = (__bf16)r; r -= (float)bp[p][j]; }
            }
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[1], bp[1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], bp[2], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[2], bp[0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], bp[1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[1], bp[0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], bp[0], c, 0, 0, 0);
        } else {
            f16x8 ah, al, bh, bl;
            const float sc = mode == 3 ? 2048.0f : 1.0f;
            for (int j = 0; j < 8; ++j) {
                ah[j] = (_Float16)a[j]; al[j] = (_Float16)((a[j] - (float)ah[j]) * sc);
                bh[j] = (_Float16)b[j]; bl[j] = (_Float16)((b[j] - (float)bh[j]) * sc);
            }
            if (mode == 3) {
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c1, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c1, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
            } else {
                if (mode == 4) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bl, c, 0, 0, 0);
                if (mode != 5) {
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0);
                }
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
            }
        }
    }
    for (int r = 0; r < 16; ++r) D[frag_unit(r, h) * 32 + n] = mode == 3 ? c[r] + c1[r] * (1.0f / 2048.0f) : c[r];
}

// 4. the five-instruction split of policy_splith_kernels.hip (v_fma_mix_f32 residuals, v_fma_mixlo/hi_f16 scaled parts) against
//    the plain C form, bit for bit
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
__global__ void split_forms(const float* a, unsigned* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    const float a0 = a[2 * i], a1 = a[2 * i + 1];
    unsigned hb, lb;
    float r0, r1;
    const float k = 2048.0f;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hb) : "v"(a0), "v"(a1));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hb), "v"(a0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hb), "v"(a1));
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(lb) : "v"(r0), "s"(k));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(lb) : "v"(r1), "s"(k));
    f16x2 h, l;
    h[0] = (_Float16)a0; h[1] = (_Float16)a1;
    l[0] = (_Float16)((a0 - (float)h[0]) * 2048.0f); l[1] = (_Float16)((a1 - (float)h[1]) * 2048.0f);
    out[4 * i] = hb; out[4 * i + 1] = lb;
    out[4 * i + 2] = __builtin_bit_cast(unsigned, h); out[4 * i + 3] = __builtin_bit_cast(unsigned, l);
}

template <bool F16>
__global__ void rate(float* out, int iters) {
    f16x8 a, b; bf16x8 a2, b2;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 0.001f); b[j] = (_Float16)1.0f; a2[j] = (__bf16)(threadIdx.x * 0.001f); b2[j] = (__bf16)1.0f; }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int i = 0; i < iters; ++i) {
        if (F16) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
        } else {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, c3, 0, 0, 0);
        }
    }
    float s = 0.0f;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    constexpr int KB = 2, K = 16 * KB;
    float *dA, *dB, *dD;
    hipMalloc(&dA, 32 * K * 4); hipMalloc(&dB, K * 32 * 4); hipMalloc(&dD, 32 * 32 * 4);
    std::vector<float> A(32 * K), B(K * 32), D(32 * 32);
    // 1. subnormal inputs
    for (auto& v : A) v = ldexpf(1.0f, -20);          // f16 subnormal
    for (auto& v : B) v = 1024.0f;
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    mm<KB><<<1, 64>>>(dA, dB, dD, 5);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    printf("subnormal A (2^-20) x 2^10, K = %d: D[0] = %g (honoured: %g, flushed: 0)\n", K, D[0], K * ldexp(1.0, -10));
    // lo of a small value: x = 0.01 (1 + 2^-12): lo = 0.01 * 2^-12 = 2.4e-6, subnormal
    for (auto& v : A) v = 0.01f * (1.0f + ldexpf(1.0f, -12));
    for (auto& v : B) v = 1.0f;
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    for (int mode : {5, 2}) {
        mm<KB><<<1, 64>>>(dA, dB, dD, mode);
        hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
        printf("  sum of %d x 0.01 (1 + 2^-12), mode %d: rel err %.3e\n", K, mode, D[0] / (K * (double)A[0]) - 1.0);
    }
    // 2. accuracy
    std::mt19937 rng(1);
    std::normal_distribution<float> nd;
    const float scales[][2] = {{1, 1}, {0.1f, 1}, {1e-2f, 1}, {1e-3f, 10}, {1e-4f, 1}, {100, 30}, {0.3f, 0.3f}, {1e-2f, 1e-2f}, {1e-5f, 1}, {3000, 10}};
    printf("%-16s %-10s | componentwise max |d - d64| / sum|a||b|, and normwise ||d - d64|| / ||d64||\n", "scales", "");
    printf("%-16s   f32 chain          bf16x3/6           f16x2/3            f16x2/3 scaled     f16x2/4            f16 hi only\n", "");
    for (auto& sc : scales) {
        double cw[6] = {0}, nw_num[6] = {0}, nw_den = 0;
        for (int trial = 0; trial < 20; ++trial) {
            for (auto& v : A) v = nd(rng) * sc[0];
            for (auto& v : B) v = nd(rng) * sc[1];
            hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
            std::vector<double> ref(32 * 32), mag(32 * 32);
            std::vector<float> chain(32 * 32);
            for (int i = 0; i < 32; ++i)
                for (int n = 0; n < 32; ++n) {
                    double s = 0, m = 0; float f = 0;
                    for (int k = 0; k < K; ++k) {
                        s += (double)A[i * K + k] * B[k * 32 + n]; m += fabs((double)A[i * K + k] * B[k * 32 + n]);
                        f = fmaf(A[i * K + k], B[k * 32 + n], f);
                    }
                    ref[i * 32 + n] = s; mag[i * 32 + n] = m; chain[i * 32 + n] = f; nw_den += s * s;
                }
            for (int v = 0; v < 6; ++v) {
                if (v > 0) { mm<KB><<<1, 64>>>(dA, dB, dD, v); hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost); }
                for (int e = 0; e < 32 * 32; ++e) {
                    const double d = (v == 0 ? chain[e] : D[e]) - ref[e];
                    cw[v] = fmax(cw[v], fabs(d) / mag[e]); nw_num[v] += d * d;
                }
            }
        }
        printf("%-7g %-8g", sc[0], sc[1]);
        for (int v = 0; v < 6; ++v) printf("  %.2e %.2e", cw[v], sqrt(nw_num[v] / nw_den));
        printf("\n");
    }
    // 4. split forms
    {
        const int n = 1 << 20;
        std::vector<float> X(n);
        std::uniform_real_distribution<float> ud(-1.0f, 1.0f);
        for (int i = 0; i < n; ++i) X[i] = ud(rng) * ldexpf(1.0f, (int)(rng() % 46) - 30);    // magnitudes 2^-30 .. 2^15
        X[0] = 0.0f; X[1] = -0.0f; X[2] = 65504.0f; X[3] = -32768.0f; X[4] = 6.1e-5f; X[5] = 5.9e-8f; X[6] = 1e-9f; X[7] = -3.1e-5f;
        float* dX; unsigned* dO;
        hipMalloc(&dX, n * 4); hipMalloc(&dO, 2 * n * 4);
        hipMemcpy(dX, X.data(), n * 4, hipMemcpyHostToDevice);
        split_forms<<<n / 2 / 256, 256>>>(dX, dO, n);
        std::vector<unsigned> O(2 * n);
        hipMemcpy(O.data(), dO, 2 * n * 4, hipMemcpyDeviceToHost);
        long bad = 0;
        for (int i = 0; i < n / 2; ++i) bad += (O[4 * i] != O[4 * i + 2]) + (O[4 * i + 1] != O[4 * i + 3]);
        printf("five-instruction split against the C form over %d values (2^-30 .. 2^15, zeros, subnormals): %ld differing words\n", n, bad);
    }
    // 3. rate
    float* dout; hipMalloc(&dout, 1024 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int f16 = 0; f16 < 2; ++f16) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (f16) rate<true><<<1024, 256>>>(dout, 20000); else rate<false><<<1024, 256>>>(dout, 20000);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = 1024.0 * 4 * 20000 * 4 * 2.0 * 32 * 32 * 16;
            if (rep) printf("%s: %.3f ms, %.1f TFLOP/s\n", f16 ? "v_mfma_f32_32x32x16_f16 " : "v_mfma_f32_32x32x16_bf16", ms, flops / ms * 1e-9);
        }
    }
    return 0;
}
