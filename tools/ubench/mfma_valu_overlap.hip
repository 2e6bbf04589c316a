// Does an f32-input MFMA stream of one wavefront overlap with the FP32 VALU stream of another wavefront on the
// same SIMD?  Workgroup = 8 wavefronts = 2 per SIMD (w and w + 4 share a SIMD).  role(w) selects the stream.
//   mode 0: both waves MFMA      mode 1: both VALU      mode 2: one MFMA + one VALU      mode 3: MFMA wave + idle
//   mode 4: VALU wave + idle     mode 5: one wave alternating MFMA / VALU blocks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int KIND>
__global__ void __launch_bounds__(512) k(float* out, int iters, int mode) {
    const int wave = threadIdx.x >> 6;
    const int hi = wave >> 2;   // 0: first wave of its SIMD, 1: second
    int role;                   // 0 mfma, 1 valu, 2 idle, 3 alternate
    if (mode == 0) role = 0;
    else if (mode == 1) role = 1;
    else if (mode == 2) role = hi;
    else if (mode == 3) role = hi ? 2 : 0;
    else if (mode == 4) role = hi ? 2 : 1;
    else role = hi ? 2 : 3;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    f32x16 acc0 = {}, acc1 = {};
    float v0 = a, v1 = b, v2 = a + b, v3 = a - b, v4 = a * b, v5 = 1.f, v6 = 2.f, v7 = 3.f;
    if (role == 0) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
            }
        }
    } else if (role == 1) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 32; ++u) {   // 256 independent-ish FMAs = 1024 cycles, same as 16 MFMAs
                v0 = __builtin_fmaf(v0, b, a); v1 = __builtin_fmaf(v1, b, a); v2 = __builtin_fmaf(v2, b, a);
                v3 = __builtin_fmaf(v3, b, a); v4 = __builtin_fmaf(v4, b, a); v5 = __builtin_fmaf(v5, b, a);
                v6 = __builtin_fmaf(v6, b, a); v7 = __builtin_fmaf(v7, b, a);
            }
        }
    } else if (role == 3) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    v0 = __builtin_fmaf(v0, b, a); v1 = __builtin_fmaf(v1, b, a); v2 = __builtin_fmaf(v2, b, a);
                    v3 = __builtin_fmaf(v3, b, a); v4 = __builtin_fmaf(v4, b, a); v5 = __builtin_fmaf(v5, b, a);
                    v6 = __builtin_fmaf(v6, b, a); v7 = __builtin_fmaf(v7, b, a);
                }
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    v0 = __builtin_fmaf(v0, b, a); v1 = __builtin_fmaf(v1, b, a); v2 = __builtin_fmaf(v2, b, a);
                    v3 = __builtin_fmaf(v3, b, a); v4 = __builtin_fmaf(v4, b, a); v5 = __builtin_fmaf(v5, b, a);
                    v6 = __builtin_fmaf(v6, b, a); v7 = __builtin_fmaf(v7, b, a);
                }
            }
        }
    }
    float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    if (s == 123.456f) out[threadIdx.x] = s;
}

int main() {
    float* d;
    hipMalloc(&d, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    const char* names[] = {"MFMA + MFMA", "VALU + VALU", "MFMA + VALU", "MFMA alone", "VALU alone", "one wave alternating"};
    for (int mode = 0; mode < 6; ++mode) {
        k<0><<<256, 512>>>(d, 10, mode);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<0><<<256, 512>>>(d, iters, mode);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        // per iteration per wave: 16 MFMAs (1024 matrix cycles) or 256 FMAs (1024 VALU cycles)
        printf("mode %d %-22s %8.3f ms  = %6.0f ns per iteration (1024 cycles of one stream = %.0f ns at 2.4 GHz)\n", mode,
               names[mode], ms, ms * 1e6 / iters, 1024 / 2.4);
    }
    return 0;
}
