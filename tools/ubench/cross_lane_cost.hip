// Issue cost of the cross-lane forms available on gfx950, one wavefront per SIMD, 8 independent chains:
//   quad_perm DPP fused into an add, bare v_mov_b32_dpp, row_shr DPP, ds_swizzle, v_permlane32_swap, 4x4x1 MFMA gather,
//   and a DPP op diluted by K plain FMAs (does the DPP overlap with the plain stream?).
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ float dppq(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true)); }
__device__ __forceinline__ float dpprow(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true)); }
template <int FORM, int K>
__global__ void __launch_bounds__(64) k(float* out, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = a + c;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 32; ++u) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (FORM == 0) v[c] = v[c] + dppq(v[c]);
                else if (FORM == 1) v[c] = __builtin_fmaf(dppq(v[c]), dppq(v[(c + 1) & 7]), a);   // two bare movs + fma
                else if (FORM == 2) v[c] = v[c] + dpprow(v[c]);
                else if (FORM == 3) v[c] = v[c] + __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v[c]), 0x80B1));
                else if (FORM == 4) {
                    float x = v[c], y = v[c];
                    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
                    v[c] = x + y;
                } else if (FORM == 5) {
                    f32x4 z = {0, 0, 0, 0};
                    f32x4 g = __builtin_amdgcn_mfma_f32_4x4x1f32(v[c], 1.0f, z, 0, 0, 0);
                    v[c] = (g[0] + g[1]) + (g[2] + g[3]);
                } else if (FORM == 6) {
                    v[c] = v[c] + dppq(v[c]);
#pragma unroll
                    for (int q = 0; q < K; ++q) v[c] = __builtin_fmaf(v[c], b, a);
                } else {
#pragma unroll
                    for (int q = 0; q < K + 1; ++q) v[c] = __builtin_fmaf(v[c], b, a);
                }
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) s += v[c];
    if (s == 123.456f) out[threadIdx.x] = s;
}
template <int FORM, int K>
void run(float* d, const char* name, int ops_per_slot) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    k<FORM, K><<<1024, 64>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<FORM, K><<<1024, 64>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %.2f ns per slot (%d instr)\n", name, ms * 1e6 / (iters * 256.0), ops_per_slot);
}
int main() {
    float* d; (void)hipMalloc(&d, 4096);
    run<7, 0>(d, "fma", 1);
    run<0, 0>(d, "add quad_perm dpp (fused)", 1);
    run<1, 0>(d, "2 x v_mov_dpp + fma", 3);
    run<2, 0>(d, "add row_shr:1 dpp", 1);
    run<3, 0>(d, "ds_swizzle + add", 2);
    run<4, 0>(d, "2 mov + permlane32_swap + add", 4);
    run<5, 0>(d, "mfma 4x4x1 gather + 3 add", 4);
    run<6, 1>(d, "dpp add + 1 fma", 2);  run<7, 1>(d, "2 fma", 2);
    run<6, 3>(d, "dpp add + 3 fma", 4);  run<7, 3>(d, "4 fma", 4);
    run<6, 7>(d, "dpp add + 7 fma", 8);  run<7, 7>(d, "8 fma", 8);
    return 0;
}
