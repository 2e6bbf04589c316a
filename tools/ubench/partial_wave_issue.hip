// Does a wavefront with only 16 (or 32) active lanes issue vector instructions faster than 1 per 4 cycles?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(64) k(float* out, int iters, int active) {
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    float v0 = a, v1 = b, v2 = a + b, v3 = a - b;
    if ((int)threadIdx.x < active) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 64; ++u) {     // 4 independent chains: 256 dependent-free-enough FMAs
                v0 = __builtin_fmaf(v0, b, a); v1 = __builtin_fmaf(v1, b, a);
                v2 = __builtin_fmaf(v2, b, a); v3 = __builtin_fmaf(v3, b, a);
            }
        }
    }
    float s = v0 + v1 + v2 + v3;
    if (s == 123.456f) out[threadIdx.x] = s;
}
int main() {
    float* d; hipMalloc(&d, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    for (int active : {64, 32, 16, 8, 1}) {
        k<<<256, 64>>>(d, 10, active);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<<<256, 64>>>(d, iters, active);     // one wavefront per CU
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("active lanes %2d: %7.3f ms  -> %.2f ns per FMA instruction (4 cycles at 2.4 GHz = 1.67 ns)\n", active, ms,
               ms * 1e6 / (iters * 256.0));
    }
    return 0;
}
