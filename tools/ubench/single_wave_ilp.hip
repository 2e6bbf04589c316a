// How fast does ONE wavefront per SIMD issue FP32 vector instructions, as a function of the number of independent
// dependency chains in its stream (1, 2, 4, 8), and for DPP-modified / VOP3 / literal-operand forms?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAINS, int FORM>
__global__ void __launch_bounds__(64) k(float* out, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = a + c;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 256 / CHAINS; ++u) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if (FORM == 0) v[c] = __builtin_fmaf(v[c], b, a);                       // v_fmac / v_fma
                else if (FORM == 1) v[c] = v[c] + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v[c]), 0xB1, 0xf, 0xf, true));  // add_dpp (quad_perm 1,0,3,2)
                else if (FORM == 2) v[c] = __builtin_fmaf(v[c], 1.000123f, 0.371f);    // literal operands
                else v[c] = __builtin_fmaf(-v[c], b, a);                                // VOP3 (neg modifier)
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) s += v[c];
    if (s == 123.456f) out[threadIdx.x] = s;
}
template <int CHAINS, int FORM>
void run(float* d, const char* name) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    for (int waves_per_cu : {1, 4, 8}) {
        k<CHAINS, FORM><<<256 * waves_per_cu, 64>>>(d, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<CHAINS, FORM><<<256 * waves_per_cu, 64>>>(d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-8s chains %d, %d wave(s)/CU: %.2f ns per instruction per wave (4 cycles at 2.4 GHz = 1.67 ns)\n", name, CHAINS, waves_per_cu,
               ms * 1e6 / (iters * 256.0));
    }
}
int main() {
    float* d; hipMalloc(&d, 4096);
    run<1, 0>(d, "fma"); run<2, 0>(d, "fma"); run<4, 0>(d, "fma"); run<8, 0>(d, "fma");
    run<1, 1>(d, "add_dpp"); run<2, 1>(d, "add_dpp"); run<4, 1>(d, "add_dpp"); run<8, 1>(d, "add_dpp");
    run<1, 2>(d, "literal"); run<4, 2>(d, "literal");
    run<1, 3>(d, "vop3"); run<4, 3>(d, "vop3");
    return 0;
}
