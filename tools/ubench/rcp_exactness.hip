// Is v_rcp_f32 on gfx950 the correctly rounded reciprocal?  And rcp + one / two Newton steps?
// All 2^23 mantissas at several binades, checked on the host against 1.0f / x (IEEE).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
__global__ void k(uint32_t expo, float* r0, float* r1, float* r2) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    const float x = __builtin_bit_cast(float, (expo << 23) | m);
    const float a = __builtin_amdgcn_rcpf(x);
    const float e0 = __builtin_fmaf(-x, a, 1.0f);
    const float b = __builtin_fmaf(e0, a, a);
    const float e1 = __builtin_fmaf(-x, b, 1.0f);
    const float c = __builtin_fmaf(e1, b, b);
    r0[m] = a; r1[m] = b; r2[m] = c;
}
int main() {
    const size_t N = 1u << 23;
    float *d0, *d1, *d2;
    (void)hipMalloc(&d0, N * 4); (void)hipMalloc(&d1, N * 4); (void)hipMalloc(&d2, N * 4);
    std::vector<float> h0(N), h1(N), h2(N);
    for (uint32_t expo : {127u, 128u, 100u, 126u, 150u, 2u, 252u}) {
        k<<<N / 256, 256>>>(expo, d0, d1, d2);
        (void)hipMemcpy(h0.data(), d0, N * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(h1.data(), d1, N * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(h2.data(), d2, N * 4, hipMemcpyDeviceToHost);
        size_t bad0 = 0, bad1 = 0, bad2 = 0; int maxulp0 = 0;
        for (uint32_t m = 0; m < N; ++m) {
            uint32_t bits = (expo << 23) | m; float x; memcpy(&x, &bits, 4);
            volatile float ref = 1.0f / x;
            float rf = ref;
            uint32_t rb, b0; memcpy(&rb, &rf, 4); memcpy(&b0, &h0[m], 4);
            if (h0[m] != rf) { ++bad0; int u = (int)b0 - (int)rb; if (u < 0) u = -u; if (u > maxulp0) maxulp0 = u; }
            if (h1[m] != rf) ++bad1;
            if (h2[m] != rf) ++bad2;
        }
        printf("exponent %3u: v_rcp_f32 != RN(1/x) for %zu of %zu (max %d ulp); after 1 Newton step %zu; after 2 steps %zu\n", expo, bad0, N,
               maxulp0, bad1, bad2);
    }
    return 0;
}
