// Which (row, column) of D does register j of lane l hold for v_mfma_f64_16x16x4_f64?  (operand layout assumed:
// A[m = l % 16][k = l / 16], B[k = l / 16][n = l % 16]; A[m][0] = m + 1, B[0][n] = 100 (n + 1), other k zero.)
#include <hip/hip_runtime.h>
#include <cstdio>
using f64x4 = __attribute__((ext_vector_type(4))) double;
__global__ void k(double* out) {
    const int l = threadIdx.x;
    const double a = (l / 16 == 0) ? (double)(l % 16 + 1) : 0.0;
    const double b = (l / 16 == 0) ? 100.0 * (l % 16 + 1) : 0.0;
    f64x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = acc[j];
}
int main() {
    double* d; double h[256];
    hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 1) {
        if (l % 16 > 1 && l % 16 < 15) continue;
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) { int v = (int)h[l * 4 + j]; int m = (v % 100 == 0) ? 0 : 0; (void)m;
            // v = (m+1) * 100 * (n+1): not unique; print raw
            printf(" %6.0f", h[l * 4 + j]); }
        printf("\n");
    }
    return 0;
}
