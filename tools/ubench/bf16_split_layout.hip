// bf16_split_layout.hip -- the layout facts policy_split_kernels.hip is built on, checked on the device:
//   1. v_mfma_f32_32x32x16_bf16: A lane (i = l & 31, h = l >> 5) element j and B lane (n = l & 31, h) element j meet at
//      the same k; D register r of lane (n, h) is row (r & 3) + 8 (r >> 2) + 4 h, column n.
//   2. ds_read_b64_tr_b16 in a [32 rows][stride 72 B] image of 16-bit elements: a lane of 16-lane group g = l >> 4 that
//      supplies the address of row 4 sq + ((l & 15) >> 2), columns 16 cg + 4 (l & 3) .. + 3 receives rows 4 sq .. 4 sq + 3
//      of column 16 cg + (l & 15).
//   3. the three-way bf16 split of an f32 is exact (hi + mid + lo == x) and six cross terms reproduce a float64 dot
//      product to f32 accuracy.
// hipcc --offload-arch=gfx950 -O2 -o tools/ubench/bf16_split_layout tools/ubench/bf16_split_layout.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__host__ __device__ constexpr int frag_unit(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

__global__ void mfma_layout(const float* A, const float* B, float* D) {   // A [32][16], B [16][32], D [32][32]
    const int l = threadIdx.x, n = l & 31, h = l >> 5;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (__bf16)A[n * 16 + 8 * h + j];
        b[j] = (__bf16)B[(8 * h + j) * 32 + n];
    }
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[frag_unit(r, h) * 32 + n] = c[r];
}

constexpr int STRIDE = 72;
__global__ void tr_layout(unsigned short* out /* [2 kb][2 jh][64 lanes][4] */, unsigned short* raw /* [64][4] */) {
    __shared__ __attribute__((aligned(16))) unsigned char img[32 * STRIDE + 4096];
    const int l = threadIdx.x;
    typedef __attribute__((address_space(3))) s16x4* lp;
    // natural pattern: element index e at byte 2 e, lane l supplies byte 8 l
    unsigned short* lin = reinterpret_cast<unsigned short*>(img);
    for (int e = l; e < 1024; e += 64) lin[e] = (unsigned short)e;
    __syncthreads();
    s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(img + 8 * l));
    for (int e = 0; e < 4; ++e) raw[l * 4 + e] = (unsigned short)t[e];
    __syncthreads();
    // the image of the kernel: value = 32 row + column
    for (int e = l; e < 32 * 32; e += 64) {
        const int row = e / 32, col = e % 32;
        *reinterpret_cast<unsigned short*>(img + row * STRIDE + 2 * col) = (unsigned short)(32 * row + col);
    }
    __syncthreads();
    const int lh = l >> 5, cg = (l >> 4) & 1, i = l & 15;
    const int base = (4 * lh + (i >> 2)) * STRIDE + 32 * cg + 8 * (i & 3);
    for (int kb = 0; kb < 2; ++kb)
        for (int jh = 0; jh < 2; ++jh) {
            s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(img + base + (2 * kb + jh) * 8 * STRIDE));
            for (int e = 0; e < 4; ++e) out[((kb * 2 + jh) * 64 + l) * 4 + e] = (unsigned short)v[e];
        }
}

// dot products of length 16 by six bf16 cross terms: D = A B with A, B split three ways
__global__ void split_dot(const float* A, const float* B, float* D, float* resid) {
    const int l = threadIdx.x, n = l & 31, h = l >> 5;
    bf16x8 a[3], b[3];
    float worst = 0.0f;
    for (int j = 0; j < 8; ++j) {
        const float xs[2] = {A[n * 16 + 8 * h + j], B[(8 * h + j) * 32 + n]};
        for (int w = 0; w < 2; ++w) {
            const float x = xs[w];
            const __bf16 p0 = (__bf16)x;
            const float r1 = x - (float)p0;
            const __bf16 p1 = (__bf16)r1;
            const float r2 = r1 - (float)p1;
            const __bf16 p2 = (__bf16)r2;
            worst = fmaxf(worst, fabsf(((float)p0 + (float)p1 + (float)p2) - x) + fabsf(r2 - (float)p2));
            if (w == 0) { a[0][j] = p0; a[1][j] = p1; a[2][j] = p2; }
            else { b[0][j] = p0; b[1][j] = p1; b[2][j] = p2; }
        }
    }
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[frag_unit(r, h) * 32 + n] = c[r];
    resid[l] = worst;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
    std::vector<float> A(32 * 16), B(16 * 32), D(32 * 32);
    srand(3);
    for (auto& v : A) v = (float)(rand() % 17 - 8);          // exact in bf16
    for (auto& v : B) v = (float)(rand() % 13 - 6);
    float *dA, *dB, *dD, *dR;
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dD, D.size() * 4));
    CK(hipMalloc(&dR, 64 * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(mfma_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 32; ++i)
        for (int n = 0; n < 32; ++n) {
            float s = 0;
            for (int k = 0; k < 16; ++k) s += A[i * 16 + k] * B[k * 32 + n];
            if (s != D[i * 32 + n]) { if (bad < 4) printf("  mfma D[%d][%d] = %g, expected %g\n", i, n, D[i * 32 + n], s); ++bad; }
        }
    printf("mfma_f32_32x32x16_bf16 layout: %d mismatches of 1024\n", bad);

    unsigned short *dO, *dRaw;
    CK(hipMalloc(&dO, 2 * 2 * 64 * 4 * 2)); CK(hipMalloc(&dRaw, 64 * 4 * 2));
    hipLaunchKernelGGL(tr_layout, dim3(1), dim3(64), 0, 0, dO, dRaw);
    std::vector<unsigned short> O(2 * 2 * 64 * 4), Raw(64 * 4);
    CK(hipMemcpy(O.data(), dO, O.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(Raw.data(), dRaw, Raw.size() * 2, hipMemcpyDeviceToHost));
    int badraw = 0;
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 4; ++j)
            if (Raw[l * 4 + j] != (l & 15) + 16 * j + 64 * (l >> 4)) ++badraw;
    printf("ds_read_b64_tr_b16 natural pattern (lane l, elem j = (l&15) + 16 j + 64 (l>>4)): %d mismatches of 256\n", badraw);
    if (badraw) {
        for (int l = 0; l < 20; ++l)
            printf("  lane %2d: %4d %4d %4d %4d\n", l, Raw[l * 4], Raw[l * 4 + 1], Raw[l * 4 + 2], Raw[l * 4 + 3]);
    }
    int badtr = 0;
    for (int kb = 0; kb < 2; ++kb)
        for (int jh = 0; jh < 2; ++jh)
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 4; ++e) {
                    const int lh = l >> 5, unit = l & 31;
                    const int sample = frag_unit(8 * kb + 4 * jh + e, lh);
                    const int got = O[((kb * 2 + jh) * 64 + l) * 4 + e];
                    if (got != 32 * sample + unit) {
                        if (badtr < 8) printf("  tr kb %d jh %d lane %d e %d: got row %d col %d, expected row %d col %d\n", kb, jh, l, e,
                                              got / 32, got % 32, sample, unit);
                        ++badtr;
                    }
                }
    printf("transposing read of the [32][72 B] image: %d mismatches of 1024\n", badtr);

    for (auto& v : A) v = (float)((rand() / (double)RAND_MAX - 0.5) * 2.0);
    for (auto& v : B) v = (float)((rand() / (double)RAND_MAX - 0.5) * 1e-3);
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(split_dot, dim3(1), dim3(64), 0, 0, dA, dB, dD, dR);
    CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
    std::vector<float> R(64);
    CK(hipMemcpy(R.data(), dR, 64 * 4, hipMemcpyDeviceToHost));
    double worst_rel = 0, worst_f32 = 0, worst_res = 0;
    for (int l = 0; l < 64; ++l) worst_res = fmax(worst_res, R[l]);
    for (int i = 0; i < 32; ++i)
        for (int n = 0; n < 32; ++n) {
            double s = 0, sabs = 0;
            float f = 0;
            for (int k = 0; k < 16; ++k) {
                s += (double)A[i * 16 + k] * (double)B[k * 32 + n];
                sabs += fabs((double)A[i * 16 + k] * (double)B[k * 32 + n]);
                f = fmaf(A[i * 16 + k], B[k * 32 + n], f);
            }
            worst_rel = fmax(worst_rel, fabs(D[i * 32 + n] - s) / sabs);
            worst_f32 = fmax(worst_f32, fabs((double)f - s) / sabs);
        }
    printf("three-way split: residual of hi + mid + lo = %g; six-term dot product vs float64: %.3g of sum|a b| (an f32 fma chain: %.3g)\n",
           worst_res, worst_rel, worst_f32);
    return (bad || badtr) ? 1 : 0;
}
