// One wavefront per SIMD: how many vector instructions of which kind hide in the shadow of an f32-input MFMA
// (v_mfma_f32_16x16x4_f32, 32 cycles of matrix pipe)?  Four rotating accumulators (no accumulator dependency),
// K filler instructions of one kind after every MFMA; shader cycles per (MFMA + fillers).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
#define F_FMA "v_fmac_f32_e32 %4, %6, %4\n\t"
#define F_MUL "v_mul_f32_e32 %5, %6, %5\n\t"
#define F_EXP "v_exp_f32_e32 %5, %5\n\t"
#define F_MOV "v_mov_b32_e32 %5, %4\n\t"
#define F_INT "v_add_u32_e32 %5, %5, %4\n\t"
#define F_PK "v_pk_mul_f32 %7, %7, %7\n\t"
#define MF(a) "v_mfma_f32_16x16x4_f32 " a ", %6, %6, " a "\n\t"
#define DEFK(NAME, FILL)                                                                                          \
    __global__ void __launch_bounds__(64) NAME(float* out, uint64_t* cyc, int iters) {                           \
        f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;                                                      \
        float v0 = threadIdx.x * 1e-3f, v1 = 1.0f, w = 0.5f;                                                     \
        typedef float f2 __attribute__((ext_vector_type(2)));                                                    \
        f2 p = {1.0f, 1.0f};                                                                                     \
        uint64_t t0 = __builtin_readcyclecounter();                                                              \
        for (int i = 0; i < iters; ++i) {                                                                        \
            asm volatile(REP4(MF("%0") FILL MF("%1") FILL MF("%2") FILL MF("%3") FILL)                           \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(v0), "+v"(v1) : "v"(w), "v"(p));         \
        }                                                                                                        \
        uint64_t t1 = __builtin_readcyclecounter();                                                              \
        if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;                                               \
        if (a0[0] + a1[0] + a2[0] + a3[0] + v0 + v1 == 123.456f) out[0] = v0;                                    \
    }
DEFK(k_none, "")
DEFK(k_fma2, F_FMA F_MUL) DEFK(k_fma4, F_FMA F_MUL F_FMA F_MUL) DEFK(k_fma6, F_FMA F_MUL F_FMA F_MUL F_FMA F_MUL)
DEFK(k_fma8, REP4(F_FMA F_MUL))
DEFK(k_exp2, F_EXP F_EXP) DEFK(k_exp4, REP4(F_EXP))
DEFK(k_mov4, REP4(F_MOV)) DEFK(k_mov8, REP8(F_MOV))
DEFK(k_int4, REP4(F_INT)) DEFK(k_int8, REP8(F_INT))
DEFK(k_pk2, F_PK F_PK) DEFK(k_pk4, REP4(F_PK))
template <class K> void run(K kern, const char* name, float* d, uint64_t* c) {
    const int iters = 2000;
    kern<<<1024, 64>>>(d, c, 10); (void)hipDeviceSynchronize();
    kern<<<1024, 64>>>(d, c, iters); (void)hipDeviceSynchronize();
    uint64_t h = 0; (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("%-28s %6.1f cycles per MFMA + fillers\n", name, (double)h / (iters * 16.0));
}
int main() {
    float* d; uint64_t* c; (void)hipMalloc(&d, 4096); (void)hipMalloc(&c, 64);
    run(k_none, "MFMA only", d, c);
    run(k_fma2, "+ 2 fma/mul", d, c); run(k_fma4, "+ 4 fma/mul", d, c); run(k_fma6, "+ 6 fma/mul", d, c); run(k_fma8, "+ 8 fma/mul", d, c);
    run(k_exp2, "+ 2 v_exp", d, c); run(k_exp4, "+ 4 v_exp", d, c);
    run(k_mov4, "+ 4 v_mov", d, c); run(k_mov8, "+ 8 v_mov", d, c);
    run(k_int4, "+ 4 v_add_u32", d, c); run(k_int8, "+ 8 v_add_u32", d, c);
    run(k_pk2, "+ 2 v_pk_mul", d, c); run(k_pk4, "+ 4 v_pk_mul", d, c);
    return 0;
}
