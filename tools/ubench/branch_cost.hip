// What a branch instruction costs ONE wavefront on a SIMD, in shader cycles (s_memtime): a body of 16 dependent vector
// instructions per block, 16 blocks per loop iteration, with one scalar branch per block -- never taken / always taken
// (over an 8-instruction block that then never runs) / an exec-mask branch (s_cbranch_execz, not taken) -- against the
// same body without branches.  The lane-group rollouts run exactly this regime (one instruction stream per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define VALU16 REP16("v_fmac_f32_e32 %0, %1, %0\n\t")
#define VALU8 REP4("v_fmac_f32_e32 %0, %1, %0\n\t") REP4("v_fmac_f32_e32 %0, %1, %0\n\t")
#define DEFK(NAME, BLOCK)                                                                                \
    __global__ void __launch_bounds__(64) NAME(float* out, uint64_t* cyc, int iters, int zero) {         \
        float v = threadIdx.x * 1e-3f + 1.f, a = 0.999f;                                                 \
        uint64_t t0 = __builtin_readcyclecounter();                                                      \
        for (int i = 0; i < iters; ++i) {                                                                \
            asm volatile(REP16(BLOCK) : "+v"(v) : "v"(a), "s"(zero) : "scc", "vcc", "s20", "s21");        \
        }                                                                                                \
        uint64_t t1 = __builtin_readcyclecounter();                                                      \
        if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;                                       \
        if (v == 123.456f) out[0] = v;                                                                   \
    }
// no branch: 16 vector instructions + the scalar compare the branchy forms also carry
DEFK(plain, VALU16 "s_cmp_eq_u32 %2, 1\n\t")
// never taken: falls through into 8 more instructions (so: 24 per block) -- compare with plain24
DEFK(not_taken, VALU16 "s_cmp_eq_u32 %2, 1\n\ts_cbranch_scc1 1f\n\t" VALU8 "1:\n\t")
DEFK(plain24, VALU16 "s_cmp_eq_u32 %2, 1\n\t" VALU8)
// always taken over 8 instructions that never run -- compare with plain
DEFK(taken, VALU16 "s_cmp_eq_u32 %2, 0\n\ts_cbranch_scc1 1f\n\t" VALU8 "1:\n\t")
// exec-mask branch, not taken (exec != 0) -- compare with plain24
DEFK(execz_not_taken, VALU16 "s_cmp_eq_u32 %2, 1\n\ts_cbranch_execz 1f\n\t" VALU8 "1:\n\t")
// vcc branch fed by a vector compare (the rl_any_pos pattern): v_cmp + s_cbranch_vccz, taken
DEFK(vccz_taken, VALU16 "v_cmp_lt_f32_e32 vcc, 1.0e30, %0\n\ts_cbranch_vccz 1f\n\t" VALU8 "1:\n\t")
DEFK(plain_vcmp, VALU16 "v_cmp_lt_f32_e32 vcc, 1.0e30, %0\n\t")

int main() {
    float* out; uint64_t* cyc;
    hipMalloc(&out, 4); hipMalloc(&cyc, 8);
    const int iters = 2000;
#define RUN(K) { hipLaunchKernelGGL(K, dim3(1), dim3(64), 0, 0, out, cyc, iters, 0); hipDeviceSynchronize();                   \
                 hipLaunchKernelGGL(K, dim3(1), dim3(64), 0, 0, out, cyc, iters, 0); hipDeviceSynchronize();                   \
                 uint64_t c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);                                                     \
                 printf("%-18s %8.2f cycles per block (16 blocks x %d iterations)\n", #K, (double)c / (16.0 * iters), iters); }
    RUN(plain) RUN(plain24) RUN(not_taken) RUN(taken) RUN(execz_not_taken) RUN(plain_vcmp) RUN(vccz_taken)
    return 0;
}
