// Dependent-issue latency and independent issue interval of the vector instruction forms the Swimmer sub-step is made
// of, for ONE wavefront on a SIMD, in shader cycles (s_memtime): C interleaved dependency chains of each form.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
// one "slot" = the instruction applied to chains 0..C-1 in turn
#define SLOT1(I0) I0
#define SLOT2(I0, I1) I0 I1
#define SLOT4(I0, I1, I2, I3) I0 I1 I2 I3
#define DEFK(NAME, C, BODY)                                                                                     \
    __global__ void __launch_bounds__(64) NAME(float* out, uint64_t* cyc, int iters) {                          \
        float v0 = threadIdx.x * 1e-3f + 1.f, v1 = v0 + 1.f, v2 = v0 + 2.f, v3 = v0 + 3.f, a = 0.999f, b = 1e-3f; \
        uint64_t t0 = __builtin_readcyclecounter();                                                             \
        for (int i = 0; i < iters; ++i) {                                                                       \
            asm volatile(REP16(BODY) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(a), "v"(b) : "vcc");        \
        }                                                                                                       \
        uint64_t t1 = __builtin_readcyclecounter();                                                             \
        if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;                                              \
        if (v0 + v1 + v2 + v3 == 123.456f) out[0] = v0;                                                         \
    }
#define I_FMAC(v) "v_fmac_f32_e32 " v ", %4, " v "\n\t"          /* VOP2: v = a * v + v  (dependent through dst and src1) */
#define I_MUL(v) "v_mul_f32_e32 " v ", %4, " v "\n\t"
#define I_FMA3(v) "v_fma_f32 " v ", " v ", %4, %5\n\t"           /* VOP3 */
#define I_FMAK(v) "v_fmac_f32_e32 " v ", 0x3a83126f, " v "\n\t"  /* VOP2 + literal */
#define I_ADDDPP(v) "v_add_f32_dpp " v ", " v ", " v " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define I_MOVDPP(v) "v_mov_b32_dpp " v ", " v " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define I_RCP(v) "v_rcp_f32_e32 " v ", " v "\n\t"
#define I_MED3(v) "v_med3_f32 " v ", " v ", %4, %5\n\t"
#define I_CMPSEL(v) "v_cmp_neq_f32_e32 vcc, 0, " v "\n\tv_cndmask_b32_e32 " v ", %4, " v ", vcc\n\t"
#define I_MULNEG(v) "v_mul_f32_e64 " v ", " v ", -%4\n\t"
#define ALLC(NAME, I)                          \
    DEFK(NAME##_1, 1, SLOT1(I("%0")))          \
    DEFK(NAME##_2, 2, SLOT2(I("%0"), I("%1"))) \
    DEFK(NAME##_4, 4, SLOT4(I("%0"), I("%1"), I("%2"), I("%3")))
ALLC(fmac, I_FMAC) ALLC(mul, I_MUL) ALLC(fma3, I_FMA3) ALLC(fmak, I_FMAK) ALLC(adddpp, I_ADDDPP) ALLC(movdpp, I_MOVDPP)
ALLC(rcp, I_RCP) ALLC(med3, I_MED3) ALLC(cmpsel, I_CMPSEL) ALLC(mulneg, I_MULNEG)
// a DPP consumer right behind a plain producer and vice versa (the two hazards of the exchange pattern)
DEFK(mul_then_dpp_1, 1, "v_mul_f32_e32 %0, %4, %0\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t")
DEFK(mul_then_dpp_2, 2, "v_mul_f32_e32 %0, %4, %0\n\tv_mul_f32_e32 %1, %4, %1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t")

#define DEFK2(NAME, BODY)                                                                                      \
    __global__ void __launch_bounds__(64) NAME(float* out, uint64_t* cyc, int iters) {                          \
        typedef float f2 __attribute__((ext_vector_type(2)));                                                   \
        f2 v0 = {threadIdx.x * 1e-3f + 1.f, 2.f}, v1 = v0 + 1.f, a = {0.999f, 1.001f}, b = {1e-3f, 2e-3f};      \
        uint64_t t0 = __builtin_readcyclecounter();                                                             \
        for (int i = 0; i < iters; ++i) {                                                                       \
            asm volatile(REP16(BODY) : "+v"(v0), "+v"(v1) : "v"(a), "v"(b));                                    \
        }                                                                                                       \
        uint64_t t1 = __builtin_readcyclecounter();                                                             \
        if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;                                              \
        if (v0.x + v1.x + v0.y + v1.y == 123.456f) out[0] = v0.x;                                               \
    }
DEFK2(pkfma_1, "v_pk_fma_f32 %0, %0, %2, %3\n\t")
DEFK2(pkfma_2, "v_pk_fma_f32 %0, %0, %2, %3\n\tv_pk_fma_f32 %1, %1, %2, %3\n\t")
DEFK2(pkfma_mod_1, "v_pk_fma_f32 %0, %0, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]\n\t")
DEFK2(pkmul_1, "v_pk_mul_f32 %0, %0, %2\n\t")
DEFK2(pkmul_2, "v_pk_mul_f32 %0, %0, %2\n\tv_pk_mul_f32 %1, %1, %2\n\t")
DEFK2(pkadd_1, "v_pk_add_f32 %0, %0, %3\n\t")
DEFK2(pkadd_mod_2, "v_pk_add_f32 %0, %0, %3 neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %1, %1, %3 op_sel:[1,0] op_sel_hi:[0,1]\n\t")
DEFK(fmac_nop0, 1, "v_fmac_f32_e32 %0, %4, %0\n\ts_nop 0\n\t")
DEFK(fmac_nop1, 1, "v_fmac_f32_e32 %0, %4, %0\n\ts_nop 1\n\t")
DEFK(fmac_nop3, 1, "v_fmac_f32_e32 %0, %4, %0\n\ts_nop 3\n\t")
DEFK(fmac_salu, 1, "v_fmac_f32_e32 %0, %4, %0\n\ts_mov_b32 vcc_lo, 0\n\t")
DEFK(fmac_2salu, 1, "v_fmac_f32_e32 %0, %4, %0\n\ts_mov_b32 vcc_lo, 0\n\ts_mov_b32 vcc_hi, 0\n\t")
template <class K>
void run(K kern, const char* name, int chains, int per_body, float* d, uint64_t* c) {
    const int iters = 2000;
    kern<<<1024, 64>>>(d, c, 10);
    (void)hipDeviceSynchronize();
    kern<<<1024, 64>>>(d, c, iters);
    (void)hipDeviceSynchronize();
    uint64_t h = 0; (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    const double per_instr = (double)h / ((double)iters * 16 * per_body);
    printf("%-16s chains %d: %6.2f cycles per instruction  (%6.2f per chain step)\n", name, chains, per_instr, per_instr * chains);
}
int main() {
    float* d; uint64_t* c; (void)hipMalloc(&d, 4096); (void)hipMalloc(&c, 64);
#define RUN3(N, PB) run(N##_1, #N, 1, PB, d, c); run(N##_2, #N, 2, 2 * PB, d, c); run(N##_4, #N, 4, 4 * PB, d, c);
    RUN3(fmac, 1) RUN3(mul, 1) RUN3(fma3, 1) RUN3(fmak, 1) RUN3(adddpp, 1) RUN3(movdpp, 1) RUN3(rcp, 1) RUN3(med3, 1) RUN3(cmpsel, 2) RUN3(mulneg, 1)
    run(mul_then_dpp_1, "mul->dpp", 1, 2, d, c); run(mul_then_dpp_2, "mul->dpp", 2, 4, d, c);
    run(pkfma_1, "pk_fma", 1, 1, d, c); run(pkfma_2, "pk_fma", 2, 2, d, c); run(pkfma_mod_1, "pk_fma op_sel,neg", 1, 1, d, c);
    run(pkmul_1, "pk_mul", 1, 1, d, c); run(pkmul_2, "pk_mul", 2, 2, d, c); run(pkadd_1, "pk_add", 1, 1, d, c); run(pkadd_mod_2, "pk_add mods", 2, 2, d, c);
    run(fmac_nop0, "fmac + s_nop 0 (per pair)", 1, 1, d, c); run(fmac_nop1, "fmac + s_nop 1 (per pair)", 1, 1, d, c); run(fmac_nop3, "fmac + s_nop 3 (per pair)", 1, 1, d, c);
    run(fmac_salu, "fmac + s_mov (per pair)", 1, 1, d, c); run(fmac_2salu, "fmac + 2 s_mov (per triple)", 1, 1, d, c);
    return 0;
}
