// tools/ubench/dot2_residual.hip -- is  r = a - bf16(a)  through v_dot2c_f32_bf16 (packed hi pair x (-1, 0) / (0, -1),
// accumulated onto a) EXACT, and what does it cost a lone wavefront?  The split of policy_split_kernels.hip spends two
// unpack instructions (v_and / v_lshlrev) and a packed subtraction per pair and stage to form the residual; the dot
// product form needs two instructions per pair and no unpack.  Checked here against the subtraction form, bit for bit,
// over magnitudes from 2^-60 to 2^20, zeros, and residuals that are f32 denormals (a dot instruction that flushed them
// would lose the lo part of tiny operands); timed as C interleaved chains like valu_latency.hip.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/dot2_residual tools/ubench/dot2_residual.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __forceinline__ bf16x2 sel(unsigned bits) {
    unsigned v;
    asm("v_mov_b32 %0, %1" : "=v"(v) : "s"(bits));
    return __builtin_bit_cast(bf16x2, v);
}

__global__ void split_both(const float* a, int n, float* sub, float* dot, float* cst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    const f32x2 v = {a[2 * i], a[2 * i + 1]};
    // reference: the subtraction form (exact by construction: every subtraction is of a value and its rounding)
    const bf16x2 h = __builtin_convertvector(v, bf16x2);
    const f32x2 r = v - __builtin_convertvector(h, f32x2);
    const bf16x2 m = __builtin_convertvector(r, bf16x2);
    const f32x2 l = r - __builtin_convertvector(m, f32x2);
    // dot form, the two selectors handed over as CONSTANTS: the compiler encodes (-1, 0) as the inline constant "-1.0",
    // which the VOP2 instruction reads as (0, -1) -- the .x results of this form are wrong (kept: it is the finding)
    const bf16x2 c0 = {(__bf16)-1.0f, (__bf16)0.0f}, c1 = {(__bf16)0.0f, (__bf16)-1.0f};
    const f32x2 rc = {__builtin_amdgcn_fdot2_f32_bf16(h, c0, v[0], false), __builtin_amdgcn_fdot2_f32_bf16(h, c1, v[1], false)};
    cst[2 * i + 0] = rc[0]; cst[2 * i + 1] = rc[1];
    // dot form, selectors in registers (what policy_split_kernels.hip does)
    const bf16x2 e0 = sel(0x0000bf80u), e1 = sel(0xbf800000u);
    const f32x2 rd = {__builtin_amdgcn_fdot2_f32_bf16(h, e0, v[0], false), __builtin_amdgcn_fdot2_f32_bf16(h, e1, v[1], false)};
    const bf16x2 md = __builtin_convertvector(rd, bf16x2);
    const f32x2 ld = {__builtin_amdgcn_fdot2_f32_bf16(md, e0, rd[0], false), __builtin_amdgcn_fdot2_f32_bf16(md, e1, rd[1], false)};
    sub[4 * i + 0] = r[0]; sub[4 * i + 1] = r[1]; sub[4 * i + 2] = l[0]; sub[4 * i + 3] = l[1];
    dot[4 * i + 0] = rd[0]; dot[4 * i + 1] = rd[1]; dot[4 * i + 2] = ld[0]; dot[4 * i + 3] = ld[1];
}

// timing: eight independent pairs per lane, each iteration = one split stage per pair (cvt_pk + residual) and one packed
// multiply that feeds the residual back (common to both forms); the compiler's own instruction selection, as in the
// kernel that would use it
template <bool DOT>
__global__ void __launch_bounds__(64) stage_time(float* out, uint64_t* cyc, int iters) {
    f32x2 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = f32x2{threadIdx.x * 1e-3f + 1.f + j, 2.f + j};
    const bf16x2 e0 = sel(0x0000bf80u), e1 = sel(0xbf800000u);
    const f32x2 big = {300.0f, 300.0f};
    unsigned acc = 0;
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bf16x2 h = __builtin_convertvector(v[j], bf16x2);
            f32x2 r;
            if (DOT) r = f32x2{__builtin_amdgcn_fdot2_f32_bf16(h, e0, v[j][0], false), __builtin_amdgcn_fdot2_f32_bf16(h, e1, v[j][1], false)};
            else r = v[j] - __builtin_convertvector(h, f32x2);
            acc ^= __builtin_bit_cast(unsigned, h);
            v[j] = r * big + f32x2{1.0f, 1.0f};
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) sum += v[j][0] + v[j][1];
    if (sum == 123.456f) out[0] = sum + acc;
}

template <bool DOT>
static void run(const char* name) {
    float* out; uint64_t* cyc;
    (void)hipMalloc(&out, 4); (void)hipMalloc(&cyc, 8);
    const int iters = 4000;
    hipLaunchKernelGGL(stage_time<DOT>, dim3(1), dim3(64), 0, 0, out, cyc, 10);
    hipLaunchKernelGGL(stage_time<DOT>, dim3(1), dim3(64), 0, 0, out, cyc, iters);
    uint64_t c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("{\"form\": \"%s\", \"clock_ticks_per_pair_and_stage\": %.2f}\n", name, (double)c / ((double)iters * 8));
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    const int n = 1 << 20;
    std::vector<float> a(n);
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
    for (int i = 0; i < n; ++i) {
        const int e = (int)(rnd() % 81) - 60;                         // 2^-60 .. 2^20
        const float m = 1.0f + (rnd() >> 9) * (1.0f / 8388608.0f);
        float v = ldexpf(m, e) * ((rnd() & 1) ? 1.f : -1.f);
        if (i % 97 == 0) v = 0.0f;
        if (i % 101 == 0) v = ldexpf(m, -120 - (int)(rnd() % 6));     // residuals below 2^-126: f32 denormals
        if (i % 103 == 0) { uint32_t b; memcpy(&b, &v, 4); b &= 0xffff0000u; memcpy(&v, &b, 4); }   // already a bf16
        a[i] = v;
    }
    float *da, *ds, *dd, *dc;
    hipMalloc(&da, n * 4); hipMalloc(&ds, 2 * n * 4); hipMalloc(&dd, 2 * n * 4); hipMalloc(&dc, n * 4);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(split_both, dim3(n / 2 / 256), dim3(256), 0, 0, da, n, ds, dd, dc);
    std::vector<float> hs(2 * n), hd(2 * n);
    hipMemcpy(hs.data(), ds, 2 * n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hd.data(), dd, 2 * n * 4, hipMemcpyDeviceToHost);
    long bad = 0, bad_denorm = 0, denorm_cases = 0;
    for (int i = 0; i < 2 * n; ++i) {
        uint32_t x, y; memcpy(&x, &hs[i], 4); memcpy(&y, &hd[i], 4);
        const bool den = hs[i] != 0.0f && fabsf(hs[i]) < 1.17549435e-38f;
        denorm_cases += den;
        if (x != y && !(hs[i] == 0.0f && hd[i] == 0.0f)) { ++bad; bad_denorm += den; if (bad <= 5) printf("mismatch %d: sub %a dot %a (a = %a)\n", i, hs[i], hd[i], a[(i / 4) * 2 + (i & 1)]); }
    }
    {   // the constant-operand form against the subtraction form (first residual only)
        std::vector<float> hc(n);
        hipMemcpy(hc.data(), dc, n * 4, hipMemcpyDeviceToHost);
        long bx = 0, by = 0;
        for (int i = 0; i < n / 2; ++i) {
            bx += memcmp(&hc[2 * i], &hs[4 * i], 4) != 0 && !(hc[2 * i] == 0.0f && hs[4 * i] == 0.0f);
            by += memcmp(&hc[2 * i + 1], &hs[4 * i + 1], 4) != 0 && !(hc[2 * i + 1] == 0.0f && hs[4 * i + 1] == 0.0f);
        }
        printf("{\"selectors_as_constants\": {\"pairs\": %d, \"wrong_x\": %ld, \"wrong_y\": %ld}}\n", n / 2, bx, by);
    }
    printf("{\"exactness\": {\"values\": %d, \"mismatches\": %ld, \"mismatches_where_the_residual_is_denormal\": %ld, \"denormal_residuals\": %ld}}\n",
           2 * n, bad, bad_denorm, denorm_cases);
    run<false>("cvt_pk + unpack (lshl, and) + v_pk_add_f32");
    run<true>("cvt_pk + 2 x v_dot2c_f32_bf16");
    return bad != 0;
}
