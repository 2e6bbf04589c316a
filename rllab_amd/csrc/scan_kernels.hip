// scan_kernels.hip -- segmented reverse linear-recurrence scans over [T][n] planes.
//
// y[t] = x[t] + c[t] * y[t+1] is a composition of affine maps, hence associative:
// time is cut into KB chunks, every (env, chunk) thread reduces its chunk to an
// affine summary (A, B) with y_first = A + B * carry_in, summaries are exchanged
// through LDS, every thread folds the summaries of the chunks after it into its
// carry-in and then re-walks its chunk writing the true values (the second read
// is served by L2: a workgroup touches < 1 MB).  HBM traffic is therefore the
// algorithmic 12 B + 8 B (f64 baseline) + 1 B per sample for GAE.
// Accumulation is f64 (two flops per element are free next to the loads), the
// stored planes are f32: results sit within 1 ulp_f32 of the reference's
// float64 scipy.lfilter (rllab/misc/special.py:107-111).
#include <hip/hip_runtime.h>
#include "../../include/rllab_amd.h"
#include "capi_util.h"

namespace rl {

constexpr int SCAN_EW = 32;  // envs per workgroup (128-B row segments)
constexpr int SCAN_KB = 32;  // time chunks per workgroup

struct Affine {
    double a, b;  // y = a + b * carry
};

// ---- GAE + returns ---------------------------------------------------------
__global__ void __launch_bounds__(SCAN_EW* SCAN_KB)
gae_kernel(int T, int n, const float* __restrict__ r, const double* __restrict__ v,
           const uint8_t* __restrict__ done, double gamma, double gl, float* __restrict__ adv,
           float* __restrict__ ret, float* __restrict__ und) {
    __shared__ Affine s_adv[SCAN_KB][SCAN_EW];
    __shared__ Affine s_ret[SCAN_KB][SCAN_EW];
    __shared__ Affine s_und[SCAN_KB][SCAN_EW];   // undiscounted return-to-go (discount 1)
    const int e = threadIdx.x, k = threadIdx.y;
    const int i = blockIdx.x * SCAN_EW + e;
    const int L = (T + SCAN_KB - 1) / SCAN_KB;
    const int t0 = k * L;
    const int t1 = min(T, t0 + L);  // exclusive
    const bool live = (i < n) && (t0 < T);

    Affine sa{0.0, 1.0}, sr{0.0, 1.0}, su{0.0, 1.0};
    if (live) {
        // pass 1: chunk summary.  V[t+1] inside the chunk comes from the previous
        // loop iteration; across the chunk boundary it is read directly.
        double ya = 0.0, ba = 1.0, yr = 0.0, br = 1.0, yu = 0.0, bu = 1.0;
        double vnext = 0.0;
        if (v && t1 < T) vnext = v[(size_t)t1 * n + i];
        for (int t = t1 - 1; t >= t0; --t) {
            const size_t off = (size_t)t * n + i;
            const bool end = (t == T - 1) || done[off];
            const double rt = (double)r[off];
            const double vt = v ? v[off] : 0.0;
            const double keep = end ? 0.0 : 1.0;
            const double delta = rt + gamma * vnext * keep - vt;
            ya = delta + gl * keep * ya;
            ba = gl * keep * ba;
            yr = rt + gamma * keep * yr;
            br = gamma * keep * br;
            yu = rt + keep * yu;
            bu = keep * bu;
            vnext = vt;
        }
        sa = Affine{ya, ba};
        sr = Affine{yr, br};
        su = Affine{yu, bu};
    }
    s_adv[k][e] = sa;
    s_ret[k][e] = sr;
    s_und[k][e] = su;
    __syncthreads();
    if (!live) return;

    // carry-in = value at t1 = fold of chunks k+1 .. KB-1 (last chunk has carry 0)
    double ca = 0.0, cr = 0.0, cu = 0.0;
    for (int j = SCAN_KB - 1; j > k; --j) {
        ca = s_adv[j][e].a + s_adv[j][e].b * ca;
        cr = s_ret[j][e].a + s_ret[j][e].b * cr;
        cu = s_und[j][e].a + s_und[j][e].b * cu;
    }
    // pass 2: true values
    double ya = ca, yr = cr, yu = cu;
    double vnext = 0.0;
    if (v && t1 < T) vnext = v[(size_t)t1 * n + i];
    for (int t = t1 - 1; t >= t0; --t) {
        const size_t off = (size_t)t * n + i;
        const bool end = (t == T - 1) || done[off];
        const double rt = (double)r[off];
        const double vt = v ? v[off] : 0.0;
        const double keep = end ? 0.0 : 1.0;
        const double delta = rt + gamma * vnext * keep - vt;
        ya = delta + gl * keep * ya;
        yr = rt + gamma * keep * yr;
        yu = rt + keep * yu;
        adv[off] = (float)ya;
        ret[off] = (float)yr;
        if (und) und[off] = (float)yu;
        vnext = vt;
    }
}

// ---- GAE + returns, the chunk held in registers ------------------------------------------------
// The kernel above walks its chunk twice with one dependent load group per step: at C3's size (4096 envs x 500
// steps, 51 MB) it is latency-bound -- 128 workgroups on 256 CUs, 16 loads in a row per thread -- and reaches 1.5 TB/s.
// Here a thread first issues ALL loads of its chunk (L <= LMAX steps: 3 L independent requests in flight), keeps the
// values in registers for both passes (nothing is read twice), and EW = 16 envs per workgroup put 256 workgroups on
// the chip at 4096 envs.  Neighbouring env groups share 128-B lines of the f32 / u8 rows: the workgroup -> group map
// sends them to the same XCD (workgroups are dealt round-robin over the 8 XCDs), so the line is fetched into one L2.
// Same recurrences in the same order as gae_kernel: bit-identical results.
template <int LMAX, int EW>
__global__ void __launch_bounds__(1024)
gae_reg_kernel(int T, int n, int L, const float* __restrict__ r, const double* __restrict__ v,
               const uint8_t* __restrict__ done, double gamma, double gl, float* __restrict__ adv,
               float* __restrict__ ret, float* __restrict__ und) {
    extern __shared__ Affine s_sum[];              // [3][KB][EW]
    const int e = threadIdx.x, k = threadIdx.y, KB = blockDim.y;
    const int G = gridDim.x;
    const int w = blockIdx.x;
    const int g = (G % 8 == 0) ? (w % 8) * (G / 8) + w / 8 : w;
    const int i = g * EW + e;
    const int t0 = k * L;
    const int t1 = min(T, t0 + L);  // exclusive
    const bool live = (i < n) && (t0 < T);
    Affine* s_adv = s_sum + (size_t)k * EW + e;
    Affine* s_ret = s_adv + (size_t)KB * EW;
    Affine* s_und = s_ret + (size_t)KB * EW;

    float rr[LMAX];
    double vv[LMAX];
    uint32_t ends = 0;                             // bit j: step t0 + j ends its path
    double vlast = 0.0;
    if (live) {
        uint8_t dd[LMAX];
#pragma unroll
        for (int j = 0; j < LMAX; ++j) {
            const int t = t0 + j;
            const size_t off = (size_t)(t < t1 ? t : t0) * n + i;
            rr[j] = r[off];
            vv[j] = v ? v[off] : 0.0;
            dd[j] = done[off];
        }
        if (v && t1 < T) vlast = v[(size_t)t1 * n + i];
#pragma unroll
        for (int j = 0; j < LMAX; ++j) ends |= ((t0 + j == T - 1) || dd[j]) ? (1u << j) : 0u;
    }
    Affine sa{0.0, 1.0}, sr{0.0, 1.0}, su{0.0, 1.0};
    if (live) {
        double ya = 0.0, ba = 1.0, yr = 0.0, br = 1.0, yu = 0.0, bu = 1.0;
        double vnext = vlast;
#pragma unroll
        for (int j = LMAX - 1; j >= 0; --j) {
            if (t0 + j < t1) {
                const double rt = (double)rr[j], vt = vv[j];
                const double keep = ((ends >> j) & 1u) ? 0.0 : 1.0;
                const double delta = rt + gamma * vnext * keep - vt;
                ya = delta + gl * keep * ya;
                ba = gl * keep * ba;
                yr = rt + gamma * keep * yr;
                br = gamma * keep * br;
                yu = rt + keep * yu;
                bu = keep * bu;
                vnext = vt;
            }
        }
        sa = Affine{ya, ba};
        sr = Affine{yr, br};
        su = Affine{yu, bu};
    }
    *s_adv = sa;
    *s_ret = sr;
    *s_und = su;
    __syncthreads();
    if (!live) return;
    double ca = 0.0, cr = 0.0, cu = 0.0;
    for (int j = KB - 1; j > k; --j) {
        const Affine a = s_sum[(size_t)j * EW + e], b = s_sum[((size_t)KB + j) * EW + e], c = s_sum[((size_t)2 * KB + j) * EW + e];
        ca = a.a + a.b * ca;
        cr = b.a + b.b * cr;
        cu = c.a + c.b * cu;
    }
    double ya = ca, yr = cr, yu = cu;
    double vnext = vlast;
#pragma unroll
    for (int j = LMAX - 1; j >= 0; --j) {
        if (t0 + j < t1) {
            const size_t off = (size_t)(t0 + j) * n + i;
            const double rt = (double)rr[j], vt = vv[j];
            const double keep = ((ends >> j) & 1u) ? 0.0 : 1.0;
            const double delta = rt + gamma * vnext * keep - vt;
            ya = delta + gl * keep * ya;
            yr = rt + gamma * keep * yr;
            yu = rt + keep * yu;
            adv[off] = (float)ya;
            ret[off] = (float)yr;
            if (und) und[off] = (float)yu;
            vnext = vt;
        }
    }
}

// ---- plain segmented discount_cumsum --------------------------------------
__global__ void __launch_bounds__(SCAN_EW* SCAN_KB)
discount_cumsum_kernel(int T, int n, const float* __restrict__ x, const uint8_t* __restrict__ done,
                       double discount, float* __restrict__ y) {
    __shared__ Affine s[SCAN_KB][SCAN_EW];
    const int e = threadIdx.x, k = threadIdx.y;
    const int i = blockIdx.x * SCAN_EW + e;
    const int L = (T + SCAN_KB - 1) / SCAN_KB;
    const int t0 = k * L;
    const int t1 = min(T, t0 + L);
    const bool live = (i < n) && (t0 < T);
    Affine sm{0.0, 1.0};
    if (live) {
        double a = 0.0, b = 1.0;
        for (int t = t1 - 1; t >= t0; --t) {
            const size_t off = (size_t)t * n + i;
            const bool end = (t == T - 1) || (done && done[off]);
            const double c = end ? 0.0 : discount;
            a = (double)x[off] + c * a;
            b = c * b;
        }
        sm = Affine{a, b};
    }
    s[k][e] = sm;
    __syncthreads();
    if (!live) return;
    double carry = 0.0;
    for (int j = SCAN_KB - 1; j > k; --j) carry = s[j][e].a + s[j][e].b * carry;
    double a = carry;
    for (int t = t1 - 1; t >= t0; --t) {
        const size_t off = (size_t)t * n + i;
        const bool end = (t == T - 1) || (done && done[off]);
        const double c = end ? 0.0 : discount;
        a = (double)x[off] + c * a;
        y[off] = (float)a;
    }
}

}  // namespace rl

using namespace rl;

extern "C" int rl_gae(int T, int n, const float* rewards, const double* values, const uint8_t* dones,
                      double gamma, double lambda, float* adv, float* ret, float* undiscounted, void* stream) {
    if (T <= 0 || n <= 0 || !rewards || !dones || !adv || !ret)
        return set_error(RL_ERR_ARG, "rl_gae: bad argument");
    constexpr int LMAX = 16;
    if (T <= 64 * LMAX) {
        // register-resident chunks: up to 64 chunks of up to 16 steps; 16 envs per workgroup while 32 would leave CUs idle
#ifndef RL_GAE_NARROW_BELOW
#define RL_GAE_NARROW_BELOW 1024     // workgroups of 32 envs below which 16-env workgroups are launched instead
#endif
#ifndef RL_GAE_KB_NARROW
#define RL_GAE_KB_NARROW 32          // chunks per 16-env workgroup: 32 x 16 steps (512 threads) beat 63 x 8 (14.0 against 22.4 us
#endif                               // at 4096 envs x 500 steps, tools/exp/scan_time.py): the fold over the chunks is serial
        const bool narrow = (n + 31) / 32 < RL_GAE_NARROW_BELOW;
        const int ew = narrow ? 16 : 32, kb_max = narrow ? RL_GAE_KB_NARROW : 32;
        if (T <= kb_max * LMAX) {
            const int L = (T + kb_max - 1) / kb_max, KB = (T + L - 1) / L;
            dim3 grid((n + ew - 1) / ew), block(ew, KB);
            const size_t lds = (size_t)3 * KB * ew * sizeof(Affine);
            if (narrow)
                hipLaunchKernelGGL((gae_reg_kernel<LMAX, 16>), grid, block, lds, (hipStream_t)stream, T, n, L, rewards, values,
                                   dones, gamma, gamma * lambda, adv, ret, undiscounted);
            else
                hipLaunchKernelGGL((gae_reg_kernel<LMAX, 32>), grid, block, lds, (hipStream_t)stream, T, n, L, rewards, values,
                                   dones, gamma, gamma * lambda, adv, ret, undiscounted);
            return check_launch("gae_reg_kernel");
        }
    }
    dim3 grid((n + SCAN_EW - 1) / SCAN_EW), block(SCAN_EW, SCAN_KB);
    hipLaunchKernelGGL(gae_kernel, grid, block, 0, (hipStream_t)stream, T, n, rewards, values, dones,
                       gamma, gamma * lambda, adv, ret, undiscounted);
    return check_launch("gae_kernel");
}

extern "C" int rl_discount_cumsum(int T, int n, const float* x, const uint8_t* dones, double discount,
                                  float* y, void* stream) {
    if (T <= 0 || n <= 0 || !x || !y) return set_error(RL_ERR_ARG, "rl_discount_cumsum: bad argument");
    dim3 grid((n + SCAN_EW - 1) / SCAN_EW), block(SCAN_EW, SCAN_KB);
    hipLaunchKernelGGL(discount_cumsum_kernel, grid, block, 0, (hipStream_t)stream, T, n, x, dones,
                       discount, y);
    return check_launch("discount_cumsum_kernel");
}
