// gaussian_head_kernels.hip -- the diagonal-Gaussian head of the policy objectives on PLANES, for policies whose mean
// and log-std both come from networks (GaussianMLPPolicy(adaptive_std=True) / std_network=...,
// rllab/policies/gaussian_mlp_policy.py:60-98): the networks run as rl_mlp_forward / rl_mlp_backward
// (policy_kernels.hip), this file is what sits between them.
//
//   rl_gaussian_head    : from mean / log_std planes of the current parameters and the recorded batch,
//                           out4 = [ sum w lr adv, sum w KL(old || new), sum w logp adv, max KL ]
//                         (rllab/algos/npo.py:72-82 with rllab/distributions/diagonal_gaussian.py:14-69, as the fused
//                         head of policy_pass_kernel) and the cotangents of
//                           (-sum w {lr | logp} adv + kl_penalty sum w KL) * inv_count
//                         on the two planes:   g_mean = c z / sigma + p dKL/dmu,   g_lstd = c (z^2 - 1) + p dKL/dls,
//                         c = -w adv {lr | 1} inv_count, p = kl_penalty w inv_count; g_lstd = 0 where the floor
//                         log(min_std) is active (the floor's derivative).
//   rl_gaussian_fisher  : the Fisher metric of the mean KL at old == new in (mean, log_std) coordinates is diagonal:
//                           g_mean = w inv_count dmean 2 / (2 v + 1e-8),  g_lstd = w inv_count dlstd 4 v (2 v - e) / (2 v + e)^2,
//                         v = sigma^2 (the same two factors policy_pass_kernel<FVP> applies to its one log_std row).
// HBM-bound: ~ (6 DA + 2) floats in, 2 DA floats out per sample.
#include <hip/hip_runtime.h>
#include "../../include/rllab_amd.h"
#include "capi_util.h"

namespace rl {

int launch_reduce_loss(const double* partial, int rows, double* out, hipStream_t st);   // policy_kernels.hip

constexpr int HEAD_THREADS = 256;
constexpr int HEAD_MAX_DA = 8;
constexpr int HEAD_MAX_GRID = 1024;

struct HeadArgs {
    size_t B;
    int DA, vpg;
    const float* mean;
    const float* lstd;        // raw network output; the floor is applied here
    const float* act;
    const float* adv;
    const float* old_mean;
    const float* old_lstd;    // [DA][B] planes
    const float* w;
    float inv_count, log_min_std, kl_penalty;
    float* g_mean;            // null = sums only
    float* g_lstd;
    double* partial;          // [grid][4]
};

__device__ __forceinline__ double block_sum256(double v, double* scratch) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    return (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
}

__global__ void __launch_bounds__(HEAD_THREADS) gaussian_head_kernel(HeadArgs a) {
    __shared__ double scratch[4];
    __shared__ float smax[4];
    double s_loss = 0.0, s_kl = 0.0, s_vpg = 0.0;
    float max_kl = -INFINITY;
    const size_t B = a.B;
    for (size_t b = (size_t)blockIdx.x * HEAD_THREADS + threadIdx.x; b < B; b += (size_t)gridDim.x * HEAD_THREADS) {
        const float wgt = a.w[b], advb = a.adv[b];
        float zz_new = 0.0f, zz_old = 0.0f, sls_new = 0.0f, sls_old = 0.0f, kl = 0.0f;
        float z[HEAD_MAX_DA], isd[HEAD_MAX_DA], var[HEAD_MAX_DA], dm[HEAD_MAX_DA], num[HEAD_MAX_DA];
        bool fl[HEAD_MAX_DA];
#pragma unroll
        for (int k = 0; k < HEAD_MAX_DA; ++k)
            if (k < a.DA) {
                const size_t i = (size_t)k * B + b;
                const float raw = a.lstd[i];
                fl[k] = raw < a.log_min_std;
                const float ls = fmaxf(raw, a.log_min_std);
                isd[k] = __expf(-ls);
                var[k] = __expf(2.0f * ls);
                const float ak = a.act[i], mo = a.old_mean[i], lo = a.old_lstd[i], so = __expf(lo);
                z[k] = (ak - a.mean[i]) * isd[k];
                const float zo = (ak - mo) / so;
                zz_new = __builtin_fmaf(z[k], z[k], zz_new);
                zz_old = __builtin_fmaf(zo, zo, zz_old);
                sls_new += ls;
                sls_old += lo;
                dm[k] = mo - a.mean[i];
                num[k] = dm[k] * dm[k] + so * so - var[k];
                kl += num[k] / (2.0f * var[k] + 1e-8f) + ls - lo;
            }
        const float logp_new = -sls_new - 0.5f * zz_new;
        const float lr = __expf(logp_new - (-sls_old - 0.5f * zz_old));
        s_loss += (double)(wgt * lr * advb);
        s_kl += (double)(wgt * kl);
        s_vpg += (double)(wgt * (logp_new - 0.5f * (float)a.DA * 1.8378770664093453f) * advb);
        if (wgt > 0.0f) max_kl = fmaxf(max_kl, kl);
        if (a.g_mean) {
            const float c = -wgt * advb * (a.vpg ? 1.0f : lr) * a.inv_count;
            const float p = a.kl_penalty * wgt * a.inv_count;
#pragma unroll
            for (int k = 0; k < HEAD_MAX_DA; ++k)
                if (k < a.DA) {
                    const size_t i = (size_t)k * B + b;
                    const float den = 2.0f * var[k] + 1e-8f;
                    float gm = c * z[k] * isd[k];
                    float gl = c * (z[k] * z[k] - 1.0f);
                    if (a.kl_penalty != 0.0f) {
                        gm = __builtin_fmaf(p, -2.0f * dm[k] / den, gm);
                        gl = __builtin_fmaf(p, 1.0f - (2.0f * var[k] * den + 4.0f * var[k] * num[k]) / (den * den), gl);
                    }
                    a.g_mean[i] = gm;
                    a.g_lstd[i] = fl[k] ? 0.0f : gl;
                }
        }
    }
    const double l = block_sum256(s_loss, scratch), k = block_sum256(s_kl, scratch), v = block_sum256(s_vpg, scratch);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) max_kl = fmaxf(max_kl, __shfl_xor(max_kl, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = max_kl;
    __syncthreads();
    if (threadIdx.x == 0) {
        double* o = a.partial + (size_t)blockIdx.x * 4;
        o[0] = l; o[1] = k; o[2] = v;
        o[3] = (double)fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
    }
}

__global__ void __launch_bounds__(HEAD_THREADS) gaussian_fisher_kernel(size_t B, int DA, const float* __restrict__ dmean,
                                                                       const float* __restrict__ dlstd,
                                                                       const float* __restrict__ lstd,
                                                                       const float* __restrict__ w, float inv_count,
                                                                       float log_min_std, float* __restrict__ g_mean,
                                                                       float* __restrict__ g_lstd) {
    for (size_t b = (size_t)blockIdx.x * HEAD_THREADS + threadIdx.x; b < B; b += (size_t)gridDim.x * HEAD_THREADS) {
        const float c = w[b] * inv_count;
        for (int k = 0; k < DA; ++k) {
            const size_t i = (size_t)k * B + b;
            const float raw = lstd[i];
            const float v = __expf(2.0f * fmaxf(raw, log_min_std)), e = 1e-8f;
            g_mean[i] = c * dmean[i] * (2.0f / (2.0f * v + e));
            g_lstd[i] = raw < log_min_std ? 0.0f : c * dlstd[i] * (4.0f * v * (2.0f * v - e) / ((2.0f * v + e) * (2.0f * v + e)));
        }
    }
}

}  // namespace rl

using namespace rl;

extern "C" size_t rl_gaussian_head_workspace_bytes(void) { return (size_t)HEAD_MAX_GRID * 4 * sizeof(double); }

extern "C" int rl_gaussian_head(size_t n_samples, int act_dim, const float* mean, const float* log_std, const float* actions,
                                const float* advantages, const float* old_means, const float* old_log_stds,
                                const float* weights, float inv_count, float log_min_std, int vpg, float kl_penalty,
                                float* g_mean, float* g_log_std, void* workspace, size_t workspace_bytes, double* out4,
                                void* stream) {
    if (n_samples == 0 || act_dim < 1 || act_dim > HEAD_MAX_DA || !mean || !log_std || !actions || !advantages ||
        !old_means || !old_log_stds || !weights || !out4 || !workspace || ((g_mean == nullptr) != (g_log_std == nullptr)))
        return set_error(RL_ERR_ARG, "rl_gaussian_head: bad argument");
    if (workspace_bytes < rl_gaussian_head_workspace_bytes())
        return set_error(RL_ERR_ARG, "rl_gaussian_head: workspace too small");
    HeadArgs a;
    a.B = n_samples; a.DA = act_dim; a.vpg = vpg; a.mean = mean; a.lstd = log_std; a.act = actions; a.adv = advantages;
    a.old_mean = old_means; a.old_lstd = old_log_stds; a.w = weights; a.inv_count = inv_count;
    a.log_min_std = log_min_std; a.kl_penalty = kl_penalty; a.g_mean = g_mean; a.g_lstd = g_log_std;
    a.partial = (double*)workspace;
    size_t blocks = (n_samples + HEAD_THREADS - 1) / HEAD_THREADS;
    const int grid = (int)(blocks < (size_t)HEAD_MAX_GRID ? blocks : (size_t)HEAD_MAX_GRID);
    hipLaunchKernelGGL(gaussian_head_kernel, dim3(grid), dim3(HEAD_THREADS), 0, (hipStream_t)stream, a);
    int rc = check_launch("gaussian_head_kernel");
    if (rc) return rc;
    return launch_reduce_loss(a.partial, grid, out4, (hipStream_t)stream);
}

extern "C" int rl_gaussian_fisher(size_t n_samples, int act_dim, const float* dmean, const float* dlog_std,
                                  const float* log_std, const float* weights, float inv_count, float log_min_std,
                                  float* g_mean, float* g_log_std, void* stream) {
    if (n_samples == 0 || act_dim < 1 || act_dim > HEAD_MAX_DA || !dmean || !dlog_std || !log_std || !weights || !g_mean ||
        !g_log_std)
        return set_error(RL_ERR_ARG, "rl_gaussian_fisher: bad argument");
    size_t blocks = (n_samples + HEAD_THREADS - 1) / HEAD_THREADS;
    const int grid = (int)(blocks < (size_t)HEAD_MAX_GRID ? blocks : (size_t)HEAD_MAX_GRID);
    hipLaunchKernelGGL(gaussian_fisher_kernel, dim3(grid), dim3(HEAD_THREADS), 0, (hipStream_t)stream, n_samples, act_dim,
                       dmean, dlog_std, log_std, weights, inv_count, log_min_std, g_mean, g_log_std);
    return check_launch("gaussian_fisher_kernel");
}
