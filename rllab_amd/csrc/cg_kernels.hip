// cg_kernels.hip -- the vector algebra of one conjugate-gradient iteration as ONE launch.
//
// Replaces the NumPy body of krylov.cg (rllab/misc/krylov.py:7-39) as called from
// ConjugateGradientOptimizer.optimize (rllab/optimizers/conjugate_gradient_optimizer.py:253-256):
//     z = f_Ax(p);  v = rdotr / p.z;  x += v p;  r -= v z;  newrdotr = r.r;
//     mu = newrdotr / rdotr;  p = r + mu p;  rdotr = newrdotr;  if rdotr < tol: break
// with f_Ax(p) = F p + reg_coeff * p, where F p (the Fisher-vector product summed over all
// ranks) is produced by rl_policy_fvp.  Everything is float64, as in the reference; the next
// search direction is also emitted in float32 because that is what rl_policy_fvp consumes.
// The early exit is a device-side `active` flag: once the residual test fires, x / r / p are
// frozen, so the host never has to look at rdotr inside the loop (no synchronisation).
// P <= a few thousand: one workgroup, each thread owns a fixed strided set of elements, dot
// products are reduced through LDS in a fixed order (deterministic, identical on all ranks).
#include <hip/hip_runtime.h>
#include "../../include/rllab_amd.h"
#include "capi_util.h"
#include "cg_device.h"

namespace rl {

// scal[0] = rdotr, scal[1] = active (1 / 0), scal[2] = p.Ap of the last step, scal[3] = steps taken
__global__ void __launch_bounds__(CG_THREADS) cg_init_kernel(int n, const double* __restrict__ b,
                                                             double* __restrict__ x, double* __restrict__ r,
                                                             double* __restrict__ p, float* __restrict__ p32,
                                                             double* __restrict__ scal) {
    __shared__ double scratch[CG_THREADS / 64];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += CG_THREADS) {
        const double bi = b[i];
        x[i] = 0.0; r[i] = bi; p[i] = bi; p32[i] = (float)bi;
        acc += bi * bi;
    }
    const double rdotr = block_sum(acc, scratch);
    if (threadIdx.x == 0) { scal[0] = rdotr; scal[1] = 1.0; scal[2] = 0.0; scal[3] = 0.0; }
}

__global__ void __launch_bounds__(CG_THREADS) cg_step_kernel(int n, const double* __restrict__ fp, double reg,
                                                             double tol, double* __restrict__ x,
                                                             double* __restrict__ r, double* __restrict__ p,
                                                             float* __restrict__ p32, double* __restrict__ scal) {
    __shared__ double scratch[CG_THREADS / 64];
    cg_step_body(n, fp, reg, tol, x, r, p, p32, scal, scratch);
}

__global__ void __launch_bounds__(CG_THREADS) cg_step_large_kernel(int n, const double* __restrict__ fp, double reg,
                                                                   double tol, double* __restrict__ x,
                                                                   double* __restrict__ r, double* __restrict__ p,
                                                                   float* __restrict__ p32,
                                                                   double* __restrict__ scal) {
    __shared__ double scratch[CG_THREADS / 64];
    cg_step_body_large(n, fp, reg, tol, x, r, p, p32, scal, scratch);
}

// trpo_step_kernel for n beyond the register-cached form
__global__ void __launch_bounds__(CG_THREADS) trpo_step_large_kernel(int n, const double* __restrict__ x,
                                                                     const double* __restrict__ a,
                                                                     const double* __restrict__ b, double reg,
                                                                     double delta, double* __restrict__ step,
                                                                     double* __restrict__ out) {
    __shared__ double scratch[CG_THREADS / 64];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += CG_THREADS) {
        const double xv = x[i];
        const double hx = (b ? a[i] - b[i] : a[i]) + reg * xv;
        acc += xv * hx;
    }
    const double xHx = block_sum(acc, scratch);
    double beta = sqrt(2.0 * delta * (1.0 / (xHx + 1e-8)));
    if (beta != beta) beta = 1.0;
    for (int i = threadIdx.x; i < n; i += CG_THREADS) step[i] = beta * x[i];
    if (threadIdx.x == 0) { out[0] = xHx; out[1] = beta; }
}

// After CG (conjugate_gradient_optimizer.py:257-262):  xHx = x . (a - b + reg x),
// beta = sqrt(2 delta * (1 / (xHx + 1e-8)))  (NaN -> 1),  step = beta x.   out = {xHx, beta}
//   a = F x (rl_policy_fvp), b = null, reg = reg_coeff : H x evaluated afresh, as the reference does;
//   a = g, b = CG's residual r, reg = 0                : the same H x from CG's invariant r = g - H x
//                                                        (H = F + reg I is what rl_cg_step iterates on).
__global__ void __launch_bounds__(CG_THREADS) trpo_step_kernel(int n, const double* __restrict__ x,
                                                               const double* __restrict__ a,
                                                               const double* __restrict__ b, double reg,
                                                               double delta, double* __restrict__ step,
                                                               double* __restrict__ out) {
    __shared__ double scratch[CG_THREADS / 64];
    double xv[CG_MAX_PER_THREAD];
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < CG_MAX_PER_THREAD; ++k) {
        const int i = threadIdx.x + k * CG_THREADS;
        if (i < n) {
            xv[k] = x[i];
            const double hx = (b ? a[i] - b[i] : a[i]) + reg * xv[k];
            acc += xv[k] * hx;
        }
    }
    const double xHx = block_sum(acc, scratch);
    double beta = sqrt(2.0 * delta * (1.0 / (xHx + 1e-8)));
    if (beta != beta) beta = 1.0;
#pragma unroll
    for (int k = 0; k < CG_MAX_PER_THREAD; ++k) {
        const int i = threadIdx.x + k * CG_THREADS;
        if (i < n) step[i] = beta * xv[k];
    }
    if (threadIdx.x == 0) { out[0] = xHx; out[1] = beta; }
}

// one candidate of the backtracking line search (:266-274): theta = (float)(prev - ratio * step)
__global__ void __launch_bounds__(256) line_search_point_kernel(int n, const float* __restrict__ prev,
                                                                const double* __restrict__ step, double ratio,
                                                                float* __restrict__ theta) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) theta[i] = (float)((double)prev[i] - ratio * step[i]);
}

// The accept test of one line-search candidate (conjugate_gradient_optimizer.py:262-274) and, when the search goes on,
// the next candidate's parameters -- one workgroup, so the decision and the flag it reads are ordered without atomics.
// state[0] = accepted (0 / 1), state[1] = accepted candidate, state[2 + 4 k ..] = candidate k's folded sums.
__global__ void __launch_bounds__(CG_THREADS) line_search_decide_kernel(
        int rows, const double* __restrict__ sums, const double* __restrict__ before, double inv_count, double delta,
        int candidate, double* __restrict__ state, int32_t* __restrict__ gate, int n, const float* __restrict__ prev,
        const double* __restrict__ step, double next_ratio, float* __restrict__ theta) {
    __shared__ int go_on;
    if (threadIdx.x == 0) {
        bool accepted = state[0] != 0.0;
        if (!accepted) {
            // rows are added in rank order, then scaled: the host path's  h[:, :3].sum(axis=0) * inv
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, mx = -INFINITY, b0 = 0.0;
            for (int r = 0; r < rows; ++r) {
                s0 += sums[4 * r]; s1 += sums[4 * r + 1]; s2 += sums[4 * r + 2];
                mx = fmax(mx, sums[4 * r + 3]);            // (fmax drops a NaN like numpy's max does not -- the max is only logged)
                b0 += before[4 * r];
            }
            double* rec = state + 2 + 4 * candidate;
            rec[0] = s0; rec[1] = s1; rec[2] = s2; rec[3] = mx;
            const double loss = -(s0 * inv_count), loss_before = -(b0 * inv_count), kl = s1 * inv_count;
            if (loss < loss_before && kl <= delta) {       // NaN: false, as in the reference's comparison
                accepted = true;
                state[0] = 1.0;
                state[1] = (double)candidate;
                *gate = 1;
            }
        }
        go_on = accepted ? 0 : 1;
    }
    __syncthreads();
    if (go_on && next_ratio > 0.0)
        for (int i = threadIdx.x; i < n; i += CG_THREADS) theta[i] = (float)((double)prev[i] - next_ratio * step[i]);
}

// One Adam step in Lasagne's form (lasagne.updates.adam, used by FirstOrderOptimizer, first_order_optimizer.py:21-22):
//   m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  theta = (float)(theta - a_t m / (sqrt(v) + eps)),
//   a_t = lr sqrt(1 - b2^t) / (1 - b1^t) computed by the caller.  float64 arithmetic on the float32 parameters.
__global__ void __launch_bounds__(256) adam_step_kernel(int n, float* __restrict__ theta, const double* __restrict__ g,
                                                        double* __restrict__ m, double* __restrict__ v, double a_t,
                                                        double b1, double b2, double eps) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double gi = g[i];
    const double mi = b1 * m[i] + (1.0 - b1) * gi;
    const double vi = b2 * v[i] + (1.0 - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    theta[i] = (float)((double)theta[i] - a_t * mi / (sqrt(vi) + eps));
}

}  // namespace rl

using namespace rl;

extern "C" int rl_adam_step(int n, float* theta, const double* grad, double* m, double* v, double a_t, double beta1,
                            double beta2, double epsilon, void* stream) {
    if (n <= 0 || !theta || !grad || !m || !v) return set_error(RL_ERR_ARG, "rl_adam_step: bad argument");
    hipLaunchKernelGGL(adam_step_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, theta, grad, m, v,
                       a_t, beta1, beta2, epsilon);
    return check_launch("adam_step_kernel");
}

extern "C" int rl_trpo_step(int n, const double* x, const double* a, const double* b, double reg_coeff,
                            double max_constraint, double* step, double* out, void* stream) {
    if (n <= 0 || n > CG_MAX_N || !x || !a || !step || !out)
        return set_error(RL_ERR_ARG, "rl_trpo_step: bad argument (n = %d, max %d)", n, CG_MAX_N);
    if (n <= CG_THREADS * CG_MAX_PER_THREAD)
        hipLaunchKernelGGL(trpo_step_kernel, dim3(1), dim3(CG_THREADS), 0, (hipStream_t)stream, n, x, a, b, reg_coeff,
                           max_constraint, step, out);
    else
        hipLaunchKernelGGL(trpo_step_large_kernel, dim3(1), dim3(CG_THREADS), 0, (hipStream_t)stream, n, x, a, b,
                           reg_coeff, max_constraint, step, out);
    return check_launch("trpo_step_kernel");
}

extern "C" int rl_line_search_point(int n, const float* prev, const double* step, double ratio, float* theta,
                                    void* stream) {
    if (n <= 0 || !prev || !step || !theta) return set_error(RL_ERR_ARG, "rl_line_search_point: bad argument");
    hipLaunchKernelGGL(line_search_point_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, prev,
                       step, ratio, theta);
    return check_launch("line_search_point_kernel");
}

extern "C" int rl_line_search_decide(int rows, const double* sums, const double* before, double inv_count,
                                     double max_constraint, int candidate, double* state, int32_t* gate, int n,
                                     const float* prev, const double* step, double next_ratio, float* theta,
                                     void* stream) {
    if (rows <= 0 || !sums || !before || !state || !gate || candidate < 0 ||
        (next_ratio > 0.0 && (n <= 0 || !prev || !step || !theta)))
        return set_error(RL_ERR_ARG, "rl_line_search_decide: bad argument");
    hipLaunchKernelGGL(line_search_decide_kernel, dim3(1), dim3(CG_THREADS), 0, (hipStream_t)stream, rows, sums, before,
                       inv_count, max_constraint, candidate, state, gate, n, prev, step, next_ratio, theta);
    return check_launch("line_search_decide_kernel");
}

extern "C" int rl_cg_init(int n, const double* b, double* x, double* r, double* p, float* p32, double* scal,
                          void* stream) {
    if (n <= 0 || n > CG_MAX_N || !b || !x || !r || !p || !p32 || !scal)
        return set_error(RL_ERR_ARG, "rl_cg_init: bad argument (n = %d, max %d)", n, CG_MAX_N);
    hipLaunchKernelGGL(cg_init_kernel, dim3(1), dim3(CG_THREADS), 0, (hipStream_t)stream, n, b, x, r, p, p32, scal);
    return check_launch("cg_init_kernel");
}

extern "C" int rl_cg_step(int n, const double* fvp, double reg_coeff, double residual_tol, double* x, double* r,
                          double* p, float* p32, double* scal, void* stream) {
    if (n <= 0 || n > CG_MAX_N || !fvp || !x || !r || !p || !p32 || !scal)
        return set_error(RL_ERR_ARG, "rl_cg_step: bad argument (n = %d, max %d)", n, CG_MAX_N);
    if (n <= CG_THREADS * CG_MAX_PER_THREAD)
        hipLaunchKernelGGL(cg_step_kernel, dim3(1), dim3(CG_THREADS), 0, (hipStream_t)stream, n, fvp, reg_coeff,
                           residual_tol, x, r, p, p32, scal);
    else
        hipLaunchKernelGGL(cg_step_large_kernel, dim3(1), dim3(CG_THREADS), 0, (hipStream_t)stream, n, fvp, reg_coeff,
                           residual_tol, x, r, p, p32, scal);
    return check_launch("cg_step_kernel");
}
