// cg_device.h -- device body of one krylov.cg iteration (rllab/misc/krylov.py:7-39), shared by
//   cg_step_kernel        (cg_kernels.hip): rl_cg_step, the Fisher-vector product handed over already summed (ranks);
//   reduce_rows_cg_kernel (policy_kernels.hip): rl_policy_fvp_cg_step, the single-GPU form -- the workgroup that
//                         finishes the row reduction of the Fisher-vector product LAST runs the CG step in the same
//                         launch (one launch less per CG iteration).
#pragma once
#include <hip/hip_runtime.h>

namespace rl {

constexpr int CG_THREADS = 1024;
constexpr int CG_MAX_PER_THREAD = 16;   // n <= 16384 parameters: operands cached in registers between the passes
constexpr int CG_MAX_N = 1 << 16;       // beyond that and up to here: the re-reading form (cg_step_body_large)

__device__ __forceinline__ double block_sum(double v, double* scratch) {
    // wavefront butterfly, then the 16 wave sums in order
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < CG_THREADS / 64; ++w) s += scratch[w];
    return s;
}

// read a double other workgroups of this launch may have written (agent scope: not served from the CU's L1)
__device__ __forceinline__ double ld_agent(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one CG iteration by the CG_THREADS threads of one workgroup; `fp` = F p (summed over ranks), scratch = LDS [16]
__device__ __forceinline__ void cg_step_body(int n, const double* fp, double reg, double tol, double* __restrict__ x,
                                             double* __restrict__ r, double* __restrict__ p,
                                             float* __restrict__ p32, double* __restrict__ scal, double* scratch) {
    const double rdotr = scal[0];
    const bool active = scal[1] != 0.0;
    double z[CG_MAX_PER_THREAD], pv[CG_MAX_PER_THREAD];
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < CG_MAX_PER_THREAD; ++k) {
        const int i = threadIdx.x + k * CG_THREADS;
        if (i < n) {
            pv[k] = p[i];
            z[k] = ld_agent(fp + i) + reg * pv[k];      // Hx = F p + reg_coeff * p
            acc += pv[k] * z[k];
        }
    }
    const double pz = block_sum(acc, scratch);
    if (!active) return;                     // wave-uniform: scal[1] is one value for the whole grid
    const double v = rdotr / pz;
    double rn[CG_MAX_PER_THREAD];
    acc = 0.0;
#pragma unroll
    for (int k = 0; k < CG_MAX_PER_THREAD; ++k) {
        const int i = threadIdx.x + k * CG_THREADS;
        if (i < n) {
            x[i] += v * pv[k];
            rn[k] = r[i] - v * z[k];
            r[i] = rn[k];
            acc += rn[k] * rn[k];
        }
    }
    const double newrdotr = block_sum(acc, scratch);
    const double mu = newrdotr / rdotr;
#pragma unroll
    for (int k = 0; k < CG_MAX_PER_THREAD; ++k) {
        const int i = threadIdx.x + k * CG_THREADS;
        if (i < n) {
            const double pn = rn[k] + mu * pv[k];
            p[i] = pn;
            p32[i] = (float)pn;
        }
    }
    if (threadIdx.x == 0) {
        scal[0] = newrdotr;
        scal[1] = (newrdotr >= tol) ? 1.0 : 0.0;
        scal[2] = pz;
        scal[3] += 1.0;
    }
}


// The same iteration for LARGE parameter vectors (n > CG_THREADS * CG_MAX_PER_THREAD: the 128-unit policy nets of
// policy_wide_kernels.hip have up to ~37 k parameters): nothing is cached in registers, every pass re-reads its
// operands.  Same expressions, same per-thread strided partial sums, same fixed-order fold -- so for a size both
// forms accept they agree bit for bit.
__device__ __forceinline__ void cg_step_body_large(int n, const double* fp, double reg, double tol,
                                                   double* __restrict__ x, double* __restrict__ r,
                                                   double* __restrict__ p, float* __restrict__ p32,
                                                   double* __restrict__ scal, double* scratch) {
    const double rdotr = scal[0];
    const bool active = scal[1] != 0.0;
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += CG_THREADS) {
        const double pv = p[i];
        acc += pv * (ld_agent(fp + i) + reg * pv);
    }
    const double pz = block_sum(acc, scratch);
    if (!active) return;
    const double v = rdotr / pz;
    acc = 0.0;
    for (int i = threadIdx.x; i < n; i += CG_THREADS) {
        const double pv = p[i];
        const double z = ld_agent(fp + i) + reg * pv;
        x[i] += v * pv;
        const double rn = r[i] - v * z;
        r[i] = rn;
        acc += rn * rn;
    }
    const double newrdotr = block_sum(acc, scratch);
    const double mu = newrdotr / rdotr;
    for (int i = threadIdx.x; i < n; i += CG_THREADS) {
        const double pn = r[i] + mu * p[i];
        p[i] = pn;
        p32[i] = (float)pn;
    }
    if (threadIdx.x == 0) {
        scal[0] = newrdotr;
        scal[1] = (newrdotr >= tol) ? 1.0 : 0.0;
        scal[2] = pz;
        scal[3] += 1.0;
    }
}

}  // namespace rl
