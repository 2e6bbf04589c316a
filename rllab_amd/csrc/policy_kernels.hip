// policy_kernels.hip -- fused TRPO/VPG update kernels for GaussianMLPPolicy
// (tanh MLP mean with two hidden layers + state-independent log_std).
//
// One pass over the dense batch per launch; what the reference evaluates as the
// compiled Theano functions f_loss / f_constraint / f_loss_constraint, f_grad and
// f_Hx_plain (rllab/optimizers/conjugate_gradient_optimizer.py:27-46,194-215) on the
// surrogate loss and mean KL of rllab/algos/npo.py:72-82 with
// rllab/distributions/diagonal_gaussian.py:14-69:
//   MODE_LOSS : sum_b w_b * lr_b * adv_b ,  sum_b w_b * KL_b ,  sum_b w_b * logp_b * adv_b ,
//               max_b KL_b                                (f_loss_constraint, vpg f_kl)
//   MODE_GRAD : d/dtheta of  -sum_b w_b lr_b adv_b * inv_count          (f_grad)
//   MODE_VPG  : d/dtheta of  -sum_b w_b logp_b adv_b * inv_count        (vpg.py:91)
//   MODE_FVP  : Fisher-vector product  F v = J^T diag(2/(2 sigma^2+1e-8)) J v * inv_count
//               (+ the log_std block), which equals the Hessian of the mean KL at
//               theta_new == theta_old -- the only point where TRPO evaluates it
//               (PerlmutterHvp, :27-55; reg_coeff * v is added by the caller).
//
// Mapping.  A workgroup is WAVES wavefronts; a wavefront owns tiles of 64 samples,
// lane <-> sample for the per-sample network passes (weights broadcast from LDS,
// activations in registers / per-lane LDS columns).  The parameter gradient is a
// batch reduction of outer products (gW1 = sum_b h0_b (x) gz1_b ...): for that phase
// the roles flip, lane <-> (column, row-group) of the weight matrix, and each lane
// walks the 64 samples of the tile reading activations from LDS (row stride 65 floats:
// conflict-free in both roles).  Per-lane accumulators live in registers across the
// whole grid-stride loop; every wavefront then writes ONE partial gradient row, and a
// second kernel sums the partial rows in float64 in a fixed order (deterministic).
//
// All kernels are VALU-bound by design (arithmetic intensity ~ 6*fwd_flops / 80 B
// >> the 20 flop/B ridge); HBM sees each sample once per pass.
#include <hip/hip_runtime.h>
#include "../../include/rllab_amd.h"
#include "capi_util.h"

namespace rl {

constexpr int WV = 64;   // wavefront
constexpr int LS = 65;   // LDS row stride (floats) of activation tiles

enum { MODE_LOSS = 0, MODE_GRAD = 1, MODE_FVP = 2, MODE_VPG = 3 };
constexpr int LOSS_COLS = 4;  // sum w*lr*adv, sum w*kl, sum w*logp*adv, max kl

__device__ __forceinline__ float ftanh(float x) {
    float xc = fminf(fmaxf(x, -10.0f), 10.0f);
    float e = __expf(2.0f * xc);
    return (e - 1.0f) * __builtin_amdgcn_rcpf(e + 1.0f);
}

template <int DO_, int DA_, int H0_, int H1_>
struct Net {
    static constexpr int DO = DO_, DA = DA_, H0 = H0_, H1 = H1_;
    static constexpr int W0 = 0;
    static constexpr int B0 = W0 + DO * H0;
    static constexpr int W1 = B0 + H0;
    static constexpr int B1 = W1 + H0 * H1;
    static constexpr int W2 = B1 + H1;
    static constexpr int B2 = W2 + H1 * DA;
    static constexpr int LSTD = B2 + DA;
    static constexpr int P = LSTD + DA;
    static constexpr int PP = (P + 3) & ~3;
    // outer-product ownership: lane owns column (lane % H), row group (lane / H)
    static constexpr int NG1 = WV / H1, R1 = H0 / NG1;               // gW1: R1 rows per lane
    static constexpr int NG0 = WV / H0, R0 = (DO + NG0 - 1) / NG0;   // gW0: R0 rows per lane
    static constexpr int DOP = R0 * NG0;                             // padded obs rows (zeros)
    static constexpr int E2 = (H1 * DA + WV - 1) / WV;               // gW2 entries per lane
    static constexpr int HMAX = (H0 > H1 ? H0 : H1) > DOP ? (H0 > H1 ? H0 : H1) : DOP;
    static constexpr int TILE = HMAX * LS;
    static_assert(WV % H0 == 0 && WV % H1 == 0 && H0 % NG1 == 0, "hidden sizes must divide 64");
};

// y[j] += sum_d W[d][j] * x[d]; W row-major [IN][OUT] in LDS (broadcast reads),
// x in this lane's LDS column (stride LS).  The d loop stays rolled (see env_kernels.hip).
template <int IN, int OUT>
__device__ __forceinline__ void dense_acc(const float* __restrict__ w, const float* __restrict__ xcol,
                                          float* y) {
#pragma unroll 2
    for (int d = 0; d < IN; ++d) {
        const float xd = xcol[d * LS];
        const float* __restrict__ row = w + d * OUT;
#pragma unroll
        for (int j = 0; j < OUT; ++j) y[j] = __builtin_fmaf(xd, row[j], y[j]);
    }
}

template <int N>
__device__ __forceinline__ void store_col(float* col, const float* v) {
#pragma unroll
    for (int j = 0; j < N; ++j) col[j * LS] = v[j];
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WV);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WV);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, WV));
    return v;
}

struct PolicyBatch {
    int B;                     // samples
    const float* theta;        // [P]
    const float* vec;          // [P] tangent (MODE_FVP) or null
    const float* obs;          // [DO][B]
    const float* act;          // [DA][B]
    const float* adv;          // [B]
    const float* old_mean;     // [DA][B]
    const float* old_log_std;  // [DA]
    const float* weight;       // [B] 0/1
    float inv_count;
    float log_min_std;
    float* partial;            // [waves_total][P]   (grad-like modes)
    double* partial_loss;      // [waves_total][LOSS_COLS] (MODE_LOSS)
};

template <class N, int MODE, int WAVES>
__global__ void __launch_bounds__(WAVES* WV) policy_pass_kernel(PolicyBatch a) {
    constexpr int DO = N::DO, DA = N::DA, H0 = N::H0, H1 = N::H1, P = N::P;
    constexpr bool GRADLIKE = (MODE != MODE_LOSS);
    constexpr int R1 = N::R1, R0 = N::R0, E2 = N::E2;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sw = smem;                                              // [P] weights
    float* sv = sw + N::PP;                                        // [P] tangent (FVP only)
    float* sw1t = sv + ((MODE == MODE_FVP) ? N::PP : 0);           // [H1][H0] = W1^T (grad-like)
    float* tiles = sw1t + (GRADLIKE ? H0 * H1 : 0);
    const int wave = threadIdx.x / WV, lane = threadIdx.x % WV;
    float* bufA = tiles + wave * (2 * N::TILE + DA * LS);
    float* bufB = bufA + N::TILE;
    float* gmu_t = bufB + N::TILE;                                 // [DA][LS]

    for (int k = threadIdx.x; k < P; k += WAVES * WV) {
        sw[k] = a.theta[k];
        if (MODE == MODE_FVP) sv[k] = a.vec[k];
    }
    if (GRADLIKE) {
        for (int k = threadIdx.x; k < H0 * H1; k += WAVES * WV) {
            const int i = k / H1, j = k % H1;                      // W1[i][j]
            sw1t[j * H0 + i] = a.theta[N::W1 + k];
        }
    }
    __syncthreads();

    // effective log_std / std (state independent)
    float lstd[DA], inv_var[DA], inv_std[DA], var_[DA];
    bool floored[DA];
#pragma unroll
    for (int k = 0; k < DA; ++k) {
        const float raw = sw[N::LSTD + k];
        floored[k] = raw < a.log_min_std;
        lstd[k] = fmaxf(raw, a.log_min_std);
        inv_std[k] = __expf(-lstd[k]);
        inv_var[k] = inv_std[k] * inv_std[k];
        var_[k] = __expf(2.0f * lstd[k]);
    }

    // accumulators (registers, whole launch)
    double acc_loss = 0.0, acc_kl = 0.0, acc_vpg = 0.0;
    float max_kl = -INFINITY;
    float gW1[R1], gW0[R0], gW2[E2];
    float gb0 = 0.0f, gb1 = 0.0f, gb2 = 0.0f, gls[DA];
    float wsum = 0.0f;
#pragma unroll
    for (int r = 0; r < R1; ++r) gW1[r] = 0.0f;
#pragma unroll
    for (int r = 0; r < R0; ++r) gW0[r] = 0.0f;
#pragma unroll
    for (int r = 0; r < E2; ++r) gW2[r] = 0.0f;
#pragma unroll
    for (int k = 0; k < DA; ++k) gls[k] = 0.0f;

    const int B = a.B;
    const int n_tiles = (B + WV - 1) / WV;
    const int wave_global = blockIdx.x * WAVES + wave;
    const int waves_total = gridDim.x * WAVES;
    float* colA = bufA + lane;
    float* colB = bufB + lane;
    const int j1 = lane % H1, g1 = lane / H1;   // owned column / row group of gW1
    const int j0 = lane % H0, g0 = lane / H0;   // owned column / row group of gW0

    for (int tile = wave_global; tile < n_tiles; tile += waves_total) {
        const int b = tile * WV + lane;
        const bool live = b < B;
        const int bi = live ? b : (B - 1);
        const float wgt = live ? a.weight[bi] : 0.0f;
        float x[N::DOP];
#pragma unroll
        for (int d = 0; d < N::DOP; ++d) x[d] = (d < DO) ? a.obs[(size_t)d * B + bi] : 0.0f;

        // ---- forward ------------------------------------------------------------
        store_col<N::DOP>(colA, x);                            // bufA = x (zero padded)
        float h0[H0];
#pragma unroll
        for (int j = 0; j < H0; ++j) h0[j] = sw[N::B0 + j];
        dense_acc<DO, H0>(sw + N::W0, colA, h0);
#pragma unroll
        for (int j = 0; j < H0; ++j) h0[j] = ftanh(h0[j]);
        store_col<H0>(colB, h0);                               // bufB = h0
        float h1[H1];
#pragma unroll
        for (int j = 0; j < H1; ++j) h1[j] = sw[N::B1 + j];
        dense_acc<H0, H1>(sw + N::W1, colB, h1);
#pragma unroll
        for (int j = 0; j < H1; ++j) h1[j] = ftanh(h1[j]);
        float mean[DA];
#pragma unroll
        for (int k = 0; k < DA; ++k) mean[k] = sw[N::B2 + k];
#pragma unroll
        for (int i = 0; i < H1; ++i)
#pragma unroll
            for (int k = 0; k < DA; ++k) mean[k] = __builtin_fmaf(h1[i], sw[N::W2 + i * DA + k], mean[k]);

        // ---- per-sample scalars -> cotangent on the mean --------------------------
        float gmu[DA];
#pragma unroll
        for (int k = 0; k < DA; ++k) gmu[k] = 0.0f;
        if (MODE != MODE_FVP) {
            const float advb = a.adv[bi];
            float zz_new = 0.0f, zz_old = 0.0f, sls_new = 0.0f, sls_old = 0.0f, kl = 0.0f;
            float znew[DA];
#pragma unroll
            for (int k = 0; k < DA; ++k) {
                const float ak = a.act[(size_t)k * B + bi];
                const float mo = a.old_mean[(size_t)k * B + bi];
                const float lo = a.old_log_std[k];
                const float so = __expf(lo);
                znew[k] = (ak - mean[k]) * inv_std[k];
                const float zo = (ak - mo) / so;
                zz_new = __builtin_fmaf(znew[k], znew[k], zz_new);
                zz_old = __builtin_fmaf(zo, zo, zz_old);
                sls_new += lstd[k];
                sls_old += lo;
                const float dm = mo - mean[k];
                const float num = dm * dm + so * so - var_[k];
                kl += num / (2.0f * var_[k] + 1e-8f) + lstd[k] - lo;
            }
            // logli_new - logli_old (diagonal_gaussian.py:56-69); the 0.5*Da*log(2 pi) terms cancel
            const float logp_new = -sls_new - 0.5f * zz_new;
            const float dlog = logp_new - (-sls_old - 0.5f * zz_old);
            const float lr = __expf(dlog);
            if (MODE == MODE_LOSS) {
                acc_loss += (double)(wgt * lr * advb);
                acc_kl += (double)(wgt * kl);
                acc_vpg += (double)(wgt * (logp_new - 0.5f * (float)DA * 1.8378770664093453f) * advb);
                if (wgt > 0.0f) max_kl = fmaxf(max_kl, kl);
            } else {
                // d(-w adv lr)/dmu_k = -w adv lr z_k / sigma_k ; VPG: lr -> 1 (d logp)
                const float c = -wgt * advb * (MODE == MODE_GRAD ? lr : 1.0f) * a.inv_count;
#pragma unroll
                for (int k = 0; k < DA; ++k) {
                    gmu[k] = c * znew[k] * inv_std[k];
                    if (!floored[k]) gls[k] += c * (znew[k] * znew[k] - 1.0f);
                }
            }
        } else {
            // tangent forward: dmu = J v
            float d0[H0];
#pragma unroll
            for (int j = 0; j < H0; ++j) d0[j] = sv[N::B0 + j];
            dense_acc<DO, H0>(sv + N::W0, colA, d0);               // dW0^T x + db0   (bufA = x)
#pragma unroll
            for (int j = 0; j < H0; ++j) d0[j] *= (1.0f - h0[j] * h0[j]);   // dh0
            float d1[H1];
#pragma unroll
            for (int j = 0; j < H1; ++j) d1[j] = sv[N::B1 + j];
            dense_acc<H0, H1>(sv + N::W1, colB, d1);               // dW1^T h0        (bufB = h0)
            store_col<H0>(colA, d0);                               // bufA = dh0
            dense_acc<H0, H1>(sw + N::W1, colA, d1);               // + W1^T dh0
#pragma unroll
            for (int j = 0; j < H1; ++j) d1[j] *= (1.0f - h1[j] * h1[j]);   // dh1
            float dmu[DA];
#pragma unroll
            for (int k = 0; k < DA; ++k) dmu[k] = sv[N::B2 + k];
#pragma unroll
            for (int i = 0; i < H1; ++i)
#pragma unroll
                for (int k = 0; k < DA; ++k) {
                    dmu[k] = __builtin_fmaf(h1[i], sv[N::W2 + i * DA + k], dmu[k]);
                    dmu[k] = __builtin_fmaf(d1[i], sw[N::W2 + i * DA + k], dmu[k]);
                }
            const float c = wgt * a.inv_count;
#pragma unroll
            for (int k = 0; k < DA; ++k) gmu[k] = c * dmu[k] * (2.0f / (2.0f * var_[k] + 1e-8f));
            wsum += c;
            // restore bufA = x for the gW0 outer product below
            store_col<N::DOP>(colA, x);
        }

        if (GRADLIKE) {
            // ---- layer 2: gW2 = h1 (x) gmu, gb2 = sum gmu ---------------------------
            // bufA holds x, bufB holds h0.  gW2 needs h1 in a tile: stage it in bufA after
            // saving nothing -- x is still in registers and is re-stored for layer 0.
            store_col<H1>(colA, h1);                               // bufA = h1
#pragma unroll
            for (int k = 0; k < DA; ++k) gmu_t[k * LS + lane] = gmu[k];
#pragma unroll
            for (int r = 0; r < E2; ++r) {
                const int e = r * WV + lane;
                if (e < H1 * DA) {
                    const int i = e / DA, k = e % DA;
                    float acc = 0.0f;
#pragma unroll 4
                    for (int s = 0; s < WV; ++s) acc = __builtin_fmaf(bufA[i * LS + s], gmu_t[k * LS + s], acc);
                    gW2[r] += acc;
                }
            }
            if (lane < DA) {
                float acc = 0.0f;
#pragma unroll 4
                for (int s = 0; s < WV; ++s) acc += gmu_t[lane * LS + s];
                gb2 += acc;
            }
            // ---- layer 1: gz1 = (W2 gmu) * (1 - h1^2); gW1 = h0 (x) gz1 -------------------
            float gz1[H1];
#pragma unroll
            for (int i = 0; i < H1; ++i) {
                float g = 0.0f;
#pragma unroll
                for (int k = 0; k < DA; ++k) g = __builtin_fmaf(sw[N::W2 + i * DA + k], gmu[k], g);
                gz1[i] = g * (1.0f - h1[i] * h1[i]);
            }
            store_col<H1>(colA, gz1);                              // bufA = gz1, bufB = h0
            {
                float gbl = 0.0f;
#pragma unroll 2
                for (int s = 0; s < WV; ++s) {
                    const float bval = bufA[j1 * LS + s];
                    gbl += bval;
#pragma unroll
                    for (int r = 0; r < R1; ++r)
                        gW1[r] = __builtin_fmaf(bufB[(g1 * R1 + r) * LS + s], bval, gW1[r]);
                }
                gb1 += gbl;
            }
            // ---- layer 0: gz0 = (W1 gz1) * (1 - h0^2); gW0 = x (x) gz0 ---------------------
            float gz0[H0];
#pragma unroll
            for (int i = 0; i < H0; ++i) gz0[i] = 0.0f;
            dense_acc<H1, H0>(sw1t, colA, gz0);                    // sum_j W1[i][j] gz1[j]
#pragma unroll
            for (int i = 0; i < H0; ++i) gz0[i] *= (1.0f - h0[i] * h0[i]);
            store_col<H0>(colB, gz0);                              // bufB = gz0
            store_col<N::DOP>(colA, x);                            // bufA = x
            {
                float gbl = 0.0f;
#pragma unroll 2
                for (int s = 0; s < WV; ++s) {
                    const float bval = bufB[j0 * LS + s];
                    gbl += bval;
#pragma unroll
                    for (int r = 0; r < R0; ++r)
                        gW0[r] = __builtin_fmaf(bufA[(g0 * R0 + r) * LS + s], bval, gW0[r]);
                }
                gb0 += gbl;
            }
        }
    }

    // ---- write this wavefront's partial row ------------------------------------------
    if (MODE == MODE_LOSS) {
        const double l = wave_sum(acc_loss), k = wave_sum(acc_kl), v = wave_sum(acc_vpg);
        const float mk = wave_max(max_kl);
        if (lane == 0) {
            double* row = a.partial_loss + (size_t)wave_global * LOSS_COLS;
            row[0] = l; row[1] = k; row[2] = v; row[3] = (double)mk;
        }
    } else {
        float* row = a.partial + (size_t)wave_global * P;
#pragma unroll
        for (int r = 0; r < R0; ++r) {
            const int i = g0 * R0 + r;
            if (i < DO) row[N::W0 + i * H0 + j0] = gW0[r];
        }
        if (g0 == 0) row[N::B0 + j0] = gb0;
#pragma unroll
        for (int r = 0; r < R1; ++r) row[N::W1 + (g1 * R1 + r) * H1 + j1] = gW1[r];
        if (g1 == 0) row[N::B1 + j1] = gb1;
#pragma unroll
        for (int r = 0; r < E2; ++r) {
            const int e = r * WV + lane;
            if (e < H1 * DA) row[N::W2 + e] = gW2[r];
        }
        if (lane < DA) row[N::B2 + lane] = gb2;
        if (MODE == MODE_FVP) {
            // log_std block of the Fisher: d2KL/ds2 = 4 v (2 v - eps) / (2 v + eps)^2, v = sigma^2
            const float ws = wave_sum(wsum);
            if (lane < DA) {
                float c = 0.0f;
#pragma unroll
                for (int k = 0; k < DA; ++k)
                    if (k == lane) {
                        const float vv = var_[k], e = 1e-8f;
                        c = floored[k] ? 0.0f
                                       : 4.0f * vv * (2.0f * vv - e) / ((2.0f * vv + e) * (2.0f * vv + e));
                    }
                row[N::LSTD + lane] = c * sv[N::LSTD + lane] * ws;
            }
        } else {
#pragma unroll
            for (int k = 0; k < DA; ++k) {
                const float s = wave_sum(gls[k]);
                if (lane == 0) row[N::LSTD + k] = s;
            }
        }
    }
}

// out[c] = sum_r partial[r][c] in float64, fixed order (deterministic).
__global__ void reduce_rows_f32_kernel(const float* __restrict__ partial, int rows, int cols,
                                       double* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    double s = 0.0;
    for (int r = 0; r < rows; ++r) s += (double)partial[(size_t)r * cols + c];
    out[c] = s;
}

// loss partials: columns 0..2 summed, column 3 maxed
__global__ void reduce_loss_kernel(const double* __restrict__ partial, int rows, double* __restrict__ out) {
    const int c = threadIdx.x;
    if (c >= LOSS_COLS) return;
    double s = (c == 3) ? -INFINITY : 0.0;
    for (int r = 0; r < rows; ++r) {
        const double v = partial[(size_t)r * LOSS_COLS + c];
        s = (c == 3) ? fmax(s, v) : s + v;
    }
    out[c] = s;
}

template <class N, int MODE, int WAVES>
static size_t pass_lds_bytes() {
    size_t f = N::PP + (MODE == MODE_FVP ? N::PP : 0) + (MODE != MODE_LOSS ? N::H0 * N::H1 : 0) +
               (size_t)WAVES * (2 * N::TILE + N::DA * LS);
    return f * sizeof(float);
}

template <class N, int MODE, int WAVES>
static int launch_pass(const rl_policy_batch* g, const float* vec, void* workspace, size_t workspace_bytes,
                       double* out, hipStream_t st) {
    PolicyBatch a;
    a.B = g->n_samples; a.theta = g->theta; a.vec = vec; a.obs = g->obs; a.act = g->actions; a.adv = g->advantages;
    a.old_mean = g->old_means; a.old_log_std = g->old_log_std; a.weight = g->weights;
    a.inv_count = g->inv_count; a.log_min_std = g->log_min_std;
    const int n_tiles = (a.B + WV - 1) / WV;
    const size_t lds = pass_lds_bytes<N, MODE, WAVES>();
    int blocks_per_cu = (int)((160 * 1024) / lds);
    if (blocks_per_cu < 1) return set_error(RL_ERR_UNSUPPORTED, "policy pass needs %zu B of LDS", lds);
    if (blocks_per_cu > 8) blocks_per_cu = 8;
    int grid = 256 * blocks_per_cu;
    const int need = (n_tiles + WAVES - 1) / WAVES;
    if (grid > need) grid = need;
    const int rows = grid * WAVES;
    const size_t need_bytes = (MODE == MODE_LOSS) ? (size_t)rows * LOSS_COLS * sizeof(double)
                                                  : (size_t)rows * N::P * sizeof(float);
    if (workspace_bytes < need_bytes)
        return set_error(RL_ERR_ARG, "policy pass workspace too small: %zu < %zu bytes", workspace_bytes,
                         need_bytes);
    a.partial = (float*)workspace;
    a.partial_loss = (double*)workspace;
    auto kern = policy_pass_kernel<N, MODE, WAVES>;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * WV), lds, st, a);
    int rc = check_launch("policy_pass_kernel");
    if (rc) return rc;
    if (MODE == MODE_LOSS) {
        hipLaunchKernelGGL(reduce_loss_kernel, dim3(1), dim3(64), 0, st, a.partial_loss, rows, out);
    } else {
        hipLaunchKernelGGL(reduce_rows_f32_kernel, dim3((N::P + 255) / 256), dim3(256), 0, st, a.partial, rows,
                           N::P, out);
    }
    return check_launch("policy reduce kernel");
}

template <class N, int WAVES>
static int dispatch_mode(int mode, const rl_policy_batch* g, const float* vec, void* ws, size_t ws_bytes,
                         double* out, hipStream_t st) {
    switch (mode) {
        case MODE_LOSS: return launch_pass<N, MODE_LOSS, WAVES>(g, vec, ws, ws_bytes, out, st);
        case MODE_GRAD: return launch_pass<N, MODE_GRAD, WAVES>(g, vec, ws, ws_bytes, out, st);
        case MODE_FVP: return launch_pass<N, MODE_FVP, WAVES>(g, vec, ws, ws_bytes, out, st);
        case MODE_VPG: return launch_pass<N, MODE_VPG, WAVES>(g, vec, ws, ws_bytes, out, st);
    }
    return set_error(RL_ERR_ARG, "unknown policy pass mode %d", mode);
}

static int dispatch_net(int mode, const rl_policy_batch* g, const float* vec, void* ws, size_t ws_bytes,
                        double* out, hipStream_t st) {
    const int d = g->obs_dim, k = g->act_dim, h0 = g->hidden0, h1 = g->hidden1;
#define NETCASE(DO, DA, H, WAVES) \
    if (d == DO && k == DA && h0 == H && h1 == H) return dispatch_mode<Net<DO, DA, H, H>, WAVES>(mode, g, vec, ws, ws_bytes, out, st);
    NETCASE(4, 1, 32, 2)    // Cartpole
    NETCASE(6, 1, 32, 2)    // DoublePendulum
    NETCASE(13, 2, 32, 2)   // Swimmer
    NETCASE(20, 6, 32, 2)   // HalfCheetah
    NETCASE(4, 1, 64, 2)
    NETCASE(6, 1, 64, 2)
    NETCASE(13, 2, 64, 2)
    NETCASE(20, 6, 64, 2)
#undef NETCASE
    return set_error(RL_ERR_UNSUPPORTED,
                     "no fused policy kernel for obs_dim=%d act_dim=%d hidden=(%d,%d); the torch autograd "
                     "path handles arbitrary networks", d, k, h0, h1);
}

}  // namespace rl

using namespace rl;

static int check_batch(const rl_policy_batch* g, const char* who) {
    if (!g) return set_error(RL_ERR_ARG, "%s: null batch", who);
    if (g->n_samples <= 0 || !g->theta || !g->obs || !g->weights)
        return set_error(RL_ERR_ARG, "%s: bad batch", who);
    return 0;
}

extern "C" size_t rl_policy_workspace_bytes(int obs_dim, int act_dim, int hidden0, int hidden1) {
    // rows <= 256 CUs * 8 blocks * 2 waves; row = P floats (or LOSS_COLS doubles)
    const size_t P = (size_t)obs_dim * hidden0 + hidden0 + (size_t)hidden0 * hidden1 + hidden1 +
                     (size_t)hidden1 * act_dim + 2 * (size_t)act_dim;
    const size_t rows = 256 * 8 * 2;
    const size_t a = rows * P * sizeof(float), b = rows * LOSS_COLS * sizeof(double);
    return a > b ? a : b;
}

extern "C" int rl_policy_loss_kl(const rl_policy_batch* g, void* workspace, size_t workspace_bytes,
                                 double* out4, void* stream) {
    int rc = check_batch(g, "rl_policy_loss_kl");
    if (rc) return rc;
    if (!g->actions || !g->advantages || !g->old_means || !g->old_log_std || !out4)
        return set_error(RL_ERR_ARG, "rl_policy_loss_kl: bad argument");
    return dispatch_net(MODE_LOSS, g, nullptr, workspace, workspace_bytes, out4, (hipStream_t)stream);
}

extern "C" int rl_policy_grad(const rl_policy_batch* g, int vpg, void* workspace, size_t workspace_bytes,
                              double* grad_out, void* stream) {
    int rc = check_batch(g, "rl_policy_grad");
    if (rc) return rc;
    if (!g->actions || !g->advantages || !g->old_means || !g->old_log_std || !grad_out)
        return set_error(RL_ERR_ARG, "rl_policy_grad: bad argument");
    return dispatch_net(vpg ? MODE_VPG : MODE_GRAD, g, nullptr, workspace, workspace_bytes, grad_out,
                        (hipStream_t)stream);
}

extern "C" int rl_policy_fvp(const rl_policy_batch* g, const float* vec, void* workspace,
                             size_t workspace_bytes, double* fvp_out, void* stream) {
    int rc = check_batch(g, "rl_policy_fvp");
    if (rc) return rc;
    if (!vec || !fvp_out) return set_error(RL_ERR_ARG, "rl_policy_fvp: bad argument");
    return dispatch_net(MODE_FVP, g, vec, workspace, workspace_bytes, fvp_out, (hipStream_t)stream);
}
