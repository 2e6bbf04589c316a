// policy_csplit_kernels.hip -- the Fisher-vector product of the 64-unit, WIDE and DEEP GaussianMLPPolicy mean networks
// (two or three tanh layers of 32 / 64 / 128 units, at least one of them wider than 32) on the bf16 matrix pipe at f32
// accuracy: the split-operand arithmetic of policy_split_kernels.hip (every f32 operand = hi + mid + lo, three bf16
// parts, exactly; six cross terms per product on v_mfma_f32_32x32x16_bf16 with f32 accumulation -- dropped terms
// <= 2^-23 |a b|, tests/test_split_arithmetic.py) in the COOPERATIVE tiling of policy_wide_kernels.hip.
//
// What it computes: rl_policy_fvp on cached activations -- f_Hx_plain of rllab/optimizers/
// conjugate_gradient_optimizer.py:27-55 at theta_new == theta_old, for GaussianMLPPolicy(hidden_sizes=...) of
// rllab/policies/gaussian_mlp_policy.py:21-58 -- the same inputs and partial-row / float64 row reduction as
// policy_pass_kernel<.., MODE_FVP, cached> and wide_pass_kernel<.., WMODE_FVP>, a result that differs from theirs by
// rounding only.  TRPO launches it cg_iters times per update on the activations its gradient pass cached.
//
// Why: the f32-input matrix instruction runs at the f32 VECTOR rate on the vector datapath; at these widths its cycles
// are most of a tile ((20 -> 64 -> 64 -> 6): 17.8 k of 41 k cycles per 32 samples; a (128 x 128) layer product: 65.5 k),
// the bf16 pipe does the six terms in 6/16 of them, beside the vector ALU.
//
// Mapping.  A workgroup of WW wavefronts (WW = row tiles of the widest layer: 2 or 4) owns tiles of 32 samples; wavefront
// w owns row tile w (32 units) of every layer that has one.  All f32 fragments are sample-major (lane = sample + 32 half,
// register r = unit frag_unit(r, half) of the row tile: what the matrix pipe produces and what the gradient pass cached).
//   * A fragment's owner splits it ONCE and publishes the parts in LDS as a "parts image": chunk (k-block kbg = 2 t + kb',
//     part p) = 64 lanes x 16 B, lane (sample, half) holding the 8 units frag_unit(8 kb' + j, half) of row tile t --
//     directly the B operand of the next layer's products for every wavefront (one ds_read_b128 per part and k-block;
//     the K permutation is absorbed into the weight images).
//   * The same images serve the products that contract over SAMPLES (gW_l += h_{l-1}^T gz_l): ds_read_b64_tr_b16, the
//     transposing LDS read, hands a lane the four samples of ONE unit, so two reads per part are a unit-major operand --
//     no transposition products, no second split, no extra LDS image (tools/ubench/bf16_split_layout.hip pins the
//     read's lane map on the device).
//   * A operands (tangent weights dW^T, weights W^T, untransposed W for back-propagation, and the output layer in all
//     three roles) are split once per launch by cs_stage_kernel into operand images in global memory (a few hundred KB,
//     L2-resident): one k-block = three coalesced 1 KB loads, fetched one k-block ahead and across chain boundaries.
//   * The output layer is matrix work as well -- dmu = dWo^T h + Wo^T dh (each wavefront contracts over ITS 32 units, the
//     partial sums meet in LDS), gz = (Wo gmu)(1 - h^2) with the cotangent on the mean published as one more parts
//     image, gWo += h^T gmu over the sample axis -- so the vector ALU is left with the splits, the tanh derivatives and
//     the bias sums (per lane, folded over the lanes once per launch).
// One partial row per workgroup (every parameter has exactly one owner wavefront), reduce_rows_kernel sums the rows in
// float64 in a fixed order, as in the other families.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "../../include/rllab_amd.h"
#include "capi_util.h"
#include "policy_mfma.h"
#include "policy_wide.h"

namespace rl {

int launch_reduce_rows(const float* partial, int rows, int cols, double* out, hipStream_t st);   // policy_kernels.hip
bool net_has_narrow_kernel(int obs_dim, int act_dim, int h0, int h1, int h2);                    // policy_kernels.hip

namespace cs {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

constexpr int CH = 1088;        // bytes of one parts chunk: 64 lanes x 16 B, lane half 1 displaced (below)
constexpr int HSTR = 576;       // byte offset of lane half 1 inside a chunk (512 + 64: the transposing reads of the
                                // two halves' rows then fall on different banks)
constexpr int TILE_IMG = 6 * CH;   // one row tile of a parts image: two k-blocks x three parts
constexpr int MAX_IMG = 10;     // operand images of one launch
constexpr int CS_MAX_GRID = 512;
// experiment knobs (timing ablations; tools/exp/csplit_variants.sh).  CS_FETCH_LATE: the next tile's cached fragments
// start travelling after the tile's last operand-image load (in front of the products over the sample axis, which read
// LDS only) instead of at the top of the tile.  The memory queue answers in order: an operand-image load issued behind
// the HBM loads of the fragments is not "there" before they are, so with the early fetch the first chain of every tile
// sat out an HBM round trip (SQ_WAIT_ANY 50 % of the wave cycles, profiles/r06_csplit_pmc_sq.csv).  Round 4 measured the
// late fetch slower on the unbalanced shapes; with the k-slices it is 8 - 19 % faster on every shape (round 6,
// tools/exp/r06_call9.sh: (128, 128) 2.68 -> 2.47 ms, (100, 50, 25) 2.36 -> 2.18, (128, 128, 64) 4.23 -> 3.42) -- on.
#ifndef CS_FETCH_LATE
#define CS_FETCH_LATE 1
#endif
#ifndef CS_ABLATE_FETCH     // 1: no per-tile HBM loads at all (the first tile's registers are reused): wrong results
#define CS_ABLATE_FETCH 0
#endif
#ifndef CS_ABLATE_AIMG      // 1: operand images are not re-fetched per k-block: wrong results
#define CS_ABLATE_AIMG 0
#endif
#ifndef CS_ABLATE_OUTER     // 1: no products over the sample axis: wrong results
#define CS_ABLATE_OUTER 0
#endif
#ifndef CS_ABLATE_MFMA      // 1: no matrix instructions in the chains: wrong results
#define CS_ABLATE_MFMA 0
#endif
// CS_KSPLIT (round 6): a layer with FEWER row tiles than the workgroup has wavefronts (the narrow layers of a tapering net:
// (100, 50, 25) pads to 4 / 2 / 1 row tiles) is split over K as well -- the WW / HT wavefronts that share a row tile each
// contract over their slice of the k-blocks and the partial sums meet in LDS at the tile's owner, which adds the bias
// tangent, applies the tanh derivative and publishes as before.  Same for the output layer's two chains (per chain and
// k-block) and for the products over the sample axis whose output tiles are fewer than the wavefronts (per sample block:
// two wavefronts hold partial accumulators of the same tile for the whole launch and meet once, at the end).  A tile's
// time is the k-block steps of its BUSIEST wavefront (~620 cycles each at one wavefront per SIMD, whatever the shape:
// profiles/r04_notes.md): (100, 50, 25) 46 -> 27 steps.  0: ownership by row tile only (rounds 4 - 5), for A/B runs.
#ifndef CS_KSPLIT
#define CS_KSPLIT 1
#endif
// CS_SHAPES: 1 = the common shapes are instantiated with their row-tile counts as compile-time constants
#ifndef CS_SHAPES
#define CS_SHAPES 1
#endif

// kind 0: dW_l^T (vec)    1: W_l^T (theta)    2: W_l (theta; back-propagation through layer l)
//      3: dWo^T (vec)     4: Wo^T (theta)     rows = action slots (one row tile, rows >= DA zero), k = units of layer L-1
//      5: Wo (theta)      rows = units of layer L-1, k = action slots (one k-block: slots 0..7 in lane half 0, zeros beyond)
struct CsImage { int kind, l, base16, items, kb; };

struct CsShape {
    int L, DO, DA;
    int H[3], HT[3];
    int P, oW[3], ob[3], oWo, obo, ols;
    int KB[3];                      // k-blocks of 16 feeding layer l (KB[0]: input slots incl. the bias slot, 1 or 2)
    int iFD[3], iFT[3], iBT[3];     // operand images, offsets in 16-byte units (iFT[0] / iBT[0] unused)
    int iDO, iTO, iWO;              // ... of the output layer: dWo^T, Wo^T, Wo
    int img16;                      // 16-byte units of all images
    CsImage im[MAX_IMG];
    int n_im, stage_items;
    int lX, lH[3], lG[3], lM, ldb[3], lpart, lds_total;    // LDS byte offsets (lM: the image of the cotangent on the mean)
    int RF[3];                      // wavefronts sharing a row tile of layer l in the tangent forward pass (k-slices; 1: none)
    int RO;                         // ... sharing a unit tile of the last hidden layer in the output layer's chains (1, 2, 4)
    int SW[3], SO;                  // sample-axis products of layer l / of the output layer split per sample block (0 / 1)
    int fmt;                        // activation cache: 0 = fragment rows (policy_kernels.hip), 1 = unit rows (wide)
    int frow[3];                    // fmt 0: first 16-byte row of layer l inside a tile; fmt 1: float offset of layer l
    int rows, ctile;                // fmt 0: 16-byte rows per tile; fmt 1: floats per tile
};

struct CsArgs {
    int B;
    const float* theta;
    const float* vec;
    const bf16x8* img;
    const float* acts;
    const float* obs;
    const float* weight;
    float inv_count, log_min_std;
    float* partial;                 // [grid][P]
    CsShape s;
};

struct Parts { bf16x8 p[3]; };

__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// c += A B to f32 accuracy: the six cross terms, smallest first
__device__ __forceinline__ f32x16 mm6(const Parts& A, const Parts& B, f32x16 c) {
    c = mfma16(A.p[1], B.p[1], c);
    c = mfma16(A.p[0], B.p[2], c);
    c = mfma16(A.p[2], B.p[0], c);
    c = mfma16(A.p[0], B.p[1], c);
    c = mfma16(A.p[1], B.p[0], c);
    c = mfma16(A.p[0], B.p[0], c);
    return c;
}
// x = hi + mid + lo, each a bf16: successive round-to-nearest residuals (every subtraction is exact)
__device__ __forceinline__ void split_pair(float a0, float a1, Parts& out, int j) {
    const f32x2 a = {a0, a1};
    const bf16x2 h = __builtin_convertvector(a, bf16x2);
    const f32x2 r = a - __builtin_convertvector(h, f32x2);
    const bf16x2 m = __builtin_convertvector(r, bf16x2);
    const f32x2 l = r - __builtin_convertvector(m, f32x2);
    const bf16x2 q = __builtin_convertvector(l, bf16x2);
    out.p[0][j] = h[0]; out.p[0][j + 1] = h[1];
    out.p[1][j] = m[0]; out.p[1][j + 1] = m[1];
    out.p[2][j] = q[0]; out.p[2][j + 1] = q[1];
}
__device__ __forceinline__ void split8(const float* v, Parts& out) {
#pragma unroll
    for (int j = 0; j < 8; j += 2) split_pair(v[j], v[j + 1], out, j);
}

// unit of the row tile that lane l32 (= lane & 31) of a transposing read's result holds (see cs_tr)
__host__ __device__ constexpr int tr_unit(int l32) {
    return frag_unit(8 * ((l32 >> 4) & 1) + 4 * ((l32 >> 2) & 1) + (l32 & 3), (l32 >> 3) & 1);
}

// ---- operand images ---------------------------------------------------------------------------------------------------
// image element ((t * KB + kbg) * 3 + p) * 64 + lane: what lane (i = lane & 31, half) feeds k-block kbg of output row tile
// t, part p.  k of (kbg, half, j): layer-0 input slot 16 kbg + 8 half + j (slot DO carries the bias, the slots beyond
// are zero), action slot 8 half + j (kind 5), otherwise unit 32 (kbg >> 1) + frag_unit(8 (kbg & 1) + j, half) of the
// feeding fragment.
__global__ void __launch_bounds__(256) cs_stage_kernel(CsShape s, const float* __restrict__ th,
                                                       const float* __restrict__ vec, bf16x8* __restrict__ img) {
    for (int e = blockIdx.x * 256 + threadIdx.x; e < s.stage_items; e += gridDim.x * 256) {
        int q = 0, r = e;
        while (q + 1 < s.n_im && r >= s.im[q].items) { r -= s.im[q].items; ++q; }
        const CsImage im = s.im[q];
        const int lane = r & 63, kbg = (r >> 6) % im.kb, t = (r >> 6) / im.kb;
        const int i = lane & 31, half = lane >> 5, l = im.l;
        const float* src = (im.kind == 0 || im.kind == 3) ? vec : th;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ku = 32 * (kbg >> 1) + frag_unit(8 * (kbg & 1) + j, half);      // a unit of the feeding fragment
            if (im.kind <= 1 && l == 0) {
                const int d = 16 * kbg + 8 * half + j, o = 32 * t + i;
                v[j] = d < s.DO ? src[s.oW[0] + d * s.H[0] + o] : (d == s.DO ? src[s.ob[0] + o] : 0.0f);
            } else if (im.kind <= 1) {
                v[j] = src[s.oW[l] + ku * s.H[l] + 32 * t + i];                          // A[i][k] = W_l[k][i]
            } else if (im.kind == 2) {
                v[j] = src[s.oW[l] + (32 * t + i) * s.H[l] + ku];                        // A[i][k] = W_l[i][k]
            } else if (im.kind <= 4) {
                v[j] = i < s.DA ? src[s.oWo + ku * s.DA + i] : 0.0f;                     // A[i][k] = Wo[k][i]
            } else {
                const int k = 8 * half + j;
                v[j] = k < s.DA ? src[s.oWo + (32 * t + i) * s.DA + k] : 0.0f;           // A[i][k] = Wo[i][k]
            }
        }
        Parts pr;
        split8(v, pr);
        bf16x8* dst = img + im.base16 + (size_t)((t * im.kb + kbg) * 3) * 64 + lane;
#pragma unroll
        for (int p = 0; p < 3; ++p) dst[p * 64] = pr.p[p];
    }
}

// ---- LDS parts images --------------------------------------------------------------------------------------------------
// publish a fragment (lane = sample + 32 half) as the two k-blocks of its row tile, three parts each;
// slot = image + row tile * TILE_IMG + this lane's offset
__device__ __forceinline__ void cs_publish(char* slot, const f32x16& v) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        Parts q;
#pragma unroll
        for (int j = 0; j < 8; j += 2) split_pair(v[8 * kb + j], v[8 * kb + j + 1], q, j);
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<bf16x8*>(slot + (kb * 3 + p) * CH) = q.p[p];
    }
}
// unit-major operand of sample block kbs (16 samples) of a row tile of a parts image: lane (l32 = lane & 31, hh = lane >> 5)
// receives unit tr_unit(l32) at samples 16 kbs + 8 hh + j.  Each 16-lane group reads a 4 x 16 block (rows = samples):
// lane 4 r + q of the group supplies the address of row r, columns 4 q .. 4 q + 3, and receives column (lane & 15), rows
// 0..3.  src = image + row tile * TILE_IMG + the lane's transposing-read offset (tr_off in the kernel: chunk of its
// 16-lane group's k-block, its row, the lane half and the 8 bytes of its column group).
__device__ __forceinline__ Parts cs_tr(char* src, int kbs) {
    Parts out;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(src + p * CH + kbs * 256));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(src + p * CH + kbs * 256 + 64));
        const bf16x4 l4 = __builtin_bit_cast(bf16x4, lo), h4 = __builtin_bit_cast(bf16x4, hi);
#pragma unroll
        for (int e = 0; e < 4; ++e) { out.p[p][e] = l4[e]; out.p[p][4 + e] = h4[e]; }
    }
    return out;
}

// ---- the matrix chains ------------------------------------------------------------------------------------------------
// acc += A B over nkb k-blocks: A from the operand image (this lane's column; the next k-block's parts in flight while
// this one runs; during the last k-block the first parts of the wavefront's NEXT chain are fetched, so that every chain
// starts on operands that are already there: `pre`), B from a parts image in LDS (this lane's slot).
struct CsPre { bf16x8 v[2][3]; };      // the first two k-blocks of the wavefront's next chain (the second only if it has one)

// Two k-blocks of A in flight ahead of the one being multiplied (an L2 round trip is several six-term products long at
// one wavefront per SIMD), B one k-block ahead (an LDS round trip is about one product).  The wavefront's chains form one
// stream of k-blocks: `next` (of next_nkb >= 1 k-blocks) and `next2` are the chains it runs after this one, whose first
// k-blocks are fetched during this chain's last two.
__device__ __forceinline__ f32x16 cs_gemm(const bf16x8* __restrict__ A, int nkb, const char* B, f32x16 acc, CsPre& pre,
                                          const bf16x8* __restrict__ next, int next_nkb,
                                          const bf16x8* __restrict__ next2) {
    bf16x8 a0[3], a1[3], a2[3], b0[3], b1[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        a0[p] = pre.v[0][p];
        a1[p] = pre.v[1][p];
        b0[p] = *reinterpret_cast<const bf16x8*>(B + p * CH);
    }
    for (int kb = 0; kb < nkb; ++kb) {
        // k-block kb + 2 of the stream: of this chain, else k-block 0 / 1 of the next one, else (a next chain of ONE
        // k-block) k-block 0 of the one after
        const int ahead = kb + 2 - nkb;
        const bf16x8* src = ahead < 0 ? A + (size_t)(kb + 2) * 3 * 64
                                      : (ahead < next_nkb ? next + (size_t)ahead * 3 * 64 : next2);
#pragma unroll
        for (int p = 0; p < 3; ++p) a2[p] = CS_ABLATE_AIMG ? a0[p] : src[p * 64];
        const int kn = kb + 1 < nkb ? kb + 1 : kb;
#pragma unroll
        for (int p = 0; p < 3; ++p) b1[p] = *reinterpret_cast<const bf16x8*>(B + (kn * 3 + p) * CH);
        Parts Aop, Bop;
#pragma unroll
        for (int p = 0; p < 3; ++p) { Aop.p[p] = a0[p]; Bop.p[p] = b0[p]; }
#if CS_ABLATE_MFMA
#pragma unroll
        for (int p = 0; p < 3; ++p) acc[p] += (float)Aop.p[p][0] * (float)Bop.p[p][0];
#else
        acc = mm6(Aop, Bop, acc);
#endif
#pragma unroll
        for (int p = 0; p < 3; ++p) { a0[p] = a1[p]; a1[p] = a2[p]; b0[p] = b1[p]; }
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) { pre.v[0][p] = a0[p]; pre.v[1][p] = a1[p]; }
    return acc;
}

// L hidden layers; WW wavefronts per workgroup; MT = most output tiles of a hidden-to-hidden weight gradient one
// wavefront accumulates
// WSP >= 0: the instruction stream of wavefront WSP alone (CS_PER_WAVE: which chains it runs, which tiles it owns and every
// image offset that depends on them are then compile-time constants; the kernel branches ONCE, on the wavefront's index).
// All copies execute the same sequence of workgroup barriers.
template <int L, int WW, int MT, int SHP, int WSP>
__device__ __forceinline__ void cs_body(const CsArgs& a) {
    const CsShape& s = a.s;
    // row tiles per layer and everything derived from them: compile-time constants in the instantiations of the common
    // shapes (SHP = HT0 | HT1 << 4 | HT2 << 8), read from the shape record otherwise (SHP = 0)
    auto HTc = [&](int l) -> int { return SHP ? (SHP >> (4 * l)) & 15 : s.HT[l]; };
    auto Hc = [&](int l) -> int { return SHP ? 32 * ((SHP >> (4 * l)) & 15) : s.H[l]; };
    auto KBc = [&](int l) -> int { return (SHP && l >= 1) ? 2 * ((SHP >> (4 * (l - 1))) & 15) : s.KB[l]; };
    auto RFc = [&](int l) -> int {
        if (!SHP) return s.RF[l];
        if (!CS_KSPLIT || l < 1) return 1;
        const int r = WW / HTc(l), kb = KBc(l);
        return r > kb ? kb : (r < 1 ? 1 : r);
    };
    auto ROc = [&]() -> int { return SHP ? (CS_KSPLIT ? WW / HTc(L - 1) : 1) : s.RO; };
    auto SWc = [&](int l) -> int { return SHP ? (CS_KSPLIT && l >= 1 && 2 * HTc(l - 1) * HTc(l) <= WW) : s.SW[l]; };
    auto SOc = [&]() -> int { return SHP ? (CS_KSPLIT && 2 * HTc(L - 1) <= WW) : s.SO; };
    constexpr int NT = WW * WV;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = WSP >= 0 ? WSP : __builtin_amdgcn_readfirstlane(tid / WV), lane = tid % WV;
    const int lj = lane & 31, lh = lane >> 5;
    const int DO = s.DO, DA = s.DA;
    const float* __restrict__ th = a.theta;
    const float* __restrict__ vc = a.vec;
    // per-lane offsets inside a parts image: the lane's own slot (publish / B operand), and the slot it addresses in a
    // transposing read (column group q = lane & 3 of row (lane & 15) >> 2 of its 16-lane group: chunk of k-block
    // (lane >> 4) & 1, lane half q >> 1, 8 bytes q & 1; rows = samples 8 (lane >> 5) + ...)
    const int lane_off = lj * 16 + lh * HSTR;
    const int tr_off = (((lane >> 4) & 1) * 3) * CH + (8 * lh + ((lane & 15) >> 2)) * 16 + ((lane & 3) >> 1) * HSTR +
                       (lane & 1) * 8;
    char* const own = smem + lane_off + wave * TILE_IMG;       // + image offset: this wavefront's row tile, this lane's slot

    // ---- once per launch: zero the input image and the image of the mean's cotangent (their unused lanes stay zero),
    // stage the bias tangents in fragment order --------------------------------------------------------------------
    for (int k = tid; k < TILE_IMG / 4; k += NT) {
        reinterpret_cast<float*>(smem + s.lX)[k] = 0.0f;
        reinterpret_cast<float*>(smem + s.lM)[k] = 0.0f;
    }
#pragma unroll
    for (int l = 1; l < L; ++l) {
        float* db = reinterpret_cast<float*>(smem + s.ldb[l]);
        for (int e = tid; e < HTc(l) * 32; e += NT) {                // [(t * 2 + half) * 16 + r]
            const int r = e & 15, hf_ = (e >> 4) & 1, t = e >> 5;
            db[e] = vc[s.ob[l] + 32 * t + frag_unit(r, hf_)];
        }
    }
    __syncthreads();

    // ---- this wavefront's chains, in the order it runs them (for the operand prefetch): offsets into the operand
    // images in 16-byte units (wave-uniform), -1 = not this wavefront's ------------------------------------------------
    constexpr int NSEQ = 3 * L + 1;
    int seq[NSEQ], skb[NSEQ], nxt[NSEQ], nkbn[NSEQ], nxt2[NSEQ];   // image offset and k-blocks of a chain; of the chain this
                                                                   // wavefront runs next; offset of the one after that
    // k-slices (CS_KSPLIT): in layer l's tangent forward pass this wavefront works on row tile ft[l], k-blocks fk0[l] ..
    // + fnk[l] - 1 of both chains (fnk = 0: idle there); the tile's owner is the wavefront of slice 0 (= wave ft[l])
    int ft[L], fk0[L], fnk[L];
    ft[0] = wave; fk0[0] = 0; fnk[0] = wave < HTc(0) ? KBc(0) : 0;
#pragma unroll
    for (int l = 1; l < L; ++l) {
        const int ks = wave / HTc(l);
        ft[l] = wave % HTc(l);
        fnk[l] = ks < RFc(l) ? KBc(l) / RFc(l) : 0;
        fk0[l] = ks * fnk[l];
    }
    // output layer: unit tile ot of the last hidden layer; RO = 1: both chains, both k-blocks; 2: chain oq, both
    // k-blocks; 4: chain oq >> 1, k-block oq & 1
    const int ot = wave % HTc(L - 1), oq = wave / HTc(L - 1);
    const bool o_busy = oq < ROc();
    const bool o_c0 = o_busy && (ROc() == 1 || (ROc() == 2 ? oq == 0 : (oq >> 1) == 0));
    const bool o_c1 = o_busy && (ROc() == 1 || (ROc() == 2 ? oq == 1 : (oq >> 1) == 1));
    const int o_k0 = ROc() == 4 ? (oq & 1) : 0, o_nk = ROc() == 4 ? 1 : 2;
    {
        auto rowimg = [&](int base16, int kb, bool busy) -> int { return busy ? base16 + wave * kb * 3 * 64 : -1; };
        const bool last = wave < HTc(L - 1);
        seq[0] = rowimg(s.iFD[0], KBc(0), wave < HTc(0));
        skb[0] = KBc(0);
#pragma unroll
        for (int l = 1; l < L; ++l) {
            const int o16 = (ft[l] * KBc(l) + fk0[l]) * 3 * 64;
            seq[2 * l - 1] = fnk[l] ? s.iFD[l] + o16 : -1;
            seq[2 * l] = fnk[l] ? s.iFT[l] + o16 : -1;
            skb[2 * l - 1] = skb[2 * l] = fnk[l] ? fnk[l] : 1;
        }
        // output layer: ONE row tile (the action slots), k-blocks 2 t and 2 t + 1 = the units of tile t
        seq[2 * L - 1] = o_c0 ? s.iDO + (2 * ot + o_k0) * 3 * 64 : -1;
        seq[2 * L] = o_c1 ? s.iTO + (2 * ot + o_k0) * 3 * 64 : -1;
        seq[2 * L + 1] = rowimg(s.iWO, 1, last);
        skb[2 * L - 1] = skb[2 * L] = o_nk;
        skb[2 * L + 1] = 1;
#pragma unroll
        for (int l = L - 1; l >= 1; --l) {
            seq[2 * L + 2 + (L - 1 - l)] = rowimg(s.iBT[l], Hc(l) / 16, wave < HTc(l - 1));
            skb[2 * L + 2 + (L - 1 - l)] = Hc(l) / 16;
        }
    }
#pragma unroll
    for (int i = 0; i < NSEQ; ++i) {
        nxt[i] = nxt2[i] = seq[i];
        nkbn[i] = skb[i];
#pragma unroll
        for (int k = 2 * NSEQ; k >= 1; --k) {               // nearest last: (nxt2, nxt) end as the second-nearest and nearest
            const int c = seq[(i + k) % NSEQ];
            if (c >= 0) { nxt2[i] = nxt[i]; nxt[i] = c; nkbn[i] = skb[(i + k) % NSEQ]; }
        }
    }
    const bf16x8* const imgl = a.img + lane;
    CsPre pre;
    {
        int first = 0, first_kb = 1, second = 0;            // the wavefront's first chain; the chain after it
#pragma unroll
        for (int i = NSEQ - 1; i >= 0; --i)
            if (seq[i] >= 0) { first = seq[i]; first_kb = skb[i]; second = nxt[i]; }
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            pre.v[0][p] = imgl[first + p * 64];
            pre.v[1][p] = imgl[(first_kb > 1 ? first + 3 * 64 : second) + p * 64];
        }
    }

    // ---- accumulators ------------------------------------------------------------------------------------------------
    f32x16 gW[L - 1][MT], gW0, gWo;      // gWo: row tile `wave` of dWo (columns = action slots; beyond DA: zero)
    float gb[L - 1][16], gbo4[4], wsum = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        gW0[r] = 0.0f;
        gWo[r] = 0.0f;
#pragma unroll
        for (int l = 0; l < L - 1; ++l) {
            gb[l][r] = 0.0f;
#pragma unroll
            for (int m = 0; m < MT; ++m) gW[l][m][r] = 0.0f;
        }
    }
    // this lane's four action slots in the output layer's products: rows frag_unit(r, half), r < 4 = slots r + 4 half
    float fk4[4], dbo4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = r + 4 * lh;
        const float raw = k < DA ? th[s.ols + k] : 0.0f;
        const float var = __expf(2.0f * fmaxf(raw, a.log_min_std));
        fk4[r] = k < DA ? 2.0f / (2.0f * var + 1e-8f) : 0.0f;
        dbo4[r] = k < DA ? vc[s.obo + k] : 0.0f;
        gbo4[r] = 0.0f;
    }

    const int B = a.B, n_tiles = B / TS;
    char* const Xp = smem + s.lX;
    float* const part = reinterpret_cast<float*>(smem + s.lpart);
    f32x4* const xch = reinterpret_cast<f32x4*>(smem + s.lpart);    // k-slice exchange slots (4 KB each) alias the partial sums
                                                                    // of the output layer: different phases of a tile

    // one tile ahead: the cached fragments of the layers this wavefront owns, the observation slots of this thread's
    // item of the input image, the sample weight
    f32x16 hq[L];
    float xq[8], wq = 0.0f;
    const bool x_item = tid < KBc(0) * WV;
    const int x_kb = tid >> 6;
    auto fetch = [&](int tile) {
#pragma unroll
        for (int l = 0; l < L; ++l) {
            if (wave < HTc(l)) {
                if (s.fmt == 0) {
                    const f32x4* src = reinterpret_cast<const f32x4*>(a.acts) +
                                       ((size_t)tile * s.rows + s.frow[l] + 4 * wave) * WV + lane;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = __builtin_nontemporal_load(src + q * WV);
#pragma unroll
                        for (int e = 0; e < 4; ++e) hq[l][4 * q + e] = v[e];
                    }
                } else {
                    const float* src = a.acts + (size_t)tile * s.ctile + s.frow[l] + (32 * wave + 4 * lh) * 32 + lj;
#pragma unroll
                    for (int r = 0; r < 16; ++r) hq[l][r] = __builtin_nontemporal_load(src + frag_unit(r, 0) * 32);
                }
            }
        }
        wq = a.weight[tile * TS + lj];
        if (x_item) {
            const int b = tile * TS + lj;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int d = 16 * x_kb + 8 * lh + j;
                const float v = a.obs[(size_t)(d < DO ? d : DO - 1) * B + b];
                xq[j] = d < DO ? v : (d == DO ? 1.0f : 0.0f);
            }
        }
    };
    if ((int)blockIdx.x < n_tiles) fetch(blockIdx.x);

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        f32x16 hf[L];
#pragma unroll
        for (int l = 0; l < L; ++l) hf[l] = hq[l];
        const float wgt = wq;
        __syncthreads();                                    // everybody is done with the previous tile's images
        // ---- the tile's inputs: observation parts, the parts of the fragments this wavefront owns ----------------------
        if (x_item) {
            Parts q;
            split8(xq, q);
#pragma unroll
            for (int p = 0; p < 3; ++p) *reinterpret_cast<bf16x8*>(Xp + (x_kb * 3 + p) * CH + lane_off) = q.p[p];
        }
#pragma unroll
        for (int l = 0; l < L; ++l)
            if (wave < HTc(l)) cs_publish(own + s.lH[l], hf[l]);
        if (!CS_FETCH_LATE && !CS_ABLATE_FETCH) {
            // the next tile starts travelling (the workgroup's last tile fetches itself again: no branch around the loads)
            const int nx = tile + (int)gridDim.x < n_tiles ? tile + (int)gridDim.x : tile;
            fetch(nx);
        }
        __syncthreads();

        // ---- tangent forward: dh_0 = (dW0^T x + db0) (1 - h0^2), dh_l = (dW_l^T h_{l-1} + W_l^T dh_{l-1} + db_l) (1 - h_l^2);
        // every layer's tangent is published (its image passes to the cotangent later) ----------------------------------
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const bool owner = wave < HTc(l);
            f32x16 acc;
            if (l == 0) {
                if (owner) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
                    acc = cs_gemm(imgl + seq[0], KBc(0), Xp + lane_off, acc, pre, imgl + nxt[0], nkbn[0], imgl + nxt2[0]);
                }
            } else {
                if (fnk[l]) {
                    // the owner starts from the bias tangent, a k-slice from zero
                    const f32x4* db = reinterpret_cast<const f32x4*>(smem + s.ldb[l]) + (ft[l] * 2 + lh) * 4;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = owner ? db[q] : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[4 * q + e] = v[e];
                    }
                    const int bo = fk0[l] * 3 * CH + lane_off;
                    acc = cs_gemm(imgl + seq[2 * l - 1], fnk[l], smem + s.lH[l - 1] + bo, acc, pre, imgl + nxt[2 * l - 1], nkbn[2 * l - 1], imgl + nxt2[2 * l - 1]);
                    acc = cs_gemm(imgl + seq[2 * l], fnk[l], smem + s.lG[l - 1] + bo, acc, pre, imgl + nxt[2 * l], nkbn[2 * l], imgl + nxt2[2 * l]);
                    if (!owner) {                            // slice ks >= 1 of tile ft: exchange slot wave - HT[l]
                        f32x4* dst = xch + ((wave - HTc(l)) * 4) * WV + lane;
#pragma unroll
                        for (int q = 0; q < 4; ++q) dst[q * WV] = f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
                    }
                }
                if (RFc(l) > 1) __syncthreads();
                if (owner && RFc(l) > 1) {
                    for (int ks = 1; ks < RFc(l); ++ks) {
                        const f32x4* src = xch + ((wave + HTc(l) * (ks - 1)) * 4) * WV + lane;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 v = src[q * WV];
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[4 * q + e] += v[e];
                        }
                    }
                }
            }
            if (owner) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] *= (1.0f - hf[l][r] * hf[l][r]);
                cs_publish(own + s.lG[l], acc);
            }
            if (l + 1 < L || ROc() > 1) __syncthreads();      // (RO > 1: the output chains read other wavefronts' images)
        }

        // ---- output layer: dmu = dWo^T h + Wo^T dh + dbo.  Each wavefront contracts over its own 32 units (the two
        // k-blocks it has just published: its LDS operations complete in order, no barrier), partial sums meet in LDS ----
        if (o_busy) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            const int bo = ot * TILE_IMG + o_k0 * 3 * CH + lane_off;
            if (o_c0) acc = cs_gemm(imgl + seq[2 * L - 1], o_nk, smem + s.lH[L - 1] + bo, acc, pre, imgl + nxt[2 * L - 1], nkbn[2 * L - 1], imgl + nxt2[2 * L - 1]);
            if (o_c1) acc = cs_gemm(imgl + seq[2 * L], o_nk, smem + s.lG[L - 1] + bo, acc, pre, imgl + nxt[2 * L], nkbn[2 * L], imgl + nxt2[2 * L]);
            f32x4 v4;                                        // rows frag_unit(r, half), r < 4 = action slots r + 4 half
#pragma unroll
            for (int r = 0; r < 4; ++r) v4[r] = acc[r];
            *reinterpret_cast<f32x4*>(part + (wave * WV + lane) * 4) = v4;
        }
        __syncthreads();
        {
            const float c = wgt * a.inv_count;
            f32x4 g4 = {0.0f, 0.0f, 0.0f, 0.0f};
            for (int w = 0; w < HTc(L - 1) * ROc(); ++w) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(part + (w * WV + lane) * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) g4[r] += v[r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) g4[r] = c * (g4[r] + dbo4[r]) * fk4[r];           // cotangent on the mean, slots r + 4 half
            if (wave == 0) {
                // its image: one chunk per part, lane (sample, half 0) = the sample's 8 slots (the other half's four by a
                // half swap), and the per-lane sums of dbo and of the weights
                float o4[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    o4[r] = __shfl_xor(g4[r], 32, WV);
                    gbo4[r] += g4[r];
                }
                if (lh == 0) {
                    float g8[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { g8[r] = g4[r]; g8[4 + r] = o4[r]; }
                    Parts q;
                    split8(g8, q);
#pragma unroll
                    for (int p = 0; p < 3; ++p) *reinterpret_cast<bf16x8*>(smem + s.lM + p * CH + lj * 16) = q.p[p];
                    wsum += c;
                }
            }
        }
        __syncthreads();

        // ---- back-propagation: gz_{L-1} = (Wo gmu) (1 - h^2), gz_{l-1} = (W_l gz_l) (1 - h_{l-1}^2) ----------------------
        if (wave < HTc(L - 1)) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            acc = cs_gemm(imgl + seq[2 * L + 1], 1, smem + s.lM + lane_off, acc, pre, imgl + nxt[2 * L + 1], nkbn[2 * L + 1], imgl + nxt2[2 * L + 1]);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[r] *= (1.0f - hf[L - 1][r] * hf[L - 1][r]);
                gb[L - 2][r] += acc[r];
            }
            cs_publish(own + s.lG[L - 1], acc);                // (its tangent was consumed by this wavefront's own chain above)
        }
        __syncthreads();
#pragma unroll
        for (int l = L - 1; l >= 1; --l) {
            if (wave < HTc(l - 1)) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
                const int si = 2 * L + 2 + (L - 1 - l);
                acc = cs_gemm(imgl + seq[si], Hc(l) / 16, smem + s.lG[l] + lane_off, acc, pre, imgl + nxt[si], nkbn[si], imgl + nxt2[si]);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc[r] *= (1.0f - hf[l - 1][r] * hf[l - 1][r]);
                    if (l >= 2) gb[l >= 2 ? l - 2 : 0][r] += acc[r];            // (db0 rides in gW0's bias row)
                }
                cs_publish(own + s.lG[l - 1], acc);                             // the tangent of layer l-1 is consumed
            }
            __syncthreads();
        }

        if (CS_FETCH_LATE && !CS_ABLATE_FETCH) {
            // the next tile starts travelling now: the products below read LDS only, and a wait for an operand-image load
            // (the memory queue answers in order) can no longer land behind these HBM loads
            // (the workgroup's last tile fetches itself again: no branch around the loads)
            const int nx = tile + (int)gridDim.x < n_tiles ? tile + (int)gridDim.x : tile;
            fetch(nx);
        }
        // ---- the products over the sample axis (K = 32 samples = two sample blocks): gW_l += h_{l-1}^T gz_l,
        // gWo += h_{L-1}^T gmu, gW0 += x_ext^T gz_0 ------------------------------------------------------------------
        if (!CS_ABLATE_OUTER) {
#pragma unroll
        for (int l = 1; l < L; ++l) {
            const int HTa = HTc(l - 1), nt = HTa * HTc(l);
            if (SWc(l)) {
                // fewer output tiles than wavefronts: wavefront w takes sample block w / nt of tile w % nt (its partial
                // accumulator meets the other block's after the last tile)
                if (wave < 2 * nt) {
                    const int tau = wave % nt, kbs = wave / nt, ti = tau % HTa, tj = tau / HTa;
                    const Parts A = cs_tr(smem + s.lH[l - 1] + ti * TILE_IMG + tr_off, kbs);
                    const Parts Bq = cs_tr(smem + s.lG[l] + tj * TILE_IMG + tr_off, kbs);
                    gW[l - 1][0] = mm6(A, Bq, gW[l - 1][0]);
                }
            } else {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int tau = wave + WW * m;
                if (tau < nt) {
                    const int ti = tau % HTa, tj = tau / HTa;
#pragma unroll
                    for (int kbs = 0; kbs < 2; ++kbs) {
                        const Parts A = cs_tr(smem + s.lH[l - 1] + ti * TILE_IMG + tr_off, kbs);
                        const Parts Bq = cs_tr(smem + s.lG[l] + tj * TILE_IMG + tr_off, kbs);
                        gW[l - 1][m] = mm6(A, Bq, gW[l - 1][m]);
                    }
                }
            }
            }
        }
        if (SOc()) {
            // (from the top: wavefronts WW - 1, WW - 2, .. -- the hidden layers' split products fill the low ones first)
            const int wo = WW - 1 - wave;
            if (wo < 2 * HTc(L - 1)) {
                const int t = wo % HTc(L - 1), kbs = wo / HTc(L - 1);
                const Parts A = cs_tr(smem + s.lH[L - 1] + t * TILE_IMG + tr_off, kbs);
                const Parts Bq = cs_tr(smem + s.lM + tr_off, kbs);
                gWo = mm6(A, Bq, gWo);
            }
        } else if (wave < HTc(L - 1)) {
#pragma unroll
            for (int kbs = 0; kbs < 2; ++kbs) {
                const Parts A = cs_tr(smem + s.lH[L - 1] + wave * TILE_IMG + tr_off, kbs);
                const Parts Bq = cs_tr(smem + s.lM + tr_off, kbs);               // columns: lane l32 < 8 = action slot l32
                gWo = mm6(A, Bq, gWo);
            }
        }
        if (wave < HTc(0)) {
#pragma unroll
            for (int kbs = 0; kbs < 2; ++kbs) {
                const Parts A = cs_tr(Xp + tr_off, kbs);                          // rows: lane l32 = input slot l32
                const Parts Bq = cs_tr(smem + s.lG[0] + wave * TILE_IMG + tr_off, kbs);
                gW0 = mm6(A, Bq, gW0);
            }
        }
        }
    }

    // ---- accumulators that were split per sample block meet at the owner of their tile (the parts images are dead: the
    // fold area aliases them, one 4 KB slot per wavefront and accumulator) --------------------------------------------
    {
        bool any = SOc() != 0;
#pragma unroll
        for (int l = 1; l < L; ++l) any = any || SWc(l) != 0;
        if (any) {
            __syncthreads();
            f32x4* const fold = reinterpret_cast<f32x4*>(smem);
            auto put = [&](int slot, const f32x16& v) {
#pragma unroll
                for (int q = 0; q < 4; ++q) fold[(slot * 4 + q) * WV + lane] = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
            };
            auto add = [&](int slot, f32x16& v) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 t = fold[(slot * 4 + q) * WV + lane];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[4 * q + e] += t[e];
                }
            };
#pragma unroll
            for (int l = 1; l < L; ++l) {
                const int nt = HTc(l - 1) * HTc(l);
                if (SWc(l) && wave >= nt && wave < 2 * nt) put(wave * L + l - 1, gW[l - 1][0]);
            }
            const int wo = WW - 1 - wave, hto = HTc(L - 1);
            if (SOc() && wo >= hto && wo < 2 * hto) put(wave * L + L - 1, gWo);
            __syncthreads();
#pragma unroll
            for (int l = 1; l < L; ++l) {
                const int nt = HTc(l - 1) * HTc(l);
                if (SWc(l) && wave < nt) add((wave + nt) * L + l - 1, gW[l - 1][0]);
            }
            if (SOc() && wo < hto) add((WW - 1 - (wo + hto)) * L + L - 1, gWo);
        }
    }
    // (with SO the owner of unit tile t of the output layer's gradient is wavefront WW - 1 - t)
    const int o_owner_tile = SOc() ? WW - 1 - wave : wave;

    // ---- one partial row per workgroup: every parameter has exactly one owner -------------------------------------------
    float* row = a.partial + (size_t)blockIdx.x * s.P;
    const int cu = tr_unit(lj);
#pragma unroll
    for (int l = 1; l < L; ++l) {
        const int HTa = HTc(l - 1), nt = HTa * HTc(l);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int tau = wave + WW * m;
            if (tau < nt) {
                const int ti = tau % HTa, tj = tau / HTa;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    row[s.oW[l] + (32 * ti + tr_unit(frag_unit(r, lh))) * Hc(l) + 32 * tj + cu] = gW[l - 1][m][r];
            }
        }
    }
    if (wave < HTc(0)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = frag_unit(r, lh);                  // input slot (the input image's units are its slots)
            if (d < DO) row[s.oW[0] + d * Hc(0) + 32 * wave + cu] = gW0[r];
            else if (d == DO) row[s.ob[0] + 32 * wave + cu] = gW0[r];
        }
    }
    if (o_owner_tile >= 0 && o_owner_tile < HTc(L - 1) && lj < DA) {      // column lj of the product = action slot lj
#pragma unroll
        for (int r = 0; r < 16; ++r) row[s.oWo + (32 * o_owner_tile + tr_unit(frag_unit(r, lh))) * DA + lj] = gWo[r];
    }
    // per-lane sums of the bias gradients: fold the 32 samples of a lane half
#pragma unroll
    for (int l = 1; l < L; ++l) {
        if (wave < HTc(l)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = gb[l - 1][r];
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, WV);
                if (lj == 0) row[s.ob[l] + 32 * wave + frag_unit(r, lh)] = v;
            }
        }
    }
    if (wave == 0) {
        const float ws = wave_sum(wsum);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = gbo4[r];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, WV);
            const int k = r + 4 * lh;
            if (lj == 0 && k < DA) {
                row[s.obo + k] = v;
                // log_std block of the Fisher: d2KL/ds2 = 4 v (2 v - eps) / (2 v + eps)^2, v = sigma^2
                const float raw = th[s.ols + k];
                const float vv = __expf(2.0f * fmaxf(raw, a.log_min_std)), e = 1e-8f;
                const float cc = raw < a.log_min_std ? 0.0f : 4.0f * vv * (2.0f * vv - e) / ((2.0f * vv + e) * (2.0f * vv + e));
                row[s.ols + k] = cc * vc[s.ols + k] * ws;
            }
        }
    }
}

// CS_PER_WAVE 1: one copy of the instruction stream per wavefront index for the three commonest shapes.  MEASURED (round 6,
// tools/exp/r06_call7.sh, same box): (100, 50, 25) 2.333 against 2.346 ms, (128, 64) 1.82 against 1.87, (128, 128) 3.01
// against 2.71 (four copies of the loop no longer sit in the instruction cache together) -- off.
#ifndef CS_PER_WAVE
#define CS_PER_WAVE 0
#endif
template <int L, int WW, int MT, int SHP>
__global__ void __launch_bounds__(WW * WV, 1) csplit_fvp_kernel(CsArgs a) {
    if constexpr (CS_PER_WAVE && (SHP == 0x124 || SHP == 0x24 || SHP == 0x44)) {
        static_assert(WW == 4, "four copies");
        switch (__builtin_amdgcn_readfirstlane(threadIdx.x / WV)) {
            case 0: cs_body<L, WW, MT, SHP, 0>(a); break;
            case 1: cs_body<L, WW, MT, SHP, 1>(a); break;
            case 2: cs_body<L, WW, MT, SHP, 2>(a); break;
            default: cs_body<L, WW, MT, SHP, 3>(a); break;
        }
    } else {
        cs_body<L, WW, MT, SHP, -1>(a);
    }
}

// ---- host side -----------------------------------------------------------------------------------------------------------
static bool cs_shape(const rl_policy_batch* g, CsShape& s) {
    WideShape w;
    if (!wide_shape(g->obs_dim, g->act_dim, g->hidden0, g->hidden1, g->hidden2, w)) return false;
    s.L = w.L; s.DO = w.DO; s.DA = w.DA; s.P = w.P; s.oWo = w.oWo; s.obo = w.obo; s.ols = w.ols;
    int maxht = 0;
    for (int l = 0; l < 3; ++l) {
        s.H[l] = w.H[l]; s.HT[l] = w.HT[l]; s.oW[l] = w.oW[l]; s.ob[l] = w.ob[l];
        if (w.HT[l] > maxht) maxht = w.HT[l];
    }
    if (maxht < 2) return false;                       // all layers 32 wide: policy_split_kernels.hip / policy_kernels.hip
    s.KB[0] = (s.DO + 1 + 15) / 16;
    for (int l = 1; l < 3; ++l) s.KB[l] = l < s.L ? s.H[l - 1] / 16 : 0;
    // operand images
    int off = 0, n = 0, items = 0;
    auto add = [&](int kind, int l, int ht_out, int kb) {
        CsImage& im = s.im[n++];
        im.kind = kind; im.l = l; im.base16 = off; im.kb = kb; im.items = ht_out * kb * 64;
        off += ht_out * kb * 3 * 64;
        items += im.items;
        return im.base16;
    };
    for (int l = 0; l < 3; ++l) { s.iFD[l] = s.iFT[l] = s.iBT[l] = 0; }
    const int lastl = s.L - 1;
    for (int l = 0; l < s.L; ++l) s.iFD[l] = add(0, l, s.HT[l], s.KB[l]);
    for (int l = 1; l < s.L; ++l) s.iFT[l] = add(1, l, s.HT[l], s.KB[l]);
    for (int l = 1; l < s.L; ++l) s.iBT[l] = add(2, l, s.HT[l - 1], s.H[l] / 16);
    s.iDO = add(3, lastl, 1, s.H[lastl] / 16);
    s.iTO = add(4, lastl, 1, s.H[lastl] / 16);
    s.iWO = add(5, lastl, s.HT[lastl], 1);
    s.n_im = n; s.stage_items = items; s.img16 = off;
    // LDS plan
    int o = 0;
    s.lX = o; o += TILE_IMG;
    for (int l = 0; l < 3; ++l) { s.lH[l] = o; o += (l < s.L) ? s.HT[l] * TILE_IMG : 0; }
    for (int l = 0; l < 3; ++l) { s.lG[l] = o; o += (l < s.L) ? s.HT[l] * TILE_IMG : 0; }
    s.lM = o; o += TILE_IMG;
    for (int l = 0; l < 3; ++l) { s.ldb[l] = o; o += (l >= 1 && l < s.L) ? s.HT[l] * 128 : 0; }
    // k-slices (see CS_KSPLIT)
    const int WWh = maxht;          // wavefronts per workgroup = row tiles of the widest layer (2 or 4)
    int xslots = 1;
    s.RF[0] = 1;
    for (int l = 1; l < 3; ++l) {
        s.RF[l] = 1; s.SW[l] = 0;
        if (CS_KSPLIT && l < s.L) {
            int r = WWh / s.HT[l];
            if (r > s.KB[l]) r = s.KB[l];
            s.RF[l] = r < 1 ? 1 : r;
            if (s.HT[l] * (s.RF[l] - 1) > xslots) xslots = s.HT[l] * (s.RF[l] - 1);
            s.SW[l] = 2 * s.HT[l - 1] * s.HT[l] <= WWh;
        }
    }
    s.SW[0] = 0;
    s.RO = CS_KSPLIT ? WWh / s.HT[lastl] : 1;
    s.SO = CS_KSPLIT && 2 * s.HT[lastl] <= WWh;
    s.lpart = o; o += xslots * 4 * WV * 4 * (int)sizeof(float);
    if (o < WWh * s.L * 4 * WV * 4 * (int)sizeof(float)) o = WWh * s.L * 4 * WV * 4 * (int)sizeof(float);   // the fold area
    s.lds_total = o;
    // activation cache, as the gradient pass of the same net wrote it
    s.fmt = net_has_narrow_kernel(g->obs_dim, g->act_dim, g->hidden0, g->hidden1, g->hidden2) ? 0 : 1;
    int rows = 0, fl = 0;
    for (int l = 0; l < 3; ++l) {
        s.frow[l] = s.fmt == 0 ? rows : fl;
        rows += (l < s.L) ? 4 * s.HT[l] : 0;
        fl += (l < s.L) ? 32 * s.H[l] : 0;
    }
    s.rows = rows; s.ctile = fl;
    return o <= 160 * 1024;
}

static size_t cs_workspace(const CsShape& s) {
    const size_t rows = ((size_t)CS_MAX_GRID * s.P * sizeof(float) + 15) & ~(size_t)15;
    return rows + (size_t)s.img16 * 16 + 64;
}

template <int L, int WW, int MT, int SHP = 0>
static int launch(const CsShape& s, const rl_policy_batch* g, const float* vec, void* ws, size_t ws_bytes, double* out,
                  hipStream_t st) {
    if (ws_bytes < cs_workspace(s))
        return set_error(RL_ERR_ARG, "policy pass workspace too small: %zu < %zu bytes", ws_bytes, cs_workspace(s));
    CsArgs a;
    a.s = s;
    a.B = g->n_samples; a.theta = g->theta; a.vec = vec; a.acts = g->activations; a.obs = g->obs; a.weight = g->weights;
    a.inv_count = g->inv_count; a.log_min_std = g->log_min_std;
    const size_t rows = ((size_t)CS_MAX_GRID * s.P * sizeof(float) + 15) & ~(size_t)15;
    a.partial = (float*)ws;
    bf16x8* img = reinterpret_cast<bf16x8*>((char*)ws + rows);
    a.img = img;
    const int sg = (s.stage_items + 255) / 256;
    hipLaunchKernelGGL(cs_stage_kernel, dim3(sg < 1024 ? sg : 1024), dim3(256), 0, st, s, g->theta, vec, img);
    int rc = check_launch("cs_stage_kernel");
    if (rc) return rc;
    const int n_tiles = a.B / TS;
    int per_cu = (160 * 1024) / s.lds_total;
    if (per_cu > 4 / WW) per_cu = 4 / WW;              // one wavefront per SIMD (512 registers each)
    if (per_cu < 1) per_cu = 1;
    int grid = 256 * per_cu;
    if (grid > n_tiles) grid = n_tiles;
    if (grid > CS_MAX_GRID) grid = CS_MAX_GRID;
    auto kern = csplit_fvp_kernel<L, WW, MT, SHP>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024);
        if (e != hipSuccess) return set_error(RL_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WW * WV), s.lds_total, st, a);
    rc = check_launch("csplit_fvp_kernel");
    if (rc) return rc;
    return launch_reduce_rows(a.partial, grid, s.P, out, st);
}

template <int L>
static int launch_class(const CsShape& s, const rl_policy_batch* g, const float* vec, void* ws, size_t ws_bytes,
                        double* out, hipStream_t st) {
    int maxht = 0, mt4 = 0;
    for (int l = 0; l < s.L; ++l) {
        if (s.HT[l] > maxht) maxht = s.HT[l];
        if (l >= 1 && s.HT[l - 1] * s.HT[l] > 8) mt4 = 1;
    }
    // the tapering / equal-width shapes with a 128-unit first layer as compile-time constants (CS_SHAPES; HT0 | HT1 << 4 |
    // HT2 << 8): (128, 128), (128, 64) = the padded (100, 50), (128, 32); (128, 64, 32) = the padded (100, 50, 25) of the
    // reference's benchmark policies, ...; everything else runs the generic instantiations
    if (CS_SHAPES) {
        const int shp = s.HT[0] | s.HT[1] << 4 | (s.L == 3 ? s.HT[2] << 8 : 0);
#define CS_CASE(LL, SH, MTT) if (shp == SH) return launch<LL, 4, MTT, SH>(s, g, vec, ws, ws_bytes, out, st);
        if constexpr (L == 2) {
            CS_CASE(2, 0x44, 4) CS_CASE(2, 0x24, 2) CS_CASE(2, 0x14, 2) CS_CASE(2, 0x42, 2) CS_CASE(2, 0x41, 2)
        } else {
            CS_CASE(3, 0x244, 4) CS_CASE(3, 0x144, 4) CS_CASE(3, 0x224, 2) CS_CASE(3, 0x124, 2) CS_CASE(3, 0x114, 2)
        }
#undef CS_CASE
    }
    if (maxht == 2) return launch<L, 2, 2>(s, g, vec, ws, ws_bytes, out, st);
    return mt4 ? launch<L, 4, 4>(s, g, vec, ws, ws_bytes, out, st) : launch<L, 4, 2>(s, g, vec, ws, ws_bytes, out, st);
}

}  // namespace cs

// The cooperative split product takes a cached Fisher-vector product of a tanh net with two or three layers of 32 / 64 /
// 128 units whose batch is a whole number of 32-sample tiles.  By default only nets with a 128-unit layer (four
// wavefronts per tile): measured on MI355X (profiles/r04_notes.md) it is 13-16 % faster than wide_pass_kernel there and
// 20 % SLOWER than policy_pass_kernel on (64, 64) nets, where LDS lets only two tiles be in flight per CU on two
// wavefronts each.  rl_launch_opts.fvp_split = 1 switches it off (A/B runs; the f32-matrix-instruction kernels then keep cached ==
// recomputed bit for bit), = 2 takes every shape it is built for (the parity tests of the two-wavefront
// class run that way).
bool csplit_fvp_takes(const rl_policy_batch* g) {
    if (!g->activations || g->activation != RL_ACT_TANH || g->layer_activations != 0 || g->n_samples <= 0 ||
        g->n_samples % TS != 0)
        return false;
    const int req = g->opts ? g->opts->fvp_split : 0;          // 1: off, 2: every shape the kernel is built for
    if (req == 1) return false;
    cs::CsShape s;
    if (!cs::cs_shape(g, s)) return false;
    if (req == 2) return true;
    return s.HT[0] == 4 || s.HT[1] == 4 || s.HT[2] == 4;
}
size_t csplit_workspace_bytes_for(int obs_dim, int act_dim, int h0, int h1, int h2) {
    rl_policy_batch g = {};
    g.obs_dim = obs_dim; g.act_dim = act_dim; g.hidden0 = h0; g.hidden1 = h1; g.hidden2 = h2;
    cs::CsShape s;
    if (!cs::cs_shape(&g, s)) return 0;
    return cs::cs_workspace(s);
}
int csplit_fvp_dispatch(const rl_policy_batch* g, const float* vec, void* ws, size_t ws_bytes, double* out, hipStream_t st) {
    if (!csplit_fvp_takes(g)) return RL_SPLIT_NOT_TAKEN;
    cs::CsShape s;
    cs::cs_shape(g, s);
    return s.L == 2 ? cs::launch_class<2>(s, g, vec, ws, ws_bytes, out, st)
                    : cs::launch_class<3>(s, g, vec, ws, ws_bytes, out, st);
}

}  // namespace rl
