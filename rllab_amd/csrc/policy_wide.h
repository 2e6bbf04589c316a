// policy_wide.h -- shapes of the wide / deep GaussianMLPPolicy kernels (policy_wide_kernels.hip, and the rollout
// policy of the same nets in env_kernels.hip): two or three tanh hidden layers of 32, 64 or 128 units each
// (hidden sizes of the reference's GaussianMLPPolicy are free-form, rllab/policies/gaussian_mlp_policy.py:21-38,
// rllab/core/network.py:36-101; narrower layers are zero-padded by policies/kernel_layout.py, which is exact).
//
// Parameter vector (the kernels' layout = the reference's flat order with the padded widths):
//   W0 [DO][H0]  b0 [H0]  W1 [H0][H1]  b1 [H1]  (W2 [H1][H2]  b2 [H2])  Wo [HL][DA]  bo [DA]  log_std [DA]
//
// MFMA operand images.  Every dense layer is evaluated transposed, Z^T[unit][sample] = A[unit][k] B[k][sample] on
// v_mfma_f32_32x32x2_f32, one 32-unit row tile per wavefront; the A operands of ALL layers are pre-arranged once
// per pass as "fragment images" in global memory (L2-resident, a few hundred KB): image[(t * KS + m) * 64 + lane]
// is what lane `lane` feeds k-step m of row tile t, i.e. A[32 t + lane % 32][2 m + lane / 32], so a k-step's
// operand is one coalesced 256-byte load.  B operands come from LDS activation tiles [unit][33] (stride 33: the
// same tile is read row-wise as a B operand and column-wise as an operand of the sample-axis outer products).
#pragma once
#include <stdint.h>

namespace rl {

constexpr int WIDE_MAX_H = 128;
constexpr int WIDE_MAX_L = 3;
constexpr int WIDE_MAX_DA = 8;
constexpr int WIDE_MAX_DO = 30;        // DO + 1 (bias slot) must fit the 32 rows of the input tile
constexpr int WIDE_BS = 33;            // LDS tile row stride (floats)

struct WideShape {
    int L;                 // hidden layers: 2 or 3
    int ident;             // bit l: hidden layer l is the IDENTITY (a one-hidden-layer net of 65 .. 128 units runs as (128, 128)
                           // with W1 = I, b1 = 0, policies/kernel_layout.py); every other layer is tanh
    int DO, DA;
    int H[WIDE_MAX_L];     // padded widths, 32 / 64 / 128 (H[2] = 0 when L == 2)
    int HT[WIDE_MAX_L];    // H / 32
    int KS[WIDE_MAX_L];    // k-steps of the forward chain of layer l: KS[0] = input slots / 2 padded to 4,
                           // KS[l] = H[l-1] / 2 (the bias initialises the accumulator)
    int KT[WIDE_MAX_L];    // k-steps of the backward chain THROUGH layer l (l >= 1): H[l] / 2
    // offsets into the flat parameter vector
    int oW[WIDE_MAX_L], ob[WIDE_MAX_L], oWo, obo, ols, P;
    // offsets (floats) of the images inside one image set
    int oF[WIDE_MAX_L];    // forward image of layer l:  [HT[l]][KS[l]][64],   A[i][k] = W_l[k][i]
    int oT[WIDE_MAX_L];    // backward image through layer l >= 1: [HT[l-1]][KT[l]][64], A[i][k] = W_l[i][k]
    int img_fwd;           // floats of the forward images (all layers)
    int img_all;           // forward + backward images
    // tail parameters staged in LDS: biases of layers >= 1, Wo, bo, log_std
    int tb[WIDE_MAX_L];    // offset of b_l inside the tail (tb[0] unused: b0 rides in the input tile's bias slot)
    int tWo, tbo, tls, tail;
};

inline bool wide_shape(int DO, int DA, int h0, int h1, int h2, WideShape& s) {
    const int hs[3] = {h0, h1, h2};
    s.L = (h2 > 0) ? 3 : 2;
    s.ident = 0;
    if (DO < 1 || DO > WIDE_MAX_DO || DA < 1 || DA > WIDE_MAX_DA) return false;
    for (int l = 0; l < WIDE_MAX_L; ++l) {
        s.H[l] = (l < s.L) ? hs[l] : 0;
        if (l < s.L && hs[l] != 32 && hs[l] != 64 && hs[l] != 128) return false;
        s.HT[l] = s.H[l] / 32;
    }
    s.DO = DO; s.DA = DA;
    int off = 0, in = DO;
    for (int l = 0; l < s.L; ++l) {
        s.oW[l] = off; off += in * s.H[l];
        s.ob[l] = off; off += s.H[l];
        in = s.H[l];
    }
    s.oWo = off; off += in * DA;
    s.obo = off; off += DA;
    s.ols = off; off += DA;
    s.P = off;
    s.KS[0] = ((DO + 2) / 2 + 3) & ~3;
    for (int l = 1; l < WIDE_MAX_L; ++l) s.KS[l] = (l < s.L) ? s.H[l - 1] / 2 : 0;
    for (int l = 0; l < WIDE_MAX_L; ++l) s.KT[l] = (l >= 1 && l < s.L) ? s.H[l] / 2 : 0;
    int img = 0;
    for (int l = 0; l < WIDE_MAX_L; ++l) { s.oF[l] = img; img += s.HT[l] * s.KS[l] * 64; }
    s.img_fwd = img;
    s.oT[0] = img;
    for (int l = 1; l < WIDE_MAX_L; ++l) { s.oT[l] = img; img += (l < s.L ? s.HT[l - 1] : 0) * s.KT[l] * 64; }
    s.img_all = img;
    int t = 0;
    s.tb[0] = 0;
    for (int l = 1; l < WIDE_MAX_L; ++l) { s.tb[l] = t; t += (l < s.L) ? s.H[l] : 0; }
    s.tWo = t; t += s.H[s.L - 1] * DA;
    s.tbo = t; t += DA;
    s.tls = t; t += DA;
    s.tail = (t + 3) & ~3;
    return true;
}

}  // namespace rl

namespace rl {
// layer_activations (rl_policy_batch / rl_rollout_args: 2 bits per layer, 0 = the default, else code + 1) -> WideShape.ident;
// false when a layer asks for anything but tanh (code 0) or the identity (code 2)
inline bool wide_activations(int layer_activations, WideShape& s) {
    s.ident = 0;
    for (int l = 0; l < s.L; ++l) {
        const int f = (layer_activations >> (2 * l)) & 3;
        if (f == 3) s.ident |= 1 << l;            // RL_ACT_IDENTITY + 1
        else if (f != 0 && f != 1) return false;  // RL_ACT_RECTIFY + 1
    }
    return true;
}
}  // namespace rl
