"""How a GaussianMLPPolicy's flat parameter vector is presented to the HIP kernels.

The reference's ``GaussianMLPPolicy(hidden_sizes=...)`` is free-form (rllab/policies/gaussian_mlp_policy.py:21-38,
rllab/core/network.py:36-101; its MuJoCo experiments use (100, 50, 25)).  The kernels come in two families:

  * two EQUAL hidden layers of H = 32 or 64 tanh units, one wavefront per 32-sample tile with everything on chip
    (csrc/policy_kernels.hip, the lane-group rollouts of csrc/env_kernels.hip) -- any two-layer policy with both
    sizes <= 64 runs there;
  * two or three hidden layers of 32 / 64 / 128 units EACH, four wavefronts cooperating on a tile
    (csrc/policy_wide_kernels.hip, ``rollout_wide_kernel``) -- everything else up to 128 units per layer.

Either way a policy whose sizes are not tile sizes runs on the next tile size by ZERO PADDING:

    W_l [in, h_l] -> [in_pad, H_l]      b_l [h_l] -> [H_l]      Wout [h_last, Da] -> [H_last, Da]

A padded unit has zero input weights and zero bias, so its activation is tanh(0) = 0, it feeds nothing (zero
outgoing weights), and every gradient / Fisher-vector-product entry of a padded parameter is exactly zero; what the
real parameters see is the unpadded arithmetic plus exact zeros.  The policy keeps its parameters in the
reference's layout (``flat_params``, what ``get_param_values`` / snapshots / the optimizers see); ``KernelLayout``
owns the padded copy, refreshes it when the parameters have moved, scatters vectors into the padded space
(``pack``) and gathers results back (``unpack``).  For sizes that are tile sizes already it is the identity and hands
out ``flat_params`` itself.
"""
import numpy as np
import torch

TILE_SIZES = (32, 64)            # the equal-width two-layer kernels
WIDE_TILE_SIZES = (32, 64, 128)  # per-layer widths of the cooperative kernels
MAX_OBS_DIM = 30                 # obs_dim + 1 (bias slot) must fit one 32-row input tile
MAX_ACT_DIM = 8
ACT_TANH, ACT_RECTIFY, ACT_IDENTITY = 0, 1, 2      # rl_activation (include/rllab_amd.h)


def tile_for(hidden_sizes):
    """Hidden width H of the equal-width two-layer kernel that runs ``hidden_sizes``, or None."""
    hs = tuple(int(h) for h in hidden_sizes)
    if len(hs) != 2 or min(hs) < 1:
        return None
    for H in TILE_SIZES:
        if max(hs) <= H:
            return H
    return None


def padded_sizes(hidden_sizes):
    """The kernels' hidden widths for ``hidden_sizes``: (H, H) of the equal-width family when it applies, else every
    layer on its own next size of 32 / 64 / 128 (two or three layers); None when no kernel runs the net.  ONE hidden
    layer of at most 128 units runs as (H, H) with an identity second layer (``KernelLayout.identity_layer``)."""
    hs = tuple(int(h) for h in hidden_sizes)
    if len(hs) == 1 and 1 <= hs[0] <= WIDE_TILE_SIZES[-1]:
        # (65 .. 128 units: the cooperative family's (128, 128) shape, its second layer the identity as well)
        H = next(t for t in WIDE_TILE_SIZES if hs[0] <= t)
        return (H, H)
    H = tile_for(hs)
    if H is not None:
        return (H, H)
    if len(hs) not in (2, 3) or min(hs) < 1 or max(hs) > WIDE_TILE_SIZES[-1]:
        return None
    return tuple(next(t for t in WIDE_TILE_SIZES if h <= t) for h in hs)


def layer_padded_sizes(hidden_sizes):
    """Every layer on its own next size of 32 / 64 / 128 (two or three layers); None when no kernel runs the net.
    (The networks-on-planes entry points -- adaptive_std -- take any such triple.)  ONE hidden layer of at most 128 units
    runs as (H, H) with the identity as its second layer (``mlp_identity_ones`` / ``mlp_layer_activations``)."""
    hs = tuple(int(h) for h in hidden_sizes)
    if len(hs) == 1 and 1 <= hs[0] <= WIDE_TILE_SIZES[-1]:
        H = next(t for t in WIDE_TILE_SIZES if hs[0] <= t)
        return (H, H)
    if len(hs) not in (2, 3) or min(hs) < 1 or max(hs) > WIDE_TILE_SIZES[-1]:
        return None
    return tuple(next(t for t in WIDE_TILE_SIZES if h <= t) for h in hs)


def mlp_layer_activations(hidden_sizes):
    """``layer_activations`` word (2 bits per layer: code + 1) of the KERNEL copy of a tanh MLP: 0 (tanh layers) unless the net
    has one hidden layer -- then tanh, identity."""
    return ((ACT_TANH + 1) | ((ACT_IDENTITY + 1) << 2)) if len(tuple(hidden_sizes)) == 1 else 0


def mlp_identity_ones(in_dim, hidden_sizes, padded):
    """Positions, in the padded kernel-layout vector, of the ones of the identity second layer W1 = I that the kernel copy of a
    ONE-hidden-layer MLP carries as constants (empty otherwise)."""
    hs, Hs = tuple(int(h) for h in hidden_sizes), tuple(int(h) for h in padded)
    if len(hs) != 1:
        return np.zeros(0, dtype=np.int64)
    H = Hs[0]
    oW1 = in_dim * H + H
    return oW1 + np.arange(H) * (H + 1)


def mlp_pad_index(in_dim, hidden_sizes, padded, out_dim):
    """Position of every parameter of an MLP (flat order W_0, b_0, ..., W_out, b_out; W_l is [in, out] row-major) inside
    the same net with its hidden widths zero-padded to ``padded``; returns (index array [P_real], P_pad)."""
    hs, Hs = tuple(int(h) for h in hidden_sizes), tuple(int(h) for h in padded)
    if len(hs) == 1:
        # (in -> h -> out) inside (in -> H -> H [identity] -> out): W0, b0, then the constants W1 = I, b1 = 0, then Wout, bout
        h, H = hs[0], Hs[0]
        oWo = in_dim * H + H + H * H + H
        idx = [(np.arange(in_dim)[:, None] * H + np.arange(h)[None, :]).reshape(-1), in_dim * H + np.arange(h),
               oWo + (np.arange(h)[:, None] * out_dim + np.arange(out_dim)[None, :]).reshape(-1),
               oWo + H * out_dim + np.arange(out_dim)]
        return np.concatenate(idx), oWo + H * out_dim + out_dim
    ins_real, ins_pad = (in_dim,) + hs, (in_dim,) + Hs
    idx, off = [], 0
    for l in range(len(Hs)):
        rows, cols, cols_pad = ins_real[l], hs[l], Hs[l]
        idx.append(off + (np.arange(rows)[:, None] * cols_pad + np.arange(cols)[None, :]).reshape(-1))
        off += ins_pad[l] * cols_pad
        idx.append(off + np.arange(cols))
        off += cols_pad
    idx.append(off + (np.arange(hs[-1])[:, None] * out_dim + np.arange(out_dim)[None, :]).reshape(-1))
    off += Hs[-1] * out_dim
    idx.append(off + np.arange(out_dim))
    off += out_dim
    return np.concatenate(idx), off




class KernelLayout(object):
    def __init__(self, policy):
        self.policy = policy
        hs = tuple(int(h) for h in policy.hidden_sizes)
        do, da = policy.obs_dim, policy.action_dim
        Hs = padded_sizes(hs)
        assert Hs is not None
        # hidden activations per KERNEL layer: the policy's nonlinearity, and the identity for the second layer a
        # one-hidden-layer policy gets in its kernel copy (W1 = I, b1 = 0: h1 = h0 exactly, products with 1 and 0)
        from rllab_amd.policies.gaussian_mlp_policy import is_rectify
        code = ACT_RECTIFY if is_rectify(getattr(policy, "hidden_nonlinearity", None)) else ACT_TANH
        self.identity_layer = len(hs) == 1
        self.layer_codes = (code, ACT_IDENTITY) if self.identity_layer else (code,) * len(Hs)
        if self.identity_layer:
            self._init_one_layer(policy, hs[0], Hs[0], do, da)
            return
        self.hidden = Hs                         # what the kernels are told (rl_policy_batch.hidden0..2)
        self.H = Hs[0]                           # kept for callers of the equal-width family
        self.wide = not (len(Hs) == 2 and Hs[0] == Hs[1] and Hs[0] in TILE_SIZES)
        self.exact = (hs == Hs)
        self.P = policy.flat_params.numel()
        ins_real, ins_pad = (do,) + hs, (do,) + Hs
        self.P_pad = sum(ins_pad[l] * Hs[l] + Hs[l] for l in range(len(Hs))) + Hs[-1] * da + 2 * da
        if self.exact:
            assert self.P == self.P_pad
            self.index = None
            return
        # position of every real parameter inside the padded vector, in the reference's flat order
        idx, off = [], 0
        for l in range(len(Hs)):
            rows, cols, cols_pad = ins_real[l], hs[l], Hs[l]
            idx.append(off + (np.arange(rows)[:, None] * cols_pad + np.arange(cols)[None, :]).reshape(-1))   # W_l
            off += ins_pad[l] * cols_pad
            idx.append(off + np.arange(cols))                                                               # b_l
            off += cols_pad
        idx.append(off + (np.arange(hs[-1])[:, None] * da + np.arange(da)[None, :]).reshape(-1))            # Wout
        off += Hs[-1] * da
        idx.append(off + np.arange(da))                                                                     # bout
        off += da
        idx.append(off + np.arange(da))                                                                     # log_std
        off += da
        assert off == self.P_pad
        idx = np.concatenate(idx)
        assert idx.size == self.P and len(set(idx.tolist())) == self.P
        dev = policy.flat_params.device
        self.index = torch.as_tensor(idx, dtype=torch.long, device=dev)
        self._theta = torch.zeros(self.P_pad, dtype=torch.float32, device=dev)
        self._tag = None

    def _init_one_layer(self, policy, h, H, do, da):
        """hidden_sizes = (h,): kernel net (do -> H -> H -> da) whose second layer is the identity.  Real parameters
        W0 [do, h], b0 [h], Wout [h, da], bout, log_std sit at their padded positions; W1 = I and b1 = 0 are constants of
        the kernel copy (never trained: ``unpack`` gathers the real entries only, ``pack`` leaves their tangents zero)."""
        self.hidden, self.H = (H, H), H
        self.wide, self.exact = H not in TILE_SIZES, False      # 128: the cooperative kernels (identity layers since round 6)
        self.P = policy.flat_params.numel()
        oW0, ob0 = 0, do * H
        oW1 = ob0 + H
        ob1 = oW1 + H * H
        oWo = ob1 + H
        obo = oWo + H * da
        ols = obo + da
        self.P_pad = ols + da
        idx = [oW0 + (np.arange(do)[:, None] * H + np.arange(h)[None, :]).reshape(-1), ob0 + np.arange(h),
               oWo + (np.arange(h)[:, None] * da + np.arange(da)[None, :]).reshape(-1), obo + np.arange(da),
               ols + np.arange(da)]
        idx = np.concatenate(idx)
        assert idx.size == self.P and len(set(idx.tolist())) == self.P
        dev = policy.flat_params.device
        self.index = torch.as_tensor(idx, dtype=torch.long, device=dev)
        self._theta = torch.zeros(self.P_pad, dtype=torch.float32, device=dev)
        self._theta[oW1 + torch.arange(H, device=dev) * (H + 1)] = 1.0        # W1 = I
        self._tag = None
        # 1 at the real parameters' positions: the constants W1 = I, b1 = 0 are not parameters, but the kernels differentiate
        # with respect to every entry of their vector -- products and gradients are confined to the real subspace with this
        self.real_mask = torch.zeros(self.P_pad, dtype=torch.float64, device=dev)
        self.real_mask[self.index] = 1.0

    @property
    def layer_activations(self):
        """``rl_policy_batch.layer_activations`` / ``rl_rollout_args.layer_activations`` of this layout (0 = tanh layers)."""
        if all(c == ACT_TANH for c in self.layer_codes):
            return 0
        return sum((int(c) + 1) << (2 * l) for l, c in enumerate(self.layer_codes))

    @property
    def hidden3(self):
        """(hidden0, hidden1, hidden2) as the C ABI takes them (hidden2 = 0: two layers)."""
        return tuple(self.hidden) + (0,) * (3 - len(self.hidden))

    # -- the parameter vector the kernels read --------------------------------------------------------------------
    def theta(self):
        """Contiguous float32 device vector in the kernels' layout holding the CURRENT parameters (for padded
        layouts a persistent buffer, refreshed in place -- its address never changes)."""
        flat = self.policy.flat_params.detach()
        if self.exact:
            return flat
        tag = self.policy.param_version()
        if tag != self._tag:
            self._theta.index_copy_(0, self.index, flat)
            self._tag = tag
        return self._theta

    # -- vectors -------------------------------------------------------------------------------------------------
    def pack(self, vec):
        """Real-layout vector [P] -> kernel layout [P_pad] (zeros at the padded positions); same dtype."""
        if self.exact:
            return vec
        out = torch.zeros(self.P_pad, dtype=vec.dtype, device=vec.device)
        out.index_copy_(0, self.index, vec)
        return out

    def unpack(self, vec_pad):
        """Kernel-layout vector [P_pad] -> real layout [P]."""
        if self.exact:
            return vec_pad
        return vec_pad.index_select(0, self.index)
