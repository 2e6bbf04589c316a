"""How a GaussianMLPPolicy's flat parameter vector is presented to the HIP kernels.

The fused rollout and update kernels (csrc/env_kernels.hip, csrc/policy_kernels.hip) are instantiated for two
equal hidden layers of H = 32 or H = 64 tanh units -- the tile sizes of the matrix cores.  The reference's
``GaussianMLPPolicy(hidden_sizes=...)`` is free-form (rllab/policies/gaussian_mlp_policy.py:24), so any two-layer
tanh policy with hidden sizes (h0, h1), h0, h1 <= 64, is run on the kernel of the next tile size by ZERO PADDING:

    W0 [Do, h0] -> [Do, H]   b0 [h0] -> [H]   W1 [h0, h1] -> [H, H]   b1 [h1] -> [H]   Wout [h1, Da] -> [H, Da]

A padded unit has zero input weights and zero bias, so its activation is tanh(0) = 0, it feeds nothing (zero
outgoing weights), and every gradient / Fisher-vector-product entry of a padded parameter is exactly zero; what the
real parameters see is the unpadded arithmetic plus exact zeros.  The policy keeps its parameters in the
reference's layout (``flat_params``, what ``get_param_values`` / snapshots / the optimizers see); ``KernelLayout``
owns the padded copy, refreshes it when the parameters have moved, scatters vectors into the padded space
(``pack``) and gathers results back (``unpack``).  For (32, 32) and (64, 64) it is the identity and hands out
``flat_params`` itself.
"""
import numpy as np
import torch

TILE_SIZES = (32, 64)


def tile_for(hidden_sizes):
    """Hidden width H of the kernel that runs ``hidden_sizes`` (two layers), or None."""
    hs = tuple(int(h) for h in hidden_sizes)
    if len(hs) != 2 or min(hs) < 1:
        return None
    for H in TILE_SIZES:
        if max(hs) <= H:
            return H
    return None


class KernelLayout(object):
    def __init__(self, policy):
        self.policy = policy
        h0, h1 = (int(h) for h in policy.hidden_sizes)
        do, da = policy.obs_dim, policy.action_dim
        H = tile_for((h0, h1))
        assert H is not None
        self.H = H
        self.exact = (h0 == H and h1 == H)
        self.P = policy.flat_params.numel()
        self.P_pad = do * H + H + H * H + H + H * da + 2 * da
        if self.exact:
            assert self.P == self.P_pad
            self.index = None
            return
        # position of every real parameter inside the padded vector, in the reference's flat order
        off_b0 = do * H
        off_w1 = off_b0 + H
        off_b1 = off_w1 + H * H
        off_w2 = off_b1 + H
        off_b2 = off_w2 + H * da
        off_ls = off_b2 + da
        idx = [
            (np.arange(do)[:, None] * H + np.arange(h0)[None, :]).reshape(-1),                 # W0 [Do, h0]
            off_b0 + np.arange(h0),                                                           # b0
            off_w1 + (np.arange(h0)[:, None] * H + np.arange(h1)[None, :]).reshape(-1),       # W1 [h0, h1]
            off_b1 + np.arange(h1),                                                           # b1
            off_w2 + (np.arange(h1)[:, None] * da + np.arange(da)[None, :]).reshape(-1),      # Wout [h1, Da]
            off_b2 + np.arange(da),                                                           # bout
            off_ls + np.arange(da),                                                           # log_std
        ]
        idx = np.concatenate(idx)
        assert idx.size == self.P and len(set(idx.tolist())) == self.P
        dev = policy.flat_params.device
        self.index = torch.as_tensor(idx, dtype=torch.long, device=dev)
        self._theta = torch.zeros(self.P_pad, dtype=torch.float32, device=dev)
        self._tag = None

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
