"""HIP-kernel loss / gradient / Fisher-vector product for a GaussianMLPPolicy whose log-std is a NETWORK
(``adaptive_std=True`` or ``std_network=...``, rllab/policies/gaussian_mlp_policy.py:60-98; the configuration of the
reference's regression test tests/regression_tests/test_issue_3.py).

The flat parameter vector is [mean network | std network].  Both networks run on the matrix-core kernels as plain
functions on planes (``rl_mlp_forward`` / ``rl_mlp_backward``, csrc/policy_kernels.hip, modes OUT / OUT_TAN / BWD), the
diagonal-Gaussian head between them is ``rl_gaussian_head`` / ``rl_gaussian_fisher`` (csrc/gaussian_head_kernels.hip):

    loss, KL   : forward x 2 -> head (sums only)
    gradient   : forward x 2 -> head (sums + cotangents on the mean / log-std planes) -> backward x 2
    F v        : forward-with-tangent x 2 -> Fisher head (diagonal in (mean, log_std) at old == new) -> backward x 2

Everything else -- evaluation caching, the device CG, the line search writes, the sharded sums -- is inherited from
``FusedGaussianMLPOps``; this class only replaces its three passes.  Each network: two or three tanh layers of at most
128 units, run zero-padded to 32 / 64 / 128 per layer (exact: a padded unit has zero weights and bias, tanh(0) = 0;
policies/kernel_layout.py::mlp_pad_index; two equal layers of 32 / 64 on a HIP-native (obs, action) pair run one
wavefront per tile, every other shape the cooperative kernels of csrc/policy_wide_kernels.hip -- their modes OUT /
OUT_TAN / BWD), obs_dim <= 30, action_dim <= 8.
"""
import ctypes
import math

import torch

from rllab_amd import _lib
from rllab_amd.core.network import tanh
from rllab_amd.policies.fused_ops import FusedGaussianMLPOps
from rllab_amd.sampler import dist as D


class _IdentityLayout(object):
    """The optimizer-facing parameter space IS the policy's flat vector; the networks' zero padding happens per network,
    where their kernel-layout copies are made (``FusedAdaptiveStdOps.nets``)."""
    wide = False
    exact = True

    def __init__(self, policy):
        self.policy = policy
        self.P = self.P_pad = policy.flat_params.numel()

    def theta(self):
        return self.policy.flat_params.detach()

    def pack(self, vec):
        return vec

    def unpack(self, vec):
        return vec


def _net_ok(net, obs_dim, act_dim):
    from rllab_amd.policies.kernel_layout import layer_padded_sizes
    return (layer_padded_sizes(net.hidden_sizes) is not None and net.hidden_nonlinearity is tanh
            and net.output_nonlinearity is None and net.input_dim == obs_dim and net.output_dim == act_dim)


class FusedAdaptiveStdOps(FusedGaussianMLPOps):
    # its loss evaluation is a chain of launches over two networks (no single rl_policy_loss_kl to gate): the host reads
    # every line-search candidate, as the reference does
    device_line_search = False

    @staticmethod
    def supported(policy):
        if not getattr(policy, "state_dependent_std", False) or not policy.flat_params.is_cuda \
                or policy.flat_params.dtype != torch.float32:
            return False
        from rllab_amd.policies.kernel_layout import MAX_ACT_DIM, MAX_OBS_DIM
        if policy.obs_dim > MAX_OBS_DIM or policy.action_dim > MAX_ACT_DIM:
            return False
        return _net_ok(policy._mean_network, policy.obs_dim, policy.action_dim) and \
            _net_ok(policy._std_network, policy.obs_dim, policy.action_dim)

    def __init__(self, policy):
        self.policy = policy
        self.layout = _IdentityLayout(policy)
        do, da = policy.obs_dim, policy.action_dim
        from rllab_amd.policies.kernel_layout import layer_padded_sizes, mlp_identity_ones, mlp_layer_activations, mlp_pad_index
        self.acts, self.ones = [], []    # per network: layer_activations word of its kernel copy, positions of its constant ones
        self.nets = []                   # (offset, size, padded hidden triple) of [mean net, std net] in the flat vector
        self.pad = []                    # per network: (index of its real parameters in its padded copy or None, padded size)
        for net in (policy._mean_network, policy._std_network):
            off = net.params[0].offset
            hs = tuple(int(h) for h in net.hidden_sizes)
            Hs = layer_padded_sizes(hs)
            self.nets.append((off, net.end_offset - off, Hs + (0,) * (3 - len(Hs))))
            idx, p_pad = mlp_pad_index(do, hs, Hs, da)
            assert idx.size == net.end_offset - off
            self.acts.append(mlp_layer_activations(hs))          # (a one-hidden-layer network: tanh, then the identity W1 = I)
            self.ones.append(torch.as_tensor(mlp_identity_ones(do, hs, Hs), dtype=torch.long, device=policy.flat_params.device))
            self.pad.append((None if hs == Hs else torch.as_tensor(idx, dtype=torch.long, device=policy.flat_params.device),
                             int(p_pad)))
        assert self.nets[0][0] == 0 and self.nets[1][0] == self.nets[0][1]
        assert self.nets[1][0] + self.nets[1][1] == policy.flat_params.numel()
        self.dims = (do, da) + self.nets[0][2]
        self.n_kernel = self.layout.P_pad
        self.wide_kernels = False                        # (the base __init__ is not run: every attribute it sets is set here)
        self._ws = None
        self._loss_cache = None
        self._bound = {}
        self._acts = None
        self._acts_tag = None
        self._net_theta = [None, None]                   # [net params | Da zeros]: the kernels' policy layout
        self._theta_tag = None
        self._head_ws = None

    # the kernels read a policy-shaped parameter vector (log_std row at the end, ignored): a persistent padded copy
    # per network, refreshed when the parameters have moved
    def _thetas(self):
        pol = self.policy
        tag = pol.param_version()
        if self._theta_tag != tag:
            flat = pol.flat_params.detach()
            for i, (off, size, _) in enumerate(self.nets):
                idx, p_pad = self.pad[i]
                if self._net_theta[i] is None:       # (padded positions and the Da trailing floats stay zero for good)
                    self._net_theta[i] = torch.zeros(p_pad + pol.action_dim, dtype=torch.float32, device=flat.device)
                    if self.ones[i].numel():
                        self._net_theta[i][self.ones[i]] = 1.0
                self._scatter(i, self._net_theta[i], flat[off:off + size])
            self._theta_tag = tag
        return self._net_theta

    def _scatter(self, i, dst, src):
        """Real parameters of network i -> their places in its (padded) kernel-layout vector ``dst``."""
        idx, _ = self.pad[i]
        if idx is None:
            dst[:src.numel()].copy_(src)
        else:
            dst.index_copy_(0, idx, src.to(dst.dtype))

    def _gather(self, i, src):
        """... and back: the real entries of a kernel-layout vector of network i."""
        idx, _ = self.pad[i]
        return src[:self.nets[i][1]] if idx is None else src.index_select(0, idx)

    def accepts(self, inputs):
        """Per-sample old log_std planes [Da, B] (what a state-dependent std records)."""
        return inputs[0].is_cuda and inputs[4].dim() == 2 and inputs[4].shape[-1] == inputs[0].shape[-1] \
            and inputs[0].shape[-1] > 1

    def _workspace(self, device):
        if self._ws is None or self._ws.device != device:
            n = max(_lib.lib.rl_policy_workspace_bytes(self.dims[0], self.dims[1], h[0], h[1], h[2]) for _, _, h in self.nets)
            self._ws = torch.empty(n, dtype=torch.uint8, device=device)
            self._head_ws = torch.empty(_lib.lib.rl_gaussian_head_workspace_bytes(), dtype=torch.uint8, device=device)
        return self._ws

    def _batch(self, inputs):
        key = tuple(id(t) for t in inputs)
        hit = self._bound.get(key)
        if hit is not None:
            self._point_at_current_parameters(hit[0])
            return hit
        obs, act, adv, old_mean, old_ls, w, inv_count = inputs
        keep = [t.contiguous() for t in (obs, act, adv, old_mean, old_ls.float(), w)]
        obs, act, adv, old_mean, old_ls, w = keep
        B = obs.shape[-1]
        dev = obs.device
        f32 = dict(dtype=torch.float32, device=dev)
        da = self.dims[1]
        planes = dict(mean=torch.empty((da, B), **f32), lstd=torch.empty((da, B), **f32),
                      dmean=torch.empty((da, B), **f32), dlstd=torch.empty((da, B), **f32),
                      gmean=torch.empty((da, B), **f32), glstd=torch.empty((da, B), **f32))
        inv = float(inv_count)
        pol = self.policy
        log_min = math.log(pol.min_std) if pol.min_std is not None else -1e30
        structs = []
        for (_, _, h), acts in zip(self.nets, self.acts):
            structs.append(_lib.PolicyBatch(
                n_samples=B, obs_dim=self.dims[0], act_dim=da, hidden0=h[0], hidden1=h[1], hidden2=h[2], inv_count=inv,
                layer_activations=acts,
                log_min_std=log_min, theta=None, obs=obs.data_ptr(), actions=act.data_ptr(),
                advantages=adv.data_ptr(), old_means=old_mean.data_ptr(), old_log_std=old_ls.data_ptr(),
                weights=w.data_ptr()))
        # per-network scratch of the passes, allocated once per bound batch (ten Fisher-vector products per update
        # would otherwise allocate and fill four tensors each): the tangent in the kernels' policy layout
        # [network parameters | Da zeros] and the float64 gradient row rl_mlp_backward writes
        scratch = [dict(tangent=torch.zeros(p_pad + da, **f32),
                        grad=torch.empty(p_pad + da, dtype=torch.float64, device=dev)) for _, p_pad in self.pad]
        b = dict(structs=structs, planes=planes, log_min=log_min, B=B, scratch=scratch, tensors=dict(obs=obs, act=act, adv=adv,
                                                                                      old_mean=old_mean, old_ls=old_ls, w=w))
        self._point_at_current_parameters(b)
        if len(self._bound) >= 2:
            self._bound.clear()
        self._bound[key] = (b, keep + list(inputs), inv)
        return self._bound[key]

    def _point_at_current_parameters(self, b):
        for st, th in zip(b["structs"], self._thetas()):
            st.theta = th.data_ptr()

    # -- the three passes ---------------------------------------------------------------------------------------------
    def _forward(self, b, tangents=None):
        """mean / lstd planes (and their tangents in direction ``tangents`` = two kernel-layout float32 vectors)."""
        p, st = b["planes"], _lib.stream_ptr()
        ws = self._ws                                     # (the cooperative kernels keep their operand images there)
        for i, (name, dname) in enumerate((("mean", "dmean"), ("lstd", "dlstd"))):
            vec = None if tangents is None else _lib.ptr(tangents[i])
            dout = None if tangents is None else _lib.ptr(p[dname])
            _lib.check(_lib.lib.rl_mlp_forward_ws(ctypes.byref(b["structs"][i]), vec, _lib.ptr(ws), ws.numel(),
                                                  _lib.ptr(p[name]), dout, st), "rl_mlp_forward_ws")

    def _head(self, b, inv, out4, vpg=False, penalty=0.0, with_cotangents=False):
        p, t = b["planes"], b["tensors"]
        _lib.check(_lib.lib.rl_gaussian_head(
            b["B"], self.dims[1], _lib.ptr(p["mean"]), _lib.ptr(p["lstd"]), _lib.ptr(t["act"]), _lib.ptr(t["adv"]),
            _lib.ptr(t["old_mean"]), _lib.ptr(t["old_ls"]), _lib.ptr(t["w"]), inv, b["log_min"], int(vpg), float(penalty),
            _lib.ptr(p["gmean"]) if with_cotangents else None, _lib.ptr(p["glstd"]) if with_cotangents else None,
            _lib.ptr(self._head_ws), self._head_ws.numel(), _lib.ptr(out4), _lib.stream_ptr()), "rl_gaussian_head")

    def _backward(self, b, ws, out):
        """out [P] float64 <- [d mean-net | d std-net] of the cotangent planes."""
        p, st = b["planes"], _lib.stream_ptr()
        for i, gname in enumerate(("gmean", "glstd")):
            off, size, _ = self.nets[i]
            g = b["scratch"][i]["grad"]
            _lib.check(_lib.lib.rl_mlp_backward(ctypes.byref(b["structs"][i]), _lib.ptr(p[gname]), _lib.ptr(ws),
                                                ws.numel(), _lib.ptr(g), st), "rl_mlp_backward")
            out[off:off + size].copy_(self._gather(i, g))
        return out

    def _loss_eval(self, inputs):
        tag = self._eval_point(inputs)
        c = self._loss_cache
        if c is not None and c["tag"] == tag:
            return c
        b, keep, inv = self._batch(inputs)
        self._workspace(keep[0].device)
        out = torch.empty(4, dtype=torch.float64, device=keep[0].device)
        self._forward(b)
        self._head(b, inv, out)
        return self._loss_record(tag, out, inv)

    def loss_grad(self, inputs, vpg=False, keep_activations=False, with_loss=False, penalty=0.0):
        b, keep, inv = self._batch(inputs)
        ws = self._workspace(keep[0].device)
        out = torch.empty(self.n_kernel, dtype=torch.float64, device=keep[0].device)
        out4 = torch.empty(4, dtype=torch.float64, device=keep[0].device)
        tag = self._eval_point(inputs)
        self._forward(b)
        self._head(b, inv, out4, vpg=vpg, penalty=penalty, with_cotangents=True)
        if with_loss and not (self._loss_cache is not None and self._loss_cache["tag"] == tag):
            self._loss_record(tag, out4, inv)
        self._last_out4 = out4
        return D.update_sum_(self._backward(b, ws, out))

    def value_and_grad(self, inputs, penalty=0.0):
        g = self.loss_grad(inputs, penalty=penalty)
        _, _, inv = self._batch(inputs)
        sums = D.all_reduce_sum_(self._last_out4[:3].clone()).cpu().numpy()
        idx = self.policy._flat_index(trainable=True)
        host = g.cpu().numpy()
        return float((-sums[0] + penalty * sums[1]) * inv), (host if idx is None else host[idx.cpu().numpy()]).copy()

    def _fvp_into(self, b, ws, vec32, out, inputs=None):
        """F vec: tangents of both networks, the (diagonal) Fisher metric of the head, back through both networks."""
        da = self.dims[1]
        tangents = []
        for i, ((off, size, _), sc) in enumerate(zip(self.nets, b["scratch"])):
            self._scatter(i, sc["tangent"], vec32[off:off + size])      # (padded positions and the Da trailing floats stay zero)
            tangents.append(sc["tangent"])
        self._forward(b, tangents)
        p, t_ = b["planes"], b["tensors"]
        _lib.check(_lib.lib.rl_gaussian_fisher(
            b["B"], da, _lib.ptr(p["dmean"]), _lib.ptr(p["dlstd"]), _lib.ptr(p["lstd"]), _lib.ptr(t_["w"]),
            float(b["structs"][0].inv_count), b["log_min"], _lib.ptr(p["gmean"]), _lib.ptr(p["glstd"]),
            _lib.stream_ptr()), "rl_gaussian_fisher")
        return D.update_sum_(self._backward(b, ws, out))

    def fvp_variant(self, inputs):
        """The networks-on-planes products run on the f32 matrix instructions (variant 0)."""
        return 0

    def _cg_loop(self, b, ws, inputs, cg_iters, reg_coeff, residual_tol, x, r, p, p32, z, scal, st):
        for _ in range(cg_iters):
            self._fvp_into(b, ws, p32, z, inputs)
            _lib.check(_lib.lib.rl_cg_step(self.n_kernel, _lib.ptr(z), float(reg_coeff), float(residual_tol),
                                           _lib.ptr(x), _lib.ptr(r), _lib.ptr(p), _lib.ptr(p32), _lib.ptr(scal), st),
                       "rl_cg_step")
