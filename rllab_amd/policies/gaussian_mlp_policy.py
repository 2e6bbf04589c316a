"""GaussianMLPPolicy (API of rllab/policies/gaussian_mlp_policy.py:20-161).

mean = MLP(tanh hidden layers, linear output), log_std = a free trainable vector
(state independent).  The reference builds this with Lasagne (``MLP``,
rllab/core/network.py:36-101; ``ParamLayer``, rllab/core/lasagne_layers.py:9-30);
here all parameters live in ONE flat float32 device vector in the reference's
flat order W0,b0,W1,b1,...,Wout,bout,log_std with W stored [in, out] row-major
(SURVEY.md section 8), which is exactly what the fused rollout / update kernels
read.  Initialisation follows the reference: Glorot-uniform W
(U(+-sqrt(6/(fan_in+fan_out)))), zero biases, log_std = log(init_std).
"""
import numpy as np
import torch

from rllab_amd.core.network import MLP, rectify
from rllab_amd.core.parameterized import Param
from rllab_amd.core.serializable import Serializable
from rllab_amd.distributions.diagonal_gaussian import DiagonalGaussian
from rllab_amd.misc import logger
from rllab_amd.policies.base import StochasticPolicy
from rllab_amd.spaces import Box

tanh = torch.tanh


def is_rectify(f):
    """rllab's rectify (core/network.py, lasagne.nonlinearities.rectify in the reference) under any of its torch names."""
    return f is rectify or f is torch.relu or f is torch.nn.functional.relu


def _default_device():
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() \
        else torch.device("cpu")


class GaussianMLPPolicy(StochasticPolicy, Serializable):
    def __init__(self, env_spec, hidden_sizes=(32, 32), learn_std=True, init_std=1.0,
                 adaptive_std=False, std_share_network=False, std_hidden_sizes=(32, 32),
                 min_std=1e-6, std_hidden_nonlinearity=tanh, hidden_nonlinearity=tanh,
                 output_nonlinearity=None, mean_network=None, std_network=None,
                 dist_cls=DiagonalGaussian):
        Serializable.quick_init(self, locals())
        assert isinstance(env_spec.action_space, Box)
        StochasticPolicy.__init__(self, env_spec)
        obs_dim = env_spec.observation_space.flat_dim
        action_dim = env_spec.action_space.flat_dim
        self.obs_dim, self.action_dim = obs_dim, action_dim
        self.learn_std = learn_std
        self.min_std = min_std
        self._dist = dist_cls(action_dim)

        # mean network (reference :60-69); a custom ``mean_network`` is an MLP description (core/network.py) whose
        # layer sizes / nonlinearities are taken over -- its parameters are re-created in this policy's flat vector
        if mean_network is not None:
            assert mean_network.input_dim == obs_dim and mean_network.output_dim == action_dim
            hidden_sizes, hidden_nonlinearity = mean_network.hidden_sizes, mean_network.hidden_nonlinearity
            output_nonlinearity = mean_network.output_nonlinearity
        self.hidden_sizes = tuple(int(h) for h in hidden_sizes)
        self.hidden_nonlinearity = hidden_nonlinearity
        self.output_nonlinearity = output_nonlinearity
        self._mean_network = MLP((obs_dim,), action_dim, self.hidden_sizes, hidden_nonlinearity, output_nonlinearity,
                                 offset=0)
        params = list(self._mean_network.params)
        off = self._mean_network.end_offset
        # log-std head (:73-92): a network of its own on the same input (adaptive_std / std_network), or one free
        # row.  ``std_share_network`` is accepted and, as in the reference's constructor body, has no effect.
        self._std_network = None
        self._log_std_param = None
        if std_network is not None or adaptive_std:
            if std_network is not None:
                assert std_network.input_dim == obs_dim and std_network.output_dim == action_dim
                std_hidden_sizes, std_hidden_nonlinearity = std_network.hidden_sizes, std_network.hidden_nonlinearity
                std_out = std_network.output_nonlinearity
            else:
                std_out = None
            self._std_network = MLP((obs_dim,), action_dim, tuple(std_hidden_sizes), std_hidden_nonlinearity, std_out,
                                    name="std", offset=off)
            params += self._std_network.params
            off = self._std_network.end_offset
        else:
            ls = Param("output_log_std.param", (action_dim,), off, trainable=learn_std, regularizable=False)
            off += ls.size
            params.append(ls)
            self._log_std_param = ls
        self._params = params
        for p in params:
            p._owner = self

        # host-side init with np.random (so CPU oracle and GPU share theta under a seed): Glorot-uniform weights,
        # zero biases (network.py:38-39), log_std row = log(init_std) (:88-94)
        flat = np.zeros(off, dtype=np.float32)
        self._mean_network.init_values(flat)
        if self._std_network is not None:
            self._std_network.init_values(flat)
        else:
            ls = self._log_std_param
            flat[ls.offset:ls.offset + ls.size] = np.log(init_std)
        self.flat_params = torch.tensor(flat, dtype=torch.float32, device=_default_device())

    # -- Parameterized ----------------------------------------------------------
    def get_params_internal(self, **tags):
        return [p for p in self._params if all(p.tags.get(k, False) == v for k, v in tags.items())]

    # -- forward ------------------------------------------------------------------
    @property
    def state_dependent_std(self):
        """True with a log-std network (adaptive_std / std_network): agent_info["log_std"] then varies per sample."""
        return self._std_network is not None

    @property
    def fusable(self):
        """True when the in-kernel policy (tanh hidden layers, linear output, one free log_std row) is this policy."""
        return ((self.hidden_nonlinearity is tanh or is_rectify(self.hidden_nonlinearity)) and self.output_nonlinearity is None
                and not self.state_dependent_std)

    @property
    def vectorized(self):
        return True

    # -- what the HIP kernels read ---------------------------------------------------------------------------------
    def param_version(self):
        """Changes whenever the parameters do: torch's in-place version counter plus the writes our own kernels
        make through raw pointers (rl_line_search_point, rl_adam_step), which torch cannot see."""
        return (self.flat_params._version, getattr(self, "_raw_writes", 0))

    def note_raw_write(self):
        self._raw_writes = getattr(self, "_raw_writes", 0) + 1

    def kernel_layout(self):
        """``KernelLayout`` (policies/kernel_layout.py) when the fused kernels can run this policy -- two or three
        tanh hidden layers of at most 128 units (zero-padded to the kernels' tile sizes), linear output, learned
        state-independent std, float32 parameters on the device, observation / action widths the kernels take --
        else None."""
        if not hasattr(self, "_kernel_layout"):
            from rllab_amd.policies.kernel_layout import MAX_ACT_DIM, MAX_OBS_DIM, KernelLayout, padded_sizes
            Hs = padded_sizes(self.hidden_sizes)
            ok = (self.fusable and Hs is not None and self.flat_params.is_cuda
                  and self.flat_params.dtype == torch.float32
                  and self.obs_dim <= MAX_OBS_DIM and self.action_dim <= MAX_ACT_DIM)
            # rectify layers: the equal-width two-layer kernels of the HIP-native (obs, action) pairs only (the cooperative
            # family evaluates tanh and identity layers); the identity layer of a one-hidden-layer policy: those, or the
            # cooperative family's (128, 128) shape when the layer is 65 .. 128 tanh units wide
            if ok and (is_rectify(self.hidden_nonlinearity) or len(tuple(self.hidden_sizes)) == 1):
                from rllab_amd.policies.fused_ops import FusedGaussianMLPOps
                narrow = (len(Hs) == 2 and Hs[0] == Hs[1] and Hs[0] in (32, 64)
                          and (self.obs_dim, self.action_dim) in FusedGaussianMLPOps.NARROW_PAIRS)
                one_wide = (len(tuple(self.hidden_sizes)) == 1 and Hs == (128, 128)
                            and not is_rectify(self.hidden_nonlinearity))
                ok = narrow or one_wide
            self._kernel_layout = KernelLayout(self) if ok else None
        return self._kernel_layout

    def why_no_kernel_layout(self):
        """One sentence naming what keeps this policy off the fused kernels (``kernel_layout() is None``), or None.
        The sampler and the algorithms log it once, so that a 5-20x slower path is never taken silently."""
        if self.kernel_layout() is not None:
            return None
        from rllab_amd.policies.kernel_layout import MAX_ACT_DIM, MAX_OBS_DIM, padded_sizes
        hs = tuple(int(h) for h in self.hidden_sizes)
        if self.state_dependent_std:
            return "the log-std is a network (adaptive_std / std_network): the two-network kernels apply instead"
        if self.hidden_nonlinearity is not tanh and not is_rectify(self.hidden_nonlinearity):
            return "hidden_nonlinearity is %s (the kernels evaluate tanh and rectify layers)" % getattr(
                self.hidden_nonlinearity, "__name__", repr(self.hidden_nonlinearity))
        if self.output_nonlinearity is not None:
            return "output_nonlinearity is not None (the kernels' output layer is linear)"
        if padded_sizes(hs) is None:
            if len(hs) == 1:
                return "hidden_sizes=%r: one hidden layer runs on the kernels up to 128 units" % (hs,)
            if len(hs) not in (2, 3):
                return "hidden_sizes=%r has %d hidden layers (the kernels run one, two or three)" % (hs, len(hs))
            return "hidden_sizes=%r has a layer wider than 128 units" % (hs,)
        if is_rectify(self.hidden_nonlinearity) or len(hs) == 1:
            return ("rectify layers / one hidden layer run on the (32, 32) / (64, 64) kernels of the HIP-native (obs, action) "
                    "pairs: hidden_sizes=%r, obs_dim %d, action_dim %d is not one of them" % (hs, self.obs_dim, self.action_dim))
        if not self.flat_params.is_cuda or self.flat_params.dtype != torch.float32:
            return "the parameters are not float32 on a HIP device"
        if self.obs_dim > MAX_OBS_DIM or self.action_dim > MAX_ACT_DIM:
            return "obs_dim %d / action_dim %d exceed the kernels' %d / %d" % (self.obs_dim, self.action_dim, MAX_OBS_DIM,
                                                                             MAX_ACT_DIM)
        return "no kernel layout"

    def rollout_networks(self):
        """For a policy with a log-std NETWORK (adaptive_std / std_network) whose two networks the rollout kernels take
        -- two or three tanh hidden layers of at most 128 units each (zero-padded to 32 / 64 / 128 per layer), linear
        outputs, float32 parameters on the device: ``(theta_mean, hidden3_mean, theta_std, hidden3_std, layer_activations_mean,
        layer_activations_std)`` with each theta in the kernels' policy layout [network parameters | action_dim unused floats] (persistent buffers, refreshed when the
        parameters have moved).  None otherwise (such policies are sampled through the per-transition loop)."""
        if not self.state_dependent_std:
            return None
        if not hasattr(self, "_rollout_nets"):
            from rllab_amd.policies.kernel_layout import (MAX_ACT_DIM, MAX_OBS_DIM, layer_padded_sizes, mlp_identity_ones,
                                                         mlp_layer_activations, mlp_pad_index)
            nets = (self._mean_network, self._std_network)
            ok = (self.flat_params.is_cuda and self.flat_params.dtype == torch.float32
                  and self.obs_dim <= MAX_OBS_DIM and self.action_dim <= MAX_ACT_DIM
                  and all(n.hidden_nonlinearity is tanh and n.output_nonlinearity is None
                          and layer_padded_sizes(n.hidden_sizes) is not None for n in nets))
            self._rollout_nets = None
            if ok:
                dev = self.flat_params.device
                spans = [(n.params[0].offset, n.end_offset - n.params[0].offset) for n in nets]
                pads, hid, bufs, acts = [], [], [], []
                for n in nets:          # every layer zero-padded to 32 / 64 / 128 (exact: tanh(0) = 0); a one-hidden-layer
                    hs = tuple(int(h) for h in n.hidden_sizes)      # network runs with the identity as its second layer
                    Hs = layer_padded_sizes(hs)
                    idx, p_pad = mlp_pad_index(self.obs_dim, hs, Hs, self.action_dim)
                    pads.append(None if hs == Hs else torch.as_tensor(idx, dtype=torch.long, device=dev))
                    hid.append(Hs + (0,) * (3 - len(Hs)))
                    buf = torch.zeros(p_pad + self.action_dim, dtype=torch.float32, device=dev)
                    ones = mlp_identity_ones(self.obs_dim, hs, Hs)
                    if ones.size:
                        buf[torch.as_tensor(ones, dtype=torch.long, device=dev)] = 1.0          # W1 = I: constants for good
                    bufs.append(buf)
                    acts.append(mlp_layer_activations(hs))
                self._rollout_nets = dict(spans=spans, bufs=bufs, hidden=hid, pads=pads, acts=acts, tag=None)
        r = self._rollout_nets
        if r is None:
            return None
        tag = self.param_version()
        if r["tag"] != tag:
            flat = self.flat_params.detach()
            for (off, size), buf, idx in zip(r["spans"], r["bufs"], r["pads"]):
                if idx is None:
                    buf[:size].copy_(flat[off:off + size])
                else:
                    buf.index_copy_(0, idx, flat[off:off + size])
            r["tag"] = tag
        return r["bufs"][0], r["hidden"][0], r["bufs"][1], r["hidden"][1], r["acts"][0], r["acts"][1]

    def effective_log_std(self, flat=None):
        if self._log_std_param is None:
            raise AttributeError("this policy's log_std depends on the observation (adaptive_std): use "
                                 "log_std_planes(obs) / dist_info_planes(obs)")
        flat = self.flat_params if flat is None else flat
        ls = self._log_std_param.view(flat)
        if self.min_std is not None:
            ls = ls.clamp_min(float(np.log(self.min_std)))   # scalar bound: no host -> device copy per call
        return ls

    def recorded_log_std(self):
        """The log_std row a rollout records as agent_info (the "old" distribution of the update): always a
        COPY -- with ``min_std=None`` ``effective_log_std`` is a view into ``flat_params``, and the line search /
        Adam step rewrite that vector in place, which would make old and new log_std the same tensor."""
        if self.state_dependent_std:
            return None                      # per-sample log_std planes are recorded instead (sampler)
        return self.effective_log_std().detach().clone()

    def mean_planes(self, obs_planes, flat=None):
        """obs [Do, B] -> mean [Da, B] ("planes": feature axis first, the engine's layout)."""
        flat = self.flat_params if flat is None else flat
        return self._mean_network.forward_planes(obs_planes, flat)

    def log_std_planes(self, obs_planes, flat=None):
        """obs [Do, B] -> log_std [Da, B] (a broadcastable [Da, 1] column when the std is state independent),
        floored at log(min_std) (reference :100-101,120-121)."""
        flat = self.flat_params if flat is None else flat
        if self._std_network is None:
            return self.effective_log_std(flat)[:, None]
        ls = self._std_network.forward_planes(obs_planes, flat)
        if self.min_std is not None:
            ls = ls.clamp_min(float(np.log(self.min_std)))
        return ls

    def dist_info_planes(self, obs_planes, flat=None):
        return dict(mean=self.mean_planes(obs_planes, flat), log_std=self.log_std_planes(obs_planes, flat))

    def dist_info_sym(self, obs_var, state_info_vars=None):
        """[B, Do] tensor -> dict(mean [B, Da], log_std [B, Da]) (reference :118-122)."""
        obs_var = torch.as_tensor(obs_var, dtype=self.flat_params.dtype, device=self.flat_params.device)
        planes = obs_var.t()
        mean = self.mean_planes(planes).t()
        log_std = self.log_std_planes(planes).t().expand_as(mean)
        return dict(mean=mean, log_std=log_std)

    def dist_info(self, obs, state_infos=None):
        with torch.no_grad():
            d = self.dist_info_sym(np.asarray(obs))
        return {k: v.cpu().numpy().astype(np.float64) for k, v in d.items()}

    def get_action(self, observation):
        flat_obs = self.observation_space.flatten(observation)
        d = self.dist_info(flat_obs[None, :])
        mean, log_std = d["mean"][0], d["log_std"][0]
        rnd = np.random.normal(size=mean.shape)
        action = rnd * np.exp(log_std) + mean
        return action, dict(mean=mean, log_std=log_std)

    def get_actions(self, observations):
        """[n, Do] numpy array or device tensor -> (actions, dict(mean, log_std)), same container kind."""
        if torch.is_tensor(observations):
            with torch.no_grad():
                d = self.dist_info_sym(observations)
                rnd = torch.randn_like(d["mean"])
                return rnd * torch.exp(d["log_std"]) + d["mean"], d
        flat_obs = self.observation_space.flatten_n(observations)
        d = self.dist_info(flat_obs)
        rnd = np.random.normal(size=d["mean"].shape)
        return rnd * np.exp(d["log_std"]) + d["mean"], d

    def get_reparam_action_sym(self, obs_var, action_var, old_dist_info_vars):
        """The old actions re-expressed through the current parameters: mean_new + eps * exp(log_std_new) with
        eps = (a - mean_old) / (exp(log_std_old) + 1e-8)   (gaussian_mlp_policy.py:138-152).  [B, .] tensors."""
        new = self.dist_info_sym(obs_var, action_var)
        dev, dt = new["mean"].device, new["mean"].dtype
        action = torch.as_tensor(action_var, device=dev, dtype=dt)
        old_mean = torch.as_tensor(old_dist_info_vars["mean"], device=dev, dtype=dt)
        old_log_std = torch.as_tensor(old_dist_info_vars["log_std"], device=dev, dtype=dt)
        eps = (action - old_mean) / (torch.exp(old_log_std) + 1e-8)
        return new["mean"] + eps * torch.exp(new["log_std"])

    def log_diagnostics(self, paths):
        """AveragePolicyStd (reference :155-157).  log_std is state independent, so the
        per-sample mean of exp(log_std) equals the mean over action dims."""
        traj = getattr(paths, "traj", None)
        if traj is not None and getattr(traj, "log_std_host", None) is not None:
            # the row the rollout recorded, already on the host (read with the batch statistics)
            logger.record_tabular('AveragePolicyStd', float(np.mean(np.exp(traj.log_std_host))))
            return
        if traj is not None and getattr(traj, "log_std_planes", None) is not None:
            w = traj.valid.to(torch.float64) if getattr(traj, "valid", None) is not None else None
            e = torch.exp(traj.log_std_planes.double())
            val = float(e.mean()) if w is None else float((e * w).sum() / (w.sum() * e.shape[0]).clamp_min(1.0))
            logger.record_tabular('AveragePolicyStd', val)
            return
        if self.state_dependent_std:
            log_stds = np.vstack([path["agent_infos"]["log_std"] for path in paths])
            logger.record_tabular('AveragePolicyStd', float(np.mean(np.exp(log_stds))))
            return
        ls = self.effective_log_std().detach()
        logger.record_tabular('AveragePolicyStd', float(torch.exp(ls.double()).mean()))

    @property
    def distribution(self):
        return self._dist

    def fused_ops(self):
        """HIP-kernel loss / gradient / Fisher-vector product (policies/fused_ops.py), or
        None when this policy has no fused kernel (then torch autograd is used)."""
        from rllab_amd.policies.fused_ops import FusedGaussianMLPOps
        if self.state_dependent_std:
            # adaptive_std / std_network: both networks on the kernels, the Gaussian head between them
            from rllab_amd.policies.fused_adaptive_ops import FusedAdaptiveStdOps
            return FusedAdaptiveStdOps(self) if FusedAdaptiveStdOps.supported(self) else None
        if not FusedGaussianMLPOps.supported(self):
            return None
        return FusedGaussianMLPOps(self)
