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

from rllab_amd.core.parameterized import Param
from rllab_amd.core.serializable import Serializable
from rllab_amd.distributions.diagonal_gaussian import DiagonalGaussian
from rllab_amd.misc import logger
from rllab_amd.policies.base import StochasticPolicy
from rllab_amd.spaces import Box

tanh = torch.tanh


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
        if adaptive_std or std_network is not None or mean_network is not None:
            raise NotImplementedError("GaussianMLPPolicy: adaptive_std / custom networks are outside "
                                      "the hot path built here (SURVEY.md section 8)")
        StochasticPolicy.__init__(self, env_spec)
        obs_dim = env_spec.observation_space.flat_dim
        action_dim = env_spec.action_space.flat_dim
        self.obs_dim, self.action_dim = obs_dim, action_dim
        self.hidden_sizes = tuple(int(h) for h in hidden_sizes)
        self.hidden_nonlinearity = hidden_nonlinearity
        self.output_nonlinearity = output_nonlinearity
        self.learn_std = learn_std
        self.min_std = min_std
        self._dist = dist_cls(action_dim)

        sizes = (obs_dim,) + self.hidden_sizes + (action_dim,)
        params, off = [], 0
        n_layers = len(sizes) - 1
        for li in range(n_layers):
            lname = "output" if li == n_layers - 1 else "hidden_%d" % li
            w = Param("%s.W" % lname, (sizes[li], sizes[li + 1]), off)
            off += w.size
            b = Param("%s.b" % lname, (sizes[li + 1],), off, regularizable=False)
            off += b.size
            params += [w, b]
        ls = Param("output_log_std.param", (action_dim,), off, trainable=learn_std, regularizable=False)
        off += ls.size
        params.append(ls)
        self._params = params
        self._log_std_param = ls
        for p in params:
            p._owner = self

        # host-side init with np.random (so CPU oracle and GPU share theta under a seed)
        flat = np.zeros(off, dtype=np.float32)
        for li in range(n_layers):
            w = params[2 * li]
            bound = np.sqrt(6.0 / (w.shape[0] + w.shape[1]))
            flat[w.offset:w.offset + w.size] = np.random.uniform(-bound, bound, size=w.shape).reshape(-1)
        flat[ls.offset:ls.offset + ls.size] = np.log(init_std)
        self.flat_params = torch.tensor(flat, dtype=torch.float32, device=_default_device())

    # -- Parameterized ----------------------------------------------------------
    def get_params_internal(self, **tags):
        return [p for p in self._params if all(p.tags.get(k, False) == v for k, v in tags.items())]

    # -- forward ------------------------------------------------------------------
    @property
    def fusable(self):
        """True when the in-kernel MLP (tanh hidden, linear output) matches this policy."""
        return self.hidden_nonlinearity is tanh and self.output_nonlinearity is None

    @property
    def vectorized(self):
        return True

    def effective_log_std(self, flat=None):
        flat = self.flat_params if flat is None else flat
        ls = self._log_std_param.view(flat)
        if self.min_std is not None:
            ls = ls.clamp_min(float(np.log(self.min_std)))   # scalar bound: no host -> device copy per call
        return ls

    def recorded_log_std(self):
        """The log_std row a rollout records as agent_info (the "old" distribution of the update): always a
        COPY -- with ``min_std=None`` ``effective_log_std`` is a view into ``flat_params``, and the line search /
        Adam step rewrite that vector in place, which would make old and new log_std the same tensor."""
        return self.effective_log_std().detach().clone()

    def mean_planes(self, obs_planes, flat=None):
        """obs [Do, B] -> mean [Da, B] ("planes": feature axis first, the engine's layout)."""
        flat = self.flat_params if flat is None else flat
        h = obs_planes
        n_layers = len(self._params) // 2
        for li in range(n_layers):
            W = self._params[2 * li].view(flat)
            b = self._params[2 * li + 1].view(flat)
            h = W.t() @ h + b[:, None]
            if li < n_layers - 1:
                h = self.hidden_nonlinearity(h)
            elif self.output_nonlinearity is not None:
                h = self.output_nonlinearity(h)
        return h

    def dist_info_planes(self, obs_planes, flat=None):
        return dict(mean=self.mean_planes(obs_planes, flat), log_std=self.effective_log_std(flat)[:, None])

    def dist_info_sym(self, obs_var, state_info_vars=None):
        """[B, Do] tensor -> dict(mean [B, Da], log_std [B, Da]) (reference :118-122)."""
        obs_var = torch.as_tensor(obs_var, dtype=self.flat_params.dtype, device=self.flat_params.device)
        mean = self.mean_planes(obs_var.t()).t()
        log_std = self.effective_log_std().unsqueeze(0).expand_as(mean)
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
        ls = self.effective_log_std().detach()
        logger.record_tabular('AveragePolicyStd', float(torch.exp(ls.double()).mean()))

    @property
    def distribution(self):
        return self._dist

    def fused_ops(self):
        """HIP-kernel loss / gradient / Fisher-vector product (policies/fused_ops.py), or
        None when this policy has no fused kernel (then torch autograd is used)."""
        from rllab_amd.policies.fused_ops import FusedGaussianMLPOps
        if not FusedGaussianMLPOps.supported(self):
            return None
        return FusedGaussianMLPOps(self)
