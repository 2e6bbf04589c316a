"""ctypes binding of librllab_amd.so (the C ABI declared in include/rllab_amd.h).

The product path has NO CPU fallback: if the HIP library is missing this module
raises at import time, and every entry point raises ``RuntimeError`` with
``rl_last_error()`` on a non-zero status.  PyTorch is used only as plumbing
(device memory, streams); no torch type crosses the ABI -- tensors are handed
over as raw device pointers.
"""
import ctypes
import os

import torch  # noqa: F401  (must be imported first: it loads the HIP runtime the library binds to)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librllab_amd.so")

ENV_CARTPOLE = 0
ENV_DOUBLE_PENDULUM = 1
ENV_SWIMMER = 2
ENV_HALF_CHEETAH = 3
ENV_CARTPOLE_SWINGUP = 4
ENV_WALKER2D = 5
ENV_HOPPER = 6
ENV_INVERTED_DOUBLE_PENDULUM = 7

# every symbol include/rllab_amd.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = [
    "rl_last_error", "rl_abi_version", "rl_env_query", "rl_env_terminates", "rl_env_action_bounds", "rl_env_default_cfg", "rl_vecenv_com",
    "rl_vecenv_reset", "rl_vecenv_step", "rl_vecenv_step_graph", "rl_counter_add", "rl_vecenv_observe", "rl_rollout_gaussian_mlp", "rl_rollout_plan_query", "rl_rollout_lds_bytes", "rl_gae",
    "rl_discount_cumsum", "rl_debug_philox", "rl_policy_workspace_bytes", "rl_policy_activation_bytes", "rl_policy_loss_kl",
    "rl_policy_grad", "rl_policy_grad_loss", "rl_policy_fvp", "rl_policy_fvp_variant", "rl_policy_fvp_cg_step", "rl_cg_init", "rl_cg_step", "rl_trpo_step", "rl_line_search_point", "rl_line_search_decide", "rl_adam_step",
    "rl_path_scan", "rl_process_workspace_bytes", "rl_sample_stats_cols", "rl_sample_stats", "rl_adv_finish",
    "rl_lfb_normal_eq",
    "rl_peer_mailbox_bytes", "rl_peer_alloc", "rl_peer_free", "rl_peer_export", "rl_peer_open", "rl_peer_close",
    "rl_peer_allreduce_sum",
    "rl_mlp_forward", "rl_mlp_forward_ws", "rl_mlp_backward", "rl_gaussian_head_workspace_bytes", "rl_gaussian_head", "rl_gaussian_fisher",
]


class EnvCfg(ctypes.Structure):
    """Mirror of ``rl_env_cfg`` (include/rllab_amd.h): constructor options of the reference's env classes."""
    _fields_ = [
        ("ctrl_cost_coeff", ctypes.c_float), ("alive_coeff", ctypes.c_float), ("action_noise", ctypes.c_float),
        ("obs_noise", ctypes.c_float), ("frame_skip", ctypes.c_int32), ("flags", ctypes.c_int32),
        ("link_len", ctypes.c_float), ("reserved", ctypes.c_float), ("action_noise_z", ctypes.c_void_p), ("obs_noise_z", ctypes.c_void_p),
    ]


CFG_POLE_FOLLOWS_CART, CFG_FIXED_START, CFG_LIMIT_MUJOCO, CFG_CONTACT_MUJOCO = 1, 2, 4, 8


class RolloutArgs(ctypes.Structure):
    """Mirror of ``rl_rollout_args`` (include/rllab_amd.h)."""
    _fields_ = [
        ("kind", ctypes.c_int32), ("n_envs", ctypes.c_int32), ("horizon", ctypes.c_int32),
        ("max_path_length", ctypes.c_int32), ("normalize", ctypes.c_int32),
        ("reset_at_start", ctypes.c_int32), ("hidden0", ctypes.c_int32), ("hidden1", ctypes.c_int32),
        ("hidden2", ctypes.c_int32), ("env_offset", ctypes.c_int32), ("scale_reward", ctypes.c_float),
        ("log_min_std", ctypes.c_float), ("seed", ctypes.c_uint64), ("step_counter", ctypes.c_uint64),
        ("state", ctypes.c_void_p), ("ts", ctypes.c_void_p), ("theta", ctypes.c_void_p),
        ("eps", ctypes.c_void_p), ("reset_draws", ctypes.c_void_p), ("obs", ctypes.c_void_p),
        ("actions", ctypes.c_void_p), ("means", ctypes.c_void_p), ("rewards", ctypes.c_void_p),
        ("dones", ctypes.c_void_p), ("last_obs", ctypes.c_void_p), ("cfg", ctypes.POINTER(EnvCfg)),
        ("theta_std", ctypes.c_void_p), ("log_stds", ctypes.c_void_p), ("std_hidden0", ctypes.c_int32),
        ("std_hidden1", ctypes.c_int32), ("std_hidden2", ctypes.c_int32), ("layer_activations", ctypes.c_int32),
        ("opts", ctypes.c_void_p), ("norm", ctypes.c_void_p), ("std_layer_activations", ctypes.c_int32),
        ("reserved_tail", ctypes.c_int32),
    ]


class PolicyBatch(ctypes.Structure):
    """Mirror of ``rl_policy_batch`` (include/rllab_amd.h)."""
    _fields_ = [
        ("n_samples", ctypes.c_int32), ("obs_dim", ctypes.c_int32), ("act_dim", ctypes.c_int32),
        ("hidden0", ctypes.c_int32), ("hidden1", ctypes.c_int32), ("hidden2", ctypes.c_int32),
        ("inv_count", ctypes.c_float),
        ("log_min_std", ctypes.c_float), ("theta", ctypes.c_void_p), ("obs", ctypes.c_void_p),
        ("actions", ctypes.c_void_p), ("advantages", ctypes.c_void_p), ("old_means", ctypes.c_void_p),
        ("old_log_std", ctypes.c_void_p), ("weights", ctypes.c_void_p), ("activations", ctypes.c_void_p),
        ("kl_penalty", ctypes.c_float), ("activation", ctypes.c_int32), ("opts", ctypes.c_void_p),
        ("layer_activations", ctypes.c_int32), ("reserved_pad", ctypes.c_int32), ("gate", ctypes.c_void_p),
        ("obs_absmax", ctypes.c_void_p),
    ]


ACT_TANH, ACT_RECTIFY, ACT_IDENTITY = 0, 1, 2


def layer_activation_bits(codes):
    """Per-layer activation codes -> ``layer_activations`` (two bits per layer holding code + 1); all tanh -> 0."""
    if all(c == ACT_TANH for c in codes):
        return 0
    return sum((int(c) + 1) << (2 * l) for l, c in enumerate(codes))


class LaunchOpts(ctypes.Structure):
    """Mirror of ``rl_launch_opts`` (include/rllab_amd.h): launch-shape requests, 0 = the library's own rule."""
    _fields_ = [
        ("rollout_epw", ctypes.c_int32), ("rollout_wpb", ctypes.c_int32), ("swimmer_lane_kernel", ctypes.c_int32),
        ("swimmer_coop", ctypes.c_int32), ("two_leg_lane_kernel", ctypes.c_int32), ("two_leg_wave_kernel", ctypes.c_int32),
        ("fvp_split", ctypes.c_int32), ("fvp_split_wps", ctypes.c_int32), ("lfb_valu", ctypes.c_int32),
        ("reserved", ctypes.c_int32 * 7),
    ]


class RunningNorm(ctypes.Structure):
    """Mirror of ``rl_running_norm``: NormalizedEnv's running observation / reward normalisation inside the fused rollout."""
    _fields_ = [
        ("obs_mean", ctypes.c_void_p), ("obs_var", ctypes.c_void_p), ("reward_mean", ctypes.c_void_p),
        ("reward_var", ctypes.c_void_p), ("obs_alpha", ctypes.c_double), ("reward_alpha", ctypes.c_double),
        ("normalize_obs", ctypes.c_int32), ("normalize_reward", ctypes.c_int32),
    ]


class RolloutPlan(ctypes.Structure):
    """Mirror of ``rl_rollout_plan``: which kernel, in which shape, a rollout call launches."""
    _fields_ = [
        ("kernel", ctypes.c_int32), ("envs_per_wavefront", ctypes.c_int32), ("wavefronts", ctypes.c_int32),
        ("wavefronts_per_workgroup", ctypes.c_int32), ("workgroups", ctypes.c_int32), ("lds_bytes", ctypes.c_int32),
        ("lds_limit", ctypes.c_int32), ("reserved", ctypes.c_int32), ("name", ctypes.c_char * 96),
    ]


ROLLOUT_UNSUPPORTED = 0
_OPTS = LaunchOpts()        # ONE instance, refreshed in place: its address is what the argument structs carry


def _env_int(name, allowed):
    v = os.environ.get(name)
    if v is None:
        return 0
    try:
        v = int(v)
    except ValueError:
        return 0
    return v if v in allowed else 0


def launch_opts():
    """Address of the process's ``rl_launch_opts``, refreshed from the RLLAB_* switches of INTEGRATION.md section 4 (read
    HERE, per call -- the library reads no environment variable; tests flip the switches between two launches of one
    process).  Unset switches leave every field 0 = the library's own launch rules."""
    e, o = os.environ.get, _OPTS
    o.rollout_epw = _env_int("RLLAB_ROLLOUT_EPW", (16, 64))
    o.rollout_wpb = _env_int("RLLAB_ROLLOUT_WPB", (1, 2, 4))
    o.swimmer_lane_kernel = 1 if e("RLLAB_SWIMMER_LANE_KERNEL") is not None else 0
    v = e("RLLAB_SWIMMER_COOP")
    o.swimmer_coop = 0 if not v else (1 if v[0] == "1" else 2)
    v = e("RLLAB_TWO_LEG_LANE_KERNEL")
    o.two_leg_lane_kernel = 2 if v and v[0] == "0" else 0
    v = e("RLLAB_TWO_LEG_WAVE_KERNEL")
    o.two_leg_wave_kernel = 0 if not v else (1 if v[0] == "1" else 2)
    v = e("RLLAB_FVP_SPLIT")
    o.fvp_split = 0 if not v else (1 if v[0] == "0" else int(v[0]) if v[0] in "2345" else 0)
    v = e("RLLAB_FVP_SPLIT_WPS")
    o.fvp_split_wps = int(v[0]) if v and v[0] in "134" else 0
    o.lfb_valu = 1 if e("RLLAB_LFB_VALU") is not None else 0
    o.reserved[0] = _env_int("RLLAB_SPLIT16_ABLATE", range(1, 8))     # timing ablations of fvp_split16_kernel (wrong results)
    return ctypes.addressof(o)


def peer_spin_limit():
    """RLLAB_PEER_SPIN_LIMIT (polls before rl_peer_allreduce_sum gives up on a silent peer; tests), 0 = wall clock only."""
    try:
        return max(0, int(os.environ.get("RLLAB_PEER_SPIN_LIMIT", "0")))
    except ValueError:
        return 0


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "rllab_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    vp, i32, u64, f32, f64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_float, ctypes.c_double
    u32 = ctypes.c_uint32
    ip = ctypes.POINTER(ctypes.c_int)
    fp = ctypes.POINTER(ctypes.c_float)
    lib.rl_last_error.restype = ctypes.c_char_p
    lib.rl_last_error.argtypes = []
    lib.rl_abi_version.restype = i32
    lib.rl_env_query.argtypes = [i32, ip, ip, ip, ip, ip]
    lib.rl_env_terminates.argtypes = [i32]
    lib.rl_env_action_bounds.argtypes = [i32, fp, fp]
    cfgp = ctypes.POINTER(EnvCfg)
    lib.rl_env_default_cfg.argtypes = [i32, cfgp]
    lib.rl_vecenv_com.argtypes = [i32, i32, vp, vp, vp]
    lib.rl_vecenv_reset.argtypes = [i32, i32, vp, vp, vp, vp, u64, u64, i32, cfgp, vp, vp]
    lib.rl_vecenv_observe.argtypes = [i32, i32, vp, vp, vp]
    lib.rl_vecenv_step_graph.argtypes = [i32, i32, i32, f32, i32, i32, vp, vp, vp, u64, vp, i32, cfgp, vp, vp, vp, vp]
    lib.rl_counter_add.argtypes = [vp, u64, vp]
    lib.rl_vecenv_step.argtypes = [i32, i32, i32, f32, i32, i32, vp, vp, vp, vp, u64, u64, i32, cfgp, vp, vp, vp, vp]
    lib.rl_rollout_gaussian_mlp.argtypes = [ctypes.POINTER(RolloutArgs), vp]
    lib.rl_rollout_plan_query.argtypes = [ctypes.POINTER(RolloutArgs), ctypes.POINTER(RolloutPlan)]
    lib.rl_rollout_lds_bytes.argtypes = [i32, i32, i32, i32, i32, i32, i32, ctypes.POINTER(ctypes.c_size_t),
                                         ctypes.POINTER(ctypes.c_size_t)]
    lib.rl_gae.argtypes = [i32, i32, vp, vp, vp, f64, f64, vp, vp, vp, vp]
    lib.rl_discount_cumsum.argtypes = [i32, i32, vp, vp, f64, vp, vp]
    lib.rl_debug_philox.argtypes = [u32, u32, u32, u32, u32, u32, i32, vp, vp]
    pb = ctypes.POINTER(PolicyBatch)
    lib.rl_policy_workspace_bytes.restype = ctypes.c_size_t
    lib.rl_policy_workspace_bytes.argtypes = [i32, i32, i32, i32, i32]
    lib.rl_policy_activation_bytes.restype = ctypes.c_size_t
    lib.rl_policy_activation_bytes.argtypes = [i32, i32, i32, i32]
    lib.rl_policy_loss_kl.argtypes = [pb, vp, ctypes.c_size_t, vp, vp]
    lib.rl_policy_grad.argtypes = [pb, i32, vp, ctypes.c_size_t, vp, vp]
    lib.rl_policy_grad_loss.argtypes = [pb, i32, vp, ctypes.c_size_t, vp, vp, vp]
    lib.rl_policy_fvp.argtypes = [pb, vp, vp, ctypes.c_size_t, vp, vp]
    lib.rl_policy_fvp_variant.argtypes = [pb]
    lib.rl_policy_fvp_cg_step.argtypes = [pb, vp, ctypes.c_size_t, f64, f64, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.rl_cg_init.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp]
    lib.rl_cg_step.argtypes = [i32, vp, f64, f64, vp, vp, vp, vp, vp, vp]
    lib.rl_trpo_step.argtypes = [i32, vp, vp, vp, f64, f64, vp, vp, vp]
    lib.rl_line_search_point.argtypes = [i32, vp, vp, f64, vp, vp]
    lib.rl_line_search_decide.argtypes = [i32, vp, vp, f64, f64, i32, vp, vp, i32, vp, vp, f64, vp, vp]
    lib.rl_adam_step.argtypes = [i32, vp, vp, vp, vp, f64, f64, f64, f64, vp]
    sz = ctypes.c_size_t
    lib.rl_path_scan.argtypes = [i32, i32, i32, vp, vp, vp, i32, vp, vp, vp, vp]
    lib.rl_process_workspace_bytes.restype = sz
    lib.rl_process_workspace_bytes.argtypes = [i32]
    lib.rl_sample_stats_cols.restype = i32
    lib.rl_sample_stats.argtypes = [sz, vp, vp, vp, vp, vp, vp, f64, f64, vp, i32, vp, sz, vp, vp]
    lib.rl_adv_finish.argtypes = [sz, vp, vp, f64, f64, f64, vp, vp]
    lib.rl_lfb_normal_eq.argtypes = [sz, i32, vp, vp, vp, vp, vp, sz, vp, i32, vp]
    lib.rl_mlp_forward.argtypes = [pb, vp, vp, vp, vp]
    lib.rl_mlp_forward_ws.argtypes = [pb, vp, vp, sz, vp, vp, vp]
    lib.rl_mlp_backward.argtypes = [pb, vp, vp, sz, vp, vp]
    lib.rl_gaussian_head_workspace_bytes.restype = sz
    lib.rl_gaussian_head_workspace_bytes.argtypes = []
    lib.rl_gaussian_head.argtypes = [sz, i32, vp, vp, vp, vp, vp, vp, vp, f32, f32, i32, f32, vp, vp, vp, sz, vp, vp]
    lib.rl_gaussian_fisher.argtypes = [sz, i32, vp, vp, vp, vp, f32, f32, vp, vp, vp]
    vpp = ctypes.POINTER(ctypes.c_void_p)
    lib.rl_peer_mailbox_bytes.restype = sz
    lib.rl_peer_mailbox_bytes.argtypes = [i32, i32]
    lib.rl_peer_alloc.argtypes = [sz, vpp]
    lib.rl_peer_free.argtypes = [vp]
    lib.rl_peer_export.argtypes = [vp, vp]
    lib.rl_peer_open.argtypes = [vp, vpp]
    lib.rl_peer_close.argtypes = [vp]
    lib.rl_peer_allreduce_sum.argtypes = [i32, vp, i32, i32, vpp, i32, u64, vp, ctypes.c_int64, vp]
    for name in SYMBOLS:
        getattr(lib, name)  # AttributeError here = header / library mismatch
    return lib


lib = _load()


def check(status, what="rllab_amd"):
    if status != 0:
        raise RuntimeError("%s failed (%d): %s" % (what, status, lib.rl_last_error().decode()))


def ptr(t):
    """Raw device pointer of a contiguous tensor (or None)."""
    if t is None:
        return None
    assert t.is_contiguous(), "rllab_amd C ABI takes contiguous planes"
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    if not torch.cuda.is_available():
        return None
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def env_query(kind):
    o, a, s, r, nrm = (ctypes.c_int() for _ in range(5))
    check(lib.rl_env_query(kind, ctypes.byref(o), ctypes.byref(a), ctypes.byref(s), ctypes.byref(r),
                           ctypes.byref(nrm)), "rl_env_query")
    term = lib.rl_env_terminates(kind)
    check(min(term, 0), "rl_env_terminates")
    return dict(obs_dim=o.value, act_dim=a.value, state_dim=s.value, reset_draws=r.value,
                reset_is_normal=bool(nrm.value), terminates=bool(term))


def rollout_plan(kind, n_envs, horizon, hidden3, std_hidden3=(0, 0, 0), cfg_flags=0, layer_activations=0, norm=None,
                 obs_noise=0.0, std_layer_activations=0):
    """``RolloutPlan`` of a fused rollout of these sizes under the current launch options (rl_rollout_plan_query), or None
    when the library has no kernel for it (the reason is then in ``lib.rl_last_error()``)."""
    a = RolloutArgs(kind=kind, n_envs=int(n_envs), horizon=int(horizon), hidden0=hidden3[0], hidden1=hidden3[1],
                    hidden2=hidden3[2], std_hidden0=std_hidden3[0], std_hidden1=std_hidden3[1], std_hidden2=std_hidden3[2],
                    layer_activations=int(layer_activations), std_layer_activations=int(std_layer_activations))
    cfg = None
    if cfg_flags or obs_noise:
        cfg = EnvCfg()
        check(lib.rl_env_default_cfg(kind, ctypes.byref(cfg)), "rl_env_default_cfg")
        cfg.flags = int(cfg_flags)
        cfg.obs_noise = float(obs_noise)
        a.cfg = ctypes.pointer(cfg)
    nrm = None
    if norm is not None:                      # (normalize_obs, normalize_reward): only the flags are read by the query
        nrm = RunningNorm(normalize_obs=int(bool(norm[0])), normalize_reward=int(bool(norm[1])))
        a.norm = ctypes.addressof(nrm)
        a.reset_at_start = 1
    if tuple(std_hidden3) != (0, 0, 0):
        a.theta_std = 1          # (only its being non-NULL is read by the query: a log-std network is present)
        a.log_stds = 1
    a.opts = launch_opts()
    plan = RolloutPlan()
    rc = lib.rl_rollout_plan_query(ctypes.byref(a), ctypes.byref(plan))
    if rc != 0 or plan.kernel == ROLLOUT_UNSUPPORTED:
        return None
    return plan


def rollout_lds_fits(kind, hidden3, std_hidden3=(0, 0, 0), n_envs=64, horizon=1):
    """Does the library have a fused rollout for these hidden sizes (and, with ``std_hidden3``, a log-std network next to the
    mean network) on env ``kind`` -- the launcher's own answer (rl_rollout_plan_query: the shape it would choose fits
    the LDS of a CU), not a copy of its rules."""
    return rollout_plan(kind, n_envs, horizon, hidden3, std_hidden3) is not None

def env_default_cfg(kind, **overrides):
    """``EnvCfg`` of env ``kind``: its defaults (rl_env_default_cfg) with ``overrides`` (field = value) applied."""
    c = EnvCfg()
    check(lib.rl_env_default_cfg(kind, ctypes.byref(c)), "rl_env_default_cfg")
    names = dict(EnvCfg._fields_)
    for k, v in overrides.items():
        if k not in names:
            raise TypeError("unknown env option %r" % (k,))
        setattr(c, k, v)
    return c


def env_action_bounds(kind):
    import numpy as np
    q = env_query(kind)
    lb = (ctypes.c_float * q["act_dim"])()
    ub = (ctypes.c_float * q["act_dim"])()
    check(lib.rl_env_action_bounds(kind, lb, ub), "rl_env_action_bounds")
    return np.array(lb[:], dtype=np.float64), np.array(ub[:], dtype=np.float64)
