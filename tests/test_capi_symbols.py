"""The C-ABI library loads without a GPU and exports every symbol include/rllab_amd.h
declares; argument errors come back as status codes + rl_last_error (no compute)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "rllab_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rl_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    from rllab_amd import _lib
    names = _declared()
    assert len(names) >= 14
    for n in names:
        assert hasattr(_lib.lib, n), "librllab_amd.so does not export %s" % n
    assert sorted(_lib.SYMBOLS) == names
    assert _lib.lib.rl_abi_version() == 6


def test_env_query_and_errors():
    from rllab_amd import _lib
    assert _lib.env_query(_lib.ENV_CARTPOLE) == dict(obs_dim=4, act_dim=1, state_dim=16, reset_draws=4,
                                                     reset_is_normal=False)
    assert _lib.env_query(_lib.ENV_DOUBLE_PENDULUM) == dict(obs_dim=6, act_dim=1, state_dim=17, reset_draws=4,
                                                            reset_is_normal=True)
    assert _lib.env_query(_lib.ENV_SWIMMER) == dict(obs_dim=13, act_dim=2, state_dim=10, reset_draws=10,
                                                    reset_is_normal=True)
    assert _lib.env_query(_lib.ENV_HALF_CHEETAH) == dict(obs_dim=20, act_dim=6, state_dim=18, reset_draws=18,
                                                         reset_is_normal=True)
    assert _lib.env_query(_lib.ENV_CARTPOLE_SWINGUP) == _lib.env_query(_lib.ENV_CARTPOLE)
    assert _lib.env_query(_lib.ENV_WALKER2D) == dict(obs_dim=21, act_dim=6, state_dim=18, reset_draws=18,
                                                     reset_is_normal=True)
    lb, ub = _lib.env_action_bounds(_lib.ENV_WALKER2D)
    assert list(ub) == [150, 100, 100, 150, 100, 100] and list(lb) == [-150, -100, -100, -150, -100, -100]
    lb, ub = _lib.env_action_bounds(_lib.ENV_HALF_CHEETAH)
    assert list(lb) == [-1] * 6 and list(ub) == [1] * 6
    lb, ub = _lib.env_action_bounds(_lib.ENV_SWIMMER)
    assert list(lb) == [-50, -50] and list(ub) == [50, 50]
    assert _lib.lib.rl_env_query(99, None, None, None, None, None) == -1
    assert b"env kind 99" in _lib.lib.rl_last_error()
    assert _lib.lib.rl_vecenv_reset(0, 0, None, None, None, None, 0, 0, 0, None, None, None) == -1
    assert _lib.lib.rl_gae(0, 0, None, None, None, 0.99, 1.0, None, None, None, None) == -1
    assert _lib.lib.rl_rollout_gaussian_mlp(None, None) == -1
    assert _lib.lib.rl_policy_fvp(None, None, None, 0, None, None) == -1
    assert _lib.lib.rl_policy_workspace_bytes(13, 2, 32, 32) > 0


def test_env_default_cfg():
    """rl_env_default_cfg: the reward coefficients / frame skips the reference's env constructors default to."""
    from rllab_amd import _lib
    want = {_lib.ENV_CARTPOLE: (0.0, 0.0, 1), _lib.ENV_DOUBLE_PENDULUM: (0.0, 0.0, 2),
            _lib.ENV_SWIMMER: (1e-2, 0.0, 1), _lib.ENV_WALKER2D: (1e-2, 0.0, 1), _lib.ENV_HOPPER: (0.01, 1.0, 1)}
    for kind, (cc, alive, fs) in want.items():
        c = _lib.env_default_cfg(kind)
        assert abs(c.ctrl_cost_coeff - cc) < 1e-9 and c.alive_coeff == alive and c.frame_skip == fs
        assert c.action_noise == 0.0 and c.obs_noise == 0.0 and c.flags == 0
        assert not c.action_noise_z and not c.obs_noise_z
    c = _lib.env_default_cfg(_lib.ENV_SWIMMER, ctrl_cost_coeff=0.5, action_noise=0.1)
    assert c.ctrl_cost_coeff == 0.5 and abs(c.action_noise - 0.1) < 1e-8
    import pytest
    with pytest.raises(TypeError):
        _lib.env_default_cfg(_lib.ENV_SWIMMER, gravity=3.0)
    assert _lib.lib.rl_env_default_cfg(99, ctypes.byref(_lib.EnvCfg())) == -1
    assert _lib.lib.rl_vecenv_com(_lib.ENV_SWIMMER, 0, None, None, None) == -1


def test_structs_match_header_layout():
    """ctypes mirrors have the field order / count of the C structs."""
    from rllab_amd import _lib
    text = open(os.path.join(ROOT, "include", "rllab_amd.h")).read()
    for cname, cls in (("rl_rollout_args", _lib.RolloutArgs), ("rl_policy_batch", _lib.PolicyBatch),
                       ("rl_env_cfg", _lib.EnvCfg)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), text, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            names = decl.split(",")
            first = names[0].split()[-1].lstrip("*")
            fields.append(first)
            fields += [n.strip().lstrip("*") for n in names[1:]]
        assert fields == [f[0] for f in cls._fields_], (cname, fields)


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: no module of the product may reference it."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "rllab_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(base, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M) or '#include "../../oracle' in src:
                    bad.append(os.path.join(base, f))
    assert not bad, bad


def test_integration_md_bindings_match_the_library():
    """Every ``lib.rl_*.argtypes = [...]`` a maintainer would copy out of INTEGRATION.md has the argument count and
    types ``rllab_amd/_lib.py`` binds (which test_structs / the GPU tests exercise against the real library)."""
    from rllab_amd import _lib
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    short = {"vp": ctypes.c_void_p, "i32": ctypes.c_int, "u64": ctypes.c_uint64, "f32": ctypes.c_float,
             "f64": ctypes.c_double, "sz": ctypes.c_size_t, "u32": ctypes.c_uint32,
             "cfgp": ctypes.POINTER(_lib.EnvCfg), "pb": ctypes.POINTER(_lib.PolicyBatch),
             "ctypes.POINTER(RolloutArgs)": ctypes.POINTER(_lib.RolloutArgs),
             "ip": ctypes.POINTER(ctypes.c_int), "fp": ctypes.POINTER(ctypes.c_float)}
    found = re.findall(r"^\s*lib\.(rl_\w+)\.argtypes\s*= \[([^\]]*)\]", text, flags=re.M)
    assert len(found) >= 15
    for fn, args in found:
        toks = [a.strip() for a in args.split(",") if a.strip()]
        want = list(getattr(_lib.lib, fn).argtypes)
        assert len(toks) == len(want), "%s: INTEGRATION.md lists %d arguments, the ABI has %d" % (fn, len(toks), len(want))
        for k, (t, w) in enumerate(zip(toks, want)):
            assert short[t] is w, "%s argument %d: INTEGRATION.md says %s, the binding is %s" % (fn, k, t, w)
    assert "rl_abi_version() == %d" % _lib.lib.rl_abi_version() in text
