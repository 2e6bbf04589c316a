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
    assert _lib.lib.rl_abi_version() == 14


def test_env_query_and_errors():
    from rllab_amd import _lib
    assert _lib.env_query(_lib.ENV_CARTPOLE) == dict(obs_dim=4, act_dim=1, state_dim=16, reset_draws=4,
                                                     reset_is_normal=False, terminates=True)
    assert _lib.env_query(_lib.ENV_DOUBLE_PENDULUM) == dict(obs_dim=6, act_dim=1, state_dim=17, reset_draws=4,
                                                            reset_is_normal=True, terminates=False)
    assert _lib.env_query(_lib.ENV_SWIMMER) == dict(obs_dim=13, act_dim=2, state_dim=10, reset_draws=10,
                                                    reset_is_normal=True, terminates=False)
    assert _lib.env_query(_lib.ENV_HALF_CHEETAH) == dict(obs_dim=20, act_dim=6, state_dim=18, reset_draws=18,
                                                         reset_is_normal=True, terminates=False)
    assert _lib.env_query(_lib.ENV_CARTPOLE_SWINGUP) == _lib.env_query(_lib.ENV_CARTPOLE)
    assert _lib.env_query(_lib.ENV_WALKER2D) == dict(obs_dim=21, act_dim=6, state_dim=18, reset_draws=18,
                                                     reset_is_normal=True, terminates=True)
    lb, ub = _lib.env_action_bounds(_lib.ENV_WALKER2D)
    assert list(ub) == [150, 100, 100, 150, 100, 100] and list(lb) == [-150, -100, -100, -150, -100, -100]
    lb, ub = _lib.env_action_bounds(_lib.ENV_HALF_CHEETAH)
    assert list(lb) == [-1] * 6 and list(ub) == [1] * 6
    lb, ub = _lib.env_action_bounds(_lib.ENV_SWIMMER)
    assert list(lb) == [-50, -50] and list(ub) == [50, 50]
    assert [_lib.lib.rl_env_terminates(k) for k in range(8)] == [1, 0, 0, 0, 1, 1, 1, 1] and _lib.lib.rl_env_terminates(99) == -1
    assert _lib.lib.rl_env_query(99, None, None, None, None, None) == -1
    assert b"env kind 99" in _lib.lib.rl_last_error()
    assert _lib.lib.rl_vecenv_reset(0, 0, None, None, None, None, 0, 0, 0, None, None, None) == -1
    assert _lib.lib.rl_gae(0, 0, None, None, None, 0.99, 1.0, None, None, None, None) == -1
    assert _lib.lib.rl_rollout_gaussian_mlp(None, None) == -1
    assert _lib.lib.rl_policy_fvp(None, None, None, 0, None, None) == -1
    assert _lib.lib.rl_policy_workspace_bytes(13, 2, 32, 32, 0) > 0
    # the wide / deep family: two or three layers of 32 / 64 / 128 units
    assert _lib.lib.rl_policy_workspace_bytes(13, 2, 128, 64, 32) > _lib.lib.rl_policy_workspace_bytes(13, 2, 128, 64, 0) > 0
    assert _lib.lib.rl_policy_workspace_bytes(13, 2, 128, 96, 0) == 0 and _lib.lib.rl_policy_workspace_bytes(13, 2, 256, 32, 0) == 0
    assert _lib.lib.rl_policy_activation_bytes(1000, 128, 128, 0) == 32 * 32 * 256 * 4      # 1000 samples -> 32 tiles
    assert _lib.lib.rl_policy_activation_bytes(1000, 100, 50, 25) == 0                        # the caller pads first
    # peer all-reduce: argument errors without touching a device
    assert _lib.lib.rl_peer_mailbox_bytes(8, 1572) == 128 + 2 * 8 * 1572 * 8 and _lib.lib.rl_peer_mailbox_bytes(9, 4) == 0
    assert _lib.lib.rl_peer_allreduce_sum(0, None, 0, 1, None, 4, 1, None, 0, None) == -1
    assert _lib.lib.rl_peer_export(None, None) == -1 and _lib.lib.rl_peer_open(None, None) == -1


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
             "ctypes.POINTER(RolloutPlan)": ctypes.POINTER(_lib.RolloutPlan), "ctypes.c_int64": ctypes.c_int64,
             "ip": ctypes.POINTER(ctypes.c_int), "fp": ctypes.POINTER(ctypes.c_float),
             "vpp": ctypes.POINTER(ctypes.c_void_p), "szp": ctypes.POINTER(ctypes.c_size_t)}
    found = re.findall(r"^\s*lib\.(rl_\w+)\.argtypes\s*= \[([^\]]*)\]", text, flags=re.M)
    assert len(found) >= 24
    for fn, args in found:
        toks = [a.strip() for a in args.split(",") if a.strip()]
        want = list(getattr(_lib.lib, fn).argtypes)
        assert len(toks) == len(want), "%s: INTEGRATION.md lists %d arguments, the ABI has %d" % (fn, len(toks), len(want))
        for k, (t, w) in enumerate(zip(toks, want)):
            assert short[t] is w, "%s argument %d: INTEGRATION.md says %s, the binding is %s" % (fn, k, t, w)
    assert "rl_abi_version() == %d" % _lib.lib.rl_abi_version() in text


def _md_struct_fields(text):
    """{class name: [(field, type token), ...]} of every ``class X(ctypes.Structure)`` printed in INTEGRATION.md."""
    out = {}
    for name, body in re.findall(r"class (\w+)\(ctypes\.Structure\):.*?_fields_\s*=\s*\[(.*?)\]\s*\n", text, flags=re.S):
        out[name] = re.findall(r'\(\s*"(\w+)"\s*,\s*([^()]*?(?:\([^()]*\))?)\s*\)', body)
    return out


def test_integration_md_structs_are_the_library_structs():
    """Every ctypes.Structure a maintainer would copy out of INTEGRATION.md has the fields, order and types of the
    mirror in rllab_amd/_lib.py (which test_structs_match_header_layout ties to include/rllab_amd.h): a struct
    printed short or stale shifts every pointer behind the gap and the device dereferences garbage."""
    from rllab_amd import _lib
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    types_ = {"f32": ctypes.c_float, "i32f": ctypes.c_int32, "ctypes.c_int32": ctypes.c_int32, "vp": ctypes.c_void_p,
              "u64f": ctypes.c_uint64, "ctypes.c_uint64": ctypes.c_uint64, "f64": ctypes.c_double,
              "ctypes.POINTER(EnvCfg)": ctypes.POINTER(_lib.EnvCfg), "cfgp": ctypes.POINTER(_lib.EnvCfg),
              "i32f * 7": ctypes.c_int32 * 7, "ctypes.c_char * 96": ctypes.c_char * 96}
    printed = _md_struct_fields(text)
    want = {"EnvCfg": _lib.EnvCfg, "RolloutArgs": _lib.RolloutArgs, "PolicyBatch": _lib.PolicyBatch,
            "LaunchOpts": _lib.LaunchOpts, "RolloutPlan": _lib.RolloutPlan, "RunningNorm": _lib.RunningNorm}
    assert set(printed) == set(want), sorted(printed)
    for name, cls in want.items():
        got = [(f, types_[t.strip()]) for f, t in printed[name]]
        assert got == [(f, t) for f, t in cls._fields_], \
            "%s in INTEGRATION.md differs from the library's struct:\n  printed %s\n  library %s" % (
                name, [f for f, _ in got], [f for f, _ in cls._fields_])
        # and the layout a copier would get: same size, same offset of the last field
        rebuilt = type("Md" + name, (ctypes.Structure,), {"_fields_": got})
        assert ctypes.sizeof(rebuilt) == ctypes.sizeof(cls)
        assert getattr(rebuilt, got[-1][0]).offset == getattr(cls, cls._fields_[-1][0]).offset


def _md_calls(text):
    """(function, number of top-level arguments) of every ``lib.rl_*( ... )`` call printed in INTEGRATION.md."""
    calls = []
    for m in re.finditer(r"\blib\.(rl_\w+)\(", text):
        i, depth, n_args, seen = m.end(), 1, 0, False
        while depth:
            c = text[i]
            if c in "([{":
                depth += 1
            elif c in ")]}":
                depth -= 1
            elif c == "," and depth == 1:
                n_args += 1
            if depth and not c.isspace():
                seen = True
            i += 1
        calls.append((m.group(1), n_args + 1 if seen else 0))
    return calls


def test_integration_md_example_calls_have_the_abi_arity():
    """Every example call ``lib.rl_*(...)`` in INTEGRATION.md passes exactly as many arguments as the entry point
    takes (an 8-argument rl_trpo_step against 9 argtypes went unnoticed for a round)."""
    from rllab_amd import _lib
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    calls = _md_calls(text)
    assert len(calls) >= 15
    checked = 0
    for fn, n_args in calls:
        argtypes = getattr(_lib.lib, fn).argtypes
        if argtypes is None:
            assert n_args == 0, fn
            continue
        assert n_args == len(argtypes), "INTEGRATION.md calls %s with %d arguments, the ABI takes %d" % (
            fn, n_args, len(argtypes))
        checked += 1
    assert checked >= 15


def test_struct_offsets_against_the_compiled_header(tmp_path):
    """include/rllab_amd.h compiled as plain C (gcc): sizeof and every offsetof equal the ctypes mirrors' -- names
    in the right order (test_structs_match_header_layout) do not catch a wrong width."""
    import subprocess
    from rllab_amd import _lib
    structs = (("rl_env_cfg", _lib.EnvCfg), ("rl_rollout_args", _lib.RolloutArgs), ("rl_policy_batch", _lib.PolicyBatch),
               ("rl_launch_opts", _lib.LaunchOpts), ("rl_rollout_plan", _lib.RolloutPlan),
               ("rl_running_norm", _lib.RunningNorm))
    lines = ['#include "rllab_amd.h"', "#include <stdio.h>", "#include <stddef.h>", "int main(void) {"]
    for cname, cls in structs:
        lines.append('  printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for f, _ in cls._fields_:
            lines.append('  printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, f, cname, f))
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = str(tmp_path / "layout")
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe])
    got = dict(l.split() for l in subprocess.check_output([exe]).decode().splitlines())
    for cname, cls in structs:
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        for f, _ in cls._fields_:
            assert int(got["%s.%s" % (cname, f)]) == getattr(cls, f).offset, (cname, f)


def test_every_env_switch_is_documented():
    """Every RLLAB_* environment switch the product (package, bench.py, the build entry) reads has a row in
    INTEGRATION.md section 4, and every test file that row names exists and mentions the switch (or the attribute the
    row says stands for it)."""
    import glob
    import re
    names = set()
    files = [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
    for ext_ in ("py", "hip", "h"):
        files += glob.glob(os.path.join(ROOT, "rllab_amd", "**", "*." + ext_), recursive=True)
    for f in files:
        names |= set(re.findall(r"RLLAB_[A-Z0-9_]+", open(f, errors="replace").read()))
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    section = text[text.index("## 4. Environment switches"):]
    rows = {m.group(1): m.group(0) for m in re.finditer(r"^\| `(RLLAB_[A-Z0-9_]+)` \|.*$", section, re.M)}
    assert names, "no switches found: the grep is broken"
    missing = sorted(names - set(rows))
    assert not missing, "switches read by the product but absent from INTEGRATION.md section 4: %s" % missing
    stale = sorted(set(rows) - names)
    assert not stale, "INTEGRATION.md documents switches nothing reads: %s" % stale
    for name, row in rows.items():
        for tf in re.findall(r"`((?:tests/)?test_[a-z0-9_]+\.py)", row):
            path = os.path.join(ROOT, tf if tf.startswith("tests/") else os.path.join("tests", tf))
            assert os.path.exists(path), (name, tf)
