"""The launch rules as data (host logic, no GPU): rl_rollout_plan_query answers which kernel, in which shape,
rl_rollout_gaussian_mlp launches for a set of arguments, and rl_launch_opts -- filled by rllab_amd/_lib.py::launch_opts from
the RLLAB_* switches, per call -- is the only way to ask for another shape: the library reads no environment variable."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SWITCHES = ["RLLAB_ROLLOUT_EPW", "RLLAB_ROLLOUT_WPB", "RLLAB_SWIMMER_LANE_KERNEL", "RLLAB_SWIMMER_COOP",
            "RLLAB_TWO_LEG_LANE_KERNEL", "RLLAB_TWO_LEG_WAVE_KERNEL", "RLLAB_FVP_SPLIT", "RLLAB_FVP_SPLIT_WPS", "RLLAB_LFB_VALU"]


@pytest.fixture
def clean_env(monkeypatch):
    for k in SWITCHES:
        monkeypatch.delenv(k, raising=False)
    return monkeypatch


def _plan(kind, n, hidden, T=500, std=(0, 0, 0), flags=0):
    from rllab_amd import _lib
    p = _lib.rollout_plan(kind, n, T, tuple(hidden) + (0,) * (3 - len(hidden)), std, cfg_flags=flags)
    return None if p is None else (p.kernel, p.envs_per_wavefront, p.wavefronts, p.wavefronts_per_workgroup, p.workgroups,
                                   p.name.decode())


def test_the_library_reads_no_environment_variable():
    for base, _, files in os.walk(os.path.join(ROOT, "rllab_amd", "csrc")):
        for f in files:
            assert "getenv" not in open(os.path.join(base, f), errors="replace").read(), f


def test_baseline_configs_resolve_to_their_kernels(clean_env):
    from rllab_amd import _lib
    # C3 / C4's shard: 4096 Swimmer envs, (32, 32): four lanes per env, 16 envs per wavefront, one wavefront per SIMD
    assert _plan(_lib.ENV_SWIMMER, 4096, (32, 32)) == (4, 16, 256, 1, 256, "rollout_swimmer_quad_kernel<32>")
    # C5's shard: 1024 HalfCheetah envs, (64, 64): one env per wavefront
    assert _plan(_lib.ENV_HALF_CHEETAH, 1024, (64, 64)) == (7, 1, 1024, 4, 256, "rollout_two_leg_wave_kernel<HalfCheetah, 64>")
    # ... beyond 2048 envs a lane group per env, 16 envs per wavefront
    assert _plan(_lib.ENV_HALF_CHEETAH, 8192, (64, 64))[:3] == (8, 16, 512)
    # C2: Cartpole, the generic kernel; one env per lane beyond 16 384 envs
    assert _plan(_lib.ENV_CARTPOLE, 4096, (32, 32), T=100) == (1, 16, 256, 1, 256, "rollout_kernel<Cartpole, 32, 32, 16>")
    assert _plan(_lib.ENV_CARTPOLE, 65536, (32, 32), T=100)[:3] == (1, 64, 1024)
    # wide nets on the Swimmer: four wavefronts per group of 16 envs while that is at most 1024 wavefronts
    k = _plan(_lib.ENV_SWIMMER, 4096, (128, 128))
    assert k[0] == 6 and k[2] == 1024 and k[3] == 4 and k[5] == "rollout_swimmer_quad_coop_kernel"
    assert _plan(_lib.ENV_SWIMMER, 16384, (128, 128))[0] == 5
    # SwimmerEnv(limit_model="mujoco") lives in the scalar sub-step program: the generic kernel
    assert _plan(_lib.ENV_SWIMMER, 4096, (32, 32), flags=_lib.CFG_LIMIT_MUJOCO)[0] == 1


def test_unsupported_shapes_say_why(clean_env):
    from rllab_amd import _lib
    assert _plan(_lib.ENV_HOPPER, 512, (128, 128), std=(128, 128, 0)) is None            # 172 KB of weight fragments
    assert "LDS" in _lib.lib.rl_last_error().decode()
    assert _plan(_lib.ENV_SWIMMER, 512, (256, 256)) is None
    assert "no fused kernel" in _lib.lib.rl_last_error().decode()
    assert _plan(_lib.ENV_HOPPER, 512, (128, 128), std=(32, 32, 0))[0] == 3              # this pair fits


def test_every_switch_is_a_field_of_the_options_struct(clean_env):
    from rllab_amd import _lib
    mp = clean_env
    mp.setenv("RLLAB_ROLLOUT_EPW", "64")
    assert _plan(_lib.ENV_CARTPOLE, 4096, (32, 32))[:3] == (1, 64, 64)
    assert _plan(_lib.ENV_HALF_CHEETAH, 1024, (64, 64))[:3] == (1, 64, 16)        # (keeps the lane-group kernels out too)
    assert _plan(_lib.ENV_SWIMMER, 4096, (128, 128))[:3] == (2, 64, 64)
    mp.setenv("RLLAB_ROLLOUT_EPW", "16")
    assert _plan(_lib.ENV_CARTPOLE, 65536, (32, 32))[:3] == (1, 16, 4096)
    mp.delenv("RLLAB_ROLLOUT_EPW")
    mp.setenv("RLLAB_ROLLOUT_WPB", "4")
    assert _plan(_lib.ENV_SWIMMER, 4096, (32, 32))[3:5] == (4, 64)
    mp.delenv("RLLAB_ROLLOUT_WPB")
    mp.setenv("RLLAB_SWIMMER_LANE_KERNEL", "1")
    assert _plan(_lib.ENV_SWIMMER, 4096, (32, 32))[0] == 1
    mp.delenv("RLLAB_SWIMMER_LANE_KERNEL")
    mp.setenv("RLLAB_SWIMMER_COOP", "0")
    assert _plan(_lib.ENV_SWIMMER, 4096, (128, 128))[0] == 5
    mp.setenv("RLLAB_SWIMMER_COOP", "1")
    assert _plan(_lib.ENV_SWIMMER, 16384, (128, 128))[0] == 6
    mp.delenv("RLLAB_SWIMMER_COOP")
    mp.setenv("RLLAB_TWO_LEG_WAVE_KERNEL", "0")
    assert _plan(_lib.ENV_WALKER2D, 1024, (32, 32))[0] == 8
    mp.setenv("RLLAB_TWO_LEG_WAVE_KERNEL", "1")
    assert _plan(_lib.ENV_WALKER2D, 4096, (32, 32))[:3] == (7, 1, 4096)
    mp.delenv("RLLAB_TWO_LEG_WAVE_KERNEL")
    mp.setenv("RLLAB_TWO_LEG_LANE_KERNEL", "0")
    assert _plan(_lib.ENV_HALF_CHEETAH, 1024, (64, 64))[0] == 1
    mp.delenv("RLLAB_TWO_LEG_LANE_KERNEL")
    # the remaining fields steer the update's kernels: launch_opts maps the switch values onto the struct
    import ctypes
    for name, val, field, want in (("RLLAB_FVP_SPLIT", "0", "fvp_split", 1), ("RLLAB_FVP_SPLIT", "2", "fvp_split", 2),
                                   ("RLLAB_FVP_SPLIT_WPS", "1", "fvp_split_wps", 1), ("RLLAB_LFB_VALU", "1", "lfb_valu", 1)):
        mp.setenv(name, val)
        o = ctypes.cast(_lib.launch_opts(), ctypes.POINTER(_lib.LaunchOpts)).contents
        assert getattr(o, field) == want
        mp.delenv(name)
    o = ctypes.cast(_lib.launch_opts(), ctypes.POINTER(_lib.LaunchOpts)).contents
    assert all(getattr(o, f) == 0 for f, _ in _lib.LaunchOpts._fields_[:-1])


def test_header_documents_every_field():
    text = open(os.path.join(ROOT, "include", "rllab_amd.h")).read()
    body = text[text.index("typedef struct rl_launch_opts {"):text.index("} rl_launch_opts;")]
    from rllab_amd import _lib
    fields = re.findall(r"int32_t (\w+)(?:\[\d+\])?;", body)
    assert fields == [f for f, _ in _lib.LaunchOpts._fields_]
