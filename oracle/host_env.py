"""ctypes face of oracle/_build/liboracle_env.so (host build of the env dynamics
headers; see oracle/env_host.cpp for what it pins and what it cannot pin)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "liboracle_env.so")


def build(force=False):
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return LIB_PATH


def _load():
    build()
    lib = ctypes.CDLL(LIB_PATH)
    return lib


lib = _load()
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_ip = ctypes.POINTER(ctypes.c_int)
lib.oracle_env_query.argtypes = [ctypes.c_int, _ip, _ip, _ip, _ip, _ip]
lib.oracle_env_action_bounds.argtypes = [ctypes.c_int, _f64p, _f64p]
lib.oracle_env_reset_f32.argtypes = [ctypes.c_int, _f32p, _f32p]
lib.oracle_env_reset_f64.argtypes = [ctypes.c_int, _f64p, _f64p]
lib.oracle_env_observe_f32.argtypes = [ctypes.c_int, _f32p, _f32p]
lib.oracle_env_observe_f64.argtypes = [ctypes.c_int, _f64p, _f64p]
lib.oracle_env_step_f32.argtypes = [ctypes.c_int, _f32p, _f32p, ctypes.c_int, _f32p, _f32p, _ip]
lib.oracle_env_step_f64.argtypes = [ctypes.c_int, _f64p, _f64p, ctypes.c_int, _f64p, _f64p, _ip]
lib.oracle_vecenv_step_f32.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int,
                                       ctypes.c_int, _f32p, _i32p, _f32p, ctypes.c_void_p, _f32p, _f32p, _u8p]
lib.oracle_vecenv_reset_f32.argtypes = [ctypes.c_int, ctypes.c_int, _f32p, _i32p, ctypes.c_void_p, _f32p, _f32p]
lib.oracle_philox.argtypes = [ctypes.c_uint32] * 6 + [ctypes.c_int, _u32p]
lib.oracle_sincos_f32.argtypes = [ctypes.c_int, _f32p, _f32p, _f32p]
lib.oracle_sincos_f32.restype = None


def query(kind):
    vals = [ctypes.c_int() for _ in range(5)]
    assert lib.oracle_env_query(kind, *[ctypes.byref(v) for v in vals]) == 0, "unknown env kind %d" % kind
    o, a, s, r, nrm = [v.value for v in vals]
    return dict(obs_dim=o, act_dim=a, state_dim=s, reset_draws=r, reset_is_normal=bool(nrm))


def action_bounds(kind):
    q = query(kind)
    lb, ub = np.zeros(q["act_dim"]), np.zeros(q["act_dim"])
    assert lib.oracle_env_action_bounds(kind, lb, ub) == 0
    return lb, ub


class HostEnv(object):
    """One env copy on the CPU (float32 = bit-exact leg, float64 = physics leg)."""

    def __init__(self, kind, dtype=np.float32, normalize=False):
        self.kind, self.dtype, self.normalize = kind, np.dtype(dtype), int(normalize)
        self.q = query(kind)
        self.state = np.zeros(self.q["state_dim"], dtype=self.dtype)
        sfx = "f32" if self.dtype == np.float32 else "f64"
        self._reset = getattr(lib, "oracle_env_reset_" + sfx)
        self._step = getattr(lib, "oracle_env_step_" + sfx)
        self._observe = getattr(lib, "oracle_env_observe_" + sfx)

    def reset(self, draws):
        draws = np.ascontiguousarray(draws, dtype=self.dtype)
        assert draws.shape == (self.q["reset_draws"],)
        assert self._reset(self.kind, self.state, draws) == 0
        return self.observe()

    def observe(self):
        o = np.zeros(self.q["obs_dim"], dtype=self.dtype)
        assert self._observe(self.kind, self.state, o) == 0
        return o

    def step(self, action):
        a = np.ascontiguousarray(action, dtype=self.dtype).reshape(self.q["act_dim"])
        o = np.zeros(self.q["obs_dim"], dtype=self.dtype)
        r = np.zeros(1, dtype=self.dtype)
        d = ctypes.c_int()
        assert self._step(self.kind, self.state, a, self.normalize, o, r, ctypes.byref(d)) == 0
        return o, r[0], bool(d.value)


class HostVecEnv(object):
    """Serial float32 replay of the lock-step executor on the GPU's plane layout."""

    def __init__(self, kind, n, max_path_length=0, normalize=False, scale_reward=1.0, auto_reset=True):
        self.kind, self.n = kind, n
        self.q = query(kind)
        self.max_path_length, self.normalize = int(max_path_length or 0), int(normalize)
        self.scale_reward, self.auto_reset = float(scale_reward), int(auto_reset)
        self.state = np.zeros((self.q["state_dim"], n), np.float32)
        self.ts = np.zeros(n, np.int32)

    def reset(self, draws, mask=None):
        draws = np.ascontiguousarray(draws, np.float32)
        obs = np.zeros((self.q["obs_dim"], self.n), np.float32)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8).ctypes.data
        assert lib.oracle_vecenv_reset_f32(self.kind, self.n, self.state, self.ts, m, draws, obs) == 0
        return obs

    def step(self, actions, reset_draws=None):
        actions = np.ascontiguousarray(actions, np.float32)
        obs = np.zeros((self.q["obs_dim"], self.n), np.float32)
        rew = np.zeros(self.n, np.float32)
        done = np.zeros(self.n, np.uint8)
        rd = None
        if reset_draws is not None:
            reset_draws = np.ascontiguousarray(reset_draws, np.float32)
            rd = reset_draws.ctypes.data
        assert lib.oracle_vecenv_step_f32(self.kind, self.n, self.normalize, self.scale_reward,
                                          self.max_path_length, self.auto_reset, self.state, self.ts, actions,
                                          rd, obs, rew, done) == 0
        return obs, rew, done


def philox(c0, c1, c2, c3, k0, k1, count):
    out = np.zeros(4 * count, np.uint32)
    lib.oracle_philox(c0, c1, c2, c3, k0, k1, count, out)
    return out.reshape(count, 4)


def sincos_f32(x):
    x = np.ascontiguousarray(x, np.float32)
    s, c = np.zeros_like(x), np.zeros_like(x)
    lib.oracle_sincos_f32(x.size, x, s, c)
    return s, c


lib.oracle_swim_quad_compare_f32.argtypes = [_f32p, _f32p, ctypes.c_int, _f32p, _f32p]
lib.oracle_swim_quad_compare_f64.argtypes = [_f64p, _f64p, ctypes.c_int, _f64p, _f64p]


def swim_quad_compare(state, ctrl, nsub, dtype=np.float32):
    """(scalar, emulated-quad) results of ``nsub`` swimmer sub-steps from one state: 16 values each
    (qpos 5, qvel 5, carried sin 3, cos 3)."""
    dtype = np.dtype(dtype)
    fn = lib.oracle_swim_quad_compare_f32 if dtype == np.float32 else lib.oracle_swim_quad_compare_f64
    a, b = np.zeros(16, dtype), np.zeros(16, dtype)
    fn(np.ascontiguousarray(state, dtype), np.ascontiguousarray(ctrl, dtype), int(nsub), a, b)
    return a, b
