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


class OracleCfg(ctypes.Structure):
    """Env options (struct OracleCfg of env_host.cpp; the C ABI's rl_env_cfg without the injected-draw pointers)."""
    _fields_ = [("ctrl_cost_coeff", ctypes.c_double), ("alive_coeff", ctypes.c_double),
                ("action_noise", ctypes.c_double), ("obs_noise", ctypes.c_double),
                ("frame_skip", ctypes.c_int), ("flags", ctypes.c_int), ("link_len", ctypes.c_double)]


CFG_POLE_FOLLOWS_CART, CFG_FIXED_START = 1, 2
_cfgp = ctypes.POINTER(OracleCfg)
lib.oracle_env_default_cfg.argtypes = [ctypes.c_int, _cfgp]
for _sfx, _p in (("f32", _f32p), ("f64", _f64p)):
    getattr(lib, "oracle_env_reset_cfg_" + _sfx).argtypes = [ctypes.c_int, _p, _p, _cfgp]
    getattr(lib, "oracle_env_step_cfg_" + _sfx).argtypes = [ctypes.c_int, _p, _p, ctypes.c_int, _cfgp, ctypes.c_void_p,
                                                            _p, _p, _ip]
    getattr(lib, "oracle_env_obs_noise_" + _sfx).argtypes = [ctypes.c_int, _cfgp, _p, _p]
    getattr(lib, "oracle_env_com_" + _sfx).argtypes = [ctypes.c_int, _p, _p]
lib.oracle_vecenv_step_cfg_f32.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int,
                                           ctypes.c_int, _f32p, _i32p, _f32p, ctypes.c_void_p, _f32p, _f32p, _u8p,
                                           _cfgp, ctypes.c_void_p, ctypes.c_void_p]
lib.oracle_vecenv_reset_cfg_f32.argtypes = [ctypes.c_int, ctypes.c_int, _f32p, _i32p, ctypes.c_void_p, _f32p, _f32p,
                                            _cfgp, ctypes.c_void_p]


def default_cfg(kind):
    c = OracleCfg()
    assert lib.oracle_env_default_cfg(kind, ctypes.byref(c)) == 0, "unknown env kind %d" % kind
    return c


def make_cfg(kind, cfg=None):
    """OracleCfg of env ``kind``: its defaults overridden by the dict ``cfg`` (keys = field names)."""
    c = default_cfg(kind)
    for k, v in (cfg or {}).items():
        if k not in dict(OracleCfg._fields_):
            raise KeyError("unknown env option %r" % (k,))
        setattr(c, k, v)
    return c


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

    def __init__(self, kind, dtype=np.float32, normalize=False, cfg=None):
        """``cfg``: dict of env options (OracleCfg fields) or None = the env's defaults."""
        self.kind, self.dtype, self.normalize = kind, np.dtype(dtype), int(normalize)
        self.q = query(kind)
        self.cfg = make_cfg(kind, cfg)
        self.state = np.zeros(self.q["state_dim"], dtype=self.dtype)
        sfx = "f32" if self.dtype == np.float32 else "f64"
        self._reset = getattr(lib, "oracle_env_reset_cfg_" + sfx)
        self._step = getattr(lib, "oracle_env_step_cfg_" + sfx)
        self._observe = getattr(lib, "oracle_env_observe_" + sfx)
        self._obs_noise = getattr(lib, "oracle_env_obs_noise_" + sfx)
        self._com = getattr(lib, "oracle_env_com_" + sfx)

    def reset(self, draws, zobs=None):
        draws = np.ascontiguousarray(draws, dtype=self.dtype)
        assert draws.shape == (self.q["reset_draws"],)
        assert self._reset(self.kind, self.state, draws, ctypes.byref(self.cfg)) == 0
        return self.observe(zobs)

    def observe(self, zobs=None):
        """Observation of the present state; with ``zobs`` (N(0,1) draws) and obs_noise != 0 the noisy one."""
        o = np.zeros(self.q["obs_dim"], dtype=self.dtype)
        assert self._observe(self.kind, self.state, o) == 0
        return self.noisy(o, zobs)

    def noisy(self, o, zobs):
        if self.cfg.obs_noise != 0.0:
            assert zobs is not None, "obs_noise is on: the N(0,1) draws must be supplied"
            z = np.ascontiguousarray(zobs, dtype=self.dtype).reshape(self.q["obs_dim"])
            assert self._obs_noise(self.kind, ctypes.byref(self.cfg), z, o) == 0
        return o

    def com(self):
        """(forward, up) position and velocity of the torso subtree's centre of mass (MuJoCo-style envs)."""
        c = np.zeros(4, dtype=self.dtype)
        assert self._com(self.kind, self.state, c) == 0, "env kind %d has no subtree COM" % self.kind
        return c

    def step(self, action, zact=None, zobs=None):
        a = np.ascontiguousarray(action, dtype=self.dtype).reshape(self.q["act_dim"])
        o = np.zeros(self.q["obs_dim"], dtype=self.dtype)
        r = np.zeros(1, dtype=self.dtype)
        d = ctypes.c_int()
        zp = None
        if self.cfg.action_noise != 0.0:
            assert zact is not None, "action_noise is on: the N(0,1) draws must be supplied"
            zact = np.ascontiguousarray(zact, dtype=self.dtype).reshape(self.q["act_dim"])
            zp = zact.ctypes.data
        assert self._step(self.kind, self.state, a, self.normalize, ctypes.byref(self.cfg), zp, o, r,
                          ctypes.byref(d)) == 0
        return self.noisy(o, zobs), r[0], bool(d.value)


class HostVecEnv(object):
    """Serial float32 replay of the lock-step executor on the GPU's plane layout."""

    def __init__(self, kind, n, max_path_length=0, normalize=False, scale_reward=1.0, auto_reset=True, cfg=None):
        self.kind, self.n = kind, n
        self.q = query(kind)
        self.cfg = make_cfg(kind, cfg)
        self.max_path_length, self.normalize = int(max_path_length or 0), int(normalize)
        self.scale_reward, self.auto_reset = float(scale_reward), int(auto_reset)
        self.state = np.zeros((self.q["state_dim"], n), np.float32)
        self.ts = np.zeros(n, np.int32)

    @staticmethod
    def _plane(z):
        if z is None:
            return None, None
        z = np.ascontiguousarray(z, np.float32)
        return z, z.ctypes.data

    def reset(self, draws, mask=None, obs_z=None):
        draws = np.ascontiguousarray(draws, np.float32)
        obs = np.zeros((self.q["obs_dim"], self.n), np.float32)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8).ctypes.data
        keep, zo = self._plane(obs_z)
        assert lib.oracle_vecenv_reset_cfg_f32(self.kind, self.n, self.state, self.ts, m, draws, obs,
                                               ctypes.byref(self.cfg), zo) == 0
        return obs

    def step(self, actions, reset_draws=None, act_z=None, obs_z=None):
        actions = np.ascontiguousarray(actions, np.float32)
        obs = np.zeros((self.q["obs_dim"], self.n), np.float32)
        rew = np.zeros(self.n, np.float32)
        done = np.zeros(self.n, np.uint8)
        rd = None
        if reset_draws is not None:
            reset_draws = np.ascontiguousarray(reset_draws, np.float32)
            rd = reset_draws.ctypes.data
        keep_a, za = self._plane(act_z)
        keep_o, zo = self._plane(obs_z)
        assert self.cfg.action_noise == 0.0 or za is not None
        assert self.cfg.obs_noise == 0.0 or zo is not None
        assert lib.oracle_vecenv_step_cfg_f32(self.kind, self.n, self.normalize, self.scale_reward,
                                              self.max_path_length, self.auto_reset, self.state, self.ts, actions,
                                              rd, obs, rew, done, ctypes.byref(self.cfg), za, zo) == 0
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


lib.oracle_two_leg_compare_f32.argtypes = [ctypes.c_int, _f32p, _f32p, ctypes.c_int, _f32p, _f32p]
lib.oracle_two_leg_compare_f64.argtypes = [ctypes.c_int, _f64p, _f64p, ctypes.c_int, _f64p, _f64p]


lib.oracle_two_leg_lane_table.argtypes = [ctypes.c_int, _f64p]


def two_leg_lane_table(kind):
    """dyn_two_legs.h's per-lane constants of a two-legged env, [8 lanes = 4 * leg + role][20]: jx, jy, cx, cy, mass, inertia,
    arm, stiff, damp, lo, hi, mc, cpx[2], cpy[2], crad[2], cmu[2]."""
    out = np.zeros((8, 20), np.float64)
    assert lib.oracle_two_leg_lane_table(int(kind), out) == 0
    return out


lib.oracle_two_leg_quad_form_f32.argtypes = [ctypes.c_int, _f32p, _f32p, ctypes.c_int, _f32p]
lib.oracle_two_leg_quad_form_f64.argtypes = [ctypes.c_int, _f64p, _f64p, ctypes.c_int, _f64p]


def two_leg_quad_form(kind, state, tau, nsub, dtype=np.float32):
    """The same ``nsub`` sub-steps in the quad form (four emulated role lanes, both legs side by side in two-component
    values: the 16-envs-per-wavefront rollout): 22 values laid out like ``two_leg_compare``'s."""
    dtype = np.dtype(dtype)
    fn = lib.oracle_two_leg_quad_form_f32 if dtype == np.float32 else lib.oracle_two_leg_quad_form_f64
    out = np.zeros(22, dtype)
    rc = fn(int(kind), np.ascontiguousarray(state, dtype), np.ascontiguousarray(tau, dtype), int(nsub), out)
    assert rc == 0
    return out


def two_leg_compare(kind, state, tau, nsub, dtype=np.float32):
    """(eight-component, emulated one-body-per-scalar-lane) results of ``nsub`` sub-steps of a two-legged env (kind 3 / 5)
    from one state (q[9], qd[9]) under hinge torques tau[7]: 22 values each (q, qd, centre of mass and its velocity)."""
    dtype = np.dtype(dtype)
    fn = lib.oracle_two_leg_compare_f32 if dtype == np.float32 else lib.oracle_two_leg_compare_f64
    a, b = np.zeros(22, dtype), np.zeros(22, dtype)
    rc = fn(int(kind), np.ascontiguousarray(state, dtype), np.ascontiguousarray(tau, dtype), int(nsub), a, b)
    assert rc == 0
    return a, b
