"""CPU oracle for the UAV env step/reset path — TEST INFRASTRUCTURE ONLY.

ctypes front-end of oracle/uavx_oracle.c (a scalar C restatement of the reference's
multi_uav_world_2d.py / uav_agent.py / uav_world_2d.py).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this package; the product package never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libuavx_oracle.so")

OBS_DIM = 10
UW_OBS_DIM = 4
FLAG_DONE = 1
FLAG_COLLIDED = 2


def build(force=False):
    """Compile the oracle with gcc (make -C oracle)."""
    src = os.path.join(_HERE, "uavx_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src),
                                                   os.path.getmtime(os.path.join(_HERE, "uavx_oracle.h")))):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE] + (["-B"] if force else []),
                          stdout=subprocess.DEVNULL)
    return _LIB_PATH


class _Config(ctypes.Structure):
    _fields_ = [("x_size", ctypes.c_double), ("y_size", ctypes.c_double),
                ("max_speed", ctypes.c_double), ("max_acceleration", ctypes.c_double),
                ("collider_radius", ctypes.c_double), ("d_sense", ctypes.c_double),
                ("tau", ctypes.c_double), ("num_agents", ctypes.c_int32), ("_pad", ctypes.c_int32)]


class _State(ctypes.Structure):
    _fields_ = [("num_envs", ctypes.c_int64), ("num_agents", ctypes.c_int32), ("_pad", ctypes.c_int32),
                ("loc", ctypes.c_void_p), ("vel", ctypes.c_void_p), ("tgt", ctypes.c_void_p),
                ("init_d", ctypes.c_void_p), ("prev_d", ctypes.c_void_p), ("flags", ctypes.c_void_p),
                ("counters", ctypes.c_void_p), ("f64pos", ctypes.c_void_p)]


class _StepOpts(ctypes.Structure):
    _fields_ = [("action_mode", ctypes.c_int32), ("reset_policy", ctypes.c_int32), ("track_returns", ctypes.c_int32),
                ("step_cap", ctypes.c_uint32), ("seed", ctypes.c_uint64), ("env_offset", ctypes.c_int64)]


class _EpisodeState(ctypes.Structure):
    _fields_ = [("pending", ctypes.c_void_p), ("ep_run", ctypes.c_void_p), ("fin_counts", ctypes.c_void_p),
                ("fin_returns", ctypes.c_void_p)]


class _Level(ctypes.Structure):
    _fields_ = [("x_size", ctypes.c_double), ("y_size", ctypes.c_double), ("collider_radius", ctypes.c_double),
                ("d_sense", ctypes.c_double), ("n_active", ctypes.c_int32), ("b_active", ctypes.c_int32)]


class _Ext(ctypes.Structure):
    _fields_ = [("num_bodies", ctypes.c_int32), ("body_period", ctypes.c_int32), ("body_speed", ctypes.c_double),
                ("body_seed", ctypes.c_uint64), ("n_levels", ctypes.c_int32), ("level_lo", ctypes.c_int32),
                ("level_hi", ctypes.c_int32), ("_pad", ctypes.c_int32), ("levels", ctypes.c_void_p)]


class _ExtState(ctypes.Structure):
    _fields_ = [("body", ctypes.c_void_p), ("level", ctypes.c_void_p), ("next_level", ctypes.c_void_p)]


class _UWConfig(ctypes.Structure):
    _fields_ = [("x_size", ctypes.c_double), ("y_size", ctypes.c_double), ("max_speed", ctypes.c_double),
                ("max_acceleration", ctypes.c_double), ("tau", ctypes.c_double)]


class _UWState(ctypes.Structure):
    _fields_ = [("num_envs", ctypes.c_int64), ("loc", ctypes.c_void_p), ("vel", ctypes.c_void_p),
                ("tgt", ctypes.c_void_p), ("init_d", ctypes.c_void_p), ("prev_d", ctypes.c_void_p),
                ("steps", ctypes.c_void_p), ("episode", ctypes.c_void_p), ("vel_f32", ctypes.c_void_p)]


class _UWEpisodeState(ctypes.Structure):
    _fields_ = [("pending", ctypes.c_void_p), ("reached", ctypes.c_void_p), ("ep_return", ctypes.c_void_p),
                ("fin_counts", ctypes.c_void_p), ("fin_return", ctypes.c_void_p)]


class _MT(ctypes.Structure):
    _fields_ = [("mt", ctypes.c_uint32 * 624), ("idx", ctypes.c_int32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        vp, i64, i32, u64, u32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint32
        L.uavo_mt_seed.argtypes = [vp, u32]
        L.uavo_mt_double.argtypes = [vp]
        L.uavo_mt_double.restype = ctypes.c_double
        L.uavo_philox4x32.argtypes = [vp, vp, vp]
        L.uavo_reset_mt.argtypes = [vp, vp, i64, vp, i32]
        L.uavo_reset_philox.argtypes = [vp, vp, vp, u64, i64, i32]
        L.uavo_observe.argtypes = [vp, vp, vp, i32]
        L.uavo_step.argtypes = [vp, vp, vp, i32, vp, vp, vp, i32]
        L.uavo_polar_to_command.argtypes = [ctypes.c_float, ctypes.c_float, ctypes.c_float, vp]
        L.uavo_polar_to_command.restype = None
        L.uavo_step_ex.argtypes = [vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, i32]
        L.uavo_step_ex.restype = None
        L.uavo_fold_episode.argtypes = [vp, vp, i64]
        L.uavo_fold_episode.restype = None
        L.uavo_uw_fold_episode.argtypes = [vp, vp, i64]
        L.uavo_uw_fold_episode.restype = None
        L.uavo_uw_step_ex.argtypes = [vp, vp, vp, i32, i32, u32, i32, u64, i64, vp, i32, vp, vp, vp, vp, vp, i32]
        L.uavo_uw_step_ex.restype = None
        L.uavo_reset_philox_x.argtypes = [vp, vp, vp, vp, vp, u64, i64, i32]
        L.uavo_observe_x.argtypes = [vp, vp, vp, vp, vp, i32]
        L.uavo_step_x.argtypes = [vp, vp, vp, vp, vp, i32, i64, vp, vp, vp, i32]
        L.uavo_step_ex_x.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, i32]
        for f in ("uavo_reset_philox_x", "uavo_observe_x", "uavo_step_x", "uavo_step_ex_x"):
            getattr(L, f).restype = None
        L.uavo_uw_reset_mt.argtypes = [vp, vp, i64, vp]
        L.uavo_uw_reset_philox.argtypes = [vp, vp, vp, u64, i64, i32]
        L.uavo_uw_observe.argtypes = [vp, vp, vp, i32]
        L.uavo_uw_step.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp, i32]
        for f in ("uavo_mt_seed", "uavo_philox4x32", "uavo_reset_mt", "uavo_reset_philox", "uavo_observe",
                  "uavo_step", "uavo_uw_reset_mt", "uavo_uw_reset_philox", "uavo_uw_observe", "uavo_uw_step"):
            getattr(L, f).restype = None
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def polar_to_command(a0, a1, vmax_norm):
    out = np.zeros(2, np.float64)
    lib().uavo_polar_to_command(float(a0), float(a1), float(vmax_norm), _p(out))
    return out


def philox4x32(ctr, key):
    c = np.asarray(ctr, dtype=np.uint32)
    k = np.asarray(key, dtype=np.uint32)
    o = np.zeros(4, dtype=np.uint32)
    lib().uavo_philox4x32(_p(c), _p(k), _p(o))
    return o


class MTStream:
    """numpy-legacy MT19937 stream (np.random.seed(seed) equivalent) consumed by the MT resets."""

    def __init__(self, seed):
        self._g = _MT()
        lib().uavo_mt_seed(ctypes.byref(self._g), int(seed))

    def random_sample(self):
        return lib().uavo_mt_double(ctypes.byref(self._g))


class OracleMulti:
    """E independent MultiUAVWorld2D worlds stepped by the C restatement (MUW:10-241)."""

    def __init__(self, num_envs=1, x_size=50.0, y_size=50.0, max_speed=10.0, max_acceleration=5.0,
                 num_agents=4, collider_radius=1.0, d_sense=15, tau=0.02, nthreads=1,
                 num_bodies=0, body_speed=5.0, body_period=128, body_seed=0):
        """num_agents = learners L; num_bodies = scripted bodies B of the configs[4] extension (slots L..L+B-1)."""
        assert 1 <= num_agents and num_agents + num_bodies <= 64
        self.E, self.N = int(num_envs), int(num_agents)
        self.B = int(num_bodies)
        self.nthreads = int(nthreads)
        self.cfg = _Config(x_size, y_size, max_speed, max_acceleration, collider_radius, float(d_sense),
                           tau, num_agents, 0)
        self._world = dict(x_size=x_size, y_size=y_size, max_speed=max_speed, max_acceleration=max_acceleration,
                           collider_radius=collider_radius, d_sense=d_sense)
        E, N = self.E, self.N
        self.loc = np.zeros((E, N, 2), np.float64)
        self.vel = np.zeros((E, N, 2), np.float64)
        self.tgt = np.zeros((E, N, 2), np.float64)
        self.init_d = np.zeros((E, N), np.float64)
        self.prev_d = np.zeros((E, N), np.float64)
        self.flags = np.zeros((E, N), np.uint8)
        self.counters = np.zeros((E, 4), np.uint32)
        self.f64pos = np.zeros((E,), np.uint8)
        self._st = _State(E, N, 0, _p(self.loc), _p(self.vel), _p(self.tgt), _p(self.init_d),
                          _p(self.prev_d), _p(self.flags), _p(self.counters), _p(self.f64pos))
        # episode bookkeeping of uavx_step_ex / uavx_reset
        self.pending = np.zeros((E,), np.uint8)
        self.ep_run = np.zeros((E, 2), np.float32)
        self.fin_counts = np.zeros((E, 4), np.uint32)
        self.fin_returns = np.zeros((E, 2), np.float32)
        self._ep = _EpisodeState(_p(self.pending), _p(self.ep_run), _p(self.fin_counts), _p(self.fin_returns))
        # configs[4] extension (scripted bodies, curriculum levels); inert when B == 0 and no levels are installed
        self.body = np.zeros((E, max(self.B, 1), 6), np.float32)[:, :self.B]   # x, y, dx, dy, heading, legs
        self.body = np.ascontiguousarray(self.body)
        self.level = np.zeros((E,), np.uint8)
        self.next_level = np.zeros((E,), np.uint8)
        self._levels = None
        self._ext = _Ext(self.B, int(body_period), float(body_speed), int(body_seed), 0, -1, -1, 0, None)
        self._xs = _ExtState(_p(self.body) if self.B else None, _p(self.level), _p(self.next_level))

    def _x(self):
        """(ext, ext_state) pointers, or (None, None) for a plain reference-shaped world."""
        if self.B == 0 and self._ext.n_levels == 0:
            return None, None
        return ctypes.byref(self._ext), ctypes.byref(self._xs)

    def set_body_rule(self, speed=None, period=None, seed=None):
        if speed is not None:
            self._ext.body_speed = float(speed)
        if period is not None:
            self._ext.body_period = int(period)
        if seed is not None:
            self._ext.body_seed = int(seed)

    def set_curriculum(self, levels, lo=-1, hi=-1):
        """levels: list of dicts(x_size, y_size, collider_radius, d_sense, n_active, b_active); resets draw an env's
        level uniformly in [lo, hi], or take next_level[e] when lo < 0."""
        arr = (_Level * len(levels))(*[_Level(float(l["x_size"]), float(l["y_size"]), float(l["collider_radius"]),
                                              float(l["d_sense"]), int(l.get("n_active", self.N)),
                                              int(l.get("b_active", self.B))) for l in levels])
        self._levels = arr
        self._ext.n_levels = len(levels)
        self._ext.levels = ctypes.cast(arr, ctypes.c_void_p)
        self._ext.level_lo, self._ext.level_hi = int(lo), int(hi)

    def set_level_window(self, lo, hi):
        self._ext.level_lo, self._ext.level_hi = int(lo), int(hi)

    def set_env_levels(self, levels):
        self.next_level[...] = np.asarray(levels, dtype=np.uint8).reshape(self.E)

    # -- state exchange with the device path (float32 positions) -----------------------------------
    def get_state(self):
        return dict(loc=self.loc.astype(np.float32), vel=self.vel.copy(), tgt=self.tgt.astype(np.float32),
                    init_d=self.init_d.astype(np.float32), prev_d=self.prev_d.astype(np.float32),
                    flags=self.flags.copy(), counters=self.counters.copy())

    def set_state(self, loc=None, vel=None, tgt=None, init_d=None, prev_d=None, flags=None, counters=None):
        for name, val in (("loc", loc), ("vel", vel), ("tgt", tgt), ("init_d", init_d), ("prev_d", prev_d),
                          ("flags", flags), ("counters", counters)):
            if val is not None:
                getattr(self, name)[...] = np.asarray(val).reshape(getattr(self, name).shape)

    def set_config(self, **world):
        """Checker side of uavx_set_config: every uavo_* call takes the config by pointer, so later calls simply see
        the new scalar world parameters; state is untouched."""
        self._world.update(world)
        w = self._world
        self.cfg = _Config(w["x_size"], w["y_size"], w["max_speed"], w["max_acceleration"], w["collider_radius"],
                           float(w["d_sense"]), self.cfg.tau, self.N, 0)

    def reset_mt(self, stream, env=0, circular=False):
        lib().uavo_reset_mt(ctypes.byref(self.cfg), ctypes.byref(self._st), env,
                            ctypes.byref(stream._g), int(bool(circular)))

    def reset_philox(self, seed, mask=None, env_offset=0):
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        for e in range(self.E):  # an explicit reset ends the running episode (uavx_reset does the same)
            if m is None or m[e]:
                lib().uavo_fold_episode(ctypes.byref(self._st), ctypes.byref(self._ep), e)
        ext, xs = self._x()
        lib().uavo_reset_philox_x(ctypes.byref(self.cfg), ext, ctypes.byref(self._st), xs,
                                  None if m is None else _p(m), int(seed), int(env_offset), self.nthreads)

    def step_ex(self, actions, evaluate=False, action_mode=0, reset_policy=0, step_cap=0, track_returns=False,
                seed=0, env_offset=0, with_end=False):
        """with_end=True additionally returns (ended, truncated) [E] uint8 of the ending call."""
        a = np.ascontiguousarray(np.asarray(actions, dtype=np.float64).reshape(self.E, self.N, 2))
        obs = np.empty((self.E, self.N, OBS_DIM), np.float64)
        rew = np.empty((self.E, self.N), np.float64)
        done = np.empty((self.E, self.N), np.uint8)
        rmask = np.zeros((self.E,), np.uint8)
        ended = np.zeros((self.E,), np.uint8)
        trunc = np.zeros((self.E,), np.uint8)
        opt = _StepOpts(int(action_mode), int(reset_policy), int(bool(track_returns)), int(step_cap), int(seed),
                        int(env_offset))
        ext, xs = self._x()
        lib().uavo_step_ex_x(ctypes.byref(self.cfg), ext, ctypes.byref(self._st), xs, ctypes.byref(self._ep),
                             ctypes.byref(opt), _p(a), int(bool(evaluate)), _p(obs), _p(rew), _p(done), _p(rmask),
                             _p(ended), _p(trunc), self.nthreads)
        if with_end:
            return obs, rew, done, rmask, ended, trunc
        return obs, rew, done, rmask

    def observe(self):
        obs = np.empty((self.E, self.N, OBS_DIM), np.float64)
        ext, xs = self._x()
        lib().uavo_observe_x(ctypes.byref(self.cfg), ext, ctypes.byref(self._st), xs, _p(obs), self.nthreads)
        return obs

    def step(self, actions, evaluate=False, env_offset=0):
        a = np.ascontiguousarray(np.asarray(actions, dtype=np.float64).reshape(self.E, self.N, 2))
        obs = np.empty((self.E, self.N, OBS_DIM), np.float64)
        rew = np.empty((self.E, self.N), np.float64)
        done = np.empty((self.E, self.N), np.uint8)
        ext, xs = self._x()
        lib().uavo_step_x(ctypes.byref(self.cfg), ext, ctypes.byref(self._st), xs, _p(a), int(bool(evaluate)),
                          int(env_offset), _p(obs), _p(rew), _p(done), self.nthreads)
        return obs, rew, done


class OracleSingle:
    """E independent UAVWorld2D worlds (UW:11-173)."""

    def __init__(self, num_envs=1, x_size=100.0, y_size=100.0, max_speed=12.0, max_acceleration=5.0,
                 tau=0.02, nthreads=1):
        self.E = int(num_envs)
        self.nthreads = int(nthreads)
        self.cfg = _UWConfig(x_size, y_size, max_speed, max_acceleration, tau)
        E = self.E
        self.loc = np.zeros((E, 2), np.float64)
        self.vel = np.zeros((E, 2), np.float64)
        self.tgt = np.zeros((E, 2), np.float64)
        self.init_d = np.zeros((E,), np.float64)
        self.prev_d = np.zeros((E,), np.float64)
        self.steps = np.zeros((E,), np.uint32)
        self.episode = np.zeros((E,), np.uint32)
        self.vel_f32 = np.zeros((E,), np.uint8)
        self._st = _UWState(E, _p(self.loc), _p(self.vel), _p(self.tgt), _p(self.init_d), _p(self.prev_d),
                            _p(self.steps), _p(self.episode), _p(self.vel_f32))
        self.pending = np.zeros((E,), np.uint8)
        self.reached = np.zeros((E,), np.uint8)
        self.ep_return = np.zeros((E,), np.float32)
        self.fin_counts = np.zeros((E, 4), np.uint32)
        self.fin_return = np.zeros((E,), np.float32)
        self._ep = _UWEpisodeState(_p(self.pending), _p(self.reached), _p(self.ep_return), _p(self.fin_counts),
                                   _p(self.fin_return))

    def get_state(self):
        return dict(loc=self.loc.astype(np.float32), vel=self.vel.copy(), tgt=self.tgt.astype(np.float32),
                    init_d=self.init_d.astype(np.float32), prev_d=self.prev_d.astype(np.float32),
                    steps=self.steps.copy(), episode=self.episode.copy(), vel_f32=self.vel_f32.copy())

    def set_state(self, **kw):
        for name, val in kw.items():
            getattr(self, name)[...] = np.asarray(val).reshape(getattr(self, name).shape)

    def reset_mt(self, stream, env=0):
        lib().uavo_uw_reset_mt(ctypes.byref(self.cfg), ctypes.byref(self._st), env, ctypes.byref(stream._g))

    def step_ex(self, actions, polar=False, auto_reset=False, step_cap=0, track_returns=True, seed=0, env_offset=0,
                action_is_f32=None):
        arr = np.asarray(actions)
        if action_is_f32 is None:
            action_is_f32 = arr.dtype == np.float32
        a = np.ascontiguousarray(arr.astype(np.float64).reshape(self.E, 2))
        obs = np.empty((self.E, UW_OBS_DIM), np.float64)
        rew = np.empty((self.E,), np.float64)
        done = np.empty((self.E,), np.uint8)
        info = np.empty((self.E,), np.float64)
        rmask = np.zeros((self.E,), np.uint8)
        lib().uavo_uw_step_ex(ctypes.byref(self.cfg), ctypes.byref(self._st), ctypes.byref(self._ep), int(bool(polar)),
                              int(bool(auto_reset)), int(step_cap), int(bool(track_returns)), int(seed), int(env_offset),
                              _p(a), int(bool(action_is_f32)), _p(obs), _p(rew), _p(done), _p(info), _p(rmask),
                              self.nthreads)
        return obs, rew, done, info, rmask

    def reset_philox(self, seed, mask=None, env_offset=0):
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        for e in range(self.E):
            if m is None or m[e]:
                lib().uavo_uw_fold_episode(ctypes.byref(self._st), ctypes.byref(self._ep), e)
        lib().uavo_uw_reset_philox(ctypes.byref(self.cfg), ctypes.byref(self._st),
                                   None if m is None else _p(m), int(seed), int(env_offset), self.nthreads)

    def observe(self):
        obs = np.empty((self.E, UW_OBS_DIM), np.float64)
        lib().uavo_uw_observe(ctypes.byref(self.cfg), ctypes.byref(self._st), _p(obs), self.nthreads)
        return obs

    def step(self, actions, action_is_f32=None):
        arr = np.asarray(actions)
        if action_is_f32 is None:
            action_is_f32 = arr.dtype == np.float32
        a = np.ascontiguousarray(arr.astype(np.float64).reshape(self.E, 2))
        obs = np.empty((self.E, UW_OBS_DIM), np.float64)
        rew = np.empty((self.E,), np.float64)
        done = np.empty((self.E,), np.uint8)
        info = np.empty((self.E,), np.float64)
        lib().uavo_uw_step(ctypes.byref(self.cfg), ctypes.byref(self._st), _p(a), int(bool(action_is_f32)),
                           _p(obs), _p(rew), _p(done), _p(info), self.nthreads)
        self.reached[:] = info < 0.5
        return obs, rew, done, info
