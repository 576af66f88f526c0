"""Batched (vector) environments: E independent worlds advanced by one HIP kernel launch per step.

This is the new surface the MI355X build adds on top of the reference's single-env gym API
(SURVEY.md §8b): tensors in, tensors out, all resident in HBM, no host synchronisation on the step
path.  Buffers are PyTorch-ROCm tensors (zero-copy: the kernels read/write `tensor.data_ptr()`);
any object exposing `__hip_array_interface__` / `__cuda_array_interface__` is accepted as input.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .spaces import Box

_TORCH_DT = {torch.float32: _lib.F32, torch.float64: _lib.F64}


class HipArray:
    """Zero-copy view of a device tensor that publishes `__hip_array_interface__` (same schema as
    the CUDA array interface v3) next to torch's own `__cuda_array_interface__`/DLPack."""

    def __init__(self, tensor):
        self.tensor = tensor

    @property
    def __hip_array_interface__(self):
        t = self.tensor
        typestr = {torch.float32: "<f4", torch.float64: "<f8", torch.uint8: "|u1", torch.bool: "|b1",
                   torch.uint32: "<u4", torch.int32: "<i4", torch.int64: "<i8"}[t.dtype]
        return dict(shape=tuple(t.shape), typestr=typestr, data=(t.data_ptr(), False), version=3,
                    strides=None if t.is_contiguous() else tuple(s * t.element_size() for s in t.stride()),
                    stream=None)

    __cuda_array_interface__ = __hip_array_interface__


def _device_index(device):
    if not torch.cuda.is_available():
        raise RuntimeError("uavx: no MI355X/HIP device visible to PyTorch; the batched env has no CPU fallback")
    if device is None:
        return torch.cuda.current_device()
    d = torch.device(device)
    if d.type != "cuda":
        raise ValueError(f"uavx: device must be a HIP ('cuda') device, got {device!r}")
    return d.index if d.index is not None else torch.cuda.current_device()


class _Base:
    """Shared plumbing: device, stream, tensor argument checking."""

    def _init_device(self, device):
        self._L = _lib.load()
        self._dev_index = _device_index(device)
        self.device = torch.device("cuda", self._dev_index)
        self._h = ctypes.c_void_p()

    def _stream(self):
        """Raw hipStream_t of torch's current stream on the env's device, as an int (ctypes takes it as void*)."""
        try:
            return torch._C._cuda_getCurrentRawStream(self._dev_index)   # no Stream object: ~0.2 us
        except AttributeError:
            return torch.cuda.current_stream(self.device).cuda_stream

    def _actions_arg(self, actions, shape):
        """-> (tensor kept alive, dtype code).  float32/float64, contiguous, on self.device."""
        if isinstance(actions, torch.Tensor):
            t = actions
        elif hasattr(actions, "__hip_array_interface__") and not hasattr(actions, "__array__"):
            t = torch.as_tensor(HipArrayAdapter(actions), device=self.device)
        elif hasattr(actions, "__cuda_array_interface__"):
            t = torch.as_tensor(actions, device=self.device)
        else:
            a = np.asarray(actions)
            if a.dtype not in (np.float32, np.float64):
                a = a.astype(np.float64)
            t = torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
        if t.dtype not in _TORCH_DT:
            raise ValueError(f"uavx: actions must be float32 or float64, got {t.dtype}")
        if t.device != self.device:
            raise ValueError(f"uavx: actions live on {t.device}, env is on {self.device}")
        if tuple(t.shape) != tuple(shape):
            if t.numel() != int(np.prod(shape)):
                raise ValueError(f"uavx: actions have shape {tuple(t.shape)}, expected {tuple(shape)}")
            t = t.reshape(shape)
        if not t.is_contiguous():
            t = t.contiguous()
        return t, _TORCH_DT[t.dtype]

    def _out(self, t, shape, dtype, name):
        if t.device != self.device or t.dtype != dtype or tuple(t.shape) != tuple(shape) or not t.is_contiguous():
            raise ValueError(f"uavx: `{name}` must be a contiguous {dtype} tensor of shape {tuple(shape)} on {self.device}")
        return t


class HipArrayAdapter:
    """Lets torch.as_tensor consume an object that only publishes `__hip_array_interface__`."""

    def __init__(self, obj):
        self._obj = obj
        self.__cuda_array_interface__ = obj.__hip_array_interface__


class BatchedMultiUAVWorld2D(_Base):
    """E x MultiUAVWorld2D (MUW:10).  Constructor keywords match MUW:13.

    step()/reset() return views of internal HBM buffers; observations, rewards and dones are double-buffered, so
    what step t returned stays valid while step t+1 is produced (replay tuples (s, a, r, s'), test_sac_multi.py:101-105;
    n-step / GAE collectors that hold r_t across the next step) and is overwritten by step t+2: clone what must live
    longer, or hand in your own buffers (`out=`, DeviceReplay)."""

    def __init__(self, num_envs, x_size=50.0, y_size=50.0, max_speed=10.0, max_acceleration=5.0, num_agents=4,
                 collider_radius=1.0, d_sense=15, device=None, env_offset=0, seed=0, num_bodies=0, body_speed=5.0,
                 body_period=128, body_seed=0):
        """num_agents = learning UAVs L (every tensor is [E, L, ...]).  num_bodies = scripted dynamic obstacles per env
        (BASELINE configs[4]; an extension with no reference counterpart, see include/uavx.h uavx_set_body_rule): they are
        further entries of the learners' neighbour model, stepped inside the kernel, and cost no action / obs / reward
        traffic."""
        self._init_device(device)
        if not (1 <= int(num_agents) and int(num_agents) + int(num_bodies) <= _lib.MAX_AGENTS and int(num_bodies) >= 0):
            raise ValueError(f"uavx: need num_agents >= 1, num_bodies >= 0 and num_agents + num_bodies <= {_lib.MAX_AGENTS}")
        self.num_envs, self.num_agents = int(num_envs), int(num_agents)
        self.num_bodies = int(num_bodies)
        self.tau = 0.02                                                                  # MUW:26
        self.env_offset, self.seed = int(env_offset), int(seed)
        cfg = self._adopt_world(x_size, y_size, max_speed, max_acceleration, collider_radius, d_sense)
        _lib.check(self._L.uavx_create(ctypes.byref(cfg), self.num_envs, self.env_offset, self._dev_index,
                                       ctypes.byref(self._h)))
        if self.num_bodies:
            self.set_body_rule(speed=body_speed, period=body_period, seed=body_seed)
        self.levels = None
        E, N = self.num_envs, self.num_agents
        with torch.cuda.device(self.device):
            self._obs = [torch.zeros((E, N, _lib.OBS_DIM), dtype=torch.float32, device=self.device) for _ in range(2)]
            self._rews = [torch.zeros((E, N), dtype=torch.float32, device=self.device) for _ in range(2)]
            self._dones = [torch.zeros((E, N), dtype=torch.uint8, device=self.device) for _ in range(2)]
        self._rew, self._done = self._rews[0], self._dones[0]     # (shape / dtype templates for the out= checks)
        self._flip = 0
        self._resets = 0
        # hot-path caches: eager stepping is host-bound (a launch is ~6.6 us of GPU time at 65 536 x 4), so the
        # per-call Python work is kept to a pointer fetch, one shape/dtype test and the ctypes call
        self._act_shape = torch.Size((E, N, 2))
        self._obs_ptr = [o.data_ptr() for o in self._obs]
        self._rew_ptr, self._done_ptr = [r.data_ptr() for r in self._rews], [d.data_ptr() for d in self._dones]
        self._done_bools = [d.view(torch.bool) for d in self._dones]
        self._step_fn = self._L.uavx_step
        self._info = {"distance": 0}

    def _adopt_world(self, x_size, y_size, max_speed, max_acceleration, collider_radius, d_sense):
        """Mirrors the scalar world parameters as the attributes MUW:13-47 exposes; returns the uavx_config."""
        self.x_size, self.y_size = float(x_size), float(y_size)
        self.max_speed = np.array([max_speed, max_speed], dtype=np.float64)            # MUW:21
        self.min_speed = -self.max_speed
        self.max_acceleratoin = np.array([max_acceleration, max_acceleration], dtype=np.float64)  # sic, MUW:23
        self.min_acceleratoin = -self.max_acceleratoin
        self.map_diagonal_size = float(np.hypot(x_size, y_size))                         # MUW:17
        self.min_location = np.array([-x_size / 2.0, -y_size / 2.0])
        self.max_location = np.array([x_size / 2.0, y_size / 2.0])
        self.collider_radius, self.d_sense = collider_radius, d_sense
        # MUW:44-47 (low[9] = 1 is the reference's own typo; kept for space equality)
        self.observation_space = Box(np.array([0, -1, 0, -1, 0, -1, -1, 0, -1, 1]), np.ones(10), shape=(10,),
                                     dtype=np.float32)
        self.action_space = Box(-max_speed, max_speed, shape=(2,), dtype=np.float32)
        return _lib.Config(x_size, y_size, max_speed, max_acceleration, collider_radius, float(d_sense), self.tau,
                           self.num_agents, self.num_bodies)

    def set_config(self, **world):
        """Curriculum hook: change any of x_size, y_size, max_speed, max_acceleration, collider_radius, d_sense for
        all LATER launches (the reference re-creates the env object for this, test_sac_multi_score.py:37).  Agent
        state is kept; a hipGraph captured before the call keeps the old parameters."""
        cur = dict(x_size=self.x_size, y_size=self.y_size, max_speed=float(self.max_speed[0]),
                   max_acceleration=float(self.max_acceleratoin[0]), collider_radius=self.collider_radius,
                   d_sense=self.d_sense)
        unknown = set(world) - set(cur)
        if unknown:
            raise TypeError(f"uavx: unknown world parameter(s) {sorted(unknown)}")
        cur.update(world)
        cfg = _lib.Config(cur["x_size"], cur["y_size"], cur["max_speed"], cur["max_acceleration"],
                          cur["collider_radius"], float(cur["d_sense"]), self.tau, self.num_agents, self.num_bodies)
        _lib.check(self._L.uavx_set_config(self._h, ctypes.byref(cfg)), self._h)
        self._adopt_world(**cur)

    # -- configs[4] extension: scripted bodies + randomized-reset curriculum (include/uavx.h) ------------------------
    def set_body_rule(self, speed=5.0, period=128, seed=0):
        """Cruise speed (m/s), waypoint period (env steps, a power of two) and Philox seed of the scripted bodies."""
        rule = _lib.BodyRule(float(speed), int(period), 0, int(seed))
        _lib.check(self._L.uavx_set_body_rule(self._h, ctypes.byref(rule)), self._h)
        self.body_rule = dict(speed=float(speed), period=int(period), seed=int(seed))

    def set_curriculum(self, levels, lo=-1, hi=-1):
        """levels: list of dicts(x_size, y_size, collider_radius, d_sense[, n_active, b_active]) -- per-env worlds.  An
        env takes its level when it is (re-)initialised (reset() or the auto-reset of step_ex): drawn uniformly in
        [lo, hi] (the randomized-reset curriculum; move the window as training progresses), or, with lo < 0, the level
        assigned by set_env_levels.  levels=None / [] removes the table."""
        levels = list(levels or [])
        arr = (_lib.Level * max(1, len(levels)))(*[
            _lib.Level(float(l["x_size"]), float(l["y_size"]), float(l["collider_radius"]), float(l["d_sense"]),
                       int(l.get("n_active", self.num_agents)), int(l.get("b_active", self.num_bodies))) for l in levels])
        _lib.check(self._L.uavx_set_curriculum(self._h, arr, len(levels), int(lo), int(hi), self._stream()), self._h)
        self.levels = levels or None

    def set_level_window(self, lo, hi):
        self.set_curriculum(self.levels, lo, hi)

    def set_env_levels(self, levels):
        """[E] uint8: the level each env takes at its NEXT reset (used while the random window is off, lo < 0)."""
        t = torch.as_tensor(levels, device=self.device).to(torch.uint8).contiguous()
        self._out(t, (self.num_envs,), torch.uint8, "levels")
        _lib.check(self._L.uavx_set_env_levels(self._h, t.data_ptr(), self._stream()), self._h)
        torch.cuda.current_stream(self.device).synchronize()

    def env_levels(self):
        """[E] uint8: level in force in each env."""
        t = torch.empty((self.num_envs,), dtype=torch.uint8, device=self.device)
        _lib.check(self._L.uavx_get_env_levels(self._h, t.data_ptr(), self._stream()), self._h)
        return t

    def set_prefetch(self, every):
        """Reset layouts drawn ahead of time (uavx_set_prefetch): every auto-resetting step_ex launch carries one staging
        workgroup per `every` env-workgroups; each looks at a window of envs and draws the missing layouts of their next
        episodes.  0 / False = off (every auto-reset draws in place, inside its step workgroup).  The handle starts with a
        cadence sized for num_envs / 128 layouts per launch (32 at 4 UAVs, 64 at 8, 16 with 8 learners + 16 bodies): lower it
        when episodes are much shorter than 128 steps.
        Results do not depend on it."""
        _lib.check(self._L.uavx_set_prefetch(self._h, int(every)), self._h)

    def get_bodies(self):
        """[E, B, 6] float32 {x, y, dx, dy, heading, legs}: position, displacement per env step, direction of travel, steps of
        the current leg that move (include/uavx.h, uavx_set_body_rule); a body switched off by its env's level sits at +inf."""
        t = torch.empty((self.num_envs, self.num_bodies, _lib.BODY_DIM), dtype=torch.float32, device=self.device)
        _lib.check(self._L.uavx_get_bodies(self._h, t.data_ptr(), self._stream()), self._h)
        return t

    def set_bodies(self, records):
        t = torch.as_tensor(records, device=self.device).to(torch.float32).reshape(self.num_envs, self.num_bodies, _lib.BODY_DIM).contiguous()
        _lib.check(self._L.uavx_set_bodies(self._h, t.data_ptr(), self._stream()), self._h)
        torch.cuda.current_stream(self.device).synchronize()

    # -- lifecycle ---------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.uavx_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _next_obs_buf(self):
        self._flip ^= 1
        return self._obs[self._flip]

    # -- MUW:116-175 -------------------------------------------------------------------------------
    def reset(self, mask=None, seed=None, out=None):
        """Resets the envs selected by `mask` (bool/uint8 [E] device tensor; None = all) with the
        on-device Philox sampler and returns the observations of all envs, [E, N, 10] float32."""
        if seed is not None:
            self.seed = int(seed)
        m = None
        if mask is not None:
            m = torch.as_tensor(mask, device=self.device)
            if m.dtype == torch.bool:
                m = m.to(torch.uint8)
            m = self._out(m.contiguous(), (self.num_envs,), torch.uint8, "mask")
        obs = self._next_obs_buf() if out is None else self._out(out, self._obs[0].shape, torch.float32, "out")
        _lib.check(self._L.uavx_reset(self._h, None if m is None else m.data_ptr(), self.seed, obs.data_ptr(),
                                      self._stream()), self._h)
        self._resets += 1
        return obs

    def reset_circular(self, radius=20.0, target_radius=23.0, float64=True):
        """reset(circular=True) of the reference (MUW:157-163) for every env: UAV i starts at angle 2*pi*i/N on a
        circle of `radius` and must reach the antipodal point on `target_radius`.  The reference installs float64
        arrays here, so the episode runs in float64-position mode (set_position_mode); float64=False keeps the
        float32 fast path on float32-rounded points instead (positions then drift ~1e-4 m from the reference's
        over a few hundred steps)."""
        import math
        E, N = self.num_envs, self.num_agents
        loc = np.empty((N, 2)); tgt = np.empty((N, 2)); init_d = np.empty(N)
        for i in range(N):                                   # python floats / libm exactly as MUW:158-162
            theta = 2 * i * math.pi / N
            loc[i] = 20 * np.ones(2) * np.array([math.cos(theta), math.sin(theta)]) * (radius / 20.0)
            tgt[i] = 23 * np.ones(2) * np.array([math.cos(theta + math.pi), math.sin(theta + math.pi)]) * (target_radius / 23.0)
            init_d[i] = np.linalg.norm(tgt[i] - loc[i])      # MUW:162
        cnt = self.metrics().cpu().numpy().astype(np.int64)
        cnt[:, :3] = 0
        cnt[:, 3] += 1
        if float64:
            self.set_state(vel=np.zeros((E, N, 2)), flags=np.zeros((E, N), np.uint8), counters=cnt)
            self.set_state_f64(loc=np.broadcast_to(loc, (E, N, 2)), tgt=np.broadcast_to(tgt, (E, N, 2)),
                               init_d=np.broadcast_to(init_d, (E, N)), prev_d=np.broadcast_to(init_d, (E, N)))
            return self.observe()
        self.set_position_mode("float32")
        loc32, tgt32 = loc.astype(np.float32), tgt.astype(np.float32)
        d = tgt32 - loc32
        init32 = np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32)
        self.set_state(loc=np.broadcast_to(loc32, (E, N, 2)), tgt=np.broadcast_to(tgt32, (E, N, 2)),
                       vel=np.zeros((E, N, 2)), init_d=np.broadcast_to(init32, (E, N)),
                       prev_d=np.broadcast_to(init32, (E, N)), flags=np.zeros((E, N), np.uint8), counters=cnt)
        return self.observe()

    # -- MUW:177-241 -------------------------------------------------------------------------------
    def step(self, actions, evaluate=False, out=None):
        """actions: [E, N, 2] float32/float64 velocity commands.  Returns (obs [E,N,10] f32,
        rewards [E,N] f32, dones [E,N] bool, info) with info == {"distance": 0} (MUW:111-114)."""
        if (out is None and type(actions) is torch.Tensor and actions.dtype is torch.float32
                and actions.shape == self._act_shape and actions.is_contiguous() and actions.device == self.device):
            self._flip ^= 1  # fast path: device float32 tensor of the right shape, internal output buffers
            f = self._flip
            rc = self._step_fn(self._h, actions.data_ptr(), _lib.F32, 1 if evaluate else 0, self._obs_ptr[f],
                               self._rew_ptr[f], self._done_ptr[f], self._stream())
            if rc:
                _lib.check(rc, self._h)
            return self._obs[f], self._rews[f], self._done_bools[f], self._info
        a, code = self._actions_arg(actions, (self.num_envs, self.num_agents, 2))
        if out is None:
            obs = self._next_obs_buf()
            rew, done = self._rews[self._flip], self._dones[self._flip]
        else:
            obs = self._out(out[0], self._obs[0].shape, torch.float32, "out[0]")
            rew = self._out(out[1], self._rew.shape, torch.float32, "out[1]")
            done = out[2]
            if done.dtype == torch.bool:
                done = done.view(torch.uint8)
            done = self._out(done, self._done.shape, torch.uint8, "out[2]")
        _lib.check(self._L.uavx_step(self._h, a.data_ptr(), code, int(bool(evaluate)), obs.data_ptr(),
                                     rew.data_ptr(), done.data_ptr(), self._stream()), self._h)
        return obs, rew, done.view(torch.bool), {"distance": 0}

    _POLICIES = {None: _lib.RESET_NEVER, "never": _lib.RESET_NEVER, "agent0_done": _lib.RESET_AGENT0_DONE,
                 "all_done": _lib.RESET_ALL_DONE}

    def step_ex(self, actions, evaluate=False, polar=False, auto_reset=None, step_cap=0, track_returns=True,
                out=None, flags_out=None, packed_flags=False):
        """env.step plus what the reference's trainer loops do around it, in the same launch:
          polar=True       actions are policy outputs a in [-1,1]^2, converted like test_sac_multi.py:77-80
          auto_reset       "agent0_done" (test_sac_multi.py:112) / "all_done" (:116,161) / None; step_cap (:17,67)
          track_returns    accumulate episode return / evaluation score per env (:106,157)
        Auto-reset is next-step: an ended env keeps its terminal observation in this call's outputs and
        is re-initialised by the NEXT call instead of being stepped (that call's reset_mask[e] is True,
        reward 0, done False).  Returns (obs, rew, done, info) with [E] bool tensors info["reset_mask"],
        info["ended"] (this call ended the env's episode) and info["truncated"] (it ended by the step cap alone: a
        time-limit cut to bootstrap through, not a terminal state).
        out=(obs, rew, done) / flags_out=(reset_mask, ended, truncated): caller-owned tensors the launch writes instead
        of the env's own buffers (DeviceReplay hands in slots of its ring: nothing is copied afterwards).
        packed_flags=True (UAVX_FLAGS_IN_DONE): no per-env flag arrays are written; `done` comes back as the raw uint8
        tensor whose entry [e, 0] carries done | reset_mask << 1 | ended << 2 | truncated << 3 (the other agents' entries
        are 0 / 1) -- one launch with no one-byte-per-env stores; unpack_done() splits it (four small torch ops, only when
        the caller wants the flags)."""
        fast = (out is None and type(actions) is torch.Tensor and actions.shape == self._act_shape
                and actions.is_contiguous() and actions.device == self.device and actions.dtype in _TORCH_DT)
        if fast:
            a, code = actions, _TORCH_DT[actions.dtype]
        else:
            a, code = self._actions_arg(actions, (self.num_envs, self.num_agents, 2))
        if out is None:
            self._flip ^= 1
            f = self._flip
            obs, rew, done, done_bool = self._obs[f], self._rews[f], self._dones[f], self._done_bools[f]
            obs_ptr, rew_ptr, done_ptr = self._obs_ptr[f], self._rew_ptr[f], self._done_ptr[f]
        else:  # caller-owned output buffers (DeviceReplay): same shape / dtype / device / contiguity checks as step()
            obs = self._out(out[0], self._obs[0].shape, torch.float32, "out[0]")
            rew = self._out(out[1], self._rew.shape, torch.float32, "out[1]")
            done = out[2].view(torch.uint8) if out[2].dtype == torch.bool else out[2]
            done = self._out(done, self._done.shape, torch.uint8, "out[2]")
            done_bool = done.view(torch.bool)
            obs_ptr, rew_ptr, done_ptr = obs.data_ptr(), rew.data_ptr(), done.data_ptr()
            if obs_ptr % 16:
                raise ValueError("uavx: `out[0]` must be 16-byte aligned")
        if not hasattr(self, "_reset_mask"):
            self._reset_mask = torch.zeros((3, self.num_envs), dtype=torch.uint8, device=self.device)
            self._reset_mask_bool = self._reset_mask.view(torch.bool)
            self._ex_args = _lib.StepArgs()
            self._flag_ptrs = tuple(self._reset_mask[i].data_ptr() for i in range(3))
            self._ex_ref = ctypes.byref(self._ex_args)
            self._ex_info = {"distance": 0, "reset_mask": self._reset_mask_bool[0], "ended": self._reset_mask_bool[1],
                             "truncated": self._reset_mask_bool[2]}
            self._info_packed = {"distance": 0, "flags_in_done": True}
        args = self._ex_args  # one struct reused across calls: only the fields that change are written
        info = self._ex_info
        args.flags_mode = _lib.FLAGS_IN_DONE if packed_flags else _lib.FLAGS_ARRAYS
        if packed_flags:
            if flags_out is not None:
                raise ValueError("uavx: packed_flags=True writes no flag arrays; drop flags_out")
            args.reset_mask = args.ended = args.truncated = None
            info = self._info_packed
        elif flags_out is not None:
            fl = [self._out(t.view(torch.uint8) if t.dtype == torch.bool else t, (self.num_envs,), torch.uint8, f"flags_out[{i}]")
                  for i, t in enumerate(flags_out)]
            args.reset_mask, args.ended, args.truncated = (t.data_ptr() for t in fl)
            info = {"distance": 0, "reset_mask": fl[0].view(torch.bool), "ended": fl[1].view(torch.bool),
                    "truncated": fl[2].view(torch.bool)}
        else:
            args.reset_mask, args.ended, args.truncated = self._flag_ptrs
        args.actions, args.action_dtype = a.data_ptr(), code
        args.action_mode = _lib.ACTION_POLAR if polar else _lib.ACTION_CARTESIAN
        args.evaluate = 1 if evaluate else 0
        args.reset_policy = self._POLICIES[auto_reset]
        args.step_cap, args.track_returns, args.seed = int(step_cap), 1 if track_returns else 0, self.seed
        args.obs, args.rew, args.done = obs_ptr, rew_ptr, done_ptr
        rc = self._L.uavx_step_ex(self._h, self._ex_ref, self._stream())
        if rc:
            _lib.check(rc, self._h)
        return obs, rew, (done if packed_flags else done_bool), info

    @staticmethod
    def unpack_done(done):
        """Splits the uint8 `done` tensor of step_ex(packed_flags=True): -> (done [E, N] bool, reset_mask [E] bool,
        ended [E] bool, truncated [E] bool)."""
        first = done[:, 0]
        return (done & 1).to(torch.bool), (first & 2).to(torch.bool), (first & 4).to(torch.bool), (first & 8).to(torch.bool)

    def episode_stats(self):
        """Statistics over the episodes ended so far (auto-reset or reset()): dict of [E] tensors
        episodes, steps, reach, coll (int32) and return0, score (float32); see uavx_get_episode_stats."""
        c = torch.empty((self.num_envs, 4), dtype=torch.int32, device=self.device)
        r = torch.empty((self.num_envs, 2), dtype=torch.float32, device=self.device)
        _lib.check(self._L.uavx_get_episode_stats(self._h, c.data_ptr(), r.data_ptr(), self._stream()), self._h)
        return dict(episodes=c[:, 0], steps=c[:, 1], reach=c[:, 2], coll=c[:, 3], return0=r[:, 0], score=r[:, 1])

    def clear_episode_stats(self):
        _lib.check(self._L.uavx_clear_episode_stats(self._h, self._stream()), self._h)

    def evaluation_summary(self):
        """SR, CR and average score exactly as test_sac_multi.py:174-176 computes them, over all ended episodes."""
        st = self.episode_stats()
        episodes = int(st["episodes"].sum().item())
        denom = max(1, self.num_agents * episodes)
        return dict(episodes=episodes, success_rate=float(st["reach"].sum().item()) / denom,
                    collision_rate=float(st["coll"].sum().item()) / denom,
                    avg_score=float(st["score"].double().sum().item()) / denom,
                    mean_steps=float(st["steps"].sum().item()) / max(1, episodes))

    def step_k(self, action_tape, evaluate=False, tape_out=False):
        """K steps in one launch from an action tape [K, E, N, 2] (open-loop rollouts, benchmarks).
        tape_out=False returns the last step's (obs, rew, done); True returns [K, ...] tapes."""
        K = int(action_tape.shape[0])
        a, code = self._actions_arg(action_tape, (K, self.num_envs, self.num_agents, 2))
        E, N = self.num_envs, self.num_agents
        if tape_out:
            obs = torch.empty((K, E, N, _lib.OBS_DIM), dtype=torch.float32, device=self.device)
            rew = torch.empty((K, E, N), dtype=torch.float32, device=self.device)
            done = torch.empty((K, E, N), dtype=torch.uint8, device=self.device)
        else:
            obs = self._next_obs_buf()
            rew, done = self._rews[self._flip], self._dones[self._flip]
        _lib.check(self._L.uavx_step_k(self._h, K, a.data_ptr(), code, int(bool(evaluate)), int(bool(tape_out)),
                                       obs.data_ptr(), rew.data_ptr(), done.data_ptr(), self._stream()), self._h)
        return obs, rew, done.view(torch.bool), {"distance": 0}

    def observe(self, out=None):
        obs = self._next_obs_buf() if out is None else self._out(out, self._obs[0].shape, torch.float32, "out")
        _lib.check(self._L.uavx_observe(self._h, obs.data_ptr(), self._stream()), self._h)
        return obs

    # -- state exchange (AG:13-20 fields, MUW:166-168 counters) --------------------------------------
    _STATE_SPEC = (("loc", torch.float32, 2), ("vel", torch.float64, 2), ("tgt", torch.float32, 2),
                   ("init_d", torch.float32, 0), ("prev_d", torch.float32, 0), ("flags", torch.uint8, 0))

    def get_state(self):
        E, N = self.num_envs, self.num_agents
        st = {}
        for name, dt, last in self._STATE_SPEC:
            st[name] = torch.empty((E, N, last) if last else (E, N), dtype=dt, device=self.device)
        st["counters"] = torch.empty((E, 4), dtype=torch.int32, device=self.device)
        view = _lib.StateView(*[st[n].data_ptr() for n in ("loc", "vel", "tgt", "init_d", "prev_d", "flags", "counters")])
        _lib.check(self._L.uavx_get_state(self._h, ctypes.byref(view), self._stream()), self._h)
        return st

    def set_state(self, **fields):
        """Overwrites any subset of loc, vel, tgt, init_d, prev_d, flags, counters (host or device arrays)."""
        E, N = self.num_envs, self.num_agents
        keep, ptrs = [], {}
        spec = {n: (dt, (E, N, last) if last else (E, N)) for n, dt, last in self._STATE_SPEC}
        spec["counters"] = (torch.int32, (E, 4))
        for name, val in fields.items():
            if name not in spec:
                raise ValueError(f"uavx: unknown state field {name!r}")
            dt, shape = spec[name]
            if isinstance(val, torch.Tensor):
                t = val.to(device=self.device, dtype=dt)
            else:
                arr = np.asarray(val)
                if name == "counters":
                    arr = arr.astype(np.int64).astype(np.int32)
                t = torch.from_numpy(np.array(arr, order="C")).to(device=self.device, dtype=dt)
            t = t.reshape(shape).contiguous()
            keep.append(t)
            ptrs[name] = t.data_ptr()
        view = _lib.StateView(*[ptrs.get(n) for n in ("loc", "vel", "tgt", "init_d", "prev_d", "flags", "counters")])
        _lib.check(self._L.uavx_set_state(self._h, ctypes.byref(view), self._stream()), self._h)
        torch.cuda.current_stream(self.device).synchronize()  # `keep` may be freed after return

    # -- float64-position episodes (reset(circular=True) / caller-assigned float64 arrays, MUW:157-163) ------------
    @property
    def position_mode(self):
        """"float32" (after reset(), MUW:126) or "float64" (see set_position_mode)."""
        return "float64" if self._L.uavx_get_position_mode(self._h) == _lib.POS_F64 else "float32"

    def set_position_mode(self, mode):
        """"float64": positions / targets / init and prev distances are held and stepped as doubles, like the
        reference does once float64 arrays were assigned to its agents; "float32" rounds them back.  All envs of the
        batch share the mode; reset() returns to float32."""
        code = {"float32": _lib.POS_F32, "float64": _lib.POS_F64}[mode]
        _lib.check(self._L.uavx_set_position_mode(self._h, code, self._stream()), self._h)

    _STATE64 = (("loc", 2), ("tgt", 2), ("init_d", 0), ("prev_d", 0))

    def set_state_f64(self, **fields):
        """Overwrites any subset of loc, tgt, init_d, prev_d with float64 values (enters float64-position mode)."""
        E, N = self.num_envs, self.num_agents
        keep, ptrs = [], {}
        spec = {n: ((E, N, last) if last else (E, N)) for n, last in self._STATE64}
        for name, val in fields.items():
            if name not in spec:
                raise ValueError(f"uavx: unknown float64 state field {name!r}")
            t = val if isinstance(val, torch.Tensor) else torch.from_numpy(np.array(val, dtype=np.float64, order="C"))
            t = t.to(device=self.device, dtype=torch.float64).reshape(spec[name]).contiguous()
            keep.append(t)
            ptrs[name] = t.data_ptr()
        view = _lib.StateViewF64(*[ptrs.get(n) for n, _ in self._STATE64])
        _lib.check(self._L.uavx_set_state_f64(self._h, ctypes.byref(view), self._stream()), self._h)
        torch.cuda.current_stream(self.device).synchronize()  # `keep` may be freed after return

    def get_state_f64(self):
        E, N = self.num_envs, self.num_agents
        st = {n: torch.empty((E, N, last) if last else (E, N), dtype=torch.float64, device=self.device)
              for n, last in self._STATE64}
        view = _lib.StateViewF64(*[st[n].data_ptr() for n, _ in self._STATE64])
        _lib.check(self._L.uavx_get_state_f64(self._h, ctypes.byref(view), self._stream()), self._h)
        return st

    # -- exact snapshot / restore (uavx_save / uavx_load) -------------------------------------------------------------
    def state_dict(self):
        """Everything needed to continue this batch bit for bit: {"snapshot": uint8 device tensor written by uavx_save (agent
        state, counters, running and ended-episode statistics, bodies, levels, parked auto-reset layouts, world / body rule /
        curriculum / staging parameters), "host": the few Python-side values (seed, constructor shape)}.  The reference
        checkpoints only its agents (sac.py:101-139); with this a 65 536-env run resumes where it stopped.  torch.save-able."""
        n = int(self._L.uavx_snapshot_bytes(self._h))
        blob = torch.empty((n + 256,), dtype=torch.uint8, device=self.device)
        off = (-blob.data_ptr()) % 256                       # uavx_save wants a 256-byte aligned buffer
        snap = blob[off:off + n]
        _lib.check(self._L.uavx_save(self._h, snap.data_ptr(), self._stream()), self._h)
        host = dict(num_envs=self.num_envs, num_agents=self.num_agents, num_bodies=self.num_bodies, env_offset=self.env_offset,
                    seed=self.seed, levels=self.levels, body_rule=getattr(self, "body_rule", None),
                    world=dict(x_size=self.x_size, y_size=self.y_size, max_speed=float(self.max_speed[0]),
                               max_acceleration=float(self.max_acceleratoin[0]), collider_radius=self.collider_radius,
                               d_sense=self.d_sense))
        return {"snapshot": snap, "host": host}

    def load_state_dict(self, sd):
        """Restores a state_dict() of a batch of the same shape (envs, learners, bodies); waits for the current stream once."""
        host = sd["host"]
        if (host["num_envs"], host["num_agents"], host["num_bodies"]) != (self.num_envs, self.num_agents, self.num_bodies):
            raise ValueError("uavx: the state dict was taken from a batch of another shape")
        snap = sd["snapshot"].to(self.device)
        need = int(self._L.uavx_snapshot_bytes(self._h))
        if snap.dtype != torch.uint8 or snap.dim() != 1 or not snap.is_contiguous() or snap.numel() < need:
            raise ValueError(f"uavx: the snapshot must be a contiguous uint8 tensor of at least {need} bytes for this batch "
                             f"(got {tuple(snap.shape)} {snap.dtype})")
        if snap.data_ptr() % 256:
            blob = torch.empty((snap.numel() + 256,), dtype=torch.uint8, device=self.device)
            off = (-blob.data_ptr()) % 256
            blob[off:off + snap.numel()].copy_(snap)
            snap = blob[off:off + snap.numel()]
        _lib.check(self._L.uavx_load(self._h, snap.data_ptr(), self._stream()), self._h)
        torch.cuda.current_stream(self.device).synchronize()      # `snap` may be a temporary
        self.seed, self.env_offset = int(host["seed"]), int(host["env_offset"])
        self.levels = host["levels"]
        if host.get("body_rule"):
            self.body_rule = dict(host["body_rule"])
        self._adopt_world(**host["world"])

    def metrics(self):
        """[E, 4] int32: steps, target_reach_count, collision_count, episode index (MUW:166-168)."""
        c = torch.empty((self.num_envs, 4), dtype=torch.int32, device=self.device)
        _lib.check(self._L.uavx_get_metrics(self._h, c.data_ptr(), self._stream()), self._h)
        return c

    def nonfinite_count(self):
        """[E] int32: agent-steps of each env's running episode whose reward came out NaN / Inf (a poisoned command or
        state; cleared at reset).  `bool(env.nonfinite_count().any())` is the batched form of the trainers' NaN tripwire
        (test_ddpg_multi.py:114-130)."""
        c = torch.empty((self.num_envs,), dtype=torch.int32, device=self.device)
        _lib.check(self._L.uavx_get_nonfinite(self._h, c.data_ptr(), self._stream()), self._h)
        return c

    @property
    def steps(self):
        return self.metrics()[:, 0]

    @property
    def target_reach_count(self):
        return self.metrics()[:, 1]

    @property
    def collision_count(self):
        return self.metrics()[:, 2]

    def render(self, mode="human"):  # MUW:243 — no display on a compute node
        return None


class BatchedUAVWorld2D(_Base):
    """E x UAVWorld2D (UW:11).  Constructor keywords match UW:14."""

    def __init__(self, num_envs, x_size=100.0, y_size=100.0, agent_num=4, max_speed=12.0, max_acceleration=5.0,
                 device=None, env_offset=0, seed=0):
        self._init_device(device)
        self.num_envs = int(num_envs)
        self.x_size, self.y_size = float(x_size), float(y_size)
        self.max_speed = np.array([max_speed, max_speed], dtype=np.float64)
        self.min_speed = -self.max_speed
        self.max_acceleratoin = np.array([max_acceleration, max_acceleration], dtype=np.float64)
        self.min_acceleratoin = -self.max_acceleratoin
        self.map_diagonal_size = float(np.hypot(x_size, y_size))
        self.min_location = np.array([-x_size / 2.0, -y_size / 2.0])
        self.max_location = np.array([x_size / 2.0, y_size / 2.0])
        self.tau = 0.02
        self.env_offset, self.seed = int(env_offset), int(seed)
        self.observation_space = Box(np.array([0., -1., 0., -1.]), np.ones(4), shape=(4,), dtype=np.float32)  # UW:55
        self.action_space = Box(-max_speed, max_speed, shape=(2,), dtype=np.float32)                            # UW:64
        cfg = _lib.UWConfig(x_size, y_size, max_speed, max_acceleration, self.tau)
        _lib.check(self._L.uavx_uw_create(ctypes.byref(cfg), self.num_envs, self.env_offset, self._dev_index,
                                          ctypes.byref(self._h)))
        E = self.num_envs
        with torch.cuda.device(self.device):
            self._obs = [torch.zeros((E, _lib.UW_OBS_DIM), dtype=torch.float32, device=self.device) for _ in range(2)]
            # double-buffered like the observations: what step t returned is overwritten by step t + 2
            self._rews = [torch.zeros((E,), dtype=torch.float32, device=self.device) for _ in range(2)]
            self._dones = [torch.zeros((E,), dtype=torch.uint8, device=self.device) for _ in range(2)]
            self._infos = [torch.zeros((E,), dtype=torch.float32, device=self.device) for _ in range(2)]
        self._flip = 0
        # what an eager loop would otherwise rebuild on every call (the host cost of a call is what bounds small batches)
        self._act_shape = torch.Size((E, 2))
        self._ptrs = [(self._obs[f].data_ptr(), self._rews[f].data_ptr(), self._dones[f].data_ptr(), self._infos[f].data_ptr())
                      for f in range(2)]
        self._done_bools = [d.view(torch.bool) for d in self._dones]
        self._step_infos = [{"distance": self._infos[f]} for f in range(2)]
        self._step_fn = self._L.uavx_uw_step

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.uavx_uw_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _next_obs_buf(self):
        self._flip ^= 1
        return self._obs[self._flip]

    def reset(self, mask=None, seed=None):  # UW:119-135
        if seed is not None:
            self.seed = int(seed)
        m = None
        if mask is not None:
            m = torch.as_tensor(mask, device=self.device)
            if m.dtype == torch.bool:
                m = m.to(torch.uint8)
            m = self._out(m.contiguous(), (self.num_envs,), torch.uint8, "mask")
        obs = self._next_obs_buf()
        _lib.check(self._L.uavx_uw_reset(self._h, None if m is None else m.data_ptr(), self.seed, obs.data_ptr(),
                                         self._stream()), self._h, uw=True)
        return obs

    def step(self, actions, out=None):  # UW:137-173
        """out = (obs [E,4] f32, rew [E] f32, done [E] u8 / bool, distance [E] f32): caller-owned device buffers the launch
        writes into (the single-env façade packs them into one block so that a step is one launch and ONE copy back)."""
        if (out is None and type(actions) is torch.Tensor and actions.shape == self._act_shape and actions.dtype in _TORCH_DT
                and actions.is_contiguous() and actions.device == self.device):
            self._flip ^= 1  # fast path: a device tensor of the right shape, internal output buffers
            f = self._flip
            o, r, d, i = self._ptrs[f]
            rc = self._step_fn(self._h, actions.data_ptr(), _TORCH_DT[actions.dtype], o, r, d, i, self._stream())
            if rc:
                _lib.check(rc, self._h, uw=True)
            return self._obs[f], self._rews[f], self._done_bools[f], self._step_infos[f]
        a, code = self._actions_arg(actions, (self.num_envs, 2))
        if out is None:
            obs = self._next_obs_buf()
            rew, done, info = self._rews[self._flip], self._dones[self._flip], self._infos[self._flip]
        else:
            obs = self._out(out[0], self._obs[0].shape, torch.float32, "out[0]")
            rew = self._out(out[1], self._rews[0].shape, torch.float32, "out[1]")
            done = out[2].view(torch.uint8) if out[2].dtype == torch.bool else out[2]
            done = self._out(done, self._dones[0].shape, torch.uint8, "out[2]")
            info = self._out(out[3], self._infos[0].shape, torch.float32, "out[3]")
        _lib.check(self._L.uavx_uw_step(self._h, a.data_ptr(), code, obs.data_ptr(), rew.data_ptr(),
                                        done.data_ptr(), info.data_ptr(), self._stream()), self._h, uw=True)
        return obs, rew, done.view(torch.bool), {"distance": info}

    def step_ex(self, actions, polar=False, auto_reset=False, step_cap=0, track_returns=True):
        """env.step plus the single-agent trainer loop's bookkeeping (test_sac.py:77-80,98,106-109): polar=True
        converts policy outputs a in [-1,1]^2 (v = (a0/2+0.5)*action_space.high[0], theta = a1*pi); auto_reset
        re-initialises an env in the call AFTER the one that returned done (info["reset_mask"]); episode returns
        and lengths are accumulated per env (episode_stats())."""
        if (type(actions) is torch.Tensor and actions.shape == self._act_shape and actions.dtype in _TORCH_DT
                and actions.is_contiguous() and actions.device == self.device):
            a, code = actions, _TORCH_DT[actions.dtype]
        else:
            a, code = self._actions_arg(actions, (self.num_envs, 2))
        if not hasattr(self, "_reset_mask"):
            self._reset_mask = torch.zeros((3, self.num_envs), dtype=torch.uint8, device=self.device)
            mb = self._reset_mask.view(torch.bool)
            self._ex_args = [None, None]
            for f in range(2):   # one argument struct per output buffer set, reused: only what changes is written per call
                o, r, d, i = self._ptrs[f]
                self._ex_args[f] = _lib.UWStepArgs(0, 0, 0, 0, 0, 0, 0, 0, o, r, d, i, self._reset_mask[0].data_ptr(),
                                                   self._reset_mask[1].data_ptr(), self._reset_mask[2].data_ptr())
            self._ex_refs = [ctypes.byref(x) for x in self._ex_args]
            self._ex_infos = [{"distance": self._infos[f], "reset_mask": mb[0], "ended": mb[1], "truncated": mb[2]} for f in range(2)]
            self._ex_fn = self._L.uavx_uw_step_ex
        self._flip ^= 1
        f = self._flip
        args = self._ex_args[f]
        args.actions, args.action_dtype = a.data_ptr(), code
        args.action_mode = _lib.ACTION_POLAR if polar else _lib.ACTION_CARTESIAN
        args.auto_reset, args.step_cap, args.track_returns, args.seed = (1 if auto_reset else 0), int(step_cap), (1 if track_returns else 0), self.seed
        rc = self._ex_fn(self._h, self._ex_refs[f], self._stream())
        if rc:
            _lib.check(rc, self._h, uw=True)
        return self._obs[f], self._rews[f], self._done_bools[f], self._ex_infos[f]

    def episode_stats(self):
        c = torch.empty((self.num_envs, 4), dtype=torch.int32, device=self.device)
        r = torch.empty((self.num_envs,), dtype=torch.float32, device=self.device)
        _lib.check(self._L.uavx_uw_get_episode_stats(self._h, c.data_ptr(), r.data_ptr(), self._stream()), self._h, uw=True)
        return dict(episodes=c[:, 0], steps=c[:, 1], reached=c[:, 2], returns=r)

    def clear_episode_stats(self):
        _lib.check(self._L.uavx_uw_clear_episode_stats(self._h, self._stream()), self._h, uw=True)

    def observe(self):
        obs = self._next_obs_buf()
        _lib.check(self._L.uavx_uw_observe(self._h, obs.data_ptr(), self._stream()), self._h, uw=True)
        return obs

    _STATE_SPEC = (("loc", torch.float32, (2,)), ("vel", torch.float64, (2,)), ("tgt", torch.float32, (2,)),
                   ("init_d", torch.float32, ()), ("prev_d", torch.float32, ()), ("flags", torch.uint8, ()),
                   ("counters", torch.int32, (2,)))

    def get_state(self):
        st = {n: torch.empty((self.num_envs,) + tail, dtype=dt, device=self.device) for n, dt, tail in self._STATE_SPEC}
        view = _lib.UWStateView(*[st[n].data_ptr() for n, _, _ in self._STATE_SPEC])
        _lib.check(self._L.uavx_uw_get_state(self._h, ctypes.byref(view), self._stream()), self._h, uw=True)
        return st

    def set_state(self, **fields):
        spec = {n: (dt, (self.num_envs,) + tail) for n, dt, tail in self._STATE_SPEC}
        keep, ptrs = [], {}
        for name, val in fields.items():
            if name not in spec:
                raise ValueError(f"uavx: unknown state field {name!r}")
            dt, shape = spec[name]
            if isinstance(val, torch.Tensor):
                t = val.to(device=self.device, dtype=dt)
            else:
                arr = np.asarray(val)
                if name == "counters":
                    arr = arr.astype(np.int64).astype(np.int32)
                t = torch.from_numpy(np.ascontiguousarray(arr)).to(device=self.device, dtype=dt)
            t = t.reshape(shape).contiguous()
            keep.append(t)
            ptrs[name] = t.data_ptr()
        view = _lib.UWStateView(*[ptrs.get(n) for n, _, _ in self._STATE_SPEC])
        _lib.check(self._L.uavx_uw_set_state(self._h, ctypes.byref(view), self._stream()), self._h, uw=True)
        torch.cuda.current_stream(self.device).synchronize()

    @property
    def steps(self):
        return self.get_state()["counters"][:, 0]

    def render(self, mode="human"):
        return None
