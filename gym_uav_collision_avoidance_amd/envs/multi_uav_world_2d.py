"""Drop-in single-env façade of MultiUAVWorld2D (MUW:10) over the batched HIP path (E = 1).

Same constructor keywords, attributes, list-valued returns and gym-0.24 call signatures as the
reference, so run_multi.py:5-23 and the trainers' loops (test_sac_multi.py:62-119) drive it
unchanged.  All arithmetic of step()/_get_obs() runs in the HIP kernels; the façade only converts
between Python lists and device tensors.  reset() keeps the reference's contract of drawing from
the process-global `np.random` stream in the same order (MUW:126-153), so a seeded script sees the
same start/target layout as with the reference; the layout is then uploaded with uavx_set_state.
"""
import colorsys
import os

import numpy as np
import torch

from .. import _lib
from ..batched import BatchedMultiUAVWorld2D
from ..spaces import Box
from .uav_agent import UAVAgentView

HARD_COLLISION_RADIUS = 0.5  # MUW:8


def _norm32(v):
    v = np.asarray(v, dtype=np.float32)
    return np.sqrt(v[0] * v[0] + v[1] * v[1])


class MultiUAVWorld2D:
    metadata = {"render_fps": 1000}  # MUW:11

    def __init__(self, x_size=50.0, y_size=50.0, max_speed=10.0, max_acceleration=5.0, num_agents=4,
                 collider_radius=1.0, d_sense=15, device=None):
        self._batched = BatchedMultiUAVWorld2D(1, x_size, y_size, max_speed, max_acceleration, num_agents,
                                               collider_radius, d_sense, device=device)
        b = self._batched
        self.x_size, self.y_size, self.num_agents = x_size, y_size, num_agents
        self.map_diagonal_size = b.map_diagonal_size
        self.map_dimension = np.array([x_size, y_size])
        self.min_location, self.max_location = b.min_location, b.max_location
        self.max_speed, self.min_speed = b.max_speed, b.min_speed
        self.max_acceleratoin, self.min_acceleratoin = b.max_acceleratoin, b.min_acceleratoin
        self.tau = b.tau
        self.collider_radius, self.d_sense = collider_radius, d_sense
        self.max_window_size = 800  # MUW:25 / UW:25 (only used by the rgb_array rasteriser here)
        if x_size > y_size:
            self.window_size_x, self.window_size_y = self.max_window_size, self.max_window_size / x_size * y_size
        else:
            self.window_size_y, self.window_size_x = self.max_window_size, self.max_window_size / y_size * x_size
        self.observation_space: Box = b.observation_space
        self.action_space: Box = b.action_space
        self.agent_list = []
        for i in range(num_agents):  # MUW:37-41
            r, g, bl = colorsys.hsv_to_rgb(i / num_agents, 1.0, 1.0)
            self.agent_list.append(UAVAgentView(self, i, (int(255 * r), int(255 * g), int(255 * bl)), max_speed,
                                                max_acceleration, self.tau))
        self.window = None
        self.clock = None
        # One packed device buffer for a step's outputs (obs | rew | done) and one pinned host mirror: a step
        # is one async H2D of the commands, one launch and ONE D2H instead of three .cpu() round trips.
        n = num_agents
        dev = b.device
        self._pack = torch.zeros(n * 45 + 3, dtype=torch.uint8, device=dev)
        self._obs_d = self._pack[: n * 40].view(torch.float32).view(1, n, 10)
        self._rew_d = self._pack[n * 40: n * 44].view(torch.float32).view(1, n)
        self._done_d = self._pack[n * 44: n * 45].view(1, n)
        self._host = torch.zeros(n * 45 + 3, dtype=torch.uint8).pin_memory()
        self._host_np = self._host.numpy()
        self._act_host = torch.zeros((1, n, 2), dtype=torch.float64).pin_memory()
        self._act_np = self._act_host.numpy()
        self._act_d = torch.zeros((1, n, 2), dtype=torch.float64, device=dev)
        # Mapped host memory: pinned (hipHostMalloc) buffers are addressable by the GPU at their host address, so the launch can
        # read the commands from and write obs | rew | done to the pinned blocks directly -- a step is then ONE launch and a
        # stream synchronize, no copy calls (UAVX_FACADE_COPIES=1 keeps the H2D / D2H copies, for A/B; tools/facade_rate.py).
        self._mapped = os.environ.get("UAVX_FACADE_COPIES") != "1" and self._host.is_pinned()   # (only page-locked memory is mapped)
        hp = self._host.data_ptr()
        self._io_ptrs = (self._act_host.data_ptr(), hp, hp + n * 40, hp + n * 44)

    # -- counters live on the device (MUW:166-168,209,221,238) ----------------------------------------
    def _counter(self, k):
        return int(self._batched.metrics()[0, k].item())

    def _set_counter(self, k, v):
        c = self._batched.metrics()
        c[0, k] = int(v)
        self._batched.set_state(counters=c)

    steps = property(lambda s: s._counter(0), lambda s, v: s._set_counter(0, v))
    target_reach_count = property(lambda s: s._counter(1), lambda s, v: s._set_counter(1, v))
    collision_count = property(lambda s: s._counter(2), lambda s, v: s._set_counter(2, v))

    def _get_info(self):
        return {"distance": 0}  # MUW:111-114

    def _obs_list(self, obs):
        """N arrays of shape (10,), dtype float64 like the reference's `np.array([...python floats...])` (MUW:98-109); the
        values are the kernel's float32 ones, widened exactly."""
        o = obs[0].cpu().numpy().astype(np.float64)
        return [o[i].copy() for i in range(self.num_agents)]

    def _draw_layout(self):
        """Start/target points with the reference's rejection rules and np.random draw order."""
        n, lo, hi = self.num_agents, self.min_location, self.max_location
        two_r = 2 * self.collider_radius
        loc = np.zeros((n, 2), np.float32)
        tgt = np.zeros((n, 2), np.float32)
        for i in range(n):  # MUW:126-137
            while True:
                cand = np.random.uniform(lo, high=hi, size=(2,)).astype(np.float32)
                if all(_norm32(loc[j] - cand) > two_r for j in range(i)):
                    break
            loc[i] = cand
        for i in range(n):  # MUW:140-153
            while True:
                cand = np.random.uniform(lo, high=hi, size=(2,)).astype(np.float32)
                if _norm32(cand - loc[i]) > two_r and all(_norm32(tgt[j] - cand) > two_r for j in range(i)):
                    break
            tgt[i] = cand
        return loc, tgt

    def reset(self, return_info=False, circular=False):  # MUW:116
        n = self.num_agents
        loc, tgt = self._draw_layout()
        if circular:
            # MUW:157-163 installs float64 arrays: the episode runs in the float64-position mode of the library
            return_obs = self._batched.reset_circular()
            obs = self._obs_list(return_obs)
            return (obs, self._get_info()) if return_info else obs
        self._batched.set_position_mode("float32")                                   # MUW:126: float32 arrays again
        d = tgt - loc
        init_d = np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32)  # MUW:154
        self._batched.set_state(loc=loc[None], tgt=tgt[None], vel=np.zeros((1, n, 2)), init_d=init_d[None],
                                prev_d=init_d[None], flags=np.zeros((1, n), np.uint8),
                                counters=np.array([[0, 0, 0, int(self._batched.metrics()[0, 3].item()) + 1]]))
        obs = self._obs_list(self._batched.observe())  # MUW:170-172
        return (obs, self._get_info()) if return_info else obs

    def step(self, n_action, evaluate=False):  # MUW:177
        n = self.num_agents
        if len(n_action) != n:
            raise IndexError(f"n_action must hold {n} actions of 2 components")
        for i in range(n):
            self._act_np[0, i] = n_action[i]        # float32 or float64 commands widen exactly
        b = self._batched
        if self._mapped:
            a, o, r, d = self._io_ptrs
            rc = b._L.uavx_step(b._h, a, _lib.F64, 1 if evaluate else 0, o, r, d, b._stream())
            if rc:
                _lib.check(rc, b._h)
        else:
            self._act_d.copy_(self._act_host, non_blocking=True)
            b.step(self._act_d, evaluate=evaluate, out=(self._obs_d, self._rew_d, self._done_d))
            self._host.copy_(self._pack, non_blocking=True)
        torch.cuda.current_stream(b.device).synchronize()   # (hipStreamSynchronize on the raw stream measured no faster)
        h = self._host_np
        obs = h[: n * 40].view(np.float32).reshape(n, 10).astype(np.float64)   # MUW:98-109 returns float64 arrays
        # N row views of this step's fresh array, Python floats (float32 widened exactly) and bools, as MUW:241 returns them
        return (list(obs), h[n * 40: n * 44].view(np.float32).tolist(), h[n * 44: n * 45].view(np.bool_).tolist(), self._get_info())

    def render(self, mode="human"):
        """MUW:243-331.  mode="human" is a no-op (no display / pygame on a compute node, and the trainers call
        it every step); mode="rgb_array" rasterises the same scene with numpy from one get_state(): targets as
        squares, UAVs as discs with a heading tick and collider ring, agent 0's sensing ring and the lines to its
        (up to) two nearest neighbours."""
        if mode != "rgb_array":
            return None
        from .render import draw_world
        st = {k: v[0].cpu().numpy() for k, v in self._batched.get_state().items()}
        return draw_world(st["loc"], st["tgt"], st["vel"], [a.color for a in self.agent_list], self.x_size, self.y_size,
                          self.collider_radius, self.d_sense)

    def close(self):  # MUW:333
        self._batched.close()
