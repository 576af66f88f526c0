"""Drop-in single-env façade of UAVWorld2D (UW:11) over the batched HIP path (E = 1); drives
run.py:6-16 and test_sac.py-style loops unchanged.  reset() draws location, velocity and target
from np.random in the reference's order (UW:121-126)."""
import os

import numpy as np
import torch

from .. import _lib
from ..batched import BatchedUAVWorld2D


class UAVWorld2D:
    metadata = {"render_fps": 1000}  # UW:12

    def __init__(self, x_size=100.0, y_size=100.0, agent_num=4, max_speed=12.0, max_acceleration=5.0, device=None):
        self._batched = BatchedUAVWorld2D(1, x_size, y_size, agent_num, max_speed, max_acceleration, device=device)
        b = self._batched
        self.x_size, self.y_size = x_size, y_size
        self.map_diagonal_size = b.map_diagonal_size
        self.map_dimension = np.array([x_size, y_size])
        self.min_location, self.max_location = b.min_location, b.max_location
        self.max_speed, self.min_speed = b.max_speed, b.min_speed
        self.max_acceleratoin, self.min_acceleratoin = b.max_acceleratoin, b.min_acceleratoin
        self.tau = b.tau
        self.max_window_size = 800  # MUW:25 / UW:25 (only used by the rgb_array rasteriser here)
        if x_size > y_size:
            self.window_size_x, self.window_size_y = self.max_window_size, self.max_window_size / x_size * y_size
        else:
            self.window_size_y, self.window_size_x = self.max_window_size, self.max_window_size / y_size * x_size
        self.observation_space, self.action_space = b.observation_space, b.action_space
        self.window = None
        self.clock = None
        # pinned staging for the commands and ONE device->host copy of (obs | rew | info | done) per step
        dev = b.device
        self._pack = torch.zeros(32, dtype=torch.uint8, device=dev)      # obs 16 B | reward 4 | distance 4 | done 1
        self._out = (self._pack[:16].view(torch.float32).view(1, 4), self._pack[16:20].view(torch.float32),
                     self._pack[24:25], self._pack[20:24].view(torch.float32))
        self._host = torch.zeros(32, dtype=torch.uint8).pin_memory()
        self._host_np = self._host.numpy()
        self._act_host = {np.dtype(np.float32): torch.zeros((1, 2), dtype=torch.float32).pin_memory(),
                          np.dtype(np.float64): torch.zeros((1, 2), dtype=torch.float64).pin_memory()}
        self._act_dev = {k: torch.zeros_like(v, device=dev) for k, v in self._act_host.items()}
        # Mapped host memory (see MultiUAVWorld2D): the launch reads the command from and writes its outputs to the pinned
        # blocks themselves; UAVX_FACADE_COPIES=1 keeps the copies (A/B).
        self._mapped = os.environ.get("UAVX_FACADE_COPIES") != "1" and self._host.is_pinned()   # (only page-locked memory is mapped)
        hp = self._host.data_ptr()
        self._io_ptrs = (hp, hp + 16, hp + 24, hp + 20)     # obs | reward | done | distance, as in self._out
        self._act_code = {np.dtype(np.float32): _lib.F32, np.dtype(np.float64): _lib.F64}
        self._act_ptr = {k: v.data_ptr() for k, v in self._act_host.items()}

    @property
    def steps(self):
        return int(self._batched.steps[0].item())

    def _get_info(self):  # UW:114-117
        st = self._batched.get_state()
        d = (st["tgt"][0] - st["loc"][0]).cpu().numpy()
        return {"distance": np.sqrt(d[0] * d[0] + d[1] * d[1])}

    def reset(self, return_info=False, options=None):  # UW:119
        loc = np.random.uniform(self.min_location, high=self.max_location, size=(2,)).astype(np.float32)
        vel = np.random.uniform(self.min_speed, high=self.max_speed, size=(2,)).astype(np.float32)
        tgt = np.random.uniform(self.min_location, high=self.max_location, size=(2,)).astype(np.float32)
        d = tgt - loc
        init_d = np.sqrt(d[0] * d[0] + d[1] * d[1])
        episode = int(self._batched.get_state()["counters"][0, 1].item()) + 1
        self._batched.set_state(loc=loc[None], vel=vel.astype(np.float64)[None], tgt=tgt[None], init_d=[init_d],
                                prev_d=[init_d], flags=[_lib.FLAG_VEL_F32], counters=[[0, episode]])
        obs = self._batched.observe()[0].cpu().numpy().astype(np.float64)   # UW:106-111: a float64 array(4,)
        return (obs, self._get_info()) if return_info else obs

    def step(self, action):  # UW:137
        a = np.asarray(action)
        key = np.dtype(np.float32) if a.dtype == np.float32 else np.dtype(np.float64)  # float32 matters on step 1 (UW:142)
        self._act_host[key].numpy()[0] = a
        b = self._batched
        if self._mapped:
            o, r, d, i = self._io_ptrs
            rc = b._L.uavx_uw_step(b._h, self._act_ptr[key], self._act_code[key], o, r, d, i, b._stream())
            if rc:
                _lib.check(rc, b._h, uw=True)
        else:
            self._act_dev[key].copy_(self._act_host[key], non_blocking=True)
            b.step(self._act_dev[key], out=self._out)        # the launch writes straight into the packed block
            self._host.copy_(self._pack, non_blocking=True)
        torch.cuda.current_stream(b.device).synchronize()   # (hipStreamSynchronize on the raw stream measured no faster)
        h = self._host_np
        return (h[:16].view(np.float32).astype(np.float64), np.float32(h[16:20].view(np.float32)[0]), bool(h[24]),
                {"distance": np.float32(h[20:24].view(np.float32)[0])})

    def render(self, mode="human"):  # UW:175 — no-op on a headless node
        return None

    def close(self):  # UW:230
        self._batched.close()
