#!/usr/bin/env python3
"""Per-wave phase timeline of one fused step launch (diagnostic build with s_memtime stamps, -DUAVX_STAMPS):
when do waves start, when is their state in registers, when is the arithmetic done, when are the stores issued.
    hipcc ... -DUAVX_STAMPS -o tools/dbg/libuavx_stamps.so ; python tools/exp_phase.py"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gym_uav_collision_avoidance_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "dbg", "libuavx_stamps.so")
from gym_uav_collision_avoidance_amd import BatchedMultiUAVWorld2D

dev = torch.device("cuda", 0)
E, N, R = int(sys.argv[1]) if len(sys.argv) > 1 else 65536, 4, 20
g = torch.Generator(device=dev).manual_seed(1)
cart = (torch.rand((R, E, N, 2), generator=g, device=dev) * 2 - 1) * 10
env = BatchedMultiUAVWorld2D(E, num_agents=N, device=dev)
env.reset()
L = _lib.load()
buf = (ctypes.c_ulonglong * (8 * 4096))(); n = ctypes.c_uint(0)
for k in range(60):
    env.step_ex(cart[k % R], track_returns=False)
L.uavx_debug_stamps(buf, ctypes.byref(n))
for rep in range(3):
    env.step_ex(cart[rep], track_returns=False)
    L.uavx_debug_stamps(buf, ctypes.byref(n))
    a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8)[: n.value].astype(np.int64)
    t = a[:, :6] - a[:, 0].min()
    tick = 0.01  # s_memtime: 100 MHz reference clock -> us
    q = lambda x: "min %5.2f  med %5.2f  p90 %5.2f  max %5.2f" % tuple(np.percentile(x * tick, [0, 50, 90, 100]))
    print(f"launch {rep}: {len(a)} sampled waves (us since the first sampled wave started)")
    print("  wave start                 ", q(t[:, 0]))
    print("  env record arrived         ", q(t[:, 1]))
    print("  state + action in registers", q(t[:, 3]))
    print("  arithmetic done            ", q(t[:, 4]))
    print("  stores issued (wave end)   ", q(t[:, 5]))
    print("  per-wave: load wait %5.2f  compute %5.2f  store issue %5.2f  (medians)" % (
        np.median(t[:, 3] - t[:, 0]) * tick, np.median(t[:, 4] - t[:, 3]) * tick, np.median(t[:, 5] - t[:, 4]) * tick))
    if rep == 2:
        order = np.argsort(a[:, 7])
        print("  block : start -> end (10 ns units), in block order")
        print("  " + "  ".join(f"{int(a[i,7])}:{int(t[i,0])}->{int(t[i,5])}" for i in order))
