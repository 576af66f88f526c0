import sys, os, ctypes, shutil
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gym_uav_collision_avoidance_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "dbg", "libuavx_stamps.so")
from gym_uav_collision_avoidance_amd import BatchedMultiUAVWorld2D
dev = torch.device("cuda", 0)
E, N, R = 65536, 4, 20
g = torch.Generator(device=dev).manual_seed(1)
cart = (torch.rand((R, E, N, 2), generator=g, device=dev) * 2 - 1) * 10
env = BatchedMultiUAVWorld2D(E, num_agents=N, device=dev)
env.reset()
L = _lib.load()
buf = (ctypes.c_ulonglong * (8 * 4096))(); n = ctypes.c_uint(0)
for k in range(400):
    env.step_ex(cart[k % R], track_returns=False, auto_reset="agent0_done")
L.uavx_debug_stamps(buf, ctypes.byref(n))
for k in range(1):
    env.step_ex(cart[k % R], track_returns=False, auto_reset="agent0_done")
L.uavx_debug_stamps(buf, ctypes.byref(n))
a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8)[: n.value].astype(np.int64)
names = {0: "normal waves", 1: "waves folding a fresh env", 2: "waves drawing a new layout", 3: "both"}
for kind in (0, 1, 2, 3):
    r = a[a[:, 6] == kind]
    if len(r) == 0: continue
    d = np.diff(r[:, :6], axis=1)
    print(names[kind], len(r), "median segment cycles [loads->ballot, fold_load issue, polar, step, stores(+draw)]:",
          np.median(d, axis=0).astype(int).tolist(), " total", int(np.median(r[:, 5] - r[:, 0])), " max total", int((r[:, 5] - r[:, 0]).max()))
