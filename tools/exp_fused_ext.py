"""Experiment: cost of the step_ex options on a world with scripted bodies (configs[4] shape), same process / box.
usage: python tools/exp_fused_ext.py [E] [L] [B]"""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_uav_collision_avoidance_amd import BatchedMultiUAVWorld2D
dev = torch.device("cuda", 0)
E, N, B = (int(x) for x in (sys.argv[1:4] + ["65536", "8", "16"][len(sys.argv) - 1:]))
R, K = 16, 1600
g = torch.Generator(device=dev).manual_seed(1)
ring = (torch.rand((R, E, N, 2), generator=g, device=dev) * 2 - 1)
cart = ring * 10

PREFETCH = int(os.environ.get("UAVX_PREFETCH", "16"))

def timeit(fn, levels=False):
    env = BatchedMultiUAVWorld2D(E, num_agents=N, num_bodies=B, device=dev)
    env.set_prefetch(PREFETCH)
    if levels:
        env.set_curriculum([dict(x_size=40, y_size=40, collider_radius=1.0, d_sense=15), dict(x_size=50, y_size=50, collider_radius=1.0, d_sense=15)], lo=0, hi=1)
    env.reset()
    f = fn(env)
    for i in range(3): f(i)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for i in range(R): f(i)
    for _ in range(5): graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K // R): graph.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    eps = int(env.episode_stats()["episodes"].sum().item())
    env.close()
    return dt * 1e6, eps / (K + 5 * R + 3)

cases = {
 "step": lambda env: (lambda i: env.step(cart[i])),
 "ex_plain": lambda env: (lambda i: env.step_ex(cart[i], track_returns=False)),
 "ex_polar_track": lambda env: (lambda i: env.step_ex(ring[i], polar=True, track_returns=True)),
 "ex_cap1500": lambda env: (lambda i: env.step_ex(cart[i], track_returns=False, step_cap=1500)),
 "ex_cap64": lambda env: (lambda i: env.step_ex(cart[i], track_returns=False, step_cap=64)),
 "ex_agent0": lambda env: (lambda i: env.step_ex(cart[i], track_returns=False, auto_reset="agent0_done")),
 "ex_all": lambda env: (lambda i: env.step_ex(ring[i], polar=True, track_returns=True, auto_reset="agent0_done", step_cap=1500)),
}
for rep in range(2):
    for name, fn in cases.items():
        us, rate = timeit(fn)
        print(f"{name:16s} {us:7.2f} us   resets/launch {rate:8.1f}", flush=True)
us, rate = timeit(cases["ex_all"], levels=True)
print(f"{'ex_all+levels':16s} {us:7.2f} us   resets/launch {rate:8.1f}", flush=True)
