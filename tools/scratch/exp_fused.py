"""Experiment: cost of each step_ex option vs the bare step (same process, same box)."""
import sys, os, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_uav_collision_avoidance_amd import BatchedMultiUAVWorld2D
dev = torch.device("cuda", 0)
E, N, R, K = 65536, 4, 20, 2000
g = torch.Generator(device=dev).manual_seed(1)
ring = (torch.rand((R, E, N, 2), generator=g, device=dev) * 2 - 1)
cart = ring * 10

def timeit(fn):
    env = BatchedMultiUAVWorld2D(E, num_agents=N, device=dev)
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
    env.close()
    return dt * 1e6

cases = {
 "step": lambda env: (lambda i: env.step(cart[i])),
 "ex_plain": lambda env: (lambda i: env.step_ex(cart[i], track_returns=False)),
 "ex_polar": lambda env: (lambda i: env.step_ex(ring[i], polar=True, track_returns=False)),
 "ex_track": lambda env: (lambda i: env.step_ex(cart[i], track_returns=True)),
 "ex_cap": lambda env: (lambda i: env.step_ex(cart[i], track_returns=False, step_cap=1500)),
 "ex_agent0": lambda env: (lambda i: env.step_ex(cart[i], track_returns=False, auto_reset="agent0_done")),
 "ex_all": lambda env: (lambda i: env.step_ex(ring[i], polar=True, track_returns=True, auto_reset="agent0_done", step_cap=1500)),
}
for rep in range(2):
    for name, fn in cases.items():
        print(name, "%.2f us" % timeit(fn), flush=True)
