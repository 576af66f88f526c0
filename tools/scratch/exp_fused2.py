import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_uav_collision_avoidance_amd import BatchedMultiUAVWorld2D
dev = torch.device("cuda", 0)
E, N, R, K = 65536, 4, 20, 2000
g = torch.Generator(device=dev).manual_seed(1)
cart = (torch.rand((R, E, N, 2), generator=g, device=dev) * 2 - 1) * 10
zero = torch.zeros_like(cart)
def timeit(fn, warm=0):
    env = BatchedMultiUAVWorld2D(E, num_agents=N, device=dev)
    env.reset()
    f = fn(env)
    for i in range(warm): f(i % R)
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
    m = env.metrics().float().mean(0).tolist()
    env.close()
    return dt * 1e6, m
cases = {
 "ex_cap2 (every env resets every 2nd call)": (lambda env: (lambda i: env.step_ex(cart[i], track_returns=False, step_cap=1)), 3),
 "ex_cap8": (lambda env: (lambda i: env.step_ex(cart[i], track_returns=False, step_cap=7)), 3),
 "ex_cap64": (lambda env: (lambda i: env.step_ex(cart[i], track_returns=False, step_cap=63)), 3),
 "step_fresh": (lambda env: (lambda i: env.step(cart[i])), 3),
 "step_zero_actions": (lambda env: (lambda i: env.step(zero[i])), 3),
 "ex_agent0_zero_actions": (lambda env: (lambda i: env.step_ex(zero[i], track_returns=False, auto_reset="agent0_done")), 3),
 "ex_agent0_random": (lambda env: (lambda i: env.step_ex(cart[i], track_returns=False, auto_reset="agent0_done")), 3),
 "ex_never_random_cap300": (lambda env: (lambda i: env.step_ex(cart[i], track_returns=False, step_cap=300)), 3),
}
for name, (fn, warm) in cases.items():
    t, m = timeit(fn, warm)
    print(name, "%.2f us" % t, "mean counters", [round(x, 2) for x in m], flush=True)
