"""Experiment: fused step_ex with agent-0-done auto-reset on the configs[4] shape for several staging cadences
(uavx_set_prefetch).  usage: python tools/exp_pf.py [E] [L] [B]"""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_uav_collision_avoidance_amd import BatchedMultiUAVWorld2D
dev = torch.device("cuda", 0)
E, N, B = (int(x) for x in (sys.argv[1:4] + ["65536", "8", "16"][len(sys.argv) - 1:]))
R, K = 16, 1600
g = torch.Generator(device=dev).manual_seed(1)
ring = (torch.rand((R, E, N, 2), generator=g, device=dev) * 2 - 1)

def timeit(pf, kw):
    env = BatchedMultiUAVWorld2D(E, num_agents=N, num_bodies=B, device=dev)
    env.set_prefetch(pf)
    env.reset()
    f = lambda i: env.step_ex(ring[i], polar=True, track_returns=True, **kw)
    for i in range(300): f(i % R)     # every env gets its first layout parked
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
    return dt * 1e6, eps / (K + 5 * R + 300)

for pf in [int(v) for v in os.environ.get('UAVX_PF_LIST', '16,64,256,1024,0').split(',')]:
    for name, kw in (("agent0", dict(auto_reset="agent0_done", step_cap=1500)), ("none", dict())):
        us, rate = timeit(pf, kw)
        print(f"prefetch {pf:5d} {name:8s} {us:7.2f} us   resets/launch {rate:8.1f}", flush=True)
