"""Fused step_ex with agent-0-done auto-reset for one agent count, layouts drawn ahead off / on (slices = every).
usage: python tools/exp_prefetch_n.py N [E]"""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_uav_collision_avoidance_amd import BatchedMultiUAVWorld2D
N = int(sys.argv[1]); E = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
dev = torch.device("cuda", 0)
R, K = 16, 3200
g = torch.Generator(device=dev).manual_seed(1)
ring = torch.rand((R, E, N, 2), generator=g, device=dev) * 2 - 1
for every in (0, 4, 16, 64, 0, 16):
    env = BatchedMultiUAVWorld2D(E, num_agents=N, device=dev)
    env.set_prefetch(every)
    env.reset()
    f = lambda i: env.step_ex(ring[i], polar=True, track_returns=True, auto_reset="agent0_done", step_cap=1500)
    for i in range(3): f(i)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for i in range(R): f(i)
    for _ in range(20): graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K // R): graph.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    eps = int(env.episode_stats()["episodes"].sum().item())
    print(f"N={N} E={E} every={every:3d}: {dt * 1e6:7.2f} us   resets/launch {eps / (K + 20 * R + 3):7.1f}", flush=True)
    env.close()
