"""Experiment: what the reset machinery of a fused step_ex launch costs in steady state, piece by piece (same handle shape, 16-step
graphs, wall clock over many replays after a long warm-up so that every env has its layouts parked).
usage: python tools/exp_reset_cost.py [E] [L] [B] [warm]"""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_uav_collision_avoidance_amd import BatchedMultiUAVWorld2D
dev = torch.device("cuda", 0)
E, N, B, WARM = (int(x) for x in (sys.argv[1:5] + ["65536", "8", "16", "3000"][len(sys.argv) - 1:]))
R, K = 16, 3200
g = torch.Generator(device=dev).manual_seed(1)
ring = (torch.rand((R, E, N, 2), generator=g, device=dev) * 2 - 1)
MODES = (("agent-0-done + cap 1500 (resets, staging)", dict(auto_reset="agent0_done", step_cap=1500)),
         ("cap 2^30 only (staging workgroups scan, nothing ever resets)", dict(step_cap=1 << 30)),
         ("no policy (no staging workgroups)", dict()))
for name, kw in MODES:
    env = BatchedMultiUAVWorld2D(E, num_agents=N, num_bodies=B, device=dev)
    env.reset()
    f = lambda i: env.step_ex(ring[i], polar=True, track_returns=True, **kw)
    graph = torch.cuda.CUDAGraph()
    for i in range(R): f(i)
    with torch.cuda.graph(graph):
        for i in range(R): f(i)
    for _ in range(WARM // R): graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K // R): graph.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    eps = int(env.episode_stats()["episodes"].sum().item())
    print(f"{name:64s} {dt * 1e6:7.2f} us   episodes ended per launch {eps / (K + WARM + 2 * R):6.1f}", flush=True)
    env.close()
