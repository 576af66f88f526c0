"""Experiment: step time vs batch size, and one batch split into C independent chains on C streams
captured in one hipGraph (envs are independent, so chains never synchronise until the end)."""
import sys, os, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_uav_collision_avoidance_amd import BatchedMultiUAVWorld2D

dev = torch.device("cuda", 0)
N = 4
def actions(R, E):
    g = torch.Generator(device=dev).manual_seed(1)
    a = torch.rand((R, E, N, 2), generator=g, device=dev) * 2 - 1
    v = (a[..., 0] / 2 + 0.5) * 14.142; th = a[..., 1] * np.pi
    return torch.stack([v * torch.cos(th), v * torch.sin(th)], -1).contiguous()

def run(E, chains, R=20, K=2000):
    per = E // chains
    envs = [BatchedMultiUAVWorld2D(per, num_agents=N, device=dev, env_offset=c * per) for c in range(chains)]
    rings = [actions(R, per) for _ in range(chains)]
    for e in envs: e.reset()
    for e, r in zip(envs, rings):
        e.step(r[0])
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(dev) for _ in range(chains)]
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        cur = torch.cuda.current_stream(dev)
        if chains == 1:
            for i in range(R): envs[0].step(rings[0][i])
        else:
            for s in streams: s.wait_stream(cur)
            for c, s in enumerate(streams):
                with torch.cuda.stream(s):
                    for i in range(R): envs[c].step(rings[c][i])
            for s in streams: cur.wait_stream(s)
    for _ in range(5): graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K // R): graph.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (K // R * R)
    for e in envs: e.close()
    return dt

for E, chains in ((65536, 1), (65536, 2), (65536, 4), (16384, 1), (32768, 1), (131072, 1), (262144, 1), (1048576, 1), (1048576, 2), (4194304, 1)):
    K = 2000 if E <= 262144 else 400
    dt = run(E, chains, K=K)
    print(json.dumps(dict(E=E, chains=chains, us_per_step=dt * 1e6, env_steps_per_s=E / dt, alg_GBs=E * 452 / dt / 1e9)), flush=True)
