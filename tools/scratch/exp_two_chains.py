"""Experiment: the 65 536-env batch as S independent handles (E/S envs each) whose step chains are captured into ONE hipGraph as
parallel branches (fork at the start, join at the end of the graph): do the launch boundaries of one chain hide under the kernels
of the other?   usage: python tools/exp_two_chains.py [E] [N]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_uav_collision_avoidance_amd import BatchedMultiUAVWorld2D
E = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda", 0)
R, K = 50, 4000
g = torch.Generator(device=dev).manual_seed(1)
for S in (1, 2, 4, 1, 2):
    envs = [BatchedMultiUAVWorld2D(E // S, num_agents=N, device=dev, env_offset=k * (E // S)) for k in range(S)]
    rings = [(torch.rand((R, E // S, N, 2), generator=g, device=dev) * 20 - 10) for _ in range(S)]
    streams = [torch.cuda.Stream(dev) for _ in range(S)]
    for e in envs: e.reset()
    for k in range(S):
        with torch.cuda.stream(streams[k]):
            for i in range(3): envs[k].step(rings[k][i])
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        main = torch.cuda.current_stream(dev)
        for k in range(S):
            streams[k].wait_stream(main)
            with torch.cuda.stream(streams[k]):
                for i in range(R): envs[k].step(rings[k][i])
        for k in range(S):
            main.wait_stream(streams[k])
    for _ in range(5): graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K // R): graph.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(f"E={E} N={N} handles={S}: {dt * 1e6:7.3f} us per step of the whole batch  ({E / dt / 1e9:.2f} G env-steps/s)", flush=True)
    for e in envs: e.close()
