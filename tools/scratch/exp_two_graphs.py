"""Experiment: the batch as S half/quarter-batch handles, each with its OWN hipGraph replayed on its OWN stream (S independent chains,
no join until the end of the timed region).  usage: python tools/exp_two_graphs.py [E] [N]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_uav_collision_avoidance_amd import BatchedMultiUAVWorld2D
E = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda", 0)
R, K = 50, 4000
g = torch.Generator(device=dev).manual_seed(1)
for S in (1, 2, 3, 4, 1, 2):
    e = E // S
    envs = [BatchedMultiUAVWorld2D(e, num_agents=N, device=dev, env_offset=k * e) for k in range(S)]
    rings = [(torch.rand((R, e, N, 2), generator=g, device=dev) * 20 - 10) for _ in range(S)]
    streams = [torch.cuda.Stream(dev) for _ in range(S)]
    graphs = []
    for k in range(S):
        envs[k].reset()
        with torch.cuda.stream(streams[k]):
            for i in range(3): envs[k].step(rings[k][i])
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=streams[k]):
            for i in range(R): envs[k].step(rings[k][i])
        graphs.append(gr)
    def burst(n):
        for _ in range(n):
            for k in range(S):
                with torch.cuda.stream(streams[k]):
                    graphs[k].replay()
    burst(5); torch.cuda.synchronize()
    t0 = time.perf_counter()
    burst(K // R)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(f"E={e * S} N={N} chains={S}: {dt * 1e6:7.3f} us per step of the whole batch  ({e * S / dt / 1e9:.2f} G env-steps/s)", flush=True)
    for x in envs: x.close()
