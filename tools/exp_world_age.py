"""Experiment: µs per bare `step` launch as a function of the world's age (steps since reset, no auto-reset): random polar commands
scatter the UAVs, neighbours leave sensing range, episodes end -- the launch gets cheaper.  usage: python tools/exp_world_age.py [E] [L] [B]"""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_uav_collision_avoidance_amd import BatchedMultiUAVWorld2D
dev = torch.device("cuda", 0)
E, N, B = (int(x) for x in (sys.argv[1:4] + ["65536", "4", "0"][len(sys.argv) - 1:]))
R = 50
g = torch.Generator(device=dev).manual_seed(1234)
mag = torch.rand((R, E, N), generator=g, device=dev) * float(np.sqrt(200.0))
ang = (torch.rand((R, E, N), generator=g, device=dev) * 2 - 1) * np.pi
ring = torch.stack((mag * torch.cos(ang), mag * torch.sin(ang)), dim=-1).contiguous()
env = BatchedMultiUAVWorld2D(E, num_agents=N, num_bodies=B, device=dev)
env.reset()
graph = torch.cuda.CUDAGraph()
for i in range(R): env.step(ring[i])
env.reset()
with torch.cuda.graph(graph):
    for i in range(R): obs, rew, done, info = env.step(ring[i])
env.reset()
age = 0
for blk in range(60):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    graph.replay(); graph.replay()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / (2 * R)
    age += 2 * R
    if blk < 10 or blk % 5 == 4:
        d = done.float().mean().item()
        o = obs.reshape(-1, 10)
        far = (o[:, 4] >= 1.0).float().mean().item()      # no first neighbour within d_sense
        print(f"age {age:5d}: {dt * 1e6:6.2f} us/step   done {d:.3f}   no neighbour in range {far:.3f}", flush=True)
