"""Experiment: how long ONE accept / reject chain takes (uavx_reset with a sparse mask: every wavefront that has work runs the
chain for one env, all of them in parallel) next to an empty launch.  usage: python tools/exp_chain.py [E] [L] [B]"""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_uav_collision_avoidance_amd import BatchedMultiUAVWorld2D
dev = torch.device("cuda", 0)
E, N, B = (int(x) for x in (sys.argv[1:4] + ["65536", "8", "16"][len(sys.argv) - 1:]))
env = BatchedMultiUAVWorld2D(E, num_agents=N, num_bodies=B, device=dev)
env.reset()
g = torch.Generator(device=dev).manual_seed(3)
for label, n in (("none", 0), ("1 env", 1), ("100 envs", 100), ("1000 envs", 1000), ("every 8th env", -8), ("all", -1)):
    mask = torch.zeros(E, dtype=torch.uint8, device=dev)
    if n > 0:
        mask[torch.randperm(E, generator=g, device=dev)[:n]] = 1
    elif n < 0:
        mask[::-n] = 1
    for _ in range(3):
        env.reset(mask=mask)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(20):
        ev0.record(); env._L.uavx_reset(env._h, mask.data_ptr(), env.seed, None, env._stream()); ev1.record()
        torch.cuda.synchronize()
        ts.append(ev0.elapsed_time(ev1) * 1e3)
    ts.sort()
    print(f"{label:14s} reset launch {ts[len(ts) // 2]:8.2f} us (min {ts[0]:.2f})", flush=True)
