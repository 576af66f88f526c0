import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_uav_collision_avoidance_amd import BatchedMultiUAVWorld2D
dev = torch.device("cuda", 0)
E, N = 65536, 4
g = torch.Generator(device=dev).manual_seed(1)
env = BatchedMultiUAVWorld2D(E, num_agents=N, device=dev)
env.reset()
tot = 0
for t in range(600):
    a = torch.rand((E, N, 2), generator=g, device=dev) * 20 - 10
    obs, rew, done, info = env.step_ex(a, auto_reset="agent0_done", track_returns=False)
    if t % 100 == 99:
        print(t, "reset frac this step", float(info["reset_mask"].float().mean()), "done0 frac", float(done[:, 0].float().mean()))
st = env.episode_stats()
print("mean episode len", float(st["steps"].sum()) / max(1, float(st["episodes"].sum())), "episodes", int(st["episodes"].sum()))
