import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_uav_collision_avoidance_amd import BatchedMultiUAVWorld2D
dev = torch.device("cuda", 0)
E, N, R = 65536, 4, 20
mode = sys.argv[1]
g = torch.Generator(device=dev).manual_seed(1)
cart = (torch.rand((R, E, N, 2), generator=g, device=dev) * 2 - 1) * 10
env = BatchedMultiUAVWorld2D(E, num_agents=N, device=dev)
env.reset()
for k in range(600):
    if mode == "agent0":
        env.step_ex(cart[k % R], track_returns=False, auto_reset="agent0_done")
    elif mode == "cap300":
        env.step_ex(cart[k % R], track_returns=False, step_cap=300)
    else:
        env.step(cart[k % R])
torch.cuda.synchronize()
