"""Single-env façade cost (the literal drop-in for run_multi.py / the trainers' loops)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_uav_collision_avoidance_amd.envs import MultiUAVWorld2D, UAVWorld2D
for n in (1, 4, 8):
    env = MultiUAVWorld2D(num_agents=n)
    np.random.seed(0)
    env.reset()
    acts = [np.array([3.0, -2.0]) for _ in range(n)]
    for _ in range(50): env.step(acts)
    t0 = time.perf_counter()
    K = 2000
    for _ in range(K):
        obs, rew, done, info = env.step(acts)
    dt = (time.perf_counter() - t0) / K
    print(f"MultiUAVWorld2D(num_agents={n}) facade: {dt * 1e6:.1f} us/step = {1 / dt:.0f} env-steps/s")
    env.close()
env = UAVWorld2D()
env.reset()
a = np.array([3.0, -2.0], dtype=np.float32)
for _ in range(50): env.step(a)
t0 = time.perf_counter()
for _ in range(2000): env.step(a)
dt = (time.perf_counter() - t0) / 2000
print(f"UAVWorld2D facade: {dt * 1e6:.1f} us/step = {1 / dt:.0f} env-steps/s")
