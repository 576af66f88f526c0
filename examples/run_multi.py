#!/usr/bin/env python3
"""The reference's run_multi.py (random-action demo, run_multi.py:5-23) against the MI355X build, with the reference's
own import line: install_alias() (opt-in) makes `gym_uav_collision_avoidance` resolve to this build.  The reference blocks
on input() and renders with pygame each step; here the loop runs a fixed number of steps and can dump rgb_array frames."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gym_uav_collision_avoidance_amd
gym_uav_collision_avoidance_amd.install_alias()                 # or: UAVX_ALIAS=1 / PYTHONPATH=<package>/compat
from gym_uav_collision_avoidance.envs import MultiUAVWorld2D    # run_multi.py:2, unchanged

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--frames", type=str, default=None, help="directory for .npy rgb_array frames (every 20th step)")
args = ap.parse_args()

num_agent = 5
env = MultiUAVWorld2D(num_agents=num_agent)
observation, info = env.reset(return_info=True)
for t in range(args.steps):
    n_action = [env.action_space.sample() for _ in range(num_agent)]
    observation, reward, done, info = env.step(n_action)
    env.render()
    if args.frames and t % 20 == 0:
        os.makedirs(args.frames, exist_ok=True)
        np.save(os.path.join(args.frames, f"frame_{t:05d}.npy"), env.render(mode="rgb_array"))
    if t % 50 == 0:
        print(t, observation[0])
    if done[0]:
        observation, info = env.reset(return_info=True)
print("steps", env.steps, "reached", env.target_reach_count, "collisions", env.collision_count)
env.close()
