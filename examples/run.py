#!/usr/bin/env python3
"""The reference's run.py (run.py:6-16) against the MI355X build: single-UAV world, random actions."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gym_uav_collision_avoidance_amd
gym_uav_collision_avoidance_amd.install_alias()            # opt-in alias: the reference's import line below then means this build
from gym_uav_collision_avoidance.envs import UAVWorld2D    # run.py:2, unchanged

env = UAVWorld2D()
observation, info = env.reset(return_info=True)
episodes = 0
for t in range(2000):
    observation, reward, done, info = env.step(env.action_space.sample())
    env.render()
    if done:
        episodes += 1
        observation, info = env.reset(return_info=True)
print("episodes finished:", episodes, "last obs:", observation)
env.close()
