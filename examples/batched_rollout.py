#!/usr/bin/env python3
"""Batched counterpart of the reference's SAC data-collection loop (test_sac_multi.py:62-119) with
everything resident in HBM: one policy forward over all (env, agent) rows, the step launch converts the
policy output to a velocity command, steps, auto-resets on dones[0] / step cap, accumulates episode
statistics and writes observation / reward / done straight into the replay ring.

  --bodies K      BASELINE configs[4] (extension, no reference semantics): K scripted obstacles per world, stepped inside
                  the kernel between Philox waypoints; every tensor stays [E, agents, ...] (no action / observation rows
                  for bodies).
  --curriculum    randomized-reset curriculum: three world levels (box, d_sense, collider radius, active learners /
                  bodies); the window an env draws its level from at each auto-reset widens as the run progresses.
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_uav_collision_avoidance_amd import BatchedMultiUAVWorld2D
from gym_uav_collision_avoidance_amd.policy import GaussianPolicy, load_reference_checkpoint
from gym_uav_collision_avoidance_amd.replay import DeviceReplay

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=65536)
ap.add_argument("--agents", type=int, default=8)
ap.add_argument("--bodies", type=int, default=0)
ap.add_argument("--curriculum", action="store_true")
ap.add_argument("--steps", type=int, default=500)
ap.add_argument("--horizon", type=int, default=64)
ap.add_argument("--checkpoint", type=str, default=None, help="reference weights.chpt (policy_state_dict)")
args = ap.parse_args()

dev = torch.device("cuda", 0)
N, K = args.agents, args.bodies
env = BatchedMultiUAVWorld2D(args.envs, num_agents=N, num_bodies=K, device=dev, seed=0)
levels = [dict(x_size=30.0, y_size=30.0, collider_radius=0.5, d_sense=10.0, n_active=max(1, N // 4), b_active=K // 4),
          dict(x_size=40.0, y_size=40.0, collider_radius=0.8, d_sense=12.0, n_active=max(1, N // 2), b_active=K // 2),
          dict(x_size=50.0, y_size=50.0, collider_radius=1.0, d_sense=15.0, n_active=N, b_active=K)]
if args.curriculum:
    env.set_curriculum(levels, lo=0, hi=0)
policy = load_reference_checkpoint(args.checkpoint, dev) if args.checkpoint else GaussianPolicy().to(dev)
mem = DeviceReplay(env, horizon=args.horizon)
mem.begin(env.reset())
gen = torch.Generator(device=dev).manual_seed(0)

torch.cuda.synchronize()
t0 = time.perf_counter()
with torch.no_grad():
    for t in range(args.steps):
        if args.curriculum and t in (args.steps // 3, 2 * args.steps // 3):
            env.set_level_window(0, 1 if t < 2 * args.steps // 3 else 2)   # harder worlds enter the draw
        slot = mem.action_slot()                                  # [E, N, 2] view inside the ring
        slot.copy_(policy.act(mem.state, evaluate=False, generator=gen))
        obs, rew, done, info = mem.step(polar=True, auto_reset="agent0_done", step_cap=1500, track_returns=True)
        if t % 100 == 99:                                         # what a learner update would consume
            s, a, r, s1, mask, truncated, ended = mem.sample(256, generator=gen, with_flags=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{args.envs} envs x {N} UAVs + {K} scripted bodies: {args.steps} steps in {dt:.3f} s "
      f"= {args.envs * args.steps / dt / 1e6:.1f} M env-steps/s incl. policy forward; replay holds {len(mem)} transitions")
print("ended episodes:", env.evaluation_summary())
if args.curriculum:
    print("levels in force:", torch.bincount(env.env_levels().long(), minlength=3).tolist())
env.close()
