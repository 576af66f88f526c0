#!/usr/bin/env python3
"""Batched counterpart of the reference's SAC data-collection loop (test_sac_multi.py:62-119) with
everything resident in HBM: one policy forward over all (env, agent) rows, the step launch converts the
policy output to a velocity command, steps, auto-resets on dones[0] / step cap, accumulates episode
statistics and writes observation / reward / done straight into the replay ring.

  --bodies K   BASELINE config 5 style extension (no reference semantics, parity unpinned): the last K
               "UAVs" of every world are scripted moving obstacles (they wander between random waypoints);
               the learner only consumes rows of the first N-K agents.
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
ap.add_argument("--steps", type=int, default=500)
ap.add_argument("--horizon", type=int, default=64)
ap.add_argument("--checkpoint", type=str, default=None, help="reference weights.chpt (policy_state_dict)")
args = ap.parse_args()

dev = torch.device("cuda", 0)
N, K = args.agents + args.bodies, args.bodies
env = BatchedMultiUAVWorld2D(args.envs, num_agents=N, device=dev, seed=0)
policy = load_reference_checkpoint(args.checkpoint, dev) if args.checkpoint else GaussianPolicy().to(dev)
mem = DeviceReplay(env, horizon=args.horizon, num_learners=args.agents)
mem.begin(env.reset())
gen = torch.Generator(device=dev).manual_seed(0)
body_heading = torch.rand((args.envs, K), generator=gen, device=dev) * 2 - 1 if K else None

torch.cuda.synchronize()
t0 = time.perf_counter()
with torch.no_grad():
    for t in range(args.steps):
        slot = mem.action_slot()                                  # [E, N, 2] view inside the ring
        slot[:, : args.agents] = policy.act(mem.state[:, : args.agents], evaluate=False, generator=gen)
        if K:                                                     # scripted bodies: slow drift, heading random walk
            body_heading = torch.remainder(body_heading + 0.02 * torch.randn(body_heading.shape, generator=gen, device=dev) + 1, 2) - 1
            slot[:, args.agents:, 0] = -0.4
            slot[:, args.agents:, 1] = body_heading
        obs, rew, done, info = mem.step(polar=True, auto_reset="agent0_done", step_cap=1500, track_returns=True)
        if t % 100 == 99:
            s, a, r, s1, mask = mem.sample(256, generator=gen)    # what a learner update would consume
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{args.envs} envs x {N} agents ({K} scripted bodies): {args.steps} steps in {dt:.3f} s "
      f"= {args.envs * args.steps / dt / 1e6:.1f} M env-steps/s incl. policy forward; replay holds {len(mem)} transitions")
print("ended episodes:", env.evaluation_summary())
env.close()
