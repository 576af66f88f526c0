#!/usr/bin/env python3
"""Where the fused launch's extra time goes: graph-replayed step variants at 65 536 x 4, us per launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_uav_collision_avoidance_amd import BatchedMultiUAVWorld2D

dev = torch.device("cuda", 0)
E, N, R = 65536, 4, 50
g = torch.Generator(device=dev).manual_seed(1)
pol = torch.rand((R, E, N, 2), generator=g, device=dev) * 2 - 1
cart = pol * 10


def timed(name, fn, ring):
    env = BatchedMultiUAVWorld2D(E, num_agents=N, device=dev)
    env.reset()
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for i in range(3):
            fn(env, ring[i])
    torch.cuda.current_stream(dev).wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for i in range(R):
            fn(env, ring[i])
    for _ in range(4):
        graph.replay()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / (20 * R))
    print(f"{name:58s} {best:6.2f} us", flush=True)
    env.close()


timed("step", lambda e, a: e.step(a), cart)
timed("step_ex defaults, track_returns=False", lambda e, a: e.step_ex(a, track_returns=False), cart)
timed("step_ex track_returns", lambda e, a: e.step_ex(a, track_returns=True), cart)
timed("step_ex polar", lambda e, a: e.step_ex(a, polar=True, track_returns=False), pol)
timed("step_ex polar + track", lambda e, a: e.step_ex(a, polar=True, track_returns=True), pol)
timed("step_ex polar + track + agent0_done (resets happen)", lambda e, a: e.step_ex(a, polar=True, auto_reset="agent0_done"), pol)
timed("step_ex polar + track + agent0_done + cap 1500", lambda e, a: e.step_ex(a, polar=True, auto_reset="agent0_done", step_cap=1500), pol)
