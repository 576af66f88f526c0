#!/usr/bin/env python3
"""Host cost of one env call in an EAGER Python loop (no hipGraph): calls per second with a batch so small (4 096 envs) that the GPU
is never the bound -- what a trainer that does not capture its loop pays per step before the kernel even matters.

    python tools/host_cost.py          (on the GPU box)
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gym_uav_collision_avoidance_amd import BatchedMultiUAVWorld2D, BatchedUAVWorld2D, UAVVectorEnv  # noqa: E402


def rate(fn, n=20000):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()          # host time to ENQUEUE n calls
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return dict(host_us_per_call=(t1 - t0) / n * 1e6, us_per_call_incl_drain=(t2 - t0) / n * 1e6)


def main():
    E = 4096
    dev = torch.device("cuda", 0)
    out = {}
    m = BatchedMultiUAVWorld2D(E, num_agents=4, device=dev); m.reset()
    a = torch.rand((E, 4, 2), device=dev) * 2 - 1
    out["multi.step"] = rate(lambda: m.step(a))
    out["multi.step_ex"] = rate(lambda: m.step_ex(a, polar=True, auto_reset="agent0_done", step_cap=1500))
    v = UAVVectorEnv(E, num_agents=4, polar=True, step_cap=1500, device=dev); v.reset(seed=0)
    out["vector.step"] = rate(lambda: v.step(a))
    u = BatchedUAVWorld2D(E, device=dev); u.reset()
    au = torch.rand((E, 2), device=dev) * 2 - 1
    out["uw.step"] = rate(lambda: u.step(au))
    out["uw.step_ex"] = rate(lambda: u.step_ex(au, polar=True, auto_reset=True, step_cap=1500))
    for k, r in out.items():
        print(f"{k:14s} host {r['host_us_per_call']:.2f} us per call   (with the queue drained: {r['us_per_call_incl_drain']:.2f})")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
