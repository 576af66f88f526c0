#!/usr/bin/env python3
"""The closed rollout loop of the reference's trainers (test_sac_multi.py:67-123: policy forward -> polar conversion ->
env.step -> replay push, "Steps Per Sec" printed for the WHOLE loop) in its batched form, captured as ONE hipGraph per ring
pass: batched actor forward over all (env, agent) rows -> written into the replay ring's action slot -> fused uavx_step_ex
(conversion, step, auto-reset, statistics) writing observation / reward / done into the ring.  Reports the step time of the
whole loop, of the env launch alone and of the actor alone, i.e. what share of a real rollout step the env launch is.

    python tools/closed_loop.py [--out profiles/r03_closed_loop.json]      (GPU box; both BASELINE configs[2] and [4])
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gym_uav_collision_avoidance_amd import BatchedMultiUAVWorld2D, _lib      # noqa: E402
from gym_uav_collision_avoidance_amd.policy import DDPGActor, GaussianPolicy, TD3Actor   # noqa: E402
from gym_uav_collision_avoidance_amd.replay import DeviceReplay               # noqa: E402

dev = torch.device("cuda", 0)
RING = 8           # ring slots = steps per graph (the slot indices of a pass repeat, so one graph serves every pass)
STEP_KW = dict(polar=True, auto_reset="agent0_done", step_cap=1500, track_returns=True)


def graph_of(fn, warm=3):
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(warm):
            fn()
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize(dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g


def time_graph(g, steps_per_replay, replays=60):
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize(dev)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(replays):
            g.replay()
        torch.cuda.synchronize(dev)
        ts.append((time.perf_counter() - t0) / (replays * steps_per_replay))
    return sorted(ts)[1] * 1e6


def workload(name, E, N, B, curriculum, actor, dtype=torch.float32):
    kw = dict(num_bodies=B) if B else {}
    env = BatchedMultiUAVWorld2D(E, num_agents=N, device=dev, seed=0, **kw)
    if curriculum:
        f = lambda a, b, k: a + (b - a) * k / 3
        env.set_curriculum([dict(x_size=f(30.0, 60.0, k), y_size=f(30.0, 60.0, k), collider_radius=1.0, d_sense=f(10.0, 18.0, k),
                                 n_active=max(1, round(f(N / 2, N, k))), b_active=round(f(B / 4, B, k))) for k in range(4)], lo=0, hi=3)
    mem = DeviceReplay(env, horizon=RING - 1)
    mem.begin(env.reset())
    pol = actor().to(dev).to(dtype).eval()          # bfloat16: the matrix cores' native input type (the env stays float32 / float64)
    gen = torch.Generator(device=dev).manual_seed(0)

    def loop_pass():                      # RING steps of the closed loop
        with torch.no_grad():
            for _ in range(RING):
                mem.action_slot().copy_(pol.act(mem.state.to(dtype)))     # (copy_ converts back to the ring's float32)
                mem.step(**STEP_KW)

    def env_pass():                       # the same launches fed by whatever sits in the action slots
        for _ in range(RING):
            mem.step(**STEP_KW)

    obs_rows = torch.rand((E, N, 10), generator=gen, device=dev).to(dtype)
    out_rows = torch.empty((E, N, 2), device=dev)

    def actor_pass():
        with torch.no_grad():
            for _ in range(RING):
                out_rows.copy_(pol.act(obs_rows))

    for _ in range(40):                   # parked layouts in place, caches warm
        loop_pass()
    res = dict(workload=name, envs=E, learners=N, bodies=B, curriculum_levels=4 if curriculum else 0, actor=actor.__name__,
               actor_dtype=str(dtype).replace("torch.", ""),
               actor_rows_per_step=E * N, ring_slots=RING)
    res["loop_us_per_step"] = time_graph(graph_of(loop_pass), RING)
    res["env_launch_us_per_step"] = time_graph(graph_of(env_pass), RING)
    res["actor_us_per_step"] = time_graph(graph_of(actor_pass), RING)
    res["env_share_of_loop"] = res["env_launch_us_per_step"] / res["loop_us_per_step"]
    res["env_steps_per_s_whole_loop"] = E / (res["loop_us_per_step"] * 1e-6)
    res["ended_episodes"] = env.evaluation_summary()
    env.close()
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "closed_loop.json"))
    ap.add_argument("--envs", type=int, default=65536)
    args = ap.parse_args()
    rows = []
    for actor in (GaussianPolicy, TD3Actor, DDPGActor):
        rows.append(workload("BASELINE configs[2]: 65 536 envs x 4 UAVs", args.envs, 4, 0, False, actor))
        print(json.dumps(rows[-1]), flush=True)
    rows.append(workload("BASELINE configs[2]: 65 536 envs x 4 UAVs", args.envs, 4, 0, False, GaussianPolicy, torch.bfloat16))
    print(json.dumps(rows[-1]), flush=True)
    for dt in (torch.float32, torch.bfloat16):
        rows.append(workload("BASELINE configs[4]: 65 536 envs x (8 UAVs + 16 scripted bodies), 4-level curriculum", args.envs, 8, 16, True, GaussianPolicy, dt))
        print(json.dumps(rows[-1]), flush=True)
    doc = dict(what="closed rollout loop (batched actor forward -> replay action slot -> fused uavx_step_ex into the replay ring) captured "
                    "as one hipGraph per ring pass; random-initialised actors of the reference's architectures (no checkpoint ships with "
                    "the reference); the reference prints the same quantity for its host loop as 'Steps Per Sec' (test_sac_multi.py:120-123)",
               device=torch.cuda.get_device_name(0), csrc_sha=_lib.source_hash(), torch=torch.__version__, rows=rows)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(doc, open(args.out, "w"), indent=1)
    print("wrote", args.out)
