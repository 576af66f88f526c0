#!/usr/bin/env python3
"""Soak test (not part of the default suite): long closed-loop rollouts with auto-reset, device vs oracle,
state / masks / reset streams compared bit-for-bit every step, observations and rewards every 16th step.

    python tools/soak.py [steps] [envs] [n1,n2,...]
    python tools/soak.py [steps] [envs] ext      configs[4] extension: scripted bodies + randomized-reset curriculum + truncated flags
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from golden_util import obs_err  # noqa: E402
from gym_uav_collision_avoidance_amd import BatchedMultiUAVWorld2D  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
E = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
CASES = ((4, {}), (8, dict(x_size=40.0, y_size=40.0)), (10, {}), (3, dict(x_size=16.0, y_size=16.0, d_sense=6.0)),
         (24, dict(x_size=70.0, y_size=70.0)))
EXT = len(sys.argv) > 3 and sys.argv[3] == "ext"
if EXT:   # (learners, world + body kwargs): bodies and a three-level curriculum on top of the same closed loop
    CASES = ((8, dict(num_bodies=16, body_period=32, body_seed=7)), (4, dict(num_bodies=5, body_period=16, body_speed=7.0)),
             (6, dict()), (12, dict(num_bodies=30, body_period=64)))
elif len(sys.argv) > 3:   # e.g. "1,2,5,16,33,64": other agent counts, box scaled with sqrt(n)
    CASES = tuple((int(k), dict(x_size=12.0 * int(k) ** 0.5 + 8, y_size=12.0 * int(k) ** 0.5 + 8)) for k in sys.argv[3].split(","))
for n, kw in CASES:
    Ecur = E if n <= 10 else E // 4
    env = BatchedMultiUAVWorld2D(Ecur, num_agents=n, seed=100 + n, **kw)
    orc = oracle.OracleMulti(num_envs=Ecur, num_agents=n, nthreads=16, **kw)
    if EXT:
        nb = kw.get("num_bodies", 0)
        levels = [dict(x_size=22.0, y_size=22.0, collider_radius=0.4, d_sense=8.0, n_active=max(1, n // 3), b_active=nb // 3),
                  dict(x_size=34.0, y_size=30.0, collider_radius=0.7, d_sense=11.0, n_active=max(1, n // 2), b_active=nb // 2),
                  dict(x_size=46.0, y_size=46.0, collider_radius=1.0, d_sense=15.0, n_active=n, b_active=nb)]
        env.set_curriculum(levels, lo=0, hi=2); orc.set_curriculum(levels, lo=0, hi=2)
    env.reset(); orc.reset_philox(100 + n)
    rng = np.random.default_rng(n)
    t0 = time.time()
    worst_o = worst_r = 0.0
    resets = 0
    for t in range(steps):
        d = orc.tgt - orc.loc
        d = np.where(np.isfinite(d), d, 0.0)                      # parked learners (extension) sit at +inf
        dist = np.linalg.norm(d, axis=-1, keepdims=True)
        act = d / np.maximum(dist, 1e-9) * np.where(dist > 0.3, np.minimum(8.0, np.sqrt(4.0 * dist)), 0.0)
        noisy = rng.random((Ecur, n, 1)) < 0.15
        act = np.where(noisy, rng.uniform(-10, 10, size=act.shape), act)
        policy = "all_done" if (t // 500) % 2 else "agent0_done"
        og, rg, dg, info = env.step_ex(act, evaluate=(policy == "all_done"), auto_reset=policy, step_cap=700)
        oo, ro, do, rm, en, tr = orc.step_ex(act, evaluate=(policy == "all_done"), reset_policy=1 if policy == "agent0_done" else 2,
                                             step_cap=700, track_returns=True, seed=100 + n, with_end=True)
        resets += int(rm.sum())
        assert np.array_equal(info["ended"].cpu().numpy().astype(np.uint8), en), (n, t)
        assert np.array_equal(info["truncated"].cpu().numpy().astype(np.uint8), tr), (n, t)
        if EXT:
            assert np.array_equal(env.env_levels().cpu().numpy(), orc.level), (n, t)
            if orc.B:
                assert np.array_equal(env.get_bodies().cpu().numpy(), orc.body), (n, t)
        assert np.array_equal(info["reset_mask"].cpu().numpy().astype(np.uint8), rm), (n, t)
        assert np.array_equal(dg.cpu().numpy().astype(np.uint8), do), (n, t)
        st = env.get_state(); ref = orc.get_state()
        on = (ref["flags"] & 32) == 0
        assert np.array_equal(st["flags"].cpu().numpy(), ref["flags"]), (n, t)
        for k in ("loc", "vel", "tgt", "prev_d"):
            assert np.array_equal(st[k].cpu().numpy()[on], ref[k][on]), (n, t, k)
        assert np.array_equal(st["counters"].cpu().numpy(), ref["counters"].astype(np.int32)), (n, t)
        if t % 16 == 0:
            worst_o = max(worst_o, obs_err(og.cpu().numpy(), oo))
            worst_r = max(worst_r, float((np.abs(rg.cpu().numpy() - ro) / np.maximum(1.0, np.abs(ro))).max()))
            assert worst_o <= 1e-5 and worst_r <= 1e-5, (n, t, worst_o, worst_r)
    s = env.evaluation_summary()
    print(f"N={n:2d} E={Ecur}: {steps} steps OK in {time.time() - t0:.1f} s; resets {resets}; reach {int(orc.fin_counts[:, 2].sum())} "
          f"hard collisions {int(orc.fin_counts[:, 3].sum())}; worst obs err {worst_o:.2e} rew err {worst_r:.2e}; "
          f"SR {s['success_rate']:.3f} CR {s['collision_rate']:.3f}", flush=True)
    assert np.array_equal(env.episode_stats()["episodes"].cpu().numpy(), orc.fin_counts[:, 0])
    env.close()
print("soak passed")
