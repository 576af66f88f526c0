#!/usr/bin/env python3
"""Generates the golden fixtures in tests/golden/*.npz by running the REFERENCE env code.

Run in the build container only (needs /root/reference; see ref_loader.py):
    python tests/golden/make_golden.py
The .npz files hold data only: initial states, the actions fed, and the reference's per-step
observations / rewards / dones / agent state / counters.  numpy version is recorded in `meta`
(NEP 50 promotion affects a few float32 sub-expressions, SURVEY.md §7 hard part 4).
"""
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_loader  # noqa: E402

MUW, UW, _AG = ref_loader.load()


# ---------------------------------------------------------------------------------------------
def snap_multi(env):
    ags = env.agent_list
    pos_dtype = ags[0].location.dtype
    return dict(
        loc=np.array([a.location for a in ags], dtype=np.float64),
        vel=np.array([a.velocity for a in ags], dtype=np.float64),
        tgt=np.array([a.target_location for a in ags], dtype=np.float64),
        init_d=np.array([a.init_distance for a in ags], dtype=np.float64),
        prev_d=np.array([a.prev_distance for a in ags], dtype=np.float64),
        flags=np.array([(1 if a.done else 0) | (2 if a.collided else 0) for a in ags], dtype=np.uint8),
        counters=np.array([env.steps, env.target_reach_count, env.collision_count], dtype=np.uint32),
        f64pos=np.uint8(pos_dtype == np.float64),
    )


def run_multi(env, policy, T, evaluate=None, action_dtype=np.float32):
    """Steps `env` T times with actions from policy(env, t) and records everything."""
    init = snap_multi(env)
    rec = {k: [] for k in ("actions", "obs", "rew", "done", "loc", "vel", "prev_d", "flags", "counters", "evaluate")}
    for t in range(T):
        acts = [np.asarray(a, dtype=action_dtype) for a in policy(env, t)]
        ev = bool(evaluate[t]) if evaluate is not None else False
        obs, rew, done, _info = env.step(acts, evaluate=ev)
        s = snap_multi(env)
        rec["actions"].append(np.array(acts, dtype=action_dtype))
        rec["obs"].append(np.array(obs, dtype=np.float64))
        rec["rew"].append(np.array([float(r) for r in rew], dtype=np.float64))
        rec["done"].append(np.array([bool(d) for d in done], dtype=np.uint8))
        for k in ("loc", "vel", "prev_d", "flags", "counters"):
            rec[k].append(s[k])
        rec["evaluate"].append(np.uint8(ev))
    out = {"init_" + k: v for k, v in init.items()}
    out.update({k: np.array(v) for k, v in rec.items()})
    return out


def cfg_of(env):
    return dict(x_size=float(env.x_size), y_size=float(env.y_size), max_speed=float(env.max_speed[0]),
                max_acceleration=float(env.max_acceleratoin[0]), num_agents=int(env.num_agents),
                collider_radius=float(env.collider_radius), d_sense=float(env.d_sense), tau=float(env.tau))


def save(name, data, meta):
    meta = dict(meta)
    meta["numpy"] = np.__version__
    meta["generator"] = "tests/golden/make_golden.py"
    data = dict(data)
    data["meta"] = np.array(json.dumps(meta))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **data)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


# ---------------------------------------------------------------------------------------------
# policies
def random_box_policy(rng):
    """action_space.sample()-like: float32 uniform in [-max_speed, max_speed]^2 (run_multi.py:13)."""
    def pol(env, t):
        hi = float(env.max_speed[0])
        return [rng.uniform(-hi, hi, size=2).astype(np.float32) for _ in range(env.num_agents)]
    return pol


def polar_policy(rng):
    """Trainer-style conversion of a ~ U(-1,1)^2 (test_sac_multi.py:73,77-80), float64 actions."""
    def pol(env, t):
        out = []
        for _ in range(env.num_agents):
            a = rng.uniform(-1, 1, size=2)
            v = (a[0] / 2 + 0.5) * np.linalg.norm(env.action_space.high)
            th = a[1] * math.pi
            out.append(np.array([v * math.cos(th), v * math.sin(th)]))
        return out
    return pol


def seek_policy(rng, noise=0.3, vcap=8.0, brake=2.0):
    """Flies every agent at its target and brakes so that the success test (MUW:218) fires."""
    def pol(env, t):
        out = []
        for a in env.agent_list:
            d = np.asarray(a.target_location, dtype=np.float64) - np.asarray(a.location, dtype=np.float64)
            dist = float(np.linalg.norm(d))
            if dist < 1e-9:
                out.append(np.zeros(2))
                continue
            sp = min(vcap, math.sqrt(2 * brake * dist)) if dist > 0.3 else 0.0
            out.append(d / dist * sp + rng.normal(0, noise, size=2) * (dist > 3.0))
        return out
    return pol


# ---------------------------------------------------------------------------------------------
def gen_random_rollouts():
    for n, seed, T, pol_kind in ((1, 0, 256, "box"), (2, 1, 256, "box"), (4, 2, 384, "box"), (4, 3, 384, "polar"),
                                 (5, 4, 256, "box"), (8, 5, 256, "polar"), (10, 6, 128, "polar"),
                                 (24, 7, 64, "box")):
        np.random.seed(1000 + seed)
        env = MUW(num_agents=n)
        env.reset()
        rng = np.random.default_rng(seed)
        if pol_kind == "box":
            data = run_multi(env, random_box_policy(rng), T, action_dtype=np.float32)
        else:
            data = run_multi(env, polar_policy(rng), T, action_dtype=np.float64)
        save(f"multi_random_n{n}_s{seed}", data, dict(kind="multi", cfg=cfg_of(env), np_seed=1000 + seed,
                                                      policy=pol_kind))


def gen_seek_rollouts():
    for n, seed, T, ev in ((4, 10, 900, False), (4, 11, 900, True), (8, 12, 700, False), (2, 13, 700, False)):
        np.random.seed(2000 + seed)
        env = MUW(num_agents=n)
        env.reset()
        rng = np.random.default_rng(seed)
        evaluate = np.full(T, ev)
        data = run_multi(env, seek_policy(rng), T, evaluate=evaluate, action_dtype=np.float64)
        save(f"multi_seek_n{n}_s{seed}", data, dict(kind="multi", cfg=cfg_of(env), np_seed=2000 + seed,
                                                    policy="seek", reach=int(env.target_reach_count),
                                                    coll=int(env.collision_count)))
        print("   reach", env.target_reach_count, "coll", env.collision_count)


def gen_dense_rollouts():
    """Small box, many agents: frequent (hard) collisions, OOB, ties are plausible."""
    for n, seed, T in ((6, 20, 300), (4, 21, 300)):
        np.random.seed(3000 + seed)
        env = MUW(x_size=12.0, y_size=12.0, num_agents=n, d_sense=6)
        env.reset()
        rng = np.random.default_rng(seed)
        evaluate = rng.random(T) < 0.5
        data = run_multi(env, random_box_policy(rng), T, evaluate=evaluate)
        save(f"multi_dense_n{n}_s{seed}", data, dict(kind="multi", cfg=cfg_of(env), np_seed=3000 + seed,
                                                     policy="box", coll=int(env.collision_count)))
        print("   coll", env.collision_count)


def place(env, locs, tgts, vels=None):
    """Pokes agent state the way test_sac_multi_plot_trajectory.py:43-49 does (float32 positions)."""
    for i, a in enumerate(env.agent_list):
        a.location = np.array(locs[i], dtype=np.float32)
        a.target_location = np.array(tgts[i], dtype=np.float32)
        a.velocity = np.array(vels[i] if vels is not None else [0.0, 0.0], dtype=np.float64)
        a.velocity_prev = a.velocity
        a.init_distance = np.linalg.norm(a.target_location - a.location)
        a.prev_distance = a.init_distance
        a.done = False
        a.collided = False
    env.steps = 0
    env.target_reach_count = 0
    env.collision_count = 0


def const_policy(actions):
    return lambda env, t: [np.asarray(a) for a in actions]


def gen_crafted():
    np.random.seed(4000)
    # (1) Gauss-Seidel ordering: agent 0 evaluates against agent 1's OLD position (MUW:181-231)
    env = MUW(num_agents=2); env.reset()
    place(env, [[0.0, 0.0], [2.003, 0.0]], [[10.0, 0.0], [-10.0, 0.0]])
    save("crafted_gauss_seidel", run_multi(env, const_policy([[10.0, 0.0], [-10.0, 0.0]]), 40),
         dict(kind="multi", cfg=cfg_of(env), note="agents approach head-on; order-dependent -2 rewards"))
    # (2) done agent keeps +10/step, then gets crowded (-2, done False) (MUW:203-205,218-227)
    env = MUW(num_agents=2); env.reset()
    place(env, [[0.0, 0.0], [8.0, 0.0]], [[0.2, 0.0], [-20.0, 0.0]])
    save("crafted_done_sticky", run_multi(env, const_policy([[0.0, 0.0], [-10.0, 0.0]]), 120),
         dict(kind="multi", cfg=cfg_of(env), note="agent0 finishes at once (v=0 -> NaN -> zeros); agent1 flies through it"))
    # (3) OOB with evaluate toggling (MUW:224-225)
    env = MUW(num_agents=2); env.reset()
    place(env, [[24.5, 0.0], [-24.9, -24.9]], [[0.0, 0.0], [0.0, 0.0]], vels=[[9.0, 0.0], [-9.0, -9.0]])
    ev = np.array([0, 0, 0, 1, 1, 0, 1, 0] * 5, dtype=bool)
    save("crafted_oob_evaluate", run_multi(env, const_policy([[10.0, 0.0], [-10.0, -10.0]]), 40, evaluate=ev),
         dict(kind="multi", cfg=cfg_of(env), note="agents leave the box; evaluate flag alternates"))
    # (4) hard collision counted once per agent (MUW:207-210)
    env = MUW(num_agents=3); env.reset()
    place(env, [[-3.0, 0.0], [3.0, 0.0], [0.0, 20.0]], [[20.0, 0.0], [-20.0, 0.0], [0.0, -20.0]])
    save("crafted_hard_collision", run_multi(env, const_policy([[6.0, 0.0], [-6.0, 0.0], [0.0, 0.0]]), 120),
         dict(kind="multi", cfg=cfg_of(env), note="pair passes through each other; collision_count == 2"))
    # (5) no neighbours in range -> defaults [1, +-1, 0] (MUW:77-95), d_sense boundary
    env = MUW(num_agents=3); env.reset()
    place(env, [[-20.0, -20.0], [20.0, 20.0], [-20.0, 20.0]], [[-10.0, -20.0], [10.0, 20.0], [-20.0, 10.0]])
    rng = np.random.default_rng(5)
    save("crafted_no_neighbours", run_multi(env, random_box_policy(rng), 60),
         dict(kind="multi", cfg=cfg_of(env), note="all pair distances > d_sense"))
    # (6) velocity saturation at [10,10] -> speed feature 1.0 (AG:27, MUW:62)
    env = MUW(num_agents=1, x_size=400.0, y_size=400.0); env.reset()
    place(env, [[-150.0, -150.0]], [[150.0, 150.0]])
    save("crafted_saturation", run_multi(env, const_policy([[10.0, 10.0]]), 220),
         dict(kind="multi", cfg=cfg_of(env), note="accelerates to the per-axis speed limit"))
    # (7) exact ties in neighbour distance -> lower index first (AG:62 stable argsort on small arrays)
    env = MUW(num_agents=5); env.reset()
    place(env, [[0.0, 0.0], [4.0, 0.0], [-4.0, 0.0], [0.0, 4.0], [0.0, -4.0]],
          [[0.0, 10.0], [14.0, 0.0], [-14.0, 0.0], [0.0, 14.0], [0.0, -14.0]])
    save("crafted_ties", run_multi(env, const_policy([[0.0, 0.0]] * 5), 6),
         dict(kind="multi", cfg=cfg_of(env), note="four agents equidistant from agent 0; nobody moves"))
    # (8) circular reset: float64 positions (MUW:157-163)
    for n in (4, 6):
        env = MUW(num_agents=n)
        np.random.seed(4100 + n)
        env.reset(circular=True)
        rng = np.random.default_rng(n)
        save(f"crafted_circular_n{n}", run_multi(env, seek_policy(rng, noise=0.0), 500, action_dtype=np.float64),
             dict(kind="multi", cfg=cfg_of(env), note="reset(circular=True): float64 location/target arrays",
                  np_seed=4100 + n))
    # (9) non-default parameters
    env = MUW(x_size=80.0, y_size=40.0, max_speed=6.0, max_acceleration=3.0, num_agents=3,
              collider_radius=0.7, d_sense=11.5)
    np.random.seed(4200); env.reset()
    rng = np.random.default_rng(9)
    save("crafted_params", run_multi(env, random_box_policy(rng), 200),
         dict(kind="multi", cfg=cfg_of(env), note="non-default ctor args; 2R=1.4 and d_sense not f32-exact",
              np_seed=4200))


def gen_resets():
    """Initial states + observations of reset() under np.random.seed, several resets per stream."""
    out = {}
    specs = []
    for k, (n, seed, kwargs) in enumerate(((1, 11, {}), (4, 12, {}), (8, 13, {}), (24, 14, {}),
                                           (16, 15, dict(x_size=14.0, y_size=14.0)))):
        np.random.seed(seed)
        env = MUW(num_agents=n, **kwargs)
        for r in range(3):
            obs = env.reset()
            s = snap_multi(env)
            for key in ("loc", "tgt", "init_d", "prev_d"):
                out[f"r{k}_{r}_{key}"] = s[key]
            out[f"r{k}_{r}_obs"] = np.array(obs, dtype=np.float64)
        specs.append(dict(cfg=cfg_of(env), np_seed=seed, resets=3))
    save("multi_resets", out, dict(kind="multi_resets", specs=specs))


def snap_uw(env):
    return dict(loc=np.array(env._agent_location, dtype=np.float64), vel=np.array(env._agent_speed, dtype=np.float64),
                tgt=np.array(env._target_location, dtype=np.float64), init_d=np.float64(env._init_target_distance),
                prev_d=np.float64(env._prev_distance), steps=np.uint32(env.steps),
                vel_f32=np.uint8(env._agent_speed.dtype == np.float32))


def gen_uw():
    for seed, T, kind in ((30, 400, "box32"), (31, 400, "box64"), (32, 900, "seek")):
        np.random.seed(5000 + seed)
        env = UW()
        obs0 = env.reset()
        init = snap_uw(env)
        rng = np.random.default_rng(seed)
        rec = {k: [] for k in ("actions", "obs", "rew", "done", "info", "loc", "vel", "prev_d", "steps")}
        dt = np.float32 if kind == "box32" else np.float64
        for t in range(T):
            if kind == "seek":
                d = env._target_location.astype(np.float64) - env._agent_location.astype(np.float64)
                dist = np.linalg.norm(d)
                a = d / max(dist, 1e-9) * min(9.0, math.sqrt(2 * 2.0 * dist))
            else:
                a = rng.uniform(-12, 12, size=2)
            a = np.asarray(a, dtype=dt)
            obs, r, done, info = env.step(a)
            s = snap_uw(env)
            rec["actions"].append(a); rec["obs"].append(np.array(obs, dtype=np.float64))
            rec["rew"].append(np.float64(r)); rec["done"].append(np.uint8(done)); rec["info"].append(np.float64(info["distance"]))
            for k in ("loc", "vel", "prev_d", "steps"):
                rec[k].append(s[k])
        data = {"init_" + k: v for k, v in init.items()}
        data["init_obs"] = np.array(obs0, dtype=np.float64)
        data.update({k: np.array(v) for k, v in rec.items()})
        save(f"uw_{kind}_s{seed}", data, dict(kind="uw", np_seed=5000 + seed, policy=kind,
                                              cfg=dict(x_size=100.0, y_size=100.0, max_speed=12.0,
                                                       max_acceleration=5.0, tau=0.02)))
    # resets from seed
    np.random.seed(77)
    env = UW()
    out = {}
    for r in range(4):
        obs = env.reset()
        s = snap_uw(env)
        for k in ("loc", "vel", "tgt", "init_d"):
            out[f"r{r}_{k}"] = s[k]
        out[f"r{r}_obs"] = np.array(obs, dtype=np.float64)
    save("uw_resets", out, dict(kind="uw_resets", np_seed=77, resets=4))


if __name__ == "__main__":
    gen_random_rollouts()
    gen_seek_rollouts()
    gen_dense_rollouts()
    gen_crafted()
    gen_resets()
    gen_uw()
