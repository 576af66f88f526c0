"""Live cross-check of the oracle against the REFERENCE ITSELF, beyond the committed fixtures: fresh seeds,
agent counts and policies every run.  Only possible where /root/reference is mounted (the build container);
skipped on the GPU box, where the committed fixtures (tests/golden/*.npz) carry the pinning."""
import math
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import ref_loader  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_loader.available(), reason="reference tree not mounted")


def _snap(env):
    ags = env.agent_list
    return (np.array([a.location for a in ags], np.float64), np.array([a.velocity for a in ags], np.float64),
            np.array([a.prev_distance for a in ags], np.float64),
            np.array([(1 if a.done else 0) | (2 if a.collided else 0) for a in ags], np.uint8),
            np.array([env.steps, env.target_reach_count, env.collision_count], np.uint32))


@pytest.mark.parametrize("n,seed", [(1, 101), (2, 102), (3, 103), (4, 104), (4, 105), (7, 106), (12, 107), (20, 108)])
def test_oracle_equals_live_reference(oracle_mod, n, seed):
    MUW, _, _ = ref_loader.load()
    np.random.seed(seed)
    kw = dict(num_agents=n) if seed % 2 else dict(num_agents=n, x_size=30.0, y_size=36.0, d_sense=9, collider_radius=0.8)
    env = MUW(**kw)
    orc = oracle_mod.OracleMulti(num_envs=1, **kw)
    g = oracle_mod.MTStream(seed)
    rng = np.random.default_rng(seed)
    for episode in range(3):
        obs_ref = env.reset()
        orc.reset_mt(g)
        np.testing.assert_array_equal(orc.observe()[0], np.array(obs_ref))
        for t in range(150):
            if t % 3 == 0:
                acts = [rng.uniform(-10, 10, 2).astype(np.float32) for _ in range(n)]
            else:  # goal seeking, float64 like the trainers' converted actions
                acts = []
                for a in env.agent_list:
                    d = np.asarray(a.target_location, np.float64) - np.asarray(a.location, np.float64)
                    dist = float(np.linalg.norm(d))
                    acts.append(d / max(dist, 1e-9) * (min(8.0, math.sqrt(4 * dist)) if dist > 0.3 else 0.0))
            ev = bool(t % 7 == 0)
            obs, rew, done, _ = env.step(acts, evaluate=ev)
            o_obs, o_rew, o_done = orc.step(np.array(acts, np.float64), evaluate=ev)
            loc, vel, pd, flags, cnt = _snap(env)
            ctx = f"n={n} seed={seed} ep={episode} t={t}"
            np.testing.assert_array_equal(o_done[0], np.array(done, np.uint8), err_msg=ctx)
            np.testing.assert_array_equal(orc.loc[0], loc, err_msg=ctx)
            np.testing.assert_array_equal(orc.vel[0], vel, err_msg=ctx)
            np.testing.assert_array_equal(orc.prev_d[0], pd, err_msg=ctx)
            np.testing.assert_array_equal(orc.flags[0], flags, err_msg=ctx)
            np.testing.assert_array_equal(orc.counters[0, :3], cnt, err_msg=ctx)
            np.testing.assert_array_equal(o_rew[0], np.array([float(r) for r in rew]), err_msg=ctx)
            np.testing.assert_array_equal(o_obs[0], np.array(obs), err_msg=ctx)


@pytest.mark.parametrize("seed", [201, 202])
def test_oracle_uw_equals_live_reference(oracle_mod, seed):
    _, UW, _ = ref_loader.load()
    np.random.seed(seed)
    env = UW()
    orc = oracle_mod.OracleSingle(num_envs=1)
    g = oracle_mod.MTStream(seed)
    rng = np.random.default_rng(seed)
    for episode in range(3):
        obs_ref = env.reset()
        orc.reset_mt(g)
        np.testing.assert_array_equal(orc.observe()[0], obs_ref)
        for t in range(200):
            a = rng.uniform(-12, 12, 2).astype(np.float32) if (t + seed) % 2 else rng.uniform(-12, 12, 2)
            obs, rew, done, info = env.step(a)
            o_obs, o_rew, o_done, o_info = orc.step(a)
            assert bool(o_done[0]) == bool(done) and o_rew[0] == float(rew) and o_info[0] == float(info["distance"])
            np.testing.assert_array_equal(o_obs[0], obs)
            np.testing.assert_array_equal(orc.loc[0], np.asarray(env._agent_location, np.float64))
            np.testing.assert_array_equal(orc.vel[0], np.asarray(env._agent_speed, np.float64))


def test_committed_fixture_regenerates_identically():
    """The committed .npz really is what the reference produces here (one fixture re-generated in memory)."""
    from golden_util import load_fixture
    data, meta = load_fixture("multi_random_n4_s2")
    MUW, _, _ = ref_loader.load()
    np.random.seed(meta["np_seed"])
    env = MUW(num_agents=4)
    env.reset()
    rng = np.random.default_rng(2)
    for t in range(60):
        acts = [rng.uniform(-10, 10, size=2).astype(np.float32) for _ in range(4)]
        np.testing.assert_array_equal(np.array(acts), data["actions"][t])
        obs, rew, done, _ = env.step(acts)
        np.testing.assert_array_equal(np.array(obs), data["obs"][t])
        np.testing.assert_array_equal(np.array([float(r) for r in rew]), data["rew"][t])
