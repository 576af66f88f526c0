"""CPU checks of the oracle's restatement of the configs[4] extension (scripted bodies, curriculum levels, ended /
truncated).  There is no reference behaviour to pin these semantics to: the tests check (i) the committed known-answer
fixture the oracle itself produced (tests/golden/make_ext_golden.py), so the definition cannot drift silently, (ii) that the
extension degenerates EXACTLY to the reference-pinned step when nothing of it is switched on, (iii) invariants the
definition in include/uavx.h promises."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def test_extension_known_answers(oracle_mod):
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_ext_golden as g
    fx = np.load(os.path.join(HERE, "golden", "ext_bodies_levels.npz"))
    got = g.run()
    assert set(got) | {"meta"} == set(fx.files)
    for k in got:
        np.testing.assert_array_equal(got[k], fx[k], err_msg=k)
    # the scenario exercises what it is meant to: both levels, parked learners, resets, truncations, terminal ends
    assert set(np.unique(fx["level"])) == {0, 1} and (fx["flags"] & 32).any()
    assert fx["reset_mask"].sum() > 0 and fx["truncated"].sum() > 0 and (fx["ended"] & ~fx["truncated"]).sum() >= 0


def test_trivial_extension_equals_the_pinned_step(oracle_mod):
    kw = dict(num_envs=24, num_agents=5, x_size=30.0, y_size=30.0, d_sense=9.0)
    a, b = oracle_mod.OracleMulti(**kw), oracle_mod.OracleMulti(**kw)
    b.set_curriculum([dict(x_size=30.0, y_size=30.0, collider_radius=1.0, d_sense=9.0)], lo=0, hi=0)
    a.reset_philox(4, env_offset=9); b.reset_philox(4, env_offset=9)
    rng = np.random.default_rng(0)
    for t in range(120):
        act = rng.uniform(-10, 10, size=(24, 5, 2))
        ra = a.step_ex(act, reset_policy=1, step_cap=40, seed=4, env_offset=9, with_end=True)
        rb = b.step_ex(act, reset_policy=1, step_cap=40, seed=4, env_offset=9, with_end=True)
        for x, y in zip(ra, rb):
            np.testing.assert_array_equal(x, y)
    np.testing.assert_array_equal(a.loc, b.loc)
    np.testing.assert_array_equal(a.counters, b.counters)


def test_extension_invariants(oracle_mod):
    L, B, E = 4, 9, 40
    levels = [dict(x_size=18.0, y_size=14.0, collider_radius=0.5, d_sense=7.0, n_active=2, b_active=4),
              dict(x_size=26.0, y_size=26.0, collider_radius=1.0, d_sense=12.0, n_active=4, b_active=9)]
    mk = lambda: oracle_mod.OracleMulti(num_envs=E, num_agents=L, num_bodies=B, body_speed=4.0, body_period=8, body_seed=2)
    o = mk(); o.set_curriculum(levels, lo=0, hi=1); o.reset_philox(3)
    # start points: every pair of slots taking part is more than 2R apart (MUW:127-137 extended to the bodies)
    for e in range(E):
        lv = levels[o.level[e]]
        pts = np.concatenate([o.loc[e, :lv["n_active"]], o.body[e, :lv["b_active"], :2].astype(np.float64)])
        d = np.linalg.norm(pts[:, None] - pts[None], axis=-1) + np.eye(len(pts)) * 1e9
        assert d.min() > 2 * lv["collider_radius"]
        assert np.isinf(o.loc[e, lv["n_active"]:]).all() and np.isinf(o.body[e, lv["b_active"]:, 0]).all()
        assert ((o.flags[e] & 32) != 0).tolist() == [i >= lv["n_active"] for i in range(L)]
    rng = np.random.default_rng(1)
    prev = o.body.copy()
    for t in range(60):
        obs, rew, done, rm, en, tr = o.step_ex(rng.uniform(-1, 1, size=(E, L, 2)), action_mode=1, reset_policy=2, step_cap=25,
                                               seed=3, with_end=True)
        parked = (o.flags & 32) != 0
        stepping = (rm == 0)[:, None] & parked
        assert (obs[parked] == 0).all() and (rew[parked] == 0).all() and done[stepping].all()
        assert (tr <= en).all() and not (en & rm).any()                # truncated implies ended; a reset call ends nothing
        for e in range(E):
            lv = levels[o.level[e]]
            on = o.body[e, :lv["b_active"]]
            assert (np.abs(on[:, 0]) <= lv["x_size"] / 2 + 1e-4).all() and (np.abs(on[:, 1]) <= lv["y_size"] / 2 + 1e-4).all()
            if not rm[e]:   # a body covers speed * tau per step while its leg lasts, and rests afterwards
                mv = np.linalg.norm(on[:, :2] - prev[e, :lv["b_active"], :2], axis=-1)
                assert (mv <= 4.0 * 0.02 + 1e-5).all()
                assert (np.isclose(mv, 4.0 * 0.02, atol=1e-5) | (mv == 0)).all()
            # the record: displacement of length speed * tau (or 0), heading = its direction, a whole number of moving steps
            dlen = np.linalg.norm(on[:, 2:4].astype(np.float64), axis=-1)
            assert (np.isclose(dlen, 4.0 * 0.02, atol=1e-6) | (dlen == 0)).all()
            mvg = dlen > 0
            assert np.allclose(on[mvg, 4], np.arctan2(on[mvg, 3].astype(np.float64), on[mvg, 2].astype(np.float64)), atol=2e-5)   # (float32 direction of a 0.08 m vector)
            assert (on[:, 5] == np.floor(on[:, 5])).all() and (on[:, 5] >= 0).all()
        prev = o.body.copy()
    # determinism and independence of the shard cut (Philox keyed by global env id), bodies and levels included
    whole = mk(); whole.set_curriculum(levels, lo=0, hi=1); whole.reset_philox(3)
    lo_, hi_ = (oracle_mod.OracleMulti(num_envs=n, num_agents=L, num_bodies=B, body_speed=4.0, body_period=8, body_seed=2)
                for n in (17, E - 17))
    for part, off in ((lo_, 0), (hi_, 17)):
        part.set_curriculum(levels, lo=0, hi=1); part.reset_philox(3, env_offset=off)
    acts = rng.uniform(-1, 1, size=(30, E, L, 2))
    for t in range(30):
        whole.step_ex(acts[t], action_mode=1, reset_policy=2, step_cap=12, seed=3)
        lo_.step_ex(acts[t, :17], action_mode=1, reset_policy=2, step_cap=12, seed=3, env_offset=0)
        hi_.step_ex(acts[t, 17:], action_mode=1, reset_policy=2, step_cap=12, seed=3, env_offset=17)
    np.testing.assert_array_equal(whole.body, np.concatenate([lo_.body, hi_.body]))
    np.testing.assert_array_equal(whole.level, np.concatenate([lo_.level, hi_.level]))
    on = (whole.flags & 32) == 0
    np.testing.assert_array_equal(whole.loc[on], np.concatenate([lo_.loc, hi_.loc])[on])


def test_body_trajectories_against_independent_arithmetic(oracle_mod):
    """A hand check that does not come from the oracle: the waypoint stream (Philox4x32-10 written from its published
    definition, Random123 known answers first), the leg record and the positions of every body over three legs, computed with
    numpy float32 operations from the text of include/uavx.h (tests/golden_util.py: body_waypoint / body_leg / body_track).
    Positions, displacements and leg counts must agree bit for bit; the heading (the build's own polynomial) must be the
    direction of the displacement to float32 accuracy."""
    from golden_util import body_track, philox4x32_10
    assert philox4x32_10([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert philox4x32_10([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert philox4x32_10([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]
    L, B, E, period, speed, seed, off = 3, 5, 6, 8, 6.5, 0x1234567890, 7
    xs, ys = 44.0, 36.0
    o = oracle_mod.OracleMulti(num_envs=E, num_agents=L, num_bodies=B, x_size=xs, y_size=ys, body_speed=speed, body_period=period,
                               body_seed=seed)
    o.reset_philox(11, env_offset=off)
    start = o.body[:, :, :2].copy()
    episode = int(o.counters[0, 3]) - 1          # the episode index the reset drew with
    steps = 3 * period + 2
    want = {(e, b): body_track(start[e, b], off + e, L + b, episode, seed, xs, ys, speed, 0.02, period, steps)
            for e in range(E) for b in range(B)}
    for (e, b), (_, legs) in want.items():       # leg 0 is installed by the reset
        _, dx, dy, n, heading, _, _ = legs[0]
        rec = o.body[e, b]
        assert rec[2] == dx and rec[3] == dy and rec[5] == n, (e, b, rec, legs[0])
        assert abs(float(rec[4]) - heading) <= 4e-7 * max(1.0, abs(heading))
    rng = np.random.default_rng(5)
    for t in range(steps):
        o.step(rng.uniform(-3, 3, size=(E, L, 2)), env_offset=off)
        for (e, b), (track, legs) in want.items():
            np.testing.assert_array_equal(o.body[e, b, :2], track[t], err_msg=f"step {t} env {e} body {b}")
            cur = [l for l in legs if l[0] <= t][-1]
            assert o.body[e, b, 2] == cur[1] and o.body[e, b, 3] == cur[2] and o.body[e, b, 5] == cur[3], (t, e, b)
            assert abs(float(o.body[e, b, 4]) - cur[4]) <= 4e-7 * max(1.0, abs(cur[4]))
    assert len(want[(0, 0)][1]) == 4             # legs 0..3 were started
