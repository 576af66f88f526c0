"""GPU tests of the BASELINE.json configs[4] extension: scripted bodies stepped inside the kernel, per-env curriculum
levels taken at (auto-)reset, and the ended / truncated outputs of uavx_step_ex.

The reference has no scripted obstacle and no per-env worlds: these semantics are the build's own
(include/uavx.h) and PARITY IS UNPINNED BY THE REFERENCE.  What is checked here is the HIP path against the oracle's
restatement of the same definition (oracle/uavx_oracle.c: uavo_*_x) -- bit-exact on masks, learner state, body records,
levels and counters, 1e-5 on observations / rewards -- with the learners' own step still the reference-pinned one (an
extension handle with no bodies and one level equal to the config must reproduce the plain kernels bit for bit)."""
import ctypes

import os

import numpy as np
import pytest

from golden_util import obs_err

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def amd():
    import torch
    assert torch.cuda.is_available()
    import gym_uav_collision_avoidance_amd as pkg
    return pkg


def _np(t):
    return t.detach().cpu().numpy()


def _seek_actions(orc, rng, noise=0.3):
    d = orc.tgt - orc.loc
    d = np.where(np.isfinite(d), d, 0.0)
    dist = np.linalg.norm(d, axis=-1, keepdims=True)
    act = d / np.maximum(dist, 1e-9) * np.where(dist > 0.3, np.minimum(8.0, np.sqrt(4.0 * dist)), 0.0)
    return act + rng.normal(0, noise, size=act.shape) * (dist > 2.0)


def _compare_state(env, orc, ctx):
    st, ref = env.get_state(), orc.get_state()
    on = (ref["flags"] & 32) == 0          # parked learners hold +inf / unspecified values: compare flags only
    np.testing.assert_array_equal(_np(st["flags"]), ref["flags"], err_msg=f"{ctx} flags")
    for key in ("loc", "vel", "tgt", "init_d", "prev_d"):
        a, b = _np(st[key]), ref[key]
        np.testing.assert_array_equal(a[on], b[on], err_msg=f"{ctx} {key}")
    np.testing.assert_array_equal(_np(st["counters"]), ref["counters"].astype(np.int32), err_msg=ctx)
    if orc.B:
        np.testing.assert_array_equal(_np(env.get_bodies()), orc.body, err_msg=f"{ctx} bodies")


@pytest.mark.parametrize("L,B,E,period", [(8, 16, 1000, 16), (4, 3, 777, 4), (1, 6, 300, 8), (5, 20, 257, 32), (24, 40, 65, 16)])
def test_bodies_step_vs_oracle(amd, oracle_mod, L, B, E, period):
    """Plain uavx_step on worlds with scripted bodies: learners' masks / state, body records and waypoint re-targets
    against the oracle (dense boxes so that bodies are sensed, collided with and re-target many times)."""
    import torch
    kw = dict(x_size=30.0, y_size=24.0, num_agents=L, d_sense=9.0, num_bodies=B, body_speed=6.0, body_period=period, body_seed=99)
    env = amd.BatchedMultiUAVWorld2D(E, seed=5, env_offset=11, **kw)
    orc = oracle_mod.OracleMulti(num_envs=E, nthreads=8, **kw)
    obs = env.reset()
    orc.reset_philox(5, env_offset=11)
    _compare_state(env, orc, "reset")
    assert obs_err(_np(obs), orc.observe()) <= TOL
    rng = np.random.default_rng(L * 100 + B)
    hits = 0
    for t in range(3 * period + 40):
        act = _seek_actions(orc, rng) if t % 3 else rng.uniform(-10, 10, size=(E, L, 2))
        o_g, r_g, d_g, _ = env.step(torch.from_numpy(act).to(env.device), evaluate=bool(t % 2))
        o_o, r_o, d_o = orc.step(act, evaluate=bool(t % 2), env_offset=11)
        ctx = f"L{L} B{B} step {t}"
        np.testing.assert_array_equal(_np(d_g).astype(np.uint8), d_o, err_msg=ctx)
        _compare_state(env, orc, ctx)
        assert obs_err(_np(o_g), o_o) <= TOL, ctx
        assert float(np.abs(_np(r_g) - r_o).max()) <= TOL, ctx
        hits += int((r_o == -2).sum())
    assert hits > 0, "scenario must produce learner/body proximity events"
    assert np.isfinite(orc.body).all() and (np.abs(orc.body[..., 0]) <= 15.0).all() and (np.abs(orc.body[..., 1]) <= 12.0).all()
    # observe() (no motion) sees the same bodies
    assert obs_err(_np(env.observe()), orc.observe()) <= TOL
    env.close()


LEVELS = [dict(x_size=24.0, y_size=24.0, collider_radius=0.5, d_sense=8.0, n_active=2, b_active=4),
          dict(x_size=32.0, y_size=28.0, collider_radius=0.8, d_sense=12.0, n_active=5, b_active=9),
          dict(x_size=40.0, y_size=40.0, collider_radius=1.0, d_sense=15.0, n_active=8, b_active=16)]


@pytest.mark.parametrize("prefetch", [16, 1, 0])
def test_config5_combined_vs_oracle(amd, oracle_mod, prefetch):
    """BASELINE configs[4] put together at a size the oracle follows: 8 learners + 16 scripted bodies, randomized-reset
    curriculum (per-env box / d_sense / collider / active counts drawn at every auto-reset), uavx_step_ex with polar
    actions, all-done auto-reset and a step cap, observations written zero-copy into DeviceReplay."""
    import torch
    from gym_uav_collision_avoidance_amd.replay import DeviceReplay
    E, L, B, cap = 1024, 8, 16, 140
    small = [dict(x_size=14.0, y_size=14.0, collider_radius=0.3, d_sense=6.0, n_active=2, b_active=3),
             dict(x_size=18.0, y_size=16.0, collider_radius=0.4, d_sense=8.0, n_active=4, b_active=8),
             dict(x_size=24.0, y_size=24.0, collider_radius=0.5, d_sense=10.0, n_active=8, b_active=16)]
    kw = dict(num_agents=L, num_bodies=B, body_speed=2.0, body_period=16, body_seed=3)
    env = amd.BatchedMultiUAVWorld2D(E, seed=21, env_offset=7, **kw)
    orc = oracle_mod.OracleMulti(num_envs=E, nthreads=8, **kw)
    env.set_prefetch(prefetch)   # layouts drawn ahead by staging workgroups (1/16 or all envs per launch) vs. in place: same results
    env.set_curriculum(small, lo=0, hi=1)
    orc.set_curriculum(small, lo=0, hi=1)
    mem = DeviceReplay(env, horizon=24)
    mem.begin(env.reset())
    orc.reset_philox(21, env_offset=7)
    np.testing.assert_array_equal(_np(env.env_levels()), orc.level)
    assert set(np.unique(orc.level)) == {0, 1}
    rng = np.random.default_rng(8)
    n_reset = n_trunc = n_term = 0
    for t in range(330):
        if t == 150:   # the curriculum window moves on
            env.set_level_window(1, 2)
            orc.set_level_window(1, 2)
        a = rng.uniform(-1, 1, size=(E, L, 2)).astype(np.float32)
        if t % 4:  # mostly goal seeking (so that episodes also END by all_done), expressed as policy outputs
            act = _seek_actions(orc, rng, noise=0.1)
            v = np.linalg.norm(act, axis=-1)
            a[..., 0] = np.clip(v / np.sqrt(200.0) * 2 - 1, -1, 1)
            a[..., 1] = np.arctan2(act[..., 1], act[..., 0]) / np.pi
        obs_g, rew_g, done_g, info = mem.step(torch.from_numpy(a).to(env.device), evaluate=True, polar=True,
                                              auto_reset="all_done", step_cap=cap, track_returns=True)
        obs_o, rew_o, done_o, rm_o, en_o, tr_o = orc.step_ex(a, evaluate=True, action_mode=1, reset_policy=2, step_cap=cap,
                                                             track_returns=True, seed=21, env_offset=7, with_end=True)
        ctx = f"step {t}"
        np.testing.assert_array_equal(_np(info["reset_mask"]).astype(np.uint8), rm_o, err_msg=ctx)
        np.testing.assert_array_equal(_np(info["ended"]).astype(np.uint8), en_o, err_msg=ctx)
        np.testing.assert_array_equal(_np(info["truncated"]).astype(np.uint8), tr_o, err_msg=ctx)
        np.testing.assert_array_equal(_np(done_g).astype(np.uint8), done_o, err_msg=ctx)
        np.testing.assert_array_equal(_np(env.env_levels()), orc.level, err_msg=ctx)
        _compare_state(env, orc, ctx)
        assert obs_err(_np(obs_g), obs_o) <= TOL and float(np.abs(_np(rew_g) - rew_o).max()) <= TOL, ctx
        assert obs_g.data_ptr() == mem.obs[(t + 1) % 25].data_ptr()           # zero-copy: the kernel wrote into the ring
        n_reset += int(rm_o.sum()); n_trunc += int(tr_o.sum()); n_term += int((en_o & ~tr_o).sum())
    assert n_reset > E and n_trunc > 0 and n_term > 0, (n_reset, n_trunc, n_term)
    stats = {k: _np(v) for k, v in env.episode_stats().items()}
    np.testing.assert_array_equal(stats["episodes"], orc.fin_counts[:, 0])
    np.testing.assert_array_equal(stats["reach"], orc.fin_counts[:, 2])
    np.testing.assert_array_equal(stats["coll"], orc.fin_counts[:, 3])
    # replay: reset rows never sampled; truncated transitions keep mask 1 unless the learner itself was done
    S, A, R, S1, M, TR, EN = mem.sample(8192, generator=torch.Generator(device=env.device).manual_seed(1), with_flags=True)
    assert S.shape == (8192, 10) and bool(((M == 0) | (M == 1)).all()) and bool((EN | ~TR).all())
    env.close()


@pytest.mark.parametrize("cap,L,B,behind", [(1, 8, 16, 0), (1, 8, 16, 1), (2, 8, 16, 1), (2, 3, 5, 0), (2, 3, 5, 1), (1, 4, 0, 0), (1, 4, 0, 1),
                                            (2, 24, 0, 0), (2, 13, 0, 1)])
def test_staging_overlaps_the_reset_it_serves(amd, oracle_mod, monkeypatch, cap, L, B, behind):
    """Episodes of one and two steps with staging workgroups in EVERY launch (prefetch every = 1): each launch re-initialises
    half (or a third) of the envs from the parked layouts while staging workgroups of the same launch scan, and redraw, the
    layouts of those very envs -- from in front of the env-workgroups and from behind them (the library picks by launch shape;
    UAVX_STAGE_BEHIND, read when the handle is made, forces either).  What keeps that correct is written down at stage_ahead
    (csrc/uavx_multi.hip): the slot a re-initialising env reads is left alone while its "ended" mark stands, and the mark is
    cleared by the env's last store, after every load from the staging arrays has returned.  Every output and the whole state
    against the oracle, every step; a seed change in the middle invalidates everything that is parked."""
    import torch
    monkeypatch.setenv("UAVX_STAGE_BEHIND", str(behind))
    E = 4096 if L <= 8 else 1024        # (24 and 13 learners with levels: the extension kernels on 3- and 4-wavefront workgroups)
    kw = dict(num_agents=L, num_bodies=B, body_period=4, x_size=26.0, y_size=22.0, d_sense=9.0, collider_radius=0.6)
    levels = [dict(x_size=20.0, y_size=18.0, collider_radius=0.5, d_sense=8.0, n_active=max(1, L // 2), b_active=B // 2),
              dict(x_size=26.0, y_size=22.0, collider_radius=0.6, d_sense=9.0, n_active=L, b_active=B)]
    env = amd.BatchedMultiUAVWorld2D(E, seed=31, env_offset=11, **kw)
    orc = oracle_mod.OracleMulti(num_envs=E, nthreads=8, **kw)
    env.set_prefetch(1)
    env.set_curriculum(levels, lo=0, hi=1); orc.set_curriculum(levels, lo=0, hi=1)
    env.reset(); orc.reset_philox(31, env_offset=11)
    rng = np.random.default_rng(cap * 100 + L)
    seed, hits = 31, 0
    for t in range(60):
        if t == 37:
            seed = 32
            env.seed = seed
        a = rng.uniform(-1, 1, size=(E, L, 2)).astype(np.float32)
        obs_g, rew_g, done_g, info = env.step_ex(torch.from_numpy(a).to(env.device), polar=True, auto_reset="agent0_done", step_cap=cap)
        obs_o, rew_o, done_o, rm_o, en_o, tr_o = orc.step_ex(a, action_mode=1, reset_policy=1, step_cap=cap, seed=seed, env_offset=11,
                                                             with_end=True)
        ctx = f"cap {cap}, step {t}"
        np.testing.assert_array_equal(_np(info["reset_mask"]).astype(np.uint8), rm_o, err_msg=ctx)
        np.testing.assert_array_equal(_np(info["ended"]).astype(np.uint8), en_o, err_msg=ctx)
        np.testing.assert_array_equal(_np(done_g).astype(np.uint8), done_o, err_msg=ctx)
        np.testing.assert_array_equal(_np(env.env_levels()), orc.level, err_msg=ctx)
        _compare_state(env, orc, ctx)
        assert obs_err(_np(obs_g), obs_o) <= TOL and float(np.abs(_np(rew_g) - rew_o).max()) <= TOL, ctx
        hits += int(rm_o.sum())
    assert hits >= E * 60 // (cap + 1) - E
    env.close()


@pytest.mark.parametrize("L,B", [(8, 16), (4, 0)])
def test_results_do_not_depend_on_the_staging_hints(amd, oracle_mod, L, B):
    """A staging workgroup draws what its last scan wrote into its hint slots {env + 1, episode} -- after checking against the
    env's record and the slot's tag that the layout is still wanted.  Here the hint area (the tail of the handle's slab, hence of
    a snapshot) is overwritten with random pairs twice in the middle of a run with two-step episodes and staging workgroups in
    every launch: plausible env ids with wrong episodes, right ones by chance, ids past the batch, doubled entries.  Every output
    and the whole state stay equal to the oracle's, which knows nothing of hints."""
    import torch
    E, cap = 4096, 2
    kw = dict(num_agents=L, num_bodies=B, body_period=4, x_size=26.0, y_size=22.0, d_sense=9.0, collider_radius=0.6)
    env = amd.BatchedMultiUAVWorld2D(E, seed=5, env_offset=7, **kw)
    orc = oracle_mod.OracleMulti(num_envs=E, nthreads=8, **kw)
    env.set_prefetch(1)
    env.reset(); orc.reset_philox(5, env_offset=7)
    rng = np.random.default_rng(77)
    groups = E // (64 // L if B == 0 else 8)              # env-workgroups = staging workgroups at every = 1
    tail = groups * 8 * 8                                 # eight {uint32, uint32} hint slots per staging workgroup
    for t in range(40):
        if t in (9, 22):
            sd = env.state_dict()
            snap = sd["snapshot"]
            junk = np.empty((tail // 8, 2), np.uint32)
            junk[:, 0] = rng.integers(0, 2 * E, size=tail // 8)       # env + 1: half of them past the batch, 0 = empty
            junk[:, 1] = rng.integers(0, 24, size=tail // 8)          # episode indices around the real ones (t / 3)
            k = len(junk[1::7])
            junk[::7][:k] = junk[1::7]                                # doubled entries
            snap[snap.numel() - tail:] = torch.from_numpy(junk.view(np.uint8).reshape(-1)).to(snap.device)
            env.load_state_dict(sd)
        a = rng.uniform(-1, 1, size=(E, L, 2)).astype(np.float32)
        obs_g, rew_g, done_g, info = env.step_ex(torch.from_numpy(a).to(env.device), polar=True, auto_reset="agent0_done", step_cap=cap)
        obs_o, rew_o, done_o, rm_o, en_o, tr_o = orc.step_ex(a, action_mode=1, reset_policy=1, step_cap=cap, seed=5, env_offset=7, with_end=True)
        ctx = f"L{L} B{B} step {t}"
        np.testing.assert_array_equal(_np(info["reset_mask"]).astype(np.uint8), rm_o, err_msg=ctx)
        np.testing.assert_array_equal(_np(done_g).astype(np.uint8), done_o, err_msg=ctx)
        _compare_state(env, orc, ctx)
        assert obs_err(_np(obs_g), obs_o) <= TOL and float(np.abs(_np(rew_g) - rew_o).max()) <= TOL, ctx
    env.close()


def test_explicit_env_levels_and_parked_learners(amd, oracle_mod):
    """No bodies; levels assigned per env by the caller; parked learners report obs 0 / reward 0 / done 1 and are nobody's
    neighbour; a one-level curriculum equal to the config reproduces the plain kernels bit for bit."""
    import torch
    E, L = 900, 6
    kw = dict(num_agents=L, x_size=30.0, y_size=30.0, d_sense=10.0)
    env = amd.BatchedMultiUAVWorld2D(E, seed=2, **kw)
    orc = oracle_mod.OracleMulti(num_envs=E, nthreads=8, **kw)
    levels = [dict(x_size=20.0, y_size=20.0, collider_radius=0.6, d_sense=6.0, n_active=1),
              dict(x_size=26.0, y_size=22.0, collider_radius=1.0, d_sense=10.0, n_active=3),
              dict(x_size=30.0, y_size=30.0, collider_radius=1.0, d_sense=10.0, n_active=6)]
    env.set_curriculum(levels)          # lo < 0: explicit per-env levels
    orc.set_curriculum(levels)
    assign = (np.arange(E) * 7 % 3).astype(np.uint8)
    env.set_env_levels(assign)
    orc.set_env_levels(assign)
    obs = env.reset()
    orc.reset_philox(2)
    np.testing.assert_array_equal(_np(env.env_levels()), assign)
    assert obs_err(_np(obs), orc.observe()) <= TOL
    rng = np.random.default_rng(4)
    for t in range(120):
        act = _seek_actions(orc, rng)
        o_g, r_g, d_g, info = env.step_ex(act, auto_reset="agent0_done", step_cap=50)
        o_o, r_o, d_o, rm_o = orc.step_ex(act, reset_policy=1, step_cap=50, seed=2)
        np.testing.assert_array_equal(_np(info["reset_mask"]).astype(np.uint8), rm_o)
        np.testing.assert_array_equal(_np(d_g).astype(np.uint8), d_o)
        _compare_state(env, orc, f"step {t}")
        assert obs_err(_np(o_g), o_o) <= TOL and float(np.abs(_np(r_g) - r_o).max()) <= TOL
        parked = (orc.flags & 32) != 0
        stepping = (rm_o == 0)[:, None] & parked
        assert (_np(o_g)[parked] == 0).all() and (_np(r_g)[parked] == 0).all() and (_np(d_g)[stepping]).all()
    env.close()
    # one level == the config: the EXT kernels must give exactly what the plain kernels give
    a = amd.BatchedMultiUAVWorld2D(E, seed=9, **kw)
    b = amd.BatchedMultiUAVWorld2D(E, seed=9, **kw)
    b.set_curriculum([dict(x_size=30.0, y_size=30.0, collider_radius=1.0, d_sense=10.0)], lo=0, hi=0)
    oa, ob = a.reset(), b.reset()
    assert torch.equal(oa, ob)
    for t in range(60):
        act = torch.from_numpy(rng.uniform(-10, 10, size=(E, L, 2)).astype(np.float32)).to(a.device)
        ra, rb = a.step(act), b.step(act)
        assert torch.equal(ra[0], rb[0]) and torch.equal(ra[1], rb[1]) and torch.equal(ra[2], rb[2])
    sa, sb = a.get_state(), b.get_state()
    for k in ("loc", "vel", "flags", "counters"):
        assert torch.equal(sa[k], sb[k]), k
    a.close(); b.close()


def test_config5_full_size_properties(amd):
    """configs[4] at BASELINE size (65 536 envs x 8 learners + 16 bodies): size-independent properties -- determinism,
    independence of the shard cut (Philox keyed by global env id), bodies stay inside their env's box, parked / active
    bookkeeping is consistent.  GPU against GPU; the comparison with the oracle at this size is
    test_config5_full_size_oracle_differential below."""
    import torch
    E, L, B = 65536, 8, 16
    kw = dict(num_agents=L, num_bodies=B, body_period=32, body_seed=5)

    def run(E_, off, steps=48):
        env = amd.BatchedMultiUAVWorld2D(E_, seed=13, env_offset=off, **kw)
        env.set_curriculum(LEVELS, lo=0, hi=2)
        env.reset()
        g = torch.Generator(device=env.device).manual_seed(77)
        acts = torch.rand((steps, E, L, 2), generator=g, device=env.device)[:, off:off + E_] * 2 - 1
        outs = []
        for t in range(steps):
            o, r, d, info = env.step_ex(acts[t], polar=True, auto_reset="all_done", step_cap=20, evaluate=True)
            if t in (0, 19, 20, 21, steps - 1):
                outs.append((o.clone(), r.clone(), d.clone(), info["reset_mask"].clone(), info["truncated"].clone()))
        st = env.get_state()
        bodies, lv = env.get_bodies(), env.env_levels()
        env.close()
        return outs, st, bodies, lv

    whole = run(E, 0)
    again = run(E, 0)
    for (a, b) in zip(whole[0], again[0]):
        for x, y in zip(a, b):
            assert torch.equal(x, y)                                   # deterministic
    cut = 40000                                                        # not a multiple of the envs per wavefront
    lo, hi = run(cut, 0), run(E - cut, cut)
    for w, a, b in zip(whole[0], lo[0], hi[0]):
        for x, p, q in zip(w, a, b):
            assert torch.equal(x, torch.cat([p, q], dim=0))            # sharding by env index changes nothing
    assert torch.equal(whole[2], torch.cat([lo[2], hi[2]], dim=0)) and torch.equal(whole[3], torch.cat([lo[3], hi[3]], dim=0))
    outs, st, bodies, lv = whole
    assert all(torch.isfinite(o[0]).all() and torch.isfinite(o[1]).all() for o in outs)
    assert int(outs[2][3].sum()) >= 0.99 * E and int(outs[1][4].sum()) >= 0.99 * E   # cap 20: truncated at call 19, re-initialised by call 20
    half = torch.tensor([[l["x_size"] / 2, l["y_size"] / 2] for l in LEVELS], device=bodies.device)[lv.long()]   # [E, 2]
    nb = torch.tensor([l["b_active"] for l in LEVELS], device=bodies.device)[lv.long()]
    on = torch.arange(B, device=bodies.device)[None, :] < nb[:, None]
    assert bool((bodies[..., :2].abs() <= half[:, None, :] + 1e-4)[on].all()) and bool(torch.isinf(bodies[..., 0][~on]).all())
    nl = torch.tensor([l["n_active"] for l in LEVELS], device=bodies.device)[lv.long()]
    parked = (st["flags"] & 32) != 0
    assert torch.equal(parked, torch.arange(L, device=bodies.device)[None, :] >= nl[:, None])
    assert len(torch.unique(lv)) == 3


def test_config5_full_size_oracle_differential(amd, oracle_mod):
    """configs[4] at BASELINE size against the ORACLE, every env: 65 536 envs x (8 learners + 16 bodies), 3-level randomized-
    reset curriculum, fused step_ex with polar commands, all-done auto-reset and a step cap of 12 (so that every env is
    re-initialised from a parked layout -- or draws in place -- two or three times), 36 steps: reset / ended / truncated
    masks, done masks, levels, learner state, body records and counters bit for bit, observations / rewards to 1e-5."""
    import torch
    E, L, B, T, cap = 65536, 8, 16, 36, 12
    nthreads = os.cpu_count() or 8
    kw = dict(num_agents=L, num_bodies=B, body_period=8, body_seed=5)
    env = amd.BatchedMultiUAVWorld2D(E, seed=13, **kw)
    orc = oracle_mod.OracleMulti(num_envs=E, nthreads=nthreads, **kw)
    env.set_curriculum(LEVELS, lo=0, hi=2); orc.set_curriculum(LEVELS, lo=0, hi=2)
    env.reset(); orc.reset_philox(13)
    rng = np.random.default_rng(9)
    n_reset = n_trunc = 0
    for t in range(T):
        a = rng.uniform(-1, 1, size=(E, L, 2)).astype(np.float32)
        o_g, r_g, d_g, info = env.step_ex(torch.from_numpy(a).to(env.device), polar=True, auto_reset="all_done", step_cap=cap, evaluate=True)
        o_o, r_o, d_o, rm_o, en_o, tr_o = orc.step_ex(a, evaluate=True, action_mode=1, reset_policy=2, step_cap=cap, seed=13, with_end=True)
        ctx = f"step {t}"
        np.testing.assert_array_equal(_np(info["reset_mask"]).astype(np.uint8), rm_o, err_msg=ctx)
        np.testing.assert_array_equal(_np(info["ended"]).astype(np.uint8), en_o, err_msg=ctx)
        np.testing.assert_array_equal(_np(info["truncated"]).astype(np.uint8), tr_o, err_msg=ctx)
        np.testing.assert_array_equal(_np(d_g).astype(np.uint8), d_o, err_msg=ctx)
        assert obs_err(_np(o_g), o_o) <= TOL and float(np.abs(_np(r_g) - r_o).max()) <= TOL, ctx
        if t % 6 == 5 or t == T - 1:
            np.testing.assert_array_equal(_np(env.env_levels()), orc.level, err_msg=ctx)
            _compare_state(env, orc, ctx)
        n_reset += int(rm_o.sum()); n_trunc += int(tr_o.sum())
    assert n_reset >= 2 * E and n_trunc >= E, (n_reset, n_trunc)
    env.close()


def test_extension_fixture_replay_on_device(amd):
    """The committed known-answer fixture of the extension (tests/golden/ext_bodies_levels.npz, produced by the oracle with
    tests/golden/make_ext_golden.py) replayed through the C ABI without the oracle in the loop."""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_ext_golden as g
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "ext_bodies_levels.npz"))
    env = amd.BatchedMultiUAVWorld2D(g.E, seed=g.SEED, env_offset=g.OFFSET, **g.KW)
    env.set_curriculum(g.LEVELS, lo=0, hi=1)
    env.reset()
    np.testing.assert_array_equal(_np(env.get_bodies()), fx["init_body"])
    np.testing.assert_array_equal(_np(env.env_levels()), fx["init_level"])
    np.testing.assert_array_equal(_np(env.get_state()["flags"]), fx["init_flags"])
    for t in range(g.T):
        o, r, d, info = env.step_ex(torch.from_numpy(fx["actions"][t]).to(env.device), polar=True, auto_reset="all_done",
                                    step_cap=g.CAP, evaluate=True)
        for key, got in (("done", d), ("reset_mask", info["reset_mask"]), ("ended", info["ended"]), ("truncated", info["truncated"])):
            np.testing.assert_array_equal(_np(got).astype(np.uint8), fx[key][t], err_msg=f"{key} step {t}")
        st = env.get_state()
        on = (fx["flags"][t] & 32) == 0
        np.testing.assert_array_equal(_np(st["flags"]), fx["flags"][t])
        np.testing.assert_array_equal(_np(st["loc"])[on], fx["loc"][t].astype(np.float32)[on])
        np.testing.assert_array_equal(_np(st["vel"])[on], fx["vel"][t][on])
        np.testing.assert_array_equal(_np(st["counters"]), fx["counters"][t].astype(np.int32))
        np.testing.assert_array_equal(_np(env.get_bodies()), fx["body"][t])
        np.testing.assert_array_equal(_np(env.env_levels()), fx["level"][t])
        assert obs_err(_np(o), fx["obs"][t]) <= TOL and float(np.abs(_np(r) - fx["rew"][t]).max()) <= TOL
    env.close()


def test_randomized_extension_shapes(amd, oracle_mod):
    """Differential run over odd shapes: learner / body counts from 1+0 to 64 slots, one env up to a few hundred, every
    waypoint period, random level tables, small step caps (resets every few calls), all three layout-prefetch cadences."""
    import torch
    rng = np.random.default_rng(2024)
    shapes = [(1, 63), (63, 1), (64, 0), (3, 61), (2, 1), (7, 9), (16, 16), (31, 33), (5, 0), (1, 1), (9, 23), (12, 4),
              (24, 0), (11, 0), (13, 0), (48, 0), (22, 0)]   # the last five: agent counts on multi-wavefront workgroups
    for case, (L, B) in enumerate(shapes):
        E = int(rng.choice([1, 2, 7, 65, 130, 257]))
        period = int(rng.choice([1, 2, 8, 64]))
        box = float(rng.uniform(4.0, 9.0) * np.sqrt(L + B) + 6.0)
        kw = dict(num_agents=L, num_bodies=B, body_speed=float(rng.uniform(0.0, 9.0)), body_period=period, body_seed=case,
                  x_size=box, y_size=box * float(rng.uniform(0.7, 1.0)), d_sense=float(rng.uniform(4.0, 16.0)),
                  collider_radius=float(rng.uniform(0.2, 0.9)))
        os.environ["UAVX_STAGE_BEHIND"] = str(case & 1)      # staging workgroups behind / in front of the env-workgroups
        try:
            env = amd.BatchedMultiUAVWorld2D(E, seed=case, env_offset=3 * case, **kw)
        finally:
            del os.environ["UAVX_STAGE_BEHIND"]
        orc = oracle_mod.OracleMulti(num_envs=E, nthreads=4, **kw)
        nlev = int(rng.integers(0, 4)) if B or case < 12 else int(rng.integers(1, 4))   # B == 0 needs levels to be an extension handle
        if nlev:
            levels = [dict(x_size=box * float(rng.uniform(0.7, 1.1)), y_size=box * float(rng.uniform(0.6, 1.0)),
                           collider_radius=float(rng.uniform(0.2, 0.8)), d_sense=float(rng.uniform(3.0, 14.0)),
                           n_active=int(rng.integers(1, L + 1)), b_active=int(rng.integers(0, B + 1))) for _ in range(nlev)]
            lo = int(rng.integers(-1, nlev))
            hi = int(rng.integers(max(lo, 0), nlev))
            env.set_curriculum(levels, lo=lo, hi=hi)
            orc.set_curriculum(levels, lo=lo, hi=hi)
            if lo < 0:
                assign = rng.integers(0, nlev, size=E).astype(np.uint8)
                env.set_env_levels(assign); orc.set_env_levels(assign)
        env.set_prefetch(int(rng.choice([0, 1, 3])))
        cap = int(rng.choice([1, 2, 5, 9]))
        pol, code = [("agent0_done", 1), ("all_done", 2), (None, 0)][case % 3]
        env.reset(); orc.reset_philox(case, env_offset=3 * case)
        ctx = f"case {case}: L{L} B{B} E{E} period {period} levels {nlev} cap {cap} {pol}"
        _compare_state(env, orc, ctx + " reset")
        seed = case
        for t in range(16):
            if t == 5:      # an explicit masked reset in the middle (folds the running episodes, draws fresh layouts)
                mask = rng.random(E) < 0.5
                env.reset(mask=torch.from_numpy(mask).to(env.device)); orc.reset_philox(seed, mask=mask, env_offset=3 * case)
                _compare_state(env, orc, ctx + " masked reset")
            if t == 7:      # state read back and written unchanged: a no-op (level bits, parked flags, counters survive)
                st = env.get_state()
                env.set_state(**{k: v for k, v in st.items()})
                if B:
                    env.set_bodies(env.get_bodies())
            if t == 9:      # another seed: layouts parked for the old one must not be used
                seed = case + 1000
                env.seed = seed
            if t == 11 and nlev == 0:   # handle-wide world change (uavx_set_config) under bodies
                env.set_config(x_size=box * 0.9, d_sense=kw["d_sense"] * 0.8)
                orc.set_config(x_size=box * 0.9, d_sense=kw["d_sense"] * 0.8)
            a = rng.uniform(-1, 1, size=(E, L, 2)).astype(np.float32)
            o_g, r_g, d_g, info = env.step_ex(torch.from_numpy(a).to(env.device), polar=True, auto_reset=pol, step_cap=cap,
                                              evaluate=bool(t % 2))
            o_o, r_o, d_o, rm, en, tr = orc.step_ex(a, action_mode=1, reset_policy=code, step_cap=cap, evaluate=bool(t % 2),
                                                    track_returns=True, seed=seed, env_offset=3 * case, with_end=True)
            np.testing.assert_array_equal(_np(info["reset_mask"]).astype(np.uint8), rm, err_msg=ctx)
            np.testing.assert_array_equal(_np(info["ended"]).astype(np.uint8), en, err_msg=ctx)
            np.testing.assert_array_equal(_np(info["truncated"]).astype(np.uint8), tr, err_msg=ctx)
            np.testing.assert_array_equal(_np(d_g).astype(np.uint8), d_o, err_msg=ctx)
            _compare_state(env, orc, f"{ctx} step {t}")
            assert obs_err(_np(o_g), o_o) <= TOL and float(np.abs(_np(r_g) - r_o).max()) <= TOL, ctx
        env.close()


def test_layout_staging_on_a_caller_stream(amd, oracle_mod):
    """Everything driven from a non-default torch stream, every env's next layout re-staged in every launch (every = 1), no
    host synchronisation between the calls of a burst -- results must still be the oracle's (an ordering hole between the
    staging and the step workgroups of a launch would show as a stale or torn layout)."""
    import torch
    E, L, B = 2048, 8, 16
    kw = dict(num_agents=L, num_bodies=B, body_period=4, x_size=20.0, y_size=20.0, d_sense=8.0, collider_radius=0.5)
    env = amd.BatchedMultiUAVWorld2D(E, seed=12, **kw)
    orc = oracle_mod.OracleMulti(num_envs=E, nthreads=8, **kw)
    env.set_prefetch(1)
    side = torch.cuda.Stream(env.device)
    rng = np.random.default_rng(5)
    acts = rng.uniform(-1, 1, size=(48, E, L, 2)).astype(np.float32)
    d_acts = torch.from_numpy(acts).to(env.device)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        env.reset()
        outs = []
        for t in range(48):     # a burst: nothing but launches on `side`
            o, r, d, info = env.step_ex(d_acts[t], polar=True, auto_reset="agent0_done", step_cap=5)
            outs.append((o.clone(), r.clone(), d.clone(), info["reset_mask"].clone(), info["truncated"].clone()))
        bodies = env.get_bodies()
    side.synchronize()
    orc.reset_philox(12)
    for t in range(48):
        o_o, r_o, d_o, rm, en, tr = orc.step_ex(acts[t], action_mode=1, reset_policy=1, step_cap=5, seed=12, with_end=True)
        o, r, d, m, tru = outs[t]
        np.testing.assert_array_equal(_np(m).astype(np.uint8), rm, err_msg=f"step {t}")
        np.testing.assert_array_equal(_np(tru).astype(np.uint8), tr, err_msg=f"step {t}")
        np.testing.assert_array_equal(_np(d).astype(np.uint8), d_o, err_msg=f"step {t}")
        assert obs_err(_np(o), o_o) <= TOL and float(np.abs(_np(r) - r_o).max()) <= TOL, t
    np.testing.assert_array_equal(_np(bodies), orc.body)
    env.close()


def test_step_ex_replayed_from_a_short_hipgraph(amd, oracle_mod):
    """A 6-step hipGraph of fused step_ex launches (shorter than the default 16-call prefetch cadence), replayed 8 times with
    fresh commands: the first recorded call of the capture carries the layout-prefetch side launch, so every replay draws
    ahead once -- and whatever is parked or missed, the results are the oracle's."""
    import torch
    E, L, B, G, R = 1024, 8, 16, 6, 8
    kw = dict(num_agents=L, num_bodies=B, body_period=4, x_size=20.0, y_size=20.0, d_sense=8.0, collider_radius=0.5)
    env = amd.BatchedMultiUAVWorld2D(E, seed=21, **kw)
    orc = oracle_mod.OracleMulti(num_envs=E, nthreads=8, **kw)
    rng = np.random.default_rng(9)
    acts = rng.uniform(-1, 1, size=(R, G, E, L, 2)).astype(np.float32)
    tape = torch.from_numpy(acts).to(env.device)
    slot = torch.zeros((G, E, L, 2), dtype=torch.float32, device=env.device)
    obs = torch.zeros((G, E, L, 10), dtype=torch.float32, device=env.device)
    rew = torch.zeros((G, E, L), dtype=torch.float32, device=env.device)
    done = torch.zeros((G, E, L), dtype=torch.uint8, device=env.device)
    flags = torch.zeros((G, 3, E), dtype=torch.uint8, device=env.device)
    step = lambda i: env.step_ex(slot[i], polar=True, auto_reset="agent0_done", step_cap=7, out=(obs[i], rew[i], done[i]),
                                 flags_out=(flags[i, 0], flags[i, 1], flags[i, 2]))
    env.reset()
    side = torch.cuda.Stream(env.device)
    side.wait_stream(torch.cuda.current_stream(env.device))
    with torch.cuda.stream(side):          # lazy initialisation outside the capture; these three steps count as steps
        slot.copy_(tape[0])
        for i in range(3):
            step(i)
        warm = [t.clone() for t in (obs[:3], rew[:3], done[:3], flags[:3])]
    torch.cuda.current_stream(env.device).wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(G):
            step(i)
    got = []
    for r in range(R):
        slot.copy_(tape[r])
        g.replay()
        got.append([t.clone() for t in (obs, rew, done, flags)])
    bodies = env.get_bodies()
    torch.cuda.synchronize()
    orc.reset_philox(21)

    def check(a, o, r, d, f, tag):
        o_o, r_o, d_o, rm, en, tr = orc.step_ex(a, action_mode=1, reset_policy=1, step_cap=7, seed=21, with_end=True)
        np.testing.assert_array_equal(_np(f[0]), rm, err_msg=tag)
        np.testing.assert_array_equal(_np(f[1]), en, err_msg=tag)
        np.testing.assert_array_equal(_np(f[2]), tr, err_msg=tag)
        np.testing.assert_array_equal(_np(d), d_o, err_msg=tag)
        assert obs_err(_np(o), o_o) <= TOL and float(np.abs(_np(r) - r_o).max()) <= TOL, tag
    for i in range(3):
        check(acts[0, i], warm[0][i], warm[1][i], warm[2][i], warm[3][i], f"warm-up step {i}")
    for r in range(R):
        o, w, d, f = got[r]
        for i in range(G):
            check(acts[r, i], o[i], w[i], d[i], f[i], f"replay {r} step {i}")
    np.testing.assert_array_equal(_np(bodies), orc.body)
    env.close()


def test_vector_env_surface_with_bodies(amd, oracle_mod):
    """The VectorEnv-shaped view over a world with scripted bodies: spaces stay per LEARNER, one fused launch per step,
    info carries reset_mask / ended / truncated."""
    import torch
    E, L, B = 640, 4, 6
    kw = dict(num_agents=L, num_bodies=B, body_period=8, x_size=22.0, y_size=22.0, d_sense=8.0)
    venv = amd.UAVVectorEnv(E, auto_reset="agent0_done", step_cap=9, polar=True, seed=4, **kw)
    orc = oracle_mod.OracleMulti(num_envs=E, nthreads=8, **kw)
    assert venv.observation_space.shape == (E, L, 10) and venv.action_space.shape == (E, L, 2) and venv.num_agents == L
    obs = venv.reset()
    orc.reset_philox(4)
    assert obs_err(_np(obs), orc.observe()) <= TOL
    rng = np.random.default_rng(3)
    for t in range(30):
        a = rng.uniform(-1, 1, size=(E, L, 2)).astype(np.float32)
        obs, rew, done, info = venv.step(torch.from_numpy(a).to(venv.device))
        o_o, r_o, d_o, rm, en, tr = orc.step_ex(a, action_mode=1, reset_policy=1, step_cap=9, track_returns=True, seed=4, with_end=True)
        np.testing.assert_array_equal(_np(done).astype(np.uint8), d_o)
        for key, want in (("reset_mask", rm), ("ended", en), ("truncated", tr)):
            np.testing.assert_array_equal(_np(info[key]).astype(np.uint8), want, err_msg=f"{key} step {t}")
        assert obs_err(_np(obs), o_o) <= TOL and float(np.abs(_np(rew) - r_o).max()) <= TOL
    np.testing.assert_array_equal(_np(venv.env.get_bodies()), orc.body)
    venv.close()


def test_static_obstacles_single_uav(amd, oracle_mod):
    """BASELINE configs[0] / [1] name "1 UAV + static obstacles": a body with speed 0 is a static obstacle.  One learner among
    12 of them: the records never move, the learner senses / collides with them exactly as the oracle says."""
    import torch
    E, L, B = 4096, 1, 12
    kw = dict(num_agents=L, num_bodies=B, body_speed=0.0, body_period=4, x_size=24.0, y_size=24.0, d_sense=8.0)
    env = amd.BatchedMultiUAVWorld2D(E, seed=6, **kw)
    orc = oracle_mod.OracleMulti(num_envs=E, nthreads=8, **kw)
    env.reset(); orc.reset_philox(6)
    start = _np(env.get_bodies()).copy()
    rng = np.random.default_rng(2)
    hits = 0
    for t in range(80):
        act = _seek_actions(orc, rng, noise=0.5)
        o_g, r_g, d_g, _ = env.step(torch.from_numpy(act).to(env.device))
        o_o, r_o, d_o = orc.step(act)
        np.testing.assert_array_equal(_np(d_g).astype(np.uint8), d_o)
        _compare_state(env, orc, f"step {t}")
        assert obs_err(_np(o_g), o_o) <= TOL and float(np.abs(_np(r_g) - r_o).max()) <= TOL
        hits += int((r_o == -2).sum())
    now = _np(env.get_bodies())
    np.testing.assert_array_equal(now[..., :2], start[..., :2])      # static: positions untouched (waypoints re-drawn, unused)
    assert hits > 0
    env.close()


def test_extension_argument_checks(amd):
    """Error behaviour of the new entry points and of caller-owned output buffers (no kernel must run on a bad call)."""
    import torch
    env = amd.BatchedMultiUAVWorld2D(64, num_agents=4, num_bodies=4)
    with pytest.raises(ValueError):
        env.set_body_rule(period=48)                      # not a power of two
    with pytest.raises(ValueError):
        env.set_curriculum([dict(x_size=10, y_size=10, collider_radius=1, d_sense=5, n_active=9)])   # n_active > L
    with pytest.raises(ValueError):
        env.set_curriculum(LEVELS[:1] * 17)               # more than UAVX_MAX_LEVELS
    with pytest.raises(ValueError):
        env.set_curriculum([dict(x_size=10, y_size=10, collider_radius=1, d_sense=5, n_active=2, b_active=2)], lo=0, hi=3)
    with pytest.raises(RuntimeError):
        env.step_k(torch.zeros((2, 64, 4, 2), device=env.device))          # k > 1 has no extension variant
    with pytest.raises(RuntimeError):
        env.set_position_mode("float64")
    a = torch.zeros((64, 4, 2), device=env.device)
    good = (torch.zeros((64, 4, 10), device=env.device), torch.zeros((64, 4), device=env.device),
            torch.zeros((64, 4), dtype=torch.uint8, device=env.device))
    env.reset()
    env.step_ex(a, out=good)
    for bad in ((good[0][:32], good[1], good[2]),                          # wrong shape: would write out of bounds
                (good[0].double(), good[1], good[2]),                      # wrong dtype
                (torch.zeros((64, 4, 20), device=env.device)[..., ::2], good[1], good[2]),   # strided view
                (good[0], good[1].cpu(), good[2])):                        # wrong device
        with pytest.raises(ValueError):
            env.step_ex(a, out=bad)
    with pytest.raises(ValueError):
        amd.BatchedMultiUAVWorld2D(8, num_agents=40, num_bodies=30)         # more than 64 slots
    env.close()


def test_uw_entry_points_reject_misaligned_buffers(amd):
    """uavx_uw_* cast obs to float4 and load float2 / double2 actions: a sliced (misaligned) view must be refused with
    UAVX_ERR_INVALID_ARG instead of faulting on the GPU."""
    import torch
    from gym_uav_collision_avoidance_amd import _lib
    env = amd.BatchedUAVWorld2D(128)
    env.reset()
    L, h, st = env._L, env._h, env._stream()
    obs = torch.zeros(128 * 4 + 4, device=env.device)
    act = torch.zeros(128 * 2 + 2, device=env.device)
    rew = torch.zeros(128, device=env.device)
    done = torch.zeros(128, dtype=torch.uint8, device=env.device)
    ok = L.uavx_uw_step(h, act.data_ptr(), _lib.F32, obs.data_ptr(), rew.data_ptr(), done.data_ptr(), None, st)
    assert ok == 0
    assert L.uavx_uw_step(h, act.data_ptr(), _lib.F32, obs.data_ptr() + 4, rew.data_ptr(), done.data_ptr(), None, st) == -1
    assert L.uavx_uw_step(h, act.data_ptr() + 4, _lib.F32, obs.data_ptr(), rew.data_ptr(), done.data_ptr(), None, st) == -1
    assert L.uavx_uw_step(h, act.data_ptr() + 8, _lib.F64, obs.data_ptr(), rew.data_ptr(), done.data_ptr(), None, st) == -1
    assert L.uavx_uw_observe(h, obs.data_ptr() + 8, st) == -1
    assert L.uavx_uw_reset(h, None, 0, obs.data_ptr() + 4, st) == -1
    args = _lib.UWStepArgs(act.data_ptr(), _lib.F32, 0, 0, 0, 0, 0, 0, obs.data_ptr() + 4, rew.data_ptr(), done.data_ptr(), None, None, None, None)
    assert L.uavx_uw_step_ex(h, ctypes.byref(args), st) == -1
    assert b"aligned" in L.uavx_uw_last_error(h)
    env.close()


def test_largest_supported_batch_addresses_correctly(amd):
    """E*N just under the 2^26 agent-slot limit of uavx_create (32-bit byte offsets: the obs block is 2.68 GB, above
    2^31): the last envs of the big batch must equal a small handle created at the same global env offset (Philox is
    keyed by global env id), i.e. every offset computation at the top of the range is right."""
    import torch
    N = 4
    E = (1 << 26) // N - 1
    tail = 4099
    with pytest.raises(RuntimeError):
        amd.BatchedMultiUAVWorld2D(E + 1, num_agents=N)                     # at the limit: refused, not wrapped
    big = amd.BatchedMultiUAVWorld2D(E, num_agents=N, seed=3)
    small = amd.BatchedMultiUAVWorld2D(tail, num_agents=N, seed=3, env_offset=E - tail)
    ob, os_ = big.reset(), small.reset()
    assert torch.equal(ob[E - tail:], os_)
    g = torch.Generator(device=big.device).manual_seed(5)
    for t in range(3):
        act = torch.rand((tail, N, 2), generator=g, device=big.device) * 20 - 10
        full = torch.zeros((E, N, 2), device=big.device)
        full[E - tail:] = act
        rb, rs = big.step(full), small.step(act)
        assert torch.equal(rb[0][E - tail:], rs[0]) and torch.equal(rb[1][E - tail:], rs[1]) and torch.equal(rb[2][E - tail:], rs[2])
        del full
    sb, ss = big.get_state(), small.get_state()
    for k in ("loc", "vel", "tgt", "flags"):
        assert torch.equal(sb[k][E - tail:], ss[k]), k
    assert torch.equal(sb["counters"][E - tail:, :3], ss["counters"][:, :3])
    big.close(); small.close()


def test_body_trajectories_against_independent_arithmetic_on_device(amd):
    """The device against arithmetic that comes from neither the oracle nor the HIP source (tests/golden_util.py: Philox4x32-10
    from its published definition, leg records and positions in numpy float32 from the text of include/uavx.h): three legs of
    every body, positions / displacements / leg counts bit for bit, headings to float32 accuracy.  The CPU twin of this test
    (tests/test_oracle_ext.py) holds the oracle to the same arithmetic."""
    import torch
    from golden_util import body_track
    L, B, E, period, speed, seed, off = 3, 5, 6, 8, 6.5, 0x1234567890, 7
    xs, ys = 44.0, 36.0
    env = amd.BatchedMultiUAVWorld2D(E, num_agents=L, num_bodies=B, x_size=xs, y_size=ys, body_speed=speed, body_period=period,
                                     body_seed=seed, env_offset=off, seed=11)
    env.reset()
    rec0 = _np(env.get_bodies())
    episode = int(_np(env.metrics())[0, 3]) - 1
    steps = 3 * period + 2
    want = {(e, b): body_track(rec0[e, b, :2], off + e, L + b, episode, seed, xs, ys, speed, 0.02, period, steps)
            for e in range(E) for b in range(B)}
    for (e, b), (_, legs) in want.items():
        _, dx, dy, n, heading, _, _ = legs[0]
        assert rec0[e, b, 2] == dx and rec0[e, b, 3] == dy and rec0[e, b, 5] == n, (e, b, rec0[e, b], legs[0])
        assert abs(float(rec0[e, b, 4]) - heading) <= 4e-7 * max(1.0, abs(heading))
    g = torch.Generator(device="cpu").manual_seed(5)
    for t in range(steps):
        env.step((torch.rand((E, L, 2), generator=g) * 6 - 3).to(env.device))
        rec = _np(env.get_bodies())
        for (e, b), (track, legs) in want.items():
            np.testing.assert_array_equal(rec[e, b, :2], track[t], err_msg=f"step {t} env {e} body {b}")
            cur = [l for l in legs if l[0] <= t][-1]
            assert rec[e, b, 2] == cur[1] and rec[e, b, 3] == cur[2] and rec[e, b, 5] == cur[3], (t, e, b)
            assert abs(float(rec[e, b, 4]) - cur[4]) <= 4e-7 * max(1.0, abs(cur[4]))
    env.close()
