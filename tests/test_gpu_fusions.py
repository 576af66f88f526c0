"""GPU tests of the trainer-loop fusions (uavx_step_ex: polar actions, next-step auto-reset, episode
statistics) and the zero-copy device replay.  These are NEW semantics of the build:
the env step inside is the reference-pinned one; the bookkeeping is checked against the oracle's
restatement (oracle/uavx_oracle.c: uavo_step_ex) and against the reference trainers' own formulas."""
import math

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


def test_polar_conversion_matches_the_trainers_formula(amd, oracle_mod):
    """Device float32 conversion == oracle restatement bit for bit, and == test_sac_multi.py:77-80 to 1e-6."""
    import torch
    E, n = 4096, 4
    env = amd.BatchedMultiUAVWorld2D(E, num_agents=n, seed=1)
    orc = oracle_mod.OracleMulti(num_envs=E, num_agents=n, nthreads=8)
    env.reset()
    orc.reset_philox(1)
    rng = np.random.default_rng(0)
    a = rng.uniform(-1, 1, size=(E, n, 2)).astype(np.float32)
    a[0, 0] = [1.0, 1.0]; a[0, 1] = [-1.0, -1.0]; a[0, 2] = [0.0, 0.5]; a[0, 3] = [0.3, -0.5]
    env.step_ex(torch.from_numpy(a).to(env.device), polar=True)
    orc.step_ex(a, action_mode=1)
    st = env.get_state()
    np.testing.assert_array_equal(_np(st["vel"]), orc.vel)
    np.testing.assert_array_equal(_np(st["loc"]), orc.loc.astype(np.float32))
    vmax = np.linalg.norm(env.action_space.high)
    for (a0, a1) in a.reshape(-1, 2)[:2000]:
        v = (np.float64(a0) / 2 + 0.5) * np.float64(vmax)
        th = np.float64(a1) * math.pi
        want = np.array([v * math.cos(th), v * math.sin(th)])
        got = oracle_mod.polar_to_command(a0, a1, np.float32(vmax))
        assert np.abs(got - want).max() < 4e-6  # float32 arithmetic on |command| <= 14.2
    env.close()


@pytest.mark.parametrize("policy,code,cap,n", [("agent0_done", 1, 0, 4), ("all_done", 2, 90, 4), (None, 0, 40, 8),
                                                 ("agent0_done", 1, 55, 5), ("all_done", 2, 70, 1), ("agent0_done", 1, 1, 4),
                                                 ("agent0_done", 1, 3, 24), ("noprefetch", 1, 30, 4), ("tiles", 1, 40, 8),
                                                 ("all_done", 2, 45, 10), ("agent0_done", 1, 30, 12)])
def test_auto_reset_and_episode_stats_vs_oracle(amd, oracle_mod, monkeypatch, policy, code, cap, n):
    import torch
    E = 1536
    kw = dict(x_size=26.0, y_size=26.0, num_agents=n, d_sense=9.0)
    if policy == "tiles":   # two one-wavefront tiles per workgroup (what uavx_create picks at 65 536 x 8), forced at this size
        monkeypatch.setenv("UAVX_TILES", "2")
        policy = "agent0_done"
    env = amd.BatchedMultiUAVWorld2D(E, seed=77, env_offset=5, **kw)
    if policy == "noprefetch":   # every reset drawn inside the step launch (the pre-drawn layouts switched off)
        env.set_prefetch(0)
        policy = "agent0_done"
    orc = oracle_mod.OracleMulti(num_envs=E, nthreads=8, **kw)
    env.reset()
    orc.reset_philox(77, env_offset=5)
    rng = np.random.default_rng(3)
    resets = 0
    for t in range(260):
        d = orc.tgt - orc.loc
        dist = np.linalg.norm(d, axis=-1, keepdims=True)
        act = d / np.maximum(dist, 1e-9) * np.where(dist > 0.3, np.minimum(8.0, np.sqrt(4.0 * dist)), 0.0)
        act = act + rng.normal(0, 0.3, size=act.shape) * (dist > 2.0)
        ev = policy == "all_done"
        obs_g, rew_g, done_g, info = env.step_ex(act, evaluate=ev, auto_reset=policy, step_cap=cap, track_returns=True)
        obs_o, rew_o, done_o, rmask_o = orc.step_ex(act, evaluate=ev, reset_policy=code, step_cap=cap, track_returns=True,
                                                    seed=77, env_offset=5)
        ctx = f"{policy}/{cap} step {t}"
        np.testing.assert_array_equal(_np(info["reset_mask"]).astype(np.uint8), rmask_o, err_msg=ctx)
        np.testing.assert_array_equal(_np(done_g).astype(np.uint8), done_o, err_msg=ctx)
        resets += int(rmask_o.sum())
        st = env.get_state()
        ref = orc.get_state()
        for key in ("loc", "vel", "tgt", "init_d", "prev_d", "flags"):
            np.testing.assert_array_equal(_np(st[key]), ref[key], err_msg=f"{ctx} {key}")
        np.testing.assert_array_equal(_np(st["counters"]), ref["counters"].astype(np.int32), err_msg=ctx)
        assert obs_err(_np(obs_g), obs_o) <= TOL and float(np.abs(_np(rew_g) - rew_o).max()) <= TOL, ctx
        assert (_np(rew_g)[rmask_o == 1] == 0).all() and (_np(done_g)[rmask_o == 1] == 0).all()
    assert resets > E // 4, "scenario must actually auto-reset"
    stats = {k: _np(v) for k, v in env.episode_stats().items()}
    np.testing.assert_array_equal(stats["episodes"], orc.fin_counts[:, 0])
    np.testing.assert_array_equal(stats["steps"], orc.fin_counts[:, 1])
    np.testing.assert_array_equal(stats["reach"], orc.fin_counts[:, 2])
    np.testing.assert_array_equal(stats["coll"], orc.fin_counts[:, 3])
    np.testing.assert_allclose(stats["return0"], orc.fin_returns[:, 0], atol=2e-3, rtol=1e-5)
    np.testing.assert_allclose(stats["score"], orc.fin_returns[:, 1], atol=2e-3, rtol=1e-5)
    summ = env.evaluation_summary()
    eps = int(orc.fin_counts[:, 0].sum())
    assert summ["episodes"] == eps
    assert summ["success_rate"] == pytest.approx(orc.fin_counts[:, 2].sum() / (n * eps))   # test_sac_multi.py:174
    assert summ["collision_rate"] == pytest.approx(orc.fin_counts[:, 3].sum() / (n * eps))  # test_sac_multi.py:175
    # an explicit reset also ends the running episodes
    env.reset()
    orc.reset_philox(77, env_offset=5)
    np.testing.assert_array_equal(_np(env.episode_stats()["episodes"]), orc.fin_counts[:, 0])
    env.clear_episode_stats()
    assert int(env.episode_stats()["episodes"].sum()) == 0
    env.close()


@pytest.mark.parametrize("n,bodies,policy,code,cap", [(4, 0, "agent0_done", 1, 40), (8, 16, "all_done", 2, 25), (5, 0, None, 0, 7), (1, 0, "agent0_done", 1, 3)])
def test_flags_packed_into_the_done_byte_vs_oracle(amd, oracle_mod, n, bodies, policy, code, cap):
    """uavx_step_ex with flags_mode = UAVX_FLAGS_IN_DONE (ABI v3): reset_mask / ended / truncated in bits 1..3 of the done byte
    of every env's agent 0, no flag arrays written.  Done bits, the three flags, state, observations and rewards against the
    oracle, and against a twin handle driven with the three plain arrays (the same launch otherwise: outputs bit for bit)."""
    import torch
    E = 2048
    kw = dict(x_size=24.0, y_size=24.0, num_agents=n, d_sense=9.0, **(dict(num_bodies=bodies, body_period=8) if bodies else {}))
    env = amd.BatchedMultiUAVWorld2D(E, seed=5, env_offset=3, **kw)
    twin = amd.BatchedMultiUAVWorld2D(E, seed=5, env_offset=3, **kw)
    orc = oracle_mod.OracleMulti(num_envs=E, nthreads=8, **kw)
    env.reset(); twin.reset(); orc.reset_philox(5, env_offset=3)
    rng = np.random.default_rng(11)
    seen = np.zeros(3, np.int64)
    for t in range(120):
        a = rng.uniform(-1, 1, size=(E, n, 2)).astype(np.float32)
        ad = torch.from_numpy(a).to(env.device)
        obs_g, rew_g, done_raw, info = env.step_ex(ad, polar=True, auto_reset=policy, step_cap=cap, packed_flags=True)
        assert done_raw.dtype == torch.uint8 and info.get("flags_in_done") is True
        obs_t, rew_t, done_t, info_t = twin.step_ex(ad, polar=True, auto_reset=policy, step_cap=cap)
        obs_o, rew_o, done_o, rm_o, en_o, tr_o = orc.step_ex(a, action_mode=1, reset_policy=code, step_cap=cap, seed=5, env_offset=3, with_end=True)
        raw = _np(done_raw)
        assert (raw[:, 1:] <= 1).all() and (raw[:, 0] < 16).all()                 # flags only in agent 0's byte
        done_g, rm_g, en_g, tr_g = (_np(x) for x in env.unpack_done(done_raw))
        ctx = f"step {t}"
        np.testing.assert_array_equal(done_g.astype(np.uint8), done_o, err_msg=ctx)
        np.testing.assert_array_equal(rm_g.astype(np.uint8), rm_o, err_msg=ctx)
        np.testing.assert_array_equal(en_g.astype(np.uint8), en_o, err_msg=ctx)
        np.testing.assert_array_equal(tr_g.astype(np.uint8), tr_o, err_msg=ctx)
        assert torch.equal(obs_g, obs_t) and torch.equal(rew_g, rew_t) and np.array_equal(done_g, _np(done_t))
        assert np.array_equal(rm_g, _np(info_t["reset_mask"])) and np.array_equal(en_g, _np(info_t["ended"])) and np.array_equal(tr_g, _np(info_t["truncated"]))
        assert obs_err(_np(obs_g), obs_o) <= TOL and float(np.abs(_np(rew_g) - rew_o).max()) <= TOL, ctx
        seen += [int(rm_o.sum()), int(en_o.sum()), int(tr_o.sum())]
    assert (seen > 0).all(), seen
    for key, v in env.get_state().items():
        assert torch.equal(v, twin.get_state()[key]), key
    with pytest.raises(ValueError):
        env.step_ex(ad, packed_flags=True, flags_out=(torch.zeros(E, dtype=torch.uint8, device=env.device),) * 3)
    env.close(); twin.close()


@pytest.mark.parametrize("n,bodies", [(4, 0), (8, 16)])
def test_snapshot_restores_a_running_batch_exactly(amd, n, bodies, tmp_path):
    """uavx_save / uavx_load: 100 fused steps (auto-reset, statistics, curriculum, bodies) -> save -> 50 steps -> load -> the
    same 50 steps again: every output of every step and episode_stats() identical; the snapshot also restores a FRESH handle
    (through torch.save / torch.load of the state dict), which then produces the same 50 steps."""
    import torch
    E = 4096
    kw = dict(num_agents=n, x_size=24.0, y_size=24.0, d_sense=9.0, **(dict(num_bodies=bodies, body_period=8, body_seed=4) if bodies else {}))
    levels = [dict(x_size=18.0, y_size=18.0, collider_radius=0.5, d_sense=7.0, n_active=max(1, n // 2), b_active=bodies // 2),
              dict(x_size=24.0, y_size=24.0, collider_radius=1.0, d_sense=9.0, n_active=n, b_active=bodies)]
    g = torch.Generator(device="cpu").manual_seed(2)
    tape = torch.rand((150, E, n, 2), generator=g) * 2 - 1
    step_kw = dict(polar=True, auto_reset="agent0_done", step_cap=23, track_returns=True)

    def run(env, lo, hi):
        outs = []
        for t in range(lo, hi):
            o, r, d, info = env.step_ex(tape[t].to(env.device), **step_kw)
            outs.append((o.clone(), r.clone(), d.clone(), info["reset_mask"].clone(), info["ended"].clone(), info["truncated"].clone()))
        return outs

    env = amd.BatchedMultiUAVWorld2D(E, seed=17, env_offset=9, **kw)
    env.set_curriculum(levels, lo=0, hi=1)
    env.reset()
    run(env, 0, 100)
    sd = env.state_dict()
    torch.save(sd, tmp_path / "env.pt")
    first = run(env, 100, 150)
    stats_first = {k: v.clone() for k, v in env.episode_stats().items()}
    env.set_level_window(1, 1)                      # wander off: another world version, other statistics
    run(env, 0, 7)
    env.load_state_dict(sd)
    again = run(env, 100, 150)
    for a, b in zip(first, again):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    for k, v in env.episode_stats().items():
        assert torch.equal(v, stats_first[k]), k
    state_first = {k: v.clone() for k, v in env.get_state().items()}
    env.close()
    fresh = amd.BatchedMultiUAVWorld2D(E, seed=0, env_offset=0, **kw)     # another seed / offset: the snapshot brings its own
    fresh.load_state_dict(torch.load(tmp_path / "env.pt", weights_only=True))
    third = run(fresh, 100, 150)
    for a, b in zip(first, third):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    for k, v in fresh.episode_stats().items():
        assert torch.equal(v, stats_first[k]), k
    for k, v in fresh.get_state().items():
        assert torch.equal(v, state_first[k]), k
    small = amd.BatchedMultiUAVWorld2D(E // 2, **kw)
    with pytest.raises(ValueError):
        small.load_state_dict(sd)
    # a truncated or damaged snapshot is refused before anything is copied (header fields feed copy lengths, kernel arguments
    # and level-table indices), and the refused handle still runs
    cut = dict(sd, snapshot=sd["snapshot"][: sd["snapshot"].numel() // 2].clone())
    with pytest.raises(ValueError):
        fresh.load_state_dict(cut)
    import struct
    for off, fmt, bad in ((24, "<Q", 1 << 40),      # wide_bytes: would be the length of a device-to-device copy
                          (64, "<i", 99),           # n_levels: indexes the level table
                          (76, "<i", -5),           # prefetch_every
                          (84, "<I", 7)):           # envs per workgroup: what the per-workgroup step counters are indexed by
        broken = sd["snapshot"].clone()
        broken[off:off + struct.calcsize(fmt)] = torch.tensor(list(struct.pack(fmt, bad)), dtype=torch.uint8, device=broken.device)
        with pytest.raises(ValueError):
            fresh.load_state_dict(dict(sd, snapshot=broken))
    fresh.load_state_dict(sd)                       # ... and the handle that refused them is intact
    fourth = run(fresh, 100, 150)
    for a, b in zip(first, fourth):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    fresh.close(); small.close()


def test_curriculum_set_config_between_launches_vs_oracle(amd, oracle_mod):
    """uavx_set_config: world scalars change between launches (box shrinks, sensing range / collider / speed
    limits move); the same schedule on the oracle gives the same masks, states and reset streams."""
    E, n = 1024, 6
    kw = dict(x_size=40.0, y_size=30.0, num_agents=n, d_sense=12.0)
    env = amd.BatchedMultiUAVWorld2D(E, seed=9, **kw)
    orc = oracle_mod.OracleMulti(num_envs=E, nthreads=8, **kw)
    env.reset()
    orc.reset_philox(9)
    schedule = {60: dict(x_size=24.0, y_size=24.0), 120: dict(d_sense=5.0, collider_radius=1.5),
                180: dict(max_speed=6.0, max_acceleration=9.0), 240: dict(x_size=60.0, y_size=20.0, d_sense=20)}
    rng = np.random.default_rng(5)
    oob_after_shrink = 0
    for t in range(300):
        if t in schedule:
            env.set_config(**schedule[t])
            orc.set_config(**schedule[t])
        d = orc.tgt - orc.loc
        dist = np.linalg.norm(d, axis=-1, keepdims=True)
        act = d / np.maximum(dist, 1e-9) * np.where(dist > 0.3, np.minimum(8.0, np.sqrt(4.0 * dist)), 0.0)
        act = act + rng.normal(0, 0.5, size=act.shape)
        obs_g, rew_g, done_g, info = env.step_ex(act, auto_reset="agent0_done", step_cap=80)
        obs_o, rew_o, done_o, rmask_o = orc.step_ex(act, reset_policy=1, step_cap=80, track_returns=True, seed=9)
        ctx = f"step {t}"
        np.testing.assert_array_equal(_np(info["reset_mask"]).astype(np.uint8), rmask_o, err_msg=ctx)
        np.testing.assert_array_equal(_np(done_g).astype(np.uint8), done_o, err_msg=ctx)
        if t == 60:
            oob_after_shrink = int(done_o.sum())
        st, ref = env.get_state(), orc.get_state()
        for key in ("loc", "vel", "tgt", "init_d", "prev_d", "flags"):
            np.testing.assert_array_equal(_np(st[key]), ref[key], err_msg=f"{ctx} {key}")
        assert obs_err(_np(obs_g), obs_o) <= TOL, ctx
        assert float((np.abs(_np(rew_g) - rew_o) / np.maximum(1.0, np.abs(rew_o))).max()) <= TOL, ctx
    assert oob_after_shrink > E, "shrinking the box must terminate the agents left outside (MUW:224-229)"
    assert env.map_diagonal_size == pytest.approx(math.hypot(60.0, 20.0)) and env.d_sense == 20
    assert float(env.action_space.high[0]) == 6.0
    with pytest.raises(TypeError):
        env.set_config(num_agents=3)
    with pytest.raises(ValueError):
        env.set_config(x_size=-1.0)
    assert env.x_size == 60.0, "a refused config must leave the mirror attributes alone"
    env.close()


def test_vector_env_surface_drives_one_launch_per_step(amd, oracle_mod):
    """UAVVectorEnv: gym.vector.VectorEnv call surface (spaces, reset, step_async/step_wait, set_attr) over the
    fused launch; trajectories equal the oracle's step_ex under the trainers' reset rule."""
    import torch
    from gym_uav_collision_avoidance_amd import UAVSingleVectorEnv, UAVVectorEnv
    E, n = 640, 4
    venv = UAVVectorEnv(E, num_agents=n, step_cap=50, polar=True, seed=21, x_size=30.0, y_size=30.0)
    assert len(venv) == E and venv.is_vector_env and venv.num_envs == E
    assert venv.single_observation_space.shape == (n, 10) and venv.observation_space.shape == (E, n, 10)
    assert venv.single_action_space.shape == (n, 2) and venv.action_space.shape == (E, n, 2)
    assert float(venv.action_space.high.max()) == 1.0 and venv.observation_space.dtype == np.float32
    assert venv.get_attr("d_sense") == (15,) * E
    orc = oracle_mod.OracleMulti(num_envs=E, num_agents=n, nthreads=8, x_size=30.0, y_size=30.0)
    obs = venv.reset(seed=21)
    orc.reset_philox(21)
    assert obs.shape == (E, n, 10) and obs.is_cuda
    rng = np.random.default_rng(2)
    resets = 0
    for t in range(150):
        a = rng.uniform(-1, 1, size=(E, n, 2)).astype(np.float32)
        a[..., 0] = np.minimum(a[..., 0], 0.2)
        venv.step_async(torch.from_numpy(a).to(venv.device))
        with pytest.raises(RuntimeError):
            venv.step_async(torch.from_numpy(a).to(venv.device))
        og, rg, dg, info = venv.step_wait()
        oo, ro, do, rm = orc.step_ex(a, action_mode=1, reset_policy=1, step_cap=50, track_returns=True, seed=21)
        np.testing.assert_array_equal(_np(info["reset_mask"]).astype(np.uint8), rm)
        np.testing.assert_array_equal(_np(dg).astype(np.uint8), do)
        assert obs_err(_np(og), oo) <= TOL and float(np.abs(_np(rg) - ro).max()) <= TOL
        resets += int(rm.sum())
    assert resets > E
    with pytest.raises(RuntimeError):
        venv.step_wait()
    venv.set_attr("d_sense", [7.0] * 3)
    assert venv.env.d_sense == 7.0
    with pytest.raises(ValueError):
        venv.set_attr("d_sense", [7.0, 8.0])
    assert venv.evaluation_summary()["episodes"] == int(orc.fin_counts[:, 0].sum())
    venv.close(); venv.close()
    sv = UAVSingleVectorEnv(512, seed=4, step_cap=40)
    o = sv.reset()
    assert o.shape == (512, 4) and sv.observation_space.shape == (512, 4) and sv.single_action_space.shape == (2,)
    ended = 0
    for t in range(90):
        o, r, d, info = sv.step(torch.from_numpy(rng.uniform(-12, 12, size=(512, 2)).astype(np.float32)).to(sv.device))
        ended += int(info["reset_mask"].sum().item())
    assert r.shape == (512,) and d.dtype == torch.bool and ended >= 512
    sv.close()


@pytest.mark.parametrize("E", [1536, 1531, 65536])    # (1 531: the last tile of the last workgroup holds three envs of eight)
def test_tile_pairs_equal_one_wavefront_workgroups(amd, monkeypatch, E):
    """8 UAVs: a launch that fills the wavefront slots once runs two one-wavefront tiles per workgroup (tiles_for in
    uavx_create; UAVX_TILES forces either).  Same results bit for bit, bare and fused, staging workgroups included."""
    import torch
    n = 8
    if E < 65536: monkeypatch.setenv("UAVX_TILES", "2")    # (65 536 x 8 picks the pairs by itself)
    a = amd.BatchedMultiUAVWorld2D(E, num_agents=n, seed=9)
    monkeypatch.setenv("UAVX_TILES", "1")
    b = amd.BatchedMultiUAVWorld2D(E, num_agents=n, seed=9)
    a.reset(); b.reset()
    g = torch.Generator(device="cpu").manual_seed(2)
    for t in range(24):
        act = (torch.rand((E, n, 2), generator=g) * 20 - 10).to(a.device)
        if t % 3 == 0:
            ra, rb = a.step(act), b.step(act)
        else:
            ra = a.step_ex(act, auto_reset="agent0_done", step_cap=7, track_returns=True)
            rb = b.step_ex(act, auto_reset="agent0_done", step_cap=7, track_returns=True)
            for k in ("reset_mask", "ended", "truncated"):
                assert torch.equal(ra[3][k], rb[3][k]), (t, k)
        for x, y in zip(ra[:3], rb[:3]):
            assert torch.equal(x, y), t
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    for k, v in a.episode_stats().items():
        assert torch.equal(v, b.episode_stats()[k]), k
    # a curriculum installed later moves the handle to the kernels with levels, which keep ONE tile per workgroup: the launch
    # must be laid out for those from then on (every env stepped, same results as the handle that never had the pairs)
    levels = [dict(x_size=40.0, y_size=40.0, collider_radius=1.0, d_sense=12.0, n_active=6),
              dict(x_size=55.0, y_size=50.0, collider_radius=1.0, d_sense=16.0)]
    for env in (a, b):
        env.set_curriculum(levels, lo=0, hi=1)
    oa, ob = a.reset(seed=5), b.reset(seed=5)
    assert torch.equal(oa, ob)
    for t in range(12):
        act = (torch.rand((E, n, 2), generator=g) * 20 - 10).to(a.device)
        ra = a.step_ex(act, auto_reset="agent0_done", step_cap=5, track_returns=True)
        rb = b.step_ex(act, auto_reset="agent0_done", step_cap=5, track_returns=True)
        for x, y in zip(ra[:3], rb[:3]):
            assert torch.equal(x, y), ("levels", t)
        assert torch.equal(ra[3]["reset_mask"], rb[3]["reset_mask"])
        seen = ra[3]["reset_mask"].clone() if t == 0 else seen | ra[3]["reset_mask"]
    assert bool(seen.all())                                # the cap re-initialised EVERY env: none was left out of a launch
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    a.close(); b.close()


@pytest.mark.parametrize("n", [3, 10, 12])
def test_workgroup_width_does_not_change_results(amd, monkeypatch, n):
    """The runtime-N kernels pick their wavefronts per workgroup from a measured table (three at 3 and 12 UAVs, two at 10);
    UAVX_GW forces one.  A launch-shape choice only: same trajectories bit for bit, bare and fused, resets included."""
    import torch
    E = 1000
    a = amd.BatchedMultiUAVWorld2D(E, num_agents=n, seed=4, x_size=30.0, y_size=30.0)
    monkeypatch.setenv("UAVX_GW", "1")
    b = amd.BatchedMultiUAVWorld2D(E, num_agents=n, seed=4, x_size=30.0, y_size=30.0)
    monkeypatch.setenv("UAVX_GW", "4")
    c = amd.BatchedMultiUAVWorld2D(E, num_agents=n, seed=4, x_size=30.0, y_size=30.0)
    envs = (a, b, c)
    obs = [e.reset() for e in envs]
    assert torch.equal(obs[0], obs[1]) and torch.equal(obs[0], obs[2])
    g = torch.Generator(device="cpu").manual_seed(8)
    for t in range(30):
        act = (torch.rand((E, n, 2), generator=g) * 20 - 10).to(a.device)
        if t % 4 == 0:
            rs = [e.step(act) for e in envs]
        else:
            rs = [e.step_ex(act, auto_reset="agent0_done", step_cap=6, track_returns=True) for e in envs]
            assert torch.equal(rs[0][3]["reset_mask"], rs[1][3]["reset_mask"]) and torch.equal(rs[0][3]["reset_mask"], rs[2][3]["reset_mask"])
        for k in range(3):
            assert torch.equal(rs[0][k], rs[1][k]) and torch.equal(rs[0][k], rs[2][k]), (t, k)
    st = [e.get_state() for e in envs]
    for k in st[0]:
        assert torch.equal(st[0][k], st[1][k]) and torch.equal(st[0][k], st[2][k]), k
    for e in envs:
        e.close()


def test_step_ex_defaults_equal_plain_step(amd):
    import torch
    E, n = 3000, 4
    a = amd.BatchedMultiUAVWorld2D(E, num_agents=n, seed=9)
    b = amd.BatchedMultiUAVWorld2D(E, num_agents=n, seed=9)
    a.reset(); b.reset()
    g = torch.Generator(device="cpu").manual_seed(2)
    for t in range(30):
        act = (torch.rand((E, n, 2), generator=g) * 20 - 10).to(a.device)
        o1, r1, d1, _ = a.step(act)
        o2, r2, d2, info = b.step_ex(act, track_returns=False)
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2) and not info["reset_mask"].any()
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    a.close(); b.close()


def test_device_replay_is_zero_copy_and_samples_true_transitions(amd):
    import torch
    from gym_uav_collision_avoidance_amd.replay import DeviceReplay
    E, n, T = 512, 4, 16
    env = amd.BatchedMultiUAVWorld2D(E, num_agents=n, seed=4, x_size=26.0, y_size=26.0)
    mem = DeviceReplay(env, horizon=T)
    mem.begin(env.reset())
    g = torch.Generator(device=env.device).manual_seed(0)
    tape = []
    for t in range(40):  # wraps the ring more than twice
        s = mem.state.clone()
        slot = mem.action_slot()
        slot.copy_(torch.rand((E, n, 2), generator=g, device=env.device) * 2 - 1)  # "policy output" written in place
        obs, rew, done, info = mem.step(polar=True, auto_reset="agent0_done", step_cap=12)
        assert obs.data_ptr() == mem.obs[(t + 1) % (T + 1)].data_ptr()  # the kernel wrote straight into the ring
        tape.append((s, slot.clone(), rew.clone(), obs.clone(), done.clone(), info["reset_mask"].clone()))
    assert len(mem) == T * E * n
    S, A, R, S1, M = mem.sample(20000, generator=g)
    # every sampled row must be one of the true (non-reset) transitions of the last T steps
    keys = {}
    for (s, a, r, s1, d, rm) in tape[-T:]:
        ok = ~rm
        rows = torch.cat([s[ok].reshape(-1, 10), a[ok].reshape(-1, 2), r[ok].reshape(-1, 1), s1[ok].reshape(-1, 10),
                          1 - d[ok].reshape(-1, 1).float()], dim=1)
        for row in rows.cpu().numpy().round(6):
            keys[row.tobytes()] = True
    got = torch.cat([S, A, R[:, None], S1, M[:, None]], dim=1).cpu().numpy().round(6)
    miss = sum(1 for row in got if row.tobytes() not in keys)
    assert miss == 0, f"{miss} sampled rows are not recorded transitions"
    assert S.shape == (20000, 10) and A.shape == (20000, 2) and M.min() >= 0 and M.max() <= 1
    env.close()


def test_evaluation_harness_statistics_match_oracle_loop(amd, oracle_mod):
    """evaluate_policy with an observation-driven controller vs the same closed loop run on the oracle:
    the two loops see observations that differ by <=1e-5, so trajectories are not bit-identical, but the
    SR/CR statistics of test_sac_multi_score.py:63-68 must agree closely."""
    import torch
    from gym_uav_collision_avoidance_amd.evaluate import evaluate_policy, seek_policy
    n, E, T = 4, 400, 900
    pol = seek_policy()
    got = evaluate_policy(pol, n, episodes=E, max_steps=T, evaluate=True, seed=3)
    orc = oracle_mod.OracleMulti(num_envs=E, num_agents=n, nthreads=8)
    orc.reset_philox(3)
    obs = orc.observe()
    ended = np.zeros(E, bool); reach = np.zeros(E); coll = np.zeros(E); total = np.zeros(E)
    for t in range(T):
        a = pol(torch.from_numpy(obs.astype(np.float32))).numpy()
        obs, rew, done, _ = orc.step_ex(a, evaluate=True, action_mode=1)
        total += np.where(~ended, (rew * (1 - done)).sum(axis=1), 0.0)
        newly = ~ended & (done.all(axis=1) | (t + 1 >= T))
        reach[newly] = orc.counters[newly, 1]; coll[newly] = orc.counters[newly, 2]
        ended |= newly
        if ended.all():
            break
    sr, cr = reach.sum() / (n * E), coll.sum() / (n * E)
    assert got["success_rate"] > 0.8, got
    assert abs(got["success_rate"] - sr) < 0.02 and abs(got["collision_rate"] - cr) < 0.02, (got, sr, cr)
    assert abs(got["avg_score"] - total.sum() / (n * E)) < 0.05 * max(1.0, abs(total.sum() / (n * E))), (got, total.sum() / (n * E))


def test_circular_scenario_and_agent_sweep(amd):
    import torch
    from gym_uav_collision_avoidance_amd.evaluate import seek_policy, sweep_num_agents
    env = amd.BatchedMultiUAVWorld2D(3, num_agents=6)
    obs = env.reset_circular()
    st = env.get_state()
    th = 2 * np.arange(6) * np.pi / 6
    np.testing.assert_allclose(_np(st["loc"])[1], 20 * np.stack([np.cos(th), np.sin(th)], 1), atol=1e-5)   # MUW:160
    np.testing.assert_allclose(_np(st["tgt"])[2], -23 * np.stack([np.cos(th), np.sin(th)], 1), atol=1e-5)  # MUW:161
    np.testing.assert_allclose(_np(st["init_d"]), 43.0, atol=1e-4)
    assert obs.shape == (3, 6, 10) and int(env.metrics()[:, :3].abs().sum()) == 0
    env.close()
    res = sweep_num_agents(seek_policy(), agent_counts=(1, 2, 5, 12), episodes=64, max_steps=700, circular=False, seed=1)
    assert [r["num_agents"] for r in res] == [1, 2, 5, 12]
    assert res[0]["success_rate"] > 0.95 and res[0]["collision_rate"] == 0.0   # a lone UAV always arrives
    assert all(0.0 <= r["success_rate"] <= 1.0 and r["collision_rate"] >= 0.0 for r in res)


def test_trajectory_rollout_matches_the_reference_circular_episode(amd):
    """evaluate.rollout_trajectories (the plotting scenario, test_sac_multi_plot_trajectory.py:43-76) replaying the actions
    of the reference's recorded reset(circular=True) episodes: every recorded location is the reference's float64 location
    bit for bit, agents stop being recorded exactly when their done flag is set, the episode ends at all(dones)."""
    import os
    import torch
    from gym_uav_collision_avoidance_amd.evaluate import rollout_trajectories
    for n in (4, 6):
        fx = np.load(os.path.join(os.path.dirname(__file__), "golden", f"crafted_circular_n{n}.npz"))
        acts = torch.from_numpy(fx["actions"]).to("cuda")          # [T, N, 2] float64 velocity commands
        E, T = 3, acts.shape[0]
        step = {"t": 0}

        def replay(obs):
            a = acts[min(step["t"], T - 1)]                         # float64 commands, passed through unchanged
            step["t"] += 1
            return a[None].expand(E, n, 2)

        out = rollout_trajectories(replay, n, episodes=E, max_steps=T, circular=True, polar=False)
        pos, valid, length = _np(out["positions"]), _np(out["valid"]), _np(out["length"])
        want = np.concatenate([fx["init_loc"][None], fx["loc"][:-1]], axis=0)          # location BEFORE step t
        done_before = np.concatenate([np.zeros((1, n), bool), (fx["flags"][:-1] & 1) != 0], axis=0)
        all_done = np.where(fx["done"].all(axis=1))[0]
        L = int(all_done[0]) + 1 if len(all_done) else T
        assert (length == L).all(), (length, L)
        assert out["positions"].dtype == torch.float64
        for e in range(E):
            np.testing.assert_array_equal(pos[:L, e], want[:L])
            np.testing.assert_array_equal(valid[:L, e], ~done_before[:L])
        assert not valid[L:].any()
        np.testing.assert_array_equal(_np(out["depots"])[0], fx["init_loc"])
        np.testing.assert_array_equal(_np(out["goals"])[0], fx["init_tgt"])


def test_reference_checkpoint_layout_and_batched_policy(amd, tmp_path):
    """A file with the reference's SAC.save_checkpoint layout (sac.py:108-112) loads into the batched actor."""
    import torch
    from gym_uav_collision_avoidance_amd.policy import GaussianPolicy, load_reference_checkpoint
    torch.manual_seed(0)
    ref_like = GaussianPolicy()
    path = tmp_path / "weights.chpt"
    torch.save({"policy_state_dict": ref_like.state_dict(), "critic_state_dict": {}, "critic_target_state_dict": {},
                "critic_optimizer_state_dict": {}, "policy_optimizer_state_dict": {}}, path)
    pol = load_reference_checkpoint(str(path), device="cuda")
    env = amd.BatchedMultiUAVWorld2D(2048, num_agents=4, seed=2)
    obs = env.reset()
    for _ in range(20):
        a = pol.act(obs, evaluate=True)
        assert a.shape == (2048, 4, 2) and float(a.abs().max()) <= 1.0
        obs, rew, done, info = env.step_ex(a, polar=True, auto_reset="agent0_done", step_cap=1500)
    want = torch.tanh(ref_like.to("cuda")(obs)[0])
    assert torch.allclose(pol.act(obs), want, atol=1e-6)
    env.close()


def test_rgb_array_render(amd):
    from gym_uav_collision_avoidance_amd.envs import MultiUAVWorld2D
    np.random.seed(1)
    env = MultiUAVWorld2D(num_agents=3)
    env.reset()
    assert env.render() is None                      # "human": no display, no-op
    img = env.render(mode="rgb_array")
    assert img.shape == (800, 800, 3) and img.dtype == np.uint8
    assert (img != 255).any() and (img == 255).mean() > 0.8
    env.close()


@pytest.mark.parametrize("polar,cap", [(False, 0), (True, 60), (True, 0)])
def test_uw_step_ex_auto_reset_vs_oracle(amd, oracle_mod, polar, cap):
    """UAVWorld2D: fused conversion / next-step auto-reset / episode statistics vs the oracle restatement."""
    E = 4000
    env = amd.BatchedUAVWorld2D(E, seed=31, env_offset=9)
    orc = oracle_mod.OracleSingle(num_envs=E, nthreads=8)
    env.reset()
    orc.reset_philox(31, env_offset=9)
    rng = np.random.default_rng(6)
    resets = 0
    for t in range(220):
        d = orc.tgt - orc.loc
        dist = np.linalg.norm(d, axis=-1, keepdims=True)
        cmd = d / np.maximum(dist, 1e-9) * np.minimum(9.0, np.sqrt(4.0 * dist)) + rng.normal(0, 0.5, size=d.shape)
        if polar:  # express the same command as a policy output in [-1,1]^2 (float32)
            sp = np.clip(np.linalg.norm(cmd, axis=-1) / 12.0, 0, 1)
            act = np.stack([sp * 2 - 1, np.arctan2(cmd[:, 1], cmd[:, 0]) / np.pi], axis=-1).astype(np.float32)
        else:
            act = cmd.astype(np.float32) if t % 2 else cmd
        og, rg, dg, info = env.step_ex(act, polar=polar, auto_reset=True, step_cap=cap)
        oo, ro, do, io, rm = orc.step_ex(act, polar=polar, auto_reset=True, step_cap=cap, seed=31, env_offset=9)
        ctx = f"step {t}"
        np.testing.assert_array_equal(_np(info["reset_mask"]).astype(np.uint8), rm, err_msg=ctx)
        np.testing.assert_array_equal(_np(dg).astype(np.uint8), do, err_msg=ctx)
        # ended / truncated of the ending call: the oracle's `pending` is "ended"; truncated = ended without a done
        stepped = rm == 0
        np.testing.assert_array_equal(_np(info["ended"]).astype(np.uint8), orc.pending * stepped, err_msg=ctx)
        np.testing.assert_array_equal(_np(info["truncated"]).astype(np.uint8), orc.pending * stepped * (do == 0), err_msg=ctx)
        resets += int(rm.sum())
        st = env.get_state()
        np.testing.assert_array_equal(_np(st["loc"]), orc.loc.astype(np.float32), err_msg=ctx)
        np.testing.assert_array_equal(_np(st["vel"]), orc.vel, err_msg=ctx)
        np.testing.assert_array_equal(_np(st["tgt"]), orc.tgt.astype(np.float32), err_msg=ctx)
        np.testing.assert_array_equal(_np(st["counters"])[:, 0], orc.steps, err_msg=ctx)
        np.testing.assert_array_equal(_np(st["counters"])[:, 1], orc.episode, err_msg=ctx)
        assert obs_err(_np(og), oo, (1, 3)) <= TOL, ctx
        tol_r = np.maximum(TOL, np.spacing(np.abs(ro).astype(np.float32)).astype(np.float64))
        assert (np.abs(_np(rg) - ro) <= tol_r).all(), ctx
        np.testing.assert_array_equal(_np(info["distance"]), io.astype(np.float32), err_msg=ctx)
    assert resets > E // 10
    stats = {k: _np(v) for k, v in env.episode_stats().items()}
    np.testing.assert_array_equal(stats["episodes"], orc.fin_counts[:, 0])
    np.testing.assert_array_equal(stats["steps"], orc.fin_counts[:, 1])
    np.testing.assert_array_equal(stats["reached"], orc.fin_counts[:, 2])
    np.testing.assert_allclose(stats["returns"], orc.fin_return, rtol=1e-5, atol=0.05)
    assert stats["reached"].sum() > 0
    env.reset()
    orc.reset_philox(31, env_offset=9)
    np.testing.assert_array_equal(_np(env.episode_stats()["episodes"]), orc.fin_counts[:, 0])
    np.testing.assert_array_equal(_np(env.episode_stats()["reached"]), orc.fin_counts[:, 2])
    env.close()


def test_split_batch_chains_equal_the_undivided_batch(amd):
    """sharding.SplitBatch: the batch as two handles with a stream each (the double-buffered trainer layout), stepped as
    independent fused chains without any host synchronisation in between -- every env must come out exactly as in the
    undivided batch (auto-resets included: Philox streams are keyed by global env id)."""
    import torch
    from gym_uav_collision_avoidance_amd.sharding import SplitBatch
    E, N, T = 1001, 4, 90            # 501 + 500 envs
    whole = amd.BatchedMultiUAVWorld2D(E, num_agents=N, seed=5, x_size=30.0, y_size=30.0)
    sb = SplitBatch(amd.BatchedMultiUAVWorld2D, E, parts=2, num_agents=N, seed=5, x_size=30.0, y_size=30.0)
    assert [c for _, c in sb.ranges] == [501, 500]
    g = torch.Generator(device=whole.device).manual_seed(3)
    acts = torch.rand((T, E, N, 2), generator=g, device=whole.device) * 2 - 1
    kw = dict(polar=True, auto_reset="agent0_done", step_cap=25, track_returns=True)
    o_w = whole.reset(seed=5).clone()
    o_p = sb.reset(seed=5)
    assert torch.equal(torch.cat(o_p), o_w)
    outs = [[] for _ in sb.envs]
    for k, env in enumerate(sb.envs):            # chain k runs ahead on its own stream, nothing waits for the other
        off, cnt = sb.ranges[k]
        with torch.cuda.stream(sb.streams[k]):
            for t in range(T):
                o, r, d, info = env.step_ex(acts[t, off:off + cnt], **kw)
                outs[k].append((o.clone(), r.clone(), d.clone(), info["reset_mask"].clone(), info["truncated"].clone()))
    sb.synchronize()
    for t in range(T):
        o, r, d, info = whole.step_ex(acts[t], **kw)
        for j, name in enumerate(("obs", "rew", "done")):
            assert torch.equal(torch.cat([outs[k][t][j] for k in range(2)]), (o, r, d)[j]), (t, name)
        assert torch.equal(torch.cat([outs[k][t][3] for k in range(2)]), info["reset_mask"]), t
        assert torch.equal(torch.cat([outs[k][t][4] for k in range(2)]), info["truncated"]), t
    assert torch.equal(sb.metrics(), whole.metrics())
    sb.close(); whole.close()
