"""Parity tests proper (need a real MI355X): the HIP path, called through the C ABI via the Python
host layer, against (a) the committed fixtures generated from the reference and (b) the CPU oracle
on seeded random batches.  Bar (BASELINE.json north_star): done/collision masks, flags, counters and
float32 positions BIT-EXACT; float32 observations / rewards within 1e-5 (angle features measured on
the circle)."""
import os

import numpy as np
import pytest

from golden_util import UW_ANGLE_COLS, fixture_names, load_fixture, obs_err, tie_agents

pytestmark = pytest.mark.gpu

TOL = 1e-5
MULTI = fixture_names("multi")
UW = fixture_names("uw")


@pytest.fixture(scope="module")
def amd():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    import gym_uav_collision_avoidance_amd as pkg
    return pkg


def _np(t):
    return t.detach().cpu().numpy()


def _ctor(cfg):
    return {k: v for k, v in cfg.items() if k != "tau"}  # tau is fixed at 0.02 in the ctor (MUW:26)


def _check_multi_state(env, ref, ctx, vel_exact=True):
    st = {k: _np(v) for k, v in env.get_state().items()}
    np.testing.assert_array_equal(st["flags"], ref["flags"], err_msg=ctx + " flags")
    np.testing.assert_array_equal(st["loc"], ref["loc"].astype(np.float32), err_msg=ctx + " loc")
    np.testing.assert_array_equal(st["prev_d"], ref["prev_d"].astype(np.float32), err_msg=ctx + " prev_d")
    if vel_exact:
        np.testing.assert_array_equal(st["vel"], ref["vel"], err_msg=ctx + " vel")
    else:
        np.testing.assert_allclose(st["vel"], ref["vel"], rtol=1e-12, atol=1e-15, err_msg=ctx + " vel")
    np.testing.assert_array_equal(st["counters"][:, :3], ref["counters"], err_msg=ctx + " counters")


@pytest.mark.parametrize("name", [n for n in MULTI if "circular" not in n])
def test_multi_fixture_replay(amd, name):
    """Replays the reference's recorded rollouts (float32 positions) through the HIP step kernel."""
    data, meta = load_fixture(name)
    cfg = meta["cfg"]
    n = cfg["num_agents"]
    env = amd.BatchedMultiUAVWorld2D(1, **_ctor(cfg))
    env.set_state(loc=data["init_loc"][None], vel=data["init_vel"][None], tgt=data["init_tgt"][None],
                  init_d=data["init_init_d"][None], prev_d=data["init_prev_d"][None], flags=data["init_flags"][None],
                  counters=np.concatenate([data["init_counters"], [0]])[None])
    worst_obs = worst_rew = 0.0
    for t in range(data["actions"].shape[0]):
        obs, rew, done, info = env.step(data["actions"][t][None], evaluate=bool(data["evaluate"][t]))
        ctx = f"{name} step {t}"
        np.testing.assert_array_equal(_np(done)[0].astype(np.uint8), data["done"][t], err_msg=ctx + " done")
        ref = dict(flags=data["flags"][t][None], loc=data["loc"][t][None], prev_d=data["prev_d"][t][None],
                   vel=data["vel"][t][None], counters=data["counters"][t][None])
        _check_multi_state(env, ref, ctx)
        got, want = _np(obs)[0].astype(np.float64), data["obs"][t].copy()
        ties = tie_agents(data["loc"][t], cfg["d_sense"], False)
        got[ties, 4:] = 0
        want[ties, 4:] = 0
        worst_obs = max(worst_obs, obs_err(got, want))
        worst_rew = max(worst_rew, float(np.abs(_np(rew)[0] - data["rew"][t]).max()))
        assert worst_obs <= TOL and worst_rew <= TOL, f"{ctx}: obs err {worst_obs:.3g} rew err {worst_rew:.3g}"
        assert info == {"distance": 0}
    env.close()
    assert n == env.num_agents


@pytest.mark.parametrize("name", [n for n in MULTI if "circular" in n])
def test_multi_circular_fixture_float32_positions(amd, name):
    """The float32 fast path fed with the circular layout (reset_circular(float64=False)): the reference holds
    float64 positions in this scenario (MUW:157-163), so positions are compared to float32 resolution and masks
    within a step.  The exact counterpart is test_multi_circular_fixture_float64_mode_exact below."""
    data, meta = load_fixture(name)
    cfg = meta["cfg"]
    env = amd.BatchedMultiUAVWorld2D(1, **_ctor(cfg))
    env.set_state(loc=data["init_loc"][None], vel=data["init_vel"][None], tgt=data["init_tgt"][None],
                  init_d=data["init_init_d"][None], prev_d=data["init_prev_d"][None], flags=data["init_flags"][None])
    mism = 0
    for t in range(data["actions"].shape[0]):
        obs, rew, done, _ = env.step(data["actions"][t][None])
        st = env.get_state()
        np.testing.assert_allclose(_np(st["loc"])[0], data["loc"][t], atol=2e-4, err_msg=f"{name} step {t}")
        mism += int((_np(done)[0].astype(np.uint8) != data["done"][t]).sum())
        ties = tie_agents(data["loc"][t], cfg["d_sense"], True) | tie_agents(_np(st["loc"])[0], cfg["d_sense"], False)
        got, want = _np(obs)[0].astype(np.float64), data["obs"][t].copy()
        got[ties, 4:] = 0
        want[ties, 4:] = 0
        # symmetric layout: neighbour pairs are near-ties, so only the ego/target columns are pinned tightly
        assert obs_err(got[:, :4], want[:, :4], angle_cols=(1, 3)) < 1e-4, f"{name} step {t}"
    assert mism <= 2, f"{name}: {mism} done-mask mismatches (success fires within a step of the reference)"
    np.testing.assert_array_equal(_np(env.metrics())[0, 1], data["counters"][-1][1])
    env.close()


@pytest.mark.parametrize("name", [n for n in MULTI if "circular" in n])
def test_multi_circular_fixture_float64_mode_exact(amd, name):
    """The same recorded circular episodes in the library's float64-position mode (uavx_set_position_mode): done
    masks, float64 positions / prev distances, velocities, flags and counters bit for bit with the reference."""
    data, meta = load_fixture(name)
    cfg = meta["cfg"]
    env = amd.BatchedMultiUAVWorld2D(1, **_ctor(cfg))
    env.set_state(vel=data["init_vel"][None], flags=data["init_flags"][None],
                  counters=np.concatenate([data["init_counters"], [0]])[None])
    env.set_state_f64(loc=data["init_loc"][None], tgt=data["init_tgt"][None], init_d=data["init_init_d"][None],
                      prev_d=data["init_prev_d"][None])
    assert env.position_mode == "float64"
    worst_obs = worst_rew = 0.0
    for t in range(data["actions"].shape[0]):
        obs, rew, done, _ = env.step(data["actions"][t][None], evaluate=bool(data["evaluate"][t]))
        ctx = f"{name} step {t}"
        np.testing.assert_array_equal(_np(done)[0].astype(np.uint8), data["done"][t], err_msg=ctx + " done")
        st, st64 = env.get_state(), env.get_state_f64()
        np.testing.assert_array_equal(_np(st64["loc"])[0], data["loc"][t], err_msg=ctx + " loc")
        np.testing.assert_array_equal(_np(st64["prev_d"])[0], data["prev_d"][t], err_msg=ctx + " prev_d")
        np.testing.assert_array_equal(_np(st["vel"])[0], data["vel"][t], err_msg=ctx + " vel")
        np.testing.assert_array_equal(_np(st["flags"])[0], data["flags"][t], err_msg=ctx + " flags")
        np.testing.assert_array_equal(_np(st["counters"])[0, :3], data["counters"][t], err_msg=ctx + " counters")
        np.testing.assert_array_equal(_np(st["loc"])[0], data["loc"][t].astype(np.float32))   # float32 view = rounded
        got, want = _np(obs)[0].astype(np.float64), data["obs"][t].copy()
        ties = tie_agents(data["loc"][t], cfg["d_sense"], True)   # exact float64 ties: argsort order is platform dependent
        got[ties, 4:] = 0
        want[ties, 4:] = 0
        worst_obs = max(worst_obs, obs_err(got, want))
        worst_rew = max(worst_rew, float(np.abs(_np(rew)[0] - data["rew"][t]).max()))
        assert worst_obs <= TOL and worst_rew <= TOL, f"{ctx}: obs err {worst_obs:.3g} rew err {worst_rew:.3g}"
    env.close()


@pytest.mark.parametrize("n,E", [(3, 400), (7, 150), (24, 40)])
def test_float64_position_mode_vs_oracle_and_back(amd, oracle_mod, n, E):
    """Random batches: reset (float32) -> a few float32 steps -> float64 mode (exact widening) -> float64 steps incl.
    polar actions and episode-return tracking -> back to float32 (rounding, prev_distance override) -> float32 steps,
    the oracle following the same schedule through its per-env f64pos flag."""
    kw = dict(x_size=30.0, y_size=24.0, num_agents=n, d_sense=9.0)
    if n >= 24:
        kw.update(x_size=60.0, y_size=60.0)
    env = amd.BatchedMultiUAVWorld2D(E, seed=77, **kw)
    orc = oracle_mod.OracleMulti(num_envs=E, nthreads=8, **kw)
    env.reset(); orc.reset_philox(77)
    rng = np.random.default_rng(n)

    def seek():
        d = orc.tgt - orc.loc
        act = d * rng.uniform(0.3, 2.0, size=(E, n, 1)) + rng.normal(0, 0.4, size=d.shape)
        wild = rng.random((E, n, 1)) < 0.1
        return np.where(wild, rng.uniform(-10, 10, size=d.shape), act)

    def compare(obs_g, rew_g, done_g, obs_o, rew_o, done_o, ctx, wide):
        np.testing.assert_array_equal(_np(done_g).astype(np.uint8), done_o, err_msg=ctx)
        st = env.get_state()
        np.testing.assert_array_equal(_np(st["vel"]), orc.vel, err_msg=ctx + " vel")
        np.testing.assert_array_equal(_np(st["flags"]), orc.flags, err_msg=ctx + " flags")
        np.testing.assert_array_equal(_np(st["counters"])[:, :3], orc.counters[:, :3].astype(np.int32), err_msg=ctx)
        if wide:
            s64 = env.get_state_f64()
            for k in ("loc", "tgt", "init_d", "prev_d"):
                np.testing.assert_array_equal(_np(s64[k]), getattr(orc, k), err_msg=f"{ctx} {k}")
        else:
            for k in ("loc", "tgt", "init_d", "prev_d"):
                np.testing.assert_array_equal(_np(st[k]), getattr(orc, k).astype(np.float32), err_msg=f"{ctx} {k}")
        assert obs_err(_np(obs_g), obs_o) <= TOL, ctx
        assert (np.abs(_np(rew_g) - rew_o) <= TOL * np.maximum(1.0, np.abs(rew_o))).all(), ctx

    for t in range(15):
        a = seek()
        compare(*env.step(a)[:3], *orc.step(a), f"f32 warm-up {t}", False)
    env.set_position_mode("float64")
    orc.f64pos[:] = 1
    assert env.position_mode == "float64"
    assert obs_err(_np(env.observe()), orc.observe()) <= TOL
    for t in range(120):
        if t % 2:
            a = rng.uniform(-1, 1, size=(E, n, 2)).astype(np.float32)
            og, rg, dg, info = env.step_ex(a, polar=True, evaluate=bool(t % 3 == 0), track_returns=True)
            assert not bool(info["reset_mask"].any())
            oo, ro, do, _ = orc.step_ex(a, action_mode=1, evaluate=bool(t % 3 == 0), track_returns=True)
        else:
            a = seek()
            og, rg, dg, _ = env.step(a, evaluate=bool(t % 3 == 0))
            oo, ro, do = orc.step(a, evaluate=bool(t % 3 == 0))
        compare(og, rg, dg, oo, ro, do, f"f64 step {t}", True)
    assert int(orc.counters[:, 1].sum()) > 0 and int(orc.counters[:, 2].sum()) > 0, "scenario too tame"
    # what float64 mode refuses
    with pytest.raises(RuntimeError):
        env.reset(mask=np.arange(E) % 2 == 0)
    with pytest.raises(RuntimeError):
        env.step_ex(seek(), auto_reset="agent0_done")
    with pytest.raises(RuntimeError):
        env.step_k(np.stack([seek()] * 4))
    # back to float32: values round, prev_distance keeps its (rounded) value through the override slot
    env.set_position_mode("float32")
    assert env.position_mode == "float32"
    orc.f64pos[:] = 0
    for k in ("loc", "tgt", "init_d", "prev_d"):
        getattr(orc, k)[...] = getattr(orc, k).astype(np.float32)
    for t in range(25):
        a = seek()
        compare(*env.step(a)[:3], *orc.step(a), f"f32 again {t}", False)
    # reset() always returns to float32 arrays (MUW:126)
    env.set_position_mode("float64")
    env.reset(); orc.reset_philox(77)
    assert env.position_mode == "float32"
    stats = env.episode_stats()   # the explicit reset folded the episode, returns tracked by the float64 step_ex calls
    np.testing.assert_array_equal(_np(stats["episodes"]), orc.fin_counts[:, 0])
    np.testing.assert_array_equal(_np(stats["steps"]), orc.fin_counts[:, 1])
    np.testing.assert_allclose(_np(stats["return0"]), orc.fin_returns[:, 0], atol=2e-3, rtol=1e-5)
    np.testing.assert_allclose(_np(stats["score"]), orc.fin_returns[:, 1], atol=2e-3, rtol=1e-5)
    a = seek()
    compare(*env.step(a)[:3], *orc.step(a), "after reset", False)
    env.close()


# (1, 4096) is BASELINE configs[1] literally: 4 096 parallel envs x 1 UAV
# (3, 6, 7, 11, 12, 24 run three wavefronts per workgroup, 9, 10, 15, 20 two: pick_group_waves' measured table)
N_CASES = [(1, 4096), (2, 2048), (3, 1000), (4, 4096), (5, 777), (6, 700), (7, 500), (8, 1024), (10, 400), (11, 300), (12, 300),
           (15, 200), (16, 256), (20, 150), (24, 130), (33, 64), (64, 70)]


@pytest.mark.parametrize("n,E", N_CASES)
def test_multi_oracle_random_batch(amd, oracle_mod, n, E):
    """Seeded batch: device Philox reset == oracle Philox reset bit for bit, then T steps of random
    actions compared step by step (small boxes so collisions, OOB and finishes all occur)."""
    import torch
    kw = dict(x_size=30.0, y_size=24.0, num_agents=n, d_sense=9.0, collider_radius=1.0)
    if n >= 24:
        kw.update(x_size=60.0, y_size=60.0)
    env = amd.BatchedMultiUAVWorld2D(E, seed=1234 + n, env_offset=10 * n, **kw)
    orc = oracle_mod.OracleMulti(num_envs=E, nthreads=8, **kw)
    for round_ in range(2):  # second round re-resets a masked half: episode counter advances
        mask = None if round_ == 0 else (np.arange(E) % 2 == 0)
        obs_g = env.reset(mask=None if mask is None else torch.from_numpy(mask).to(env.device))
        orc.reset_philox(1234 + n, mask=mask, env_offset=10 * n)
        ref = orc.get_state()
        _check_multi_state(env, dict(flags=ref["flags"], loc=ref["loc"], prev_d=ref["prev_d"], vel=ref["vel"],
                                     counters=ref["counters"][:, :3]), f"n={n} reset {round_}")
        st = env.get_state()
        np.testing.assert_array_equal(_np(st["tgt"]), ref["tgt"])
        np.testing.assert_array_equal(_np(st["init_d"]), ref["init_d"])
        np.testing.assert_array_equal(_np(st["counters"])[:, 3], ref["counters"][:, 3])
        assert obs_err(_np(obs_g), orc.observe()) <= TOL
        rng = np.random.default_rng(n * 100 + round_)
        T = 60
        for t in range(T):
            if t % 3 == 0:  # float32 box actions / float64 goal-seeking actions alternate
                act = rng.uniform(-10, 10, size=(E, n, 2)).astype(np.float32)
            else:
                d = orc.tgt - orc.loc
                act = d * rng.uniform(0.2, 3.0, size=(E, n, 1))
                act += rng.normal(0, 0.5, size=act.shape)
            ev = bool(t % 5 == 4)
            obs_g, rew_g, done_g, _ = env.step(act, evaluate=ev)
            obs_o, rew_o, done_o = orc.step(act, evaluate=ev)
            ctx = f"n={n} round {round_} step {t}"
            np.testing.assert_array_equal(_np(done_g).astype(np.uint8), done_o, err_msg=ctx)
            ref = orc.get_state()
            _check_multi_state(env, dict(flags=ref["flags"], loc=ref["loc"], prev_d=ref["prev_d"], vel=ref["vel"],
                                         counters=ref["counters"][:, :3]), ctx)
            e_obs = obs_err(_np(obs_g), obs_o)
            e_rew = float(np.abs(_np(rew_g) - rew_o).max())
            assert e_obs <= TOL and e_rew <= TOL, f"{ctx}: obs err {e_obs:.3g}, rew err {e_rew:.3g}"
    c = orc.counters
    if 4 <= n <= 16:
        assert c[:, 2].sum() > 0, "test must exercise hard collisions"
    env.close()


def test_multi_goal_reaching_batch(amd, oracle_mod):
    """Braking controller on 2048 envs: most agents finish; exercises finish(), +10 stickiness and
    the float64 velocity rescale (AG:38-42) at scale."""
    E, n = 2048, 4
    env = amd.BatchedMultiUAVWorld2D(E, num_agents=n, seed=7)
    orc = oracle_mod.OracleMulti(num_envs=E, num_agents=n, nthreads=8)
    env.reset()
    orc.reset_philox(7)
    for t in range(700):
        d = orc.tgt - orc.loc
        dist = np.linalg.norm(d, axis=-1, keepdims=True)
        sp = np.where(dist > 0.3, np.minimum(8.0, np.sqrt(4.0 * dist)), 0.0)
        act = d / np.maximum(dist, 1e-9) * sp
        obs_g, rew_g, done_g, _ = env.step(act)
        obs_o, rew_o, done_o = orc.step(act)
        if t % 25 == 0 or t > 650:
            np.testing.assert_array_equal(_np(done_g).astype(np.uint8), done_o, err_msg=f"step {t}")
            ref = orc.get_state()
            _check_multi_state(env, dict(flags=ref["flags"], loc=ref["loc"], prev_d=ref["prev_d"], vel=ref["vel"],
                                         counters=ref["counters"][:, :3]), f"step {t}")
            assert obs_err(_np(obs_g), obs_o) <= TOL
            assert float(np.abs(_np(rew_g) - rew_o).max()) <= TOL
    assert orc.counters[:, 1].sum() > E * n * 0.5, "controller should make most agents reach their target"
    env.close()


@pytest.mark.parametrize("n,E", [(4, 1500), (24, 333), (13, 200), (7, 77)])   # compile-time N, multi-wavefront workgroups, runtime N
def test_step_k_equals_k_steps(amd, n, E):
    import torch
    K = 9
    a = amd.BatchedMultiUAVWorld2D(E, num_agents=n, seed=3)
    b = amd.BatchedMultiUAVWorld2D(E, num_agents=n, seed=3)
    a.reset(); b.reset()
    g = torch.Generator(device="cpu").manual_seed(5)
    tape = (torch.rand((K, E, n, 2), generator=g) * 20 - 10).to(a.device)
    obs_k, rew_k, done_k, _ = a.step_k(tape, tape_out=True)
    for k in range(K):
        obs, rew, done, _ = b.step(tape[k])
        assert torch.equal(obs, obs_k[k]) and torch.equal(rew, rew_k[k]) and torch.equal(done, done_k[k])
    sa, sb = a.get_state(), b.get_state()
    for key in sa:
        assert torch.equal(sa[key], sb[key]), key
    last = a.step_k(tape, tape_out=False)
    for k in range(K):
        cur = b.step(tape[k])
    assert torch.equal(last[0], cur[0]) and torch.equal(last[1], cur[1]) and torch.equal(last[2], cur[2])
    a.close(); b.close()


def test_full_size_properties(amd):
    """BASELINE.json's headline size (65 536 envs x 4 UAVs): size-independent properties —
    determinism, env independence (any sub-batch evolves identically inside the full batch) and
    shard independence (env_offset keys the Philox streams by global env id).  GPU against GPU; the comparison with the
    oracle at this size is test_full_size_oracle_differential below."""
    import torch
    E, n, T = 65536, 4, 40
    g = torch.Generator(device="cpu").manual_seed(11)
    tape = (torch.rand((T, E, n, 2), generator=g) * 20 - 10)

    def rollout(num_envs, env_offset, acts):
        env = amd.BatchedMultiUAVWorld2D(num_envs, num_agents=n, seed=99, env_offset=env_offset)
        env.reset()
        acts = acts.to(env.device)
        rsum = torch.zeros((num_envs, n), device=env.device, dtype=torch.float64)
        dsum = torch.zeros((num_envs, n), device=env.device, dtype=torch.int64)
        for t in range(acts.shape[0]):
            obs, rew, done, _ = env.step(acts[t])
            rsum += rew
            dsum += done
        out = (obs.clone(), rsum, dsum, {k: v.clone() for k, v in env.get_state().items()})
        env.close()
        return out

    full = rollout(E, 0, tape)
    again = rollout(E, 0, tape)
    for x, y in zip(full[:3], again[:3]):
        assert torch.equal(x, y), "same seed, same actions -> identical buffers"
    lo, hi = 32768 - 100, 32768 + 1948  # a shard cut that is not wave aligned
    part = rollout(hi - lo, lo, tape[:, lo:hi])
    assert torch.equal(part[0], full[0][lo:hi]) and torch.equal(part[1], full[1][lo:hi]) and torch.equal(part[2], full[2][lo:hi])
    for k in ("loc", "vel", "flags", "prev_d", "tgt"):
        assert torch.equal(part[3][k], full[3][k][lo:hi]), k
    assert int(full[3]["counters"][:, 0].min()) == T and int(full[3]["counters"][:, 0].max()) == T
    flags = full[3]["flags"]
    assert torch.equal(full[3]["counters"][:, 1].long(), (flags & 1).sum(dim=1).long()), "reach count == done agents"
    assert torch.isfinite(full[0]).all() and float(full[0][..., [0, 2, 4, 7]].min()) >= 0.0
    assert float(full[0][..., [0, 1, 3, 4, 5, 6, 7, 8, 9]].abs().max()) <= 1.0 + 1e-6  # only the target distance may exceed 1


def test_full_size_oracle_differential(amd, oracle_mod):
    """BASELINE.json's headline size against the ORACLE, every env: 65 536 envs x 4 UAVs, Philox reset (a quarter of the envs
    then get targets 0.8 - 1.6 m from their UAVs, through set_state on both sides, so that arrivals happen within the run), 64
    steps of goal-seeking + noise commands -- arrivals, collisions and out-of-box terminations all occur: done masks at every
    step, positions, velocities, flags and counters bit for bit, observations / rewards to 1e-5.  The oracle runs on all host
    cores (~0.5 s for the 4.2 M env-steps; most of the test's time is numpy building the commands)."""
    import torch
    E, n, T = 65536, 4, 64
    nthreads = os.cpu_count() or 8
    env = amd.BatchedMultiUAVWorld2D(E, num_agents=n, seed=99)
    orc = oracle_mod.OracleMulti(num_envs=E, num_agents=n, nthreads=nthreads)
    env.reset(); orc.reset_philox(99)
    rng = np.random.default_rng(5)
    st0 = orc.get_state()
    near = np.arange(E) % 4 == 0
    ang, rad = rng.uniform(-np.pi, np.pi, size=(E, n)), rng.uniform(0.8, 1.6, size=(E, n))
    tgt = st0["tgt"].copy()
    tgt[near] = (st0["loc"] + np.stack([rad * np.cos(ang), rad * np.sin(ang)], axis=-1).astype(np.float32))[near]
    dd = tgt - st0["loc"]                                                  # float32, like MUW:154
    init = np.sqrt(dd[..., 0] * dd[..., 0] + dd[..., 1] * dd[..., 1]).astype(np.float32)
    env.set_state(tgt=tgt, init_d=init, prev_d=init); orc.set_state(tgt=tgt, init_d=init, prev_d=init)
    assert obs_err(_np(env.observe()), orc.observe()) <= TOL
    reached = collided = 0
    for t in range(T):
        d = orc.tgt - orc.loc
        dist = np.linalg.norm(d, axis=-1, keepdims=True)
        act = d / np.maximum(dist, 1e-9) * np.where(dist > 0.3, np.minimum(8.0, np.sqrt(4.0 * dist)), 0.0)
        act = (act + rng.normal(0, 1.0, size=d.shape) * (dist > 2.0)).astype(np.float32)
        o_g, r_g, d_g, _ = env.step(torch.from_numpy(act).to(env.device))
        o_o, r_o, d_o = orc.step(act)
        np.testing.assert_array_equal(_np(d_g).astype(np.uint8), d_o, err_msg=f"done mask, step {t}")
        assert obs_err(_np(o_g), o_o) <= TOL and float(np.abs(_np(r_g) - r_o).max()) <= TOL, f"step {t}"
        if t % 16 == 15 or t == T - 1:
            st, ref = env.get_state(), orc.get_state()
            for key in ("loc", "vel", "tgt", "init_d", "prev_d", "flags"):
                np.testing.assert_array_equal(_np(st[key]), ref[key], err_msg=f"{key}, step {t}")
            np.testing.assert_array_equal(_np(st["counters"]), ref["counters"].astype(np.int32), err_msg=f"counters, step {t}")
    reached, collided = int(orc.counters[:, 1].sum()), int(orc.counters[:, 2].sum())
    assert reached > 0 and collided > 0, (reached, collided)
    env.close()


def test_config3_eight_shards_equal_one_batch(amd):
    """BASELINE configs[3] (262 144 envs x 4 UAVs sharded over 8 GPUs, gather of episode metrics) on ONE GPU: the eight
    32 768-env shards a rank-r process would own (env_offset = r * 32 768) are run one after the other and must reproduce the
    single 262 144-env batch exactly -- state, outputs and the gathered [E, 4] metric rows the SR / CR come from.  (The
    8-process RCCL run itself is the driver's; the gather is covered by the world-2 gloo test and bench.py's 1-rank RCCL path.)"""
    import torch
    from gym_uav_collision_avoidance_amd.sharding import shard_range, summarize_metrics
    total, world, n, T = 262144, 8, 4, 24
    g = torch.Generator(device="cuda").manual_seed(5)
    tape = torch.rand((T, total, n, 2), generator=g, device="cuda") * 2 - 1

    def run(count, offset):
        env = amd.BatchedMultiUAVWorld2D(count, num_agents=n, seed=7, env_offset=offset)
        env.reset()
        for t in range(T):
            obs, rew, done, info = env.step_ex(tape[t, offset:offset + count], polar=True, auto_reset="agent0_done", step_cap=10)
        out = (obs.clone(), rew.clone(), done.clone(), env.metrics().clone(), env.get_state()["loc"].clone(),
               {k: v.clone() for k, v in env.episode_stats().items()})
        env.close()
        return out

    whole = run(total, 0)
    shards = [run(*reversed(shard_range(total, world, r))) for r in range(world)]
    for k in range(5):
        assert torch.equal(whole[k], torch.cat([s[k] for s in shards], dim=0)), k
    for key in whole[5]:
        assert torch.equal(whole[5][key], torch.cat([s[5][key] for s in shards], dim=0)), key
    gathered = torch.cat([s[3] for s in shards], dim=0)                      # what rank 0 holds after the one gather
    assert summarize_metrics(gathered, n) == summarize_metrics(whole[3], n)
    assert int(whole[5]["episodes"].sum()) >= 2 * total                      # the step cap ended every env at least twice


# ---------------------------------------------------------------------------------------------------
# UAVWorld2D
@pytest.mark.parametrize("name", UW)
def test_uw_fixture_replay(amd, name):
    data, meta = load_fixture(name)
    env = amd.BatchedUAVWorld2D(1, **{k: v for k, v in meta["cfg"].items() if k != "tau"})
    env.set_state(loc=data["init_loc"][None], vel=data["init_vel"][None], tgt=data["init_tgt"][None],
                  init_d=[data["init_init_d"]], prev_d=[data["init_prev_d"]],
                  flags=[4 if data["init_vel_f32"] else 0], counters=[[int(data["init_steps"]), 0]])
    assert obs_err(_np(env.observe())[0], data["init_obs"], UW_ANGLE_COLS) <= TOL
    for t in range(data["actions"].shape[0]):
        obs, rew, done, info = env.step(data["actions"][t][None])
        ctx = f"{name} step {t}"
        assert bool(_np(done)[0]) == bool(data["done"][t]), ctx
        st = env.get_state()
        np.testing.assert_array_equal(_np(st["loc"])[0], data["loc"][t].astype(np.float32), err_msg=ctx)
        np.testing.assert_array_equal(_np(st["vel"])[0], data["vel"][t], err_msg=ctx)
        assert obs_err(_np(obs)[0], data["obs"][t], UW_ANGLE_COLS) <= TOL, ctx
        # rewards reach ~1000 on success (UW:161): float32 resolution there is 6e-5, so the bar is
        # 1e-5 absolute or one float32 ulp, whichever is larger
        r_ref = data["rew"][t]
        assert abs(float(_np(rew)[0]) - r_ref) <= max(TOL, float(np.spacing(np.float32(abs(r_ref))))), ctx
        assert abs(float(_np(info["distance"])[0]) - data["info"][t]) == 0.0, ctx
    env.close()


def test_uw_oracle_random_batch(amd, oracle_mod):
    E = 4096     # BASELINE configs[1] literally (the single-UAV world)
    env = amd.BatchedUAVWorld2D(E, seed=21, env_offset=3)
    orc = oracle_mod.OracleSingle(num_envs=E, nthreads=8)
    obs_g = env.reset()
    orc.reset_philox(21, env_offset=3)
    st = env.get_state()
    ref = orc.get_state()
    for k in ("loc", "vel", "tgt", "init_d", "prev_d"):
        np.testing.assert_array_equal(_np(st[k]), ref[k], err_msg=k)
    assert obs_err(_np(obs_g), orc.observe(), UW_ANGLE_COLS) <= TOL
    rng = np.random.default_rng(2)
    for t in range(120):
        if t % 2 == 0:
            act = rng.uniform(-12, 12, size=(E, 2)).astype(np.float32)
        else:
            act = (orc.tgt - orc.loc) * rng.uniform(0.1, 2.0, size=(E, 1))
        obs_g, rew_g, done_g, info = env.step(act)
        obs_o, rew_o, done_o, info_o = orc.step(act)
        np.testing.assert_array_equal(_np(done_g).astype(np.uint8), done_o, err_msg=f"step {t}")
        st = env.get_state()
        np.testing.assert_array_equal(_np(st["loc"]), orc.loc.astype(np.float32))
        np.testing.assert_array_equal(_np(st["vel"]), orc.vel)
        assert obs_err(_np(obs_g), obs_o, UW_ANGLE_COLS) <= TOL
        tol_r = np.maximum(TOL, np.spacing(np.abs(rew_o).astype(np.float32)).astype(np.float64))
        assert (np.abs(_np(rew_g) - rew_o) <= tol_r).all(), f"step {t}"
        np.testing.assert_array_equal(_np(info["distance"]), info_o.astype(np.float32))
    env.close()


# ---------------------------------------------------------------------------------------------------
# drop-in façades (single env, gym-0.24 call surface)
def test_facade_multi_reset_follows_np_random(amd):
    """np.random.seed(s); env.reset() gives the reference's own start/target layout and observation."""
    from gym_uav_collision_avoidance_amd.envs import MultiUAVWorld2D
    data, meta = load_fixture("multi_resets")
    for k, spec in enumerate(meta["specs"]):
        np.random.seed(spec["np_seed"])
        env = MultiUAVWorld2D(**{kk: v for kk, v in spec["cfg"].items() if kk != "tau"})
        for r in range(spec["resets"]):
            obs = env.reset()
            st = env._batched.get_state()
            np.testing.assert_array_equal(_np(st["loc"])[0], data[f"r{k}_{r}_loc"].astype(np.float32))
            np.testing.assert_array_equal(_np(st["tgt"])[0], data[f"r{k}_{r}_tgt"].astype(np.float32))
            assert isinstance(obs, list) and len(obs) == spec["cfg"]["num_agents"] and obs[0].shape == (10,)
            assert obs_err(np.array(obs), data[f"r{k}_{r}_obs"]) <= TOL
        env.close()


def test_facade_run_multi_call_pattern(amd):
    """run_multi.py:5-23 driven against the façade (minus sleep/input), plus the attributes trainers read."""
    from gym_uav_collision_avoidance_amd.envs import MultiUAVWorld2D
    num_agent = 5
    env = MultiUAVWorld2D(num_agents=num_agent)
    np.random.seed(0)
    observation, info = env.reset(return_info=True)
    assert info == {"distance": 0} and observation[0].dtype == np.float64 and observation[0].shape == (10,)
    for _ in range(30):
        n_action = [env.action_space.sample() for _ in range(num_agent)]
        observation, reward, done, info = env.step(n_action)
        env.render()
        assert len(observation) == num_agent and len(reward) == num_agent and len(done) == num_agent
        assert all(isinstance(r, float) for r in reward) and all(isinstance(d, bool) for d in done)
        # MUW:98-109: float64 arrays of shape (10,) (here: the kernel's float32 values, widened exactly)
        assert all(o.dtype == np.float64 and o.shape == (10,) for o in observation)
        assert all((o.astype(np.float32).astype(np.float64) == o).all() for o in observation)
        if done[0]:
            observation, info = env.reset(return_info=True)
    assert env.observation_space.shape == (10,) and env.action_space.shape == (2,)
    assert env.steps >= 1 and env.target_reach_count >= 0 and env.collision_count >= 0
    assert float(np.linalg.norm(env.action_space.high)) == pytest.approx(14.1421356, rel=1e-6)
    a0 = env.agent_list[0]
    a0.location = np.array([1.5, -2.5])
    assert np.allclose(a0.location, [1.5, -2.5]) and a0.done in (True, False)
    env.close()


def test_examples_run_with_the_reference_import_lines(amd):
    """examples/run_multi.py and examples/run.py keep the reference's own import line (run_multi.py:2, run.py:2:
    `from gym_uav_collision_avoidance.envs import ...`); the opt-in alias makes it resolve to the MI355X façades."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for script, args in (("run_multi.py", ["--steps", "40"]), ("run.py", [])):
        src = open(os.path.join(root, "examples", script)).read()
        assert "from gym_uav_collision_avoidance.envs import" in src
        out = subprocess.run([sys.executable, os.path.join(root, "examples", script)] + args, capture_output=True, text=True,
                             timeout=300, cwd=root)
        assert out.returncode == 0, out.stderr[-2000:]
    assert "steps" in out.stdout or "episodes" in out.stdout


def test_facade_circular_reset_and_float64_pokes(amd, oracle_mod):
    """reset(circular=True) through the drop-in class reproduces the reference's float64 episode exactly
    (MUW:157-163), and assigning float64 arrays to agent_list[i].location / target_location the way
    test_sac_multi_plot_trajectory.py:43-49 does turns a random episode into a float64 one as well."""
    import math
    from gym_uav_collision_avoidance_amd.envs import MultiUAVWorld2D
    data, _ = load_fixture("crafted_circular_n4")
    n = 4
    env = MultiUAVWorld2D(num_agents=n)
    np.random.seed(5)
    obs = env.reset(circular=True)
    assert env._batched.position_mode == "float64" and obs_err(np.array(obs), data["init_obs"]) <= TOL \
        if "init_obs" in data else env._batched.position_mode == "float64"
    for i in range(n):
        a = env.agent_list[i]
        assert a.location.dtype == np.float64 and a.target_location.dtype == np.float64
        np.testing.assert_array_equal(a.location, data["init_loc"][i])
        np.testing.assert_array_equal(a.target_location, data["init_tgt"][i])
        assert a.init_distance == data["init_init_d"][i] and a.prev_distance == data["init_prev_d"][i]
    for t in range(250):
        o, r, d, _ = env.step([data["actions"][t][i] for i in range(n)])
        assert [bool(x) for x in d] == [bool(x) for x in data["done"][t]], t
        assert float(np.abs(np.array(r) - data["rew"][t]).max()) <= TOL
    for i in range(n):
        np.testing.assert_array_equal(env.agent_list[i].location, data["loc"][249][i])
    assert env.target_reach_count == int(data["counters"][249][1])
    # a plain reset() installs float32 arrays again; then the plotting script's pokes
    np.random.seed(9)
    env.reset()
    assert env._batched.position_mode == "float32" and env.agent_list[0].location.dtype == np.float32
    offset = 0.5
    for i in range(n):
        theta = 2 * i * math.pi / n
        env.agent_list[i].location = 20 * np.ones(2) * np.array([math.cos(theta), math.sin(theta)])
        env.agent_list[i].target_location = 23 * np.ones(2) * np.array([math.cos(theta + math.pi - offset * math.pi / n),
                                                                      math.sin(theta + math.pi - offset * math.pi / n)])
    assert env._batched.position_mode == "float64"
    orc = oracle_mod.OracleMulti(num_envs=1, num_agents=n, nthreads=1)
    s32, s64 = env._batched.get_state(), env._batched.get_state_f64()
    orc.set_state(loc=_np(s64["loc"]), tgt=_np(s64["tgt"]), init_d=_np(s64["init_d"]), prev_d=_np(s64["prev_d"]),
                  vel=_np(s32["vel"]), flags=_np(s32["flags"]))
    orc.f64pos[:] = 1
    assert orc.loc[0, 1, 0] == 20 * math.cos(2 * math.pi / n)             # the poked python floats arrived unrounded
    assert float(orc.init_d[0, 0]) == float(np.float32(orc.init_d[0, 0]))  # init / prev distance stay the stale float32 ones
    for t in range(300):
        dvec = orc.tgt[0] - orc.loc[0]
        act = [dvec[i] * 0.8 for i in range(n)]
        o, r, d, _ = env.step(act)
        oo, ro, do = orc.step(np.array(act)[None])
        assert [bool(x) for x in d] == [bool(x) for x in do[0]], t
        assert obs_err(np.array(o), oo[0]) <= TOL and float(np.abs(np.array(r) - ro[0]).max()) <= TOL * max(1.0, float(np.abs(ro).max()))
    np.testing.assert_array_equal(_np(env._batched.get_state_f64()["loc"]), orc.loc)
    env.close()


def test_facade_uw_matches_reference_stream(amd):
    from gym_uav_collision_avoidance_amd.envs import UAVWorld2D
    data, meta = load_fixture("uw_resets")
    np.random.seed(meta["np_seed"])
    env = UAVWorld2D()
    for r in range(meta["resets"]):
        obs, info = env.reset(return_info=True)
        assert obs.dtype == np.float64 and obs.shape == (4,)      # UW:106-111
        assert obs_err(obs, data[f"r{r}_obs"], UW_ANGLE_COLS) <= TOL
        assert abs(float(info["distance"]) - float(data[f"r{r}_init_d"])) == 0.0
    # UW rollout from a seeded reset == fixture (reset state there came from the same stream position)
    d2, m2 = load_fixture("uw_box32_s30")
    np.random.seed(m2["np_seed"])
    env.reset()
    for t in range(50):
        obs, rew, done, info = env.step(d2["actions"][t])
        assert obs.dtype == np.float64 and obs.shape == (4,)
        assert obs_err(obs, d2["obs"][t], UW_ANGLE_COLS) <= TOL and bool(done) == bool(d2["done"][t])
    env.close()


def test_facade_mapped_host_buffers_equal_the_copy_path(amd, monkeypatch):
    """The single-env façades hand the launch their PINNED host blocks (commands in, obs | reward | done out: mapped host memory,
    no copy calls); UAVX_FACADE_COPIES=1 keeps the H2D / D2H copies.  Same values either way, float64-position episodes too."""
    from gym_uav_collision_avoidance_amd.envs import MultiUAVWorld2D, UAVWorld2D
    envs = []
    for copies in ("0", "1"):
        monkeypatch.setenv("UAVX_FACADE_COPIES", copies)
        envs.append((MultiUAVWorld2D(num_agents=5), UAVWorld2D()))
    (ma, ua), (mb, ub) = envs
    assert ma._mapped and ua._mapped and not mb._mapped and not ub._mapped
    rng = np.random.default_rng(5)
    for circular in (False, True):
        outs = []
        for m in (ma, mb):
            np.random.seed(11)
            outs.append(m.reset(circular=circular))
        assert all((x == y).all() for x, y in zip(*outs))
        for t in range(40):
            act = [rng.uniform(-8, 8, size=2) for _ in range(5)]
            ra, rb = ma.step(act), mb.step(act)
            assert all((x == y).all() for x, y in zip(ra[0], rb[0])) and ra[1] == rb[1] and ra[2] == rb[2], (circular, t)
    for u in (ua, ub):
        np.random.seed(3)
        u.reset()
    for t in range(60):
        act = rng.uniform(-10, 10, size=2).astype(np.float32 if t % 2 else np.float64)
        ra, rb = ua.step(act), ub.step(act)
        assert (ra[0] == rb[0]).all() and ra[1] == rb[1] and ra[2] == rb[2] and ra[3] == rb[3], t
    for e in (ma, mb, ua, ub):
        e.close()


def test_hand_written_sqrt_is_ieee_on_every_float(amd):
    """uavx_selftest: the kernels' 9-instruction square root equals the compiler's correctly rounded sqrtf on all
    float32 bit patterns (0 ... +inf and a block of NaNs) on this device."""
    import ctypes
    from gym_uav_collision_avoidance_amd import _lib
    bad = ctypes.c_uint64(123)
    assert _lib.load().uavx_selftest(0, ctypes.byref(bad)) == 0
    assert bad.value == 0


def test_errors_are_loud(amd):
    import torch
    env = amd.BatchedMultiUAVWorld2D(8, num_agents=4)
    with pytest.raises(ValueError):
        env.step(torch.zeros((8, 3, 2), device=env.device))
    with pytest.raises(ValueError):
        env.step(torch.zeros((8, 4, 2), dtype=torch.float16, device=env.device))
    with pytest.raises(ValueError):
        env.step(torch.zeros((8, 4, 2)))  # host tensor is not silently copied
    with pytest.raises(ValueError):
        amd.BatchedMultiUAVWorld2D(8, num_agents=65)
    env.close()


@pytest.mark.parametrize("n", [6, 8, 13, 24])
def test_neighbour_order_on_near_ties(amd, oracle_mod, n):
    """The N>5 scan orders neighbours by an integer key (squared-distance bits truncated to 2^-17 relative, index
    in the low bits) and must fall back to the exact (float32 distance, index) order whenever that truncation
    could matter.  Rings of neighbours whose radii differ by 0 ... a few thousand float32 ulps, in shuffled index
    order, around the sensing limit too: nearest-two identity (their heading columns differ) and every mask must
    equal the oracle's, for observe() and for a step."""
    E = 1500
    kw = dict(num_agents=n, d_sense=9.0, x_size=60.0, y_size=60.0)
    env = amd.BatchedMultiUAVWorld2D(E, seed=3, **kw)
    orc = oracle_mod.OracleMulti(num_envs=E, nthreads=8, **kw)
    env.reset(); orc.reset_philox(3)
    rng = np.random.default_rng(n)
    loc = np.zeros((E, n, 2), np.float32)
    centre = rng.uniform(-8, 8, size=(E, 1, 2)).astype(np.float32)
    r0 = np.where(rng.random((E, 1)) < 0.3, 9.0, rng.uniform(1.2, 8.5, size=(E, 1)))   # 30 %: ring at d_sense
    eps = rng.choice([0.0, 0.0, 2.0 ** -23, -2.0 ** -23, 2.0 ** -21, 2.0 ** -18, -2.0 ** -17, 2.0 ** -16, 2.0 ** -12],
                     size=(E, n - 1))
    ang = rng.uniform(-np.pi, np.pi, size=(E, n - 1))
    ring = (r0 * (1.0 + eps))[..., None] * np.stack([np.cos(ang), np.sin(ang)], -1)
    far = rng.random((E, n - 1)) < 0.25                                                  # some well outside range
    ring = np.where(far[..., None], ring * 2.5, ring)
    who = np.argsort(rng.random((E, n)), axis=1)                                         # which agent is the centre
    others = np.stack([np.delete(np.arange(n), who[e, 0]) for e in range(E)])
    loc[np.arange(E), who[:, 0]] = centre[:, 0]
    loc[np.arange(E)[:, None], others] = (centre + ring).astype(np.float32)
    vel = rng.uniform(-3, 3, size=(E, n, 2))
    env.set_state(loc=loc, vel=vel); orc.set_state(loc=loc, vel=vel)
    obs_g, obs_o = _np(env.observe()), orc.observe()
    assert obs_err(obs_g, obs_o) <= TOL
    act = rng.uniform(-10, 10, size=(E, n, 2)).astype(np.float32)
    og, rg, dg, _ = env.step(act)
    oo, ro, do = orc.step(act)
    np.testing.assert_array_equal(_np(dg).astype(np.uint8), do)
    ref = orc.get_state()
    _check_multi_state(env, dict(flags=ref["flags"], loc=ref["loc"], prev_d=ref["prev_d"], vel=ref["vel"],
                                 counters=ref["counters"][:, :3]), f"near ties n={n}")
    # the poke leaves prev_distance stale, so |reward| reaches hundreds (MUW:190): relative bar beyond |r| = 1
    assert obs_err(_np(og), oo) <= TOL and (np.abs(_np(rg) - ro) <= TOL * np.maximum(1.0, np.abs(ro))).all()
    env.close()


def test_poked_state_keeps_reference_prev_distance_semantics(amd, oracle_mod):
    """prev_distance is not stored on the device (it is derived from position/target); values that break
    that identity — a caller moving an agent without touching prev_distance, as
    test_sac_multi_plot_trajectory.py:43-49 does, or setting done flags — must still act exactly like the
    reference's explicit field for the next step."""
    E, n = 700, 4
    env = amd.BatchedMultiUAVWorld2D(E, num_agents=n, seed=8)
    orc = oracle_mod.OracleMulti(num_envs=E, num_agents=n, nthreads=8)
    env.reset()
    orc.reset_philox(8)
    rng = np.random.default_rng(4)

    def step_both(ctx):
        act = rng.uniform(-10, 10, size=(E, n, 2)).astype(np.float32)
        obs_g, rew_g, done_g, _ = env.step(act)
        obs_o, rew_o, done_o = orc.step(act)
        np.testing.assert_array_equal(_np(done_g).astype(np.uint8), done_o, err_msg=ctx)
        ref = orc.get_state()
        _check_multi_state(env, dict(flags=ref["flags"], loc=ref["loc"], prev_d=ref["prev_d"], vel=ref["vel"],
                                     counters=ref["counters"][:, :3]), ctx)
        # a teleported agent earns |reward| up to ~50 (MUW:190): beyond |r| = 1 the bar is relative (float32 output)
        assert (np.abs(_np(rew_g) - rew_o) <= TOL * np.maximum(1.0, np.abs(rew_o))).all(), ctx
        assert obs_err(_np(obs_g), obs_o) <= TOL, ctx

    for t in range(5):
        step_both(f"warm {t}")
    # (1) move agents, prev_distance left stale
    new_loc = orc.loc.copy()
    new_loc[::2, 1] = rng.uniform(-20, 20, size=new_loc[::2, 1].shape).astype(np.float32)
    env.set_state(loc=new_loc)
    orc.set_state(loc=new_loc)
    np.testing.assert_array_equal(_np(env.get_state()["prev_d"]), orc.prev_d.astype(np.float32))
    step_both("after loc poke")
    step_both("after loc poke +1")
    # (2) explicit prev_distance and done flags (a done agent never moves again, AG:24-25)
    flags = orc.flags.copy(); flags[1::3, 2] |= 1
    pd = orc.prev_d.copy(); pd[:, 0] = 3.25
    env.set_state(flags=flags, prev_d=pd)
    orc.set_state(flags=flags, prev_d=pd)
    st = env.get_state()
    np.testing.assert_array_equal(_np(st["prev_d"]), pd.astype(np.float32))
    np.testing.assert_array_equal(_np(st["flags"]), flags)
    for t in range(4):
        step_both(f"after flag/prev_d poke {t}")
    # (3) get_state -> set_state round trip is the identity
    st = env.get_state()
    env.set_state(**{k: v for k, v in st.items()})
    st2 = env.get_state()
    import torch
    for k in st:
        assert torch.equal(st[k], st2[k]), k
    step_both("after round trip")
    env.close()


def test_nonfinite_actions_propagate_like_numpy(amd, oracle_mod):
    """np.clip lets NaN through and clips +-inf (AG:26-27): a NaN command poisons that agent's velocity and
    position for good, an infinite one just saturates the acceleration.  The device path must do the same."""
    E, n = 64, 4
    env = amd.BatchedMultiUAVWorld2D(E, num_agents=n, seed=2)
    orc = oracle_mod.OracleMulti(num_envs=E, num_agents=n)
    env.reset(); orc.reset_philox(2)
    rng = np.random.default_rng(1)
    for t in range(6):
        act = rng.uniform(-10, 10, size=(E, n, 2))
        if t == 1:
            act[0, 1, 0] = np.inf; act[1, 2, 1] = -np.inf; act[2, 0, :] = [np.inf, -np.inf]
        if t == 2:
            act[3, 3, 0] = np.nan; act[4, 0, 1] = np.nan
        obs_g, rew_g, done_g, _ = env.step(act)
        obs_o, rew_o, done_o = orc.step(act)
        st = env.get_state()
        np.testing.assert_array_equal(_np(st["vel"]), orc.vel, err_msg=f"step {t}")          # NaN == NaN here
        np.testing.assert_array_equal(_np(st["loc"]), orc.loc.astype(np.float32), err_msg=f"step {t}")
        np.testing.assert_array_equal(_np(done_g).astype(np.uint8), done_o, err_msg=f"step {t}")
        np.testing.assert_array_equal(_np(st["flags"]), orc.flags, err_msg=f"step {t}")
        finite = np.isfinite(obs_o).all(axis=-1) & np.isfinite(rew_o)
        assert obs_err(_np(obs_g)[finite], obs_o[finite]) <= TOL
        assert float(np.abs(_np(rew_g)[finite] - rew_o[finite]).max()) <= TOL
        assert np.isnan(_np(obs_g)[~finite]).any(axis=-1).all() if (~finite).any() else True
        bad_steps = bad_steps + (~np.isfinite(rew_o)).sum(axis=1) if t else (~np.isfinite(rew_o)).sum(axis=1)
        np.testing.assert_array_equal(_np(env.nonfinite_count()), bad_steps, err_msg=f"tripwire, step {t}")
    assert np.isnan(orc.vel[3, 3]).any() and np.isnan(orc.vel[4, 0]).any() and np.isfinite(orc.vel[0, 1]).all()
    assert bad_steps[3] == 4 and bad_steps[4] == 4 and bad_steps[:3].sum() == 0 and bad_steps[5:].sum() == 0   # envs 3, 4 from step 2 on
    env.reset()
    assert int(env.nonfinite_count().sum()) == 0                          # cleared with the MUW:166-168 counters
    env.close()


def test_side_stream_and_hip_array_interface(amd):
    """Launches go to the caller's current HIP stream (no hidden sync), and any object that publishes
    __hip_array_interface__ (same schema as the CUDA array interface) is accepted zero-copy as the action buffer."""
    import torch
    E, n = 4096, 4
    a = amd.BatchedMultiUAVWorld2D(E, num_agents=n, seed=12)
    b = amd.BatchedMultiUAVWorld2D(E, num_agents=n, seed=12)
    a.reset(); b.reset()
    g = torch.Generator(device="cpu").manual_seed(3)
    tape = (torch.rand((12, E, n, 2), generator=g) * 20 - 10).to(a.device)
    side = torch.cuda.Stream(a.device)
    side.wait_stream(torch.cuda.current_stream(a.device))
    with torch.cuda.stream(side):
        for t in range(12):
            oa, ra, da, _ = a.step(tape[t])
    side.synchronize()

    class Foreign:  # a non-torch device buffer that only speaks __hip_array_interface__
        def __init__(self, t):
            self._keep = t
            self.__hip_array_interface__ = amd.HipArray(t).__hip_array_interface__

    for t in range(12):
        ob, rb, db, _ = b.step(Foreign(tape[t]))
    torch.cuda.synchronize()
    assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db)
    iface = amd.HipArray(oa).__hip_array_interface__
    assert iface["data"][0] == oa.data_ptr() and iface["shape"] == (E, n, 10) and iface["typestr"] == "<f4"
    a.close(); b.close()


@pytest.mark.parametrize("case", range(16))
def test_randomized_constructor_parameters(amd, oracle_mod, case):
    """Random world sizes / speed and acceleration limits / collider radii / sensing ranges / agent counts,
    including degenerate ones (collider_radius 0, d_sense below 2R, long thin boxes): device vs oracle, masks
    and state bit-exact, observations / rewards within tolerance."""
    rng = np.random.default_rng(9000 + case)
    n = int(rng.choice([1, 2, 3, 4, 6, 8, 9, 13, 17, 32]))
    kw = dict(x_size=float(rng.uniform(8, 120)), y_size=float(rng.uniform(8, 120)),
              max_speed=float(rng.uniform(1, 25)), max_acceleration=float(rng.uniform(0.5, 12)),
              collider_radius=float(rng.choice([0.0, 0.3, 0.5, 1.0, 1.7])),
              d_sense=float(rng.choice([0.5, 3.0, 7.25, 15.0, 40.0])), num_agents=n)
    if case == 3:
        kw.update(x_size=200.0, y_size=6.0)
    while n * 3.2 * (2 * kw["collider_radius"]) ** 2 > 0.5 * kw["x_size"] * kw["y_size"]:
        kw["x_size"] *= 1.5; kw["y_size"] *= 1.5          # keep the rejection sampler's acceptance rate sane
    E = 300
    env = amd.BatchedMultiUAVWorld2D(E, seed=case, **kw)
    orc = oracle_mod.OracleMulti(num_envs=E, nthreads=8, **kw)
    obs_g = env.reset()
    orc.reset_philox(case)
    assert obs_err(_np(obs_g), orc.observe()) <= TOL, kw
    vmax = kw["max_speed"]
    for t in range(40):
        if t % 2:
            act = rng.uniform(-vmax, vmax, size=(E, n, 2)).astype(np.float32)
        else:
            d = orc.tgt - orc.loc
            act = d * rng.uniform(0.1, 2.0, size=(E, n, 1))
        ev = bool(t % 4 == 3)
        obs_g, rew_g, done_g, _ = env.step(act, evaluate=ev)
        obs_o, rew_o, done_o = orc.step(act, evaluate=ev)
        ctx = f"case {case} {kw} step {t}"
        np.testing.assert_array_equal(_np(done_g).astype(np.uint8), done_o, err_msg=ctx)
        ref = orc.get_state()
        _check_multi_state(env, dict(flags=ref["flags"], loc=ref["loc"], prev_d=ref["prev_d"], vel=ref["vel"],
                                     counters=ref["counters"][:, :3]), ctx)
        assert obs_err(_np(obs_g), obs_o) <= TOL, ctx
        assert (np.abs(_np(rew_g) - rew_o) <= TOL * np.maximum(1.0, np.abs(rew_o))).all(), ctx
    env.close()
