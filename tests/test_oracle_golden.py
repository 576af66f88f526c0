"""Pins the CPU oracle (oracle/uavx_oracle.c) against fixtures generated from the reference itself
(tests/golden/make_golden.py).  The oracle uses the same libm and the same float32/float64 op
order as numpy evaluates for the reference, so it is expected to be BIT-EXACT on every recorded
quantity, not just within the 1e-5 product tolerance."""
import numpy as np
import pytest

from golden_util import fixture_names, load_fixture, tie_agents

MULTI = fixture_names("multi")
UW = fixture_names("uw")


def _mk_multi(oracle_mod, meta, init):
    cfg = meta["cfg"]
    o = oracle_mod.OracleMulti(num_envs=1, **cfg)
    o.set_state(loc=init["init_loc"], vel=init["init_vel"], tgt=init["init_tgt"], init_d=init["init_init_d"],
                prev_d=init["init_prev_d"], flags=init["init_flags"])
    o.counters[0, :3] = init["init_counters"]
    o.f64pos[0] = init["init_f64pos"]
    return o


@pytest.mark.parametrize("name", MULTI)
def test_multi_rollout_bit_exact(oracle_mod, name):
    data, meta = load_fixture(name)
    o = _mk_multi(oracle_mod, meta, data)
    T = data["actions"].shape[0]
    for t in range(T):
        obs, rew, done = o.step(data["actions"][t], evaluate=bool(data["evaluate"][t]))
        ctx = f"{name} step {t}"
        np.testing.assert_array_equal(done[0], data["done"][t], err_msg=ctx)
        np.testing.assert_array_equal(o.flags[0], data["flags"][t], err_msg=ctx)
        np.testing.assert_array_equal(o.counters[0, :3], data["counters"][t], err_msg=ctx)
        np.testing.assert_array_equal(o.loc[0], data["loc"][t], err_msg=ctx)
        np.testing.assert_array_equal(o.vel[0], data["vel"][t], err_msg=ctx)
        np.testing.assert_array_equal(o.prev_d[0], data["prev_d"][t], err_msg=ctx)
        np.testing.assert_array_equal(rew[0], data["rew"][t], err_msg=ctx)
        got, ref = obs[0].copy(), data["obs"][t].copy()
        if not np.array_equal(got, ref):  # only legal cause: platform-dependent order of exact ties
            ties = tie_agents(o.loc[0], meta["cfg"]["d_sense"], bool(o.f64pos[0]))
            assert ties.any(), ctx
            got[ties, 4:] = 0
            ref[ties, 4:] = 0
        np.testing.assert_array_equal(got, ref, err_msg=ctx)


def test_multi_fixtures_cover_the_edge_cases():
    """The fixture set must actually exercise success, stickiness, collisions, OOB and evaluate."""
    seen = dict(success=0, collision=0, hard=0, oob_done=0, sticky=0, evaluate=0, noneigh=0)
    for name in MULTI:
        d, _ = load_fixture(name)
        seen["success"] += int(d["counters"][-1][1])
        seen["hard"] += int(d["counters"][-1][2])
        seen["collision"] += int((d["rew"] == -2.0).sum())
        seen["sticky"] += int(((d["flags"][:-1] & 1) & (d["done"][1:] == 1)).sum())
        seen["evaluate"] += int(d["evaluate"].sum())
        seen["oob_done"] += int(((d["done"] == 1) & ((d["flags"] & 1) == 0)).sum())
        seen["noneigh"] += int((d["obs"][..., 4] == 1.0).sum())
    assert all(v > 0 for v in seen.values()), seen


def test_multi_reset_stream_matches_np_random(oracle_mod):
    data, meta = load_fixture("multi_resets")
    for k, spec in enumerate(meta["specs"]):
        o = oracle_mod.OracleMulti(num_envs=1, **spec["cfg"])
        g = oracle_mod.MTStream(spec["np_seed"])
        for r in range(spec["resets"]):
            o.reset_mt(g)
            for key in ("loc", "tgt", "init_d", "prev_d"):
                np.testing.assert_array_equal(getattr(o, key)[0], data[f"r{k}_{r}_{key}"], err_msg=f"{k}/{r}/{key}")
            assert not o.vel.any() and not o.flags.any() and not o.counters[0, :3].any()
            np.testing.assert_array_equal(o.observe()[0], data[f"r{k}_{r}_obs"])


def test_multi_circular_reset(oracle_mod):
    for n in (4, 6):
        data, meta = load_fixture(f"crafted_circular_n{n}")
        o = oracle_mod.OracleMulti(num_envs=1, **meta["cfg"])
        o.reset_mt(oracle_mod.MTStream(meta["np_seed"]), circular=True)
        np.testing.assert_array_equal(o.loc[0], data["init_loc"])
        np.testing.assert_array_equal(o.tgt[0], data["init_tgt"])
        np.testing.assert_array_equal(o.init_d[0], data["init_init_d"])
        assert o.f64pos[0] == 1


@pytest.mark.parametrize("name", UW)
def test_uw_rollout_bit_exact(oracle_mod, name):
    data, meta = load_fixture(name)
    o = oracle_mod.OracleSingle(num_envs=1, **meta["cfg"])
    o.set_state(loc=data["init_loc"], vel=data["init_vel"], tgt=data["init_tgt"], init_d=data["init_init_d"],
                prev_d=data["init_prev_d"], steps=data["init_steps"], vel_f32=data["init_vel_f32"])
    np.testing.assert_array_equal(o.observe()[0], data["init_obs"])
    for t in range(data["actions"].shape[0]):
        obs, rew, done, info = o.step(data["actions"][t])
        ctx = f"{name} step {t}"
        assert done[0] == data["done"][t], ctx
        np.testing.assert_array_equal(o.loc[0], data["loc"][t], err_msg=ctx)
        np.testing.assert_array_equal(o.vel[0], data["vel"][t], err_msg=ctx)
        assert rew[0] == data["rew"][t], ctx
        assert info[0] == data["info"][t], ctx
        np.testing.assert_array_equal(obs[0], data["obs"][t], err_msg=ctx)
        assert o.steps[0] == data["steps"][t]


def test_uw_reset_stream_matches_np_random(oracle_mod):
    data, meta = load_fixture("uw_resets")
    o = oracle_mod.OracleSingle(num_envs=1)
    g = oracle_mod.MTStream(meta["np_seed"])
    for r in range(meta["resets"]):
        o.reset_mt(g)
        for k in ("loc", "vel", "tgt"):
            np.testing.assert_array_equal(getattr(o, k)[0], data[f"r{r}_{k}"])
        assert o.init_d[0] == data[f"r{r}_init_d"]
        np.testing.assert_array_equal(o.observe()[0], data[f"r{r}_obs"])
