"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol include/uavx.h
declares, host logic (sharding, spaces), and the world_size-2 gloo gather."""
import os
import re
import socket

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from gym_uav_collision_avoidance_amd import _lib
    _lib.build()
    hdr = open(os.path.join(ROOT, "include", "uavx.h")).read()
    declared = set(re.findall(r"\b(uavx_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"uavx_handle", "uavx_uw_handle"}
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert lib.uavx_version() == 3
    assert lib.uavx_strerror(-1) == b"invalid argument"


def test_no_gpu_means_loud_failure_not_fallback():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import gym_uav_collision_avoidance_amd as pkg
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pkg.BatchedMultiUAVWorld2D(4)
    from gym_uav_collision_avoidance_amd.envs import MultiUAVWorld2D, UAVWorld2D
    with pytest.raises(RuntimeError):
        MultiUAVWorld2D()
    with pytest.raises(RuntimeError):
        UAVWorld2D()


def test_product_never_imports_the_oracle():
    pkg_dir = os.path.join(ROOT, "gym_uav_collision_avoidance_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "uavx_oracle" not in txt, f


def test_box_space():
    from gym_uav_collision_avoidance_amd.spaces import Box
    b = Box(-10.0, 10.0, shape=(2,), dtype=np.float32)
    np.random.seed(3)
    s = b.sample()
    assert s.dtype == np.float32 and s.shape == (2,) and b.contains(s)
    assert float(np.linalg.norm(b.high)) == pytest.approx(14.142135, rel=1e-6)  # test_sac_multi.py:77


def test_shard_range_partitions_exactly():
    from gym_uav_collision_avoidance_amd.sharding import shard_range
    for total, world in ((262144, 8), (65536, 1), (10, 3), (7, 8)):
        spans = [shard_range(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == total
        for (o1, c1), (o2, _) in zip(spans, spans[1:]):
            assert o1 + c1 == o2


def test_philox_streams_are_keyed_by_global_env(oracle_mod):
    """Shard independence of reset: resetting global envs [0,12) in one piece equals
    resetting [0,5) and [5,12) separately with env_offset."""
    kw = dict(num_agents=4)
    whole = oracle_mod.OracleMulti(num_envs=12, **kw)
    whole.reset_philox(42)
    a = oracle_mod.OracleMulti(num_envs=5, **kw)
    b = oracle_mod.OracleMulti(num_envs=7, **kw)
    a.reset_philox(42, env_offset=0)
    b.reset_philox(42, env_offset=5)
    np.testing.assert_array_equal(whole.loc, np.concatenate([a.loc, b.loc]))
    np.testing.assert_array_equal(whole.tgt, np.concatenate([a.tgt, b.tgt]))
    # Philox4x32-10 known-answer vectors (Random123 kat_vectors)
    np.testing.assert_array_equal(oracle_mod.philox4x32([0, 0, 0, 0], [0, 0]),
                                  np.array([0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8], dtype=np.uint32))
    np.testing.assert_array_equal(oracle_mod.philox4x32([0xffffffff] * 4, [0xffffffff] * 2),
                                  np.array([0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd], dtype=np.uint32))
    np.testing.assert_array_equal(oracle_mod.philox4x32([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]),
                                  np.array([0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1], dtype=np.uint32))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _gather_worker(rank, world, port, total, q):
    import torch.distributed as dist
    from gym_uav_collision_avoidance_amd.sharding import (gather_episode_metrics, reduce_episode_totals, shard_range,
                                                        summarize_metrics)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    off, cnt = shard_range(total, world, rank)
    rows = torch.arange(off, off + cnt, dtype=torch.int32)
    local = torch.stack([rows * 3, rows % 5, rows % 2, torch.ones_like(rows)], dim=1)  # fake counters keyed by global env
    # exactly ONE collective per call: count what torch.distributed is asked to do
    calls = []
    for name in ("gather", "all_gather", "reduce", "all_reduce", "broadcast"):
        orig = getattr(dist, name)
        setattr(dist, name, (lambda o, n: (lambda *a, **k: (calls.append(n), o(*a, **k))[1]))(orig, name))
    out = gather_episode_metrics(local, dst=0, total_envs=total)
    assert calls == ["gather"], calls
    calls.clear()
    red = reduce_episode_totals(local, 4, dst=0)
    assert calls == ["reduce"], calls
    if rank == 0:
        q.put((out.numpy(), summarize_metrics(out, 4), red))
    else:
        assert out is None and red is None
    dist.barrier()
    dist.destroy_process_group()


def _summary_worker(rank, world, port, total, q):
    """gather_evaluation_summary over an ODD total split by shard_range, the size taken from the env object the way
    make_sharded_env records it (the env here is a stand-in with the three things the function reads)."""
    import torch.distributed as dist
    from gym_uav_collision_avoidance_amd.sharding import gather_evaluation_summary, shard_range
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    off, cnt = shard_range(total, world, rank)
    rows = torch.arange(off, off + cnt)

    class Shard:
        num_agents = 4
        total_envs = total                      # what make_sharded_env sets

        def episode_stats(self):
            return dict(episodes=torch.ones_like(rows), steps=rows * 2, reach=rows % 3, coll=rows % 2,
                        return0=rows.float(), score=rows.float() * 0.5)
    calls = []
    orig = dist.gather
    dist.gather = lambda *a, **k: (calls.append("gather"), orig(*a, **k))[1]
    out = gather_evaluation_summary(Shard(), dst=0)
    assert calls == ["gather"], calls
    if rank == 0:
        q.put(out)
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_evaluation_summary_over_an_odd_total():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port, total, world = _free_port(), 13, 2   # 7 + 6 rows: ranks hold different row counts
    procs = [ctx.Process(target=_summary_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    rows = np.arange(total)
    assert got["episodes"] == total and got["success_rate"] == pytest.approx((rows % 3).sum() / (4 * total))
    assert got["collision_rate"] == pytest.approx((rows % 2).sum() / (4 * total)) and got["mean_steps"] == pytest.approx((rows * 2).sum() / total)
    assert got["avg_score"] == pytest.approx((rows * 0.5).sum() / (4 * total))


@pytest.mark.parametrize("world,total", [(2, 11), (8, 37)])   # uneven shards: 6 + 5; 5 x 5 + 3 x 4 (the node's eight ranks)
def test_gloo_metrics_gather(world, total):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, summary, reduced = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    rows = np.arange(total)
    want = np.stack([rows * 3, rows % 5, rows % 2, np.ones_like(rows)], axis=1)
    np.testing.assert_array_equal(got, want)
    assert summary["success_rate"] == pytest.approx((rows % 5).sum() / (4 * total))
    # the 4-scalar reduce path gives the same SR / CR / mean length without moving the rows
    assert reduced["episodes"] == total and reduced["success_rate"] == pytest.approx(summary["success_rate"])
    assert reduced["collision_rate"] == pytest.approx(summary["collision_rate"])
    assert reduced["mean_steps"] == pytest.approx(summary["mean_steps"])


def test_abi_argument_validation_without_gpu():
    """Status codes of the C ABI for bad arguments (checked before any device is touched) and for a
    box without a GPU; nothing computes here."""
    import ctypes
    from gym_uav_collision_avoidance_amd import _lib
    L = _lib.load()
    h = ctypes.c_void_p()
    good = _lib.Config(50.0, 50.0, 10.0, 5.0, 1.0, 15.0, 0.02, 4, 0)
    assert L.uavx_create(None, 8, 0, 0, ctypes.byref(h)) == -1                       # NULL config
    assert L.uavx_create(ctypes.byref(good), 0, 0, 0, ctypes.byref(h)) == -1         # no envs
    assert L.uavx_create(ctypes.byref(good), 8, -1, 0, ctypes.byref(h)) == -1        # negative env_offset
    for field, val in (("num_agents", 0), ("num_agents", 65), ("tau", 0.0), ("max_speed", -1.0), ("x_size", 0.0),
                       ("d_sense", 0.0), ("collider_radius", -0.5)):
        bad = _lib.Config(50.0, 50.0, 10.0, 5.0, 1.0, 15.0, 0.02, 4, 0)
        setattr(bad, field, val)
        assert L.uavx_create(ctypes.byref(bad), 8, 0, 0, ctypes.byref(h)) == -1, field
    assert L.uavx_create(ctypes.byref(good), 1 << 24, 0, 0, ctypes.byref(h)) == -4   # E*N >= 2^26: unsupported
    if not torch.cuda.is_available():
        assert L.uavx_create(ctypes.byref(good), 8, 0, 0, ctypes.byref(h)) == -3     # UAVX_ERR_NO_DEVICE
    assert L.uavx_step(None, None, 0, 0, None, None, None, None) == -1
    assert L.uavx_destroy(None) == -1
    assert L.uavx_selftest(0, None) == -1
    assert L.uavx_set_config(None, ctypes.byref(good)) == -1
    assert L.uavx_set_position_mode(None, 1, None) == -1 and L.uavx_get_position_mode(None) == -1
    assert L.uavx_set_state_f64(None, None, None) == -1 and L.uavx_get_state_f64(None, None, None) == -1
    assert b"null handle" in L.uavx_last_error(None)
    uw = _lib.UWConfig(100.0, 100.0, 12.0, 5.0, 0.0)
    assert L.uavx_uw_create(ctypes.byref(uw), 8, 0, 0, ctypes.byref(h)) == -1       # tau == 0


def test_policy_checkpoint_layout_on_cpu(tmp_path):
    """The batched actor mirrors the reference's GaussianPolicy parameter names (model.py:64-78) and loads the
    'policy_state_dict' entry of SAC.save_checkpoint (sac.py:108) with weights_only=True."""
    from gym_uav_collision_avoidance_amd.policy import GaussianPolicy, load_reference_checkpoint
    torch.manual_seed(1)
    src = GaussianPolicy()
    assert set(src.state_dict()) == {"linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias",
                                     "mean_linear.weight", "mean_linear.bias", "log_std_linear.weight",
                                     "log_std_linear.bias"}
    path = tmp_path / "weights.chpt"
    torch.save({"policy_state_dict": src.state_dict(), "critic_state_dict": {}, "critic_target_state_dict": {},
                "critic_optimizer_state_dict": {}, "policy_optimizer_state_dict": {}}, path)
    pol = load_reference_checkpoint(str(path), device="cpu")
    obs = torch.rand((7, 3, 10))
    a = pol.act(obs)
    assert a.shape == (7, 3, 2) and float(a.abs().max()) <= 1.0
    assert torch.allclose(a, torch.tanh(src(obs)[0]))
    g = torch.Generator().manual_seed(0)
    assert not torch.equal(pol.act(obs, evaluate=False, generator=g), a)


def test_seek_policy_output_convention():
    """The observation-driven controller used by the evaluation tests emits policy outputs in [-1,1]^2 whose
    polar conversion (test_sac_multi.py:77-80) points at the target."""
    import math
    from gym_uav_collision_avoidance_amd.evaluate import seek_policy
    pol = seek_policy()
    obs = torch.zeros((1, 1, 10))
    obs[0, 0, 1] = 0.25          # heading theta_v = pi/4
    obs[0, 0, 2] = 30.0 / math.hypot(50.0, 50.0)   # 30 m from the target
    obs[0, 0, 3] = 0.5           # target is 90 degrees to the left of the heading
    a = pol(obs)[0, 0]
    v = (float(a[0]) / 2 + 0.5) * math.sqrt(200.0)
    th = float(a[1]) * math.pi
    assert abs(th - 0.75 * math.pi) < 1e-6 and abs(v - 8.0) < 1e-5   # capped cruise speed, bearing 135 degrees


def test_lane_to_env_multiply_shift_is_exact():
    """lane_map() of the runtime-N step kernel divides the lane index by N with (lane * (65536//N + 1)) >> 16."""
    for n in range(1, 65):
        magic = 65536 // n + 1
        assert all(((lane * magic) >> 16) == lane // n for lane in range(64)), n
        # staging workgroups map a lane per SLOT with the same trick over up to four wavefronts (stage_ahead, magic_s)
        assert all(((lane * magic) >> 16) == lane // n for lane in range(256)), n


def test_checkpoint_written_in_the_reference_layout(tmp_path):
    """policy.save_reference_checkpoint writes the five keys of SAC.save_checkpoint (pytorch_sac_temp/sac.py:108-112) with
    state dicts that reference-shaped modules and Adam optimisers load; the policy part round-trips through the loader."""
    from gym_uav_collision_avoidance_amd.policy import (CHECKPOINT_KEYS, GaussianPolicy, TwinQ, load_reference_checkpoint,
                                                        save_reference_checkpoint)
    torch.manual_seed(0)
    pol = GaussianPolicy()
    popt = torch.optim.Adam(pol.parameters(), lr=3e-4)
    mean, log_std = pol(torch.randn(5, 10))
    (mean.sum() + log_std.sum()).backward()
    popt.step()                                            # an optimiser WITH state (exp_avg, exp_avg_sq, step)
    f = save_reference_checkpoint(str(tmp_path / "ckpt" / "weights.chpt"), pol, policy_optimizer=popt)
    ck = torch.load(f, weights_only=True)                  # tensors and plain containers only
    assert tuple(ck) == CHECKPOINT_KEYS
    assert set(ck["critic_state_dict"]) == {f"linear{i}.{w}" for i in range(1, 7) for w in ("weight", "bias")}
    assert ck["critic_state_dict"]["linear1.weight"].shape == (256, 12) and ck["critic_state_dict"]["linear6.weight"].shape == (1, 256)
    for k in ck["critic_state_dict"]:
        assert torch.equal(ck["critic_state_dict"][k], ck["critic_target_state_dict"][k])     # hard_update, sac.py:26
    # what SAC.load_checkpoint does (sac.py:126-130) with modules / optimisers of the reference's shapes
    pol2, q, qt = GaussianPolicy(), TwinQ(), TwinQ()
    pol2.load_state_dict(ck["policy_state_dict"]); q.load_state_dict(ck["critic_state_dict"]); qt.load_state_dict(ck["critic_target_state_dict"])
    torch.optim.Adam(q.parameters(), lr=3e-4).load_state_dict(ck["critic_optimizer_state_dict"])
    o2 = torch.optim.Adam(pol2.parameters(), lr=3e-4)
    o2.load_state_dict(ck["policy_optimizer_state_dict"])
    assert len(o2.state_dict()["state"]) == 8              # 4 layers x (weight, bias) carry their Adam moments
    back = load_reference_checkpoint(f, device="cpu")
    for a, b in zip(pol.state_dict().values(), back.state_dict().values()):
        assert torch.equal(a, b)


def test_td3_and_ddpg_checkpoint_layouts_on_cpu(tmp_path):
    """The batched actors mirror the parameter names of the reference's TD3 Actor (td3.py:14-27) and DDPG ActorNetwork
    (model.py:6-31, incl. its unused BatchNorm1d buffers), load what the reference writes (weights_only=True), and write files
    with exactly the reference's keys whose entries load into reference-shaped modules and optimisers."""
    from gym_uav_collision_avoidance_amd import policy as P
    torch.manual_seed(3)
    # --- TD3: a checkpoint in the reference's layout (td3.py:163-169) -> loader -> batched act
    src = P.TD3Actor()
    assert set(src.state_dict()) == {f"l{i}.{w}" for i in (1, 2, 3) for w in ("weight", "bias")}
    assert src.l1.weight.shape == (256, 10) and src.l3.weight.shape == (2, 256)
    f = tmp_path / "td3" / "weights.chpt"
    f.parent.mkdir()
    torch.save({"actor_state_dict": src.state_dict(), "actor_target_state_dict": src.state_dict(), "critic_state_dict": {},
                "critic_target_state_dict": {}, "actor_optimizer_state_dict": {}, "critic_optimizer_state_dict": {}}, f)
    pol = P.load_td3_checkpoint(str(f), device="cpu")
    obs = torch.rand((6, 3, 10))
    a = pol.act(obs)
    assert a.shape == (6, 3, 2) and float(a.abs().max()) <= 1.0 and torch.equal(a, src(obs))
    g = torch.Generator().manual_seed(0)
    noisy = pol.act(obs, evaluate=False, generator=g)
    assert not torch.equal(noisy, a) and float(noisy.abs().max()) <= 1.0
    out = P.save_td3_checkpoint(str(tmp_path / "td3_out" / "weights.chpt"), pol)
    ck = torch.load(out, weights_only=True)
    assert tuple(ck) == P.TD3_CHECKPOINT_KEYS
    assert set(ck["critic_state_dict"]) == {f"l{i}.{w}" for i in range(1, 7) for w in ("weight", "bias")} and ck["critic_state_dict"]["l4.weight"].shape == (256, 12)
    a2, q2 = P.TD3Actor(), P.TD3TwinQ()                                   # what TD3.load_checkpoint does (td3.py:176-182)
    a2.load_state_dict(ck["actor_state_dict"]); a2.load_state_dict(ck["actor_target_state_dict"])
    q2.load_state_dict(ck["critic_state_dict"]); q2.load_state_dict(ck["critic_target_state_dict"])
    torch.optim.Adam(a2.parameters(), lr=3e-4).load_state_dict(ck["actor_optimizer_state_dict"])
    torch.optim.Adam(q2.parameters(), lr=3e-4).load_state_dict(ck["critic_optimizer_state_dict"])
    assert torch.equal(a2(obs), a)
    # --- DDPG: actor.chpt / critic.chpt (ddpg.py:124-135)
    dsrc = P.DDPGActor()
    assert set(dsrc.state_dict()) == {"input.weight", "input.bias", "fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias",
                                      "bn1.running_mean", "bn1.running_var", "bn1.num_batches_tracked"}
    assert dsrc.input.weight.shape == (400, 10) and dsrc.fc1.weight.shape == (300, 400) and dsrc.fc2.weight.shape == (2, 300)
    d = tmp_path / "ddpg"
    d.mkdir()
    torch.save({"model_state_dict": dsrc.state_dict(), "target_model_state_dict": dsrc.state_dict(), "optimizer_state_dict": {}}, d / "actor.chpt")
    dpol = P.load_ddpg_checkpoint(str(d), device="cpu")
    da = dpol.act(obs)
    assert da.shape == (6, 3, 2) and torch.equal(da, dsrc.eval()(obs))
    want = torch.tanh(dsrc.fc2(torch.nn.functional.leaky_relu(dsrc.fc1(torch.nn.functional.leaky_relu(dsrc.input(obs))))))   # model.py:23-31, bn1 unused
    assert torch.allclose(da, want)
    assert torch.equal(dpol.act(obs, evaluate=False, noise=torch.full((2,), 5.0)), torch.ones_like(da))     # OU noise added, clipped
    outd = P.save_ddpg_checkpoint(str(tmp_path / "ddpg_out"), dpol)
    for name, mod in (("actor.chpt", P.DDPGActor()), ("critic.chpt", P.DDPGCritic())):
        ck = torch.load(os.path.join(outd, name), weights_only=True)
        assert tuple(ck) == P.DDPG_CHECKPOINT_KEYS
        mod.load_state_dict(ck["model_state_dict"]); mod.load_state_dict(ck["target_model_state_dict"])
        torch.optim.Adam(mod.parameters(), lr=1e-3, amsgrad=True).load_state_dict(ck["optimizer_state_dict"])
    # --- the three layouts told apart by their keys
    sac = P.GaussianPolicy()
    torch.save({"policy_state_dict": sac.state_dict()}, tmp_path / "sac.chpt")
    assert isinstance(P.load_actor(str(tmp_path / "sac.chpt"), device="cpu"), P.GaussianPolicy)
    assert isinstance(P.load_actor(str(f), device="cpu"), P.TD3Actor) and isinstance(P.load_actor(str(d), device="cpu"), P.DDPGActor)
    torch.save({"something": 1}, tmp_path / "x.chpt")
    with pytest.raises(ValueError):
        P.load_actor(str(tmp_path / "x.chpt"), device="cpu")


def test_bench_algorithmic_bytes_and_refusal_without_gpu():
    """The per-env-step byte figures bench.py prices the roofline with and its refusal to run without a GPU."""
    import subprocess
    import sys
    sys.path.insert(0, ROOT)
    import bench
    assert bench.algorithmic_bytes_per_env_step(4) == 452 and bench.algorithmic_bytes_per_env_step(1) == 131
    assert bench.algorithmic_bytes_per_env_step(8) == 880 and bench.algorithmic_bytes_per_env_step(8, 16) == 880 + 16 * 32
    # a curriculum that switches on 6 of 8 learners and 10 of 16 bodies on average: parked learners only have their 45 B of outputs written
    assert bench.algorithmic_bytes_per_env_step(8, 16, 6, 10) == 107 * 6 + 45 * 2 + 24 + 32 * 10
    import torch
    if not torch.cuda.is_available():
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "0"], capture_output=True, text=True, timeout=300)
        assert out.returncode != 0 and "no CPU fallback" in (out.stderr + out.stdout)


def test_bench_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (what a driver may run for the scaling curve) starts
    torch.distributed.run itself as a child process and leaves with the child's exit code.  Without a GPU every rank
    refuses to run: the refusal must be the RANKS' ("no CPU fallback"), printed by processes that carry WORLD_SIZE=2,
    not a complaint of the parent about a missing launcher."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("the GPU form of this test is tests/test_gpu_bench_contract.py::test_gpus_2_without_a_launcher")
    env = dict(os.environ, UAVX_REHEARSAL="1")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0"],
                         capture_output=True, text=True, timeout=600, env=env)
    text = out.stderr + out.stdout
    assert out.returncode != 0
    assert "no CPU fallback" in text and "launcher must start" not in text, text[-2000:]


def test_source_hash_ignores_comments_and_follows_code(tmp_path, monkeypatch):
    """The identity a profile / a built library carries (`_lib.source_hash`) is taken over the CODE of the kernel sources: a
    reworded comment or re-indented line must not orphan a profile, a changed token must.  Literals are code (`//` inside a
    string is not a comment)."""
    from gym_uav_collision_avoidance_amd import _lib
    a = 'int f() { return 1; }   // one\n/* block\n   comment */ const char *s = "// not a comment"; char q = \'"\';\n'
    b = 'int f() {\n    return 1;\n}\nconst char *s = "// not a comment";   /* other words */   char q = \'"\';   // two\n'
    assert _lib._code_only(a) == _lib._code_only(b)
    assert _lib._code_only(a) != _lib._code_only(a.replace("return 1", "return 2"))
    assert _lib._code_only(a) != _lib._code_only(a.replace("// not a comment", "// not  a comment"))   # inside the literal: code
    # the hash of the tree: stable under a comment appended to a kernel source, moved by a token
    csrc = tmp_path / "csrc"
    csrc.mkdir()
    inc = tmp_path / "include"
    inc.mkdir()
    (csrc / "k.hip").write_text("__global__ void k(int *p) { *p = 1; }\n")
    (csrc / "d.hpp").write_text("// helpers\nstatic int two() { return 2; }\n")
    (inc / "uavx.h").write_text("/* ABI */\nint uavx_version(void);\n")
    pkg = tmp_path / "pkg"
    pkg.mkdir()
    monkeypatch.setattr(_lib, "CSRC", str(csrc))
    monkeypatch.setattr(_lib, "_HERE", str(pkg))
    (tmp_path / "pkg").rmdir()
    monkeypatch.setattr(_lib, "_HERE", str(tmp_path / "x"))      # include/ is looked up next to the package directory's parent
    h0 = _lib.source_hash()
    (csrc / "k.hip").write_text("// a new comment\n__global__ void k(int *p) {\n    *p = 1;   // same code\n}\n")
    assert _lib.source_hash() == h0
    (csrc / "k.hip").write_text("__global__ void k(int *p) { *p = 2; }\n")
    assert _lib.source_hash() != h0
    # where a preprocessor directive ends is code
    assert _lib._code_only("#define A 1\nint x = A;\n") != _lib._code_only("#define A 1 int x = A;\n")
    assert _lib._code_only("#define A 1   // one\nint x = A;\n") == _lib._code_only("#define   A 1\n\nint x =\n A;\n")
    # the build recipe belongs to the identity: a changed flag (-ffp-contract decides bit-exactness) or -D knob moves the hash,
    # a reworded Makefile comment does not; so does an override from the environment
    (csrc / "k.hip").write_text("__global__ void k(int *p) { *p = 1; }\n")
    (csrc / "Makefile").write_text("# builds\nHIPFLAGS ?= -O3 -ffp-contract=off   # no FMA\n")
    h1 = _lib.source_hash()
    assert h1 != h0
    (csrc / "Makefile").write_text("# builds the library\nHIPFLAGS ?= -O3 -ffp-contract=off   # exact chains\n")
    assert _lib.source_hash() == h1
    (csrc / "Makefile").write_text("# builds\nHIPFLAGS ?= -O3 -ffp-contract=fast\n")
    assert _lib.source_hash() != h1
    (csrc / "Makefile").write_text("# builds\nHIPFLAGS ?= -O3 -ffp-contract=off   # no FMA\n")
    monkeypatch.setenv("HIPFLAGS", "-O3 -DUAVX_EXB=6")
    assert _lib.source_hash() != h1


def test_alias_package_is_opt_in(tmp_path):
    """`from gym_uav_collision_avoidance.envs import MultiUAVWorld2D` (run_multi.py:2) resolves to the MI355X façade only after
    install_alias() / UAVX_ALIAS=1; by default the name is not importable from this repo, and an importable reference checkout
    is never shadowed without force=True."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import importlib.util as u\n"
            "import gym_uav_collision_avoidance_amd as g\n"
            "print('default', u.find_spec('gym_uav_collision_avoidance') is not None)\n" % ROOT)
    env = {k: v for k, v in os.environ.items() if k not in ("UAVX_ALIAS", "PYTHONPATH")}
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env, cwd=str(tmp_path))
    assert out.returncode == 0 and "default False" in out.stdout, out.stderr[-1500:]
    code2 = code + ("g.install_alias()\n"
                    "import gym_uav_collision_avoidance, gym_uav_collision_avoidance.envs as e\n"
                    "from gym_uav_collision_avoidance_amd.envs import MultiUAVWorld2D, UAVWorld2D\n"
                    "print('same', e.MultiUAVWorld2D is MultiUAVWorld2D and e.UAVWorld2D is UAVWorld2D)\n")
    out = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True, timeout=300, env=env, cwd=str(tmp_path))
    assert out.returncode == 0 and "same True" in out.stdout, out.stderr[-1500:]
    out = subprocess.run([sys.executable, "-c", code.replace("print('default'", "import gym_uav_collision_avoidance.envs as e; print('viaenv'")],
                         capture_output=True, text=True, timeout=300, env=dict(env, UAVX_ALIAS="1"), cwd=str(tmp_path))
    assert out.returncode == 0 and "viaenv True" in out.stdout, out.stderr[-1500:]
    # a "real" checkout on the path: refused unless forced
    fake = tmp_path / "ref" / "gym_uav_collision_avoidance"
    fake.mkdir(parents=True)
    (fake / "__init__.py").write_text("REAL = True\n")
    code3 = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
             "import gym_uav_collision_avoidance_amd as g\n"
             "try:\n    g.install_alias()\n    print('installed')\nexcept ImportError as e:\n    print('refused')\n"
             "g.install_alias(force=True)\nimport gym_uav_collision_avoidance as m\nprint('forced', not hasattr(m, 'REAL'))\n"
             % (ROOT, str(tmp_path / "ref")))
    out = subprocess.run([sys.executable, "-c", code3], capture_output=True, text=True, timeout=300, env=env, cwd=str(tmp_path))
    assert out.returncode == 0 and "refused" in out.stdout and "forced True" in out.stdout, out.stdout + out.stderr[-1500:]


def test_hot_kernels_keep_their_scalars_in_registers():
    """Code-object metadata of the built library (llvm-readelf --notes through tools/kernel_resources.py; nothing runs).
    * no step / observe kernel of the float32 path spills a VGPR or touches scratch memory;
    * the kernels a one-round launch of 8 192 wavefronts depends on fit 8 wavefronts per SIMD: <= 64 VGPRs AND <= 80 SGPRs
      (800 per SIMD in blocks of 16, plus the trap handler's 16 per wavefront: tools/micro/occupancy.hip measured exactly that);
    * the headline kernels park no scalar in VGPR lanes; the two kernels of BASELINE configs[4] may hold a few there: the builds
      without any (late kernel-argument loads at every site) were measured SLOWER -- bare step with bodies 17.4 -> 18.2 us --
      so the bound is what the faster build has (profiles/r04_ab_notes.md section 1), and a regression past it fails here."""
    import importlib.util
    from gym_uav_collision_avoidance_amd import _lib
    lib = _lib.build()
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "tools", "kernel_resources.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    rows = {r["name"]: r for r in kr.kernel_table(lib)}
    assert len(rows) > 50
    for name, r in rows.items():
        if name.startswith(("step_kernel", "step_ex_kernel", "observe_kernel", "uw_step")):
            assert r["vgpr_spill_count"] == 0 and r["private_segment_fixed_size"] == 0, (name, r)
    for name in ("step_kernel<1, false, false, 1, 1>", "step_kernel<2, false, false, 1, 1>", "step_kernel<4, false, false, 1, 1>",
                 "step_kernel<8, false, false, 1, 1>", "step_ex_kernel<4, false, false, 1, 1>", "uw_step_kernel<false>"):
        assert rows[name]["sgpr_spill_count"] == 0, (name, rows[name])
    for name in ("step_kernel<0, false, true, 1, 1>", "step_ex_kernel<8, false, false, 1, 1>"):
        assert rows[name]["sgpr_spill_count"] <= 8, (name, rows[name])
    # (two tiles per workgroup, what 65 536 x 8 runs: the tile offset rides in the row base, no register of its own)
    assert rows["step_kernel<8, false, false, 1, 2>"]["sgpr_spill_count"] == 0
    assert rows["step_ex_kernel<8, false, false, 1, 2>"]["sgpr_spill_count"] <= 10
    # (the fused kernel with bodies: a dozen since its leading arguments are preloaded -- and 2 % faster with them, notes section 10)
    assert rows["step_ex_kernel<0, false, true, 1, 1>"]["sgpr_spill_count"] <= 14, rows["step_ex_kernel<0, false, true, 1, 1>"]
    for name in ("step_kernel<0, false, true, 1, 1>", "step_ex_kernel<0, false, true, 1, 1>", "step_kernel<8, false, false, 1, 1>",
                 "step_ex_kernel<8, false, false, 1, 1>", "step_kernel<4, false, false, 1, 1>",
                 "step_kernel<8, false, false, 1, 2>", "step_ex_kernel<8, false, false, 1, 2>"):
        r = rows[name]
        assert r["waves_per_simd"] == 8 and r["vgpr_count"] <= 64 and r["sgpr_count"] <= 80, (name, r)
