"""bench.py contract checks on the GPU box: the one-line JSON at N=1, and a 2-rank rehearsal of the N>1
launch path (both ranks share the one GPU and talk over gloo: RCCL refuses two ranks per device; the real
multi-GPU run is the driver's)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"}


def _last_json(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


def test_single_gpu_line():
    out = subprocess.run([sys.executable, "bench.py", "--steps", "300", "--warmup", "30", "--no-cpu-baseline"],
                         cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    assert REQUIRED <= set(d) and d["n_gpus"] == 1 and d["steps"] == 300 and d["warmup"] == 30
    assert d["metric"] == "env-steps/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["scaling"] == "weak" and d["data"] == "synthetic" and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["achieved"] > 400.0   # > 5 % of peak or something broke
    assert d["value"] > 5e7, "target of BASELINE.json: >= 50 M env-steps/s"
    assert abs(d["value"] - 65536 * 300 / (d["ms_per_step"] * 300 / 1e3)) / d["value"] < 1e-6
    assert d["repeats"] == 5 and len(d["repeat_ms_per_step"]) == 5
    assert d["ms_per_step"] == pytest.approx(sorted(d["repeat_ms_per_step"])[2])          # the median region
    assert d["config"]["mode"] == "graph" and d["config"]["graph_replays"] == 6            # 300 = 6 x 50-step graphs
    assert "L3-resident" in r["note"] and d["gather_ms"] > 0 and d["latency_us"]["single_step_launch_to_done"] > 0
    ro = d["reset_observe"]               # the path's other two launches on the same batch (informational)
    assert 0 < ro["observe"]["us_per_launch"] < ro["reset"]["us_per_launch"] < 1e3
    sp = d["split_batch"]                 # the same batch as two independent half-batch chains (informational, not `value`)
    assert sp["chains"] == 2 and sp["envs"] == 65536 and sp["value"] > 0.8 * d["value"] and 0.3 < sp["frac"] < 1.0
    big = d["roofline_large"]
    assert big["envs"] == 1 << 20 and big["working_set_bytes"] > 2 * 256 * 2 ** 20 and 0.3 < big["frac"] < 1.0


def test_driver_sized_run_is_a_pure_graph_replay():
    """The driver's round-end command (--steps 20 --warmup 5): 20 < ring length, so the region must be ONE replay of a
    20-node graph (round 1 silently ran it as 20 eager launches and reported half the rate the kernel earns)."""
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu-baseline"],
                         cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    assert d["steps"] == 20 and d["warmup"] == 5 and d["config"]["mode"] == "graph" and d["config"]["graph_replays"] == 1
    # the line says what THIS run stepped: the note carries the real ages of the world, not a constant
    assert d["world_age"]["wall_regions"] == [8, 108] and d["world_age"]["event_regions"] == [108, 108 + 5 * 11 * 20]
    assert "8-108 launches after its reset" in d["world_note"] and "thousands of steps old" not in d["world_note"]
    assert d["roofline"]["traffic_measured_in_this_run"] is False and "traffic_note" in d["roofline"]
    assert d["hsa_env"]["HSA_ENABLE_IPC_MODE_LEGACY"] is not None and "distributed" not in d and "omitted_for_n_gpus>1" not in d
    assert d["steps_executed"] == 3 + 5 + (5 + 5 * 11) * 20    # capture warm-up + warm-up + 5 wall-clock regions of exactly 20 launches + 5 device-time regions (an untimed pass + ten timed ones each: >= 200 launches per event bracket; counted before roofline_steady runs)
    assert d["value"] > 7e9, d["value"]   # 20 x ~6.3 us of kernels + one graph launch; 9-10 G on a quiet box
    ol = d["open_loop_step_k"]            # informational: K steps per launch from an action tape (uavx_step_k), never `value`
    assert ol["K"] == 32 and ol["tape_out"] and ol["value"] > d["value"]
    # counter-derived figures appear exactly when profiles/ holds a PMC pass of THIS build of the kernels (same source hash)
    assert ("valu_frac" in ol) == ("roofline_valu" in d) == (d["roofline"]["traffic"] is not None)
    st = d["roofline_steady"]             # the same launches over 1000-step regions, next to the 20-step figure
    assert st["steps"] == 1000 and st["frac"] >= 0.93 * d["roofline"]["frac"] and st["kernel_us"] > 3.0   # (box noise; both are pure device time now)
    assert d["roofline"]["kernel_us"] < 1e3 * d["ms_per_step"]   # the wall-clock region holds one graph-launch latency, the event region none


def test_cfg5_as_designed_line():
    """BASELINE configs[4] in one launch sequence: 8 learners + 16 scripted bodies, fused step_ex with auto-reset, a
    randomized-reset curriculum and the outputs landing in the on-device replay ring (small batch: the line, not the rate)."""
    out = subprocess.run([sys.executable, "bench.py", "--cfg5", "--envs", "2048", "--ring", "8", "--steps", "64", "--warmup", "8",
                          "--repeats", "2", "--no-cpu-baseline", "--no-large"], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    c = d["config"]
    assert c["agents"] == 8 and c["bodies"] == 16 and c["curriculum_levels"] == 4 and c["replay"] is True
    assert c["mode"] == "graph" and c["graph_replays"] == 8 and "replay ring" in c["workload"]
    assert d["roofline"]["kernel"].startswith("uavx::step_ex_kernel<0") and d["value"] > 1e6
    assert d["episode_metrics"]["ended_episodes"]["episodes"] >= 0


def test_two_rank_rehearsal():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, UAVX_REHEARSAL="1")
    env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)     # an outer launcher that does not export it: the ranks set the default themselves
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2",
                          "--steps", "200", "--warmup", "20", "--envs", "8192"],
                         cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _last_json(out.stdout)
    assert d["n_gpus"] == 2 and d["config"]["envs_per_gpu"] == 8192 and d["config"]["parallelism"] == "env-index shard x2"
    assert abs(d["value"] - 2 * 8192 * 200 / (d["ms_per_step"] * 200 / 1e3)) / d["value"] < 1e-6   # whole-job aggregate
    assert "cpu_baseline" not in d and "cpu_baseline" in d["omitted_for_n_gpus>1"]
    _check_two_ranks(d)
    assert d["steps_executed"] == 3 + 20 + 15 * 200       # capture warm-up + warm-up + 5 wall-clock and 5 device-time regions (an untimed + a timed pass each)
    assert d["episode_metrics"]["mean_steps"] == float(d["steps_executed"])   # ... on every env of both shards


def test_gpus_2_without_a_launcher():
    """`python3 bench.py --gpus 2 ...` exactly as a driver would type it for the scaling curve, NOT wrapped in
    torch.distributed.run: bench.py starts its two ranks itself (child process, before any GPU call in the parent),
    rank 0's line comes through on stdout, exit code 0.  (Rehearsal: both ranks share the one GPU and talk over gloo.)"""
    env = dict(os.environ, UAVX_REHEARSAL="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "20", "--warmup", "5", "--envs", "8192"],
                         cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]              # ONE line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["warmup"] == 5 and d["config"]["envs_per_gpu"] == 8192
    assert d["config"]["parallelism"] == "env-index shard x2" and d["scaling"] == "weak"
    assert abs(d["value"] - 2 * 8192 * 20 / (d["ms_per_step"] * 20 / 1e3)) / d["value"] < 1e-6
    _check_two_ranks(d)
    # no --envs given: the hint about configs[3] goes to stderr, the line's tag is computed from what ran
    out2 = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "20", "--warmup", "5", "--envs", "131072", "--no-large"],
                          cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert out2.returncode == 0, out2.stderr[-3000:]
    d2 = _last_json(out2.stdout)
    assert "configs[3]" in d2["config"]["workload"] and d2["config"]["envs_per_gpu"] == 131072   # 2 x 131 072 = the 262 144 envs of configs[3]
    _check_two_ranks(d2)


def test_gpus_4_without_a_launcher():
    """Four self-launched ranks (rehearsal: they share the one GPU, gloo): the line the scaling curve's N = 4 point would print --
    262 144 envs in all is BASELINE configs[3]'s total, tagged from what ran."""
    env = dict(os.environ, UAVX_REHEARSAL="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "4", "--steps", "20", "--warmup", "5"],
                         cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["config"]["envs_per_gpu"] == 65536 and d["config"]["parallelism"] == "env-index shard x4"
    assert abs(d["value"] - 4 * 65536 * 20 / (d["ms_per_step"] * 20 / 1e3)) / d["value"] < 1e-6
    assert "configs[3]" in d["config"]["workload"] and "65 536 envs PER GPU" in out.stderr
    _check_two_ranks(d, 4)


def _check_two_ranks(d, n=2):
    """The N > 1 line proves what it ran on: n ranks seen by the process group, a time per rank, the same HSA environment in
    both launch forms (an outer torchrun and bench.py's own launcher)."""
    g = d["distributed"]
    assert g["ranks_seen"] == n and g["backend"] == "gloo" and g["rehearsal_shared_gpu"] is True
    assert len(g["per_rank"]) == n and [r["rank"] for r in g["per_rank"]] == list(range(n))
    assert len(g["per_rank_ms_per_step"]) == n and all(t > 0 for t in g["per_rank_ms_per_step"])
    assert d["ms_per_step"] >= 0.999 * min(g["per_rank_ms_per_step"])     # the line's time is the slowest rank's, region by region
    assert {r["hsa_enable_ipc_mode_legacy"] for r in g["per_rank"]} == {d["hsa_env"]["HSA_ENABLE_IPC_MODE_LEGACY"]} == {"0"}
    assert len({r["pid"] for r in g["per_rank"]}) == n


def test_rccl_code_path_with_one_rank():
    """The N>1 branch of bench.py on the real RCCL backend (communicator on the device, gather, barriers, hipGraph
    capture next to the RCCL watchdog thread) with a single rank: what a one-GPU box can exercise of it."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, UAVX_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "1",
                          "--steps", "200", "--warmup", "20", "--envs", "16384", "--no-cpu-baseline"],
                         cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _last_json(out.stdout)
    assert d["n_gpus"] == 1 and d["config"]["mode"] == "graph", (d["config"], out.stderr[-1500:])
    assert d["episode_metrics"]["mean_steps"] == float(d["steps_executed"]) == 3023.0
    g = d["distributed"]
    assert g["ranks_seen"] == 1 and g["backend"] == "nccl" and g["rccl_version"] and g["rehearsal_shared_gpu"] is False
    assert g["distinct_devices"] == 1 and g["per_rank"][0]["name"] and len(g["per_rank_ms_per_step"]) == 1
