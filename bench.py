#!/usr/bin/env python3
"""bench.py — env-steps/s of the batched MultiUAVWorld2D step path on MI355X.

    python bench.py --gpus 1 --steps 2000 --warmup 200
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one launch of the fused HIP step kernel over the whole batch (all agents of all envs
advanced once, MUW:177-241).  Workload = BASELINE.json configs[2], the configuration the metric is
quoted on: 65 536 envs x 4 UAVs per GPU (weak scaling: every rank owns 65 536 envs, sharded by env
index with no step-path communication; one RCCL gather of episode metrics at the end of the timed
region).  State, action ring and outputs are resident in HBM before the timed region starts.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def algorithmic_bytes_per_env_step(n_agents):
    """lean f32 SoA, 41 B read + 66 B write per agent-step + 24 B per-env counters."""
    return 107 * n_agents + 24


def measured_traffic(E, N):
    """HBM bytes per step launch from the committed rocprofv3 PMC passes (profiles/*_pmc_summary.json,
    produced by tools/summarize_profiles.py from separate FETCH_SIZE / WRITE_SIZE runs of this
    command, with the gfx950 FETCH_SIZE x2 correction).  Only valid for the profiled grid."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("_meta", {}).get("grid") == E * N and "hbm_traffic_bytes_per_launch" in d:
            best = (d["hbm_traffic_bytes_per_launch"]["total"], os.path.basename(f))
    return best


def polar_actions(gen, shape, vmax_norm, device):
    """a ~ U(-1,1)^2 mapped like the trainers do (test_sac_multi.py:77-80)."""
    a = torch.rand(shape + (2,), generator=gen, device=device) * 2 - 1
    v = (a[..., 0] / 2 + 0.5) * vmax_norm
    th = a[..., 1] * np.pi
    return torch.stack([v * torch.cos(th), v * torch.sin(th)], dim=-1).contiguous()


def cpu_baseline(n_agents, budget_s=12.0):
    """The CPU oracle (a C port of the reference's step, oracle/uavx_oracle.c) timed on this box's
    host cores on a bounded sample of the same workload: 4 096 envs x n_agents, same reset seed and
    action distribution."""
    import oracle
    oracle.build()
    cores = min(os.cpu_count() or 1, 16)
    E = 4096
    rng = np.random.default_rng(1234)
    a = rng.uniform(-1, 1, size=(8, E, n_agents, 2))
    v = (a[..., 0] / 2 + 0.5) * np.sqrt(200.0)
    acts = np.stack([v * np.cos(a[..., 1] * np.pi), v * np.sin(a[..., 1] * np.pi)], axis=-1)
    out = {}
    for label, threads in (("1", 1), ("all", cores)):
        orc = oracle.OracleMulti(num_envs=E, num_agents=n_agents, nthreads=threads)
        orc.reset_philox(0)
        for k in range(3):
            orc.step(acts[k % 8])
        t0 = time.perf_counter()
        steps = 0
        while True:
            for k in range(8):
                orc.step(acts[k])
            steps += 8
            if time.perf_counter() - t0 > budget_s / 2:
                break
        out[label] = E * steps / (time.perf_counter() - t0)
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return dict(value=out["all"], unit="env-steps/s", cores=cores, kind="port", cpu_model=model, host_cpus=os.cpu_count(),
                sample=f"{E} envs x {n_agents} UAVs, oracle/uavx_oracle.c with OpenMP over envs on {cores} threads, ~{budget_s / 2:.0f} s",
                single_thread_value=out["1"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs", type=int, default=65536, help="envs per GPU")
    ap.add_argument("--agents", type=int, default=4)
    ap.add_argument("--world", choices=("multi", "uw"), default="multi",
                    help="multi: MultiUAVWorld2D (headline); uw: UAVWorld2D (single UAV, BASELINE configs[1] family)")
    ap.add_argument("--fused", action="store_true",
                    help="multi only: drive uavx_step_ex (polar action conversion, agent0-done auto-reset with a "
                         "1500-step cap, episode statistics) instead of the bare step")
    ap.add_argument("--mode", choices=("graph", "launch"), default="graph",
                    help="graph: steps replayed from a captured hipGraph (one kernel node per step); "
                         "launch: one ctypes->hipLaunchKernel per step")
    ap.add_argument("--ring", type=int, default=50, help="distinct action batches resident in HBM")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X; there is no CPU fallback for the env step path")
    # UAVX_REHEARSAL=1: several ranks share GPU 0 and talk over gloo (to rehearse the N>1 launch contract on
    # a one-GPU box; RCCL refuses two ranks on one device).  Never set by the driver.
    rehearsal = os.environ.get("UAVX_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # UAVX_FORCE_DIST=1 (under torch.distributed.run --nproc-per-node 1): take the N>1 code path -- RCCL communicator,
    # gather, barriers -- with a single rank, to exercise it on a one-GPU box.  Never set by the driver.
    distributed = world > 1 or os.environ.get("UAVX_FORCE_DIST") == "1"
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    from gym_uav_collision_avoidance_amd import BatchedMultiUAVWorld2D, BatchedUAVWorld2D
    from gym_uav_collision_avoidance_amd.sharding import gather_episode_metrics, summarize_metrics

    E, N, K, W = args.envs, args.agents, args.steps, args.warmup
    gen = torch.Generator(device=device).manual_seed(1234 + rank)
    if args.world == "uw":
        N = 1
        env = BatchedUAVWorld2D(E, device=device, env_offset=rank * E, seed=0)
        ring = polar_actions(gen, (args.ring, E), float(np.sqrt(288.0)), device)  # ||(12,12)||, test_sac.py:77
        step = env.step
        bytes_per_env_step, kernel_name = 93, "uavx::uw_step_kernel<false>"
    else:
        env = BatchedMultiUAVWorld2D(E, num_agents=N, device=device, env_offset=rank * E, seed=0)
        bytes_per_env_step = algorithmic_bytes_per_env_step(N)
        if args.fused:
            ring = torch.rand((args.ring, E, N, 2), generator=gen, device=device) * 2 - 1
            step = lambda a: env.step_ex(a, polar=True, auto_reset="agent0_done", step_cap=1500, track_returns=True)
            kernel_name = f"uavx::step_ex_kernel<{N if N in (1, 2, 4, 8) else 0},false>"
        else:
            ring = polar_actions(gen, (args.ring, E, N), float(np.sqrt(200.0)), device)
            step = env.step
            kernel_name = f"uavx::step_kernel<{N if N in (1, 2, 4, 8) else 0},false>"
    env.reset()

    R = args.ring
    mode = args.mode
    graph = None
    if mode == "graph":
        # capture R consecutive steps (each reading its own action batch) into one hipGraph
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            for i in range(3):
                step(ring[i % R])
        torch.cuda.current_stream(device).wait_stream(side)
        try:
            graph = torch.cuda.CUDAGraph()
            # thread_local: the RCCL watchdog thread of a multi-rank run may touch the runtime during capture
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                for i in range(R):
                    step(ring[i])
        except Exception as exc:  # eager stepping is GPU-bound as well (4.3 us host cost per call): fall back
            print(f"[bench] hipGraph capture failed ({type(exc).__name__}: {exc}); using --mode launch", file=sys.stderr)
            graph, mode = None, "launch"
            torch.cuda.synchronize(device)
    if graph is not None:
        def run(nsteps):
            full, rem = divmod(nsteps, R)
            for _ in range(full):
                graph.replay()
            for i in range(rem):
                step(ring[i])
    else:
        def run(nsteps):
            for i in range(nsteps):
                step(ring[i % R])

    run(W)
    if distributed:
        gather_episode_metrics(env.metrics() if args.world == "multi" else env.get_state()["counters"], dst=0)  # warm the communicator
    torch.cuda.synchronize(device)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize(device)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    run(K)
    ev1.record()
    counters = env.metrics() if args.world == "multi" else env.get_state()["counters"]
    gathered = gather_episode_metrics(counters, dst=0) if distributed else counters
    torch.cuda.synchronize(device)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)  # HIP events on the stream the kernels were launched on

    if distributed:
        t = torch.tensor([elapsed, dev_ms], dtype=torch.float64, device="cpu" if rehearsal else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, dev_ms = float(t[0]), float(t[1])

    if rank == 0:
        total_envs = E * world
        value = total_envs * K / elapsed
        bytes_per_launch = bytes_per_env_step * E
        kernel_s = dev_ms * 1e-3 / K  # average per-step device time over the timed region
        achieved = bytes_per_launch / kernel_s / 1e9
        summ = summarize_metrics(gathered, N) if args.world == "multi" else {"mean_steps": float(gathered[:, 0].double().mean())}
        if args.fused:
            summ["ended_episodes"] = env.evaluation_summary()
        traffic = measured_traffic(E, N) if (args.world == "multi" and not args.fused) else None
        world_name = "MultiUAVWorld2D" if args.world == "multi" else "UAVWorld2D"
        cfg_tag = "BASELINE.json configs[2]" if (args.world == "multi" and E == 65536 and N == 4) else "non-headline size"
        line = {
            "metric": "env-steps/s", "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": elapsed * 1e3 / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32+f64", "dtype_note": "float32 positions / observations / rewards, float64 velocities, as the reference",
            "data": "synthetic",
            "config": {"workload": f"{E} envs x {N} UAVs per GPU ({cfg_tag}), {world_name} defaults, "
                                   f"polar U(-1,1)^2 actions from a {R}-batch HBM ring, mode={mode}"
                                   + (", fused step_ex (polar conversion + auto-reset + episode stats)" if args.fused else ""),
                       "envs_per_gpu": E, "agents": N, "parallelism": f"env-index shard x{world}", "mode": mode},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic[0] if traffic else None,
                         "traffic_source": traffic[1] if traffic else None,
                         "bytes_per_launch": bytes_per_launch, "kernel_us": kernel_s * 1e6,
                         "kernel": kernel_name},
            "episode_metrics": summ,
        }
        if world == 1 and not args.no_cpu_baseline and args.world == "multi":
            line["cpu_baseline"] = cpu_baseline(N)
        print(json.dumps(line), flush=True)
    env.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
