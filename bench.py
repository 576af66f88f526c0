#!/usr/bin/env python3
"""bench.py — env-steps/s of the batched MultiUAVWorld2D step path on MI355X.

    python bench.py --gpus 1 --steps 2000 --warmup 200
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N --steps K --warmup W      # same thing: starts the launcher line above as a child process

One "step" = one launch of the fused HIP step kernel over the whole batch (all agents of all envs
advanced once, MUW:177-241).  Workload = BASELINE.json configs[2], the configuration the metric is
quoted on: 65 536 envs x 4 UAVs per GPU (weak scaling: every rank owns 65 536 envs, sharded by env
index with no step-path communication).  State, action ring and outputs are resident in HBM before a
timed region starts.

Timing contract.  W untimed warm-up steps, then the K-step region is timed `--repeats` times (default 5;
SURVEY.md §8d / BASELINE.md §5.3: median of 5): every region is bracketed by barrier + synchronize on both
sides, holds EXACTLY K step launches and nothing else (the episode-metrics read and the RCCL gather are
timed separately as `gather_ms`; they are per-episode work, not per-step work), per region the MAX over
ranks is taken, and `ms_per_step` / `value` are the MEDIAN region.  In graph mode the K launches are replayed
from captured hipGraphs (chunks of `--ring` steps + one graph for the remainder, so that any K is a pure
replay); `config.mode` says what actually ran and `config.graph_replays` how many replays one region held.
"""
import argparse
import json
import os
import statistics
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (about 6.3 TB/s achievable)
L3_BYTES = 256 * 2 ** 20       # Infinity Cache
SIMDS = 256 * 4                # 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9               # peak engine clock
VALU_CYCLES = 4                # a wave64 float32 VALU instruction occupies its SIMD for 4 cycles (float64 and transcendental ones longer)


def algorithmic_bytes_per_env_step(n_agents, n_bodies=0, active_agents=None, active_bodies=None):
    """lean f32 SoA, 41 B read + 66 B write per agent-step + 24 B per-env counters; a scripted
    body (BASELINE configs[4]) is a 16 B read + 16 B write record (no action, no observation / reward / done).
    Under a curriculum only the slots a level switches on count in full (mean over the levels, which envs draw uniformly):
    a parked learner still has its observation / reward / done rows written (45 B) and nothing else, a body that is
    switched off moves nothing."""
    na = n_agents if active_agents is None else active_agents
    nb = n_bodies if active_bodies is None else active_bodies
    return 107 * na + 45 * (n_agents - na) + 24 + 32 * nb


def measured_pmc(kernel_name, shape):
    """The entry of this kernel and workload shape in the committed rocprofv3 PMC passes (profiles/*_pmc_summary.json, written
    by tools/summarize_profiles.py from the separate counter runs of tools/profile_round.sh), or None.  A summary is only
    used when it was taken from THIS build of the kernels (hash of csrc/ + include/) and for this kernel and workload
    shape ("ExN[+B][f]..."): a stale file yields None rather than numbers that no longer describe the code."""
    import glob
    from gym_uav_collision_avoidance_amd import _lib
    sha = _lib.source_hash()
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary*.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        for entry in (d.get("kernels") or [d]):
            meta = entry.get("_meta", {})
            if (meta.get("csrc_sha") == sha and meta.get("shape") == shape
                    and str(meta.get("kernel", "")).replace("void ", "").startswith(kernel_name)):
                best = (entry, os.path.basename(f))
    return best


def measured_traffic(kernel_name, shape):
    """HBM bytes per step launch (FETCH_SIZE / WRITE_SIZE passes, corrected as MI355X_MICROARCH.md prescribes) -> (bytes, file)."""
    hit = measured_pmc(kernel_name, shape)
    if hit and "hbm_traffic_bytes_per_launch" in hit[0]:
        return hit[0]["hbm_traffic_bytes_per_launch"]["total"], hit[1]
    return None


def valu_roofline(kernel_name, shape, kernel_s):
    """The instruction-issue ceiling next to the bandwidth one, for shapes that are not bandwidth-bound: SQ_INSTS_VALU (wave-level
    vector instructions of one launch, PMC) x 4 cycles each, spread over 1 024 SIMDs at 2.4 GHz = the time the launch needs
    for VALU issue alone if every SIMD issued one vector instruction every 4 cycles without a bubble (a floor on that time:
    float64 and transcendental instructions take longer).  frac = that time / the launch's duration."""
    hit = measured_pmc(kernel_name, shape)
    if not hit or "SQ_INSTS_VALU" not in hit[0] or "SQ_WAVES" not in hit[0]:
        return None
    insts, waves = hit[0]["SQ_INSTS_VALU"]["median"], hit[0]["SQ_WAVES"]["median"]
    t = insts * VALU_CYCLES / (SIMDS * CLOCK_HZ)
    return dict(bound="valu", valu_insts_per_wave=insts / waves, waves=waves, issue_time_us=t * 1e6, kernel_us=kernel_s * 1e6,
                frac=t / kernel_s, source=hit[1],
                note=f"SQ_INSTS_VALU x {VALU_CYCLES} cycles / ({SIMDS} SIMDs x {CLOCK_HZ / 1e9:.1f} GHz): the floor of the launch's "
                     "vector-issue time; float64 / transcendental instructions occupy a SIMD longer than 4 cycles")


def polar_actions(gen, shape, vmax_norm, device):
    """a ~ U(-1,1)^2 mapped like the trainers do (test_sac_multi.py:77-80)."""
    a = torch.rand(shape + (2,), generator=gen, device=device) * 2 - 1
    v = (a[..., 0] / 2 + 0.5) * vmax_norm
    th = a[..., 1] * np.pi
    return torch.stack([v * torch.cos(th), v * torch.sin(th)], dim=-1).contiguous()


def cpu_baseline(n_agents, n_bodies=0, budget_s=18.0):
    """The CPU oracle (a C port of the reference's step, oracle/uavx_oracle.c) timed on this box's host cores on a bounded
    sample of the same workload: same reset seed and action distribution, OpenMP over envs on ALL host cores (SURVEY.md 8d:
    that is the baseline `speedup_vs_cpu_baseline` is taken against), with the 16-thread and the single-thread (the
    reference's own shape: it cannot use a second core) figures beside it.  The batch is sized to the thread count
    (>= 64 envs per thread, at least 4 096), so that every core has work."""
    import oracle
    oracle.build()
    host = os.cpu_count() or 1
    try:
        host = len(os.sched_getaffinity(0))     # the cores this process may be scheduled on
    except (AttributeError, OSError):
        pass
    quota = None
    try:                                         # ... and the CPU time it is granted (cgroup v2): "quota period" or "max period"
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(round(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    if quota is not None and quota < host:       # threads beyond the quota are only throttled: they measure the throttle
        host = quota
    rng = np.random.default_rng(1234)
    out, sizes = {}, {}
    runs = [("all", host)] + ([("16", 16)] if host > 16 else []) + [("1", 1)]
    for label, threads in runs:
        E = max(4096, 64 * threads)
        a = rng.uniform(-1, 1, size=(4, E, n_agents, 2))
        v = (a[..., 0] / 2 + 0.5) * np.sqrt(200.0)
        acts = np.stack([v * np.cos(a[..., 1] * np.pi), v * np.sin(a[..., 1] * np.pi)], axis=-1)
        orc = oracle.OracleMulti(num_envs=E, num_agents=n_agents, nthreads=threads, num_bodies=n_bodies)
        orc.reset_philox(0)
        for k in range(3):
            orc.step(acts[k % 4])
        t0 = time.perf_counter()
        steps = 0
        while True:
            for k in range(4):
                orc.step(acts[k])
            steps += 4
            if time.perf_counter() - t0 > budget_s / len(runs):
                break
        out[label] = E * steps / (time.perf_counter() - t0)
        sizes[label] = E
    cores = host
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    res = dict(value=out["all"], unit="env-steps/s", cores=cores, kind="port", cpu_model=model, host_cpus=os.cpu_count(),
               sample=f"{sizes['all']} envs x {n_agents} UAVs" + (f" + {n_bodies} scripted bodies" if n_bodies else "") +
                      f", oracle/uavx_oracle.c with OpenMP over envs on all {cores} host threads, ~{budget_s / len(runs):.0f} s per run",
               single_thread_value=out["1"], threads16_value=out.get("16"), cpu_quota=quota,
               note="value = every core this job may use: min(schedulable CPUs, cgroup cpu.max quota) -- on a one-GPU box of this pool "
                    "that is 16 of the host's 256 hardware threads (more threads are throttled, not run); threads16_value / "
                    "single_thread_value: the same port on 16 threads / on one (the reference itself is single-threaded Python: "
                    "reference_python); value_if_whole_host: linear extrapolation to host_cpus, an upper bound never measured")
    res["value_if_whole_host"] = out["all"] * (os.cpu_count() or cores) / cores
    # the unmodified Python reference cannot travel to the GPU box: its timing is taken in the build container by
    # tools/time_reference.py and attached here with its provenance
    import glob
    refs = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_reference_cpu.json")))
    if refs:
        try:
            ref = json.load(open(refs[-1]))
            row = next((r for r in ref["rows"] if r["world"] == "MultiUAVWorld2D" and r["num_agents"] == n_agents), None)
            if row and not n_bodies:
                res["reference_python"] = dict(value=row["env_steps_per_s"], unit="env-steps/s", cores=1,
                                               provenance=f"{os.path.basename(refs[-1])}: {ref['provenance']}")
        except Exception:
            pass
    return res


class Stepper:
    """K step launches as pure hipGraph replays (chunks of R steps + one graph for the remainder) or, in launch
    mode / if capture fails, as K eager ctypes -> hipLaunchKernel calls."""

    def __init__(self, step, ring, device, mode):
        self.step, self.ring, self.R, self.device, self.mode = step, ring, int(ring.shape[0]), device, mode
        self.graphs = {}

    def _capture(self, n):
        g = torch.cuda.CUDAGraph()
        # thread_local: the RCCL watchdog thread of a multi-rank run may touch the runtime during capture
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            for i in range(n):
                self.step(self.ring[i])
        return g

    def prepare(self, K):
        """Captures what run(K) needs.  Capture executes nothing, so the env state is untouched."""
        if self.mode != "graph":
            return
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):   # a few eager calls on the capture stream first (lazy init outside the capture)
            for i in range(3):
                self.step(self.ring[i % self.R])
        torch.cuda.current_stream(self.device).wait_stream(side)
        self.capture_warmup = 3
        full, rem = divmod(K, self.R)
        try:
            for n in ({self.R} if full else set()) | ({rem} if rem else set()):
                if n not in self.graphs:
                    self.graphs[n] = self._capture(n)
        except Exception as exc:  # eager stepping is GPU-bound as well (4.3 us host cost per call): fall back
            print(f"[bench] hipGraph capture failed ({type(exc).__name__}: {exc}); using --mode launch", file=sys.stderr)
            self.graphs, self.mode = {}, "launch"
            torch.cuda.synchronize(self.device)

    def replays(self, K):
        if self.mode != "graph":
            return 0
        full, rem = divmod(K, self.R)
        return full + (1 if rem else 0)

    def run(self, K):
        if self.mode == "graph":
            full, rem = divmod(K, self.R)
            for _ in range(full):
                self.graphs[self.R].replay()
            if rem:
                self.graphs[rem].replay()
        else:
            for i in range(K):
                self.step(self.ring[i % self.R])


def event_passes(K):
    """How many back-to-back passes of K launches one device-time region brackets: at least 200 launches, so that the ~10 us
    device-side start of a graph replay -- which rocprofv3's per-kernel average does not contain -- is a small share of the
    bracket even when K is 20 (the wall-clock regions `value` comes from are exactly K launches, always).  ONE pass from
    K = 200 on: an event region then runs 2 K launches (the untimed pass + the timed one), so `--steps 2000` spends
    5 x 4 000 launches on device time beside the 5 x 2 000 of the wall-clock regions."""
    return max(1, -(-200 // max(1, K)))


def timed_regions(stepper, K, repeats, device, dist=None, events=False):
    """`repeats` regions of exactly K step launches, each bracketed by barrier + synchronize on both sides.  events=False:
    wall-clock regions (nothing but the K launches between the two clock reads): what `value` / `ms_per_step` come from.
    events=True: device time of K step launches from HIP events on the launch stream, in regions of their own (the two event
    records would otherwise sit inside the wall-clock region), each preceded by an untimed pass of the same K launches so that
    the events bracket kernels, not the host's enqueue latency, and spanning event_passes(K) passes (>= 200 launches) per bracket.
    Returns the per-region times of this rank (s, or ms per K launches)."""
    out = []
    for _ in range(repeats):
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)
        if events:
            # device time of K step launches: an untimed pass of the same K launches goes first and the two events and the
            # timed pass are enqueued BEHIND it, while it runs, so the host's latency of enqueueing on an idle stream is not
            # inside the bracket.  What a bracket still holds beside its kernels is the DEVICE-side start of every graph
            # replay (~10 us for an isolated one, ~6 us back to back; a lone 20-step replay: 20 x 5.8 us of kernels come out as
            # 128 us) -- rocprofv3's per-kernel average does not contain it; hence >= 200 launches per bracket, and a 20-step
            # run also reports roofline_steady (1000-step regions)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = event_passes(K)
            stepper.run(K)
            ev0.record()
            for _ in range(reps):
                stepper.run(K)
            ev1.record()
            torch.cuda.synchronize(device)
            out.append(ev0.elapsed_time(ev1) / reps)  # HIP events on the stream the kernels were launched on; ms per K launches
        else:
            done = torch.cuda.Event()
            t0 = time.perf_counter()
            stepper.run(K)
            done.record()
            while not done.query():     # spin instead of a blocking wait: the wake-up latency of a sleeping host thread
                pass                    # (tens of us) would otherwise be charged to a K-step region of ~100 us
            out.append(time.perf_counter() - t0)
            torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
    return out


def single_step_latency(step, ring, device, trials=200):
    """Host-visible latency of ONE step from an idle stream: launch -> kernel done (median, us).  What a caller that
    needs every result before it can act (the single-env façades, small batches) pays per step."""
    lat = []
    ev = torch.cuda.Event()
    for i in range(trials):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        step(ring[i % ring.shape[0]])
        ev.record()
        while not ev.query():
            pass
        lat.append((time.perf_counter() - t0) * 1e6)
    return statistics.median(lat)


def reset_observe_point(env, N, bodies, device, reps=30):
    """The other two launches of the path (MUW:116-175 reset, MUW:60-109 _get_obs) on the bench's own batch: a full-batch
    uavx_reset (Philox draw + accept / reject chain of every env) and uavx_observe, device time per launch over HIP events.
    Informational: an RL loop resets inside step_ex (--fused) and reads observations from the step launch."""
    E = env.num_envs
    out = {}
    for name, fn, per_agent, per_env in (("reset", lambda: env.reset(), 8 + 16 + 16 + 40, 16 + 16 + 24 * bodies),
                                         ("observe", lambda: env.observe(), 8 + 16 + 16 + 40, 16 * bodies)):
        for _ in range(3):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(device)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize(device)
        us = a.elapsed_time(b) * 1e3 / reps
        nbytes = (per_agent * N + per_env) * E
        out[name] = {"us_per_launch": us, "algorithmic_bytes": nbytes, "achieved_GBps": nbytes / us / 1e3,
                     "frac_of_hbm_peak": nbytes / us / 1e3 / HBM_PEAK_GBS}
    out["note"] = ("full-batch uavx_reset (written: 40 B of state + the 40 B observation per agent, the env record; the time is Philox + "
                   "the accept / reject chain, not bytes) and uavx_observe (40 B read + 40 B written per agent); eager launches back to back, HIP events")
    return out


def large_batch_point(N, device, gen, bodies=0, E=1 << 20, steps=300, warmup=60):
    """The same step kernel on a batch whose working set (state + double-buffered outputs + action ring, ~0.9 GB at
    N=4) is several times the 256 MiB Infinity Cache, i.e. a launch that really streams from HBM."""
    from gym_uav_collision_avoidance_amd import BatchedMultiUAVWorld2D
    env = BatchedMultiUAVWorld2D(E, num_agents=N, device=device, seed=0, **(dict(num_bodies=bodies) if bodies else {}))
    L = N
    ring = polar_actions(gen, (6, E, L), float(np.sqrt(200.0)), device)
    env.reset()
    st = Stepper(env.step, ring, device, "graph")
    st.prepare(steps)
    st.run(warmup)
    devs = timed_regions(st, steps, 3, device, events=True)
    kernel_s = statistics.median(devs) * 1e-3 / steps
    b = algorithmic_bytes_per_env_step(N, bodies) * E
    ws = working_set_bytes(E, N + bodies, L, ring.shape[0])
    env.close()
    del ring, env
    torch.cuda.empty_cache()
    shape = f"{E}x{N}" + (f"+{bodies}" if bodies else "")
    nt = N if (N in (1, 2, 4, 5, 8) and not bodies) else 0
    traffic = measured_traffic(f"uavx::step_kernel<{nt}", shape)
    return dict(bound="hbm", achieved=b / kernel_s / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", frac=b / kernel_s / 1e9 / HBM_PEAK_GBS,
                traffic=traffic[0] if traffic else None, traffic_source=traffic[1] if traffic else None,
                achieved_traffic=(traffic[0] / kernel_s / 1e9) if traffic else None,
                bytes_per_launch=b, kernel_us=kernel_s * 1e6, envs=E, agents=N, steps=steps,
                working_set_bytes=ws, note="working set >> 256 MiB Infinity Cache: HBM-resident stream; achievable HBM rate on MI355X is ~6.3 TB/s "
                                           "(0.79 of the 8 TB/s spec peak).  Measured separately, not in this run (profiles/r04_ab_notes.md section 4): a "
                                           "launch of this size writes about as much as the Infinity Cache holds and is still helped by it -- the same "
                                           "read / write mix streams at 6.05-6.1 TB/s of REAL bytes, flat, from 1 to 4 GB per launch "
                                           "(tools/micro/stream_mix.hip), and this kernel at 2-4 Mi envs at 0.65-0.69 of the peak in algorithmic bytes")


def open_loop_point(E, N, device, gen, K=32, reps=40):
    """uavx_step_k: K consecutive steps of the same step body in ONE launch from an action tape, every step's observations /
    rewards / dones written out (open-loop rollouts: scripted or pre-drawn commands).  No kernel boundary between steps and the
    agent state stays in registers, so what is left is the step body itself: the figure to hold against the vector-instruction
    ceiling.  Informational (a trainer with a policy in the loop cannot use it), never `value`."""
    from gym_uav_collision_avoidance_amd import BatchedMultiUAVWorld2D
    env = BatchedMultiUAVWorld2D(E, num_agents=N, device=device, seed=0)
    env.reset()
    tape = polar_actions(gen, (K, E, N), float(np.sqrt(200.0)), device)
    env.step_k(tape, tape_out=True)
    torch.cuda.synchronize(device)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(reps):
        out = env.step_k(tape, tape_out=True)
    ev1.record()
    torch.cuda.synchronize(device)
    us = ev0.elapsed_time(ev1) * 1e3 / (reps * K)
    env.close()
    del out, tape
    torch.cuda.empty_cache()
    res = {"K": K, "tape_out": True, "us_per_step": us, "value": E / us * 1e6, "unit": "env-steps/s",
           "note": "uavx_step_k: K steps per launch from an action tape, all outputs of every step written; informational"}
    rv = valu_roofline(f"uavx::step_kernel<{N if N in (1, 2, 4, 5, 8) else 0}", f"{E}x{N}", us * 1e-6)
    if rv is not None:   # same step body: the single-step kernel's instruction count per wavefront stands for it
        res["valu_frac"] = rv["frac"]
    return res


def split_batch_point(E, N, device, gen, ring_len, bodies=0, chains=2, steps=1000, warmup=100):
    """The same batch as `chains` handles of E / chains envs, each replaying its OWN hipGraph on its OWN stream: independent
    step chains (what a trainer that double-buffers env halves has: the policy works on one half while the other steps).
    The launch-to-launch boundary of one chain then hides under the kernels of the other.  (Parallel branches inside ONE
    captured graph do not overlap on this runtime; separate graphs on separate streams do.)"""
    from gym_uav_collision_avoidance_amd import BatchedMultiUAVWorld2D
    e = E // chains
    kw = dict(num_bodies=bodies) if bodies else {}
    envs = [BatchedMultiUAVWorld2D(e, num_agents=N, device=device, env_offset=k * e, seed=0, **kw) for k in range(chains)]
    rings = [polar_actions(gen, (ring_len, e, N), float(np.sqrt(200.0)), device) for _ in range(chains)]
    streams = [torch.cuda.Stream(device) for _ in range(chains)]
    graphs = []
    for k in range(chains):
        envs[k].reset()
        with torch.cuda.stream(streams[k]):
            for i in range(3):
                envs[k].step(rings[k][i])
        torch.cuda.synchronize(device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=streams[k]):
            for i in range(ring_len):
                envs[k].step(rings[k][i])
        graphs.append(g)

    def burst(nsteps):
        for _ in range(nsteps // ring_len):
            for k in range(chains):
                with torch.cuda.stream(streams[k]):
                    graphs[k].replay()

    steps = max(ring_len, steps // ring_len * ring_len)
    burst(warmup)
    walls = []
    for _ in range(3):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        burst(steps)
        torch.cuda.synchronize(device)
        walls.append(time.perf_counter() - t0)
    dt = statistics.median(walls) / steps
    b = algorithmic_bytes_per_env_step(N, bodies) * e * chains
    for x in envs:
        x.close()
    del rings, envs, graphs
    torch.cuda.empty_cache()
    return dict(chains=chains, envs=e * chains, agents=N, steps=steps, us_per_step=dt * 1e6, value=e * chains / dt,
                achieved=b / dt / 1e9, frac=b / dt / 1e9 / HBM_PEAK_GBS,
                note=f"the batch as {chains} independent handles of {e} envs, one hipGraph and one stream each (wall clock over "
                     f"{steps}-step regions, all chains in flight): not `value` -- a step here is {chains} launches on {chains} queues")


def working_set_bytes(E, n_slots, n_learners, ring_len):
    """Bytes one pass over the action ring touches: state (40 B per agent slot), both obs buffers, rew, done, ring."""
    return E * (n_slots * 40 + n_learners * (2 * 40 + 4 + 1) + ring_len * n_learners * 8) + E * 48


def self_launch(n_ranks, argv):
    """`python bench.py --gpus N` (N > 1) without a launcher around it: run
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <argv>`
    as a child process and return its exit code.  Signals that end this process are passed on to the child's process
    group so that a driver-side timeout does not leave ranks behind: the handlers are installed BEFORE the child exists
    (a signal that arrives in between is remembered and delivered as soon as it does).  The rendezvous port comes from
    bind / close, so another process can take it before torchrun binds it: a child that dies on exactly that (address in
    use) is started once more on a fresh port."""
    import collections
    import signal
    import socket
    import subprocess
    import threading
    state = {"proc": None, "pending": None}

    def forward(signum, _frame):
        proc = state["proc"]
        if proc is None:
            state["pending"] = signum
            return
        try:
            os.killpg(proc.pid, signum)
        except OSError:
            pass
    for sig in (signal.SIGINT, signal.SIGTERM, signal.SIGHUP):
        signal.signal(sig, forward)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # see main(): the ranks set the same default themselves
    rc = 1
    for attempt in range(2):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
        if state["pending"] is not None:
            return 128 + state["pending"]
        tail = collections.deque(maxlen=400)   # the child's stderr streams through; its tail tells a port clash from any other failure

        def pump(pipe):
            for line in iter(pipe.readline, ""):
                tail.append(line)
                sys.stderr.write(line)
                sys.stderr.flush()
        state["proc"] = subprocess.Popen(cmd, env=env, start_new_session=True, stderr=subprocess.PIPE, text=True, bufsize=1)
        if state["pending"] is not None:
            forward(state["pending"], None)
        reader = threading.Thread(target=pump, args=(state["proc"].stderr,), daemon=True)
        reader.start()
        rc = state["proc"].wait()
        reader.join(timeout=10)
        state["proc"] = None
        text = "".join(tail)
        clash = rc != 0 and ("EADDRINUSE" in text or "ddress already in use" in text)
        if not clash or state["pending"] is not None:
            break
        print(f"[bench] rendezvous port {port} was taken before torchrun bound it; starting the ranks once more", file=sys.stderr)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--repeats", type=int, default=5, help="timed K-step regions; the median is reported")
    ap.add_argument("--envs", type=int, default=None, help="envs per GPU (default 65 536: BASELINE configs[2] on every GPU; "
                                                          "configs[3] is --gpus 8 --envs 32768)")
    ap.add_argument("--agents", type=int, default=4, help="learning UAVs per env")
    ap.add_argument("--bodies", type=int, default=0,
                    help="scripted dynamic obstacles per env, stepped in-kernel (BASELINE configs[4]: --agents 8 --bodies 16)")
    ap.add_argument("--world", choices=("multi", "uw"), default="multi",
                    help="multi: MultiUAVWorld2D (headline); uw: UAVWorld2D (single UAV, BASELINE configs[1] family)")
    ap.add_argument("--fused", action="store_true",
                    help="multi only: drive uavx_step_ex (polar action conversion, agent0-done auto-reset with a "
                         "1500-step cap, episode statistics) instead of the bare step")
    ap.add_argument("--curriculum", type=int, default=0, metavar="LEVELS",
                    help="multi only: install this many curriculum levels (box 30..60 m, d_sense 10..18 m, fewer active learners / "
                         "bodies on the lower ones); every (re-)initialised env draws its level uniformly over all of them")
    ap.add_argument("--replay", action="store_true",
                    help="with --fused: obs / reward / done / episode flags land zero-copy in a DeviceReplay ring of --ring slots "
                         "and the commands are read from its action slots")
    ap.add_argument("--packed-flags", action="store_true",
                    help="with --fused: UAVX_FLAGS_IN_DONE (ABI v3): reset_mask / ended / truncated ride in bits 1..3 of every env's "
                         "first done byte instead of three one-byte-per-env arrays")
    ap.add_argument("--prefetch", type=int, default=None,
                    help="with --fused: uavx_set_prefetch cadence (one staging workgroup per this many env-workgroups; 0 = off); "
                         "default: the handle's own")
    ap.add_argument("--cfg5", action="store_true",
                    help="BASELINE configs[4] as designed: --agents 8 --bodies 16 --fused --curriculum 4 --replay")
    ap.add_argument("--mode", choices=("graph", "launch"), default="graph",
                    help="graph: steps replayed from captured hipGraphs (one kernel node per step); "
                         "launch: one ctypes->hipLaunchKernel per step")
    ap.add_argument("--ring", type=int, default=50, help="distinct action batches resident in HBM")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-large", action="store_true",
                    help="skip the extra measurements: the HBM-sized second roofline point (1 Mi envs) and roofline_steady")
    args = ap.parse_args()
    # dmabuf IPC.  The host driver of this pool supports no other kind: without this variable RCCL (and any sharing of device
    # memory between processes) fails with `hipIpcGetMemHandle: invalid argument` (the pool's environment notes; exported there
    # already).  Set HERE, before the first call that initialises the GPU, so that ranks started by an outer torchrun and ranks
    # started by self_launch() run in the same environment; a value the caller exported wins.  Reported in the line (`hsa_env`).
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    envs_defaulted = args.envs is None
    if envs_defaulted:
        args.envs = 65536
    if args.cfg5:
        args.agents, args.bodies, args.fused, args.replay = 8, 16, True, True
        args.curriculum = args.curriculum or 4
    if args.replay and not args.fused:
        raise SystemExit("--replay needs --fused (the replay ring is fed by uavx_step_ex)")
    if args.world == "uw" and (args.curriculum or args.replay or args.bodies):
        raise SystemExit("--curriculum / --replay / --bodies belong to --world multi")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU).  This process has
        # not touched the GPU yet (importing torch does not initialise HIP) and never will: the ranks are CHILD processes
        # (no exec from here), their stdout is ours, so rank 0's JSON line streams through, and we leave with their exit code.
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    if args.gpus > 1 and envs_defaulted and int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench] --gpus {args.gpus} with the default 65 536 envs PER GPU (configs[2] on every GPU, {65536 * args.gpus} in all); "
              f"BASELINE configs[3] (262 144 envs over 8 GPUs) is --gpus 8 --envs 32768", file=sys.stderr)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} was started with WORLD_SIZE={world}: the launcher must start {args.gpus} ranks")
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
    dist = None
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            if torch.cuda.device_count() < world:
                raise SystemExit(f"--gpus {world}: this process sees {torch.cuda.device_count()} GPU(s); one rank per GPU is the "
                                 f"contract (UAVX_REHEARSAL=1 shares GPU 0 over gloo, for rehearsals only)")
            dist.init_process_group("nccl", device_id=device)
        if dist.get_world_size() != world:
            raise SystemExit(f"process group of {dist.get_world_size()} ranks, WORLD_SIZE={world}")

    from gym_uav_collision_avoidance_amd import BatchedMultiUAVWorld2D, BatchedUAVWorld2D
    from gym_uav_collision_avoidance_amd.replay import DeviceReplay
    from gym_uav_collision_avoidance_amd.sharding import gather_episode_metrics, summarize_metrics

    E, N, B, K, W = args.envs, args.agents, args.bodies, args.steps, args.warmup
    gen = torch.Generator(device=device).manual_seed(1234 + rank)
    nt = lambda n: n if n in (1, 2, 4, 5, 8) else 0
    ext = False
    if args.world == "uw":
        N, B = 1, 0
        env = BatchedUAVWorld2D(E, device=device, env_offset=rank * E, seed=0)
        ring = polar_actions(gen, (args.ring, E), float(np.sqrt(288.0)), device)  # ||(12,12)||, test_sac.py:77
        step = env.step
        bytes_per_env_step, kernel_name = 93, "uavx::uw_step_kernel"
    else:
        env = BatchedMultiUAVWorld2D(E, num_agents=N, device=device, env_offset=rank * E, seed=0,
                                     **(dict(num_bodies=B) if B else {}))
        bytes_per_env_step = algorithmic_bytes_per_env_step(N, B)
        ext = bool(B or args.curriculum)    # the kernel variant with bodies / levels (runtime agent count)
        bytes_note = ""
        if args.curriculum:
            n = args.curriculum
            f = lambda a, b, k: a + (b - a) * k / max(1, n - 1)
            levels = [dict(x_size=f(30.0, 60.0, k), y_size=f(30.0, 60.0, k), collider_radius=1.0, d_sense=f(10.0, 18.0, k),
                           n_active=max(1, round(f(N / 2, N, k))), b_active=round(f(B / 4, B, k))) for k in range(n)]
            env.set_curriculum(levels, lo=0, hi=n - 1)
            na, nb = sum(l["n_active"] for l in levels) / n, sum(l["b_active"] for l in levels) / n
            bytes_per_env_step = algorithmic_bytes_per_env_step(N, B, na, nb)
            bytes_note = (f"; algorithmic bytes count the slots the curriculum switches on ({na:g} of {N} learners, {nb:g} of {B} bodies "
                          f"on average over its {n} levels: {bytes_per_env_step:.0f} B per env-step instead of "
                          f"{algorithmic_bytes_per_env_step(N, B)})")
        if args.fused:
            if args.prefetch is not None:
                env.set_prefetch(args.prefetch)
            ring = torch.rand((args.ring, E, N, 2), generator=gen, device=device) * 2 - 1
            fused_kw = dict(polar=True, auto_reset="agent0_done", step_cap=1500, track_returns=True)
            if args.replay:
                # slot k of the ring holds the command, reward, done and flags of step k and the observation step k - 1
                # produced: the launch reads replay.act[k] and writes obs -> slot k + 1, the rest -> slot k (nothing is copied)
                replay = DeviceReplay(env, horizon=args.ring - 1, num_learners=N, packed_flags=args.packed_flags)
                replay.act.copy_(ring)
                step = lambda a: replay.step(None, **fused_kw)
            else:
                step = lambda a: env.step_ex(a, packed_flags=args.packed_flags, **fused_kw)
            kernel_name = f"uavx::step_ex_kernel<{0 if ext else nt(N)}"
        else:
            ring = polar_actions(gen, (args.ring, E, N), float(np.sqrt(200.0)), device)
            step = env.step
            kernel_name = f"uavx::step_kernel<{0 if ext else nt(N)}"
    obs0 = env.reset()
    if args.world == "multi" and args.replay:
        replay.begin(obs0)

    stepper = Stepper(step, ring, device, args.mode)
    stepper.prepare(K)
    for i in range(W):  # W untimed warm-up steps (eager launches; capture itself executes nothing)
        step(ring[i % args.ring])
    read_counters = (lambda: env.metrics()) if args.world == "multi" else (lambda: env.get_state()["counters"])
    if distributed:
        gather_episode_metrics(read_counters(), dst=0)  # warm the communicator

    walls = timed_regions(stepper, K, max(1, args.repeats), device, dist)
    devs = timed_regions(stepper, K, max(1, args.repeats), device, dist, events=True)   # device time, in regions of their own

    # episode-metrics path (test_sac_multi.py:164-165): counters read + ONE gather to rank 0, timed on its own
    torch.cuda.synchronize(device)
    if distributed:
        dist.barrier()
    tg = time.perf_counter()
    counters = read_counters()
    gathered = gather_episode_metrics(counters, dst=0) if distributed else counters
    torch.cuda.synchronize(device)
    gather_s = time.perf_counter() - tg

    ranks_info = None
    if distributed:
        # every rank's own figures first (one all_gather of a few numbers: off the clock), then the slowest rank per region
        props = torch.cuda.get_device_properties(device)
        mine = dict(rank=rank, local_rank=local_rank, device=torch.cuda.current_device(), name=props.name,
                    pci_bus_id=f"{getattr(props, 'pci_domain_id', 0):04x}:{getattr(props, 'pci_bus_id', 0):02x}:{getattr(props, 'pci_device_id', 0):02x}",
                    uuid=str(getattr(props, "uuid", "")), pid=os.getpid(),
                    ms_per_step=statistics.median(walls) * 1e3 / K, kernel_us=statistics.median(devs) * 1e3 / K,
                    hsa_enable_ipc_mode_legacy=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"))
        ranks_info = [None] * dist.get_world_size()
        dist.all_gather_object(ranks_info, mine)
        t = torch.tensor([walls, devs, [gather_s] * len(walls)], dtype=torch.float64, device="cpu" if rehearsal else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)  # per region: the slowest rank
        walls, devs, gather_s = t[0].tolist(), t[1].tolist(), float(t[2][0])

    if rank == 0:
        total_envs = E * world
        elapsed = statistics.median(walls)
        dev_ms = statistics.median(devs)
        value = total_envs * K / elapsed
        bytes_per_launch = bytes_per_env_step * E
        kernel_s = dev_ms * 1e-3 / K  # average per-step device time over the median timed region
        achieved = bytes_per_launch / kernel_s / 1e9
        summ = summarize_metrics(gathered, N) if args.world == "multi" else {"mean_steps": float(gathered[:, 0].double().mean())}
        if args.fused:
            summ["ended_episodes"] = env.evaluation_summary()
        slots = N + B
        shape = (f"{E}x{N}" + (f"+{B}" if B else "") + ("f" if args.fused else "")   # the tag tools/profile_round.sh files it under
                 + (f"c{args.curriculum}" if args.curriculum else "") + ("r" if args.replay else "") + ("p" if args.packed_flags else ""))
        traffic = measured_traffic(kernel_name, shape) if args.world == "multi" else None
        pmc_hit = measured_pmc(kernel_name, shape) if args.world == "multi" else None
        pmc_taken = (pmc_hit[0].get("_meta", {}).get("taken_utc") if pmc_hit else None)
        ws = working_set_bytes(E, slots, N, args.ring) if args.world == "multi" else E * (40 + 2 * 16 + 5 + args.ring * 8)
        if args.replay:   # every slot of the ring has its own obs / reward / done / flag rows
            ws += E * (args.ring - 1) * (N * 45 + 3)
        world_name = "MultiUAVWorld2D" if args.world == "multi" else "UAVWorld2D"
        if args.world == "multi" and N == 4 and B == 0 and total_envs == 262144 and world > 1:
            cfg_tag = f"BASELINE.json configs[3]: 262 144 envs sharded over {world} GPUs by env index"
        elif args.world == "multi" and E == 65536 and N == 4 and B == 0:
            cfg_tag = "BASELINE.json configs[2]" + (f" on each of {world} GPUs (weak scaling; configs[3] is --gpus 8 --envs 32768)" if world > 1 else "")
        elif args.world == "multi" and E == 4096 and N == 1 and B == 0 and world == 1:
            cfg_tag = "BASELINE.json configs[1]"
        elif args.world == "uw" and E == 4096 and world == 1:
            cfg_tag = "BASELINE.json configs[1], UAVWorld2D"
        elif args.world == "multi" and E == 65536 and N == 8 and B == 16:
            cfg_tag = "BASELINE.json configs[4]: 8 learners + 16 scripted dynamic obstacles (extension, parity unpinned by the reference)"
        else:
            cfg_tag = "non-headline size"
        mode = stepper.mode
        replays = stepper.replays(K)
        if mode == "graph" and replays == 0:
            mode = "launch"
        note = (f"working set {ws / 2 ** 20:.0f} MiB (state + double-buffered outputs + {args.ring}-batch action ring) "
                + ("fits the 256 MiB Infinity Cache: FETCH/WRITE_SIZE count fabric requests incl. L3 hits, so this point is "
                   "L3-resident, not HBM-streaming; see roofline_large for the HBM-sized batch" if ws < L3_BYTES else
                   "exceeds the 256 MiB Infinity Cache: HBM-streaming"))
        if args.world == "multi":
            note += bytes_note
        # how old the stepped world was in the regions the figures come from (launches since the reset at the top of this run)
        n_rep = len(walls)
        age0 = getattr(stepper, "capture_warmup", 0) + W
        ev_launches = (1 + event_passes(K)) * K
        world_age = {"wall_regions": [age0, age0 + n_rep * K], "event_regions": [age0 + n_rep * K, age0 + n_rep * K + len(devs) * ev_launches],
                     "unit": "step launches since reset"}
        if args.fused:
            ended = summ.get("ended_episodes", {}).get("episodes", 0) if isinstance(summ.get("ended_episodes"), dict) else 0
            total_launches = world_age["event_regions"][1]
            world_note = (f"fused step_ex with auto-reset: a live world -- {ended} episodes ended over the {total_launches} launches of this run "
                          f"({ended / max(1, total_launches):.1f} per launch over {E} envs)")
        else:
            world_note = (f"plain step, no auto-reset (as the reference's step): the wall-clock regions `value` comes from stepped a world "
                          f"{world_age['wall_regions'][0]}-{world_age['wall_regions'][1]} launches after its reset, the device-time regions "
                          f"`roofline` comes from one {world_age['event_regions'][0]}-{world_age['event_regions'][1]} launches old "
                          f"({n_rep} wall-clock regions of {K} launches, then the event regions); separately measured, not in this run: random commands scatter the UAVs, "
                          "and a world several thousand launches old steps 3 % (4 UAVs) to 7 % (8 + 16 bodies) cheaper than a freshly reset "
                          "one (tools/exp_world_age.py, profiles/HISTORY.md); --fused measures the live world a trainer keeps")
        line = {
            "metric": "env-steps/s", "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": elapsed * 1e3 / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32+f64", "dtype_note": "float32 positions / observations / rewards, float64 velocities, as the reference",
            "data": "synthetic", "repeats": len(walls), "repeat_ms_per_step": [w * 1e3 / K for w in walls],
            "config": {"workload": f"{E} envs x {N} UAVs" + (f" + {B} scripted bodies" if B else "") + f" per GPU ({cfg_tag}), "
                                   f"{world_name} defaults, polar U(-1,1)^2 actions from a {args.ring}-batch HBM ring, mode={mode}"
                                   + (", fused step_ex (polar conversion + auto-reset + episode stats)" if args.fused else "")
                                   + (", flags packed into the done bytes" if args.packed_flags else "")
                                   + (f", randomized-reset curriculum over {args.curriculum} levels" if args.curriculum else "")
                                   + (f", outputs written zero-copy into a {args.ring}-slot on-device replay ring" if args.replay else ""),
                       "envs_per_gpu": E, "agents": N, "bodies": B, "curriculum_levels": args.curriculum, "replay": bool(args.replay),
                       **({"body_model": "legs-v3 (include/uavx.h uavx_set_body_rule: bodies step before the learners, fixed legs; this "
                                         "build's own definition since round 3 -- round-2 figures used the costlier waypoint model)"} if B else {}), "parallelism": f"env-index shard x{world}", "mode": mode,
                       "graph_replays": replays},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic[0] if traffic else None,
                         "traffic_source": traffic[1] if traffic else None,
                         "traffic_measured_in_this_run": False,
                         "traffic_note": ("no PMC pass of this build and shape under profiles/" if not traffic else
                                          f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh, read from {traffic[1]} "
                                          f"(taken {pmc_taken or 'at an unrecorded time'}, on the kernel sources with the hash of the loaded library); "
                                          "counters cannot be collected inside a timed run"),
                         "bytes_per_launch": bytes_per_launch, "kernel_us": kernel_s * 1e6,
                         "kernel": kernel_name + ("" if kernel_name.endswith("kernel") else (",false,true>" if ext else ",false,false>")),
                         "note": note},
            "world_note": world_note, "world_age": world_age,
            "gather_ms": gather_s * 1e3,
            "gather_note": "counter read + one gather of [E,4] episode metrics to rank 0 (once per episode, not per step); "
                           "amortised over a 1500-step episode it adds gather_ms/1500 to ms_per_step",
            "ms_per_step_incl_amortised_gather": elapsed * 1e3 / K + gather_s * 1e3 / 1500.0,
            "episode_metrics": summ,
            "steps_executed": getattr(stepper, "capture_warmup", 0) + W + K * (len(walls) + (1 + event_passes(K)) * len(devs)),   # (event regions: an untimed pass + the timed ones)
        }
        line["hsa_env"] = {"HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
        if distributed:
            try:
                rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:
                rccl = None
            line["distributed"] = {"ranks_seen": dist.get_world_size(), "backend": dist.get_backend(),
                                   "rccl_version": None if rehearsal else rccl, "rehearsal_shared_gpu": rehearsal,
                                   "devices_visible": torch.cuda.device_count(),
                                   "distinct_devices": len({(r["pci_bus_id"], r["device"]) for r in ranks_info}),
                                   "per_rank": ranks_info,
                                   "per_rank_ms_per_step": [r["ms_per_step"] for r in ranks_info],
                                   "note": "value / ms_per_step: per timed region the slowest rank (all_reduce MAX), median region; "
                                           "per_rank_ms_per_step: each rank's own median region"}
        if world > 1:
            line["omitted_for_n_gpus>1"] = ["cpu_baseline", "speedup_vs_cpu_baseline", "roofline_steady", "roofline_large", "latency_us",
                                            "open_loop_step_k", "split_batch", "reset_observe"]
        if args.world == "multi":
            rv = valu_roofline(kernel_name, shape, kernel_s)
            if rv is not None:
                line["roofline_valu"] = rv
                if rv["valu_insts_per_wave"] > 600 and rv["frac"] > line["roofline"]["frac"]:
                    # the instruction-issue ceiling is the nearer one: say so, and keep the bandwidth figures beside it
                    line["roofline"]["bound_note"] = (f"nearer ceiling: vector-instruction issue ({rv['frac']:.2f} of the launch) -- see "
                                                      "roofline_valu; `bound` stays the contract's HBM figure computed from algorithmic bytes")
                    line["roofline"]["nearer_bound"] = "valu"
                else:
                    line["roofline"]["nearer_bound"] = "hbm"
        if world == 1 and K < 500 and not args.no_large:
            # A K-step region this short holds one graph replay whose launch latency (~10-15 us) is a visible share of it;
            # the same kernel over a 1000-step region of its own (not part of `value`) for the steady per-launch time
            steady = Stepper(step, ring, device, args.mode)
            steady.prepare(1000)
            sd = timed_regions(steady, 1000, 3, device, events=True)
            ks = statistics.median(sd) * 1e-3 / 1000
            line["roofline_steady"] = {"kernel_us": ks * 1e6, "achieved": bytes_per_launch / ks / 1e9, "frac": bytes_per_launch / ks / 1e9 / HBM_PEAK_GBS,
                                       "steps": 1000, "note": f"same launch sequence over 1000-step event regions; the {K}-step regions "
                                       "of the timing contract include one graph-launch latency each"}
        if world == 1:
            line["latency_us"] = {"single_step_launch_to_done": single_step_latency(step, ring, device),
                                  "kernel": kernel_s * 1e6,
                                  "note": "one step from an idle stream, host-visible; small batches are bound by this, not by bytes"}
        if world == 1 and args.world == "multi" and not args.fused:
            try:   # an extra measurement must never cost the line
                line["reset_observe"] = reset_observe_point(env, N, B, device)
            except Exception as exc:
                print(f"[bench] reset_observe skipped ({type(exc).__name__}: {exc})", file=sys.stderr)
                torch.cuda.synchronize(device)
        if world == 1 and not args.no_large and args.world == "multi" and not args.fused and E < (1 << 20):
            env.close()
            del ring
            torch.cuda.empty_cache()
            line["roofline_large"] = large_batch_point(N, device, gen, bodies=B)
            if not B and not args.curriculum:
                try:   # an extra measurement must never cost the line
                    line["open_loop_step_k"] = open_loop_point(E, N, device, gen)
                except Exception as exc:
                    print(f"[bench] open_loop_step_k skipped ({type(exc).__name__}: {exc})", file=sys.stderr)
                    torch.cuda.synchronize(device)
            if args.mode == "graph" and not args.curriculum and E % 2 == 0:
                try:   # an extra measurement must never cost the line
                    line["split_batch"] = split_batch_point(E, N, device, gen, min(args.ring, 50), bodies=B)
                except Exception as exc:
                    print(f"[bench] split_batch skipped ({type(exc).__name__}: {exc})", file=sys.stderr)
                    torch.cuda.synchronize(device)
        if world == 1 and not args.no_cpu_baseline and args.world == "multi":
            line["cpu_baseline"] = cpu_baseline(N, B)
            line["speedup_vs_cpu_baseline"] = value / line["cpu_baseline"]["value"]
            line["speedup_vs_cpu_baseline_if_whole_host"] = value / line["cpu_baseline"]["value_if_whole_host"]
        print(json.dumps(line), flush=True)
    env.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
