#!/usr/bin/env python3
"""Times the UNMODIFIED reference env (its own Python, loaded through tests/golden/ref_loader.py) in the BUILD CONTAINER
and writes profiles/<tag>_reference_cpu.json.  The reference cannot travel to the GPU box, so this is the reproducible form of
the "reference Python" baseline quoted in SURVEY.md §6 / BASELINE.md §2; bench.py attaches the file's number to its line
as cpu_baseline.reference_python with this provenance.

    python tools/time_reference.py [tag]          (needs /root/reference; about a minute)

Method: env.step only, one thread (the reference is single-threaded by design), polar U(-1,1)^2 policy
noise mapped like test_sac_multi.py:77-80, reset on dones[0] (train-loop rule :112) or a 1500-step cap, render not called."""
import json
import math
import os
import platform
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import ref_loader  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
if not ref_loader.available():
    raise SystemExit("the reference is not mounted here (build container only)")
MUW, UW, _ = ref_loader.load()


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def polar(rng, n, vmax_norm):
    a = rng.uniform(-1, 1, size=(n, 2))
    v = (a[:, 0] / 2 + 0.5) * vmax_norm
    th = a[:, 1] * math.pi
    return [np.array([v[i] * math.cos(th[i]), v[i] * math.sin(th[i])]) for i in range(n)]


def time_multi(n, budget_s):
    env = MUW(num_agents=n)
    np.random.seed(0)
    env.reset()
    rng = np.random.default_rng(1234)
    acts = [polar(rng, n, float(np.linalg.norm(env.action_space.high))) for _ in range(64)]
    for k in range(50):
        env.step(acts[k % 64])
    env.reset()
    steps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        for k in range(64):
            _, _, dones, _ = env.step(acts[k])
            steps += 1
            if dones[0] or env.steps >= 1500:
                env.reset()
    return steps / (time.perf_counter() - t0)


def time_uw(budget_s):
    env = UW()
    np.random.seed(0)
    env.reset()
    rng = np.random.default_rng(1234)
    acts = [polar(rng, 1, 12.0)[0] for _ in range(64)]
    steps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        for k in range(64):
            _, _, done, _ = env.step(acts[k])
            steps += 1
            if done or env.steps >= 1500:
                env.reset()
    return steps / (time.perf_counter() - t0)


rows = []
for n in (1, 4, 5, 8, 24):
    r = time_multi(n, 8.0)
    rows.append(dict(world="MultiUAVWorld2D", num_agents=n, env_steps_per_s=r, agent_steps_per_s=r * n))
    print(f"MultiUAVWorld2D(num_agents={n}): {r:,.0f} env-steps/s", flush=True)
r = time_uw(8.0)
rows.append(dict(world="UAVWorld2D", num_agents=1, env_steps_per_s=r, agent_steps_per_s=r))
print(f"UAVWorld2D: {r:,.0f} env-steps/s", flush=True)
out = dict(rows=rows, cores=1, cpu_model=cpu_model(), host_cpus=os.cpu_count(), python=platform.python_version(),
           numpy=np.__version__,
           provenance=f"unmodified reference env loaded by tests/golden/ref_loader.py, build container ({cpu_model()}), 1 thread, "
                      "env.step only, 8 s per row; written by tools/time_reference.py",
           method="polar U(-1,1)^2 actions (test_sac_multi.py:77-80), reset on dones[0] or 1500 steps, no render")
path = os.path.join(ROOT, "profiles", f"{tag}_reference_cpu.json")
json.dump(out, open(path, "w"), indent=1)
print("wrote", path)
