"""K steps per launch (uavx_step_k): throughput of the step body without kernel boundaries -- what amortises the launch
latency that bounds small batches (BASELINE configs[1]: 4 096 envs x 1 UAV).   usage: python tools/exp_stepk.py [E] [N]"""
import sys, os, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_uav_collision_avoidance_amd import BatchedMultiUAVWorld2D
dev = torch.device("cuda", 0)
E, N = (int(x) for x in (sys.argv[1:3] + ["65536", "4"][len(sys.argv) - 1:]))
g = torch.Generator(device=dev).manual_seed(1)
for K, tape_out in ((1, False), (8, True), (32, True), (32, False), (128, False)):
    env = BatchedMultiUAVWorld2D(E, num_agents=N, device=dev)
    env.reset()
    tape = (torch.rand((K, E, N, 2), generator=g, device=dev) * 2 - 1) * 10
    reps = max(4, 2048 // K)
    out = env.step_k(tape, tape_out=tape_out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = env.step_k(tape, tape_out=tape_out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (reps * K)
    print(json.dumps(dict(envs=E, agents=N, K=K, tape_out=tape_out, us_per_step=dt * 1e6, G_env_steps_per_s=E / dt / 1e9)), flush=True)
    env.close()
