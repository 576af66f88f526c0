"""The kernels no headline shape runs, timed once so that their cost is on record (round-3 review, weak #8):
  * float64-position episodes (`step64_kernel`: reset(circular=True) / float64 pokes; one thread per env, untuned by design);
  * `uavx_step_k` (K steps per launch, state in registers) on the 8-UAV specialisation and on the runtime-N path, whose register
    shape is poor (105 VGPRs at N = 8, 75-79 VGPRs + scalars parked in VGPR lanes on the runtime-N path).
usage: python tools/exp_offheadline.py            (GPU box; prints JSON lines)"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_uav_collision_avoidance_amd import BatchedMultiUAVWorld2D
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(reps):
        fn()
    ev1.record(); torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) * 1e3 / reps


for E, N in ((65536, 4), (65536, 8), (16384, 6)):
    env = BatchedMultiUAVWorld2D(E, num_agents=N, device=dev)
    act = (torch.rand((8, E, N, 2), generator=g, device=dev) * 2 - 1) * 8
    env.reset()
    us32 = timed(lambda: env.step(act[0]), 300)
    env.reset_circular()                       # float64-position mode (MUW:157-163)
    assert env.position_mode == "float64"
    us64 = timed(lambda: env.step(act[1]), 100)
    env.close()
    print(json.dumps(dict(kernel="step64_kernel", envs=E, agents=N, us_per_step_float64_positions=us64, us_per_step_float32=us32,
                          ratio=us64 / us32)), flush=True)
for E, N in ((65536, 8), (65536, 4), (65536, 5), (32768, 10), (16384, 24)):
    for K, tape_out in ((32, True), (32, False)):
        env = BatchedMultiUAVWorld2D(E, num_agents=N, device=dev)
        env.reset()
        tape = (torch.rand((K, E, N, 2), generator=g, device=dev) * 2 - 1) * 10
        us1 = timed(lambda: env.step(tape[0]), 200)
        usk = timed(lambda: env.step_k(tape, tape_out=tape_out), 24) / K
        env.close()
        print(json.dumps(dict(kernel="step_k_kernel", envs=E, agents=N, K=K, tape_out=tape_out, us_per_step=usk, us_single_step_launch=us1,
                              G_env_steps_per_s=E / usk / 1e3)), flush=True)
