"""Diagnostic: per-wavefront timeline of ONE fused uavx_step_ex launch (library built with -DUAVX_STAMPS into tools/dbg/).
Every wavefront logs {start, mid, end, kind}: 0 step wave, 1 step wave that re-initialised an env, 10 / 11 staging wave that
scanned (found nothing / appended), 12 staging wave whose queued ids needed nothing, 13 staging wave that drew layouts.
usage: python tools/exp_stamps.py [E] [L] [B]"""
import sys, os, ctypes
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["UAVX_LIB"] = os.path.join(ROOT, "tools", "dbg", "libuavx_stamps.so")
from gym_uav_collision_avoidance_amd import _lib
_lib.LIB_PATH = os.environ["UAVX_LIB"]
from gym_uav_collision_avoidance_amd import BatchedMultiUAVWorld2D
dev = torch.device("cuda", 0)
E, N, B = (int(x) for x in (sys.argv[1:4] + ["65536", "8", "16"][len(sys.argv) - 1:]))
R = 16
g = torch.Generator(device=dev).manual_seed(1)
ring = (torch.rand((R, E, N, 2), generator=g, device=dev) * 2 - 1)
env = BatchedMultiUAVWorld2D(E, num_agents=N, num_bodies=B, device=dev)
env.set_prefetch(int(os.environ.get("UAVX_PREFETCH", "16")))
env.reset()
L = _lib.load()
buf = (ctypes.c_ulonglong * (8 * 16384))(); n = ctypes.c_uint(0)
MODES = {"all": dict(polar=True, track_returns=True, auto_reset="agent0_done", step_cap=1500),
         "plain": dict(track_returns=False), "polar_track": dict(polar=True, track_returns=True),
         "cap": dict(polar=True, track_returns=True, step_cap=1500)}
kw = MODES[os.environ.get("UAVX_STAMP_MODE", "all")]
for k in range(int(os.environ.get("UAVX_STAMP_WARM", "400"))):
    env.step_ex(ring[k % R], **kw)
L.uavx_debug_stamps(buf, ctypes.byref(n))
names = {0: "step", 1: "step+reinit", 4: "step, exact scan", 5: "step+reinit, exact scan", 10: "scan, nothing", 11: "scan, left hints", 12: "drawing workgroup, no hints", 13: "drew layouts"}
for rep in range(3):
    for k in range(3):
        env.step_ex(ring[k % R], **kw)
    torch.cuda.synchronize()
    L.uavx_debug_stamps(buf, ctypes.byref(n))       # clears the log
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(); env.step_ex(ring[5], **kw); ev1.record()
    torch.cuda.synchronize()
    launch_us = ev0.elapsed_time(ev1) * 1e3
    L.uavx_debug_stamps(buf, ctypes.byref(n))
    a = np.frombuffer(buf, dtype=np.uint64).reshape(16384, 8)[: n.value]
    a = a[(a[:, 7] >> np.uint64(63)) == 1]            # slots actually written
    a = (a & np.uint64(0x7FFFFFFFFFFFFFFF)).astype(np.int64)
    kind, xcc, blk = a[:, 7] & 255, (a[:, 7] >> 8) & 15, (a[:, 7] >> 16) & 0xFFFFFF
    dr = a[kind == 13]
    if len(dr):   # staging waves that drew: scan | fused Philox | level + candidates | chain | leg + stores
        seg = np.stack([dr[:, 1] - dr[:, 0], dr[:, 3] - dr[:, 1], dr[:, 4] - dr[:, 3], dr[:, 5] - dr[:, 4], dr[:, 2] - dr[:, 5]], axis=1)
        print("  drawing waves, ticks per stretch [scan, philox, level+candidates, chain, leg+stores]: median", np.median(seg, axis=0).astype(int).tolist(),
              "max", seg.max(axis=0).tolist())
    t0 = np.zeros(len(a), np.int64)
    for x in np.unique(xcc):                      # clocks are per XCD: time zero = first start on the same XCD
        t0[xcc == x] = a[xcc == x, 0].min()
    start, mid, end = a[:, 0] - t0, a[:, 1] - t0, a[:, 2] - t0
    print(f"--- launch {rep} ({launch_us:.1f} us between events): {len(a)} wavefronts, last end {end.max()} ticks; per-XCD wave counts {np.bincount(xcc).tolist()}")
    for kd in sorted(set(kind.tolist())):
        s = kind == kd
        d = end[s] - start[s]
        print(f"  {names.get(kd, kd):26s} n={s.sum():5d}  start med {int(np.median(start[s])):6d} max {start[s].max():6d} | life med {int(np.median(d)):6d} "
              f"p99 {int(np.percentile(d, 99)):6d} max {d.max():6d} | end med {int(np.median(end[s])):6d} p99 {int(np.percentile(end[s], 99)):6d} max {end[s].max():6d}")
    sw = kind <= 5
    for x in np.unique(xcc):
        q = sw & (xcc == x)
        lt = np.sort(start[q])
        print(f"    xcd {x}: step waves {q.sum()}, starts p50 {int(lt[len(lt) // 2])} p90 {int(lt[int(len(lt) * .9)])} p99 {int(lt[int(len(lt) * .99)])} max {lt[-1]}, "
              f"life of the first half {int(np.median((end - start)[q & (start <= lt[len(lt) // 2])]))}, of the last tenth {int(np.median((end - start)[q & (start >= lt[int(len(lt) * .9)])]))}, last end {end[q].max()}")
    # the device-wide 100 MHz clock (slot 6: first / last stamp of the wavefront, low words): the launch's dispatch timeline
    rt0, rt1 = (a[:, 6] & 0x7FFFFFFF).astype(np.int64), ((a[:, 6] >> 32) & 0x7FFFFFFF).astype(np.int64)   # (bit 63 is the slot's valid mark)
    z = rt0.min()
    rs, re = (rt0 - z) / 100.0, (rt1 - z) / 100.0     # us since the first wavefront of the launch started
    print(f"  device clock: launch spans {re.max():.2f} us from the first start to the last end")
    for kd in sorted(set(kind.tolist())):
        q = kind == kd
        print(f"    {names.get(kd, kd):26s} n={q.sum():5d} start us p1 {np.percentile(rs[q], 1):5.2f} p50 {np.median(rs[q]):5.2f} p90 {np.percentile(rs[q], 90):5.2f} p99 {np.percentile(rs[q], 99):5.2f} max {rs[q].max():5.2f}"
              f" | life us p50 {np.median((re - rs)[q]):5.2f} p99 {np.percentile((re - rs)[q], 99):5.2f} | end us p50 {np.median(re[q]):5.2f} p90 {np.percentile(re[q], 90):5.2f} p99 {np.percentile(re[q], 99):5.2f} max {re[q].max():5.2f}")
    h = np.histogram(rs[sw], bins=np.arange(0, re.max() + 1, 1.0))[0]
    print("    step-wave starts per us:", h.tolist())
    h = np.histogram(re[sw], bins=np.arange(0, re.max() + 1, 1.0))[0]
    print("    step-wave ends per us:  ", h.tolist())
    late = sw & (rs > np.percentile(rs[sw], 95))
    print(f"    the last 5 % of step waves to start: start {rs[late].min():.2f}..{rs[late].max():.2f} us, life p50 {np.median((re - rs)[late]):.2f}, end p50 {np.median(re[late]):.2f} max {re[late].max():.2f}; blocks {int(blk[late].min())}..{int(blk[late].max())}")
    life = end - start
    slow = np.argsort(np.where(sw, life, 0))[-12:]
    print("  longest-lived step waves (block, life, mid-start, end-mid):", [(int(blk[i]), int(life[i]), int(mid[i] - start[i]), int(end[i] - mid[i])) for i in slow])
    last = np.argsort(end)[-8:]
    print("  last to end:", [(names.get(int(kind[i]), int(kind[i])), int(blk[i]), int(start[i]), int(end[i])) for i in last])
