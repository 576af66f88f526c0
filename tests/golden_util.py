"""Helpers shared by the oracle (CPU) and HIP (GPU) parity tests: fixture loading + comparisons."""
import glob
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# obs columns that are angles/pi and therefore live on a circle of circumference 2
ANGLE_COLS = (1, 3, 5, 6, 8, 9)
UW_ANGLE_COLS = (1, 3)


def fixture_names(kind):
    out = []
    for p in sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))):
        with np.load(p) as z:
            meta = json.loads(str(z["meta"]))
        if meta["kind"] == kind:
            out.append(os.path.basename(p)[:-4])
    return out


def load_fixture(name):
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz")) as z:
        data = {k: z[k] for k in z.files}
    meta = json.loads(str(data.pop("meta")))
    return data, meta


def circ_diff(a, b):
    """|a-b| measured on the circle of circumference 2 (angles are normalised by pi)."""
    d = np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))
    d = np.mod(d, 2.0)
    return np.minimum(d, 2.0 - d)


def obs_err(got, ref, angle_cols=ANGLE_COLS):
    """max abs error over an obs array [..., D]; angle columns on the circle."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    err = np.abs(got - ref)
    for c in angle_cols:
        err[..., c] = circ_diff(got[..., c], ref[..., c])
    return float(err.max()) if err.size else 0.0


def tie_agents(loc, d_sense, f64pos):
    """Agents whose in-range neighbour distances contain an exact tie.  The order numpy's argsort
    gives tied elements is platform dependent (AVX-512 builds use an unstable SIMD sort; the numpy
    the reference targeted used a stable insertion sort), so neighbour columns 4..9 of such agents
    are compared modulo that order.  The build resolves ties lower-index-first."""
    loc = np.asarray(loc, dtype=np.float64 if f64pos else np.float32)
    n = loc.shape[0]
    out = np.zeros(n, dtype=bool)
    for i in range(n):
        ds = []
        for j in range(n):
            if j == i:
                continue
            diff = loc[j] - loc[i]
            d = np.sqrt(diff[0] * diff[0] + diff[1] * diff[1])
            if d < (d_sense if f64pos else np.float32(d_sense)):
                ds.append(float(d))
        out[i] = len(set(ds)) != len(ds)
    return out


# ---------------------------------------------------------------------------------------------------------------------------
# Independent arithmetic for the scripted bodies of the configs[4] extension (include/uavx.h, uavx_set_body_rule): numpy only,
# written from the published Philox4x32-10 definition (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11;
# Random123) and from the header's text -- NOT from oracle/uavx_oracle.c and not from the HIP source -- so that a mistake the
# oracle and the kernel share (both restate make_leg / atan2_exact) cannot pass unnoticed.
def philox4x32_10(counter, key):
    """counter: 4 uint32, key: 2 uint32 -> 4 uint32 (Random123 philox4x32-10)."""
    c = [int(x) & 0xFFFFFFFF for x in counter]
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c[0], 0xCD9E8D57 * c[2]
        c = [(p1 >> 32) ^ c[1] ^ k0, p1 & 0xFFFFFFFF, (p0 >> 32) ^ c[3] ^ k1, p0 & 0xFFFFFFFF]
        k0, k1 = (k0 + 0x9E3779B9) & 0xFFFFFFFF, (k1 + 0xBB67AE85) & 0xFFFFFFFF
    return c


def body_waypoint(global_env, slot, leg, episode, seed, x_size, y_size):
    """Waypoint `leg` of the body in neighbour slot `slot`: words 0, 1 of Philox(counter = (env[31:0], env[47:32] | slot << 16,
    0x80000000 | leg, episode), key = seed), uniform over the box, cast to float32 (np.random.uniform(lo, hi).astype(float32))."""
    w = philox4x32_10([global_env & 0xFFFFFFFF, ((global_env >> 32) & 0xFFFF) | (slot << 16), 0x80000000 | leg, episode],
                      [seed & 0xFFFFFFFF, seed >> 32])
    lo_x, lo_y = -x_size / 2.0, -y_size / 2.0
    return (np.float32(lo_x + x_size * (w[0] / 4294967296.0)), np.float32(lo_y + y_size * (w[1] / 4294967296.0)))


def body_leg(px, py, wx, wy, speed, tau):
    """{dx, dy, legs} of a leg from P to W in float32 without FMA, and its heading in float64 (math.atan2: what the record's
    float32 heading must agree with to float32 accuracy; its bit pattern is the build's own polynomial)."""
    import math
    f = np.float32
    step = f(speed * tau)
    dx, dy = f(f(wx) - f(px)), f(f(wy) - f(py))
    d = f(np.sqrt(f(f(dx * dx) + f(dy * dy))))
    if not d > 0:
        return f(0), f(0), f(0), 0.0
    sc = f(step / d)
    return f(dx * sc), f(dy * sc), f(np.floor(f(d / step))), math.atan2(float(dy), float(dx))


def body_track(start, global_env, slot, episode, seed, x_size, y_size, speed, tau, period, steps):
    """Positions after env steps 1..steps of one body that starts at `start` (float32 pair) with leg 0, by the header's rule:
    a new leg at every step s > 0 with s % period == 0; in step s it moves by (dx, dy) iff s % period < legs."""
    f = np.float32
    x, y = f(start[0]), f(start[1])
    out, legs_seen = [], []
    dx = dy = legs = None
    for s in range(steps):
        if s % period == 0:
            wx, wy = body_waypoint(global_env, slot, s // period, episode, seed, x_size, y_size)
            dx, dy, legs, heading = body_leg(x, y, wx, wy, speed, tau)
            legs_seen.append((s, dx, dy, legs, heading, wx, wy))
        if (s % period) < legs:
            x, y = f(x + dx), f(y + dy)
        out.append((x, y))
    return np.array(out, dtype=np.float32), legs_seen
