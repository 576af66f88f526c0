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
