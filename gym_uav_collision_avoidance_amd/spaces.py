"""Minimal Box space with the gym 0.24 surface the reference's callers use (shape, low, high,
dtype, sample(): run_multi.py:13, test_sac_multi.py:39-40,77).  gym itself is not a dependency."""
import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        self.shape = tuple(shape) if shape is not None else tuple(np.shape(low))
        self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()

    @classmethod
    def batched(cls, space, lead):
        """gym.vector.utils.batch_space for a Box: `lead` extra leading dims.  low/high are read-only broadcast
        views of the single space's bounds (a 1 M-env batch would otherwise carry hundreds of MB of bounds)."""
        b = cls.__new__(cls)
        b.dtype = space.dtype
        b.shape = tuple(lead) + space.shape
        b.low = np.broadcast_to(space.low, b.shape)
        b.high = np.broadcast_to(space.high, b.shape)
        return b

    def sample(self):
        return np.random.uniform(self.low, self.high, size=self.shape).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
        return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"
