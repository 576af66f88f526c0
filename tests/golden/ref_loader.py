"""Loader for the *reference* env classes (generation-time only; never runs on the GPU box).

The reference's env arithmetic needs only numpy + math, but its modules import gym / pygame /
cv2 / turtle at top level (MUW:3-5, AG:2-3, UW:3-7), none of which are installed here.  This
loader registers inert placeholder modules under those names, then executes the three reference
files from /root/reference by path.  It is used ONLY by tests/golden/make_golden.py to produce
the committed .npz fixtures and by the optional local cross-check in tests (skipped when
/root/reference is absent).  Nothing from the reference is copied into this repository.
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF_ROOT = os.environ.get("UAVX_REFERENCE_ROOT", "/root/reference")


class _Box:
    """Minimal stand-in for gym.spaces.Box (only what the reference env constructors touch)."""

    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.shape = tuple(shape) if shape is not None else np.shape(low)
        self.dtype = np.dtype(dtype)
        self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()

    def sample(self):
        return np.random.uniform(self.low, self.high, size=self.shape).astype(self.dtype)


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "gym_uav_collision_avoidance", "envs"))


def load():
    """Returns (MultiUAVWorld2D, UAVWorld2D, UAVAgent) classes of the reference."""
    if "gym" not in sys.modules:
        gym = types.ModuleType("gym")
        gym.Env = object
        spaces = types.ModuleType("gym.spaces")
        spaces.Box = _Box
        gym.spaces = spaces
        sys.modules["gym"] = gym
        sys.modules["gym.spaces"] = spaces
    for name, attrs in (("pygame", ()), ("turtle", ("position",)), ("cv2", ("normalize", "resizeWindow"))):
        if name not in sys.modules:
            m = types.ModuleType(name)
            for a in attrs:
                setattr(m, a, None)
            sys.modules[name] = m
    pkg = "gym_uav_collision_avoidance"
    if pkg not in sys.modules:
        sys.modules[pkg] = types.ModuleType(pkg)
        sys.modules[pkg + ".envs"] = types.ModuleType(pkg + ".envs")
    mods = {}
    for short, fname in (("uav_agent", "uav_agent.py"), ("multi_uav_world_2d", "multi_uav_world_2d.py"),
                         ("uav_world_2d", "uav_world_2d.py")):
        full = f"{pkg}.envs.{short}"
        if full not in sys.modules:
            spec = importlib.util.spec_from_file_location(
                full, os.path.join(REF_ROOT, pkg, "envs", fname))
            mod = importlib.util.module_from_spec(spec)
            sys.modules[full] = mod
            spec.loader.exec_module(mod)
        mods[short] = sys.modules[full]
    return (mods["multi_uav_world_2d"].MultiUAVWorld2D, mods["uav_world_2d"].UAVWorld2D,
            mods["uav_agent"].UAVAgent)
