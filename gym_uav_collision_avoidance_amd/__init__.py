"""uavx — MI355X-native batched implementation of the gym_uav_collision_avoidance step/reset path.

    from gym_uav_collision_avoidance_amd import BatchedMultiUAVWorld2D      # E worlds per launch
    from gym_uav_collision_avoidance_amd.envs import MultiUAVWorld2D        # drop-in single env

The env arithmetic runs in hand-written HIP kernels (csrc/) behind the C ABI of include/uavx.h;
there is no CPU path in this package.
"""
from . import _lib
from .batched import BatchedMultiUAVWorld2D, BatchedUAVWorld2D, HipArray
from .spaces import Box
from .vector import UAVSingleVectorEnv, UAVVectorEnv

ENV_IDS = {  # gym registration ids of the reference (gym_uav_collision_avoidance/__init__.py:3-10)
    "gym_uav_collision_avoidance/UAVWorld2D-v0": "gym_uav_collision_avoidance_amd.envs:UAVWorld2D",
    "gym_uav_collision_avoidance/MultiUAVWorld2D-v0": "gym_uav_collision_avoidance_amd.envs:MultiUAVWorld2D",
}


def make(env_id, **kwargs):
    """gym.make-style constructor for the two reference ids (gym itself is not required)."""
    import importlib
    mod, cls = ENV_IDS[env_id].split(":")
    return getattr(importlib.import_module(mod), cls)(**kwargs)


def install_alias(force=False):
    """Makes the reference's import lines resolve to this build -- `from gym_uav_collision_avoidance.envs import MultiUAVWorld2D`
    (run_multi.py:2, run.py:2, the trainers) -- by putting the thin re-export package `compat/gym_uav_collision_avoidance` in
    front of sys.path.  Opt-in: call this (or start Python with UAVX_ALIAS=1, or put `<package>/compat` on PYTHONPATH); never
    done silently, so a real checkout of the reference is never shadowed by accident -- if one is importable this raises unless
    force=True.  Returns the directory that was added."""
    import importlib.util
    import os
    import sys
    compat = os.path.join(os.path.dirname(os.path.abspath(__file__)), "compat")
    spec = importlib.util.find_spec("gym_uav_collision_avoidance")
    if spec is not None and spec.origin and not os.path.abspath(spec.origin).startswith(compat):
        if not force:
            raise ImportError(f"a gym_uav_collision_avoidance package is already importable ({spec.origin}); "
                              "install_alias(force=True) puts the MI355X build in front of it")
        for name in [m for m in sys.modules if m == "gym_uav_collision_avoidance" or m.startswith("gym_uav_collision_avoidance.")]:
            del sys.modules[name]
    if compat not in sys.path:
        sys.path.insert(0, compat)
    importlib.invalidate_caches()
    return compat


import os as _os
if _os.environ.get("UAVX_ALIAS") == "1":
    install_alias(force=True)

__all__ = ["BatchedMultiUAVWorld2D", "BatchedUAVWorld2D", "HipArray", "Box", "UAVVectorEnv", "UAVSingleVectorEnv",
           "make", "ENV_IDS", "install_alias"]
