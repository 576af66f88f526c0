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


__all__ = ["BatchedMultiUAVWorld2D", "BatchedUAVWorld2D", "HipArray", "Box", "UAVVectorEnv", "UAVSingleVectorEnv",
           "make", "ENV_IDS"]
