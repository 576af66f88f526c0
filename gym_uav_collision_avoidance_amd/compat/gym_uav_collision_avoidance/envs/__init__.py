"""OPT-IN alias of gym_uav_collision_avoidance/envs/__init__.py:1-2: the reference's two class names, the MI355X façades."""
from gym_uav_collision_avoidance_amd.envs import MultiUAVWorld2D, UAVWorld2D

__all__ = ["UAVWorld2D", "MultiUAVWorld2D"]
