from .uav_world_2d import UAVWorld2D
from .multi_uav_world_2d import MultiUAVWorld2D
from .uav_agent import UAVAgentView

__all__ = ["UAVWorld2D", "MultiUAVWorld2D", "UAVAgentView"]
