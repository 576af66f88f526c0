"""OPT-IN alias: `import gym_uav_collision_avoidance` resolves to the MI355X build (reference: gym_uav_collision_avoidance/
__init__.py:1-10).  A thin re-export, not on sys.path by default -- see gym_uav_collision_avoidance_amd.install_alias()."""
from gym_uav_collision_avoidance_amd import ENV_IDS

try:   # the reference registers its two ids with gym at import; done here when a gym is importable
    from gym.envs.registration import register
    for _id, _entry in ENV_IDS.items():
        register(id=_id, entry_point=_entry.replace("gym_uav_collision_avoidance_amd.envs", "gym_uav_collision_avoidance.envs"))
except ImportError:
    pass
