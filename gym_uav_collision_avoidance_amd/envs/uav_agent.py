"""Host-side view of one agent of a device-resident world.

The reference's callers read and poke `env.agent_list[i].<field>` directly
(test_sac_multi_plot_trajectory.py:43-49,57,66; test_ddpg_multi.py:122-126).  Each attribute of
this view maps to one UAVAgent field (AG:13-20) and moves data with uavx_get_state/uavx_set_state.
"""
import numpy as np

from .. import _lib


class UAVAgentView:
    def __init__(self, world, index, color, max_speed, max_acceleration, tau):
        self._w, self._i = world, index
        self.color = color                                                  # AG:9
        self.max_speed = np.array([max_speed, max_speed])                   # AG:10
        self.max_acceleration = np.array([max_acceleration, max_acceleration])  # AG:11
        self.tau = tau                                                      # AG:12

    _WIDE = {"loc": "loc", "tgt": "tgt", "init_d": "init_d", "prev_d": "prev_d"}   # fields with a float64 twin

    def _get(self, name):
        b = self._w._batched
        if name in self._WIDE and b.position_mode == "float64":   # the reference would hand back its float64 array
            return b.get_state_f64()[name][0, self._i].cpu().numpy()
        return b.get_state()[name][0, self._i].cpu().numpy()

    def _set(self, name, value):
        b = self._w._batched
        arr = np.asarray(value)
        # Assigning a float64 array (or python floats) to a position field makes the reference compute that episode
        # in float64 (test_sac_multi_plot_trajectory.py:43-49); the device mode is per world, so the whole env
        # switches (the scripts that do this assign every agent).
        if name in self._WIDE and (b.position_mode == "float64" or (name in ("loc", "tgt") and arr.dtype == np.float64)):
            b.set_position_mode("float64")
            st = b.get_state_f64()[name]
            st[0, self._i] = st.new_tensor(np.asarray(arr, dtype=np.float64))
            b.set_state_f64(**{name: st})
            return
        st = b.get_state()[name]
        st[0, self._i] = st.new_tensor(np.asarray(arr, dtype=np.float64))
        b.set_state(**{name: st})

    location = property(lambda s: s._get("loc"), lambda s, v: s._set("loc", v))                 # AG:13
    velocity = property(lambda s: s._get("vel"), lambda s, v: s._set("vel", v))                 # AG:14
    velocity_prev = velocity                                                                    # AG:15
    target_location = property(lambda s: s._get("tgt"), lambda s, v: s._set("tgt", v))          # AG:16
    init_distance = property(lambda s: float(s._get("init_d")), lambda s, v: s._set("init_d", v))  # AG:17
    prev_distance = property(lambda s: float(s._get("prev_d")), lambda s, v: s._set("prev_d", v))  # AG:18

    def _flag(self, bit):
        return bool(int(self._get("flags")) & bit)

    def _set_flag(self, bit, on):
        f = int(self._get("flags"))
        self._set("flags", (f | bit) if on else (f & ~bit))

    done = property(lambda s: s._flag(_lib.FLAG_DONE), lambda s, v: s._set_flag(_lib.FLAG_DONE, v))              # AG:19
    collided = property(lambda s: s._flag(_lib.FLAG_COLLIDED), lambda s, v: s._set_flag(_lib.FLAG_COLLIDED, v))  # AG:20
