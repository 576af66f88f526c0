"""gym.vector.VectorEnv-shaped view of the batched worlds (gym itself is not a dependency).

The reference never vectorises its envs (every script builds ONE MultiUAVWorld2D, run_multi.py:6,
test_sac_multi.py:35); a caller that wants E of them would wrap E copies in gym.vector.SyncVectorEnv.
This class has that wrapper's call surface -- num_envs, single_/batched spaces, reset, step_async /
step_wait / step, seed, close, get_attr / set_attr -- on top of ONE launch per step:

    venv = UAVVectorEnv(65536, num_agents=4)
    obs = venv.reset(seed=0)                               # [E, N, 10] device tensor
    obs, rew, done, info = venv.step(actions)              # actions [E, N, 2]; rew / done [E, N]

Differences from gym's SyncVectorEnv, both forced by the reference env being multi-agent (its done is a
per-agent list, MUW:233) and by the batch size:
  * an env is re-initialised when its episode ends under `auto_reset` ("agent0_done" is the trainers' rule,
    test_sac_multi.py:112; "all_done" the evaluation rule, :161; None never) or at `step_cap` (:17,67), in the
    step AFTER the one that returned the terminal transition (EnvPool order, not gym's same-step order): the
    terminal observation is therefore simply the `obs` of that step, and the re-initialised row comes back
    with reward 0, done False and info["reset_mask"][e] True; the step that ENDS an episode carries
    info["ended"][e] and, when the step cap alone ended it, info["truncated"][e] (a time-limit cut, not a terminal state);
  * `info` is one dict of [E] device tensors, not a tuple of E dicts.
"""
import numpy as np

from .batched import BatchedMultiUAVWorld2D, BatchedUAVWorld2D
from .spaces import Box


def _batch_box(space, lead):
    return Box.batched(space, lead)


class UAVVectorEnv:
    is_vector_env = True
    metadata = {"render_modes": []}
    reward_range = (-float("inf"), float("inf"))

    def __init__(self, num_envs, auto_reset="agent0_done", step_cap=0, polar=False, evaluate=False, device=None,
                 seed=0, env_offset=0, **world_kwargs):
        self.env = BatchedMultiUAVWorld2D(num_envs, device=device, seed=seed, env_offset=env_offset, **world_kwargs)
        self.num_envs, self.num_agents = self.env.num_envs, self.env.num_agents
        self.auto_reset, self.step_cap, self.polar, self.evaluate = auto_reset, int(step_cap), bool(polar), bool(evaluate)
        self.device = self.env.device
        self.closed = False
        self._pending = None
        self._refresh_spaces()

    def _refresh_spaces(self):
        e, n = self.env, self.num_agents
        act = Box(-1.0, 1.0, shape=(2,), dtype=np.float32) if self.polar else e.action_space
        self.single_observation_space = _batch_box(e.observation_space, (n,))      # one world: all its agents
        self.single_action_space = _batch_box(act, (n,))
        self.observation_space = _batch_box(e.observation_space, (self.num_envs, n))
        self.action_space = _batch_box(act, (self.num_envs, n))

    # -- gym.vector.VectorEnv surface ----------------------------------------------------------------
    def seed(self, seed=None):
        if seed is not None:
            self.env.seed = int(seed)
        return [self.env.seed + self.env.env_offset + i for i in range(min(self.num_envs, 8))]  # informational

    def reset(self, seed=None, return_info=False, options=None, mask=None):
        obs = self.env.reset(mask=mask, seed=seed)
        self._pending = None
        return (obs, {"distance": 0}) if return_info else obs

    def reset_async(self, seed=None, return_info=False, options=None):
        self._reset_args = (seed, return_info)

    def reset_wait(self, timeout=None, **kwargs):
        seed, return_info = self.__dict__.pop("_reset_args", (None, False))
        return self.reset(seed=seed, return_info=return_info)

    def step_async(self, actions):
        if self._pending is not None:
            raise RuntimeError("step_async called twice without step_wait")
        # the launch is asynchronous on the caller's stream: enqueue it now, hand the views out in step_wait
        self._pending = self.env.step_ex(actions, evaluate=self.evaluate, polar=self.polar, auto_reset=self.auto_reset,
                                         step_cap=self.step_cap, track_returns=True)

    def step_wait(self, timeout=None):
        if self._pending is None:
            raise RuntimeError("step_wait called without step_async")
        out, self._pending = self._pending, None
        return out

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def get_attr(self, name):
        return (getattr(self.env, name),) * self.num_envs   # the worlds of one batch share their attributes

    def set_attr(self, name, values):
        """World scalars (x_size, y_size, max_speed, max_acceleration, collider_radius, d_sense) are shared by all
        envs of the batch: `values` is one value (or a sequence of equal values)."""
        if isinstance(values, (list, tuple)):
            if any(v != values[0] for v in values):
                raise ValueError("uavx: the worlds of one batch share their parameters")
            values = values[0]
        self.env.set_config(**{name: values})
        self._refresh_spaces()

    def episode_stats(self):
        return self.env.episode_stats()

    def evaluation_summary(self):
        return self.env.evaluation_summary()

    def render(self, *a, **k):
        return None

    def close(self, **kwargs):
        if not self.closed:
            self.env.close()
            self.closed = True

    def __len__(self):
        return self.num_envs

    def __repr__(self):
        return f"UAVVectorEnv({self.num_envs} x MultiUAVWorld2D[{self.num_agents} UAVs], auto_reset={self.auto_reset!r})"


class UAVSingleVectorEnv:
    """The same surface over E x UAVWorld2D (UW:14): obs [E, 4], actions [E, 2], reward / done [E];
    auto-reset when an env's episode ends (UW:159-169), next-step order as above."""
    is_vector_env = True
    metadata = {"render_modes": []}

    def __init__(self, num_envs, auto_reset=True, step_cap=0, polar=False, device=None, seed=0, env_offset=0,
                 **world_kwargs):
        self.env = BatchedUAVWorld2D(num_envs, device=device, seed=seed, env_offset=env_offset, **world_kwargs)
        self.num_envs = self.env.num_envs
        self.auto_reset, self.step_cap, self.polar = bool(auto_reset), int(step_cap), bool(polar)
        self.device = self.env.device
        self.closed = False
        self._pending = None
        act = Box(-1.0, 1.0, shape=(2,), dtype=np.float32) if self.polar else self.env.action_space
        self.single_observation_space, self.single_action_space = self.env.observation_space, act
        self.observation_space = _batch_box(self.env.observation_space, (self.num_envs,))
        self.action_space = _batch_box(act, (self.num_envs,))

    def seed(self, seed=None):
        if seed is not None:
            self.env.seed = int(seed)

    def reset(self, seed=None, return_info=False, options=None, mask=None):
        obs = self.env.reset(mask=mask, seed=seed)
        self._pending = None
        return (obs, {}) if return_info else obs

    def step_async(self, actions):
        if self._pending is not None:
            raise RuntimeError("step_async called twice without step_wait")
        self._pending = self.env.step_ex(actions, polar=self.polar, auto_reset=self.auto_reset, step_cap=self.step_cap)

    def step_wait(self, timeout=None):
        if self._pending is None:
            raise RuntimeError("step_wait called without step_async")
        out, self._pending = self._pending, None
        return out

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def episode_stats(self):
        return self.env.episode_stats()

    def close(self, **kwargs):
        if not self.closed:
            self.env.close()
            self.closed = True

    def __len__(self):
        return self.num_envs


__all__ = ["UAVVectorEnv", "UAVSingleVectorEnv"]
