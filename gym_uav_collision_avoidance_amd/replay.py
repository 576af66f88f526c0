"""On-device replay memory fed zero-copy by the step kernel (BASELINE config 5).

The reference pushes (state, action, reward, next_state, float(not done)) per agent into a host
python list (pytorch_sac_temp/replay_memory.py:15-24, test_sac_multi.py:101-103) and samples
uniformly (:21-24).  Here the env's own output buffers ARE the memory: a time-major ring in HBM whose
slot k holds what step k produced, and the step launch writes observation / reward / done of step k
straight into slots k+1 / k / k — nothing is copied and s'(k) is s(k+1) by construction, so every
observation is stored once.  Rows where the env was re-initialised instead of stepped (auto-reset,
`reset_mask`) are not transitions and are never returned by sample().
"""
import torch

from . import _lib


class DeviceReplay:
    def __init__(self, env, horizon, num_learners=None, packed_flags=False):
        """env: BatchedMultiUAVWorld2D; horizon: number of most recent steps kept (capacity in
        transitions = horizon * num_envs * num_learners).  num_learners: only agents [0, num_learners)
        are sampled; default all (scripted bodies created with num_bodies=... are not agents and never appear here).
        packed_flags=True: the launches run with UAVX_FLAGS_IN_DONE -- "re-initialised", "ended" and "truncated" ride in
        bits 1..3 of the done byte of every env's agent 0 inside the ring's done rows (step() then returns the raw uint8
        done row), and the skip / ended / trunc rows below are not written by the launches at all."""
        self.env, self.T = env, int(horizon)
        self.num_learners = env.num_agents if num_learners is None else int(num_learners)
        self.packed = bool(packed_flags)
        L, E, N, dev = self.T + 1, env.num_envs, env.num_agents, env.device
        if (E * N * _lib.OBS_DIM * 4) % 16:
            # the step launch writes every observation slot with 16-byte stores: each slot of the ring must start on a
            # 16-byte boundary (an odd number of agent slots would put every other one 8 bytes off)
            raise ValueError(f"uavx: DeviceReplay needs num_envs * num_agents even (got {E} x {N}): observation slots of "
                             "the ring are written with 16-byte stores")
        self.L = L
        self.obs = torch.zeros((L, E, N, _lib.OBS_DIM), dtype=torch.float32, device=dev)
        self.act = torch.zeros((L, E, N, 2), dtype=torch.float32, device=dev)
        self.rew = torch.zeros((L, E, N), dtype=torch.float32, device=dev)
        self.done = torch.zeros((L, E, N), dtype=torch.uint8, device=dev)
        self.skip = torch.zeros((L, E), dtype=torch.uint8, device=dev)    # row is not a transition (env was re-initialised)
        self.trunc = torch.zeros((L, E), dtype=torch.uint8, device=dev)   # episode cut by the step cap AT this transition
        self.ended = torch.zeros((L, E), dtype=torch.uint8, device=dev)   # episode ended AT this transition
        self.count = 0  # steps written so far

    def __len__(self):
        return min(self.count, self.T) * self.env.num_envs * self.num_learners

    def begin(self, obs0):
        """Stores the observation the first step starts from (what env.reset() returned)."""
        self.obs[self.count % self.L].copy_(obs0)

    @property
    def state(self):
        """Current observation s(k) — feed this to the policy."""
        return self.obs[self.count % self.L]

    def action_slot(self):
        """[E, N, 2] float32 view the policy can write its output into (no copy at step time)."""
        return self.act[self.count % self.L]

    def step(self, actions=None, **step_ex_kwargs):
        """Steps the env with `actions` (or whatever was written into action_slot()); outputs land in the ring."""
        k = self.count % self.L
        if actions is not None:
            self.act[k].copy_(actions)
        nxt = (self.count + 1) % self.L
        if self.packed:
            obs, rew, done, info = self.env.step_ex(self.act[k], out=(self.obs[nxt], self.rew[k], self.done[k]),
                                                    packed_flags=True, **step_ex_kwargs)
        else:
            obs, rew, done, info = self.env.step_ex(self.act[k], out=(self.obs[nxt], self.rew[k], self.done[k]),
                                                    flags_out=(self.skip[k], self.ended[k], self.trunc[k]), **step_ex_kwargs)
        self.count += 1
        return obs, rew, done, info

    def _flags(self, s, e):
        """(skip, truncated, ended) bool tensors of ring rows s, envs e."""
        if self.packed:
            b = self.done[s, e, 0]
            return (b & 2) != 0, (b & 8) != 0, (b & 4) != 0
        return self.skip[s, e] != 0, self.trunc[s, e] != 0, self.ended[s, e] != 0

    def reset(self, mask=None, **reset_kwargs):
        """A manual (masked) env.reset() between two steps: the fresh observations replace s(k) in the ring, so the
        transition that led there no longer has its successor and is taken out of sampling."""
        obs = self.env.reset(mask=mask, out=self.obs[self.count % self.L], **reset_kwargs)
        if self.count > 0:
            prev = (self.count - 1) % self.L
            m = None if mask is None else torch.as_tensor(mask, device=self.env.device).to(torch.uint8)
            if self.packed:
                self.done[prev, :, 0] |= 2 if m is None else (m != 0).to(torch.uint8) * 2
            elif m is None:
                self.skip[prev].fill_(1)
            else:
                self.skip[prev] |= m
        return obs

    def sample(self, batch_size, generator=None, with_flags=False):
        """Uniform batch of transitions like ReplayMemory.sample (replay_memory.py:21-24):
        (state [B,10], action [B,2], reward [B], next_state [B,10], mask [B] = 1 - done); with_flags=True appends
        (truncated [B] bool, ended [B] bool) of the env at that transition.  Fixed size, no host sync.
        Rows where the env was re-initialised instead of stepped are not transitions: a drawn row that hits one is
        redrawn once, and what is still invalid after that is replaced by a valid row of the same batch.  Only when NO row
        of the batch is valid after the redraw (a tiny batch right after a reset of every env) does a re-initialisation row
        come back -- recognisable by reward 0 and mask 1 with `ended` False; draw a larger batch or step first."""
        assert self.count > 0
        E, N, dev = self.env.num_envs, self.num_learners, self.env.device
        lo = max(0, self.count - self.T)
        span = self.count - lo

        def draw():
            r = torch.rand((3, batch_size), generator=generator, device=dev)
            k = lo + (r[0] * span).long().clamp_(max=span - 1)   # k in [lo, count - 1]: only written slots
            return k, (r[1] * E).long().clamp_(max=E - 1), (r[2] * N).long().clamp_(max=N - 1)

        k, e, i = draw()
        k2, e2, i2 = draw()  # one redraw for rows that hit a reset step (rare: one per episode per env)
        bad = self._flags(k % self.L, e)[0]
        k, e, i = torch.where(bad, k2, k), torch.where(bad, e2, e), torch.where(bad, i2, i)
        valid = ~self._flags(k % self.L, e)[0]
        # still invalid (both draws hit reset rows): duplicate the nearest valid row of this batch -- never a row outside
        # [lo, count - 1] and never a reset row (a neighbouring STEP is not safe: with step_cap=1, or after a manual
        # reset, reset rows can be adjacent, and count itself is not written yet)
        pos = torch.arange(batch_size, device=dev)
        before = torch.cummax(torch.where(valid, pos, torch.full_like(pos, -1)), dim=0).values
        after = torch.flip(torch.cummin(torch.flip(torch.where(valid, pos, torch.full_like(pos, batch_size)), [0]), dim=0).values, [0])
        src = torch.where(before >= 0, before, after.clamp(max=batch_size - 1))
        k, e, i = k[src], e[src], i[src]
        s, s1 = k % self.L, (k + 1) % self.L
        out = (self.obs[s, e, i], self.act[s, e, i], self.rew[s, e, i], self.obs[s1, e, i],
               1.0 - (self.done[s, e, i] & 1).float())
        if with_flags:
            _, tr, en = self._flags(s, e)
            out = out + (tr, en)
        return out
