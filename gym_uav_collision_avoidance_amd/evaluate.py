"""Batched evaluation harness: the reference evaluates one episode at a time
(test_sac_multi.py:132-183, test_sac_multi_score.py:31-80); here `episodes` worlds run side by side,
one episode each, with the same stopping rule and the same SR / CR / score formulas."""
import math

import torch

from .batched import BatchedMultiUAVWorld2D


@torch.no_grad()
def evaluate_policy(policy_fn, num_agents, episodes=100, max_steps=2000, evaluate=True, circular=False, seed=0,
                    device=None, polar=True, **env_kwargs):
    """Runs `episodes` parallel episodes of an `num_agents`-UAV world.

    policy_fn(obs [E,N,10] float32 device tensor) -> actions [E,N,2]; with polar=True they are policy
    outputs in [-1,1]^2 converted on the device like test_sac_multi_score.py:47-49, else velocity commands.
    An episode ends when all(dones) (:59) or after max_steps (:13,41); its counters are read at that
    moment (:63-64).  Returns dict(success_rate, collision_rate, avg_score, score0, mean_steps)."""
    env = BatchedMultiUAVWorld2D(episodes, num_agents=num_agents, device=device, seed=seed, **env_kwargs)
    obs = env.reset_circular() if circular else env.reset()
    E, N, dev = episodes, num_agents, env.device
    ended = torch.zeros(E, dtype=torch.bool, device=dev)
    reach = torch.zeros(E, dtype=torch.int64, device=dev)
    coll = torch.zeros(E, dtype=torch.int64, device=dev)
    length = torch.zeros(E, dtype=torch.int64, device=dev)
    score0 = torch.zeros(E, dtype=torch.float64, device=dev)
    total = torch.zeros(E, dtype=torch.float64, device=dev)
    for t in range(max_steps):
        act = policy_fn(obs)
        obs, rew, done, _ = env.step_ex(act, evaluate=evaluate, polar=polar, track_returns=False)
        live = ~ended
        score0 += torch.where(live, rew[:, 0].double(), 0.0)                                       # test_sac_multi.py:155
        total += torch.where(live, (rew.double() * (1.0 - done.double())).sum(dim=1), 0.0)         # :157
        newly = live & (done.all(dim=1) | (t + 1 >= max_steps))                                    # :161 / :140
        m = env.metrics()  # snapshot at the step an env ends (ended envs keep evolving and are ignored)
        reach = torch.where(newly, m[:, 1].long(), reach)
        coll = torch.where(newly, m[:, 2].long(), coll)
        length = torch.where(newly, torch.full_like(length, t + 1), length)
        ended |= newly
        if t % 32 == 31 and bool(ended.all()):
            break
    denom = N * E
    out = dict(num_agents=N, episodes=E,
               success_rate=float(reach.sum()) / denom,          # test_sac_multi.py:174
               collision_rate=float(coll.sum()) / denom,         # :175
               avg_score=float(total.sum()) / denom,             # :176
               score0=float(score0.mean()), mean_steps=float(length.double().mean()))
    env.close()
    return out


@torch.no_grad()
def rollout_trajectories(policy_fn, num_agents, episodes=1, max_steps=2000, circular=True, layout=None, polar=True,
                         evaluate=False, seed=0, device=None, **env_kwargs):
    """The plotting scenario of test_sac_multi_plot_trajectory.py:43-76 for `episodes` worlds side by side: reset, place
    the UAVs (circular=True: MUW:157-163's circle; layout=(loc [N,2], tgt [N,2]) float64: the script's own pokes,
    :46-47, e.g. with its OFFSET), then step until all(dones) (:75) or max_steps, recording every agent's location
    BEFORE each step (:66) and feeding agents whose `done` flag is set a zero command (:57-59).

    policy_fn(obs [E,N,10]) -> actions [E,N,2] (policy outputs in [-1,1]^2 with polar=True, :61-63; velocity commands
    otherwise).  Returns dict(positions [T,E,N,2] (float64 when the episode runs in float64-position mode, as the
    reference's does after such pokes), valid [T,E,N] bool (False once the agent is done, or after its world's episode
    is over: the script stops appending), depots [E,N,2], goals [E,N,2], length [E] steps until all(dones) / the cap)."""
    env = BatchedMultiUAVWorld2D(episodes, num_agents=num_agents, device=device, seed=seed, **env_kwargs)
    E, N, dev = episodes, num_agents, env.device
    obs = env.reset()
    if layout is not None:
        loc = torch.as_tensor(layout[0], dtype=torch.float64).reshape(N, 2)
        tgt = torch.as_tensor(layout[1], dtype=torch.float64).reshape(N, 2)
        # what the script's assignments leave behind: init / prev distances keep reset()'s values (it does not touch them)
        env.set_state_f64(loc=loc.expand(E, N, 2), tgt=tgt.expand(E, N, 2))
        obs = env.observe()
    elif circular:
        obs = env.reset_circular()
    wide = env.position_mode == "float64"

    def locations():
        return env.get_state_f64()["loc"] if wide else env.get_state()["loc"]

    depots = locations().clone()
    goals = (env.get_state_f64()["tgt"] if wide else env.get_state()["tgt"]).clone()
    ended = torch.zeros(E, dtype=torch.bool, device=dev)
    length = torch.zeros(E, dtype=torch.int64, device=dev)
    pos, valid = [], []
    for t in range(max_steps):
        done_flag = (env.get_state()["flags"] & 1) != 0                    # env.agent_list[i].done, :57
        act = policy_fn(obs)
        act = (act if act.dtype in (torch.float32, torch.float64) else act.to(torch.float32)).clone()
        if polar:
            act[..., 0] = torch.where(done_flag, torch.full_like(act[..., 0], -1.0), act[..., 0])   # v = 0 -> command (0, 0)
        else:
            act = torch.where(done_flag[..., None], torch.zeros_like(act), act)
        pos.append(locations().clone())                                     # :66 (before the step)
        valid.append(~done_flag & ~ended[:, None])
        obs, rew, done, _ = env.step_ex(act, evaluate=evaluate, polar=polar, track_returns=False)
        newly = ~ended & done.all(dim=1)                                    # :75
        length = torch.where(newly, torch.full_like(length, t + 1), length)
        ended |= newly
        if t % 16 == 15 and bool(ended.all()):
            break
    length = torch.where(ended, length, torch.full_like(length, len(pos)))
    out = dict(positions=torch.stack(pos), valid=torch.stack(valid), depots=depots, goals=goals, length=length)
    env.close()
    return out


def sweep_num_agents(policy_fn, agent_counts=range(1, 25), episodes=100, max_steps=2000, **kw):
    """SR / CR versus the number of UAVs like test_sac_multi_score.py:31-80."""
    return [evaluate_policy(policy_fn, n, episodes=episodes, max_steps=max_steps, **kw) for n in agent_counts]


def seek_policy(env_diag=math.hypot(50.0, 50.0), brake=2.0, vcap=8.0, vmax_norm=math.sqrt(200.0)):
    """A hand-written goal-seeking controller expressed on the OBSERVATION (so it can drive either the
    reference or this build): fly along the target bearing, brake with the distance.  Returns policy
    outputs in [-1,1]^2 for the polar action convention.  Used by tests and examples."""
    def fn(obs):
        dist = obs[..., 2] * env_diag
        heading = obs[..., 1] + obs[..., 3]          # (theta_v + wrap(theta_t - theta_v)) / pi
        speed = torch.where(dist > 0.3, torch.clamp(torch.sqrt(2 * brake * dist), max=vcap), torch.zeros_like(dist))
        a0 = (speed / vmax_norm) * 2 - 1
        a1 = torch.remainder(heading + 1.0, 2.0) - 1.0
        return torch.stack([a0, a1], dim=-1)
    return fn
