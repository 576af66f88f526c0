"""Multi-GPU sharding by environment index: one process per GPU, rank r owns the
global envs [offset, offset + count); the step path has NO communication (worlds are independent,
MUW:36-41 builds a private agent_list per env).  The only collective is one gather of per-env
episode metrics to rank 0 (RCCL over xGMI when the backend is "nccl"), serving the evaluation path
that reads env.target_reach_count / env.collision_count before reset (test_sac_multi.py:164-165)."""
import torch
import torch.distributed as dist


def shard_range(total_envs, world_size, rank):
    """Contiguous split of `total_envs` over `world_size` ranks; the first (total % world) ranks get
    one extra env.  Returns (offset, count)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, extra = divmod(int(total_envs), int(world_size))
    count = base + (1 if rank < extra else 0)
    offset = rank * base + min(rank, extra)
    return offset, count


def gather_episode_metrics(local, dst=0, group=None, total_envs=None):
    """Gathers per-env metric rows ([E_local, C] tensor; any dtype) from every rank to `dst` in global env order
    with ONE collective (`gather`; RCCL over xGMI with backend "nccl") and no host synchronisation before it.
    Returns the concatenated [E_total, C] tensor on dst, None elsewhere.

    Shard sizes follow from the layout, not from communication: total_envs=None means every rank owns the same
    number of rows (the weak-scaling layout of bench.py) -- the caller's promise: ranks with different row counts would
    hang the collective; total_envs=T means the contiguous split of shard_range(T, world, rank), whose first T % world
    ranks own one extra row -- rows are then padded to the largest shard (gather_evaluation_summary passes the T that
    make_sharded_env recorded on the env)."""
    if not dist.is_available() or not dist.is_initialized():
        return local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if dist.get_backend(group) == "gloo" and local.is_cuda:
        local = local.cpu()  # rehearsal backend: gloo gathers host tensors
    if total_envs is None:
        sizes = [int(local.shape[0])] * world
    else:
        sizes = [shard_range(total_envs, world, r)[1] for r in range(world)]
        if sizes[rank] != local.shape[0]:
            raise ValueError(f"rank {rank} holds {local.shape[0]} rows, shard_range({total_envs}, {world}, {rank}) says {sizes[rank]}")
    width = max(sizes)
    padded = local
    if local.shape[0] != width:
        padded = torch.zeros((width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        padded[: local.shape[0]] = local
    padded = padded.contiguous()
    bufs = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    dist.gather(padded, gather_list=bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([b[:n] for b, n in zip(bufs, sizes)], dim=0)


def reduce_episode_totals(counters, num_agents, dst=0, group=None):
    """When only SR / CR / mean episode length are wanted (test_sac_multi.py:174-176), four scalars travel instead of
    E rows: ONE `reduce` (ncclReduce, sum) of [sum steps, sum target_reach_count, sum collision_count, envs] to `dst`.
    counters: this rank's [E_local, >=3] rows (steps, reach, coll, ...).  Returns the summary dict on dst, None elsewhere."""
    c = counters[:, :3].to(torch.float64)
    tot = torch.cat([c.sum(dim=0), torch.tensor([float(c.shape[0])], dtype=torch.float64, device=c.device)])
    if dist.is_available() and dist.is_initialized():
        if dist.get_backend(group) == "gloo" and tot.is_cuda:
            tot = tot.cpu()
        dist.reduce(tot, dst=dst, op=dist.ReduceOp.SUM, group=group)
        if dist.get_rank(group) != dst:
            return None
    steps, reach, coll, envs = (float(x) for x in tot)
    return dict(success_rate=reach / (num_agents * envs), collision_rate=coll / (num_agents * envs), mean_steps=steps / envs,
                episodes=int(envs))


def summarize_metrics(counters, num_agents):
    """SR / CR as the reference's evaluation loop computes them (test_sac_multi.py:174-175) from
    gathered [E, 4] counters (steps, target_reach_count, collision_count, episode)."""
    c = counters.to(torch.float64)
    episodes = c.shape[0]
    return dict(success_rate=float(c[:, 1].sum() / (num_agents * episodes)),
                collision_rate=float(c[:, 2].sum() / (num_agents * episodes)),
                mean_steps=float(c[:, 0].mean()))


def make_sharded_env(total_envs, device=None, group=None, seed=0, **env_kwargs):
    """This rank's shard of a `total_envs`-world job as a BatchedMultiUAVWorld2D: global env ids key the
    Philox reset streams (env_offset), so the union of the shards equals the unsharded batch."""
    from .batched import BatchedMultiUAVWorld2D
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    offset, count = shard_range(total_envs, world, rank)
    env = BatchedMultiUAVWorld2D(count, device=device, env_offset=offset, seed=seed, **env_kwargs)
    # the job's size travels with the shard: shard_range gives unequal shards when total_envs % world != 0, and the gather
    # below must then be told (every rank derives every shard's size from it -- no size exchange, one collective)
    env.total_envs = int(total_envs)
    return env


def gather_evaluation_summary(env, dst=0, group=None, total_envs=None):
    """SR / CR / score over the ended episodes of ALL shards (one gather of [E_local, 6] rows).  total_envs defaults to
    what make_sharded_env recorded on `env`; an env built by hand for an UNEQUAL split must pass it (ranks that disagree
    about the row count of a gather hang the collective)."""
    if total_envs is None:
        total_envs = getattr(env, "total_envs", None)
    st = env.episode_stats()
    rows = torch.stack([st["episodes"].float(), st["steps"].float(), st["reach"].float(), st["coll"].float(),
                        st["return0"], st["score"]], dim=1)
    allrows = gather_episode_metrics(rows, dst=dst, group=group, total_envs=total_envs)
    if allrows is None:
        return None
    a = allrows.double().sum(dim=0)
    denom = max(1.0, env.num_agents * float(a[0]))
    return dict(episodes=int(a[0]), success_rate=float(a[2]) / denom, collision_rate=float(a[3]) / denom,
                avg_score=float(a[5]) / denom, mean_steps=float(a[1]) / max(1.0, float(a[0])))


class SplitBatch:
    """The batch of ONE GPU as `parts` independent handles (contiguous env ranges keyed like ranks: env_offset + shard_range),
    each with its own HIP stream -- the layout of a trainer that double-buffers env halves (the policy works on the
    observations of one part while the other part steps).  Independent launch chains overlap each other's launch-to-launch
    boundary (separate streams / separate hipGraphs: +6-12 % at 65 536 x 4 and 8; bench.py `split_batch`), which a single
    stream cannot.  Results per env are those of the undivided batch (Philox streams are keyed by global env id).

        sb = SplitBatch(BatchedMultiUAVWorld2D, 65536, parts=2, num_agents=4)
        obs = sb.reset(seed=0)                                   # list of [E_k, N, 10] tensors, one per part
        for k, env in enumerate(sb.envs):                        # each part advances on its own stream
            with torch.cuda.stream(sb.streams[k]):
                obs[k], rew, done, info = env.step_ex(policy(obs[k]), polar=True, auto_reset="agent0_done")
        sb.synchronize()
    """

    def __init__(self, env_cls, num_envs, parts=2, device=None, env_offset=0, **env_kwargs):
        if parts < 1 or num_envs < parts:
            raise ValueError("SplitBatch: need 1 <= parts <= num_envs")
        self.ranges = [shard_range(num_envs, parts, k) for k in range(parts)]
        self.envs = [env_cls(cnt, device=device, env_offset=env_offset + off, **env_kwargs) for off, cnt in self.ranges]
        self.device = self.envs[0].device
        self.streams = [torch.cuda.Stream(self.device) for _ in range(parts)]
        self.num_envs = int(num_envs)

    def reset(self, **kw):
        """Every part reset on its own stream (after the work already queued on the caller's current stream)."""
        cur = torch.cuda.current_stream(self.device)
        out = []
        for env, st in zip(self.envs, self.streams):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                out.append(env.reset(**kw))
        return out

    def join(self):
        """Makes the caller's current stream wait for every part (e.g. before reading all parts' outputs together)."""
        cur = torch.cuda.current_stream(self.device)
        for st in self.streams:
            cur.wait_stream(st)

    def synchronize(self):
        for st in self.streams:
            st.synchronize()

    def metrics(self):
        """[E, C] per-env counters of the whole batch in global env order."""
        self.join()
        return torch.cat([env.metrics() for env in self.envs], dim=0)

    def close(self):
        self.synchronize()
        for env in self.envs:
            env.close()
