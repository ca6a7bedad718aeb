"""Sharding of the env axis over ranks (one process per GPU, torch.distributed; backend 'nccl' = RCCL over xGMI
on MI355X, 'gloo' in CPU tests).  Envs never interact, so the step path has NO collective: rank r owns the global
env ids [r*B, (r+1)*B) and seeds its episodes from the GLOBAL episode id, which makes every trajectory
independent of the number of ranks.  The only exchange is an all-gather of fixed-size per-episode records at the
shard boundary (end of Explorer.run_k_episodes, crowd_nav/utils/explorer.py:74-90 needs them on one rank)."""
import torch
import torch.distributed as dist

RECORD_FIELDS = ('outcome', 'steps', 'discounted_return', 'nav_time', 'danger_steps', 'danger_dmin_sum')


def shard(rank, world, envs_per_rank):
    """(env_offset, env_stride) for cn_rollout_io on rank `rank` of `world`."""
    if not 0 <= rank < world:
        raise ValueError('rank %d outside world of %d' % (rank, world))
    return rank * envs_per_rank, world * envs_per_rank


def pack_records(bufs, max_records=None):
    """rollout buffers (BatchedCrowdSim.rollout_begin) -> float64 [B, K, 6] record tensor + int64 [B] counts."""
    K = bufs['ep_outcome'].shape[1] if max_records is None else max_records
    cols = [bufs[n][:, :K].to(torch.float64) for n in ('ep_outcome', 'ep_steps', 'ep_return', 'ep_time',
                                                      'ep_danger', 'ep_danger_dmin_sum')]
    return torch.stack(cols, dim=2).contiguous(), bufs['ep_count'].to(torch.int64).clamp(max=K)


def gather_records(records, counts, group=None):
    """All-gather [B, K, F] records and [B] counts from every rank; returns ([W*B, K, F], [W*B]) ordered by global
    env id (rank-major), identical on every rank.  One collective of B*K*F*8 bytes per rank (96 KiB at 4096 envs,
    K = 1): latency-bound, never on the step path."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return records, counts
    world = dist.get_world_size(group)
    flat = torch.cat([records.reshape(records.shape[0], -1), counts.to(records.dtype)[:, None]], dim=1).contiguous()
    out = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(out, flat, group=group)
    allr = torch.cat(out, dim=0)
    return allr[:, :-1].reshape((-1,) + tuple(records.shape[1:])), allr[:, -1].to(torch.int64)


def episodes_in_global_order(records, counts, total_envs):
    """Flatten gathered records to a list ordered by global episode id c = g + j * total_envs."""
    K = records.shape[1]
    rows = []
    for j in range(K):
        have = counts > j
        idx = torch.nonzero(have, as_tuple=False)[:, 0]
        rows.append((idx + j * total_envs, records[idx, j]))
    ids = torch.cat([r[0] for r in rows])
    vals = torch.cat([r[1] for r in rows])
    order = torch.argsort(ids)
    return ids[order], vals[order]
