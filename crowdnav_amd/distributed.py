"""Sharding of the env axis over ranks (one process per GPU, torch.distributed; backend 'nccl' = RCCL over xGMI
on MI355X, 'gloo' in CPU tests).  Envs never interact, so the step path has NO collective: rank r owns the global
env ids [r*B, (r+1)*B) and seeds its episodes from the GLOBAL episode id, which makes every trajectory
independent of the number of ranks.  The only exchange is ONE all-gather of fixed-size per-env record blocks at the
shard boundary (end of Explorer.run_k_episodes, crowd_nav/utils/explorer.py:74-90 needs them on one rank).

A record block is what cn_rollout_records packs per env (include/crowdnav_amd.h): float64 [1 + 6 K] =
(episodes finished, K x RECORD_FIELDS).  Record j is RING SLOT j of the env's record ring: its j-th finished episode while
the env has finished at most record_capacity episodes; once the ring has wrapped, the most recent episode whose ordinal
is congruent to j.  episodes_in_global_order therefore refuses wrapped rings — an env that finished more episodes than
its ring's record_capacity (NOT more than the K <= record_capacity records a block carries: that is a truncated view of
valid slots) — so size the rings for the run.
pack_blocks is its host-side restatement (CPU tests, and the reference for the GPU test of the kernel)."""
import torch
import torch.distributed as dist

RECORD_FIELDS = ('outcome', 'steps', 'discounted_return', 'nav_time', 'danger_steps', 'danger_dmin_sum')


def shard(rank, world, envs_per_rank):
    """(env_offset, env_stride) for cn_rollout_io on rank `rank` of `world`."""
    if not 0 <= rank < world:
        raise ValueError('rank %d outside world of %d' % (rank, world))
    return rank * envs_per_rank, world * envs_per_rank


def pack_blocks(bufs, max_records=None):
    """rollout buffers (BatchedCrowdSim.rollout_begin) -> float64 [B, 1 + 6 K] record blocks, exactly what
    BatchedCrowdSim.rollout_records (cn_rollout_records) produces on the device."""
    cap = bufs['ep_outcome'].shape[1]
    K = cap if max_records is None else int(max_records)
    counts = bufs['ep_count'].to(torch.int64)
    cols = []
    for n in ('ep_outcome', 'ep_steps', 'ep_return', 'ep_time', 'ep_danger', 'ep_danger_dmin_sum'):
        c = torch.zeros(counts.shape[0], K, dtype=torch.float64, device=counts.device)
        c[:, :min(K, cap)] = bufs[n][:, :K].to(torch.float64)
        cols.append(c)
    rec = torch.stack(cols, dim=2)
    j = torch.arange(K, device=counts.device)[None, :]
    rec = rec * ((j < counts[:, None]) & (j < cap))[:, :, None]
    return torch.cat([counts.to(torch.float64)[:, None], rec.reshape(counts.shape[0], -1)], dim=1).contiguous()


def split_blocks(blocks, record_capacity=None):
    """[n, 1 + 6 K] blocks -> (records [n, K, 6], counts int64 [n] = records actually held per env)."""
    K = (blocks.shape[1] - 1) // len(RECORD_FIELDS)
    cap = K if record_capacity is None else min(K, int(record_capacity))
    return blocks[:, 1:].reshape(-1, K, len(RECORD_FIELDS)), blocks[:, 0].to(torch.int64).clamp(max=cap)


def gather_blocks(blocks, group=None):
    """All-gather [B, 1 + 6 K] record blocks from every rank: [W*B, 1 + 6 K] ordered by global env id (rank-major),
    identical on every rank.  ONE collective of B * (1 + 6 K) * 8 bytes per rank (224 KiB at 4096 envs, K = 1):
    latency-bound, never on the step path.  Without an initialised process group the shard is the whole job; with one —
    a world of one rank included — the collective really runs."""
    if not (dist.is_available() and dist.is_initialized()):
        return blocks
    world = dist.get_world_size(group)
    if dist.get_backend(group) == 'gloo' and blocks.is_cuda:
        # gloo (CPU tests, and bench.py's one-GPU execution of the N > 1 path) moves the blocks through the host
        host = blocks.detach().cpu().contiguous()
        out = torch.empty((world * host.shape[0], host.shape[1]), dtype=host.dtype)
        dist.all_gather_into_tensor(out, host, group=group)
        return out.to(blocks.device)
    out = torch.empty((world * blocks.shape[0], blocks.shape[1]), dtype=blocks.dtype, device=blocks.device)
    dist.all_gather_into_tensor(out, blocks.contiguous(), group=group)
    return out


def episodes_in_global_order(records, counts, total_envs, finished, record_capacity=None):
    """Flatten gathered records to a list ordered by global episode id c = g + j * total_envs.
    finished: the unclamped episode counts (blocks[:, 0]); record_capacity: slots of an env's record ring (default: the K
    records a block carries).  An env that finished more episodes than its RING holds has overwritten slots, and slot j is
    then not episode j (include/crowdnav_amd.h: record_capacity) — refused.  A block that carries fewer records than the
    ring holds (K < record_capacity) is not a wrap: its slots are valid, the list is merely truncated to K per env."""
    K = records.shape[1]
    cap = K if record_capacity is None else int(record_capacity)
    over = finished.to(torch.int64) > cap
    if bool(over.any()):
        raise ValueError('record rings have wrapped: %d env(s) finished more episodes (up to %d) than their %d record slots, '
                         'slot j is no longer episode j; use a larger record_capacity'
                         % (int(over.sum()), int(finished.max()), cap))
    rows = []
    for j in range(K):
        have = counts > j
        idx = torch.nonzero(have, as_tuple=False)[:, 0]
        rows.append((idx + j * total_envs, records[idx, j]))
    ids = torch.cat([r[0] for r in rows])
    vals = torch.cat([r[1] for r in rows])
    order = torch.argsort(ids)
    return ids[order], vals[order]


def episodes_from_blocks(blocks, total_envs, record_capacity=None):
    """gathered [n, 1 + 6 K] blocks -> (global episode ids, records) in global episode order; refuses wrapped rings."""
    records, counts = split_blocks(blocks, record_capacity)
    return episodes_in_global_order(records, counts, total_envs, blocks[:, 0], record_capacity)
