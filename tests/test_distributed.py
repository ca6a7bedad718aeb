"""N>1 path on CPU: world_size-2 gloo processes exchange episode records exactly as ranks do over RCCL."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _fake_bufs(rank, B, K, world):
    """Deterministic stand-in for what the rollout kernel leaves in the record buffers of one shard."""
    g = torch.arange(B) + rank * B  # global env ids
    counts = (g % (K + 1)).to(torch.int32)
    j = torch.arange(K)[None, :]
    c = g[:, None] + j * (world * B)  # global episode ids
    return dict(ep_outcome=(2 + c % 3).to(torch.uint8), ep_steps=(10 + c % 50).to(torch.int32),
                ep_return=c.to(torch.float64) * 0.001, ep_time=c.to(torch.float64) * 0.25,
                ep_danger=(c % 4).to(torch.int32), ep_danger_dmin_sum=c.to(torch.float64) * 0.01, ep_count=counts)


def _worker(rank, world, port, B, K, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from crowdnav_amd import distributed as cd
    assert cd.shard(rank, world, B) == (rank * B, world * B)
    allb = cd.gather_blocks(cd.pack_blocks(_fake_bufs(rank, B, K, world)))
    allr, allc = cd.split_blocks(allb)
    ids, vals = cd.episodes_in_global_order(allr, allc, world * B, allb[:, 0])
    ids2, vals2 = cd.episodes_from_blocks(allb, world * B)
    assert torch.equal(ids, ids2) and torch.equal(vals, vals2)
    ret[rank] = (allr.clone(), allc.clone(), ids.clone(), vals.clone())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_gather_records_world2_gloo():
    world, B, K = 2, 8, 3
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), B, K, ret), nprocs=world, join=True)
    r0, r1 = ret[0], ret[1]
    for a, b in zip(r0, r1):
        assert torch.equal(a, b)  # identical on every rank
    allr, allc, ids, vals = r0
    assert allr.shape == (world * B, K, 6) and allc.tolist() == [(g % (K + 1)) for g in range(world * B)]
    # what a single process owning all 16 envs would hold
    from crowdnav_amd import distributed as cd
    want = [cd.split_blocks(cd.pack_blocks(_fake_bufs(r, B, K, world))) for r in range(world)]
    assert torch.equal(allr, torch.cat([w[0] for w in want])) and torch.equal(allc, torch.cat([w[1] for w in want]))
    # global episode order: ids strictly increasing, discounted_return column = 0.001 * id
    assert torch.all(ids[1:] > ids[:-1]) and len(ids) == int(allc.sum())
    assert torch.allclose(vals[:, 2], ids.to(torch.float64) * 0.001)


def test_gather_is_identity_without_process_group():
    from crowdnav_amd import distributed as cd
    blocks = cd.pack_blocks(_fake_bufs(0, 4, 2, 1))
    assert cd.gather_blocks(blocks) is blocks
    rec, cnt = cd.split_blocks(blocks)
    assert rec.shape == (4, 2, 6) and cnt.tolist() == [0, 1, 2, 0]
    assert float(rec[0].abs().sum()) == 0.0 and float(rec[1, 1].abs().sum()) == 0.0  # nothing beyond what an env holds
    with pytest.raises(ValueError):
        cd.shard(2, 2, 4)


def test_only_a_wrapped_ring_is_refused():
    """an env that finished more episodes than its RING holds is refused; a block that merely carries fewer
    records than the ring holds (K < record_capacity) is a truncated view of valid slots and is not."""
    from crowdnav_amd import distributed as cd
    bufs = _fake_bufs(0, 6, 3, 1)                      # ring of 3 slots, envs finished 0, 1, 2, 3, 0, 1 episodes
    blocks = cd.pack_blocks(bufs)
    ids, vals = cd.episodes_from_blocks(blocks, 6)     # nothing wrapped
    assert len(ids) == int(bufs['ep_count'].sum())
    short = cd.pack_blocks(bufs, max_records=2)        # K = 2 < capacity 3: env 3 finished 3 > K episodes
    ids2, _ = cd.episodes_from_blocks(short, 6, record_capacity=3)
    assert len(ids2) == int(bufs['ep_count'].clamp(max=2).sum())
    wrapped = dict(bufs, ep_count=bufs['ep_count'] + 2)   # env 3: 5 episodes through 3 slots
    with pytest.raises(ValueError, match='wrapped'):
        cd.episodes_from_blocks(cd.pack_blocks(wrapped), 6)


@pytest.mark.timeout(180)
def test_bench_self_launches_its_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment re-executes itself as two ranks (what the
    driver's N=1 shape of the command line does for N>1).  Without GPUs here every rank must stop with a clear message
    and the parent must return non-zero instead of hanging in a rendezvous."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip('this check is for the GPU-less container')
    env['HIP_VISIBLE_DEVICES'] = ''
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '5', '--warmup', '1'],
                       env=env, capture_output=True, text=True, timeout=150)
    assert p.returncode != 0
    # the first rank to fail makes the parent terminate the other one, which may not have reached its own message yet
    assert 1 <= p.stderr.count('needs GPU') <= 2 and 'rendezvous' not in p.stderr.lower()
