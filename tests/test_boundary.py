"""Shard boundary and launch plumbing on the GPU: the record-block kernels (explorer.py:74-90), the RCCL all-gather —
through torch.distributed's 'nccl' backend AND through the C ABI's cn_gather_records on a communicator the test owns —
and the scenario ring's fill policy (launches inside the ring's budget skip the fill kernels; scenarios the register-only
generator gives up on go through the redo pool)."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def amd():
    import torch
    assert torch.cuda.is_available(), 'gpu tests need a MI355X'
    import crowdnav_amd
    return crowdnav_amd


def _np(t):
    return t.detach().cpu().numpy()


def _rollout(amd, B, launches, K=16, **cfg):
    eng = amd.BatchedCrowdSim(num_envs=B, robot_policy=amd.ROBOT_ORCA, robot_visible=1, **cfg)
    bufs = eng.rollout_begin(seed_base=1000, seed_mod=500, episode_limit=-1, record_capacity=K)
    for n in launches:
        eng.rollout(n)
    eng.sync()
    return eng, bufs


def _oracle_rollout(oracle_mod, B, steps, K=16, **cfg):
    o = oracle_mod.CrowdOracle(num_envs=B, robot_policy=1, robot_visible=1, **cfg)
    o.reset(1000 + np.arange(B))
    ep_index, cur_steps, cur_ret = np.zeros(B, np.int32), np.zeros(B, np.int32), np.zeros(B, np.float64)
    total, rec = o.rollout(steps, 1000, 500, K, ep_index, cur_steps, cur_ret)
    return o, total, rec, cur_steps


def _same_as_oracle(eng, bufs, o, total, rec, cur_steps, B, steps):
    assert int(_np(bufs['transitions'])[0]) == total == B * steps
    cnt = _np(bufs['ep_count'])
    assert np.array_equal(cnt, rec['count'])
    for b in range(B):
        k = min(cnt[b], bufs['ep_outcome'].shape[1])
        assert np.array_equal(_np(bufs['ep_outcome'])[b, :k], rec['outcome'][b, :k])
        assert np.array_equal(_np(bufs['ep_steps'])[b, :k], rec['steps'][b, :k])
    assert np.array_equal(_np(bufs['cur_steps']), cur_steps)
    assert np.abs(_np(eng.get_state()[0]) - o.get_state()[0]).max() <= 1e-9


def test_short_launches_inside_the_ring_budget_change_nothing(amd, oracle_mod):
    """42 launches of 5 steps: only every 9th tops the 48-deep scenario ring up (5 * 9 <= 48), the others skip the fill
    kernels; the result equals the oracle's 210-step rollout, i.e. also the one-launch run."""
    B, steps = 128, 210
    o, total, rec, cur = _oracle_rollout(oracle_mod, B, steps)
    eng, bufs = _rollout(amd, B, [5] * 42)
    _same_as_oracle(eng, bufs, o, total, rec, cur, B, steps)
    eng2, bufs2 = _rollout(amd, B, [210])
    assert np.array_equal(_np(eng.get_state()[0]), _np(eng2.get_state()[0]))
    for k in ('ep_count', 'ep_steps', 'ep_return', 'ep_outcome', 'cur_steps', 'cur_return'):
        assert np.array_equal(_np(bufs[k]), _np(bufs2[k])), k


def test_ring_redo_pool_regenerates_long_rejection_chains_exactly(amd, oracle_mod):
    """A crowded circle (8 humans, radius 2.6): many scenarios need more than the 113 random() calls the register-only
    generator of the fill kernel holds and go through ring_redo_kernel's memory-backed pool; episodes and auto-resets
    still equal the oracle's (which draws every scenario from a plain MT19937)."""
    B, steps = 96, 160
    cfg = dict(num_humans=8, circle_radius=2.6)
    o, total, rec, cur = _oracle_rollout(oracle_mod, B, steps, **cfg)
    assert rec['count'].sum() > 3 * B  # plenty of ring scenarios consumed
    eng, bufs = _rollout(amd, B, [40, 40, 40, 40], **cfg)
    _same_as_oracle(eng, bufs, o, total, rec, cur, B, steps)


@pytest.mark.parametrize('B,H,R,steps', [(3, 2, 4.0, 150), (5, 7, 5.0, 150), (7, 3, 4.0, 151), (4, 12, 8.0, 120)])
def test_fused_rollout_with_auto_reset_at_edge_sizes(amd, oracle_mod, B, H, R, steps):
    """Odd batch sizes (the last wave is padded), crowds on either side of every kernel boundary (the four-barrier fused
    kernel up to 5 humans, the general 5- and 10-half-plane kernels, lane- and wave-per-scenario generators): episodes,
    auto-resets and end states of cn_rollout equal the oracle's, whatever the launch lengths."""
    cfg = dict(num_humans=H, circle_radius=R)
    o, total, rec, cur = _oracle_rollout(oracle_mod, B, steps, **cfg)
    assert rec['count'].sum() >= B  # every env went through at least one auto-reset on average
    eng, bufs = _rollout(amd, B, [steps // 3, steps - steps // 3 - 7, 7], **cfg)
    _same_as_oracle(eng, bufs, o, total, rec, cur, B, steps)


def test_record_block_kernel_equals_host_pack(amd):
    import torch
    from crowdnav_amd import distributed as cd
    eng, bufs = _rollout(amd, 256, [150], K=4)
    for K in (4, 2, 6):
        got = eng.rollout_records(max_records=K)
        want = cd.pack_blocks({k: v for k, v in bufs.items()}, max_records=K)
        assert got.shape == (256, 1 + 6 * K)
        assert torch.equal(got, want)
    rec, cnt = cd.split_blocks(eng.rollout_records(), record_capacity=4)
    assert int(cnt.max()) <= 4 and int(_np(bufs['ep_count']).max()) >= 3
    assert set(np.unique(_np(rec[:, 0, 0])).tolist()) <= {0.0, 2.0, 3.0, 4.0}


@pytest.mark.parametrize('n_envs', [1000, 2500])
def test_summary_kernel_is_the_explorer_statistics_and_bitwise_reproducible(amd, n_envs):
    """8 000 and 20 000 (env, record) items: inside one grid stride of the 64 x 256-thread kernel, and beyond it."""
    import torch
    from crowdnav_amd import distributed as cd
    eng, bufs = _rollout(amd, n_envs, [200], K=8)
    blocks = eng.rollout_records()
    s = _np(eng.records_summary(blocks))
    assert np.array_equal(s, _np(eng.records_summary(blocks.clone())))  # fixed summation order
    rec, cnt = (_np(t) for t in cd.split_blocks(blocks))
    held = [rec[b, j] for b in range(n_envs) for j in range(cnt[b])]
    out = np.array([r[0] for r in held])
    assert s[0] == _np(bufs['ep_count']).sum() and s[1] == len(held)
    assert (s[2], s[3], s[4]) == ((out == 2).sum(), (out == 3).sum(), (out == 4).sum())
    assert abs(s[5] - sum(r[3] for r in held if r[0] == 2)) < 1e-6
    assert abs(s[6] - sum(r[2] for r in held)) < 1e-9 + 1e-12 * len(held)  # another summation order
    assert s[7] == sum(r[4] for r in held)
    # explorer.py:74-80 from these eight numbers
    success_rate, collision_rate = s[2] / s[1], s[3] / s[1]
    assert 0.0 < success_rate <= 1.0 and 0.0 <= collision_rate < 1.0
    assert 8.0 < s[5] / s[2] < 25.0  # average nav time of the successful episodes
    assert torch.cuda.is_available()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_rccl_all_gather_runs_at_world_one(amd):
    """torch.distributed 'nccl' (= RCCL) initialised with ONE rank: gather_blocks does not shortcut, the collective
    runs on the GPU and returns the shard unchanged."""
    import torch
    import torch.distributed as dist
    from crowdnav_amd import distributed as cd
    eng, _ = _rollout(amd, 512, [120], K=2)
    blocks = eng.rollout_records()
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(_free_port())
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        got = cd.gather_blocks(blocks)
        torch.cuda.synchronize()
        assert got.data_ptr() != blocks.data_ptr() and torch.equal(got, blocks)
        total = eng.records_summary(got)
        dist.all_reduce(total)
        assert torch.equal(total, eng.records_summary(blocks))
    finally:
        dist.destroy_process_group()


def test_c_abi_gather_over_a_caller_owned_rccl_communicator(amd):
    """INTEGRATION seam 2: a ctypes consumer creates its own ncclComm_t (world of one here) and cn_gather_records
    all-gathers the record blocks over it on the engine's stream."""
    import torch
    from crowdnav_amd import rccl
    eng, _ = _rollout(amd, 512, [120], K=3)
    blocks = eng.rollout_records()
    torch.cuda.set_device(0)
    comm = rccl.comm_init_rank(1, rccl.get_unique_id(), 0)
    try:
        got = eng.gather_records_rccl(comm, 1, blocks)
        eng.sync()
        assert got.shape == blocks.shape and torch.equal(got, blocks)
        with pytest.raises(amd.CrowdNavAmdError):
            eng.gather_records_rccl(0, 1, blocks)  # NULL communicator
    finally:
        rccl.comm_destroy(comm)


def test_asynchronous_scenario_fill_changes_timing_not_trajectories(amd, oracle_mod):
    """CN_FLAG_ASYNC_SCENARIO_FILL (crowds of more than 8 humans): the fill kernels run beside the transition kernels on
    side streams and publish every scenario on its own; an env whose next scenario is not ready pauses.  Whatever the
    timing, every episode an env DID finish is the oracle's episode (same outcome, length, return), in order, and the env
    states are those of the oracle after the same number of that env's transitions — here checked through the records."""
    B, K, launches, n = 96, 24, 30, 10
    cfg = dict(num_humans=10, circle_radius=3.2, robot_visible=1)
    eng = amd.BatchedCrowdSim(num_envs=B, robot_policy=amd.ROBOT_ORCA, flags=amd.FLAG_ASYNC_SCENARIO_FILL, **cfg)
    bufs = eng.rollout_begin(seed_base=1000, seed_mod=500, episode_limit=-1, record_capacity=K)
    for _ in range(launches):
        eng.rollout(n)
    eng.sync()
    total = int(_np(bufs['transitions'])[0])
    assert 0 < total <= B * launches * n
    o = oracle_mod.CrowdOracle(num_envs=B, robot_policy=1, **cfg)
    o.reset(1000 + np.arange(B))
    ep_index, cur_steps, cur_ret = np.zeros(B, np.int32), np.zeros(B, np.int32), np.zeros(B, np.float64)
    _, rec = o.rollout(launches * n, 1000, 500, K, ep_index, cur_steps, cur_ret)
    cnt = _np(bufs['ep_count'])
    assert (cnt <= rec['count']).all()                      # a paused env is behind, never ahead
    assert cnt.sum() >= 0.8 * rec['count'].sum() >= B       # and pauses are the exception at this geometry
    for b in range(B):
        k = min(cnt[b], K)
        assert np.array_equal(_np(bufs['ep_outcome'])[b, :k], rec['outcome'][b, :k])
        assert np.array_equal(_np(bufs['ep_steps'])[b, :k], rec['steps'][b, :k])
        assert np.allclose(_np(bufs['ep_return'])[b, :k], rec['ret'][b, :k], rtol=0, atol=1e-9)
    # the synchronous engine on the same seeds: identical to the oracle transition for transition
    eng2, bufs2 = _rollout(amd, B, [n] * launches, K=K, **{k: v for k, v in cfg.items() if k != 'robot_visible'})
    assert int(_np(bufs2['transitions'])[0]) == B * launches * n and np.array_equal(_np(bufs2['ep_count']), rec['count'])


@pytest.mark.parametrize('B,H,R,launches', [(300, 5, 4.0, [20, 1, 100]), (37, 20, 9.0, [60, 45]), (16, 20, 9.0, [48, 90])])
def test_per_env_transition_counters_and_statistics_on_request(amd, monkeypatch, B, H, R, launches):
    """ABI v6 (`cn_rollout_io.env_transitions`): with neither the job-wide counter nor the in-kernel summary requested, a
    launch ends without any hand-off between its workgroups; every env counts its own transitions, and the statistics computed
    on request (cn_rollout_records + cn_records_summary) equal what the launches of a second, identical run left behind
    themselves (rollout_epilogue) — counts exactly, float sums to rounding; the episodes are the same ones bit for bit.  Also
    through the 20-human shard's 3-of-4 env schedule (forced on 16 envs)."""
    import torch
    monkeypatch.setenv('CROWDNAV_AMD_SCHED_FORCE', '1')
    K = 4

    def run(per_env):
        eng = amd.BatchedCrowdSim(num_envs=B, num_humans=H, robot_policy=amd.ROBOT_ORCA, robot_visible=1, circle_radius=R)
        bufs = eng.rollout_begin(seed_base=1000, seed_mod=500, episode_limit=-1, record_capacity=K,
                                 boundary_records=0 if per_env else 2, per_env_transitions=per_env)
        for n in launches:
            eng.rollout(n)
        eng.sync()
        return eng, bufs

    eng, bufs = run(True)
    assert 'transitions' not in bufs and 'summary' not in bufs
    assert torch.equal(bufs['env_transitions'], torch.full((B,), sum(launches), dtype=torch.int64, device=bufs['env_transitions'].device))
    eng2, bufs2 = run(False)
    assert int(_np(bufs2['transitions'])[0]) == B * sum(launches)
    for k in ('ep_count', 'ep_outcome', 'ep_steps', 'ep_return', 'cur_steps', 'cur_return'):
        assert torch.equal(bufs[k], bufs2[k]), k
    got = _np(eng.records_summary(eng.rollout_records(), record_capacity=K))
    assert torch.equal(eng.rollout_summary(), eng.records_summary(eng.rollout_records(), record_capacity=K))  # same bits, one kernel
    want = _np(bufs2['summary'])
    assert np.array_equal(got[:5], want[:5]) and got[7] == want[7] and got[0] == _np(bufs['ep_count']).sum() > 0
    assert np.abs(got[5:7] - want[5:7]).max() <= 1e-9 * max(1.0, np.abs(want[5:7]).max())


def test_seed_numbering_is_fixed_by_rollout_begin(amd):
    """cn_rollout / cn_rollout_step accept a changed io, but the scenario cache of a 20-human rollout was sized by
    cn_rollout_begin's seed_mod and filled for its seed_base — a later io with other values is refused (CN_ERR_INVALID) instead
    of indexing past the cache or being served the old seeds' scenarios; a new cn_rollout_begin renumbers."""
    eng = amd.BatchedCrowdSim(num_envs=8, num_humans=20, robot_policy=amd.ROBOT_ORCA, robot_visible=1, circle_radius=12.0)
    eng.rollout_begin(seed_base=1000, seed_mod=7, episode_limit=-1, record_capacity=2)
    eng.rollout(30)
    io = eng._rollout[0]
    for field, value in (('seed_mod', 100000), ('seed_base', 5)):
        keep = getattr(io, field)
        setattr(io, field, value)
        with pytest.raises(amd.CrowdNavAmdError, match='cn_rollout_begin'):
            eng.rollout(3)
        setattr(io, field, keep)
    eng.rollout(3)  # the unchanged io still runs
    eng.rollout_begin(seed_base=5, seed_mod=100000, episode_limit=-1, record_capacity=2)
    eng.rollout(30)
    eng.sync()
    eng.close()
