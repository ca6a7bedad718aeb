"""The 3-of-4 env schedule of the 20-human shard's kernel as plain arithmetic (crowdnav_amd/csrc/step_kernels.h: rollout_body's
`env_block` / `extra_env`; crowdnav_amd.hip: launch_rollout).  A call of 3 q steps is four launches of q steps over 3 B / 4
workgroups; workgroup w of sub-launch k plays env 4 (w / 3) + (w % 3) + [w % 3 >= 3 - k].  Checked here for every B that is a
multiple of four: each sub-launch maps its workgroups onto distinct envs, every env is played in exactly three of the four
sub-launches (so it makes 3 q steps, in launch order), the env a group leaves out in the LAST sub-launch is the one its first
workgroup reports for (rollout_epilogue: extra_env), and the host only chooses the schedule where it saves rounds."""
import pytest


def env_of(w, k):
    g, r = divmod(w, 3)
    return 4 * g + r + (1 if r >= 3 - k else 0)


@pytest.mark.parametrize('B', [4, 8, 36, 4096, 8192])
def test_every_env_plays_three_of_four_sub_launches(B):
    played = [0] * B
    for k in range(4):
        envs = [env_of(w, k) for w in range(B // 4 * 3)]
        assert len(set(envs)) == len(envs) and 0 <= min(envs) and max(envs) < B
        for e in envs:
            played[e] += 1
        left_out = sorted(set(range(B)) - set(envs))
        assert left_out == [4 * g + (3 - k) for g in range(B // 4)]
        if k == 3:  # the last sub-launch: workgroup 3 g (r == 0) adds the record ring of env 4 g, the one its group leaves out
            assert left_out == [4 * (w // 3) for w in range(B // 4 * 3) if w % 3 == 0]
    assert played == [3] * B


@pytest.mark.parametrize('B,cus,want', [(4096, 256, True), (3072, 256, False), (2048, 256, False), (8192, 256, True),
                                        (6144, 256, False), (4096, 304, True), (3648, 304, False), (16, 256, False)])
def test_schedule_is_chosen_only_where_it_saves_rounds(B, cus, want):
    slots = 12 * cus  # three one-wave workgroups per SIMD
    rounds_plain = -(-B // slots)
    rounds_sched = -(-(B // 4 * 3) // slots)
    assert (B % 4 == 0 and 4 * rounds_sched < 3 * rounds_plain) == want


# ------------------------------------------------------------------------------------------------ dynamic schedule (round 5)
# step_kernels.h: kSchedDynamic — persistent workgroups take (env, visit) items from a device queue, visit-major; visit k of an
# env may start only when the env's k earlier visits are complete.  Emulated as a discrete-event simulation: G workers, item v =
# (env v % B, visit v // B), arbitrary per-item durations.  Claims checked: no deadlock for any G >= 1 and any durations; every
# env's visits run in order and never overlap; the split of a call's n steps over its visits adds up; the makespan stays
# within Graham's bound (total work / G + the longest env's own chain of visits).
def _simulate(B, V, G, dur):
    import heapq
    done_at = {}                      # (env, k) -> completion time
    free = [(0.0, w) for w in range(G)]
    heapq.heapify(free)
    spans, next_item = [], 0
    while next_item < B * V:
        t, w = heapq.heappop(free)    # the worker that becomes free first dequeues the next item (atomicAdd on the queue)
        env, k = next_item % B, next_item // B
        next_item += 1
        start = t if k == 0 else max(t, done_at[(env, k - 1)])  # spins until the env's previous visit has published its state
        end = start + dur(env, k)
        done_at[(env, k)] = end
        spans.append((env, k, start, end))
        heapq.heappush(free, (end, w))
    return spans


@pytest.mark.parametrize('B,V,G', [(4096, 18, 3072), (4096, 3, 3072), (16, 3, 16), (7, 5, 3), (5000, 9, 3072), (12, 4, 1)])
def test_dynamic_schedule_runs_every_envs_visits_in_order_without_deadlock(B, V, G):
    import random
    rng = random.Random(B * 131 + V * 7 + G)
    jam = {e for e in range(B) if rng.random() < 0.1}  # envs in a jam: their visits take 60 % longer
    spans = _simulate(B, V, G, lambda e, k: (1.6 if e in jam else 1.0) * rng.uniform(0.9, 1.1))
    assert len(spans) == B * V
    last = {}
    for env, k, start, end in spans:                  # in dequeue order
        assert k == last.get(env, (-1, 0.0))[0] + 1   # visit k follows visit k - 1 ...
        assert start >= last.get(env, (-1, 0.0))[1]   # ... and starts after it has ended
        last[env] = (k, end)
    assert all(last[e][0] == V - 1 for e in range(B))
    work = sum(end - start for _, _, start, end in spans)
    makespan = max(end for _, _, _, end in spans)
    chain = {}
    for env, _, start, end in spans:
        chain[env] = chain.get(env, 0.0) + (end - start)
    # Graham's bound for list scheduling under precedence constraints: total work / workers + the longest chain (an env that
    # sits in a jam for the whole call is a critical path no schedule shortens); and never better than either of the two
    assert max(work / G, max(chain.values())) - 1e-9 <= makespan <= work / G + max(chain.values()) + 1e-9


@pytest.mark.parametrize('n,want', [(999, 18), (500, 9), (150, 3), (47, 3), (24, 3), (2997, 54)])
def test_visits_per_call_and_their_lengths(n, want):
    visits = max(3, (n + 28) // 56)          # crowdnav_amd.hip: launch_rollout (~56 steps per visit, at least three)
    visits = min(visits, n)
    assert visits == want
    q, rem = divmod(n, visits)
    lengths = [q + (1 if k < rem else 0) for k in range(visits)]   # step_kernels.h: n_steps of visit k
    assert sum(lengths) == n and max(lengths) - min(lengths) <= 1
