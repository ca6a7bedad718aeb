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
