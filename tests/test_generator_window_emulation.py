"""CPU emulation of the two round-4 shortcuts of the wave-cooperative scenario generator (crowdnav_amd/csrc/scenario_wave.h),
against the plain sequential form of the reference's rejection sampling (/root/reference crowd_sim/envs/crowd_sim.py:155-176:
`generate_circle_crossing_human` — angle, px_noise, py_noise per attempt; reject inside min_dist of any placed position or goal):

* WINDOW REUSE (`CN_GEN_WINDOW`): a wave evaluates the 64 attempts at the stream cursor once and keeps taking humans from that
  window — the lanes behind an accepted attempt are the next human's first attempts and only have to clear the human just
  placed.  Emulated lane for lane (numpy arrays of 64), including the give-up rule (a human's attempts are counted in passes of
  64 from ITS first attempt; attempt N - 64 is taken when N have failed) — must place every human exactly where the sequential
  loop does and leave the stream at the same position, for any cap and any crowding.
* FLOAT32 PREFILTER (`CN_GEN_PREFILTER`): an attempt is rejected without the float64 arithmetic only if its float32 position is
  closer than min_dist - margin to a placed point (margin = 10 x the float32 position's error bound: prefilter_margin).  Checked
  here on a million attempts per radius: whatever the prefilter rejects, the exact test rejects (numpy's float32 sin / cos,
  perturbed by the hardware V_SIN_F32 / V_COS_F32 error bound kTrigAbsError, stand in for the device's).
* BLOCKED-CELL TABLE (round 6): a 160 x 160 bitmap over the square the attempts can fall in; a cell is set when it lies entirely
  within min_dist - table_margin of a placed point (marked row by row in float32, as `BlockedGrid::mark` does); an attempt in a
  set cell is rejected without any distance test, survivors go through the prefilter and the exact test ONE BY ONE in stream
  order.  Emulated in float32 numpy: conservative on a million attempts, >= 99 % of the exact rejections at R = 4, and the
  windowed generator built on it places every human where the sequential loop does.

The GPU suite checks the kernel itself against the oracle's MT19937 stream (tests/test_gpu_parity.py: reset vs the reference
generator, draw counts; wave vs lane generators); this file pins the ALGORITHM on machines without a GPU.
"""
import numpy as np
import pytest

PI = 3.141592653589793


TRIG_ABS_ERROR = 1.0e-5   # cn::kTrigAbsError (scenario_wave.h), asserted on the device by tests/test_generator_trig.py
F = np.float32


def prefilter_margin(R):
    return F(10.0) * (F(TRIG_ABS_ERROR) * F(R) + F(1.0e-6) * (F(R) + F(2.0)))


def table_margin(R):
    return prefilter_margin(R) + F(1.0e-4)


class BlockedGrid(object):
    """scenario_wave.h: BlockedGrid, operation for operation in float32"""
    N = 160

    def __init__(self, R, v_pref):
        self.ext = F(R + 0.5 * v_pref) / (F(1.0) - F(6.0) / F(self.N))
        self.cell = F(2.0) * self.ext / F(self.N)
        self.inv = F(self.N) / (F(2.0) * self.ext)
        self.bits = np.zeros((self.N, self.N), bool)   # [iy, ix]

    def locate(self, x, y):
        ix = np.clip(((F(x) + self.ext) * self.inv).astype(np.int64), 0, self.N - 1)
        iy = np.clip(((F(y) + self.ext) * self.inv).astype(np.int64), 0, self.N - 1)
        return iy, ix

    def blocked(self, x, y):
        iy, ix = self.locate(np.asarray(x, F), np.asarray(y, F))
        return self.bits[iy, ix]

    def mark(self, cx, cy, rt):
        cx, cy, rt = F(cx), F(cy), F(rt)
        if not rt > 0:
            return
        iy0 = int(np.floor((cy - rt + self.ext) * self.inv))
        iy1 = int(np.floor((cy + rt + self.ext) * self.inv))
        for iy in range(iy0, iy1 + 1):
            if iy < 0 or iy >= self.N:
                continue
            y0 = F(iy) * self.cell - self.ext
            dy = max(abs(y0 - cy), abs(y0 + self.cell - cy))
            w2 = rt * rt - dy * dy
            if not w2 > 0:
                continue
            w = np.sqrt(w2, dtype=F) * (F(1.0) - F(1.0e-6))
            lo = max(0, int(np.ceil((cx - w + self.ext) * self.inv)))
            hi = min(self.N, int(np.floor((cx + w + self.ext) * self.inv)))
            if lo < hi:
                self.bits[iy, lo:hi] = True


def attempts_of(seed, n):
    """the first n attempts of numpy's MT19937 stream seeded `seed`: (angle fraction, noise x, noise y) in [0, 1)"""
    return np.random.RandomState(seed).random_sample((n, 3))


def position(att, R, v_pref):
    ang = att[..., 0] * PI * 2
    return R * np.cos(ang) + (att[..., 1] - 0.5) * v_pref, R * np.sin(ang) + (att[..., 2] - 0.5) * v_pref


def collides(x, y, placed, min_dist):
    """placed: list of (px, py, gx, gy); np.linalg.norm((dx, dy)) < min_dist for a position or a goal"""
    hit = np.zeros(np.shape(x), bool)
    for px, py, gx, gy in placed:
        hit |= (np.hypot(x - px, y - py) < min_dist) | (np.hypot(x - gx, y - gy) < min_dist)
    return hit


def sequential(att, humans, R, v_pref, min_dist, cap):
    """the reference's loop with the generator's give-up rule: returns placed humans, attempts consumed, error flag"""
    N = (cap + 63) // 64 * 64
    placed, cur, err = [(0.0, -R, 0.0, R)], 0, False
    for _ in range(humans):
        k = 0
        while True:
            x, y = position(att[cur + k], R, v_pref)
            if not collides(x, y, placed, min_dist):
                break
            k += 1
            if k == N:  # N attempts failed: the first one of the last pass of 64 is taken
                k, err = N - 64, True
                x, y = position(att[cur + k], R, v_pref)
                break
        placed.append((x, y, -x, -y))
        cur += k + 1
    return placed[1:], cur, err


def windowed(att, humans, R, v_pref, min_dist, cap):
    """scenario_wave.h, the CN_GEN_WINDOW path, lane for lane"""
    N = (cap + 63) // 64 * 64
    lane = np.arange(64)
    placed, cursor, err = [(0.0, -R, 0.0, R)], 0, False
    win, wstart = False, 0
    wx = wy = wcol = None
    windows = 0
    for _ in range(humans):
        hstart = cursor + (wstart if win else 0)
        tried, giveup = 0, None
        while True:
            if not win:
                wx, wy = position(att[cursor:cursor + 64], R, v_pref)
                wcol = collides(wx, wy, placed, min_dist)
                win, wstart = True, 0
                windows += 1
            left = N - tried
            mine = (lane >= wstart) & (lane - wstart < left)
            idx = tried + (lane - wstart)
            g = np.nonzero((lane >= wstart) & (idx == N - 64))[0]
            if len(g):
                giveup = (wx[g[0]], wy[g[0]])
            ok = np.nonzero(mine & ~wcol)[0]
            if len(ok):
                first = ok[0]
                x, y = wx[first], wy[first]
                placed.append((x, y, -x, -y))
                behind = (lane > first) & ~wcol
                wcol = wcol | (behind & collides(wx, wy, [placed[-1]], min_dist))
                wstart = first + 1
                if wstart == 64:
                    cursor, win, wstart = cursor + 64, False, 0
                break
            tried += min(64 - wstart, left)
            if tried >= N:
                err = True
                x, y = giveup
                placed.append((x, y, -x, -y))
                cursor, win, wstart = hstart + N - 64 + 1, False, 0
                break
            cursor, win, wstart = cursor + 64, False, 0
    if win:
        cursor += wstart
    return placed[1:], cursor, err, windows


@pytest.mark.parametrize('R,humans,cap', [(12.0, 20, 1 << 23), (4.0, 12, 1 << 23), (3.0, 9, 1 << 23), (2.6, 8, 256), (2.2, 8, 128),
                                          (2.0, 7, 64), (2.0, 9, 100), (1.6, 6, 64)])
def test_window_reuse_places_every_human_where_the_sequential_loop_does(R, humans, cap):
    v_pref, min_dist = 1.0, 0.3 + 0.3 + 0.2
    gave_up = 0
    fewer = 0
    for seed in range(40):
        att = attempts_of(1000 + seed, 400000 if cap > 1000 else 20000)
        try:
            want, cur, err = sequential(att, humans, R, v_pref, min_dist, cap)
        except IndexError:  # a crowding this seed does not get out of within the attempts drawn here
            continue
        got, cur2, err2, windows = windowed(att, humans, R, v_pref, min_dist, cap)
        assert cur2 == cur and err2 == err
        assert np.array_equal(np.array(got), np.array(want))
        gave_up += err
        fewer += windows < humans
    if cap <= 256:
        assert gave_up > 0  # the give-up rule was exercised
    if R >= 12.0:
        assert fewer == 40  # an easy scenario takes fewer window evaluations than it has humans


@pytest.mark.parametrize('R', [4.0, 12.0, 40.0])
def test_float32_prefilter_rejects_only_what_the_exact_test_rejects(R):
    rng = np.random.RandomState(7)
    v_pref, min_dist = 1.0, 0.8
    margin = prefilter_margin(R)
    # 20 placed humans on the circle (positions and goals), as the generator would have them
    ang = rng.uniform(0, 2 * PI, 20)
    placed = [(R * np.cos(a) + rng.uniform(-.5, .5), R * np.sin(a) + rng.uniform(-.5, .5)) for a in ang]
    placed = [(px, py, -px, -py) for px, py in placed]
    words = rng.randint(0, 2 ** 32, size=(1000000, 6), dtype=np.uint64)
    # the exact path: np.random.random() from two words each (random_at)
    u = ((words[:, 0::2] >> 5).astype(np.float64) * 67108864.0 + (words[:, 1::2] >> 6).astype(np.float64)) / 9007199254740992.0
    x, y = position(u, R, v_pref)
    exact = collides(x, y, placed, min_dist)
    # the prefilter: the first word's top 27 bits, float32 throughout (scenario_wave.h)
    f = ((words[:, 0::2] >> 5).astype(np.float32) * np.float32(2.0 ** -27))
    a32 = f[:, 0] * np.float32(6.2831855)
    # (the device's V_COS_F32 / V_SIN_F32 are within kTrigAbsError of the true values: the worst case, both off by the bound)
    sign = np.where(rng.randint(0, 2, size=(len(f), 2)) > 0, np.float32(1), np.float32(-1)) * np.float32(TRIG_ABS_ERROR)
    fx = np.float32(R) * (np.cos(a32, dtype=np.float32) + sign[:, 0]) + (f[:, 1] - np.float32(0.5)) * np.float32(v_pref)
    fy = np.float32(R) * (np.sin(a32, dtype=np.float32) + sign[:, 1]) + (f[:, 2] - np.float32(0.5)) * np.float32(v_pref)
    t = np.float32(min_dist) - margin
    inside = np.zeros(len(fx), bool)
    for px, py, gx, gy in placed:
        for qx, qy in ((np.float32(px), np.float32(py)), (np.float32(gx), np.float32(gy))):
            ax, ay = fx - qx, fy - qy
            inside |= (ax * ax + ay * ay) < t * t
    assert not np.any(inside & ~exact)          # conservative
    assert inside.sum() > 0.98 * exact.sum()    # and it catches nearly everything the exact test rejects
    # the float32 position is well inside the bound the margin was derived from (margin = 10 x the bound)
    assert np.max(np.hypot(fx.astype(np.float64) - x, fy.astype(np.float64) - y)) < 0.2 * float(margin)


def float32_position(att, R, v_pref, rng=None):
    """the float32 position of the conservative stages: the top 27 bits of the angle / noise fractions, float32 throughout;
    with rng, cos / sin are additionally off by the hardware bound in a random direction"""
    f = (np.floor(np.asarray(att) * 2.0 ** 27)).astype(F) * F(2.0 ** -27)
    a = f[..., 0].astype(np.float64) * 2 * PI
    c, s_ = np.cos(a).astype(F), np.sin(a).astype(F)
    if rng is not None:
        c = c + np.where(rng.randint(0, 2, size=c.shape) > 0, F(1), F(-1)) * F(TRIG_ABS_ERROR)
        s_ = s_ + np.where(rng.randint(0, 2, size=c.shape) > 0, F(1), F(-1)) * F(TRIG_ABS_ERROR)
    return F(R) * c + (f[..., 1] - F(0.5)) * F(v_pref), F(R) * s_ + (f[..., 2] - F(0.5)) * F(v_pref)


@pytest.mark.parametrize('R', [2.0, 4.0, 12.0])
def test_blocked_cell_table_rejects_only_what_the_exact_test_rejects(R):
    rng = np.random.RandomState(11)
    v_pref, min_dist = 1.0, 0.8
    ang = rng.uniform(0, 2 * PI, 20)
    placed = [(R * np.cos(a) + rng.uniform(-.5, .5), R * np.sin(a) + rng.uniform(-.5, .5)) for a in ang]
    placed = [(0.0, -R, 0.0, R)] + [(px, py, -px, -py) for px, py in placed]
    grid = BlockedGrid(R, v_pref)
    for px, py, gx, gy in placed:
        grid.mark(px, py, F(min_dist) - table_margin(R))
        grid.mark(gx, gy, F(min_dist) - table_margin(R))
    att = rng.random_sample((1000000, 3))
    x, y = position(att, R, v_pref)
    exact = collides(x, y, placed, min_dist)
    fx, fy = float32_position(att, R, v_pref, rng)
    blocked = grid.blocked(fx, fy)
    assert not np.any(blocked & ~exact)   # conservative: every table rejection is an exact rejection
    if R == 4.0:
        assert blocked.sum() > 0.97 * exact.sum()   # the reference geometry: nearly every rejection is decided by one bit
    # a set cell lies within min_dist of its disc's centre even after the float32 position's error: check the corners
    iy, ix = np.nonzero(grid.bits)
    x0, y0 = ix * float(grid.cell) - float(grid.ext), iy * float(grid.cell) - float(grid.ext)
    slack = 0.5 * float(table_margin(R))
    for dx in (-slack, float(grid.cell) + slack):
        for dy in (-slack, float(grid.cell) + slack):
            assert collides(x0 + dx, y0 + dy, placed, min_dist).all()


def windowed_table(att, humans, R, v_pref, min_dist, cap):
    """scenario_wave.h, round 6: the window path with the blocked-cell table; survivors one by one in stream order"""
    N = (cap + 63) // 64 * 64
    lane = np.arange(64)
    placed, cursor, err = [(0.0, -R, 0.0, R)], 0, False
    grid = BlockedGrid(R, v_pref)
    rt = F(min_dist) - table_margin(R)
    grid.mark(0.0, -R, rt)
    grid.mark(0.0, R, rt)
    t2 = (F(min_dist) - prefilter_margin(R)) ** 2
    win, wstart = False, 0
    fx = fy = blocked = None
    stats = dict(windows=0, survivors=0, exact=0)
    for _ in range(humans):
        hstart = cursor + (wstart if win else 0)
        tried, giveup = 0, None
        while True:
            if not win:
                fx, fy = float32_position(att[cursor:cursor + 64], R, v_pref)
                blocked = grid.blocked(fx, fy)
                win, wstart = True, 0
                stats['windows'] += 1
            left = N - tried
            mine = (lane >= wstart) & (lane - wstart < left)
            g = np.nonzero((lane >= wstart) & (tried + (lane - wstart) == N - 64))[0]
            if len(g):
                giveup = position(att[cursor + g[0]], R, v_pref)
            first = -1
            for L in np.nonzero(mine & ~blocked)[0]:
                stats['survivors'] += 1
                inside = False
                for px, py, gx, gy in placed:
                    for qx, qy in ((F(px), F(py)), (F(gx), F(gy))):
                        ax, ay = fx[L] - qx, fy[L] - qy
                        inside |= bool(ax * ax + ay * ay < t2)
                if inside:
                    continue
                stats['exact'] += 1
                x, y = position(att[cursor + L], R, v_pref)
                if collides(x, y, placed, min_dist):
                    continue
                first = L
                break
            if first >= 0:
                placed.append((x, y, -x, -y))
                grid.mark(x, y, rt)
                grid.mark(-x, -y, rt)
                blocked = blocked | grid.blocked(fx, fy)
                wstart = first + 1
                if wstart == 64:
                    cursor, win, wstart = cursor + 64, False, 0
                break
            tried += min(64 - wstart, left)
            if tried >= N:
                err = True
                x, y = giveup
                placed.append((x, y, -x, -y))
                grid.mark(x, y, rt)
                grid.mark(-x, -y, rt)
                cursor, win, wstart = hstart + N - 64 + 1, False, 0
                break
            cursor, win, wstart = cursor + 64, False, 0
    if win:
        cursor += wstart
    return placed[1:], cursor, err, stats


@pytest.mark.parametrize('R,humans,cap', [(12.0, 20, 1 << 23), (4.0, 14, 1 << 23), (3.0, 9, 1 << 23), (2.6, 8, 256), (2.2, 8, 128),
                                          (2.0, 7, 64), (2.0, 9, 100)])
def test_table_generator_places_every_human_where_the_sequential_loop_does(R, humans, cap):
    v_pref, min_dist = 1.0, 0.3 + 0.3 + 0.2
    gave_up, surv, exact, windows = 0, 0, 0, 0
    for seed in range(40):
        att = attempts_of(1000 + seed, 400000 if cap > 1000 else 20000)
        try:
            want, cur, err = sequential(att, humans, R, v_pref, min_dist, cap)
        except IndexError:
            continue
        got, cur2, err2, st = windowed_table(att, humans, R, v_pref, min_dist, cap)
        assert cur2 == cur and err2 == err
        assert np.array_equal(np.array(got), np.array(want))
        gave_up += err
        surv, exact, windows = surv + st['survivors'], exact + st['exact'], windows + st['windows']
    if cap <= 256:
        assert gave_up > 0
    if R == 4.0:   # crowded: the table leaves well under one survivor per window, and most survivors die in float32
        assert surv < windows + 2 * 40 * humans and exact <= surv
