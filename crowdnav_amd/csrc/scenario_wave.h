// Wave-cooperative seeded scenario generation: ONE wave64 builds ONE scenario.
//
// Same stream, same arithmetic and same accept/reject decisions as scenario_device.h (and therefore as
// crowd_sim.py:155-207 on numpy's MT19937), but the reference's rejection sampling is evaluated 64 attempts at a
// time: the tempered generator output is kept in an LDS window, every lane tests the attempt that starts at its own
// stream offset, and the first accepted attempt (in stream order) wins — the words of the later attempts are simply
// read again by the next human.  This is what tames the heavy tail of the reference's own defaults at H = 20
// (circle radius 4: 28 k draws per scenario on average, 1.8 M worst case): a lane-serial
// generator would hold its whole wave for the worst lane.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "orca_device.h"
#include "scenario_device.h"

namespace cn {

// A generator workgroup is ONE wave: its LDS instructions execute in order, so the phases below only need the compiler to keep
// program order between one lane's write and another lane's read — not s_barrier with the LDS queue drained in front of it
// (8 per 624-word block of the generator).
__device__ __forceinline__ void wave_sync() { wave_lds_sync(); }

// Window reuse (circle crossing, fixed attributes): every human's attempts look the same — the stream continues right behind
// the accepted attempt, same radius, same noise scale — so the 64 attempts a wave has evaluated are not thrown away when one of
// them is accepted: the lanes behind it ARE the next human's first attempts, and all they still have to clear is the human just
// placed (two exact distance tests).  An easy scenario (20 humans on the 12 m circle: one or two candidates per human) costs two
// or three window evaluations instead of twenty.

struct WaveRng {
    // RAW generator words of the latest two blocks: block n lives at ring + (n & 1) * 624, i.e. stream word p at ring[p % 1248];
    // the seeded state is "block -1".  A block is generated OUT OF PLACE from the one before it (no read-after-write hazard
    // inside a phase) and nothing is tempered until it is read: the conservative stages of the rejection test read three of an
    // attempt's six words, the exact arithmetic reads all six of a handful of attempts.
    uint32_t* ring;   // [2][624] (LDS)
    uint32_t* prev2;  // [624] the block two generations back, or NULL (kept only when the stream is handed on: persist)
    uint32_t produced;  // stream words generated so far (wave-uniform)
    uint32_t cursor;    // next unread stream word (wave-uniform)
    static constexpr uint32_t kWindow = 1248;

    // words base + J .. base + 63 of init_genrand into lanes J .. 63 of v (v_writelane_b32: the lane select is an inline constant;
    // this clang has no writelane builtin, and two SGPR operands would break the constant-bus rule)
    template <int J>
    __device__ static __forceinline__ void seed_chunk(uint32_t& s, uint32_t& v, uint32_t base) {
        asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(s), "n"(J));
        s = 1812433253u * (s ^ (s >> 30)) + base + (uint32_t)J + 1u;
        if constexpr (J + 1 < 64) seed_chunk<J + 1>(s, v, base);
    }
    __device__ void seed(uint32_t s, int lane) {
        // init_genrand: a serial chain (word i + 1 from word i).  Sixty-four words at a time are collected in one register
        // through v_writelane and stored with one LDS instruction.
        uint32_t* key = ring + 624;
        s = __builtin_amdgcn_readfirstlane(s);
        for (int base = 0; base < 624; base += 64) {  // (the last chunk runs 16 words past the state: never stored)
            uint32_t v = 0;
            seed_chunk<0>(s, v, (uint32_t)base);
            if (base + lane < 624) key[base + lane] = v;
        }
        produced = 0;
        cursor = 0;
        wave_sync();
    }

    __device__ static uint32_t twist(uint32_t a, uint32_t b) {
        const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
        return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    __device__ static uint32_t temper(uint32_t v) {
        v ^= (v >> 11);
        v ^= (v << 7) & 0x9d2c5680u;
        v ^= (v << 15) & 0xefc60000u;
        v ^= (v >> 18);
        return v;
    }

    // Block n from block n - 1: new[i] = (i < 227 ? old[i + 397] : new[i - 227]) ^ twist(old[i], old[i + 1]); new[623] takes
    // new[0].  Word i = lane + 64 k is lane's k-th word: the ten twists of a lane read only the old block (all requests up
    // front), then three dependent ranges of the recurrence.
    __device__ void next_block(int lane) {
        const uint32_t n = produced / 624u;
        uint32_t* X = ring + (n & 1u) * 624u;
        const uint32_t* O = ring + ((n + 1u) & 1u) * 624u;
        if (prev2) {
            for (int i = lane; i < 624; i += 64) prev2[i] = X[i];
            wave_sync();
        }
        uint32_t tw[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const int i = lane + 64 * k;
            tw[k] = i < 623 ? twist(O[i], O[i + 1]) : 0u;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = lane + 64 * k;
            if (i < 227) X[i] = O[i + 397] ^ tw[k];
        }
        wave_sync();
#pragma unroll
        for (int k = 3; k < 8; ++k) {
            const int i = lane + 64 * k;
            if (i >= 227 && i < 454) X[i] = X[i - 227] ^ tw[k];
        }
        wave_sync();
#pragma unroll
        for (int k = 7; k < 10; ++k) {
            const int i = lane + 64 * k;
            if (i >= 454 && i < 623) X[i] = X[i - 227] ^ tw[k];
        }
        wave_sync();
        if (lane == 0) X[623] = X[396] ^ twist(O[623], X[0]);
        produced += 624;
        wave_sync();
    }

    // make stream words [cursor, cursor + n) readable (n <= 384)
    __device__ void ensure(uint32_t n, int lane) {
        while (produced < cursor + n) next_block(lane);
    }
    // Move the cursor BACK to stream word `target` (the give-up rule of the window path: up to 62 attempts = 372 words behind
    // the window base).  The ring holds words [produced - kWindow, produced) and the window may run a block ahead of the
    // cursor, so the target can have been overwritten already; and persist() knows the generator states of the latest two
    // blocks only.  In either case the stream is regenerated from its seed up to the target's block (an error path: the
    // human that gave up has just cost max_attempts / 64 window passes).
    __device__ void rewind_to(uint32_t seed0, uint32_t target, int lane) {
        const bool held = target + kWindow >= produced;
        const bool persistable = target / 624u + 2u >= produced / 624u;
        if (!held || !persistable) {
            seed(seed0, lane);
            while (produced <= target) next_block(lane);
        }
        cursor = target;
    }
    __device__ uint32_t word(uint32_t stream_index) const { return temper(ring[stream_index % kWindow]); }
    // np.random.random() from the two words at stream_index
    __device__ double random_at(uint32_t stream_index) const {
        const uint32_t a = word(stream_index) >> 5, b = word(stream_index + 1) >> 6;
        return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
    }
    // The generator after `cursor` words in the form Mt19937 (scenario_device.h) continues from: it regenerates one word at
    // a time in place, so at position p of block kb its array holds words [0, p) of block kb and words [p, 624) of block
    // kb - 1 (the seeded state for kb = 0).  The window runs up to a block ahead of the cursor: block kb is the latest one
    // or the one before it — hence the two kept generations; everything consumed = the latest block at position 0.
    __device__ void persist(uint32_t* dst, int stride, int* pos_out, int lane) const {
        const uint32_t latest = produced / 624u;  // blocks generated
        const uint32_t kb = cursor / 624u;
        const bool drained = cursor == produced;  // also the freshly seeded generator (0 == 0)
        const int pos = drained ? 0 : (int)(cursor - kb * 624u);
        const bool in_latest = kb + 1u == latest;
        const uint32_t* key = ring + ((latest - 1u) & 1u) * 624u;  // the latest block (the seeded state when none was generated)
        const uint32_t* prev = ring + (latest & 1u) * 624u;        // the one before it
        const uint32_t* newer = in_latest ? key : prev;
        const uint32_t* older = drained ? key : (in_latest ? prev : prev2);
        for (int i = lane; i < 624; i += 64) dst[(size_t)i * stride] = i < pos ? newer[i] : older[i];
        if (lane == 0) *pos_out = pos;
    }
};

// Shared scratch of one generator wave.  KEEP: the stream is handed on when the scenario is done (cn_reset: the env's own
// numpy stream continues behind it), which takes the generator states of the two blocks before the current one; the ring
// fill and rollout-begin kernels never do that, and 5 KB less per workgroup is 14 generator waves per CU instead of 9.
// Blocked-cell table of the circle-crossing rejection test: a kGridN x kGridN bitmap over the square the attempts can fall in.
constexpr int kGridN = 160;
constexpr int kGridRowWords = kGridN / 32;
template <bool KEEP>
struct WaveScratchT {
    uint32_t ring[WaveRng::kWindow];   // raw generator words of the latest two blocks
    uint32_t prev2[KEEP ? 624 : 2];
    uint32_t grid[kGridN * kGridRowWords];
    double2 ppos[64];
    double2 pgoal[64];
    double prad[64];
    float4 fpg[64];   // float32 copies (pos.x, pos.y, goal.x, goal.y) of the placed agents: the conservative prefilter
    float fthr2[64];  // ... and, per placed agent, (min_dist - margin)^2 for the human being placed
    double2 giveup;   // window path: the attempt the generator falls back on if a human exhausts its attempts
};
using WaveScratch = WaveScratchT<true>;
using WaveScratchFill = WaveScratchT<false>;

// Conservative float32 stages of the circle-crossing rejection test (crowd_sim.py:159-175).  On the reference's own geometry
// (20 humans on the 4 m circle) the last humans of a scenario are accepted once in 10^3..10^4 attempts: almost every attempt
// lies DEEP inside some placed agent's exclusion disc, and deciding that needs neither the float64 cos / sin nor the up to 38
// exact float64 distance tests of the reference arithmetic.
//
// The float32 position of an attempt: angle fraction = the top 27 bits of the attempt's first word (of the 53 the reference
// uses), cos / sin by the hardware's V_COS_F32 / V_SIN_F32 (input in revolutions), noise from the top 27 bits of its third and
// fifth word.  Its distance from the float64 position is bounded by kTrigAbsError * R (measured exhaustively over all 2^27
// fractions: scripts/probes/trig_error.hip, tests/test_generator_trig.py) + 2 pi R 2^-27 (angle truncation) + a few ulp of R + 1.
//
// Stage 1, the BLOCKED-CELL TABLE: a bitmap over the square [-ext, ext]^2 (ext = R + v_pref / 2 + three cells); a cell's bit is
// set when the WHOLE cell lies within min_dist - table_margin of a placed position or goal — marked row by row when an agent
// is placed (one lane per grid row: the blocked x-interval of the row, conservatively rounded, OR-ed into the row's words).
// An attempt whose float32 position falls in a set cell is rejected: 0.2 % of the attempts survive at R = 4 (cell 5.7 cm).
// Stage 2: a survivor's float32 position against every placed point (one lane per placed agent), rejected if closer than
// min_dist - margin.  Stage 3: the reference arithmetic, again one lane per placed agent.  margin and table_margin are ten
// times the error bounds, so whatever stages 1 and 2 reject the exact test rejects, survivors are decided by the exact test,
// and they are examined in stream order: same stream, same decisions, same bits as the sequential loop.
constexpr float kTrigAbsError = 1.0e-5f;  // |V_COS_F32(f) - cos(2 pi f)|, same for sin, f in [0, 1): bound asserted by the probe
__device__ __forceinline__ float prefilter_margin(double R) { return 10.0f * (kTrigAbsError * (float)R + 1.0e-6f * ((float)R + 2.0f)); }

struct BlockedGrid {
    uint32_t* bits;
    float ext, inv_cell, cell;
    __device__ void init(uint32_t* g, double R, double v_pref, int lane) {
        bits = g;
        ext = (float)(R + 0.5 * v_pref) / (1.0f - 6.0f / (float)kGridN);
        cell = 2.0f * ext / (float)kGridN;
        inv_cell = (float)kGridN / (2.0f * ext);
        for (int i = lane; i < kGridN * kGridRowWords; i += 64) bits[i] = 0u;
    }
    // word index and bit of the cell a float32 position falls in
    __device__ void locate(float x, float y, int& word, uint32_t& bit) const {
        int ix = (int)((x + ext) * inv_cell), iy = (int)((y + ext) * inv_cell);
        ix = ix < 0 ? 0 : (ix > kGridN - 1 ? kGridN - 1 : ix);
        iy = iy < 0 ? 0 : (iy > kGridN - 1 ? kGridN - 1 : iy);
        word = iy * kGridRowWords + (ix >> 5);
        bit = 1u << (ix & 31);
    }
    // mark the cells that lie entirely inside the disc of radius rt around (cx, cy); `sub` of `nsub` lane groups share the rows
    __device__ void mark(float cx, float cy, float rt, int sub, int nsub) {
        const int iy0 = (int)floorf((cy - rt + ext) * inv_cell), iy1 = (int)floorf((cy + rt + ext) * inv_cell);
        const float rt2 = rt * rt;
        for (int iy = iy0 + sub; iy <= iy1; iy += nsub) {
            if (iy < 0 || iy >= kGridN) continue;
            const float y0 = (float)iy * cell - ext;
            const float dy = fmaxf(fabsf(y0 - cy), fabsf(y0 + cell - cy));
            const float w2 = rt2 - dy * dy;
            if (!(w2 > 0.0f)) continue;
            const float w = sqrtf(w2) * (1.0f - 1.0e-6f);
            int lo = (int)ceilf((cx - w + ext) * inv_cell), hi = (int)floorf((cx + w + ext) * inv_cell);  // cells [lo, hi)
            lo = lo < 0 ? 0 : lo;
            hi = hi > kGridN ? kGridN : hi;
#pragma unroll
            for (int wd = 0; wd < kGridRowWords; ++wd) {
                const int a = lo > 32 * wd ? lo : 32 * wd, b = hi < 32 * wd + 32 ? hi : 32 * wd + 32;
                if (a < b) {
                    const uint32_t m = (b - a == 32 ? 0xffffffffu : ((1u << (b - a)) - 1u)) << (a - 32 * wd);
                    atomicOr(&bits[iy * kGridRowWords + wd], m);
                }
            }
        }
    }
};
// table_margin: position error (as for prefilter_margin) + the table's own float32 rounding (cell bounds, interval ends: a few
// ulp of ext) — ten-fold
__device__ __forceinline__ float table_margin(double R) { return prefilter_margin(R) + 1.0e-4f; }

// Builds agents [0, A) at pos/vel/goal/rv[base + agent] (vel may be NULL); returns np.random.random() calls consumed.
// Must be called by all 64 lanes of a one-wave workgroup.
// mt_key_out / mt_stride / mt_pos_out (optional): where the env's own generator continues (cn_reset: epsilon-greedy draws).
template <class Scratch>
__device__ inline uint64_t generate_scenario_wave(const ScenarioCfg& c, Scratch& s, uint32_t seed, size_t base,
                                                  double2* pos, double2* vel, double2* goal, double2* rv,
                                                  uint32_t* mt_key_out = nullptr, int mt_stride = 0,
                                                  int* mt_pos_out = nullptr) {
    const double kPi = 3.141592653589793;
    const int lane = threadIdx.x & 63;
    WaveRng rng{s.ring, mt_key_out ? s.prev2 : nullptr, 0, 0};
    rng.seed(seed, lane);
    const int A = c.num_agents;
    const double R = c.circle_radius;
    const float Rf = (float)R, margin = prefilter_margin(R);
    if (lane == 0) {
        s.fpg[0] = make_float4(0.0f, -Rf, 0.0f, Rf);
        s.ppos[0] = make_double2(0.0, -R);
        s.pgoal[0] = make_double2(0.0, R);
        s.prad[0] = c.robot_radius;
        pos[base] = s.ppos[0];
        goal[base] = s.pgoal[0];
        if (vel) vel[base] = make_double2(0.0, 0.0);
        rv[base] = make_double2(c.robot_radius, c.robot_v_pref);
    }
    wave_sync();
    // window state of the window-reuse path (circle crossing, fixed attributes): lane l holds the attempt at stream position
    // rng.cursor + 6 l once `win` is set; lanes below wstart belong to humans already placed
    bool win = false, w_blocked = true;
    int wstart = 0, w_word = 0;
    uint32_t w_bit = 0u;
    float w_fx = 0.0f, w_fy = 0.0f;
    BlockedGrid grid{};
    const float tmargin = table_margin(R);
    for (int i = 1; i < A; ++i) {
        double radius = c.human_radius, v_pref = c.human_v_pref;
        if (c.randomize) {
            rng.ensure(4, lane);
            v_pref = 0.5 + (1.5 - 0.5) * rng.random_at(rng.cursor);
            radius = 0.3 + (0.5 - 0.3) * rng.random_at(rng.cursor + 2);
            rng.cursor += 4;
        }
        unsigned long long attempts = 0;
        const float vpf = (float)v_pref;
        if (c.rule == 0 && !c.randomize) {
            // the reference examines attempts until one is free (crowd_sim.py:159-175); this generator gives up after N of them,
            // counted per human in passes of 64 like the loop below (attempt N - 64 is taken then, and the error flagged)
            const unsigned long long N = ((c.max_attempts + 63ull) / 64ull) * 64ull;
            const uint32_t hstart = rng.cursor + (win ? 6u * wstart : 0u);  // stream position of this human's first attempt
            unsigned long long tried = 0;
            if (i == 1) {  // the table, with the robot's start and goal marked (lanes 0-31 / 32-63: the two discs' rows)
                grid.init(s.grid, R, v_pref, lane);
                wave_sync();
                const float rt = (float)(radius + c.robot_radius + c.discomfort_dist) - tmargin;
                if (rt > 0.0f) grid.mark(0.0f, lane < 32 ? -Rf : Rf, rt, lane & 31, 32);
            }
            if (lane == i - 1) {  // the prefilter's threshold against the agent placed last (same for every later human)
                const float t = (float)(radius + s.prad[lane] + c.discomfort_dist) - margin;
                s.fthr2[lane] = t > 0.0f ? t * t : 0.0f;
            }
            wave_sync();
            for (;;) {
                if (!win) {  // the 64 attempts at the cursor: float32 position, blocked-cell table
                    rng.ensure(6 * 64, lane);
                    const uint32_t at = (rng.cursor + 6u * lane) % WaveRng::kWindow;  // (6 | 1248: an attempt's words do not wrap)
                    const float frac = (float)(WaveRng::temper(rng.ring[at]) >> 5) * 0x1p-27f;
                    w_fx = Rf * __builtin_amdgcn_cosf(frac) + ((float)(WaveRng::temper(rng.ring[at + 2]) >> 5) * 0x1p-27f - 0.5f) * vpf;
                    w_fy = Rf * __builtin_amdgcn_sinf(frac) + ((float)(WaveRng::temper(rng.ring[at + 4]) >> 5) * 0x1p-27f - 0.5f) * vpf;
                    grid.locate(w_fx, w_fy, w_word, w_bit);
                    w_blocked = (s.grid[w_word] & w_bit) != 0u;
                    win = true, wstart = 0;
                }
                const unsigned long long left = N - tried;  // attempts this human may still examine
                const bool mine = lane >= wstart && (unsigned long long)(lane - wstart) < left;
                if (lane >= wstart && tried + (unsigned long long)(lane - wstart) == N - 64ull) {
                    // the attempt the give-up rule would take (first of the last pass of 64): remembered while its words
                    // are in the window — with the default cap (2^23 attempts) this lane exists once in a blue moon
                    const uint32_t at = rng.cursor + 6u * lane;
                    const double angle = rng.random_at(at) * kPi * 2;
                    const double nx = (rng.random_at(at + 2) - 0.5) * v_pref;
                    const double ny = (rng.random_at(at + 4) - 0.5) * v_pref;
                    s.giveup = make_double2(R * cos(angle) + nx, R * sin(angle) + ny);
                }
                // the table's survivors one by one, in stream order, every placed agent on its own lane
                unsigned long long cand = __ballot(mine && !w_blocked);
                int first = -1;
                double x = 0.0, y = 0.0;
                while (cand) {
                    const int L = __ffsll((long long)cand) - 1;
                    cand &= cand - 1ull;
                    const float sx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w_fx), L));
                    const float sy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w_fy), L));
                    bool inside = false;
                    if (lane < i) {
                        const float4 q = s.fpg[lane];
                        const float t2 = s.fthr2[lane];
                        const float ax = sx - q.x, ay = sy - q.y, bx = sx - q.z, by = sy - q.w;
                        inside = (ax * ax + ay * ay < t2) | (bx * bx + by * by < t2);
                    }
                    if (__ballot(inside) != 0ull) continue;  // deep inside somebody's disc
                    const uint32_t at = rng.cursor + 6u * (uint32_t)L;  // the reference arithmetic (wave-uniform values)
                    const double angle = rng.random_at(at) * kPi * 2;
                    const double nx = (rng.random_at(at + 2) - 0.5) * v_pref;
                    const double ny = (rng.random_at(at + 4) - 0.5) * v_pref;
                    x = R * cos(angle) + nx;
                    y = R * sin(angle) + ny;
                    bool collide = false;
                    if (lane < i) {
                        const double2 p = s.ppos[lane], g = s.pgoal[lane];
                        const double min_dist = radius + s.prad[lane] + c.discomfort_dist;
                        collide = closer_than(x - p.x, y - p.y, min_dist) || closer_than(x - g.x, y - g.y, min_dist);
                    }
                    if (__ballot(collide) != 0ull) continue;
                    first = L;
                    break;
                }
                if (first >= 0) {
                    if (lane == 0) {
                        s.ppos[i] = make_double2(x, y);
                        s.pgoal[i] = make_double2(-x, -y);
                        s.fpg[i] = make_float4((float)x, (float)y, (float)-x, (float)-y);
                        s.prad[i] = radius;
                    }
                    // the new human's two discs into the table; the lanes behind the accepted attempt are the next human's
                    // first attempts: all they still have to clear is this human — they look their cell up again
                    const float rt = (float)(radius + radius + c.discomfort_dist) - tmargin;
                    if (rt > 0.0f && i + 1 < A) grid.mark(lane < 32 ? (float)x : (float)-x, lane < 32 ? (float)y : (float)-y, rt, lane & 31, 32);
                    wave_sync();
                    w_blocked = w_blocked || (s.grid[w_word] & w_bit) != 0u;
                    wstart = first + 1;
                    if (wstart == 64) rng.cursor += 6u * 64, win = false, wstart = 0;
                    break;
                }
                const unsigned long long span = (unsigned long long)(64 - wstart);
                tried += span < left ? span : left;
                if (tried >= N) {  // give up like nobody would: take the first candidate of the last pass of 64, flag it
                    wave_sync();
                    const double gx = s.giveup.x, gy = s.giveup.y;
                    if (lane == 0) {
                        s.ppos[i] = make_double2(gx, gy);
                        s.pgoal[i] = make_double2(-gx, -gy);
                        s.fpg[i] = make_float4((float)gx, (float)gy, (float)-gx, (float)-gy);
                        s.prad[i] = radius;
                        *c.error = 1;
                    }
                    const float rt = (float)(radius + radius + c.discomfort_dist) - tmargin;
                    if (rt > 0.0f && i + 1 < A) grid.mark(lane < 32 ? (float)gx : (float)-gx, lane < 32 ? (float)gy : (float)-gy, rt, lane & 31, 32);
                    rng.rewind_to(seed, hstart + 6u * (uint32_t)(N - 64ull + 1ull), lane);
                    win = false, wstart = 0;
                    break;
                }
                rng.cursor += 6u * 64, win = false, wstart = 0;
            }
        } else if (c.rule == 0) {
            // circle crossing: attempt = (angle, px_noise, py_noise) = 6 words; reject if within min_dist of any placed
            // agent's position or goal (crowd_sim.py:159-175)
            if (lane < i) {  // this human's thresholds against everybody placed so far
                const float t = (float)(radius + s.prad[lane] + c.discomfort_dist) - margin;
                s.fthr2[lane] = t > 0.0f ? t * t : 0.0f;
            }
            wave_sync();
            for (;;) {
                rng.ensure(6 * 64, lane);
                const uint32_t at = rng.cursor + 6u * lane;
                {
                    const float frac = (float)(rng.word(at) >> 5) * 0x1p-27f;
                    float sn, cs;
                    sincosf(frac * 6.2831855f, &sn, &cs);
                    const float fx = Rf * cs + ((float)(rng.word(at + 2) >> 5) * 0x1p-27f - 0.5f) * vpf;
                    const float fy = Rf * sn + ((float)(rng.word(at + 4) >> 5) * 0x1p-27f - 0.5f) * vpf;
                    bool inside = false;
                    for (int k = 0; k < i; ++k) {
                        const float4 q = s.fpg[k];
                        const float t2 = s.fthr2[k];
                        const float ax = fx - q.x, ay = fy - q.y, bx = fx - q.z, by = fy - q.w;
                        inside = inside | (ax * ax + ay * ay < t2) | (bx * bx + by * by < t2);
                    }
                    if (__ballot(!inside) == 0ull && attempts + 64 < c.max_attempts) {  // 64 certain rejections
                        attempts += 64;
                        rng.cursor += 6u * 64;
                        continue;
                    }
                }
                const double angle = rng.random_at(at) * kPi * 2;
                const double nx = (rng.random_at(at + 2) - 0.5) * v_pref;
                const double ny = (rng.random_at(at + 4) - 0.5) * v_pref;
                const double x = R * cos(angle) + nx;
                const double y = R * sin(angle) + ny;
                bool collide = false;
                for (int k = 0; k < i; ++k) {
                    const double2 p = s.ppos[k], g = s.pgoal[k];
                    const double min_dist = radius + s.prad[k] + c.discomfort_dist;
                    if (closer_than(x - p.x, y - p.y, min_dist) || closer_than(x - g.x, y - g.y, min_dist)) {
                        collide = true;
                        break;
                    }
                }
                unsigned long long ok = __ballot(!collide);
                attempts += 64;
                if (!ok && attempts >= c.max_attempts) {  // give up like nobody would: take the first candidate, flag it
                    ok = 1;
                    if (lane == 0) *c.error = 1;
                }
                if (ok) {
                    const int first = __ffsll((long long)ok) - 1;
                    if (lane == first) {
                        s.ppos[i] = make_double2(x, y);
                        s.pgoal[i] = make_double2(-x, -y);
                        s.fpg[i] = make_float4((float)x, (float)y, (float)-x, (float)-y);
                    }
                    rng.cursor += 6u * (first + 1);
                    break;
                }
                rng.cursor += 6u * 64;
            }
        } else {
            // square crossing (crowd_sim.py:181-205): sign, then start attempts (4 words), then goal attempts (4 words)
            rng.ensure(2, lane);
            const double sign = (rng.random_at(rng.cursor) > 0.5) ? -1.0 : 1.0;
            rng.cursor += 2;
            const double w = c.square_width;
            for (int pass = 0; pass < 2; ++pass) {
                for (;;) {
                    rng.ensure(4 * 64, lane);
                    const uint32_t at = rng.cursor + 4u * lane;
                    const double x = rng.random_at(at) * w * 0.5 * (pass == 0 ? sign : -sign);
                    const double y = (rng.random_at(at + 2) - 0.5) * w;
                    bool collide = false;
                    for (int k = 0; k < i; ++k) {
                        const double2 q = pass == 0 ? s.ppos[k] : s.pgoal[k];
                        if (closer_than(x - q.x, y - q.y, radius + s.prad[k] + c.discomfort_dist)) {
                            collide = true;
                            break;
                        }
                    }
                    unsigned long long ok = __ballot(!collide);
                    attempts += 64;
                    if (!ok && attempts >= c.max_attempts) {
                        ok = 1;
                        if (lane == 0) *c.error = 1;
                    }
                    if (ok) {
                        const int first = __ffsll((long long)ok) - 1;
                        if (lane == first) {
                            if (pass == 0) {
                                s.ppos[i] = make_double2(x, y);
                            } else {
                                s.pgoal[i] = make_double2(x, y);
                            }
                        }
                        rng.cursor += 4u * (first + 1);
                        break;
                    }
                    rng.cursor += 4u * 64;
                }
                wave_sync();
            }
        }
        if (lane == 0) s.prad[i] = radius;
        wave_sync();
        if (lane == 0) {
            pos[base + i] = s.ppos[i];
            goal[base + i] = s.pgoal[i];
            if (vel) vel[base + i] = make_double2(0.0, 0.0);
            rv[base + i] = make_double2(radius, v_pref);
        }
    }
    wave_sync();
    if (win) rng.cursor += 6u * wstart;  // (window path: the stream stands behind the last accepted attempt)
    if (mt_key_out) rng.persist(mt_key_out, mt_stride, mt_pos_out, lane);
    return (uint64_t)(rng.cursor / 2);
}

}  // namespace cn
