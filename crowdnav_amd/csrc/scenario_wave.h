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
    uint32_t* key;   // [624] generator state (LDS)
    uint32_t* prev;  // [2][624] the states one and two blocks earlier, or NULL (kept only when the stream is handed on)
    uint32_t* out;   // [kWindow] tempered outputs, ring indexed by stream position % kWindow (LDS)
    uint32_t produced;  // stream words generated so far (wave-uniform)
    uint32_t cursor;    // next unread stream word (wave-uniform)
    static constexpr uint32_t kWindow = 1248;

    __device__ void seed(uint32_t s, int lane) {
        if (lane == 0) {
            for (int i = 0; i < 624; ++i) {
                key[i] = s;
                s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)i + 1u;
            }
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

    // Regenerate the 624-word block in place, in the three dependency-free ranges of the recurrence
    // new[i] = new_or_old[i + 397 mod 624] ^ twist(old[i], old[i + 1]), and append its tempered words to the window.
    __device__ void next_block(int lane) {
        if (prev) {
            for (int i = lane; i < 624; i += 64) {
                prev[624 + i] = prev[i];
                prev[i] = key[i];
            }
            wave_sync();
        }
        for (int ph = 0; ph < 3; ++ph) {
            const int lo = ph == 0 ? 0 : (ph == 1 ? 227 : 454);
            const int hi = ph == 0 ? 227 : (ph == 1 ? 454 : 623);
            uint32_t v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = lo + lane + 64 * k;
                if (i < hi) {
                    const int im = (i + 397 >= 624) ? i + 397 - 624 : i + 397;
                    v[k] = key[im] ^ twist(key[i], key[i + 1]);
                }
            }
            wave_sync();
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = lo + lane + 64 * k;
                if (i < hi) key[i] = v[k];
            }
            wave_sync();
        }
        if (lane == 0) key[623] = key[396] ^ twist(key[623], key[0]);
        wave_sync();
        for (int i = lane; i < 624; i += 64) out[(produced + i) % kWindow] = temper(key[i]);
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
    __device__ uint32_t word(uint32_t stream_index) const { return out[stream_index % kWindow]; }
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
        const uint32_t* newer = in_latest ? key : prev;
        const uint32_t* older = drained ? key : (in_latest ? prev : prev + 624);
        for (int i = lane; i < 624; i += 64) dst[(size_t)i * stride] = i < pos ? newer[i] : older[i];
        if (lane == 0) *pos_out = pos;
    }
};

// Shared scratch of one generator wave.  KEEP: the stream is handed on when the scenario is done (cn_reset: the env's own
// numpy stream continues behind it), which takes the generator states of the two blocks before the current one; the ring
// fill and rollout-begin kernels never do that, and 5 KB less per workgroup is 14 generator waves per CU instead of 9.
template <bool KEEP>
struct WaveScratchT {
    uint32_t key[624];
    uint32_t prev[KEEP ? 2 * 624 : 2];
    uint32_t out[WaveRng::kWindow];
    double2 ppos[64];
    double2 pgoal[64];
    double prad[64];
    float4 fpg[64];   // float32 copies (pos.x, pos.y, goal.x, goal.y) of the placed agents: the conservative prefilter
    float fthr2[64];  // ... and, per placed agent, (min_dist - margin)^2 for the human being placed
    double2 giveup;   // window path: the attempt the generator falls back on if a human exhausts its attempts
};
using WaveScratch = WaveScratchT<true>;
using WaveScratchFill = WaveScratchT<false>;

// Conservative float32 prefilter of the circle-crossing rejection test (crowd_sim.py:159-175).  On the reference's own
// geometry (20 humans on the 4 m circle) the last humans of a scenario are accepted once in 10^3..10^4 attempts: almost every
// attempt lies DEEP inside some placed agent's exclusion disc, and deciding that does not need the float64 cos / sin and the
// up to 38 exact float64 distance tests of the reference arithmetic.  An attempt is rejected here only if its float32
// position is closer than min_dist - margin to a placed position or goal; the float32 position is within
// 2e-6 (R + 2) m of the float64 one (angle: 27 of 53 random bits, rounded to 24: 5e-7 rad; cosf / sinf: 2 ulp; the sums
// and differences: a few ulp of R + 1), and margin = 2e-5 (R + 2) m is ten times that — so every attempt rejected here is
// rejected by the exact test, and a pass of 64 attempts without a survivor moves the cursor on exactly as the exact path
// would.  Survivors (and the give-up rule) go through the exact path unchanged: same stream, same decisions, same bits.
__device__ __forceinline__ float prefilter_margin(double R) { return 2.0e-5f * ((float)R + 2.0f); }

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
    WaveRng rng{s.key, mt_key_out ? s.prev : nullptr, s.out, 0, 0};
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
    bool win = false, w_collide = true;
    int wstart = 0;
    double w_x = 0.0, w_y = 0.0;
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
            if (lane == i - 1) {  // the prefilter's threshold against the agent placed last (same for every later human)
                const float t = (float)(radius + s.prad[lane] + c.discomfort_dist) - margin;
                s.fthr2[lane] = t > 0.0f ? t * t : 0.0f;
            }
            wave_sync();
            for (;;) {
                if (!win) {  // evaluate the 64 attempts at the cursor against everybody placed so far
                    rng.ensure(6 * 64, lane);
                    const uint32_t at = rng.cursor + 6u * lane;
                    bool inside = false;
                    {
                        const float frac = (float)(rng.word(at) >> 5) * 0x1p-27f;
                        float sn, cs;
                        sincosf(frac * 6.2831855f, &sn, &cs);
                        const float fx = Rf * cs + ((float)(rng.word(at + 2) >> 5) * 0x1p-27f - 0.5f) * vpf;
                        const float fy = Rf * sn + ((float)(rng.word(at + 4) >> 5) * 0x1p-27f - 0.5f) * vpf;
                        for (int k = 0; k < i; ++k) {
                            const float4 q = s.fpg[k];
                            const float t2 = s.fthr2[k];
                            const float ax = fx - q.x, ay = fy - q.y, bx = fx - q.z, by = fy - q.w;
                            inside = inside | (ax * ax + ay * ay < t2) | (bx * bx + by * by < t2);
                        }
                    }
                    w_collide = true;
                    if (__ballot(!inside) != 0ull) {  // somebody may be free: the reference arithmetic
                        const double angle = rng.random_at(at) * kPi * 2;
                        const double nx = (rng.random_at(at + 2) - 0.5) * v_pref;
                        const double ny = (rng.random_at(at + 4) - 0.5) * v_pref;
                        w_x = R * cos(angle) + nx;
                        w_y = R * sin(angle) + ny;
                        w_collide = inside;
                        if (!inside)
                            for (int k = 0; k < i; ++k) {
                                const double2 p = s.ppos[k], g = s.pgoal[k];
                                const double min_dist = radius + s.prad[k] + c.discomfort_dist;
                                if (closer_than(w_x - p.x, w_y - p.y, min_dist) || closer_than(w_x - g.x, w_y - g.y, min_dist)) {
                                    w_collide = true;
                                    break;
                                }
                            }
                    }
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
                const unsigned long long ok = __ballot(mine && !w_collide);
                if (ok) {
                    const int first = __ffsll((long long)ok) - 1;
                    if (lane == first) {
                        s.ppos[i] = make_double2(w_x, w_y);
                        s.pgoal[i] = make_double2(-w_x, -w_y);
                        s.fpg[i] = make_float4((float)w_x, (float)w_y, (float)-w_x, (float)-w_y);
                        s.prad[i] = radius;
                    }
                    wave_sync();
                    // the lanes behind the accepted attempt are the next human's first attempts: what they still have to clear
                    // is this human (same expression as the loop above, with prad[i] = radius)
                    if (lane > first && !w_collide) {
                        const double2 p = s.ppos[i], g = s.pgoal[i];
                        const double min_dist = radius + s.prad[i] + c.discomfort_dist;
                        w_collide = closer_than(w_x - p.x, w_y - p.y, min_dist) || closer_than(w_x - g.x, w_y - g.y, min_dist);
                    }
                    wstart = first + 1;
                    if (wstart == 64) rng.cursor += 6u * 64, win = false, wstart = 0;
                    break;
                }
                const unsigned long long span = (unsigned long long)(64 - wstart);
                tried += span < left ? span : left;
                if (tried >= N) {  // give up like nobody would: take the first candidate of the last pass of 64, flag it
                    wave_sync();
                    if (lane == 0) {
                        const double x = s.giveup.x, y = s.giveup.y;
                        s.ppos[i] = make_double2(x, y);
                        s.pgoal[i] = make_double2(-x, -y);
                        s.fpg[i] = make_float4((float)x, (float)y, (float)-x, (float)-y);
                        s.prad[i] = radius;
                        *c.error = 1;
                    }
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
