// Seeded scenario generation on device — replaces, per env,
//   np.random.seed(offset + case)                        /root/reference crowd_sim/envs/crowd_sim.py:276
//   CrowdSim.generate_random_human_position(rule)        crowd_sim.py:84-153
//   generate_circle_crossing_human / generate_square_... crowd_sim.py:155-207
//   Agent.sample_random_attributes                       crowd_sim/envs/utils/agent.py:39-45
// numpy's legacy RandomState is MT19937 seeded by init_genrand; random() takes two 32-bit draws
// (SURVEY.md Appendix C).  One lane generates one env; the 624-word generator state of lane-owner `slot`
// lives in HBM as key[i * stride + slot], so the lanes of a wave touch consecutive words.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cn {

struct Mt19937 {
    uint32_t* key;  // key[i * stride]
    int stride;
    int pos;  // next word to regenerate, 0..623

    // init_genrand(s); numpy then regenerates the whole block before the first draw, which the
    // word-at-a-time update below reproduces exactly (word i only needs words i, i+1, i+397 mod 624).
    __device__ void seed(uint32_t s) {
        for (int i = 0; i < 624; ++i) {
            key[(size_t)i * stride] = s;
            s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)i + 1u;
        }
        pos = 0;
    }
    __device__ uint32_t next32() {
        const int i = pos;
        const int i1 = (i + 1 == 624) ? 0 : i + 1;
        const int im = (i + 397 >= 624) ? i + 397 - 624 : i + 397;
        const uint32_t y = (key[(size_t)i * stride] & 0x80000000u) | (key[(size_t)i1 * stride] & 0x7fffffffu);
        uint32_t v = key[(size_t)im * stride] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        key[(size_t)i * stride] = v;
        pos = i1;
        v ^= (v >> 11);
        v ^= (v << 7) & 0x9d2c5680u;
        v ^= (v << 15) & 0xefc60000u;
        v ^= (v >> 18);
        return v;
    }
    // np.random.random(): 53-bit double from two draws
    __device__ double random() {
        const uint32_t a = next32() >> 5;
        const uint32_t b = next32() >> 6;
        return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
    }
    __device__ double uniform(double lo, double hi) { return lo + (hi - lo) * random(); }
    __device__ bool dead() const { return false; }
};

// Register-only generator for the FIRST 227 words of a freshly seeded stream: output i of the first block is
//   temper(key[i + 397] ^ twist(key[i], key[i + 1]))  with key[] straight from init_genrand,
// so two running copies of the seeding recurrence (at i and at i + 397) produce it without any memory.  Enough for
// 113 np.random.random() calls — nearly every H <= 8 scenario (23 calls on average at H = 5); a stream that needs
// more reports dead() and the caller regenerates the scenario with the memory-backed generator.
struct Mt19937Head {
    uint32_t lo0, lo1, hi;  // key[i], key[i + 1], key[i + 397]
    int i;
    bool exhausted;

    __device__ static uint32_t advance(uint32_t s, uint32_t index) { return 1812433253u * (s ^ (s >> 30)) + index; }
    __device__ void seed(uint32_t s) {
        lo0 = s;
        lo1 = advance(s, 1u);
        uint32_t k = lo1;
        for (uint32_t j = 2; j <= 397; ++j) k = advance(k, j);
        hi = k;
        i = 0;
        exhausted = false;
    }
    __device__ uint32_t next32() {
        if (i >= 227) {
            exhausted = true;
            return 0u;
        }
        const uint32_t y = (lo0 & 0x80000000u) | (lo1 & 0x7fffffffu);
        uint32_t v = hi ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        lo0 = lo1;
        lo1 = advance(lo1, (uint32_t)i + 2u);
        hi = advance(hi, (uint32_t)i + 398u);  // key[624] is never used: i stops at 226
        ++i;
        v ^= (v >> 11);
        v ^= (v << 7) & 0x9d2c5680u;
        v ^= (v << 15) & 0xefc60000u;
        v ^= (v >> 18);
        return v;
    }
    __device__ double random() {
        const uint32_t a = next32() >> 5;
        const uint32_t b = next32() >> 6;
        return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
    }
    __device__ double uniform(double lo, double hi_) { return lo + (hi_ - lo) * random(); }
    __device__ bool dead() const { return exhausted; }
};

// numpy's 2-vector norm on the reference image: sqrt(fma(y, y, x*x))
__device__ __forceinline__ double norm2(double x, double y) { return sqrt(__builtin_fma(y, y, x * x)); }

// `norm2(dx, dy) < min_dist` — the reference's rejection test, np.linalg.norm((dx, dy)) < min_dist — decided without the
// square root except in a band of relative width 2^-49 around equality: with s = fma(dy, dy, dx * dx) (the very argument of
// the reference's sqrt) and m2 = min_dist^2, s < m2 (1 - 2^-50) implies sqrt(s) (1 + 2^-53) < min_dist and
// s > m2 (1 + 2^-50) implies sqrt(s) (1 - 2^-53) > min_dist, so correctly rounded sqrt(s) compares the same way; inside the
// band the exact expression is evaluated.  The rejection loops of crowded scenarios (20 humans on the 4 m circle: 28 k draws
// per scenario, each tested against up to 19 x 2 placed points) spend most of their time in these tests.
__device__ __forceinline__ bool closer_than(double dx, double dy, double min_dist) {
    const double sq = __builtin_fma(dy, dy, dx * dx);
    const double m2 = min_dist * min_dist;
    if (sq < m2 * (1.0 - 0x1p-50)) return true;
    if (sq > m2 * (1.0 + 0x1p-50)) return false;
    return sqrt(sq) < min_dist;
}

struct ScenarioCfg {
    int num_agents;  // A
    int rule;        // 0 circle_crossing, 1 square_crossing, 2 mixed
    int randomize;
    double circle_radius, square_width, discomfort_dist;
    double human_radius, human_v_pref, robot_radius, robot_v_pref;
    // Rejection sampling has no termination guarantee in the reference either (a crowded circle can be unsatisfiable);
    // after this many attempts for one human the last candidate is taken and *error is set (cn_sync reports it).
    unsigned long long max_attempts;
    int* error;
};

// `mixed` (crowd_sim.py:103-151) draws the number of humans per episode (static obstacles 0..5, or 1..5 moving
// humans).  The engine keeps a fixed A agents per env: humans beyond the drawn count are PARKED — at rest, goal =
// position, far outside every neighbour range (10 m) and 100 m apart from each other — so no kernel needs a mask; they
// never interact with anything.  cn_get_human_count reports how many are present.
constexpr double kParkedX = 1.0e6;
__device__ __forceinline__ bool is_parked(double2 p) { return p.x >= 0.5 * kParkedX; }

// Writes agents [0, A) of one env into the SoA state (double2 planes indexed base + agent; vel may be NULL:
// every agent starts at rest) and returns the number of np.random.random() calls consumed.
template <class Rng>
__device__ inline uint64_t generate_scenario(const ScenarioCfg& c, Rng& rng, uint32_t seed, size_t base,
                                             double2* pos, double2* vel, double2* goal, double2* rv) {
    const double kPi = 3.141592653589793;
    rng.seed(seed);
    uint64_t draws = 0;
    const int A = c.num_agents;
    const double R = c.circle_radius;
    pos[base] = make_double2(0.0, -R);
    goal[base] = make_double2(0.0, R);
    if (vel) vel[base] = make_double2(0.0, 0.0);
    rv[base] = make_double2(c.robot_radius, c.robot_v_pref);

    // mixed: how many humans, and are they static obstacles? (crowd_sim.py:103-115)
    int present = A - 1;
    bool obstacles = false, dummy = false;
    if (c.rule == 2) {
        obstacles = rng.random() < 0.2;
        double prob = rng.random();
        draws += 2;
        const double p_static[6] = {0.05, 0.2, 0.2, 0.3, 0.1, 0.15};  // human_num 0..5
        const double p_dynamic[5] = {0.3, 0.3, 0.2, 0.1, 0.1};        // human_num 1..5
        const int keys = obstacles ? 6 : 5;
        for (int k = 0; k < keys; ++k) {  // sorted(dict.items()); no key selected (rounding) keeps the configured count
            const double value = obstacles ? p_static[k] : p_dynamic[k];
            if (prob - value <= 0) {
                present = obstacles ? k : k + 1;
                break;
            }
            prob -= value;
        }
        if (present > A - 1) present = A - 1;
        dummy = obstacles && present == 0;  // one placeholder human at (0, -10) (:121-124)
    }

    for (int i = 1; i < A; ++i) {
        double radius = c.human_radius, v_pref = c.human_v_pref;
        double x, y, tx, ty;
        unsigned long long attempts = 0;
        // 0 circle crossing, 1 square crossing, 2 static obstacle, 3 fixed placement (placeholder / parked)
        int kind = c.rule;
        if (c.rule == 2) kind = (i > present) ? 3 : (obstacles ? 2 : (i <= 2 ? 0 : 1));
        if (kind == 3) {
            const bool placeholder = dummy && i == 1;
            x = tx = placeholder ? 0.0 : kParkedX + 100.0 * i;
            y = ty = placeholder ? -10.0 : kParkedX;
        } else if (kind == 2) {  // crowd_sim.py:125-141: a square of width 4, height 8; goal = position
            const double sign = (rng.random() > 0.5) ? -1.0 : 1.0;
            draws += 1;
            for (;;) {
                x = rng.random() * 4 * 0.5 * sign;
                y = (rng.random() - 0.5) * 8;
                draws += 2;
                bool collide = false;
                for (int k = 0; k < i; ++k) {
                    const double2 p = pos[base + k];
                    if (closer_than(x - p.x, y - p.y, radius + rv[base + k].x + c.discomfort_dist)) {
                        collide = true;
                        break;
                    }
                }
                if (!collide) break;
                if (rng.dead()) return draws;
                if (++attempts >= c.max_attempts) {
                    *c.error = 1;
                    break;
                }
            }
            tx = x;
            ty = y;
        } else {
            if (c.randomize) {  // sample_random_attributes (agent.py:39-45), inside generate_*_crossing_human
                v_pref = rng.uniform(0.5, 1.5);
                radius = rng.uniform(0.3, 0.5);
                draws += 2;
            }
            if (kind == 0) {
                for (;;) {
                    const double angle = rng.random() * kPi * 2;
                    const double nx = (rng.random() - 0.5) * v_pref;
                    const double ny = (rng.random() - 0.5) * v_pref;
                    draws += 3;
                    x = R * cos(angle) + nx;
                    y = R * sin(angle) + ny;
                    bool collide = false;
                    for (int k = 0; k < i; ++k) {
                        const double2 p = pos[base + k], g = goal[base + k];
                        const double min_dist = radius + rv[base + k].x + c.discomfort_dist;
                        if (closer_than(x - p.x, y - p.y, min_dist) || closer_than(x - g.x, y - g.y, min_dist)) {
                            collide = true;
                            break;
                        }
                    }
                    if (!collide) break;
                    if (rng.dead()) return draws;
                    if (++attempts >= c.max_attempts) {
                        *c.error = 1;
                        break;
                    }
                }
                tx = -x;
                ty = -y;
            } else {
                const double w = c.square_width;
                const double sign = (rng.random() > 0.5) ? -1.0 : 1.0;
                draws += 1;
                for (;;) {
                    x = rng.random() * w * 0.5 * sign;
                    y = (rng.random() - 0.5) * w;
                    draws += 2;
                    bool collide = false;
                    for (int k = 0; k < i; ++k) {
                        const double2 p = pos[base + k];
                        if (closer_than(x - p.x, y - p.y, radius + rv[base + k].x + c.discomfort_dist)) {
                            collide = true;
                            break;
                        }
                    }
                    if (!collide) break;
                    if (rng.dead()) return draws;
                    if (++attempts >= c.max_attempts) {
                        *c.error = 1;
                        break;
                    }
                }
                for (;;) {
                    tx = rng.random() * w * 0.5 * -sign;
                    ty = (rng.random() - 0.5) * w;
                    draws += 2;
                    bool collide = false;
                    for (int k = 0; k < i; ++k) {
                        const double2 g = goal[base + k];
                        if (closer_than(tx - g.x, ty - g.y, radius + rv[base + k].x + c.discomfort_dist)) {
                            collide = true;
                            break;
                        }
                    }
                    if (!collide) break;
                    if (rng.dead()) return draws;
                    if (++attempts >= c.max_attempts) {
                        *c.error = 1;
                        break;
                    }
                }
            }
        }
        pos[base + i] = make_double2(x, y);
        goal[base + i] = make_double2(tx, ty);
        if (vel) vel[base + i] = make_double2(0.0, 0.0);
        rv[base + i] = make_double2(radius, v_pref);
    }
    return draws;
}

}  // namespace cn
