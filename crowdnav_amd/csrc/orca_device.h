// ORCA (optimal reciprocal collision avoidance) agent-agent solve for gfx950.
//
// Replaces what `sim.doStep(); sim.getAgentVelocity(0)` does for the calling agent in
//   /root/reference crowd_sim/envs/policy/orca.py:128-129
// i.e. the un-vendored RVO2 library's neighbour selection, half-plane construction and the three
// incremental linear programs.  All arithmetic is IEEE binary32 with NO
// contraction (the translation unit is built with -ffp-contract=off), correctly rounded / and sqrt
// (hipcc default), vector / scalar = multiply by the reciprocal.
//
// Work decomposition inside one wave (step_kernels.h, rollout_fused.h): half-planes are built by one lane per (agent,
// candidate neighbour) pair and parked in LDS as one float4 {point.x, point.y, dir.x, dir.y} per slot.  Crowds of up to 5
// neighbours solve in CANDIDATE FORM (1-D solutions of all (agent, half-plane) pairs at once, a short scan per agent; the 3-D
// fallback likewise on (agent, i, j) lanes); 10-half-plane programs run on three lanes per agent (lp_planar_tri) with the
// fallback in lazily evaluated candidate form (lp_relaxed_lazy) or in shuffle rounds (lp_relaxed_coop).  What was tried and
// rejected on the way (register-resident unrolled programs, all-lane cooperative planar programs, serial LDS walks) is
// profiles/HISTORY.md.
#pragma once
#include <hip/hip_runtime.h>

namespace cn {

constexpr int kWave = 64;      // gfx950 wavefront width; one workgroup = one wave here
constexpr int kMaxNb = 10;     // RVO2 maxNeighbors supported (the reference hard-codes 10, orca.py:62)
constexpr int kLineStride = kMaxNb + 1;  // float4 slots per agent in LDS (+1 pad: conflict-free b128 reads)
constexpr float kRvoEps = 0.00001f;
constexpr int kLazyCandFloat4 = (kWave / (kMaxNb - 1)) * ((kMaxNb - 1) + 4);  // lp_relaxed_lazy<10>'s scratch per wave: 7 x (9 + 4) float4

// "Barrier" of a ONE-wave workgroup (the fused rollout, the 20-human shard's rollout, a scenario generator wave): a wave's
// LDS instructions execute in order, so between one lane's write and another lane's read no s_barrier — and none of the
// s_waitcnt lgkmcnt(0) hipcc puts in front of one — is needed; what IS needed is that the COMPILER keeps the accesses in
// program order.  wave_barrier alone is a scheduling barrier with no memory semantics (IntrNoMem): the two wavefront-scope
// fences make it a release / acquire pair at IR level, so no LDS access may be moved or forwarded across it whatever the
// optimiser can prove about per-thread aliasing; at wavefront scope they emit no instruction on gfx9.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// threadIdx.x as a value the optimiser cannot see through: everything a phase derives from it (lane group, slot, LDS row
// addresses) is recomputed where the phase runs instead of being hoisted out of the rollout kernels' step loop and held in
// VGPRs across it (the 20-human shard's kernel needs every register it can get for its third resident wave).
__device__ __forceinline__ int opaque_tid() {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}

struct OrcaParams {
    float neighbor_dist;
    float inv_time_horizon;  // 1.0f / timeHorizon
    float inv_time_step;     // 1.0f / timeStep
    int max_neighbors;
};

// Build one ORCA half-plane (Appendix A.3): self (sp, sv), other (op, ov), rsum = combined padded radius.
// The three geometric cases (cut-off disc, left/right leg, already overlapping) are folded into ONE
// sqrt + ONE reciprocal with selected operands, so that lanes of a wave in different cases do not serialise
// three divergent branches.  Every selected value is produced by exactly the operations RVO2 performs for
// that case, so the result is bit-identical to the branchy form.
__device__ __forceinline__ float4 make_half_plane(const OrcaParams& P, float spx, float spy, float svx,
                                                  float svy, float opx, float opy, float ovx, float ovy,
                                                  float rsum) {
    const float rpx = opx - spx, rpy = opy - spy;
    const float rvx = svx - ovx, rvy = svy - ovy;
    const float dist_sq = rpx * rpx + rpy * rpy;
    const float rsum_sq = rsum * rsum;
    const bool overlap = !(dist_sq > rsum_sq);  // already colliding: resolve within one time step
    const float inv_t = overlap ? P.inv_time_step : P.inv_time_horizon;
    const float wx = rvx - inv_t * rpx;
    const float wy = rvy - inv_t * rpy;
    const float wlen_sq = wx * wx + wy * wy;
    const float dot1 = wx * rpx + wy * rpy;
    // closest point of the velocity obstacle on its cut-off disc (always so when overlapping), else on a leg
    const bool disc = overlap || (dot1 < 0.0f && dot1 * dot1 > rsum_sq * wlen_sq);
    const float root = sqrtf(disc ? wlen_sq : dist_sq - rsum_sq);  // |w|  or  leg length
    const float inv = 1.0f / (disc ? root : dist_sq);
    // disc case
    const float nx = wx * inv, ny = wy * inv;
    const float kd = rsum * inv_t - root;
    // leg case
    const bool left = rpx * wy - rpy * wx > 0.0f;
    const float llx = (rpx * root - rpy * rsum) * inv, lly = (rpx * rsum + rpy * root) * inv;
    const float lrx = -((rpx * root + rpy * rsum) * inv), lry = -((-rpx * rsum + rpy * root) * inv);
    const float gx = left ? llx : lrx, gy = left ? lly : lry;
    const float dot2 = rvx * gx + rvy * gy;
    const float ldx = disc ? ny : gx;
    const float ldy = disc ? -nx : gy;
    const float ux = disc ? kd * nx : dot2 * gx - rvx;
    const float uy = disc ? kd * ny : dot2 * gy - rvy;
    return make_float4(svx + 0.5f * ux, svy + 0.5f * uy, ldx, ldy);
}

// Start point of the 2-D program when optimising towards a point (Appendix A.4, directionOpt = false).
__device__ __forceinline__ void lp_start_point(float radius, float ox, float oy, float& rx, float& ry) {
    if (ox * ox + oy * oy > radius * radius) {
        const float inv = 1.0f / sqrtf(ox * ox + oy * oy);
        const float ux = ox * inv, uy = oy * inv;
        rx = ux * radius;
        ry = uy * radius;
    } else {
        rx = ox;
        ry = oy;
    }
}

// ------------------------------------------------------------------ candidate form of the programs
// RVO2's linearProgram1 on half-plane k (Appendix A.5) reads the half-planes 0..k, the speed disc and the optimisation
// target — NOT the running result of linearProgram2: the running result only decides WHETHER half-plane k is violated, i.e.
// whether the 1-D solution replaces it.  So the 1-D solutions ("candidates") of all half-planes of all agents can be
// computed at once, one lane per (agent, half-plane), with the divisions of one candidate independent of each other; what
// remains serial per agent is a scan of MAXL compare-and-select steps (lp_planar_scan).  Same operations on the same
// operands in the same order as RVO2's own loop, so the results are bit-identical; a half-plane the sequential program
// never reaches merely has an unused candidate.
//   lk: half-plane k; prev: half-planes 0.. of the same program in LDS (NJ slots are read, j >= k ignored)
//   returns (candidate.x, candidate.y, feasible ? 1 : 0, -)
template <int NJ>
__device__ __forceinline__ float4 lp_line_candidate(const float4 lk, const float4* prev, int k, float radius, float ox,
                                                    float oy, bool dir_opt) {
    const float px = lk.x, py = lk.y, dx = lk.z, dy = lk.w;
    const float dp = px * dx + py * dy;
    const float disc = (dp * dp + radius * radius) - (px * px + py * py);
    bool ok = !(disc < 0.0f);
    const float root = sqrtf(disc);
    float t_lo = -dp - root;
    float t_hi = -dp + root;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const float4 lj = prev[j];
        const bool use = j < k;
        const float den = dx * lj.w - dy * lj.z;
        const float num = lj.z * (py - lj.y) - lj.w * (px - lj.x);
        const bool parallel = fabsf(den) <= kRvoEps;
        const float t = num / den;
        ok = ok && !(use && parallel && num < 0.0f);
        const bool upper = use && !parallel && den >= 0.0f;
        const bool lower = use && !parallel && !(den >= 0.0f);
        t_hi = (upper && t < t_hi) ? t : t_hi;
        t_lo = (lower && t_lo < t) ? t : t_lo;
        ok = ok && !(t_lo > t_hi);
    }
    float t;
    if (dir_opt) {
        t = (ox * dx + oy * dy > 0.0f) ? t_hi : t_lo;
    } else {
        t = dx * (ox - px) + dy * (oy - py);
        t = (t < t_lo) ? t_lo : ((t > t_hi) ? t_hi : t);
    }
    return make_float4(px + t * dx, py + t * dy, ok ? 1.0f : 0.0f, 0.0f);
}

// linearProgram2 (Appendix A.4) as a scan over precomputed candidates: lines / cand = the agent's MAXL slots in LDS.
// Returns the first infeasible half-plane or n; (rx, ry) enters as the start point.
template <int MAXL>
__device__ __forceinline__ int lp_planar_scan(const float4* lines, const float4* cand, int n, float& rx, float& ry) {
    // all LDS requests first, then selects only: a short-circuit `k < fail && ...` makes hipcc branch around the loads and
    // pay one LDS round trip per half-plane
    float4 lk[MAXL], ck[MAXL];
#pragma unroll
    for (int k = 0; k < MAXL; ++k) {
        lk[k] = lines[k];
        ck[k] = cand[k];
    }
    int fail = n;
#pragma unroll
    for (int k = 0; k < MAXL; ++k) {
        const float det = lk[k].z * (lk[k].y - ry) - lk[k].w * (lk[k].x - rx);
        const bool viol = (k < fail) & (det > 0.0f);
        const bool feasible = ck[k].z != 0.0f;
        rx = (viol & feasible) ? ck[k].x : rx;
        ry = (viol & feasible) ? ck[k].y : ry;
        fail = (viol & !feasible) ? k : fail;
    }
    return fail;
}

// linearProgram3 (Appendix A.6) in the same form, for MAXL = 5: the projection of half-plane j onto half-plane i (j < i)
// depends on nothing but the two half-planes, and the 1-D solution of projected half-plane k of program i on nothing but
// the projections 0..k of that program — 10 (i, j) pairs and 10 (i, k) candidates per agent, one lane each
// (lp3_project, lp3_candidate), then one serial scan per agent (lp3_scan) of at most 5 outer and 10 inner
// compare-and-select steps.  A projection RVO2 leaves out (parallel, same direction) is stored as an inert half-plane
// (zero direction: never violated, and "parallel with a zero numerator" as a constraint), which keeps the slots fixed.
__device__ __forceinline__ int lp3_program_of(int m) { return m >= 6 ? 4 : (m >= 3 ? 3 : (m >= 1 ? 2 : 1)); }  // m = i (i - 1) / 2 + j

__device__ __forceinline__ float4 lp3_project(const float4 li, const float4 lj) {
    const float d = li.z * lj.w - li.w * lj.z;
    const bool par = fabsf(d) <= kRvoEps;
    const bool same_dir = li.z * lj.z + li.w * lj.w > 0.0f;
    const float t = (lj.z * (li.y - lj.y) - lj.w * (li.x - lj.x)) / d;
    const float qx = par ? 0.5f * (li.x + lj.x) : li.x + t * li.z;
    const float qy = par ? 0.5f * (li.y + lj.y) : li.y + t * li.w;
    const float ex = lj.z - li.z, ey = lj.w - li.w;
    const float inv = 1.0f / sqrtf(ex * ex + ey * ey);
    const bool skip = par && same_dir;
    return make_float4(skip ? 0.0f : qx, skip ? 0.0f : qy, skip ? 0.0f : ex * inv, skip ? 0.0f : ey * inv);
}

// lines: the agent's half-planes; proj / cand: its 10 projected half-planes and their candidates, slot i (i - 1) / 2 + k
__device__ __forceinline__ void lp3_scan(const float4* lines, const float4* proj, const float4* cand, int n, int begin,
                                         float radius, float& rx, float& ry) {
    float distance = 0.0f;
    float4 li[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) li[i] = lines[i];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        float4 pk[4], ck[4];  // program i's projected half-planes and their candidates: requested before they are needed
#pragma unroll
        for (int k = 0; k < i; ++k) {
            pk[k] = proj[i * (i - 1) / 2 + k];
            ck[k] = cand[i * (i - 1) / 2 + k];
        }
        const float viol_i = li[i].z * (li[i].y - ry) - li[i].w * (li[i].x - rx);
        const bool active = (i >= begin) & (i < n) & (viol_i > distance);
        float r2x = -li[i].w * radius, r2y = li[i].z * radius;  // linearProgram2, directionOpt: start at opt * radius
        bool failed = false;
#pragma unroll
        for (int k = 0; k < i; ++k) {
            const float det = pk[k].z * (pk[k].y - r2y) - pk[k].w * (pk[k].x - r2x);
            const bool viol = !failed & (det > 0.0f);
            const bool feasible = ck[k].z != 0.0f;
            r2x = (viol & feasible) ? ck[k].x : r2x;
            r2y = (viol & feasible) ? ck[k].y : r2y;
            failed = failed | (viol & !feasible);
        }
        rx = (active & !failed) ? r2x : rx;
        ry = (active & !failed) ? r2y : ry;
        const float pen = li[i].z * (li[i].y - ry) - li[i].w * (li[i].x - rx);
        distance = active ? pen : distance;
    }
}

// lp3_scan split in two (round 5).  The planar program over the projections onto half-plane i — the inner loop above —
// reads the projections and their candidates only: NOT the running result (it starts at the normal of half-plane i times the
// radius).  The running result merely decides whether program i's solution is TAKEN.  So the four non-empty programs of an
// agent (i = 1..4: 1 + 2 + 3 + 4 steps) run side by side, each on a lane of its own (lp3_inner_program: the item lane that
// holds slot (i, 0)), and the agent's lane keeps the five outer compare-and-select steps (lp3_outer_scan) — a dependent chain
// of 4 inner + 5 outer steps and one LDS exchange instead of 10 inner + 5 outer behind 25 ds_read_b128.  Same operations on
// the same operands in the same order as lp3_scan: bit-identical.
//   proj / cand: the agent's 10 projected half-planes and their candidates (slot i (i - 1) / 2 + k); li: half-plane i
//   returns (solution.x, solution.y, infeasible ? 1 : 0, -) of program i
__device__ __forceinline__ float4 lp3_inner_program(const float4* proj, const float4* cand, int i, const float4 li, float radius) {
    const int base = i * (i - 1) / 2;
    float4 pk[4], ck[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {  // slots beyond the program's own (k >= i) belong to the next program or are unused: masked
        const int slot = base + k < 10 ? base + k : 9;
        pk[k] = proj[slot];
        ck[k] = cand[slot];
    }
    float r2x = -li.w * radius, r2y = li.z * radius;  // linearProgram2, directionOpt: start at opt * radius
    bool failed = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float det = pk[k].z * (pk[k].y - r2y) - pk[k].w * (pk[k].x - r2x);
        const bool viol = (k < i) & !failed & (det > 0.0f);
        const bool feasible = ck[k].z != 0.0f;
        r2x = (viol & feasible) ? ck[k].x : r2x;
        r2y = (viol & feasible) ? ck[k].y : r2y;
        failed = failed | (viol & !feasible);
    }
    return make_float4(r2x, r2y, failed ? 1.0f : 0.0f, 0.0f);
}
//   prog: the agent's row of program solutions, slot i = lp3_inner_program(.., i, ..) for i = 1..4 (slot 0 unused)
__device__ __forceinline__ void lp3_outer_scan(const float4* lines, const float4* prog, int n, int begin, float radius,
                                               float& rx, float& ry) {
    float4 li[5], pg[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        li[i] = lines[i];
        pg[i] = prog[i];
    }
    pg[0] = make_float4(-li[0].w * radius, li[0].z * radius, 0.0f, 0.0f);  // program 0 has no projected half-plane
    float distance = 0.0f;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const float viol_i = li[i].z * (li[i].y - ry) - li[i].w * (li[i].x - rx);
        const bool active = (i >= begin) & (i < n) & (viol_i > distance);
        const bool take = active & (pg[i].z == 0.0f);
        rx = take ? pg[i].x : rx;
        ry = take ? pg[i].y : ry;
        const float pen = li[i].z * (li[i].y - ry) - li[i].w * (li[i].x - rx);
        distance = active ? pen : distance;
    }
}

// ------------------------------------------------------------------ 2-D program on three lanes per agent
// linearProgram2 (Appendix A.4-A.5) for up to 10 half-planes with lane 3 g + m of a wave working for the g-th agent of a chunk of
// 21: lane m keeps the CONSTRAINT half-planes 3 m .. 3 m + 2 of its agent in registers (half-plane 9 never constrains another
// one; lane m = 2 merely watches it for violation), 12 + 4 registers instead of the 40 an unrolled per-lane program keeps, 63 of 64 lanes
// busy instead of 21.  Per round every agent advances to its next violated half-plane i (the three lanes agree on it through
// wave-shift DPP moves), each lane intersects its three half-planes with i — three divisions instead of nine — and folds them
// into a private interval; the three private intervals are then folded IN LANE ORDER (= line order) onto the speed disc's
// with RVO2's strict comparisons, which selects exactly the element the sequential loop keeps (the first one in line order
// that attains the minimum / maximum; RVO2's early exits are monotone, so testing them after the fold is equivalent).  Rounds per step = the largest number of violated half-planes of any agent of the wave
// (1.5 on average per agent in the 20-human shard), ~150 instructions each, against ~1 500 fully unrolled, mostly masked-off
// instructions per step.
//   lines [nA][kLineStride], count [nA], sol [nA] = (pref.x, pref.y, maxSpeed, solve ? 1 : 0)
//   res   [nA] = (result.x, result.y, int bits: first infeasible half-plane or n, -)
__device__ __forceinline__ float dpp_lane_below(float v) {  // lane l reads lane l - 1 (wave_shr:1)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_lane_above(float v) {  // lane l reads lane l + 1 (wave_shl:1)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));
}
// the values the three lanes of an agent hold, in lane order, on every one of them
__device__ __forceinline__ void tri_gather(float v, int m, float& v0, float& v1, float& v2) {
    const float a1 = dpp_lane_below(v), a2 = dpp_lane_below(a1);
    const float b1 = dpp_lane_above(v), b2 = dpp_lane_above(b1);
    v0 = m == 0 ? v : (m == 1 ? a1 : a2);
    v1 = m == 0 ? b1 : (m == 1 ? v : a1);
    v2 = m == 0 ? b2 : (m == 1 ? b1 : v);
}
#ifdef CN_PHASE_TIMING
// profiling builds: lp_relaxed_lazy [0] calls, [1] rounds that solved a projected program, [2] iterations that only handed
// agents out, [3] agents taken, [4] clock ticks inside the function (wave 0 lane 0 of every workgroup); lp_planar_tri [5] calls,
// [6] rounds, [7] agents still active summed over the rounds
static __device__ unsigned long long cn_lazy_counts[16];  // [8..11]: planar rounds whose largest violated index is <= 3 / <= 6 / <= 9, sum of it
#endif
constexpr int kTriAgents = kWave / 3;  // 21 agents per wave pass
__device__ __forceinline__ void lp_planar_tri(const float4* lines, const int* count, const float4* sol, float4* res, int nA,
                                              int threads) {
    constexpr int kNone = 99;
    const int tid = opaque_tid();
    const int wl = tid & (kWave - 1);
    const int g = wl / 3, m = wl - 3 * g;
    const int waves = (threads + kWave - 1) / kWave;
    const float inf = __builtin_inff();
#ifdef CN_PHASE_TIMING
    if (tid == 0) atomicAdd(&cn_lazy_counts[5], 1ull);
#endif
    for (int chunk = tid / kWave; chunk * kTriAgents < nA; chunk += waves) {
        const int a = chunk * kTriAgents + g;
        const bool live = g < kTriAgents && a < nA;
        const int aa = live ? a : 0;
        const float4 so = sol[aa];
        const float ox = so.x, oy = so.y, radius = so.z;
        const int n = (live && so.w != 0.0f) ? count[aa] : 0;
        const float4* lq = lines + aa * kLineStride;
        const float4 c0 = lq[3 * m], c1 = lq[3 * m + 1], c2 = lq[3 * m + 2];  // (slots >= n hold stale half-planes: masked by j < n)
        const float4 c9 = lq[9];
        float rx, ry;
        lp_start_point(radius, ox, oy, rx, ry);
        int cursor = 0;  // half-planes below the cursor are settled
        int fail = n;
        while (true) {
            // the agent's first violated half-plane at or above the cursor
            const int j0 = 3 * m;
            const bool v0 = j0 >= cursor && j0 < n && (c0.z * (c0.y - ry) - c0.w * (c0.x - rx) > 0.0f);
            const bool v1 = j0 + 1 >= cursor && j0 + 1 < n && (c1.z * (c1.y - ry) - c1.w * (c1.x - rx) > 0.0f);
            const bool v2 = j0 + 2 >= cursor && j0 + 2 < n && (c2.z * (c2.y - ry) - c2.w * (c2.x - rx) > 0.0f);
            const bool v9 = m == 2 && 9 >= cursor && 9 < n && (c9.z * (c9.y - ry) - c9.w * (c9.x - rx) > 0.0f);
            const int mine = v0 ? j0 : (v1 ? j0 + 1 : (v2 ? j0 + 2 : (v9 ? 9 : kNone)));
            float f0, f1, f2;
            tri_gather(__int_as_float(mine), m, f0, f1, f2);
            const int i0 = __float_as_int(f0), i1 = __float_as_int(f1), i2 = __float_as_int(f2);
            const int first = i0 < i1 ? (i0 < i2 ? i0 : i2) : (i1 < i2 ? i1 : i2);
            const bool act = live && first != kNone;
#ifdef CN_PHASE_TIMING
            {
                const unsigned long long am = __ballot(act && m == 0);
                if (tid == 0) {
                    atomicAdd(&cn_lazy_counts[6], 1ull);
                    atomicAdd(&cn_lazy_counts[7], (unsigned long long)__popcll(am));
                }
            }
#endif
            if (__ballot(act) == 0ull) break;
            const int i = act ? first : 0;
#ifdef CN_PHASE_TIMING
            {
                int mx = i;  // largest violated index among the wave's active agents
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(mx, off); mx = o > mx ? o : mx; }
                if (tid == 0) {
                    atomicAdd(&cn_lazy_counts[mx <= 3 ? 8 : (mx <= 6 ? 9 : 10)], 1ull);
                    atomicAdd(&cn_lazy_counts[11], (unsigned long long)mx);
                }
            }
#endif
            const float4 li = lq[i];
            const float px = li.x, py = li.y, dx = li.z, dy = li.w;
            // half-plane i inside the speed disc
            const float dp = px * dx + py * dy;
            const float disc = (dp * dp + radius * radius) - (px * px + py * py);
            bool ok = !(disc < 0.0f);
            const float root = sqrtf(disc);
            float t_lo = -dp - root;
            float t_hi = -dp + root;
            // this lane's three half-planes against half-plane i, folded in line order into a private interval
            float hi = inf, lo = -inf;
            bool bad = false;
            const float4 cs[3] = {c0, c1, c2};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const bool use = act && j0 + k < i;
                const float den = dx * cs[k].w - dy * cs[k].z;
                const float num = cs[k].z * (py - cs[k].y) - cs[k].w * (px - cs[k].x);
                const bool parallel = fabsf(den) <= kRvoEps;
                const float t = num / den;
                bad = bad || (use && parallel && num < 0.0f);
                const bool upper = use && !parallel && den >= 0.0f;
                const bool lower = use && !parallel && !(den >= 0.0f);
                hi = (upper && t < hi) ? t : hi;
                lo = (lower && lo < t) ? t : lo;
            }
            float h0, h1, h2, l0, l1, l2;
            tri_gather(hi, m, h0, h1, h2);
            tri_gather(lo, m, l0, l1, l2);
            t_hi = (h0 < t_hi) ? h0 : t_hi;
            t_hi = (h1 < t_hi) ? h1 : t_hi;
            t_hi = (h2 < t_hi) ? h2 : t_hi;
            t_lo = (t_lo < l0) ? l0 : t_lo;
            t_lo = (t_lo < l1) ? l1 : t_lo;
            t_lo = (t_lo < l2) ? l2 : t_lo;
            const unsigned badm = (unsigned)(__ballot(bad) >> (3 * g)) & 7u;
            ok = ok && badm == 0u && !(t_lo > t_hi);
            float tt = dx * (ox - px) + dy * (oy - py);
            tt = (tt < t_lo) ? t_lo : ((tt > t_hi) ? t_hi : tt);
            if (act) {
                if (ok) {
                    rx = px + tt * dx;
                    ry = py + tt * dy;
                    cursor = i + 1;
                } else {  // linearProgram2 returns i with the previous result
                    fail = i;
                    cursor = n;
                }
            }
        }
        if (live && m == 0) res[a] = make_float4(rx, ry, __int_as_float(fail), 0.0f);
    }
}

// ------------------------------------------------------------------ lane-cooperative 3-D fallback
// linearProgram3 (Appendix A.6) for the agents whose planar program was infeasible, with one lane per (agent, half-plane)
// (a per-agent serial walk of LDS with a dependent load per inner iteration, the other 60-odd lanes of the wave waiting, was
// 52 % of a step at 20 humans: profiles/r01_phase_probe.txt).  Here, per outer round, every agent advances to its next half-plane i whose
// violation exceeds the running distance; lane j < i projects half-plane j onto i (one division, one rsqrt — all j at
// once), and the planar program over the projected lines (directionOpt) runs in ballot / fold rounds:
// first violated projected line k, lanes j < k contribute their bound on k, fold in line order.  Projected lines that
// RVO2 skips (parallel, same direction) simply hold no line; order and every arithmetic operation are RVO2's.
//   res [nA] in: (result.x, result.y, int bits: first infeasible line), out: result
//   todo [n_todo] the agents that need the fallback, compacted: kWave / MAXL of them share a wave pass, so a 21-agent
//        env with 6 infeasible agents takes one pass at MAXL = 10 instead of one per 6-agent slice that holds any
template <int MAXL>
__device__ __forceinline__ void lp_relaxed_coop(const float4* lines, const int* count, const float4* sol, float4* res,
                                                const int* todo, int n_todo) {
    constexpr int G = kWave / MAXL;
    constexpr unsigned kField = (1u << MAXL) - 1u;
    const int wl = threadIdx.x & (kWave - 1);
    const int g = wl / MAXL, l = wl - g * MAXL;
    const int gbase = g * MAXL;
    const int waves = ((int)blockDim.x + kWave - 1) / kWave;
    const float inf = __builtin_inff();
    for (int chunk = threadIdx.x / kWave; chunk * G < n_todo; chunk += waves) {
        const bool live = g < G && chunk * G + g < n_todo;
        const int a = live ? todo[chunk * G + g] : 0;
        const float4 r0 = res[a];
        const int n = live ? count[a] : 0;
        const int begin = __float_as_int(r0.z);
        const bool need = live && begin < n;
        const float radius = sol[a].z;
        const float4 my = (l < n) ? lines[a * kLineStride + l] : make_float4(0.f, 0.f, 0.f, 0.f);
        float rx = r0.x, ry = r0.y, distance = 0.0f;
        int icur = need ? begin : n;
        while (true) {
            const bool cond = l >= icur && l < n && (my.z * (my.y - ry) - my.w * (my.x - rx) > distance);
            const unsigned long long m = __ballot(cond);
            if (m == 0ull) break;
            const unsigned gm = (unsigned)(m >> gbase) & kField;
            const bool act = gm != 0u;
            const int i = act ? __ffs(gm) - 1 : 0;
            const float4 li = lines[a * kLineStride + i];
            // this lane's half-plane projected onto half-plane i (meaningful for l < i)
            const float d = li.z * my.w - li.w * my.z;
            const bool par = fabsf(d) <= kRvoEps;
            const bool same_dir = li.z * my.z + li.w * my.w > 0.0f;
            const float t = (my.z * (li.y - my.y) - my.w * (li.x - my.x)) / d;
            const float qx = par ? 0.5f * (li.x + my.x) : li.x + t * li.z;
            const float qy = par ? 0.5f * (li.y + my.y) : li.y + t * li.w;
            const float ex = my.z - li.z, ey = my.w - li.w;
            const float inv = 1.0f / sqrtf(ex * ex + ey * ey);
            const float pz = ex * inv, pw = ey * inv;
            const bool valid = act && l < i && !(par && same_dir);
            // planar program over the projected lines, optimising along the normal of half-plane i
            const float ox = -li.w, oy = li.z;
            float r2x = ox * radius, r2y = oy * radius;
            int cur2 = 0;
            bool failed = false;
            while (true) {
                const bool viol = valid && l >= cur2 && (pz * (qy - r2y) - pw * (qx - r2x) > 0.0f);
                const unsigned long long m2 = __ballot(viol);
                if (m2 == 0ull) break;
                const unsigned gm2 = (unsigned)(m2 >> gbase) & kField;
                const bool act2 = gm2 != 0u;
                const int k = act2 ? __ffs(gm2) - 1 : 0;
                const float kx = __shfl(qx, gbase + k), ky = __shfl(qy, gbase + k);
                const float kz = __shfl(pz, gbase + k), kw = __shfl(pw, gbase + k);
                const float den = kz * pw - kw * pz;
                const float num = pz * (ky - qy) - pw * (kx - qx);
                const bool parallel = fabsf(den) <= kRvoEps;
                const float tk = num / den;
                const bool mine = act2 && valid && l < k;
                const bool bad = mine && parallel && num < 0.0f;
                const float c_hi = (mine && !parallel && den >= 0.0f) ? tk : inf;
                const float c_lo = (mine && !parallel && !(den >= 0.0f)) ? tk : -inf;
                const unsigned badm = (unsigned)(__ballot(bad) >> gbase) & kField;
                const float dp = kx * kz + ky * kw;
                const float disc = (dp * dp + radius * radius) - (kx * kx + ky * ky);
                bool ok = !(disc < 0.0f) && badm == 0u;
                const float root = sqrtf(disc);
                float t_lo = -dp - root;
                float t_hi = -dp + root;
#pragma unroll
                for (int j = 0; j < MAXL - 1; ++j) {
                    const float hj = __shfl(c_hi, gbase + j);
                    const float lj = __shfl(c_lo, gbase + j);
                    t_hi = (hj < t_hi) ? hj : t_hi;
                    t_lo = (t_lo < lj) ? lj : t_lo;
                }
                ok = ok && !(t_lo > t_hi);
                const float tt = (ox * kz + oy * kw > 0.0f) ? t_hi : t_lo;
                if (act2) {
                    if (ok) {
                        r2x = kx + tt * kz;
                        r2y = ky + tt * kw;
                        cur2 = k + 1;
                    } else {  // the projected program is infeasible: keep the previous result (RVO2: result = tempResult)
                        failed = true;
                        cur2 = MAXL;
                    }
                }
            }
            if (act) {
                if (!failed) {
                    rx = r2x;
                    ry = r2y;
                }
                distance = li.z * (li.y - ry) - li.w * (li.x - rx);
                icur = i + 1;
            }
        }
        if (need && l == 0) res[a] = make_float4(rx, ry, r0.z, 0.0f);
    }
}

// lp_line_candidate<4> for the 5 lanes (agent, half-plane k) of the fused kernel's candidate stage, PAIR-LANE form (round 6; see
// lp_line_candidate_pairs9 below for the idea): the 10 (line, earlier line) pairs of an agent are dealt two per lane —
//   lane 0: (4, 2) (4, 3)   lane 1: (1, 0) (3, 2)   lane 2: (2, 0) (2, 1)   lane 3: (3, 0) (3, 1)   lane 4: (4, 0) (4, 1)
// — instead of four masked ones; rows 3 and 4 receive the later part of their interval from lanes 1 and 0 of their group through
// wave shuffles and fold it behind their own part (line order, RVO2's strict comparisons).  lq: the agent's half-planes in LDS;
// called by lanes (agent, k) that are contiguous in the wave, k fastest; returns what lp_line_candidate<4>(lq[k], lq, k, ...,
// dir_opt = false) returns.
__device__ __forceinline__ float4 lp_line_candidate_pairs5(const float4* lq, int k, int lane, float radius, float ox, float oy) {
    const float inf = __builtin_inff();
    const float4 lk = lq[k];
    const int row0 = k == 0 ? 4 : k, j0 = k == 0 ? 2 : 0;
    const int row1 = k == 0 ? 4 : (k == 1 ? 3 : k), j1 = k == 0 ? 3 : (k == 1 ? 2 : 1);
    const bool own0 = k != 0, own1 = k >= 2;
    const float4 K0 = lq[row0], J0 = lq[j0], K1 = lq[row1], J1 = lq[j1];
    float hi_c[2], lo_c[2];
    bool bad_c[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const float4 K = s ? K1 : K0, J = s ? J1 : J0;
        const float den = K.z * J.w - K.w * J.z;
        const float num = J.z * (K.y - J.y) - J.w * (K.x - J.x);
        const bool parallel = fabsf(den) <= kRvoEps;
        const float t = num / den;
        bad_c[s] = parallel && num < 0.0f;
        hi_c[s] = (!parallel && den >= 0.0f) ? t : inf;
        lo_c[s] = (!parallel && !(den >= 0.0f)) ? t : -inf;
    }
    // own part (slots in line order) and the part this lane computes for another row
    float hi_o = own0 ? hi_c[0] : inf, lo_o = own0 ? lo_c[0] : -inf;
    bool bad_o = own0 && bad_c[0];
    float hi_h = own0 ? inf : hi_c[0], lo_h = own0 ? -inf : lo_c[0];
    bool bad_h = !own0 && bad_c[0];
    hi_o = (own1 && hi_c[1] < hi_o) ? hi_c[1] : hi_o;
    lo_o = (own1 && lo_o < lo_c[1]) ? lo_c[1] : lo_o;
    bad_o = bad_o || (own1 && bad_c[1]);
    hi_h = (!own1 && hi_c[1] < hi_h) ? hi_c[1] : hi_h;
    lo_h = (!own1 && lo_h < lo_c[1]) ? lo_c[1] : lo_h;
    bad_h = bad_h || (!own1 && bad_c[1]);
    // rows 3 and 4 fetch the later part of their interval: from lane 1 / lane 0 of the agent's group
    const int src = lane - (k == 3 ? 2 : (k == 4 ? 4 : 0));
    const float hi_r = __shfl(hi_h, src), lo_r = __shfl(lo_h, src);
    const int bad_r = __shfl(bad_h ? 1 : 0, src);
    const bool has = k >= 3;
    const float px = lk.x, py = lk.y, dx = lk.z, dy = lk.w;
    const float dp = px * dx + py * dy;
    const float disc = (dp * dp + radius * radius) - (px * px + py * py);
    bool ok = !(disc < 0.0f);
    const float root = sqrtf(disc);
    float t_lo = -dp - root;
    float t_hi = -dp + root;
    t_hi = (hi_o < t_hi) ? hi_o : t_hi;
    t_lo = (t_lo < lo_o) ? lo_o : t_lo;
    t_hi = (has && hi_r < t_hi) ? hi_r : t_hi;
    t_lo = (has && t_lo < lo_r) ? lo_r : t_lo;
    ok = ok && !bad_o && !(has && bad_r != 0) && !(t_lo > t_hi);
    float t = dx * (ox - px) + dy * (oy - py);
    t = (t < t_lo) ? t_lo : ((t > t_hi) ? t_hi : t);
    return make_float4(px + t * dx, py + t * dy, ok ? 1.0f : 0.0f, 0.0f);
}

// lp_line_candidate<8> for the 9 lanes of an agent in the lazy fallback, PAIR-LANE form (round 6).  Lane l's candidate needs the
// intersections of projected line l with lines 0 .. l-1: 36 (line, earlier line) pairs per agent, which the masked loop above
// spreads as 8 per lane, half of them switched off.  Here every lane takes exactly 4 pairs: lanes 4..8 the first four entries
// (j = 0..3) of their own row, lanes 0..3 their own (short) row and the REST of row 8 - l (j = 4 .. 7 - l), whose private
// interval goes to the row's lane through one more LDS exchange.  Private intervals start from +-inf and are folded onto the speed
// disc's in line order with RVO2's strict comparisons — own part first, then the helper's (the later j) — which selects exactly the
// element the sequential loop keeps (as lp_planar_tri folds its three lanes); the emptiness test after the fold is equivalent to
// RVO2's early exit (t_lo only grows, t_hi only shrinks).  Four IEEE divisions per lane and round instead of eight.
//   lk: this lane's projected line (row l, also at prow[l]); bpart: 4 float4 of scratch of this lane's GROUP (rows 5..8);
//   all 64 lanes call it (one wave_lds_sync inside); returns what lp_line_candidate<8>(lk, prow, l, radius, ox, oy, true) returns
__device__ __forceinline__ float4 lp_line_candidate_pairs9(const float4 lk, const float4* prow, float4* bpart, int l, bool live,
                                                           float radius, float ox, float oy) {
    const float px = lk.x, py = lk.y, dx = lk.z, dy = lk.w;
    const float dp = px * dx + py * dy;
    const float disc = (dp * dp + radius * radius) - (px * px + py * py);
    bool ok = !(disc < 0.0f);
    const float root = sqrtf(disc);
    float t_lo = -dp - root;
    float t_hi = -dp + root;
    const float inf = __builtin_inff();
    const bool helper = l < 4;
    const float4 hk = prow[helper ? 8 - l : l];  // the row this lane helps with (lanes 4..8: their own again, unused)
    float hi_own = inf, lo_own = -inf, hi_help = inf, lo_help = -inf;
    bool bad_own = false, bad_help = false;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const bool own = !helper || s < l;        // slot s of this lane: pair (row l, j = s) or (row 8 - l, j = 4 + s - l)
        const float4 lj = prow[own ? s : 4 + s - l];
        const float kx = own ? px : hk.x, ky = own ? py : hk.y, kdx = own ? dx : hk.z, kdy = own ? dy : hk.w;
        const float den = kdx * lj.w - kdy * lj.z;
        const float num = lj.z * (ky - lj.y) - lj.w * (kx - lj.x);
        const bool parallel = fabsf(den) <= kRvoEps;
        const float t = num / den;
        const bool bad = parallel && num < 0.0f;
        const bool upper = !parallel && den >= 0.0f;
        const bool lower = !parallel && !(den >= 0.0f);
        hi_own = (own && upper && t < hi_own) ? t : hi_own;
        lo_own = (own && lower && lo_own < t) ? t : lo_own;
        bad_own = bad_own || (own && bad);
        hi_help = (!own && upper && t < hi_help) ? t : hi_help;
        lo_help = (!own && lower && lo_help < t) ? t : lo_help;
        bad_help = bad_help || (!own && bad);
    }
    if (live && helper) bpart[3 - l] = make_float4(hi_help, lo_help, bad_help ? 1.0f : 0.0f, 0.0f);  // row 8 - l -> slot (8 - l) - 5
    wave_lds_sync();
    const bool has_help = l >= 5;
    const float4 hb = bpart[has_help ? l - 5 : 0];
    t_hi = (hi_own < t_hi) ? hi_own : t_hi;
    t_lo = (t_lo < lo_own) ? lo_own : t_lo;
    t_hi = (has_help && hb.x < t_hi) ? hb.x : t_hi;
    t_lo = (has_help && t_lo < hb.y) ? hb.y : t_lo;
    ok = ok && !bad_own && !(has_help && hb.z != 0.0f) && !(t_lo > t_hi);
    const float t = (ox * dx + oy * dy > 0.0f) ? t_hi : t_lo;
    return make_float4(px + t * dx, py + t * dy, ok ? 1.0f : 0.0f, 0.0f);
}

// ------------------------------------------------------------------ 3-D fallback, candidate form evaluated lazily
// linearProgram3 (Appendix A.6) for 10 half-planes.  The all-pairs candidate form (lp3_scan at 5 half-planes) would compute the projections and 1-D
// solutions of ALL 45 (i, j) pairs of an infeasible agent up front — at 5 half-planes (10 pairs) that is what makes the
// fallback cheap, at 10 it costs as much as it saves, because the sequential program only ever visits the few outer
// half-planes i whose violation exceeds the running distance (built and rejected: profiles/HISTORY.md).  lp_relaxed_coop does visit only
// those, but solves each projected planar program in ballot / shuffle rounds: one round per violated projected line, 18
// ds_bpermute each.  This version keeps coop's outer structure — lane = (agent, half-plane), the agents of a pass advance
// together to their next violated half-plane i — and solves the projected program of that ONE i in candidate form: lane l < i
// projects its half-plane onto i (lp3_project), then computes the 1-D solution of projected line l against projected lines
// 0 .. l-1 (lp_line_candidate, direction objective), both through the agent's LDS rows; then every lane of the agent runs the
// same short scan over the ≤ 9 candidates (no broadcast needed).  Same operations on the same operands as the sequential
// program: bit-identical.  One round costs ~300 instructions whatever the number of violated projected lines.
//   lines [nA][kLineStride]; proj [nA][kLineStride] scratch rows; cand: kLazyCandFloat4 float4 of scratch per WAVE
//   (one row of MAXL - 1 per agent of the pass + 4 float4 per agent for the pair-lane candidates); res [nA] in: (result, int bits: first infeasible line), out: result; todo [n_todo]: the
//   agents that need the fallback, compacted (kWave / MAXL of them share a pass)
//   GROUP_ROWS: proj holds one row of MAXL - 1 projected half-planes per lane GROUP of the wave (kWave / (MAXL - 1) rows: the
//   compact LDS layout of the 20-human shard's kernel) instead of one row of kLineStride per agent
template <int MAXL, bool GROUP_ROWS = false>
__device__ __forceinline__ void lp_relaxed_lazy(const float4* lines, float4* proj, float4* cand, const int* count,
                                                const float4* sol, float4* res, const int* todo, int n_todo, int threads) {
    // W = MAXL - 1 lanes per agent: lane l holds half-plane l (projections and candidates exist for l < MAXL - 1 only), and the
    // agent's last lane also tests half-plane MAXL - 1 for violation — 7 agents at a time at 10 half-planes instead of 6 with a
    // lane per half-plane (a step of the 20-human shard has 6.0 infeasible agents on average: a second pass for the 7th
    // was the common case).  A lane group whose agent is done takes the next one from the list (one wave per workgroup; with
    // several waves each wave keeps its own chunks of G agents): the rounds of a step are then about (sum over its agents) / G,
    // not the sum over passes of each pass's slowest agent.
    constexpr int W = MAXL - 1, G = kWave / W;
    constexpr unsigned kField = (1u << W) - 1u;
    const int tid = opaque_tid();
    const int wl = tid & (kWave - 1);
    const int g = wl / W, l = wl - g * W;
    const int gbase = g * W;
    const int waves = (threads + kWave - 1) / kWave;
    float4* const crow = cand + ((tid / kWave) * G + (g < G ? g : 0)) * W;
    // (pair-lane candidates: the helpers' private intervals of rows 5..8, 4 float4 per group behind the waves' candidate rows)
    float4* const bpart = cand + (size_t)waves * G * W + ((tid / kWave) * G + (g < G ? g : 0)) * 4;
#ifdef CN_PHASE_TIMING
    unsigned long long pt_rounds = 0ull, pt_hand = 0ull;
    const unsigned long long pt_t0 = __builtin_readcyclecounter();
#endif
    for (int chunk = tid / kWave; chunk * G < n_todo; chunk += waves) {
        const int hand_end = waves == 1 ? n_todo : (chunk * G + G < n_todo ? chunk * G + G : n_todo);  // agents this wave works off
        int next = chunk * G + G;                                                                     // ... the next one to hand out
        // per-agent state of the lane's group
        bool live = false, need = false, tail = false;
        int a = 0, n = 0, icur = 0;
        float radius = 0.0f, rx = 0.0f, ry = 0.0f, r0z = 0.0f, distance = 0.0f;
        float4 my = make_float4(0.f, 0.f, 0.f, 0.f), last = my;
        float4* prow = proj;
        const auto take = [&](int idx) {
            live = g < G && idx < hand_end;
            a = live ? todo[idx] : 0;
            const float4 r0 = res[a];
            n = live ? count[a] : 0;
            const int begin = __float_as_int(r0.z);
            need = live && begin < n;
            radius = sol[a].z;
            my = (l < n) ? lines[a * kLineStride + l] : make_float4(0.f, 0.f, 0.f, 0.f);
            tail = l == W - 1 && W < n;  // this lane also watches half-plane W
            last = tail ? lines[a * kLineStride + W] : make_float4(0.f, 0.f, 0.f, 0.f);
            prow = GROUP_ROWS ? proj + ((tid / kWave) * G + (g < G ? g : 0)) * W : proj + a * kLineStride;
            rx = r0.x, ry = r0.y, r0z = r0.z, distance = 0.0f;
            icur = need ? begin : n;
        };
        take(chunk * G + g);
        while (true) {
            const bool cond = l >= icur && l < n && (my.z * (my.y - ry) - my.w * (my.x - rx) > distance);
            const bool condw = tail && W >= icur && (last.z * (last.y - ry) - last.w * (last.x - rx) > distance);
            const unsigned long long m = __ballot(cond), mw = __ballot(condw);
            const unsigned gm = ((unsigned)(m >> gbase) & kField) | (((unsigned)(mw >> (gbase + W - 1)) & 1u) << W);
            const bool act = gm != 0u;
            if (live && !act) {  // this group's agent has no violated half-plane left
                if (need && l == 0) res[a] = make_float4(rx, ry, r0z, 0.0f);
                live = false;
            }
            if (next < hand_end) {  // free groups take the next agents of the list, in group order
                const unsigned long long idle = __ballot(l == 0 && g < G && !act);
                if (idle != 0ull) {
                    const int idx = next + __popcll(idle & ((1ull << gbase) - 1ull));
                    if (!act) take(idx);
                    next += __popcll(idle);
#ifdef CN_PHASE_TIMING
                    ++pt_hand;
#endif
                    continue;
                }
            }
            if ((m | mw) == 0ull) break;
#ifdef CN_PHASE_TIMING
            ++pt_rounds;
#endif
            const int i = act ? __ffs(gm) - 1 : 0;
            const float4 li = lines[a * kLineStride + i];
            // my half-plane projected onto half-plane i (the ones RVO2 leaves out, and lanes l >= i, hold an inert line)
            const float4 pr = (act && l < i) ? lp3_project(li, my) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (live) prow[l] = pr;
            // 1-D solution on projected line l against projected lines 0 .. l-1, optimising along the normal of half-plane i
            static_assert(W == 9, "lp_line_candidate_pairs9 deals the 36 pairs of 9 projected lines");
            const float4 cd = lp_line_candidate_pairs9(pr, prow, bpart, l, live, radius, -li.w, li.z);
            if (live) crow[l] = cd;
            // linearProgram2 over the projected lines as a scan of the candidates (every lane of the agent, identically)
            float r2x = -li.w * radius, r2y = li.z * radius;
            bool failed = false;
#pragma unroll
            for (int k = 0; k < MAXL - 1; ++k) {
                const float4 pk = prow[k], ck = crow[k];
                const float det = pk.z * (pk.y - r2y) - pk.w * (pk.x - r2x);
                const bool viol = (k < i) & !failed & (det > 0.0f);
                const bool feasible = ck.z != 0.0f;
                r2x = (viol & feasible) ? ck.x : r2x;
                r2y = (viol & feasible) ? ck.y : r2y;
                failed = failed | (viol & !feasible);
            }
            if (act) {
                if (!failed) {
                    rx = r2x;
                    ry = r2y;
                }
                distance = li.z * (li.y - ry) - li.w * (li.x - rx);
                icur = i + 1;
            }
        }
        if (waves == 1) break;
    }
#ifdef CN_PHASE_TIMING
    if (tid == 0) {
        atomicAdd(&cn_lazy_counts[0], 1ull);
        atomicAdd(&cn_lazy_counts[1], pt_rounds);
        atomicAdd(&cn_lazy_counts[2], pt_hand);
        atomicAdd(&cn_lazy_counts[3], (unsigned long long)n_todo);
        atomicAdd(&cn_lazy_counts[4], __builtin_readcyclecounter() - pt_t0);
    }
#endif
}

}  // namespace cn
