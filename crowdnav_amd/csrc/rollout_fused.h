// The fused rollout for the small-crowd geometries (<= 5 candidate neighbours per agent, one wave per workgroup,
// holonomic robot): the same transition as step_core / rollout_kernel (step_kernels.h) — same functions for every number
// it computes — with FOUR LDS exchange points per step instead of nine:
//
//   [stage]   agent lanes   float32 view of the agents, preferred velocities        (written at the END of the previous step)
//   pairs     pair lanes    the agent's candidate distances in registers (no d2 round trip), rank, half-plane   -> barrier 1
//   cands     (agent, half-plane) lanes   1-D solutions (lp_line_candidate)                                     -> barrier 2
//   solve     agent lanes   scan (+ candidate-form 3-D fallback for the infeasible ones, 3 barriers only then);
//                           the robot's action reaches its env's lanes by a wave shuffle, not through LDS;
//                           float64 swept distance / goal distance                                             -> barrier 3
//   reduce    agent lanes   EVERY lane of an env reduces the env's distances to reward / done / info itself (idle lanes
//                           otherwise), so the episode bookkeeping that decides what the agents do next — advance, load the
//                           next scenario from the ring, pause — needs no flag broadcast; integrate; stage the next step
//                                                                                                                -> barrier 4
// Per-env episode state (global_time, episodes finished, ring fill level, running / waiting / retired) is replicated on
// the env's lanes; only the robot lane keeps the return / danger accumulators and writes records.
// Launch conditions (cn_rollout / cn_rollout_step check them, everything else runs rollout_kernel): 5-half-plane
// instantiation, NC <= 5, pairs <= 64, nA * 5 <= 64, 64 threads, holonomic robot.
#pragma once
#include "step_kernels.h"

namespace cn {

constexpr int kFusedMaxNC = 5;

struct EpisodeRegs {  // replicated on every agent lane of the env
    double gtime;
    int state, ep_count, ring_filled;
};

// A wave-uniform value pinned in vector registers.  The eight float64 parameters of the step (dt, time limit, rewards,
// discomfort distance / factor, safety space) arrive as kernel arguments in SGPRs; with ~100 SGPRs live in the step loop the
// register allocator spilled that 16-dword block into VGPR lanes and re-read it with v_readlane five times per step (107
// v_readlane of the loop's ~1 450 instructions: every one a VALU issue slot).  As VALU operands they are needed in VGPRs
// anyway; the empty asm makes the copy explicit and opaque, so the values stay there (16 VGPRs; the kernel has room up to 168).
template <typename T>
__device__ __forceinline__ T in_vgpr(T x) {
    asm volatile("" : "+v"(x));
    return x;
}

// What the NEXT step's pair and collide phases read: the float32 view of the agents, float64 position, radii.
__device__ __forceinline__ void stage_agent(const Params& P, const Smem& s, const Lane& L, const AgentRegs& r,
                                            double human_safety) {
    if (L.lane >= P.nA) return;
    s.kin[L.lane] = make_float4((float)r.px, (float)r.py, (float)r.vx, (float)r.vy);
    s.posd[L.lane] = make_double2(r.px, r.py);
    s.rad[L.lane] = r.rad;
    s.hview[L.lane] = (float)(r.rad + 0.01 + human_safety);
}

// The agent's preferred velocity — towards its goal, unit length once farther than 1 m (orca.py:113-115): a float64 norm and
// two float64 divisions, ~450 clock ticks of dependent latency on 12 lanes — and linearProgram2's start point (Appendix A.4).
// Nothing reads them before the candidates phase, so they are computed INSIDE the pair phase, branch-free on every lane: one
// basic block with the float32 pair arithmetic, and the scheduler interleaves the two dependency chains.
__device__ __forceinline__ void preferred_velocity(const AgentRegs& r, float max_speed, bool solve, float4& sol, float4& start) {
    const double gdx = r.gx - r.px, gdy = r.gy - r.py;
    const double speed = norm2(gdx, gdy);
    const float pref_x = (float)(speed > 1.0 ? gdx / speed : gdx);
    const float pref_y = (float)(speed > 1.0 ? gdy / speed : gdy);
    sol = make_float4(pref_x, pref_y, max_speed, solve ? 1.0f : 0.0f);
    float sx, sy;
    lp_start_point(max_speed, pref_x, pref_y, sx, sy);
    start = make_float4(sx, sy, 0.0f, 0.0f);
}

// The fused kernel is ONE wave per workgroup: LDS instructions of a wave execute in order, so what the phases need between a
// lane's write and another lane's read is only that the compiler keeps the accesses in program order — not s_barrier with
// its s_waitcnt lgkmcnt(0) in front (the LDS queue drained five times per step).
#define CN_FUSED_SYNC() wave_lds_sync()

// Issue priority: a launch ends with its slowest wave, and that is a wave whose env sits in a jam and takes the 3-D fallback
// every step — 1.6 x the latency of a step without it — while the wave it shares the SIMD with has slack.  A wave entering the
// fallback raises its priority (s_setprio 3) and keeps it until the first step that does without (a jam lasts many steps):
// nothing at 20 steps per launch, +3-5 % at 1000.
// The one-pass fallback runs the four planar programs of an infeasible agent on four lanes (orca_device.h: lp3_inner_program /
// lp3_outer_scan); the multi-pass form (more than six infeasible agents in a wave) keeps lp3_scan.

// CN_WAVE_TRACE (profiling builds): every wave leaves four 100 MHz timestamps (kernel entry, step loop entry / exit, kernel
// exit) and how many of its steps took the 3-D fallback / ended an episode: scripts/probes/wave_trace.py
#ifdef CN_WAVE_TRACE
static __device__ unsigned long long cn_wave_trace[8192 * 6];
#endif

template <bool HEADLINE>
// (the headline instantiation is compiled for three waves per SIMD: with the hint hipcc settles on 163-167 VGPRs and a schedule
// worth 1.2 % at 4096 envs, 2.6 % in the 20-step shape; compiled for two it loses 3 %, for four — 128 VGPRs, 42 spilled — 11 %)
__global__ __launch_bounds__(kWave, (HEADLINE ? 3 : 1)) void rollout_fused_kernel(Params P_in, const StateView* Sd, const int* ring_filled_in,
                                                              RolloutView R, int n_steps, const double* ext_action) {
#ifdef CN_WAVE_TRACE
    const unsigned long long wt_entry = __builtin_amdgcn_s_memrealtime();
    unsigned long long wt_fallbacks = 0ull, wt_ends = 0ull;
#endif
    // The 17 state pointers are needed before and after the step loop and when an episode ends — never inside a step — so
    // they stay in the engine's device copy of the StateView and are re-read where used (scalar loads) instead of holding
    // 34 SGPRs (and spilling as many into VGPR lanes) across the loop.  ring_filled_in is the one pointer the host swaps
    // between launches, hence a direct argument.
    constexpr int MAXL = 5;
    Params P = P_in;
    if (HEADLINE) {  // BASELINE configs[1]: 5 humans + robot, 2 envs per wave, 60 pairs, as compile-time constants
        P.A = 6, P.NC = 5, P.E = 2, P.nA = 12, P.pairs = 60, P.threads = 64;
    }
    const Smem s = carve<MAXL>(P);
    const Lane L = lane_of(P);
    AgentRegs r = {};
    float robot_max_speed = 0.0f;
    const bool robot = L.valid && L.a == 0;
    // the ~20 pointers of the io block are needed at launch start / end and when an episode ends: they are re-read from
    // the device copy there (scalar loads) instead of occupying SGPRs across the step loop
    const cn_rollout_io* iop = R.io;
    double theta = 0.0;
    EpisodeRegs ep{0.0, kRetired, 0, 0};
    double cur_return = 0.0, cur_dsum = 0.0;
    int cur_steps = 0, cur_danger = 0;
    {
        // Launch prologue in TWO memory round trips: every pointer it needs in one batch of scalar loads (both structs are
        // dead again before the step loop), then every per-lane value in one batch of vector loads — unconditional, on clamped
        // indices, so that none waits for another's result (the robot's captured radii are read speculatively: the buffers
        // exist whether or not the policy's simulator has been built).  As a chain of `S->field[..]` / `io->field[..]` reads
        // under their own conditions this was a dozen dependent trips, ~4 of the 113 us of a 20-step launch.
        const StateView S = *Sd;
        const cn_rollout_io io = *iop;
        const size_t gi = L.valid ? L.gi : 0;
        const int env = L.valid ? L.env : 0;
        const double2 p0 = S.pos[gi], v0 = S.vel[gi], g0 = S.goal[gi], q0 = S.rv[gi];
        const bool have = P.robot_orca ? S.rsim_valid[env] != 0 : false;
        const float rr_kept = P.robot_orca ? S.rsim_radius[gi] : 0.0f;
        const float ms_kept = P.robot_orca ? S.rsim_max_speed[env] : 0.0f;
        const double theta0 = S.theta[env], gtime0 = S.gtime[env];
        const int state0 = io.active[env], count0 = io.ep_count[env], filled0 = ring_filled_in[env];
        const int steps0 = io.cur_steps[env];
        const double return0 = io.cur_return[env];
        const int danger0 = io.cur_danger ? io.cur_danger[env] : 0;
        const double dsum0 = io.cur_danger_dmin_sum ? io.cur_danger_dmin_sum[env] : 0.0;
        if (L.valid) {
            r.px = p0.x, r.py = p0.y, r.vx = v0.x, r.vy = v0.y, r.gx = g0.x, r.gy = g0.y, r.rad = q0.x, r.vpref = q0.y;
            ep.gtime = gtime0, ep.state = state0, ep.ep_count = count0, ep.ring_filled = filled0;
            // (every lane of the env: the accumulators are carried redundantly, see the reduce phase)
            cur_steps = steps0, cur_return = return0, cur_danger = danger0, cur_dsum = dsum0;
            if (L.a == 0) theta = theta0;
            if (P.robot_orca) {  // load_robot_view (step_kernels.h), without its dependent loads
                const float rr = have ? rr_kept : (float)(r.rad + 0.01 + P.robot_safety);
                if (!have) S.rsim_radius[L.gi] = rr;
                s.rview[L.lane] = rr;
                if (L.a == 0) {
                    robot_max_speed = have ? ms_kept : (float)r.vpref;
                    if (!have) S.rsim_max_speed[L.env] = robot_max_speed;
                }
            }
        }
    }
    build_pairs(P, s);

    if (L.valid && ep.state == kWaitingScenario && ep.ep_count < ep.ring_filled) {  // the fill kernel has just produced it
        // (the asynchronous fill is for the wave generators, more than 8 humans: never this kernel)
        load_from_ring(P, *Sd, L, ep.ep_count % P.ring_depth, r);
        ep.state = kRunning;
        ep.gtime = 0.0;
        theta = 1.5707963267948966;
    }
    unsigned int transitions = 0;
    for (int t = threadIdx.x; t < kMaxDiscount; t += blockDim.x) s.disc[t] = t < R.discount_len ? R.discount[t] : 0.0;
    CN_FUSED_SYNC();  // pinfo, rview
    // this pair lane's row: the kin slots of its agent's candidates (8 bits each).  A pair that does not exist (robot
    // invisible to the humans, env beyond the batch, fewer than 5 candidates) points at slot nA = (+inf, +inf): its squared
    // distance is +inf without a select, so the pair phase below has no data-dependent control flow at all.
    int my_info = 0;
    unsigned long long row_slots = 0ull;
    int my_slot = P.nA;
    if (L.lane == 0) s.kin[P.nA] = make_float4(std::numeric_limits<float>::infinity(), std::numeric_limits<float>::infinity(), 0.0f, 0.0f);
    if (L.lane < P.pairs) {
        my_info = s.pinfo[L.lane];
        const int c = (my_info >> 16) & 0xff;
        for (int k = 0; k < kFusedMaxNC; ++k) {
            int slot = P.nA;
            if (k < P.NC) {
                const int ik = s.pinfo[L.lane - c + k];
                slot = ((ik >> 24) & 1) ? ((ik >> 8) & 0xff) : P.nA;
            }
            row_slots |= (unsigned long long)slot << (8 * k);
        }
        my_slot = (int)((row_slots >> (8 * c)) & 0xffull);
    }
    const double c_dt = in_vgpr(P.dt), c_limit = in_vgpr(P.time_limit), c_limit1 = in_vgpr(P.time_limit - 1.0);
    const double c_success = in_vgpr(P.success_reward), c_collision = in_vgpr(P.collision_penalty);
    const double c_ddist = in_vgpr(P.discomfort_dist), c_dfactor = in_vgpr(P.discomfort_factor);
    const double c_hsafety = in_vgpr(P.human_safety);
    stage_agent(P, s, L, r, c_hsafety);
    CN_FUSED_SYNC();

#ifdef CN_PHASE_TIMING
    PhaseClock clock = {};
    PhaseClock* clk = &clock;
    clock.last = __builtin_readcyclecounter();
#else
    PhaseClock* clk = nullptr;
    (void)clk;
#endif
    const float range_sq = P.orca.neighbor_dist * P.orca.neighbor_dist;
#ifdef CN_WAVE_TRACE
    const unsigned long long wt_loop = __builtin_amdgcn_s_memrealtime();
#endif
    for (int step = 0; step < n_steps; ++step) {
        const bool running = L.valid && ep.state == kRunning;
        const bool solve = running && (L.a > 0 || P.robot_orca);

        // ---- pairs: candidate distances (Appendix A.2), stable rank = RVO2's sorted-insertion slot, half-plane (A.3)
        if (L.lane < P.pairs) {
            const int q = my_info & 0xff, c = (my_info >> 16) & 0xff;
            const int ol = (my_info >> 8) & 0xff;
            const bool robot_sim = (my_info >> 25) & 1;
            // every LDS request of the phase first, none of them behind a condition
            const float4 me = s.kin[q];
            const float4 other = s.kin[my_slot];
            float4 ot[kFusedMaxNC];
#pragma unroll
            for (int k = 0; k < kFusedMaxNC; ++k) ot[k] = s.kin[(int)((row_slots >> (8 * k)) & 0xffull)];
            const float* view = robot_sim ? s.rview : s.hview;
            const float rsum = view[q] + view[ol];
            const float odx = me.x - other.x, ody = me.y - other.y;
            const float mine = odx * odx + ody * ody;
            int rank = 0, within = 0;
            {
                // squared distances are +0 .. +inf: their bit patterns order like the floats, so "v < mine, or v == mine and
                // k < c" is bit 31 of v - (mine + [k < c]) and "v < range" is bit 31 of v - range as 32-bit integers: the sign
                // bits are shifted into two words (v_alignbit_b32) and counted once — no compare through VCC / SGPR pairs, no
                // scalar mask logic, none of the hazard s_nop between them (round 6; the shard's rank loop does the same)
                const uint32_t mb = __float_as_uint(mine), rb = __float_as_uint(range_sq);
                uint32_t before = 0u, inside = 0u;
#pragma unroll
                for (int k = 0; k < kFusedMaxNC; ++k) {
                    const float dx = me.x - ot[k].x, dy = me.y - ot[k].y;
                    const uint32_t vb = __float_as_uint(dx * dx + dy * dy);  // +inf for a pair that does not exist: never in range
                    const uint32_t tie = (uint32_t)(k - c) >> 31;           // k < c
                    before = __builtin_amdgcn_alignbit(before, vb - mb - tie, 31);
                    inside = __builtin_amdgcn_alignbit(inside, vb - rb, 31);
                }
                within = __popc(inside);
                rank = __popc(before & inside);
            }
            // (agent lanes are pair lanes too: their preferred velocity, same block)
            float4 sol4, start4;
            preferred_velocity(r, (L.a == 0) ? robot_max_speed : (float)r.vpref, solve, sol4, start4);
            if (c == 0) s.count[q] = within < P.orca.max_neighbors ? within : P.orca.max_neighbors;
            if (mine < range_sq && rank < P.orca.max_neighbors)
                s.lines[q * kLineStride + rank] =
                    make_half_plane(P.orca, me.x, me.y, me.z, me.w, other.x, other.y, other.z, other.w, rsum);
            if (L.lane < P.nA) {
                s.sol[L.lane] = sol4;
                s.res[L.lane] = start4;
            }
        }
        CN_FUSED_SYNC();
        CN_TICK(clk, 2);

        // ---- candidates: lane = (agent, half-plane)
        if (L.lane < P.nA * MAXL) {
            const int q = L.lane / MAXL, k = L.lane - q * MAXL;
            const float4 so = s.sol[q];
            const float4* lq = s.lines + q * kLineStride;
            s.cand2[q * kLineStride + k] = lp_line_candidate_pairs5(lq, k, L.lane, so.z, so.x, so.y);
        }
        CN_FUSED_SYNC();

        // ---- solve: scan, then the candidate-form fallback for the infeasible agents
        float rx = 0.0f, ry = 0.0f;
        int n = 0, fail = 0;
        if (solve) {
            n = s.count[L.lane];
            const float4 start = s.res[L.lane];
            rx = start.x, ry = start.y;
            fail = lp_planar_scan<MAXL>(s.lines + L.lane * kLineStride, s.cand2 + L.lane * kLineStride, n, rx, ry);
        }
        CN_TICK(clk, 3);
        const bool need = solve && fail < n;
        const unsigned long long nm = __ballot(need);
#ifdef CN_PHASE_TIMING
        clock.acc[9] += __popcll(nm);
#endif
        if (nm == 0ull) __builtin_amdgcn_s_setprio(0);
        if (nm != 0ull) {  // wave-uniform: some agent of this wave was infeasible
#ifdef CN_WAVE_TRACE
            ++wt_fallbacks;
#endif
            __builtin_amdgcn_s_setprio(3);
            constexpr int kPairs = MAXL * (MAXL - 1) / 2;
            const int n_todo = __popcll(nm);
            bool one_pass_done = false;  // (wave-uniform)
            if (n_todo * kPairs <= kWave) {
                // one pass: item = lane = (t, m); the t-th infeasible agent is the t-th set bit of the ballot (scalar bit
                // tricks, no LDS list), and the item's half-planes are requested once for both stages
                const int t = L.lane / kPairs, m = L.lane - t * kPairs;
                int a = 0;
                unsigned long long rest = nm;
#pragma unroll
                for (int u = 0; u < kWave / kPairs; ++u) {
                    const int bit = rest ? __ffsll((long long)rest) - 1 : 0;
                    a = (u == t) ? bit : a;
                    rest &= rest - 1ull;
                }
                const bool item = L.lane < n_todo * kPairs;
                const int i = lp3_program_of(m), base = i * (i - 1) / 2;
                const float4* la = s.lines + a * kLineStride;
                const float4 li = la[i], lj = la[m - base];
                const float radius = s.sol[a].z;
                const float4 pr = lp3_project(li, lj);
                if (item) s.proj[a * kLineStride + m] = pr;
                CN_FUSED_SYNC();
                if (item) {
                    // (one (projected line, earlier line) pair per item lane + two shuffle rounds instead of these three masked pairs
                    // was built and measured neutral in round 6: 1 229.6 / 1 238.8 vs 1 229.9 / 1 233.7 M — profiles/HISTORY.md)
                    const float4* pa = s.proj + a * kLineStride + base;
                    s.cand3[a * kLineStride + m] = lp_line_candidate<MAXL - 2>(pa[m - base], pa, m - base, radius, -li.w, li.z, true);
                }
                CN_FUSED_SYNC();
                // the four planar programs of an infeasible agent side by side: the item lane of slot (i, 0) runs program i
                // and leaves its solution in the agent's cand2 row (free since the planar scan above), slot i
                if (item && m == base)
                    s.cand2[a * kLineStride + i] = lp3_inner_program(s.proj + a * kLineStride, s.cand3 + a * kLineStride, i, li, radius);
                CN_FUSED_SYNC();
                if (need)
                    lp3_outer_scan(s.lines + L.lane * kLineStride, s.cand2 + L.lane * kLineStride, n, fail, s.sol[L.lane].z, rx, ry);
                one_pass_done = true;
            } else {
                if (need) s.todo[__popcll(nm & ((1ull << L.lane) - 1ull))] = L.lane;
                CN_FUSED_SYNC();
                const int items = n_todo * kPairs;
                for (int p = L.lane; p < items; p += kWave) {  // projections: lane = (agent, i, j)
                    const int t = p / kPairs, m = p - t * kPairs;
                    const int a = s.todo[t];
                    const int i = lp3_program_of(m), j = m - i * (i - 1) / 2;
                    const float4* la = s.lines + a * kLineStride;
                    s.proj[a * kLineStride + m] = lp3_project(la[i], la[j]);
                }
                CN_FUSED_SYNC();
                for (int p = L.lane; p < items; p += kWave) {  // their candidates: lane = (agent, i, k)
                    const int t = p / kPairs, m = p - t * kPairs;
                    const int a = s.todo[t];
                    const int i = lp3_program_of(m), base = i * (i - 1) / 2;
                    const float4 li = s.lines[a * kLineStride + i];
                    const float4* pa = s.proj + a * kLineStride + base;
                    s.cand3[a * kLineStride + m] = lp_line_candidate<MAXL - 2>(pa[m - base], pa, m - base, s.sol[a].z, -li.w, li.z, true);
                }
                CN_FUSED_SYNC();
            }
            if (need && !one_pass_done)
                lp3_scan(s.lines + L.lane * kLineStride, s.proj + L.lane * kLineStride, s.cand3 + L.lane * kLineStride, n,
                         fail, s.sol[L.lane].z, rx, ry);
        }
        CN_TICK(clk, 8);

        // ---- the robot's action reaches every lane of its env (wave shuffle; caller-supplied actions: one load per lane)
        double act_x, act_y;
        if (P.robot_orca) {
            act_x = (double)__shfl(rx, L.ebase);
            act_y = (double)__shfl(ry, L.ebase);
        } else {
            act_x = L.valid ? ext_action[2 * (size_t)L.env] : 0.0;
            act_y = L.valid ? ext_action[2 * (size_t)L.env + 1] : 0.0;
        }
        const double new_vx = (L.a == 0) ? act_x : (double)rx;
        const double new_vy = (L.a == 0) ? act_y : (double)ry;

        // ---- one float64 distance per agent lane (crowd_sim.py:331-351 for a human, :364-366 for the robot)
        if (L.valid) {
            const bool human = L.a > 0;
            const double2 rp = s.posd[L.ebase];
            const double x1 = r.px - rp.x, y1 = r.py - rp.y;
            const double wx = r.vx - act_x, wy = r.vy - act_y;
            const double x2 = x1 + wx * c_dt, y2 = y1 + wy * c_dt;
            const double sx = x2 - x1, sy = y2 - y1;
            double u = ((0.0 - x1) * sx + (0.0 - y1) * sy) / (sx * sx + sy * sy);
            u = (u > 1.0) ? 1.0 : ((u < 0.0) ? 0.0 : u);
            const bool degenerate = (sx == 0.0 && sy == 0.0);  // utils.py:11-13
            const double cx = degenerate ? 0.0 - x1 : (x1 + u * sx) - 0.0;
            const double cy = degenerate ? 0.0 - y1 : (y1 + u * sy) - 0.0;
            const double endx = r.px + new_vx * c_dt, endy = r.py + new_vy * c_dt;
            const double d = norm2(human ? cx : endx - r.gx, human ? cy : endy - r.gy);
            s.closest[L.lane] = human ? d - r.rad - s.rad[L.ebase] : d;
        }
        CN_FUSED_SYNC();
        CN_TICK(clk, 5);

        // ---- reduce (every lane of the env, identically), integrate, episode bookkeeping, stage the next step
        if (running) {
            // LDS requests first: the env's distances, the robot's radius, this step's discount factor
            const double goal_dist = s.closest[L.ebase];
            const double robot_rad = s.rad[L.ebase];
            const double disc_t = s.disc[cur_steps < kMaxDiscount ? cur_steps : kMaxDiscount - 1];
            // the reference stops scanning at the first colliding human (dmin keeps the minimum seen before it)
            double dmin = std::numeric_limits<double>::infinity();
            bool collision = false;
            for (int i = 1; i < P.A; ++i) {
                const double c = s.closest[L.ebase + i];
                const bool hit = c < 0.0;
                dmin = (!collision & !hit & (c < dmin)) ? c : dmin;
                collision = collision | hit;
            }
            // crowd_sim.py:364-389 as a priority chain of selects (timeout > collision > goal > danger > nothing): the
            // same values as the if / elif ladder without its nested branches
            const bool timeout = ep.gtime >= c_limit1;
            const bool reaching = goal_dist < robot_rad;
            const bool danger = dmin < c_ddist;
            double reward = danger ? (dmin - c_ddist) * c_dfactor * c_dt : 0.0;
            int info = danger ? CN_DANGER : CN_NOTHING;
            reward = reaching ? c_success : reward, info = reaching ? CN_REACH_GOAL : info;
            reward = collision ? c_collision : reward, info = collision ? CN_COLLISION : info;
            reward = timeout ? 0.0 : reward, info = timeout ? CN_TIMEOUT : info;
            const bool done = timeout | collision | reaching;
            ep.gtime += c_dt;
            r.px = r.px + new_vx * c_dt;  // Agent.step (agent.py:127-135)
            r.py = r.py + new_vy * c_dt;
            r.vx = new_vx;
            r.vy = new_vy;
            // return / danger accumulators: every lane of the env carries them (only the robot lane's copy is written out)
            ++transitions;
            cur_return = cur_return + (cur_steps < kMaxDiscount ? disc_t : 0.0) * reward;  // python sum(): left to right
            ++cur_steps;
            cur_danger += info == CN_DANGER ? 1 : 0;
            cur_dsum = info == CN_DANGER ? cur_dsum + dmin : cur_dsum;
#ifdef CN_WAVE_TRACE
            if (__ballot(done) != 0ull) ++wt_ends;
#endif
            if (done) {  // explorer.py:50-72: record, then the env's next episode
                const cn_rollout_io io = *iop;
                if (L.a == 0) {
                    if (io.record_capacity > 0) {
                        const size_t k = (size_t)L.env * io.record_capacity + (ep.ep_count % io.record_capacity);
                        if (io.ep_outcome) io.ep_outcome[k] = (uint8_t)info;
                        if (io.ep_steps) io.ep_steps[k] = cur_steps;
                        if (io.ep_return) io.ep_return[k] = cur_return;
                        if (io.ep_time) io.ep_time[k] = (info == CN_TIMEOUT) ? c_limit : ep.gtime;
                        if (io.ep_danger) io.ep_danger[k] = cur_danger;
                        if (io.ep_danger_dmin_sum) io.ep_danger_dmin_sum[k] = cur_dsum;
                    }
                }
                cur_steps = 0, cur_return = 0.0, cur_danger = 0, cur_dsum = 0.0;
                ++ep.ep_count;
                ep.gtime = 0.0;
                const int64_t c = episode_id(io, L.env, ep.ep_count);
                if (io.episode_limit >= 0 && c >= io.episode_limit) {
                    ep.state = kRetired;
                } else if (ep.ep_count < ep.ring_filled) {
                    load_from_ring(P, *Sd, L, ep.ep_count % P.ring_depth, r);
                    theta = 1.5707963267948966;  // robot.set(..., np.pi / 2)
                } else {
                    ep.state = kWaitingScenario;  // ring ran dry: pause this env until the next launch has refilled it
                }
            }
        }
        stage_agent(P, s, L, r, c_hsafety);
        CN_FUSED_SYNC();
        CN_TICK(clk, 7);
    }
#ifdef CN_PHASE_TIMING
    if ((threadIdx.x & (kWave - 1)) == 0) {
        unsigned long long total = 0ull;
        for (int k = 0; k < 10; ++k) {
            atomicAdd(&cn_phase_cycles[k], clock.acc[k]);
            if (k != 9) total += clock.acc[k];
        }
        atomicMax(&cn_phase_cycles[14], total);  // the slowest wave's step loop (what a launch waits for) ...
        atomicMin(&cn_phase_cycles[13], total);  // ... and the fastest one's (reset to ~0ull by the probe script)
        atomicAdd(&cn_phase_cycles[15], 1ull);
    }
#endif

#ifdef CN_WAVE_TRACE
    const unsigned long long wt_exit = __builtin_amdgcn_s_memrealtime();
#endif
    const StateView S = *Sd;
    if (L.valid) {
        S.pos[L.gi] = make_double2(r.px, r.py);
        S.vel[L.gi] = make_double2(r.vx, r.vy);
        S.goal[L.gi] = make_double2(r.gx, r.gy);
        S.rv[L.gi] = make_double2(r.rad, r.vpref);
    }
    const cn_rollout_io io = *iop;
    if (robot) {
        S.gtime[L.env] = ep.gtime;
        S.theta[L.env] = theta;
        if (P.robot_orca) S.rsim_valid[L.env] = 1;
        io.active[L.env] = (uint8_t)ep.state;
        io.ep_count[L.env] = ep.ep_count;
        io.cur_steps[L.env] = cur_steps;
        io.cur_return[L.env] = cur_return;
        if (io.cur_danger) io.cur_danger[L.env] = cur_danger;
        if (io.cur_danger_dmin_sum) io.cur_danger_dmin_sum[L.env] = cur_dsum;
        S.ep_word[L.env] = (ep.ep_count << 2) | ep.state;
    }
    // transitions counter, record blocks, explorer.py:74-90 sums: the launch's own tail (step_kernels.h: rollout_epilogue)
    rollout_epilogue(P, S, io, L, robot, transitions, ep.ep_count, reinterpret_cast<double*>(s.lines));
#ifdef CN_WAVE_TRACE
    if (threadIdx.x == 0 && blockIdx.x < 8192) {
        unsigned long long* w = cn_wave_trace + 6 * blockIdx.x;
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        w[0] = wt_entry, w[1] = wt_loop, w[2] = wt_exit, w[3] = __builtin_amdgcn_s_memrealtime();
        w[4] = wt_fallbacks | (wt_ends << 32), w[5] = hw;
    }
#endif
}

}  // namespace cn
