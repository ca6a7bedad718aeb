// Device side of libcrowdnav_amd.so: the batched CrowdSim transition as a wave-level pipeline.
//
// One workgroup = E whole envs (E * A <= 64 agents) on 1..8 waves.  A transition runs in phases that each use
// the lanes differently (all separated by workgroup barriers, all data exchanged through LDS).  The per-agent
// phases run on wave 0 only; the per-pair phases are spread over every wave of the workgroup, and the waves that
// have nothing to do in a phase sleep at the barrier without taking issue slots from other workgroups:
//   stage      lane = agent            float32 view of the agents as rvo2 is told them (orca.py:100-110)
//   pairs-1    lane = (agent, cand.)   squared distance of every ordered pair            (Appendix A.2)
//   pairs-2    lane = (agent, cand.)   stable rank among the agent's candidates -> neighbour slot; the
//                                      ORCA half-plane of that pair -> LDS float4 slot   (Appendix A.3)
//   solve      lane = agent            half-planes -> VGPRs, unrolled 2-D program, rare 3-D fallback
//   collide    lane = human            float64 swept robot-human distance (crowd_sim.py:331-351)
//   reduce     lane = robot            reward / done / info (crowd_sim.py:364-389)
//   integrate  lane = agent            Agent.step (agent.py:127-135)
// The chip is issue/latency-bound on this path (4096 envs x 6 agents = 24.6 k agents on 1024 SIMDs): E and the
// workgroup size are tuning knobs chosen on the host (crowdnav_amd.hip: pick_geometry).
#pragma once
#include <hip/hip_runtime.h>
#include <limits>

#include "../../include/crowdnav_amd.h"
#include "orca_device.h"
#include "kd_order.h"
#include "scenario_device.h"
#include "scenario_wave.h"

namespace cn {

// How the 10-half-plane kernels (crowds of 6+ humans) solve an agent's ORCA program — what was measured on the way is
// profiles/HISTORY.md: the planar program runs on three lanes per agent (orca_device.h: lp_planar_tri), RVO2's 3-D fallback for
// the infeasible agents as lp_relaxed_lazy (one-wave workgroups with room for its candidate rows in d2) or lp_relaxed_coop
// (several waves per workgroup, tiny crowds).  The 5-half-plane kernels solve in candidate form (lp_line_candidate /
// lp_planar_scan, lp3_project / lp3_scan).

constexpr int kMaxBlock = 512;  // threads per workgroup of the transition kernels (1..8 waves)

struct Params {
    int B, A;          // envs, agents per env
    int E;             // envs per workgroup (their E * A agents are lanes of wave 0)
    int threads;       // workgroup size (= blockDim.x of the transition kernels; read from HERE in the step loop: blockDim.x
                       // is a load from the dispatch packet — a global load and a wait per use): wave 0 runs the per-agent
                       // phases, every wave the per-pair phases
    int nA;            // E * A agent lanes
    int NC;            // A - 1 candidate neighbours per agent
    int pairs;         // nA * NC
    int ring_depth;    // scenarios kept ahead per env
    int robot_visible, robot_orca;
    int robot_unicycle;  // external robot actions are ActionRot(v, r) (agent.py:115-135)
    int async_fill;      // CN_FLAG_ASYNC_SCENARIO_FILL: ring slots are published one by one (StateView::ring_ready)
    int sched;           // the 20-human shard's kernel: sub-launch 0..3 of the 3-of-4 env schedule (launch_rollout), -1 = all envs,
                         // kSchedDynamic = persistent workgroups taking (env, visit) items from a device queue
    int dyn_visits;      // ... visits per env and call under the dynamic schedule (launch_rollout: ~56 steps per visit)
    int kd;              // A > 10: some rvo2 simulator of an env holds more than 10 agents and splits its kd-tree (kd_order.h)
    KdLayout kdl;        // ... and where its bookkeeping lives in LDS (offsets from Smem::kd_off)
    double dt, time_limit, success_reward, collision_penalty, discomfort_dist, discomfort_factor;
    double robot_safety, human_safety;
    OrcaParams orca;
};

struct StateView {
    double2* pos;
    double2* vel;
    double2* goal;
    double2* rv;  // (radius, v_pref)
    double* gtime;
    double* theta;          // [B] robot heading (only a unicycle robot changes it)
    float* rsim_radius;     // [B*A] radii captured by the robot's persistent ORCA policy (orca.py:98-104)
    float* rsim_max_speed;  // [B]
    uint8_t* rsim_valid;    // [B]
    uint32_t* mt_key;       // [624][B]   generator state of the env's own stream
    uint32_t* ring_mt_key;  // [624][kRedoLanes] HBM generator scratch of ring_redo_kernel (scenarios the head generator gave up on)
    int2* redo_list;        // [B*D] (ring index, seed) of those scenarios; redo_count [1] = how many (zeroed before a fill)
    int* redo_count;
    int* mt_pos;            // [B]
    // scenario ring: the next ring_depth episodes of every env, generated ahead of the rollout
    double2* ring_pos;      // [B][D][A]
    double2* ring_goal;
    double2* ring_rv;
    int* ring_filled_in;    // [B] episode ordinals < this have been generated (read side)
    int* ring_filled_out;   // [B] (written by the fill kernel; the host swaps the two)
    int* ring_ready;        // [B*D] async fill: ordinal + 1 of the scenario a slot holds, stored with release once it is complete
    int* ring_claim;        // [B*D] async fill: ordinal + 1 some fill launch is generating (or has generated) for the slot
    uint8_t* kd_order;      // [B*A][kd_row_bytes(A)] the permutation each agent's rvo2 simulator partitions in place (kd_order.h)
    uint8_t* kd_valid;      // [B*A] 0 = a freshly built simulator (identity order)
    // scenario cache of the wave generators (cached_scenario): a rollout whose episode seeds come from a SMALL set (seed_mod <=
    // kScenarioCacheMax: the reference's 'val' / 'test' phases replay 100 / 500 fixed cases) generates every scenario once
    double2* cache_pos;     // [kScenarioCacheMax][A]
    double2* cache_goal;
    double2* cache_rv;
    int* cache_state;       // [kScenarioCacheMax] 0 = empty, 1 = complete (release / acquire at agent scope)
    int cache_n;            // seeds cached by the current rollout (io.seed_mod), 0 = off
    int* error;             // [1] the engine's error word (ScenarioCfg::error): bit 2 = a visit of the dynamic schedule gave up waiting
    int* dyn_queue;         // [1 + B] dynamic schedule of the shard kernel: [0] next (env, visit) item, [1 + env] visits env has COMPLETED
    int* ep_word;           // [B] (episodes finished << 2) | io.active state, ONE word stored by the rollout kernels next to the
                            // two io arrays: the asynchronous fill reads it for a consistent (state, ep_count) snapshot
    // launch epilogue of the rollout kernels (rollout_epilogue): arrival tickets and partial sums
    double* wg_partial;     // [workgroups][CN_SUMMARY_FIELDS + 1] per-workgroup sums over its envs' record rings, its transitions
    double* group_partial;  // [kEpilogueGroups][CN_SUMMARY_FIELDS + 1]
    unsigned* tickets;      // [(kEpilogueGroups + 1) * kTicketStride] arrivals per group, then of the group leaders, one
                            // counter per 256-byte line; zero between launches
};

constexpr int kEpilogueGroups = 32;  // workgroup b arrives at counter b % 32: 64 arrivals per counter at 2048 workgroups
constexpr int kTicketStride = 64;    // counters 256 bytes apart: same-line atomics serialise at ~10 ns each

struct StepIo {
    const double* action;
    double* reward;
    uint8_t* done;
    uint8_t* info;
    double* dmin;
    double* action_out;
    float* orca_vel;
    double* obs;
    int update;
};

// ---------------------------------------------------------------------------------------------- LDS carve-up
struct Smem {
    float4* kin;      // [nA + 1] float32(px, py, vx, vy); slot nA = (+inf, +inf, 0, 0): the "candidate" of a pair that does not exist
    double2* posd;    // [nA] float64 position
    double2* act;     // [nA] (robot lanes) applied robot action
    float4* lines;    // [nA][kLineStride] ORCA half-planes, slot = neighbour rank
    float4* proj;     // [nA][kLineStride] projected half-planes (3-D fallback scratch)
    float4* cand2;    // [nA][kLineStride] candidate form: 1-D solutions of the agent's half-planes (lp_line_candidate)
    float4* cand3;    // [nA][kLineStride] ... and of its projected half-planes (3-D fallback)
    double* rad;      // [nA] float64 radius
    double* closest;  // [nA] (human lanes) closest boundary distance during the step
    float* hview;     // [nA] radius as a human's rvo2 sim holds it: float32(radius + 0.01 + human_safety)
    float* rview;     // [nA] radius as the robot's rvo2 sim holds it (captured, see load_robot_view)
    float* d2;        // [pairs] squared distance agent -> candidate (+inf if the pair does not exist)
    int* pinfo;       // [pairs] packed pair descriptor
    int* count;       // [nA] neighbours kept
    int* flag;        // [nA] (robot lanes) per-env flag broadcast
    float4* sol;      // [nA] lane-cooperative program input: (pref.x, pref.y, maxSpeed, solve ? 1 : 0)
    float4* res;      // [nA] ... and output: (result.x, result.y, first infeasible line or n, -)
    int* todo;        // [nA + 1] agents that need the 3-D fallback, compacted; [nA] = how many
    double* disc;     // [kMaxDiscount] discount table gamma^(t dt v_pref) (rollout kernel only)
    uint32_t kd_off;  // byte offset (from the start of LDS) of the kd-tree region, simulators of more than 10 agents: kd_view
    double2* goal2;   // [nA] COMPACT only: the agent's goal ...
    double* vpref;    // [nA] ... and preferred speed: per-episode constants the step loop does not hold in registers
};

constexpr int kCompactScratchDoubles = 16;  // COMPACT: Smem::disc holds the step parameters and the robot lane's bookkeeping instead
constexpr int kMaxDiscount = 256;  // steps per episode the LDS copy of the discount table covers
__host__ __device__ inline size_t proj_bytes(int nA, int /*maxl*/) { return (size_t)16 * kLineStride * nA; }

// maxl: the 5-half-plane kernels have two candidate-form buffers (cand2, cand3); the 10-half-plane kernels none (their lazy
// fallback's candidate rows live in d2)
__host__ __device__ inline size_t smem_bytes(int nA, int pairs, int maxl, int A = 0, int E = 1) {
    return (size_t)nA * (16 + 16 + 16 + (maxl == 5 ? 3 : 1) * 16 * kLineStride + 16 + 16 + 8 + 8 + 4 + 4 + 4 + 4 + 4) + proj_bytes(nA, maxl) +
           (size_t)pairs * 8 + 64 + 8 + 16 + 16 + sizeof(double) * kMaxDiscount + (A > kKdLeaf ? 16 + kd_lds_bytes(nA, A, E) : 0);
}

// COMPACT: the LDS layout of the 20-human shard's kernel (rollout_kernel<10, false, true, true>: one env per one-wave
// workgroup, lazy fallback, two-sweep pair phase).  18.6 KB per workgroup allowed 8 workgroups = 2 waves per SIMD whatever the
// register count; a third resident wave needs <= 12.8 KB (160 KB in 1280-byte granules: 10 granules x 12 workgroups).  What goes:
//   disc   2048 B  the discount table: one global load per step on the robot lane, issued a step ahead of its use
//   proj   3696 -> 1008 B  rows of the lazy fallback belong to the 7 lane groups of a pass, not to the 21 agents (the `kept`
//                  table of the two-sweep pair phase, 840 B, shares them as before)
//   act     336 -> 16 B    one robot per workgroup
//   pinfo  1680 -> 840 B   16-bit pair descriptors (the candidate slot is p - 20 q)
//   kd visit order + traversal stacks  1008 -> 0 B   needed only while exact distance ties are resolved, when `lines` holds
//                  nothing that is not recomputed afterwards (kd_resolve_ties re-runs the half-plane sweep): kd_scratch_view
// and what comes in: the step parameters, the robot lane's episode bookkeeping (128 B), the agents' goals and preferred
// speeds (504 B) — values the step loop would otherwise hold in VGPRs across every phase (third resident wave per SIMD).
constexpr int kLazyRows = (kWave / (kMaxNb - 1)) * (kMaxNb - 1);  // 63 float4: 7 groups x 9 projected half-planes
__host__ __device__ inline size_t smem_bytes_compact(int nA, int pairs, int A, int E) {
    size_t n = (size_t)16 * (nA + 1) + 16 * nA + 16 * E + (size_t)16 * kLineStride * nA + 16 * kLazyRows;
    n += (size_t)nA * (16 + 16 + 16 + 8 + 8 + 8 + 4 + 4 + 4 + 4) + 4 * (nA + 1) + 16;
    n += (size_t)pairs * 4 + (size_t)pairs * 2 + 16 + 16 + 8 * kCompactScratchDoubles + (A > kKdLeaf ? 16 + kd_lds_bytes(nA, A, E) : 0);
    // the traversal scratch of the tie resolver (visit order + stacks, the LAST two arrays of the kd region) lives in `lines`
    if (A > kKdLeaf) n -= (size_t)nA * kd_row_bytes(A) + (size_t)nA * (kd_max_nodes(A) + 1) * 2;
    return n;
}

template <int MAXL, bool COMPACT = false>
__device__ __forceinline__ Smem carve(const Params& P) {
    extern __shared__ double2 smem_raw[];
    Smem s;
    char* p = reinterpret_cast<char*>(smem_raw);
    const int nA = P.nA;
    s.kin = reinterpret_cast<float4*>(p), p += 16 * (nA + 1);
    s.posd = reinterpret_cast<double2*>(p), p += 16 * nA;
    s.act = reinterpret_cast<double2*>(p), p += 16 * (COMPACT ? P.E : nA);
    s.lines = reinterpret_cast<float4*>(p), p += 16 * kLineStride * nA;
    s.proj = reinterpret_cast<float4*>(p), p += COMPACT ? (size_t)16 * kLazyRows : proj_bytes(nA, MAXL);
    s.cand2 = s.cand3 = nullptr;
    if (MAXL == 5) {
        s.cand2 = reinterpret_cast<float4*>(p), p += 16 * kLineStride * nA;
        s.cand3 = reinterpret_cast<float4*>(p), p += 16 * kLineStride * nA;
    }
    s.sol = reinterpret_cast<float4*>(p), p += 16 * nA;
    s.res = reinterpret_cast<float4*>(p), p += 16 * nA;
    s.goal2 = nullptr, s.vpref = nullptr;
    if (COMPACT) {
        s.goal2 = reinterpret_cast<double2*>(p), p += 16 * nA;
        s.vpref = reinterpret_cast<double*>(p), p += 8 * nA;
    }
    s.rad = reinterpret_cast<double*>(p), p += 8 * nA;
    s.closest = reinterpret_cast<double*>(p), p += 8 * nA;
    s.hview = reinterpret_cast<float*>(p), p += 4 * nA;
    s.rview = reinterpret_cast<float*>(p), p += 4 * nA;
    s.count = reinterpret_cast<int*>(p), p += 4 * nA;
    s.flag = reinterpret_cast<int*>(p), p += 4 * nA;
    s.todo = reinterpret_cast<int*>(p), p += 4 * (nA + 1);
    p += (16 - (reinterpret_cast<size_t>(p) & 15)) & 15;  // rows of d2 are read as float4 when NC is a multiple of 4
    s.d2 = reinterpret_cast<float*>(p), p += 4 * P.pairs;
    s.pinfo = reinterpret_cast<int*>(p), p += (COMPACT ? 2 : 4) * P.pairs;
    s.disc = reinterpret_cast<double*>(p + ((8 - (reinterpret_cast<size_t>(p) & 7)) & 7));
    {
        char* q = reinterpret_cast<char*>(s.disc + (COMPACT ? kCompactScratchDoubles : kMaxDiscount));
        q += (16 - (reinterpret_cast<size_t>(q) & 15)) & 15;
        s.kd_off = (uint32_t)(q - reinterpret_cast<char*>(smem_raw));
    }
    return s;
}

// COMPACT (the 20-human shard's kernel): the seven float64 parameters of a step are not held in registers across the step
// loop (14 VGPRs) but kept in LDS — Smem::disc[0..6], written once per launch — and requested where a phase needs them;
// Smem::disc[8..13] hold the robot lane's episode bookkeeping between steps (EpisodeLds).
enum { kParDt = 0, kParLimit, kParSuccess, kParCollision, kParDDist, kParDFactor, kParHSafety, kParCount };
template <bool COMPACT>
__device__ __forceinline__ double step_param(const Params& P, const Smem& s, int which) {
    if (COMPACT) return s.disc[which];
    return which == kParDt ? P.dt : which == kParLimit ? P.time_limit : which == kParSuccess ? P.success_reward :
           which == kParCollision ? P.collision_penalty : which == kParDDist ? P.discomfort_dist :
           which == kParDFactor ? P.discomfort_factor : P.human_safety;
}
struct EpisodeLds {  // at Smem::disc + 8 (one robot per workgroup)
    double gtime, cur_return, cur_dsum;
    int cur_steps, cur_danger, ep_count, ring_filled, state;
    unsigned int transitions;
    unsigned int dyn_total;  // dynamic schedule: transitions of all the visits this workgroup ran (not a register held across them)
};
static_assert(sizeof(EpisodeLds) <= 8 * (kCompactScratchDoubles - 8), "EpisodeLds outgrew its LDS slot");

// Workgroup barrier of the step / rollout kernels.  A workgroup of ONE wave (P.threads == 64: a compile-time fact in the
// 20-human shard's instantiation, a uniform branch elsewhere) needs no s_barrier: a wave's LDS instructions execute in order,
// so between one lane's write and another lane's read only the compiler has to keep program order — and the s_waitcnt
// lgkmcnt(0) in front of every s_barrier (the LDS queue drained a dozen times per step) goes away.
__device__ __forceinline__ void block_sync(const Params& P) {
    if (P.threads == kWave) wave_lds_sync();
    else __syncthreads();
}
__device__ __forceinline__ int block_sync_or(const Params& P, int v) {
    if (P.threads == kWave) return __ballot(v != 0) != 0ull ? 1 : 0;
    return __syncthreads_or(v);
}

// ---------------------------------------------------------------------------------------------- lanes
struct Lane {
    int lane, env, a, ebase;  // ebase = lane of this env's robot
    bool valid;               // this lane owns an agent of an existing env
    size_t gi;                // env * A + a
};

__device__ __forceinline__ Lane lane_of(const Params& P, int block = -1, int tid = -1) {
    Lane L;
    L.lane = tid < 0 ? (int)threadIdx.x : tid;
    const int el = L.lane / P.A;
    L.a = L.lane - el * P.A;
    L.env = (block < 0 ? (int)blockIdx.x : block) * P.E + el;
    L.valid = (L.lane < P.nA) && (L.env < P.B);
    L.ebase = el * P.A;
    L.gi = (size_t)L.env * P.A + L.a;
    return L;
}

struct AgentRegs {
    double px, py, vx, vy, gx, gy, rad, vpref;
};

__device__ __forceinline__ void load_agent(const StateView& S, size_t gi, AgentRegs& r) {
    const double2 p = S.pos[gi], v = S.vel[gi], g = S.goal[gi], q = S.rv[gi];
    r.px = p.x, r.py = p.y, r.vx = v.x, r.vy = v.y, r.gx = g.x, r.gy = g.y, r.rad = q.x, r.vpref = q.y;
}

// Pair descriptors (agent lane q, candidate slot c): built once per launch.  Candidate order = the order
// ORCA.predict adds the others to its rvo2 sim: the other humans by index, then the robot if it is visible
// (crowd_sim.py:325-327, orca.py:102-104); the robot's own sim holds every human.
//   bits 0-7 agent lane, 8-15 lane of the candidate, 16-23 candidate slot, 24 pair exists, 25 agent is a robot
// COMPACT: 16 bits per pair — bits 0-4 agent lane, 5-9 lane of the candidate, 10 pair exists, 11 agent is a robot; the
// candidate slot is p - NC * agent lane.  pair_info() hands out the 32-bit form either way.
template <bool COMPACT>
__device__ __forceinline__ int pair_info(const Params& P, const Smem& s, int p) {
    if (!COMPACT) return s.pinfo[p];
    const int w = reinterpret_cast<const uint16_t*>(s.pinfo)[p];
    const int q = w & 31;
    return q | (((w >> 5) & 31) << 8) | ((p - q * P.NC) << 16) | (((w >> 10) & 3) << 24);
}

template <bool COMPACT = false>
__device__ __forceinline__ void build_pairs(const Params& P, const Smem& s, bool every_env_exists = false) {
    for (int p = threadIdx.x; p < P.pairs; p += P.threads) {
        const int q = p / P.NC;
        const int c = p - q * P.NC;
        const int el = q / P.A;
        const int a = q - el * P.A;
        int j;
        bool exists = every_env_exists || ((int)blockIdx.x * P.E + el) < P.B;
        if (a == 0) {
            j = c + 1;
        } else {
            j = c + 1 + (c + 1 >= a ? 1 : 0);
            if (j >= P.A) {
                j = 0;
                exists = exists && P.robot_visible;
            }
        }
        if (COMPACT)
            reinterpret_cast<uint16_t*>(s.pinfo)[p] = (uint16_t)(q | ((el * P.A + j) << 5) | (exists ? 1 << 10 : 0) | (a == 0 ? 1 << 11 : 0));
        else
            s.pinfo[p] = q | ((el * P.A + j) << 8) | (c << 16) | (exists ? 1 << 24 : 0) | (a == 0 ? 1 << 25 : 0);
    }
}

// The robot's ORCA policy object outlives episodes and keeps the radii / max speed it saw when its rvo2
// simulator was first built (orca.py:95-104).  Load or capture them.
//   preloaded: the caller requested rsim_valid / rsim_radius / rsim_max_speed earlier (sarl_decide_step_kernel, at its top)
__device__ __forceinline__ void load_robot_view(const Params& P, const StateView& S, const Smem& s,
                                                const Lane& L, const AgentRegs& r, float& robot_max_speed, bool preloaded = false,
                                                bool pre_have = false, float pre_radius = 0.0f, float pre_max_speed = 0.0f) {
    robot_max_speed = 0.0f;
    if (!L.valid) return;
    const bool have = preloaded ? pre_have : S.rsim_valid[L.env] != 0;
    float rr;
    if (have) {
        rr = preloaded ? pre_radius : S.rsim_radius[L.gi];
    } else {
        rr = (float)(r.rad + 0.01 + P.robot_safety);
        S.rsim_radius[L.gi] = rr;
    }
    s.rview[L.lane] = rr;
    if (L.a == 0) {
        if (have) {
            robot_max_speed = preloaded ? pre_max_speed : S.rsim_max_speed[L.env];
        } else {
            robot_max_speed = (float)r.vpref;
            S.rsim_max_speed[L.env] = robot_max_speed;
        }
    }
}

// ---------------------------------------------------------------------------------------------- kd-tree order (kd_order.h)
// The kd region's pointers: region offset + the host-computed layout in the kernel arguments (KdLayout).
__device__ __forceinline__ KdSmem kd_view(const Params& P, const Smem& s) {
    extern __shared__ double2 smem_raw[];
    return kd_carve(reinterpret_cast<char*>(smem_raw) + s.kd_off, P.kdl);
}

// COMPACT: the tie resolver's traversal scratch is carved out of `lines` (see carve)
template <bool COMPACT>
__device__ __forceinline__ KdSmem kd_scratch_view(const Params& P, const Smem& s) {
    KdSmem k = kd_view(P, s);
    if (COMPACT) {
        k.visit = reinterpret_cast<uint8_t*>(s.lines);
        k.stack = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(s.lines) + (size_t)P.nA * k.row);
    }
    return k;
}

// A launch keeps the permutations of its envs' simulators in LDS: loaded here (or built fresh), stored by kd_store.
__device__ __forceinline__ void kd_load(const Params& P, const StateView& S, const Smem& s, const Lane& L) {
    if (!P.kd) return;
    const KdSmem k = kd_view(P, s);
    if (L.lane < P.nA) {
        uint8_t* row = k.ord + (size_t)L.lane * k.row;
        if (L.valid && S.kd_valid[L.gi] != 0) {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(S.kd_order + L.gi * k.row);
            for (int w = 0; w < k.row / 4; ++w) reinterpret_cast<uint32_t*>(row)[w] = src[w];
        } else {
            kd_identity_row(row, k.row, P.A, L.a, P.robot_visible);
        }
    }
    if (threadIdx.x == 0) *k.gen = 0;
    for (int i = threadIdx.x; i < P.E * 2 * 2; i += P.threads) k.count[i] = 0;  // no "last step's tree" yet
}

__device__ __forceinline__ void kd_store(const Params& P, const StateView& S, const Smem& s, const Lane& L) {
    if (!P.kd || !L.valid) return;
    const KdSmem k = kd_view(P, s);
    const uint32_t* row = reinterpret_cast<const uint32_t*>(k.ord + (size_t)L.lane * k.row);
    uint32_t* dst = reinterpret_cast<uint32_t*>(S.kd_order + L.gi * k.row);
    for (int w = 0; w < k.row / 4; ++w) dst[w] = row[w];
    S.kd_valid[L.gi] = 1;
}

// A new episode: every Human is rebuilt and with it its ORCA policy and simulator (crowd_sim.py:155-207); the robot's
// policy object, hence its simulator, lives on.  Called by the lanes of an env that has just loaded its next scenario.
__device__ __forceinline__ void kd_new_episode(const Params& P, const Smem& s, const Lane& L) {
    if (!P.kd) return;
    const KdSmem k = kd_view(P, s);
    if (L.a > 0) kd_identity_row(k.ord + (size_t)L.lane * k.row, k.row, P.A, L.a, P.robot_visible);
    if (L.a == 0) {  // nothing of this env's last tree says anything about the fresh rows
        const int last = *k.gen, el = L.ebase / P.A;
        k.count[(last * P.E + el) * 2 + 0] = 0;
        k.count[(last * P.E + el) * 2 + 1] = 0;
    }
}

// The tree(s) of every env of the workgroup from this step's float32 positions (s.kin), breadth first on agent sets: the
// nodes that split, in `generation` g of the node lists.  All threads (barriers inside).
__device__ __forceinline__ void kd_build_trees(const Params& P, const Smem& s, const Lane& L, int g) {
    const KdSmem k = kd_view(P, s);
    const int tid = threadIdx.x;
    const bool agent = L.lane < P.nA;
    const int el = agent ? L.ebase / P.A : 0;
    const float4 me = agent ? s.kin[L.lane] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const uint32_t kx = kd_key(me.x), ky = kd_key(me.y);
    const uint64_t amask = P.A >= 64 ? ~0ull : ((1ull << P.A) - 1ull);
    for (int t = 0; t < 2; ++t) {
        KdNode* list = k.nodes + ((size_t)(g * P.E + el) * 2 + t) * k.mn;
        int* cnt = k.count + (g * P.E + el) * 2 + t;
        if (!kd_tree_on(P.A, t, P.robot_visible)) {  // uniform
            if (agent && L.a == 0) *cnt = 0;
            continue;
        }
        for (int i = tid; i < P.E * k.mn; i += P.threads) {
            k.bb[4 * i + 0] = 0xffffffffu, k.bb[4 * i + 1] = 0u, k.bb[4 * i + 2] = 0xffffffffu, k.bb[4 * i + 3] = 0u;
        }
        const KdNode* old = k.nodes + ((size_t)((g ^ 1) * P.E + el) * 2 + t) * k.mn;
        const int n_old = agent ? k.count[((g ^ 1) * P.E + el) * 2 + t] : 0;
        int first_dirty = 1 << 20;
        if (agent && L.a == 0) {  // the env's robot lane keeps the books (whether or not the robot is in this tree)
            list[0].meta = (uint32_t)kd_tree_size(P.A, t) << 8;
            list[0].set = t == 0 ? amask : (amask & ~1ull);
            *cnt = 1;
        }
        block_sync(P);
        for (int r = 0;; ++r) {
            const int c = agent ? *cnt : 0;
            const bool live = agent && r < c;
            if (!block_sync_or(P, live ? 1 : 0)) break;
            KdNode nd = KdNode{0u, 0u, 0ull, 0ull, 0ull};
            bool member = false;
            uint32_t* acc = k.bb + ((size_t)el * k.mn + r) * 4;
            if (live) {
                nd = list[r];
                member = (t == 0 || L.a > 0) && ((nd.set >> L.a) & 1ull) != 0ull;
                if (member) {
                    atomicMin(acc + 0, kx), atomicMax(acc + 1, kx);
                    atomicMin(acc + 2, ky), atomicMax(acc + 3, ky);
                }
            }
            block_sync(P);
            bool lower = false;
            if (live) {
                const float min_x = kd_unkey(acc[0]), max_x = kd_unkey(acc[1]), min_y = kd_unkey(acc[2]), max_y = kd_unkey(acc[3]);
                const bool vertical = (max_x - min_x) > (max_y - min_y);
                const float split = vertical ? 0.5f * (max_x + min_x) : 0.5f * (max_y + min_y);
                lower = member && (vertical ? me.x : me.y) < split;
            }
            const unsigned long long ballot = __ballot(lower);
            if (live && L.a == 0) {
                const uint64_t lm = (ballot >> L.ebase) & amask;
                const int begin = nd.meta & 0xff, end = (nd.meta >> 8) & 0xff, nl = __popcll(lm);
                const bool degenerate = nl == 0;
                const uint32_t meta = (uint32_t)begin | ((uint32_t)end << 8) | ((uint32_t)nl << 16) | (degenerate ? 1u << 24 : 0u);
                list[r].meta = meta;
                list[r].left = lm;
                if (first_dirty > r && (r >= n_old || old[r].meta != meta || old[r].left != lm)) first_dirty = r;
                int n = c;
                if (!degenerate && nl > kKdLeaf && n < k.mn) {
                    list[n].meta = (uint32_t)begin | ((uint32_t)(begin + nl) << 8);
                    list[n].set = lm;
                    ++n;
                }
                if (!degenerate && end - begin - nl > kKdLeaf && n < k.mn) {
                    list[n].meta = (uint32_t)(begin + nl) | ((uint32_t)end << 8);
                    list[n].set = nd.set & ~lm;
                    ++n;
                }
                *cnt = n;
                k.dirty[el * 2 + t] = first_dirty;
            }
        }
    }
}

// The same for the usual geometry of these crowds — one env per single-wave workgroup: every quantity of a node is uniform
// over the wave, so the bounding box is four DPP wave reductions, the lower side a ballot, and the control flow scalar: no LDS
// atomics, no barriers (the cooperative version above spent ~1 400 clock ticks per node on them).
// Built every step although 84 % of the steps (20 humans) leave the tree as it was.  Proving "unchanged" first — the four
// agents that attained a node's box last time still bound its members, every member still on its side of the new split: one
// vector pass over (node, agent) lanes, nearly free — was built and measured: the 29 % of steps that then DID rebuild paid
// 12 900 clock ticks each instead of 3 100, because code a wave executes once in a while is not in the instruction cache
// when it comes (the step loop of this kernel is ~45 KB by itself), 3 780 vs 3 100 ticks per step on average.  What the
// builder does hand on is the first node whose record differs from last step's: in unchanged steps the simulator lanes do
// nothing.  Returns the generation that holds this step's lists.
__device__ __forceinline__ int kd_build_trees_wave(const Params& P, const Smem& s, const Lane& L, int g_last) {
    const KdSmem k = kd_view(P, s);
    const bool agent = L.lane < P.nA;  // = L.lane < P.A: one env
    const float4 me = agent ? s.kin[L.lane] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const uint32_t kx = kd_key(me.x), ky = kd_key(me.y);
    const uint64_t amask = P.A >= 64 ? ~0ull : ((1ull << P.A) - 1ull);
    const int g = g_last ^ 1;
    for (int t = 0; t < 2; ++t) {
        KdNode* list = k.nodes + ((size_t)g * 2 + t) * k.mn;
        const KdNode* old = k.nodes + ((size_t)g_last * 2 + t) * k.mn;
        int* cnt = k.count + g * 2 + t;
        if (!kd_tree_on(P.A, t, P.robot_visible)) {
            if (L.lane == 0) *cnt = 0, k.dirty[t] = 0;
            continue;
        }
        const int n_old = k.count[g_last * 2 + t];
        int first_dirty = 1 << 20, n = 1;
        if (L.lane == 0) {
            list[0].meta = (uint32_t)kd_tree_size(P.A, t) << 8;
            list[0].set = t == 0 ? amask : (amask & ~1ull);
        }
        for (int r = 0; r < n; ++r) {  // n, r, the node and everything derived from it are wave-uniform
            const KdNode nd = list[r];
            const bool had = r < n_old;
            const KdNode od = old[had ? r : 0];
            const bool member = agent && ((nd.set >> L.lane) & 1ull) != 0ull;
            uint32_t b0 = member ? kx : 0xffffffffu, b1 = member ? kx : 0u, b2 = member ? ky : 0xffffffffu, b3 = member ? ky : 0u;
            kd_wave_bbox(b0, b1, b2, b3);
            const float min_x = kd_unkey(b0), max_x = kd_unkey(b1), min_y = kd_unkey(b2), max_y = kd_unkey(b3);
            const bool vertical = (max_x - min_x) > (max_y - min_y);
            const float split = vertical ? 0.5f * (max_x + min_x) : 0.5f * (max_y + min_y);
            const uint64_t lm = __ballot(member && (vertical ? me.x : me.y) < split) & amask;
            const int begin = nd.meta & 0xff, end = (nd.meta >> 8) & 0xff, nl = __popcll(lm);
            const bool degenerate = nl == 0;
            const uint32_t meta = (uint32_t)begin | ((uint32_t)end << 8) | ((uint32_t)nl << 16) | (degenerate ? 1u << 24 : 0u);
            if (first_dirty > r && (!had || od.meta != meta || od.left != lm)) first_dirty = r;
            const bool lo = !degenerate && nl > kKdLeaf && n < k.mn;
            const bool hi = !degenerate && end - begin - nl > kKdLeaf && n + (lo ? 1 : 0) < k.mn;
            if (L.lane == 0) {
                list[r].meta = meta;
                list[r].left = lm;
                if (lo) {
                    list[n].meta = (uint32_t)begin | ((uint32_t)(begin + nl) << 8);
                    list[n].set = lm;
                }
                if (hi) {
                    list[n + (lo ? 1 : 0)].meta = (uint32_t)(begin + nl) | ((uint32_t)end << 8);
                    list[n + (lo ? 1 : 0)].set = nd.set & ~lm;
                }
            }
            n += (lo ? 1 : 0) + (hi ? 1 : 0);
        }
        if (L.lane == 0) *cnt = n, k.dirty[t] = first_dirty;
    }
    return g;
}

// Every simulator's permutation follows this step's tree (lane = the simulator's own agent).  A node whose record equals
// last step's, under ancestors that did not move anything either, is partitioned already.
__device__ __forceinline__ void kd_update_orders(const Params& P, const Smem& s, const Lane& L, int g) {
    const KdSmem k = kd_view(P, s);
    if (L.lane >= P.nA) return;
    const int el = L.ebase / P.A, t = kd_tree_of(L.a, P.robot_visible);
    if (!kd_tree_on(P.A, t, P.robot_visible)) return;
    const KdNode* cur = k.nodes + ((size_t)(g * P.E + el) * 2 + t) * k.mn;
    const KdNode* old = k.nodes + ((size_t)((g ^ 1) * P.E + el) * 2 + t) * k.mn;
    const int n_cur = k.count[(g * P.E + el) * 2 + t], n_old = k.count[((g ^ 1) * P.E + el) * 2 + t];
    uint8_t* row = k.ord + (size_t)L.lane * k.row;
    (void)old, (void)n_old;
    // every node from the first one whose record differs from last step's (the builder found it): conservative — a later node
    // of another subtree is partitioned already and is left as it is by a second partition
    for (int r = k.dirty[el * 2 + t]; r < n_cur; ++r) {
        const KdNode c = cur[r];
        if (((c.meta >> 24) & 1u) != 0u) continue;
        if (P.A <= 32)
            kd_partition32(row, c.meta & 0xff, (c.meta >> 8) & 0xff, (c.meta >> 16) & 0xff, (uint32_t)c.left);
        else
            kd_partition(row, c.meta & 0xff, (c.meta >> 8) & 0xff, (c.meta >> 16) & 0xff, c.left);
    }
}

// Position of every agent of this lane's simulator in RVO2's traversal of its tree (queryAgentTreeRecursive without the
// pruning, which only drops candidates that would be rejected): nearer child first, a leaf in permutation order.  Serial on
// the lane; runs only for simulators with an exact distance tie.
template <bool COMPACT = false>
__device__ __forceinline__ void kd_visit_order(const Params& P, const Smem& s, const Lane& L, int g) {
    const KdSmem k = kd_scratch_view<COMPACT>(P, s);
    const int el = L.ebase / P.A, t = kd_tree_of(L.a, P.robot_visible);
    const KdNode* cur = k.nodes + ((size_t)(g * P.E + el) * 2 + t) * k.mn;
    const int n_cur = kd_tree_on(P.A, t, P.robot_visible) ? k.count[(g * P.E + el) * 2 + t] : 0;
    const uint8_t* row = k.ord + (size_t)L.lane * k.row;
    uint8_t* vis = k.visit + (size_t)L.lane * k.row;
    uint16_t* st = k.stack + (size_t)L.lane * (k.mn + 1);
    const float4 me = s.kin[L.lane];
    auto box_dist = [&](int b, int e) {
        float4 q = s.kin[L.ebase + row[b]];
        float min_x = q.x, max_x = q.x, min_y = q.y, max_y = q.y;
        for (int p = b + 1; p < e; ++p) {
            q = s.kin[L.ebase + row[p]];
            max_x = fmaxf(max_x, q.x), min_x = fminf(min_x, q.x);
            max_y = fmaxf(max_y, q.y), min_y = fminf(min_y, q.y);
        }
        const float a0 = fmaxf(0.0f, min_x - me.x), a1 = fmaxf(0.0f, me.x - max_x);
        const float a2 = fmaxf(0.0f, min_y - me.y), a3 = fmaxf(0.0f, me.y - max_y);
        return ((a0 * a0 + a1 * a1) + a2 * a2) + a3 * a3;
    };
    int sp = 0, next = 0;
    st[sp++] = (uint16_t)(kd_tree_size(P.A, t) << 8);
    while (sp > 0) {
        const int e = st[--sp];
        const int b = e & 0xff, en = e >> 8;
        int nl = 0;
        if (en - b > kKdLeaf)
            for (int r = 0; r < n_cur; ++r) {
                const uint32_t m = cur[r].meta;
                if ((m & 0xffffu) == (uint32_t)e && ((m >> 24) & 1u) == 0u) nl = (m >> 16) & 0xff;
            }
        if (nl > 0) {
            const float dl = box_dist(b, b + nl), dr = box_dist(b + nl, en);
            const uint16_t lo = (uint16_t)(b | ((b + nl) << 8)), hi = (uint16_t)((b + nl) | (en << 8));
            st[sp++] = dl < dr ? hi : lo;  // the farther child waits
            st[sp++] = dl < dr ? lo : hi;
        } else {
            for (int p = b; p < en; ++p) vis[row[p]] = (uint8_t)next++;
        }
    }
}

// Second sweep of the 20-candidate pair phase (orca_phases, two_sweeps): lane = (agent, slot), 6 agents per pass of a wave
// (lanes 60..63 idle): the half-plane of every kept neighbour, the number kept.  DETECT: a simulator has an exact distance tie
// that RVO2's visiting order decides if candidates of one agent shared a slot in the first sweep (strict ranks).
template <bool DETECT>
__device__ __forceinline__ void pair_sweep2(const Params& P, const Smem& s) {
    const int* kept = reinterpret_cast<const int*>(s.proj);
    const int tid = opaque_tid();
    const int wl = tid & (kWave - 1), g = wl / 10, slot = wl - g * 10;
    const int waves = (P.threads + kWave - 1) / kWave;
    const KdSmem k = DETECT ? kd_view(P, s) : KdSmem{};
    for (int q0 = (tid / kWave) * 6; q0 < P.nA; q0 += 6 * waves) {
        const int q = q0 + g;
        const bool lane_ok = wl < 60 && q < P.nA;
        const int e = lane_ok ? kept[q * 10 + slot] : -1;
        const bool valid = e >= 0;
        const unsigned long long vm = __ballot(valid);
        const int qq = lane_ok ? q : 0;
        const int ol = valid ? (e & 0xff) : qq;  // unused slots: a finite dummy (the agent against itself), not stored
        const bool robot_sim = (e >> 8) & 1;
        const float4 me = s.kin[qq];
        const float4 ot = s.kin[ol];
        const float rq_r = s.rview[qq], ro_r = s.rview[ol], rq_h = s.hview[qq], ro_h = s.hview[ol];
        const float rsum = (valid && robot_sim) ? rq_r + ro_r : rq_h + ro_h;
        if (lane_ok && slot == 0) s.count[q] = __popcll((vm >> (10 * g)) & 0x3ffull);
        if (valid)
            s.lines[q * kLineStride + slot] = make_half_plane(P.orca, me.x, me.y, me.z, me.w, ot.x, ot.y, ot.z, ot.w, rsum);
        if (DETECT) {  // more candidates claimed a slot than slots are filled: some share one, i.e. are equally far
            if (lane_ok && slot == 0 && reinterpret_cast<const int*>(k.dnext)[q] != (int)__popcll((vm >> (10 * g)) & 0x3ffull)) k.tie[q] = 1;
        }
    }
}

// Exact distance ties (rare: never in random scenes): work out RVO2's visiting order for the simulators that have one and rank
// their candidates again, ties by visiting order.  (Tried as a real call, __attribute__((noinline)): the mere presence of a call
// cost the 20-human rollout 4-5 % on every step — 62.3 vs 59.4 M env-steps/s instrumented — so it is inlined, and the few
// registers the rare path spills go to scratch.)  All threads of the workgroup (barriers inside).
template <bool COMPACT = false>
__device__ __forceinline__ void kd_resolve_ties(const Params& P, const Smem& s, const Lane& L, int kd_gen, int two_sweeps) {
    const KdSmem k = kd_scratch_view<COMPACT>(P, s);
    const float range_sq = P.orca.neighbor_dist * P.orca.neighbor_dist;
    if (L.lane < P.nA && k.tie[L.lane] != 0) kd_visit_order<COMPACT>(P, s, L, kd_gen);
    block_sync(P);
    int* kept = reinterpret_cast<int*>(s.proj);
    for (int p = L.lane; p < P.pairs; p += P.threads) {
        const int info = pair_info<COMPACT>(P, s, p);
        const int q = info & 0xff, c = (info >> 16) & 0xff, ol = (info >> 8) & 0xff;
        if (k.tie[q] == 0) continue;
        const int qb = q / P.A * P.A;
        const uint8_t* vq = k.visit + (size_t)q * k.row;
        const float mine = s.d2[p];
        const float* row = s.d2 + (p - c);
        const int my_visit = vq[ol - qb];
        int rank = 0;
        for (int kk = 0; kk < P.NC; ++kk) {
            const float v = row[kk];
            const int visit = vq[((pair_info<COMPACT>(P, s, p - c + kk) >> 8) & 0xff) - qb];
            rank += (v < range_sq ? 1 : 0) & ((v < mine ? 1 : 0) | ((v == mine ? 1 : 0) & (visit < my_visit ? 1 : 0)));
        }
        if (mine < range_sq && rank < P.orca.max_neighbors) {
            const bool robot_sim = (info >> 25) & 1;
            if (two_sweeps) {
                kept[q * 10 + rank] = ol | ((robot_sim ? 1 : 0) << 8);
            } else {
                const float4 me = s.kin[q];
                const float4 ot = s.kin[ol];
                const float rsum = robot_sim ? s.rview[q] + s.rview[ol] : s.hview[q] + s.hview[ol];
                s.lines[q * kLineStride + rank] = make_half_plane(P.orca, me.x, me.y, me.z, me.w, ot.x, ot.y, ot.z, ot.w, rsum);
            }
        }
    }
    block_sync(P);
    if (two_sweeps) {
        pair_sweep2<false>(P, s);
        block_sync(P);
    }
}

// ---------------------------------------------------------------------------------------------- ORCA phases
// Stage + pair phases + per-agent solve.  Called by all 64 lanes (contains barriers); on return every valid
// agent lane with solve == true holds its new velocity (ORCA.predict, orca.py:82-132).
// CN_PHASE_TIMING (compile time, profiling builds only): per-phase shader-clock accumulation inside the fused rollout
// (scripts/phase_probe.py).  Off in the product build: PhaseClock is empty and CN_TICK expands to nothing.
#ifdef CN_PHASE_TIMING
static __device__ unsigned long long cn_phase_cycles[16];  // (static: step_kernels.h is part of both translation units)
struct PhaseClock {
    unsigned long long last, acc[10];
};
#define CN_TICK(clk, k)                                                \
    do {                                                               \
        if (clk) {                                                     \
            const unsigned long long now_ = __builtin_readcyclecounter(); \
            (clk)->acc[k] += now_ - (clk)->last;                       \
            (clk)->last = now_;                                        \
        }                                                              \
    } while (0)
#else
struct PhaseClock {};
#define CN_TICK(clk, k) \
    do {                \
    } while (0)
#endif

// KD (compile time): the instantiation carries the kd-tree bookkeeping of simulators with more than 10 agents (kd_order.h);
// crowds of at most 9 humans run the one without it.
template <int MAXL, bool KD, bool COMPACT = false>
__device__ __forceinline__ void orca_phases(const Params& P, const Smem& s, const Lane& L, const AgentRegs& r,
                                            float robot_max_speed, bool solve, float& out_vx, float& out_vy,
                                            PhaseClock* clk = nullptr) {
    (void)clk;
    // preferred velocity: towards the goal, unit length once farther than 1 m (orca.py:113-115)
    // (COMPACT: the per-episode constants of the agent — goal, preferred speed, radius — are read from LDS where needed)
    const float max_speed = (L.a == 0) ? robot_max_speed : (float)(COMPACT ? (L.lane < P.nA ? s.vpref[L.lane] : 0.0) : r.vpref);
    auto preferred = [&](float& pref_x, float& pref_y) {
        const double2 goal = COMPACT ? s.goal2[L.lane] : make_double2(r.gx, r.gy);
        const double gdx = goal.x - r.px, gdy = goal.y - r.py;
        const double speed = norm2(gdx, gdy);
        pref_x = (float)(speed > 1.0 ? gdx / speed : gdx);
        pref_y = (float)(speed > 1.0 ? gdy / speed : gdy);
    };
    if (MAXL == 10 && P.NC == 20 && P.orca.max_neighbors == 10) {  // two-sweep pair phase: no neighbour in any slot yet
        int4* kept4 = reinterpret_cast<int4*>(s.proj);
        for (int i = L.lane; i * 4 < P.nA * 10; i += P.threads) kept4[i] = make_int4(-1, -1, -1, -1);
    }
    if (L.lane < P.nA) {
        s.kin[L.lane] = make_float4((float)r.px, (float)r.py, (float)r.vx, (float)r.vy);
        s.posd[L.lane] = make_double2(r.px, r.py);
        if (!COMPACT) {  // (COMPACT: written when the agent is loaded, stage_constants)
            s.rad[L.lane] = r.rad;
            s.hview[L.lane] = (float)(r.rad + 0.01 + step_param<COMPACT>(P, s, kParHSafety));
        }
        float pref_x, pref_y;
        preferred(pref_x, pref_y);
        s.sol[L.lane] = make_float4(pref_x, pref_y, max_speed, solve ? 1.0f : 0.0f);
        if (KD) {
            const KdSmem k = kd_view(P, s);
            k.tie[L.lane] = 0;
            reinterpret_cast<int*>(k.dnext)[L.lane] = 0;  // slots below 10 claimed by this agent's candidates (two-sweep pair phase)
        }
    }
    block_sync(P);
    // simulators of more than 10 agents: this step's kd-tree(s) and every simulator's permutation (kd_order.h); the visiting
    // order itself is only worked out further down if a simulator turns out to have an exact distance tie
    int kd_gen = 0;  // the generation of node lists that holds this step's tree
    if (KD) {
        const int g_last = *kd_view(P, s).gen;
        CN_TICK(clk, 0);
        if (P.E == 1 && P.threads == kWave) {
            kd_gen = kd_build_trees_wave(P, s, L, g_last);
            block_sync(P);  // (one wave: orders the builder's LDS records before the simulator lanes read them)
        } else {
            kd_gen = g_last ^ 1;
            kd_build_trees(P, s, L, kd_gen);
        }
        CN_TICK(clk, 4);  // (probe builds: the tree build is booked under "robot action publish")
        kd_update_orders(P, s, L, kd_gen);
        if (L.lane == 0) *kd_view(P, s).gen = kd_gen;
    }
    CN_TICK(clk, 0);

    // pairs-1: squared distances, self.pos - other.pos (Appendix A.2)
    for (int p = L.lane; p < P.pairs; p += P.threads) {
        const int info = pair_info<COMPACT>(P, s, p);
        const float4 me = s.kin[info & 0xff];
        const float4 ot = s.kin[(info >> 8) & 0xff];
        const float dx = me.x - ot.x, dy = me.y - ot.y;
        s.d2[p] = ((info >> 24) & 1) ? dx * dx + dy * dy : std::numeric_limits<float>::infinity();
    }
    block_sync(P);
    CN_TICK(clk, 1);

    // pairs-2: neighbour slot = stable rank by (distSq, visit order) among the in-range candidates, which
    // is what RVO2's sorted insertion with strict '<' produces; slots >= maxNeighbors fall off the list.
    const float range_sq = P.orca.neighbor_dist * P.orca.neighbor_dist;
    // BASELINE configs[3]'s crowd (20 candidates per agent, 10 kept): the candidate row comes in as five 16-byte reads
    // issued together (the scalar loop waited for LDS once per two candidates: ten round trips per pass of 64 pairs), and
    // the half-plane is computed in a second sweep over the (agent, kept slot) lanes only — 210 of the 420 ordered pairs,
    // four passes of a wave instead of seven.  Same comparisons and the same half-plane arithmetic: bit-identical.
    const bool two_sweeps = MAXL == 10 && P.NC == 20 && P.orca.max_neighbors == 10;
    if (two_sweeps) {
        // kept [nA][10]: candidate lane | robot's-sim bit << 8 of the pair that ranks there, -1 = no such neighbour (cleared in
        // the stage phase; proj is free until the solve).  A pair inside the range ranks by (v < mine) | (v == mine & k < c)
        // alone: every candidate that precedes it is inside the range too — 4 vector instructions per candidate instead of 7;
        // the ranks of an agent's in-range candidates are a permutation of 0 .. within - 1, so the neighbours kept are
        // exactly the filled slots and their number falls out of the second sweep's ballot.
        // With the kd bookkeeping (always, at 20 candidates) the rank counts STRICTLY nearer candidates only — 2 vector
        // instructions per candidate instead of 4-5: candidates at exactly the same distance then share a rank and a slot, which
        // the second sweep notices (more candidates claimed a slot below 10 than slots are filled) and kd_resolve_ties ranks
        // that agent's candidates again, ties by RVO2's visiting order.
        int* kept = reinterpret_cast<int*>(s.proj);
        int* const claimed = KD ? reinterpret_cast<int*>(kd_view(P, s).dnext) : nullptr;
        for (int p = L.lane; p < P.pairs; p += P.threads) {
            const int info = pair_info<COMPACT>(P, s, p);
            const int c = (info >> 16) & 0xff;
            const float mine = s.d2[p];
            const float4* row4 = reinterpret_cast<const float4*>(s.d2 + (p - c));
            float v[20];
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const float4 t = row4[j];
                v[4 * j] = t.x, v[4 * j + 1] = t.y, v[4 * j + 2] = t.z, v[4 * j + 3] = t.w;
            }
            int rank = 0;
            if (KD) {
                // squared distances are +0 .. +inf: their bit patterns order like the floats, and bit 31 of the 32-bit difference
                // says v < mine.  v_sub_u32 + v_alignbit_b32 (shift the sign into a 20-bit word) per candidate and one popcount,
                // instead of v_cmp -> vcc -> v_cndmask / v_addc with a two-state hazard s_nop between each pair (gfx950): 41
                // issue slots per pass instead of 60, nothing through VCC.
                const uint32_t mb = __float_as_uint(mine);
                uint32_t nearer = 0u;
#pragma unroll
                for (int k = 0; k < 20; ++k) nearer = __builtin_amdgcn_alignbit(nearer, __float_as_uint(v[k]) - mb, 31);
                rank = __popc(nearer);
            } else {
#pragma unroll
                for (int k = 0; k < 20; ++k) rank += ((v[k] < mine) | ((v[k] == mine) & (k < c))) ? 1 : 0;
            }
            if (mine < range_sq && rank < 10) {
                kept[(info & 0xff) * 10 + rank] = ((info >> 8) & 0xff) | (((info >> 25) & 1) << 8);
                if (KD) atomicAdd(claimed + (info & 0xff), 1);  // (LDS, no return value)
            }
        }
        block_sync(P);
        if (KD)
            pair_sweep2<true>(P, s);
        else
            pair_sweep2<false>(P, s);
    } else
    for (int p = L.lane; p < P.pairs; p += P.threads) {
        const int info = pair_info<COMPACT>(P, s, p);
        const int q = info & 0xff, c = (info >> 16) & 0xff;
        const float mine = s.d2[p];
        const float* row = s.d2 + (p - c);
        int rank = 0, within = 0, same = 0;
        for (int k = 0; k < P.NC; ++k) {  // (bitwise, not short-circuit: no branch per candidate)
            const float v = row[k];
            const int in = v < range_sq ? 1 : 0;
            within += in;
            same += v == mine ? 1 : 0;  // itself included
            rank += in & ((v < mine ? 1 : 0) | ((v == mine ? 1 : 0) & (k < c ? 1 : 0)));
        }
        if (KD && mine < range_sq && same > 1) kd_view(P, s).tie[q] = 1;  // an exact tie: RVO2's kd-tree visiting order decides
        if (c == 0) s.count[q] = within < P.orca.max_neighbors ? within : P.orca.max_neighbors;
        if (mine < range_sq && rank < P.orca.max_neighbors) {
            const int ol = (info >> 8) & 0xff;
            const float4 me = s.kin[q];
            const float4 ot = s.kin[ol];
            const bool robot_sim = (info >> 25) & 1;
            const float rsum = robot_sim ? s.rview[q] + s.rview[ol] : s.hview[q] + s.hview[ol];
            s.lines[q * kLineStride + rank] =
                make_half_plane(P.orca, me.x, me.y, me.z, me.w, ot.x, ot.y, ot.z, ot.w, rsum);
        }
    }
    block_sync(P);
    if (KD) {
        const KdSmem k = kd_view(P, s);
        const bool my_tie = L.lane < P.nA && k.tie[L.lane] != 0;
        if (block_sync_or(P, my_tie ? 1 : 0)) kd_resolve_ties<COMPACT>(P, s, L, kd_gen, two_sweeps ? 1 : 0);
    }
    CN_TICK(clk, 2);

    out_vx = 0.0f, out_vy = 0.0f;
    float rx = 0.0f, ry = 0.0f;
    int n = 0, fail = 0;
    if (MAXL == 5) {
        // candidates: lane = (agent, half-plane)
        for (int p = L.lane; p < P.nA * MAXL; p += P.threads) {
            const int q = p / MAXL, k = p - q * MAXL;
            const float4 so = s.sol[q];
            const float4* lq = s.lines + q * kLineStride;
            s.cand2[q * kLineStride + k] = lp_line_candidate<MAXL - 1>(lq[k], lq, k, so.z, so.x, so.y, false);
            if (k == 0) {
                float sx, sy;
                lp_start_point(so.z, so.x, so.y, sx, sy);
                s.res[q] = make_float4(sx, sy, 0.0f, 0.0f);
            }
        }
        block_sync(P);
        if (solve) {
            n = s.count[L.lane];
            const float4 start = s.res[L.lane];
            rx = start.x, ry = start.y;
            fail = lp_planar_scan<MAXL>(s.lines + L.lane * kLineStride, s.cand2 + L.lane * kLineStride, n, rx, ry);
        }
    } else {
        lp_planar_tri(s.lines, s.count, s.sol, s.res, P.nA, P.threads);  // every thread: lane = (agent, third of its half-planes)
        block_sync(P);
        if (solve) {
            n = s.count[L.lane];
            const float4 got = s.res[L.lane];
            rx = got.x, ry = got.y;
            fail = __float_as_int(got.z);
        }
    }
    CN_TICK(clk, 3);
    const bool need = solve && fail < n;
#ifdef CN_PHASE_TIMING
    if (clk) clk->acc[9] += __popcll(__ballot(need));  // agents in the fallback
#endif
    if (block_sync_or(P, need ? 1 : 0)) {  // some agent of this workgroup was infeasible: compact them
        if (L.lane < kWave) {  // (agent lanes live in wave 0)
            const unsigned long long nm = __ballot(need);
            if (need) {
                if (MAXL == 10) s.res[L.lane] = make_float4(rx, ry, __int_as_float(fail), 0.0f);
                s.todo[__popcll(nm & ((1ull << L.lane) - 1ull))] = L.lane;
            }
            if (L.lane == 0) s.todo[P.nA] = __popcll(nm);
        }
        block_sync(P);
        if (MAXL == 5) {
            constexpr int kPairs = 10;  // MAXL (MAXL - 1) / 2 projections and as many candidates per infeasible agent
            const int items = s.todo[P.nA] * kPairs;
            for (int p = L.lane; p < items; p += P.threads) {  // projections: lane = (agent, i, j)
                const int t = p / kPairs, m = p - t * kPairs;
                const int a = s.todo[t];
                const int i = lp3_program_of(m), j = m - i * (i - 1) / 2;
                const float4* la = s.lines + a * kLineStride;
                s.proj[a * kLineStride + m] = lp3_project(la[i], la[j]);
            }
            block_sync(P);
            for (int p = L.lane; p < items; p += P.threads) {  // their candidates: lane = (agent, i, k)
                const int t = p / kPairs, m = p - t * kPairs;
                const int a = s.todo[t];
                const int i = lp3_program_of(m), base = i * (i - 1) / 2;
                const float4 li = s.lines[a * kLineStride + i];
                const float4* pa = s.proj + a * kLineStride + base;
                s.cand3[a * kLineStride + m] = lp_line_candidate<3>(pa[m - base], pa, m - base, s.sol[a].z, -li.w, li.z, true);
            }
            block_sync(P);
            if (need)
                lp3_scan(s.lines + L.lane * kLineStride, s.proj + L.lane * kLineStride, s.cand3 + L.lane * kLineStride, n, fail,
                         max_speed, rx, ry);
        } else {
            // (the lazy fallback's candidate rows — 7 agents x 9 float4 per wave — live in d2, free after the pair phases: enough
            // for one wave of a crowd of 12+ agents; otherwise the shuffle-round form)
            if (P.threads == kWave && P.pairs * 4 >= kLazyCandFloat4 * 16)
                lp_relaxed_lazy<10, COMPACT>(s.lines, s.proj, reinterpret_cast<float4*>(s.d2), s.count, s.sol, s.res, s.todo,
                                             s.todo[P.nA], P.threads);
            else
                lp_relaxed_coop<10>(s.lines, s.count, s.sol, s.res, s.todo, s.todo[P.nA]);
            block_sync(P);
            if (need) {
                const float4 got = s.res[L.lane];
                rx = got.x, ry = got.y;
            }
        }
    }
    out_vx = rx, out_vy = ry;
    CN_TICK(clk, 8);
}

struct StepResult {  // meaningful on the robot lane
    double reward, dmin, ax, ay;
    uint8_t done, info;
};

// One transition for the lane's agent (crowd_sim.py:317-420).  `r` is updated in place when update != 0.
// new_vx/new_vy: the velocity this lane's agent chose (float32 for ORCA agents, the action for the robot).
// Must be called by all 64 lanes (contains workgroup barriers).
__device__ __forceinline__ double python_fmod(double x, double y) {  // python's float %: result takes the sign of y
    double m = fmod(x, y);
    if (m != 0.0 && ((m < 0.0) != (y < 0.0))) m += y;
    return m;
}

// UNI (compile time): the robot may be a unicycle.  The holonomic instantiation carries none of that code — it sits
// inside the fused rollout loop, where 20 extra VGPRs and a few dead branches cost 7 % (712 -> 665 M env-steps/s).
template <int MAXL, bool UNI, bool KD, bool COMPACT = false>
__device__ __forceinline__ void step_core(const Params& P, const Smem& s, const Lane& L, AgentRegs& r,
                                          double& gtime, float robot_max_speed, const double* ext_action,
                                          int update, StepResult& res, double& new_vx, double& new_vy,
                                          double* theta_io = nullptr, PhaseClock* clk = nullptr, const float* known_vel = nullptr,
                                          const double* robot_action_regs = nullptr /* the robot lane's action, in registers */,
                                          const float* lane_vel_regs = nullptr /* this lane's entry of known_vel, requested earlier */) {
    (void)clk;
    float ovx = 0.0f, ovy = 0.0f;
    if (known_vel != nullptr) {
        // the humans' ORCA velocities of THIS state are known already ([B][A][2], orca_kernel's output: sarl_decide_step_kernel
        // computed them behind the previous transition for the decision's lookahead — the same function of the same state);
        // what the phases below read from LDS besides them: the agents' float64 positions and radii
        if (L.lane < P.nA) {
            s.posd[L.lane] = make_double2(r.px, r.py);
            if (!COMPACT) s.rad[L.lane] = r.rad;
        }
        if (L.valid && L.a > 0) {
            ovx = lane_vel_regs ? lane_vel_regs[0] : known_vel[2 * L.gi];
            ovy = lane_vel_regs ? lane_vel_regs[1] : known_vel[2 * L.gi + 1];
        }
    } else {
        orca_phases<MAXL, KD, COMPACT>(P, s, L, r, robot_max_speed, L.valid && (L.a > 0 || P.robot_orca), ovx, ovy, clk);
    }
    new_vx = ovx;
    new_vy = ovy;
    // unicycle robot (ActionRot v, r): the collision test uses v (cos, sin)(r + theta) (crowd_sim.py:339-341), the
    // move is compute_position's (agent.py:115-118)
    const bool unicycle = UNI && P.robot_unicycle && !P.robot_orca && L.valid && L.a == 0;
    double rot_v = 0.0, rot_r = 0.0, theta0 = 0.0;
    if (L.valid && L.a == 0) {
        if (!P.robot_orca) {
            new_vx = robot_action_regs ? robot_action_regs[0] : ext_action[2 * (size_t)L.env];
            new_vy = robot_action_regs ? robot_action_regs[1] : ext_action[2 * (size_t)L.env + 1];
        }
        if (unicycle) {
            rot_v = new_vx, rot_r = new_vy, theta0 = *theta_io;
            new_vx = rot_v * cos(rot_r + theta0);
            new_vy = rot_v * sin(rot_r + theta0);
        }
        s.act[L.lane] = make_double2(new_vx, new_vy);
    }
    block_sync(P);
    CN_TICK(clk, 4);

    // One float64 distance per agent lane, computed branch-free so that humans and the robot share the
    // instruction stream:
    //   human: swept robot-human distance over the step, human's CURRENT velocity vs the robot's NEW action
    //          = point_to_segment_dist(px, py, ex, ey, 0, 0) - r_h - r_r      (crowd_sim.py:331-351)
    //   robot: distance of its end position to its goal                       (crowd_sim.py:364-366)
    double goal_dist = 0.0;
    if (L.valid) {
        const bool human = L.a > 0;
        const double2 rp = s.posd[L.ebase];
        const double2 act = s.act[L.ebase];
        const double x1 = r.px - rp.x, y1 = r.py - rp.y;
        const double wx = r.vx - act.x, wy = r.vy - act.y;
        const double c_dt = step_param<COMPACT>(P, s, kParDt);
        const double x2 = x1 + wx * c_dt, y2 = y1 + wy * c_dt;
        const double sx = x2 - x1, sy = y2 - y1;
        double u = ((0.0 - x1) * sx + (0.0 - y1) * sy) / (sx * sx + sy * sy);
        u = (u > 1.0) ? 1.0 : ((u < 0.0) ? 0.0 : u);
        const bool degenerate = (sx == 0.0 && sy == 0.0);  // utils.py:11-13
        const double cx = degenerate ? 0.0 - x1 : (x1 + u * sx) - 0.0;
        const double cy = degenerate ? 0.0 - y1 : (y1 + u * sy) - 0.0;
        double endx = r.px + new_vx * c_dt, endy = r.py + new_vy * c_dt;
        if (unicycle) {
            const double th = theta0 + rot_r;
            endx = r.px + cos(th) * rot_v * c_dt;
            endy = r.py + sin(th) * rot_v * c_dt;
        }
        const double2 goal = COMPACT ? s.goal2[L.lane] : make_double2(r.gx, r.gy);
        const double d = norm2(human ? cx : endx - goal.x, human ? cy : endy - goal.y);
        if (human) {
            s.closest[L.lane] = d - (COMPACT ? s.rad[L.lane] : r.rad) - s.rad[L.ebase];
        } else {
            goal_dist = d;
        }
    }
    block_sync(P);
    CN_TICK(clk, 5);

    res.done = 0;
    if (L.valid && L.a == 0) {
        // the reference stops scanning at the first colliding human (dmin keeps the minimum seen before it)
        double dmin = std::numeric_limits<double>::infinity();
        bool collision = false;
        for (int i = 1; i < P.A; ++i) {
            const double c = s.closest[L.lane + i];
            const bool hit = c < 0.0;
            dmin = (!collision && !hit && c < dmin) ? c : dmin;
            collision = collision || hit;
        }
        const bool reaching = goal_dist < (COMPACT ? s.rad[L.lane] : r.rad);
        const double c_ddist = step_param<COMPACT>(P, s, kParDDist);
        if (gtime >= step_param<COMPACT>(P, s, kParLimit) - 1.0) {
            res.reward = 0.0, res.done = 1, res.info = CN_TIMEOUT;
        } else if (collision) {
            res.reward = step_param<COMPACT>(P, s, kParCollision), res.done = 1, res.info = CN_COLLISION;
        } else if (reaching) {
            res.reward = step_param<COMPACT>(P, s, kParSuccess), res.done = 1, res.info = CN_REACH_GOAL;
        } else if (dmin < c_ddist) {
            res.reward = (dmin - c_ddist) * step_param<COMPACT>(P, s, kParDFactor) * step_param<COMPACT>(P, s, kParDt);
            res.done = 0, res.info = CN_DANGER;
        } else {
            res.reward = 0.0, res.done = 0, res.info = CN_NOTHING;
        }
        res.dmin = dmin;
        res.ax = unicycle ? rot_v : new_vx, res.ay = unicycle ? rot_r : new_vy;
        if (update) gtime += step_param<COMPACT>(P, s, kParDt);
    }
    if (update && L.valid) {  // Agent.step (agent.py:127-135)
        const double c_dt = step_param<COMPACT>(P, s, kParDt);
        if (unicycle) {
            const double th = theta0 + rot_r;
            r.px = r.px + cos(th) * rot_v * c_dt;
            r.py = r.py + sin(th) * rot_v * c_dt;
            const double theta1 = python_fmod(theta0 + rot_r, 2 * 3.141592653589793);
            r.vx = rot_v * cos(theta1);
            r.vy = rot_v * sin(theta1);
            *theta_io = theta1;
        } else {
            r.px = r.px + new_vx * c_dt;
            r.py = r.py + new_vy * c_dt;
            r.vx = new_vx;
            r.vy = new_vy;
        }
    }
    CN_TICK(clk, 6);
}

// ---------------------------------------------------------------------------------------------- kernels

template <int MAXL, bool KD>
__global__ __launch_bounds__(kMaxBlock) void orca_kernel(Params P, StateView S, float* out_vel) {
    const Smem s = carve<MAXL>(P);
    const Lane L = lane_of(P);
    AgentRegs r = {};
    if (L.valid) load_agent(S, L.gi, r);
    float robot_max_speed;
    load_robot_view(P, S, s, L, r, robot_max_speed);
    build_pairs(P, s);
    if (KD) kd_load(P, S, s, L);
    float vx, vy;
    orca_phases<MAXL, KD>(P, s, L, r, robot_max_speed, L.valid, vx, vy);
    if (KD) kd_store(P, S, s, L);
    if (L.valid) {
        out_vel[2 * L.gi] = vx;
        out_vel[2 * L.gi + 1] = vy;
        if (L.a == 0) S.rsim_valid[L.env] = 1;
    }
}

template <int MAXL, bool UNI, bool KD>
__global__ __launch_bounds__(kMaxBlock) void step_kernel(Params P, StateView S, StepIo io) {
    const Smem s = carve<MAXL>(P);
    const Lane L = lane_of(P);
    AgentRegs r = {};
    if (L.valid) load_agent(S, L.gi, r);
    float robot_max_speed = 0.0f;
    if (P.robot_orca) load_robot_view(P, S, s, L, r, robot_max_speed);
    build_pairs(P, s);
    if (KD) kd_load(P, S, s, L);
    double gtime = (L.valid && L.a == 0) ? S.gtime[L.env] : 0.0;
    const AgentRegs before = r;

    double theta = (L.valid && L.a == 0) ? S.theta[L.env] : 0.0;
    StepResult res;
    double nvx, nvy;
    step_core<MAXL, UNI, KD>(P, s, L, r, gtime, robot_max_speed, io.action, io.update, res, nvx, nvy, &theta);
    if (KD) kd_store(P, S, s, L);  // (a lookahead rebuilds the humans' trees too: the reference's onestep_lookahead runs doStep)
    if (!L.valid) return;

    if (L.a == 0) {
        io.reward[L.env] = res.reward;
        io.done[L.env] = res.done;
        io.info[L.env] = res.info;
        if (io.dmin) io.dmin[L.env] = res.dmin;
        if (io.action_out) {
            io.action_out[2 * (size_t)L.env] = res.ax;
            io.action_out[2 * (size_t)L.env + 1] = res.ay;
        }
        if (io.update) {
            S.gtime[L.env] = gtime;
            S.theta[L.env] = theta;
        }
        if (P.robot_orca) S.rsim_valid[L.env] = 1;
    }
    if (io.orca_vel) {
        io.orca_vel[2 * L.gi] = (float)nvx;
        io.orca_vel[2 * L.gi + 1] = (float)nvy;
    }
    if (io.update) {
        S.pos[L.gi] = make_double2(r.px, r.py);
        S.vel[L.gi] = make_double2(r.vx, r.vy);
    }
    if (io.obs && L.a > 0) {
        // update: get_observable_state after the move; else get_next_observable_state (agent.py:63-74)
        double* o = io.obs + ((size_t)L.env * (P.A - 1) + (L.a - 1)) * 5;
        if (io.update) {
            o[0] = r.px, o[1] = r.py, o[2] = r.vx, o[3] = r.vy;
        } else {
            o[0] = before.px + nvx * P.dt, o[1] = before.py + nvy * P.dt, o[2] = nvx, o[3] = nvy;
        }
        o[4] = r.rad;
    }
}

// Everything below — scenario generation, rollout bookkeeping, the rollout kernels — belongs to the env translation unit
// (crowdnav_amd.hip).  sarl_abi.hip includes this header for orca_kernel / step_kernel's types only and defines CN_SARL_TU.
#ifndef CN_SARL_TU
// Lane-per-scenario generation: try the register-only head generator first (no memory traffic), redo the scenario
// with the memory-backed one (word-major HBM scratch: any number of waves per CU) in the rare case its 227 words do not
// suffice.  Returns random() calls consumed.
__device__ __forceinline__ uint64_t generate_scenario_lane(const ScenarioCfg& C, uint32_t seed, size_t base,
                                                           double2* pos, double2* vel, double2* goal, double2* rv,
                                                           uint32_t* hbm_column, int hbm_stride, bool keep_state,
                                                           int* pos_out) {
    if (!keep_state) {
        Mt19937Head head;
        const uint64_t n = generate_scenario(C, head, seed, base, pos, vel, goal, rv);
        if (!head.dead()) return n;
    }
    Mt19937 rng{hbm_column, hbm_stride, 0};
    const uint64_t n = generate_scenario(C, rng, seed, base, pos, vel, goal, rv);
    if (keep_state && pos_out) *pos_out = rng.pos;
    return n;
}

// np.random.seed(seed) + scenario of one env per lane (lane = env)
__global__ __launch_bounds__(kWave) void reset_kernel(Params P, ScenarioCfg C, StateView S, const uint32_t* seeds,
                                                     const uint8_t* mask, uint64_t* draws) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= P.B) return;
    if (mask && !mask[b]) return;
    const uint64_t n = generate_scenario_lane(C, seeds[b], (size_t)b * P.A, S.pos, S.vel, S.goal, S.rv,
                                                      S.mt_key + b, P.B, true, &S.mt_pos[b]);
    S.gtime[b] = 0.0;
    S.theta[b] = 1.5707963267948966;  // robot.set(..., np.pi / 2)
    if (draws) draws[b] = n;
    if (P.kd)  // new Human objects, new ORCA policies, new rvo2 simulators (the robot's persists)
        for (int a = 1; a < P.A; ++a) S.kd_valid[(size_t)b * P.A + a] = 0;
}

__global__ void mt_probe_kernel(uint32_t* key, uint32_t seed, int n, double* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Mt19937 rng{key, 1, 0};
    rng.seed(seed);
    for (int i = 0; i < n; ++i) out[i] = rng.random();
}

// Rollout bookkeeping lives behind ONE pointer to a device copy of the caller's cn_rollout_io: the ~20
// pointers in it are needed only when an episode ends, and keeping them out of the kernel arguments keeps
// them out of the SGPR file of the step loop.
struct RolloutView {
    const cn_rollout_io* io;
    const double* discount;  // [discount_len]
    int discount_len;
};

// io.active[] states
enum { kRetired = 0, kRunning = 1, kWaitingScenario = 2 };

__device__ __forceinline__ int64_t episode_id(const cn_rollout_io& io, int b, int ordinal) {
    return io.env_offset + b + (int64_t)ordinal * io.env_stride;
}
__device__ __forceinline__ uint32_t episode_seed(const cn_rollout_io& io, int64_t c) {
    return io.seed_base + (uint32_t)((uint64_t)c % io.seed_mod);
}

// (re)start bookkeeping: env b begins its episode ordinal 0
__global__ __launch_bounds__(kWave) void rollout_begin_kernel(Params P, ScenarioCfg C, StateView S, RolloutView R) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= P.B) return;
    const cn_rollout_io io = *R.io;
    const int64_t c0 = episode_id(io, b, 0);
    const bool on = io.episode_limit < 0 || c0 < io.episode_limit;
    io.active[b] = on ? kRunning : kRetired;
    S.ep_word[b] = on ? kRunning : kRetired;
    io.ep_count[b] = 0;
    io.cur_steps[b] = 0;
    io.cur_return[b] = 0.0;
    if (io.cur_danger) io.cur_danger[b] = 0;
    if (io.cur_danger_dmin_sum) io.cur_danger_dmin_sum[b] = 0.0;
    S.ring_filled_in[b] = 0;
    S.ring_filled_out[b] = 0;
    if (P.kd)
        for (int a = 1; a < P.A; ++a) S.kd_valid[(size_t)b * P.A + a] = 0;
    if (!on) return;
    generate_scenario_lane(C, episode_seed(io, c0), (size_t)b * P.A, S.pos, S.vel, S.goal, S.rv, S.mt_key + b,
                                   P.B, false, nullptr);
    S.mt_pos[b] = -1;
    S.gtime[b] = 0.0;
}

// Scenario ring fill: one lane per (env, ring slot) generates the episode whose ordinal maps to that slot
// if it has not been generated yet, so that ordinals [next, next + D) are resident when the rollout
// launch that follows needs them.  The lane runs the register-only head generator; the
// rare scenario whose rejection chain outruns its 227 words is queued in redo_list and regenerated by ring_redo_kernel
// with a memory-backed generator from a small fixed pool (624 x kRedoLanes words, not 624 x B x D).
constexpr int kRedoLanes = 4096;
__global__ __launch_bounds__(kWave) void ring_fill_kernel(Params P, ScenarioCfg C, StateView S, RolloutView R) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int D = P.ring_depth;
    if (idx >= P.B * D) return;
    const int b = idx / D, slot = idx - b * D;
    const cn_rollout_io* io = R.io;
    const int state = io->active[b];
    // first ordinal the rollout may still ask for (an env waiting for a scenario has not consumed ep_count yet)
    const int next = io->ep_count[b] + (state == kWaitingScenario ? 0 : 1);
    if (slot == 0) S.ring_filled_out[b] = next + D;
    if (state == kRetired) return;
    const int ordinal = next + ((slot - next % D) + D) % D;
    if (ordinal < S.ring_filled_in[b]) return;  // still resident from an earlier fill
    const int64_t c = episode_id(*io, b, ordinal);
    if (io->episode_limit >= 0 && c >= io->episode_limit) return;
    const uint32_t seed = episode_seed(*io, c);
    const size_t base = ((size_t)b * D + slot) * P.A;
    Mt19937Head head;
    generate_scenario(C, head, seed, base, S.ring_pos, nullptr, S.ring_goal, S.ring_rv);
    if (head.dead()) S.redo_list[atomicAdd(S.redo_count, 1)] = make_int2(idx, (int)seed);
}

// The scenarios ring_fill_kernel<false> queued: a fixed grid of kRedoLanes lanes strides over the list, each lane with
// its own column of the word-major generator pool.
__global__ __launch_bounds__(kWave) void ring_redo_kernel(Params P, ScenarioCfg C, StateView S) {
    const int lane = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = *S.redo_count;
    for (int k = lane; k < n; k += kRedoLanes) {
        const int2 job = S.redo_list[k];
        Mt19937 rng{S.ring_mt_key + lane, kRedoLanes, 0};
        generate_scenario(C, rng, (uint32_t)job.y, (size_t)job.x * P.A, S.ring_pos, nullptr, S.ring_goal, S.ring_rv);
    }
}

// ---- wave-cooperative variants (scenario_wave.h): one 64-lane workgroup per scenario -----------------------------
__global__ __launch_bounds__(kWave) void reset_wave_kernel(Params P, ScenarioCfg C, StateView S, const uint32_t* seeds,
                                                          const uint8_t* mask, uint64_t* draws) {
    __shared__ WaveScratch scratch;
    const int b = blockIdx.x;
    if (mask && !mask[b]) return;
    // the env's own numpy stream continues where the scenario left it (cn_sarl_explore): mt_key [624][B], mt_pos [B]
    const uint64_t n = generate_scenario_wave(C, scratch, seeds[b], (size_t)b * P.A, S.pos, S.vel, S.goal, S.rv, S.mt_key + b,
                                              P.B, &S.mt_pos[b]);
    if (threadIdx.x == 0) {
        S.gtime[b] = 0.0;
        S.theta[b] = 1.5707963267948966;
        if (draws) draws[b] = n;
    }
    if (P.kd)
        for (int a = 1 + threadIdx.x; a < P.A; a += blockDim.x) S.kd_valid[(size_t)b * P.A + a] = 0;
}

__global__ __launch_bounds__(kWave) void rollout_begin_wave_kernel(Params P, ScenarioCfg C, StateView S, RolloutView R) {
    __shared__ WaveScratchFill scratch;
    const int b = blockIdx.x;
    const cn_rollout_io io = *R.io;
    const int64_t c0 = episode_id(io, b, 0);
    const bool on = io.episode_limit < 0 || c0 < io.episode_limit;
    if (threadIdx.x == 0) {
        io.active[b] = on ? kRunning : kRetired;
        S.ep_word[b] = on ? kRunning : kRetired;
        io.ep_count[b] = 0;
        io.cur_steps[b] = 0;
        io.cur_return[b] = 0.0;
        if (io.cur_danger) io.cur_danger[b] = 0;
        if (io.cur_danger_dmin_sum) io.cur_danger_dmin_sum[b] = 0.0;
        S.ring_filled_in[b] = 0;
        S.ring_filled_out[b] = 0;
        S.gtime[b] = 0.0;
        S.mt_pos[b] = -1;
    }
    if (P.kd)
        for (int a = 1 + threadIdx.x; a < P.A; a += blockDim.x) S.kd_valid[(size_t)b * P.A + a] = 0;
    if (P.async_fill)
        for (int t = threadIdx.x; t < P.ring_depth; t += blockDim.x) {  // nothing resident, nothing claimed
            S.ring_ready[(size_t)b * P.ring_depth + t] = 0;
            S.ring_claim[(size_t)b * P.ring_depth + t] = 0;
        }
    if (!on) return;
    generate_scenario_wave(C, scratch, episode_seed(io, c0), (size_t)b * P.A, S.pos, S.vel, S.goal, S.rv);
}

// A seeded scenario is a pure function of its seed, and the reference's own evaluation phases replay a fixed set of them
// ('val': 100 cases, 'test': 500; crowd_sim.py:272-283).  With 20 humans on the 4 m circle 60 % of all rejection-sampling
// attempts of the 1021 bench seeds belong to the ten hardest (up to 3.1 M attempts = 118 ms of one wave, each time) — so a
// rollout whose episode seeds come from at most kScenarioCacheMax values keeps every scenario it has generated: the first
// workgroup that needs seed index k generates it into the ring slot and copies it to the cache (release), later ones copy it
// from there (acquire).  Two workgroups may generate the same seed at the same time: they write the same bits.
constexpr int kScenarioCacheMax = 4096;
template <class Scratch>
__device__ __forceinline__ void cached_scenario_wave(const Params& P, const ScenarioCfg& C, const StateView& S, Scratch& scratch,
                                                     const cn_rollout_io& io, int64_t c, size_t base) {
    const int lane = threadIdx.x;
    const bool use = S.cache_n > 0;
    const int k = use ? (int)((uint64_t)c % io.seed_mod) : 0;
    if (use && __hip_atomic_load(&S.cache_state[k], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 1) {
        if (lane < P.A) {
            const size_t ci = (size_t)k * P.A + lane;
            S.ring_pos[base + lane] = S.cache_pos[ci];
            S.ring_goal[base + lane] = S.cache_goal[ci];
            S.ring_rv[base + lane] = S.cache_rv[ci];
        }
        return;
    }
    generate_scenario_wave(C, scratch, episode_seed(io, c), base, S.ring_pos, nullptr, S.ring_goal, S.ring_rv);
    if (!use) return;
    __syncthreads();  // (lane 0 wrote the slot; same workgroup: visible after the barrier)
    if (lane < P.A) {
        const size_t ci = (size_t)k * P.A + lane;
        S.cache_pos[ci] = S.ring_pos[base + lane];
        S.cache_goal[ci] = S.ring_goal[base + lane];
        S.cache_rv[ci] = S.ring_rv[base + lane];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (lane == 0) __hip_atomic_store(&S.cache_state[k], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(kWave) void ring_fill_wave_kernel(Params P, ScenarioCfg C, StateView S, RolloutView R) {
    __shared__ WaveScratchFill scratch;
    const int idx = blockIdx.x;  // (env, ring slot)
    const int D = P.ring_depth;
    const int b = idx / D, slot = idx - b * D;
    const cn_rollout_io io = *R.io;
    const int state = io.active[b];
    const int next = io.ep_count[b] + (state == kWaitingScenario ? 0 : 1);
    if (slot == 0 && threadIdx.x == 0) S.ring_filled_out[b] = next + D;
    if (state == kRetired) return;
    const int ordinal = next + ((slot - next % D) + D) % D;
    if (ordinal < S.ring_filled_in[b]) return;
    const int64_t c = episode_id(io, b, ordinal);
    if (io.episode_limit >= 0 && c >= io.episode_limit) return;
    cached_scenario_wave(P, C, S, scratch, io, c, ((size_t)b * D + slot) * P.A);
}

// Asynchronous flavour (CN_FLAG_ASYNC_SCENARIO_FILL): runs on a side stream NEXT to the transition kernel.  A slot is
// claimed for the ordinal the env will need there (an earlier fill launch may still be working on it: atomicCAS on
// ring_claim), generated, and published with a device-scope release store of ordinal + 1 to ring_ready.  Only slots whose
// scenario the env consumed before the transition kernel running beside the fill was launched are ever overwritten.
__global__ __launch_bounds__(kWave) void ring_fill_wave_async_kernel(Params P, ScenarioCfg C, StateView S, RolloutView R) {
    __shared__ WaveScratchFill scratch;
    __shared__ int go;
    const int idx = blockIdx.x;  // workgroup = (env, slot)
    const int D = P.ring_depth;
    const int b = idx / D, slot = idx - b * D;
    const cn_rollout_io* io = R.io;
    // (state, episodes finished) of the env as ONE word: the transition kernel running beside this launch stores it once
    // when it ends — two separate loads of io->active / io->ep_count could pair an old state with a new count and claim a
    // slot whose scenario has not been consumed yet
    const int word = __hip_atomic_load(&S.ep_word[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int state = word & 3;
    if (state == kRetired) return;
    const int next = (word >> 2) + (state == kWaitingScenario ? 0 : 1);
    const int ordinal = next + ((slot - next % D) + D) % D;
    const int64_t c = episode_id(*io, b, ordinal);
    if (io->episode_limit >= 0 && c >= io->episode_limit) return;
    if (threadIdx.x == 0) {
        const int want = ordinal + 1;
        const int have = S.ring_claim[idx];
        go = (have < want && atomicCAS(&S.ring_claim[idx], have, want) == have) ? 1 : 0;
    }
    __syncthreads();
    if (!go) return;  // resident already, or another launch is generating exactly this scenario
    cached_scenario_wave(P, C, S, scratch, *io, c, ((size_t)b * D + slot) * P.A);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // (a cached copy is written by A lanes, not by lane 0 alone)
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&S.ring_ready[idx], ordinal + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// The same fill as a WORK LIST (round 6; the default) instead of one workgroup per (env, slot) — 4096 x 144 = 590 k workgroups of
// which a fifth have something to do cost a 999-step call 4.5 ms of dispatch beside its transition kernel.  Two launches on the
// fill stream: ring_fill_scan_kernel examines the (env, urgency) items — urgency u = how many episodes ahead of the env's next
// one the scenario lies — one per lane, claims the slots exactly as above and appends the claimed (env, ordinal) pairs to a job
// list, URGENCY-MAJOR (every env's nearest missing scenario before anybody's far one: a hard scenario — tens of milliseconds
// of one wave — far ahead in the ring has the whole ring's worth of steps to finish); ring_fill_jobs_kernel is a fixed grid of
// persistent one-wave generator workgroups that pop kFillJobBatch jobs at a time.  No workgroup ever waits for another one.
// Measured (r06, 4096 x 20 on the 4 m circle, 999-step calls): every scenario generated afresh 91.9 -> 94.3 M env-steps/s,
// scenario cache on 125.2 -> 139.4 M; 256 / 512 / 1024 / 2048 generator workgroups: 74.7 / 86.3 / 94.3 / 89.1 M.
//   list: int [2 + 2 * B * D]: [0] jobs appended, [1] jobs popped, then the (env, ordinal) pairs; [0] and [1] are zeroed on the
//   fill stream in front of the scan (one list per side stream: fill launches overlap)
__global__ __launch_bounds__(256) void ring_fill_scan_kernel(Params P, StateView S, RolloutView R, int* list) {
    const int D = P.ring_depth;
    const int total = P.B * D;
    const cn_rollout_io* io = R.io;
    const int item = blockIdx.x * blockDim.x + threadIdx.x;
    const int u = item / P.B, b = item - u * P.B;
    int ordinal = 0;
    bool claimed = false;
    if (item < total) {
        const int word = __hip_atomic_load(&S.ep_word[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (state, episodes finished)
        const int state = word & 3;
        if (state != kRetired) {
            ordinal = (word >> 2) + (state == kWaitingScenario ? 0 : 1) + u;
            const int64_t c = episode_id(*io, b, ordinal);
            if (!(io->episode_limit >= 0 && c >= io->episode_limit)) {
                const int idx = b * D + ordinal % D, want = ordinal + 1;
                const int have = S.ring_claim[idx];
                claimed = have < want && atomicCAS(&S.ring_claim[idx], have, want) == have;
            }
        }
    }
    // one atomic per wave: the wave's claimed items go to consecutive list entries in lane (= env) order
    const unsigned long long m = __ballot(claimed);
    if (m == 0ull) return;
    const int lane = threadIdx.x & (kWave - 1);
    int base = 0;
    if (lane == __ffsll((long long)m) - 1) base = atomicAdd(&list[0], __popcll(m));
    base = __shfl(base, __ffsll((long long)m) - 1);
    if (claimed) {
        const int k = base + __popcll(m & ((1ull << lane) - 1ull));
        list[2 + 2 * k] = b;
        list[3 + 2 * k] = ordinal;
    }
}

constexpr int kFillJobBatch = 4;  // jobs a generator workgroup pops at a time
__global__ __launch_bounds__(kWave) void ring_fill_jobs_kernel(Params P, ScenarioCfg C, StateView S, RolloutView R, int* list) {
    __shared__ WaveScratchFill scratch;
    const int lane = threadIdx.x;
    const int D = P.ring_depth;
    const cn_rollout_io* io = R.io;
    const int jobs = list[0];  // (complete: the scan kernel ran in front of this one on the same stream)
    int first = 0;
    do {
        if (lane == 0) first = atomicAdd(&list[1], kFillJobBatch);
        first = __shfl(first, 0);
        // lane l < kFillJobBatch holds job first + l; the batch's scenarios are generated one after the other by the whole wave
        int jb = 0, jord = 0;
        const bool have = lane < kFillJobBatch && first + lane < jobs;
        if (have) jb = list[2 + 2 * (first + lane)], jord = list[3 + 2 * (first + lane)];
        unsigned long long todo = __ballot(have);
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1ull;
            const int b = __shfl(jb, src), ord = __shfl(jord, src);
            const int idx = b * D + ord % D;
            cached_scenario_wave(P, C, S, scratch, *io, episode_id(*io, b, ord), (size_t)idx * P.A);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // (a cached copy is written by A lanes, not by lane 0 alone)
            __syncthreads();
            if (lane == 0) __hip_atomic_store(&S.ring_ready[idx], ord + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    } while (first + kFillJobBatch < jobs);
}

// Is scenario `ordinal` of env b resident?  Synchronous fill: the launch-time fill level; asynchronous: the slot's own flag
// (acquire at device scope: the scenario data written by the concurrently running fill kernel is visible after it).
__device__ __forceinline__ bool scenario_ready(const Params& P, const StateView& S, int env, int ordinal, int ring_filled) {
    if (!P.async_fill) return ordinal < ring_filled;
    const int* flag = S.ring_ready + (size_t)env * P.ring_depth + ordinal % P.ring_depth;
    return __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == ordinal + 1;
}

__device__ __forceinline__ void load_from_ring(const Params& P, const StateView& S, const Lane& L, int slot,
                                               AgentRegs& r) {
    const size_t ri = ((size_t)L.env * P.ring_depth + slot) * P.A + L.a;
    const double2 p = S.ring_pos[ri], g = S.ring_goal[ri], q = S.ring_rv[ri];
    r.px = p.x, r.py = p.y, r.vx = 0.0, r.vy = 0.0, r.gx = g.x, r.gy = g.y, r.rad = q.x, r.vpref = q.y;
}

// Episode end on the robot lane (explorer.py:50-72): append the record, pick the next episode.  Returns the
// per-env flag: 0 = stop stepping (retired, or waiting for the ring to be refilled), 2 + slot = load ring slot.
__device__ __forceinline__ int finish_episode(const Params& P, const StateView& S, const RolloutView R, int env, int ring_depth,
                                          int ring_filled, double time_limit, int info, double gtime, int& ep_count,
                                          int cur_steps, double cur_return, int cur_danger, double cur_dsum,
                                          int& state) {
    const cn_rollout_io io = *R.io;
    if (io.record_capacity > 0) {
        const size_t k = (size_t)env * io.record_capacity + (ep_count % io.record_capacity);
        if (io.ep_outcome) io.ep_outcome[k] = (uint8_t)info;
        if (io.ep_steps) io.ep_steps[k] = cur_steps;
        if (io.ep_return) io.ep_return[k] = cur_return;
        if (io.ep_time) io.ep_time[k] = (info == CN_TIMEOUT) ? time_limit : gtime;
        if (io.ep_danger) io.ep_danger[k] = cur_danger;
        if (io.ep_danger_dmin_sum) io.ep_danger_dmin_sum[k] = cur_dsum;
    }
    ++ep_count;
    const int64_t c = episode_id(io, env, ep_count);
    if (io.episode_limit >= 0 && c >= io.episode_limit) {
        state = kRetired;
        return 0;
    }
    if (scenario_ready(P, S, env, ep_count, ring_filled)) return 2 + ep_count % ring_depth;
    state = kWaitingScenario;  // ring ran dry (or this scenario is still being generated): pause until a later launch
    return 0;
}

// ---------------------------------------------------------------------------------------------- launch epilogue
// Behind every rollout launch there used to be three more kernels on the stream: rollout_finish_kernel (the transitions
// counter), records_pack_kernel (the shard's record blocks) and records_summary_kernel (explorer.py:74-90) — 18 us of
// kernels plus their boundaries behind a 106 us launch in the driver's 20-step shape.  They are the tail of the rollout
// kernel now: every workgroup leaves its share (its envs' record blocks; the sums over its envs' record rings and its
// transitions as one partial record), takes an arrival ticket of its group (workgroup b -> counter b % 32: 64 arrivals per counter at
// 2048 workgroups; every counter on its own 256-byte line — nine counters in ONE line serialised all 2048 arrivals of a
// 1-step launch: 35 us instead of 17), the last arrival of a group adds the group's partial sums in workgroup order and
// arrives at the top counter, and the last of those writes the results.  The summation
// order is a function of the workgroup indices only, never of the arrival order: the same bits on every run.
// Hand-off between workgroups (MI355X_MICROARCH.md, inter-workgroup visibility): payload as 8-byte agent-scope atomic
// stores (write-through), s_waitcnt vmcnt(0), then the ticket (agent-scope RMW); the reader loads the payload with
// agent-scope atomic loads after its own ticket came back.  No L2 write-back / L1 invalidate fences.
__device__ __forceinline__ void agent_store(double* p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double agent_load(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT));
}

// scratch: LDS, at least (E * (CN_SUMMARY_FIELDS + 1) + 1) doubles, free after the last barrier of the step loop.
// Called by every thread of the workgroup (contains barriers); `transitions` / `ep_count` are read on the robot lanes.
//   extra_env (>= 0, robot lanes): an env that is NOT part of this launch (the 3-of-4 schedule's last sub-launch) whose record
//   ring this workgroup adds to its sums, so that the statistics the launch leaves behind cover every env of the engine
__device__ __forceinline__ void rollout_epilogue(const Params& P, const StateView& S, const cn_rollout_io& io, const Lane& L,
                                                 bool robot, unsigned int transitions, int ep_count, double* scratch,
                                                 int extra_env = -1) {
    constexpr int F = CN_SUMMARY_FIELDS + 1;  // the eight sums + this launch's transitions (exact in a double)
    const int tid = threadIdx.x;
    const int cap = io.record_capacity;
    const int held = ep_count < cap ? ep_count : cap;
    if (robot && io.blocks) {  // cn_rollout_records' block of this env
        const int K = io.blocks_records;
        double* blk = io.blocks + (size_t)L.env * (1 + (size_t)K * CN_RECORD_FIELDS);
        blk[0] = (double)ep_count;
        for (int j = 0; j < K; ++j) {
            double* rec = blk + 1 + (size_t)j * CN_RECORD_FIELDS;
            const bool have = j < held;
            const size_t k = (size_t)L.env * cap + j;
            rec[0] = (have && io.ep_outcome) ? (double)io.ep_outcome[k] : 0.0;
            rec[1] = (have && io.ep_steps) ? (double)io.ep_steps[k] : 0.0;
            rec[2] = (have && io.ep_return) ? io.ep_return[k] : 0.0;
            rec[3] = (have && io.ep_time) ? io.ep_time[k] : 0.0;
            rec[4] = (have && io.ep_danger) ? (double)io.ep_danger[k] : 0.0;
            rec[5] = (have && io.ep_danger_dmin_sum) ? io.ep_danger_dmin_sum[k] : 0.0;
        }
    }
    if (robot && io.env_transitions) io.env_transitions[L.env] += (uint64_t)transitions;  // (ABI v6) this env's own counter
    const bool want = io.summary != nullptr;  // uniform over the launch
    // nothing job-wide asked for: the launch ends here — no partial sums, no arrival tickets, no hand-off between workgroups
    // (eight dependent memory round trips of the LAST wave, ~9 of the 113 us of a 20-step launch)
    if (!want && io.transitions == nullptr) return;
    if (robot) {  // this env's sums over its record ring (cn_records_summary's fields) + its transitions
        double acc[F] = {};
        acc[F - 1] = (double)transitions;
        if (want) {
            acc[0] = (double)ep_count;
            acc[1] = (double)held;
            for (int j = 0; j < held; ++j) {
                const size_t k = (size_t)L.env * cap + j;
                const int outcome = io.ep_outcome ? (int)io.ep_outcome[k] : 0;
                acc[2] += outcome == CN_REACH_GOAL ? 1.0 : 0.0;
                acc[3] += outcome == CN_COLLISION ? 1.0 : 0.0;
                acc[4] += outcome == CN_TIMEOUT ? 1.0 : 0.0;
                acc[5] += (outcome == CN_REACH_GOAL && io.ep_time) ? io.ep_time[k] : 0.0;
                acc[6] += io.ep_return ? io.ep_return[k] : 0.0;
                acc[7] += io.ep_danger ? (double)io.ep_danger[k] : 0.0;
            }
            if (extra_env >= 0) {
                const int n2 = io.ep_count[extra_env], held2 = n2 < cap ? n2 : cap;
                acc[0] += (double)n2;
                acc[1] += (double)held2;
                for (int j = 0; j < held2; ++j) {
                    const size_t k = (size_t)extra_env * cap + j;
                    const int outcome = io.ep_outcome ? (int)io.ep_outcome[k] : 0;
                    acc[2] += outcome == CN_REACH_GOAL ? 1.0 : 0.0;
                    acc[3] += outcome == CN_COLLISION ? 1.0 : 0.0;
                    acc[4] += outcome == CN_TIMEOUT ? 1.0 : 0.0;
                    acc[5] += (outcome == CN_REACH_GOAL && io.ep_time) ? io.ep_time[k] : 0.0;
                    acc[6] += io.ep_return ? io.ep_return[k] : 0.0;
                    acc[7] += io.ep_danger ? (double)io.ep_danger[k] : 0.0;
                }
            }
        }
        const int el = L.ebase / P.A;
#pragma unroll
        for (int f = 0; f < F; ++f) scratch[el * F + f] = acc[f];
    }
    block_sync(P);
    const int f0 = want ? 0 : F - 1;  // without a summary only the transitions travel
    if (tid >= f0 && tid < F) {  // the workgroup's envs in env order
        double t = 0.0;
        for (int el = 0; el < P.E; ++el)
            if ((int)blockIdx.x * P.E + el < P.B) t += scratch[el * F + tid];
        agent_store(S.wg_partial + (size_t)blockIdx.x * F + tid, t);
    }
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__GFX9__)
#error "rollout_epilogue orders its payload stores before the ticket with s_waitcnt vmcnt(0): on gfx9 (gfx950) vmcnt counts stores; gfx10+ counts them in vscnt - use release / acquire atomics there"
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this lane's payload has left before the ticket is taken
    block_sync(P);
    int* const flag = reinterpret_cast<int*>(scratch + P.E * F);
    const int n_wg = (int)gridDim.x;
    const int g = blockIdx.x % kEpilogueGroups;
    const int members = (n_wg - g + kEpilogueGroups - 1) / kEpilogueGroups;  // workgroups b with b % groups == g
    if (tid == 0)
        flag[0] = __hip_atomic_fetch_add(&S.tickets[g * kTicketStride], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
                  (unsigned)(members - 1);
    block_sync(P);
    if (!flag[0]) return;
    // last arrival of group g: the group's sums, members in index order (lane l: members l, l + 64, ..; lanes by a fixed tree)
    if (tid < kWave) {
        double acc[F] = {};
        for (int m = tid; m < members; m += kWave) {
            const double* p = S.wg_partial + (size_t)(g + kEpilogueGroups * m) * F;
#pragma unroll
            for (int f = 0; f < F; ++f)
                if (f >= f0) acc[f] += agent_load(p + f);
        }
#pragma unroll
        for (int f = 0; f < F; ++f) {
            if (f < f0) continue;
            double v = acc[f];
#pragma unroll
            for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off);
            if (tid == 0) agent_store(S.group_partial + g * F + f, v);
        }
    }
    const int groups = n_wg < kEpilogueGroups ? n_wg : kEpilogueGroups;
    if (tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        flag[1] = __hip_atomic_fetch_add(&S.tickets[kEpilogueGroups * kTicketStride], 1u, __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(groups - 1);
    }
    block_sync(P);
    if (!flag[1]) return;
    // last arrival of all: results out, counters back to zero for the next launch (ordered by the kernel boundary)
    if (tid >= f0 && tid < F) {
        double v = 0.0;
        for (int gg = 0; gg < groups; ++gg) v += agent_load(S.group_partial + gg * F + tid);
        if (tid < F - 1) {
            io.summary[tid] = v;
        } else if (io.transitions && v != 0.0) {
            *io.transitions += (uint64_t)v;
        }
    }
    if (tid <= kEpilogueGroups) S.tickets[tid * kTicketStride] = 0u;
}

// Up to n_steps transitions per running env in one launch; state lives in VGPRs between steps, finished envs
// take their next scenario from the ring.
// The shard's kernel (HEADLINE instantiation of rollout_kernel<10>) is compiled for THREE resident waves per SIMD (<= 168
// VGPRs) and uses the compact LDS layout (carve<.., COMPACT>: 12 workgroups per CU); the generic 10-half-plane instantiations are
// left to the register allocator (two waves).
constexpr int kGeom20Waves = 3;
// DYNAMIC SCHEDULE of the shard's kernel (Params::sched == kSchedDynamic; round 5).  The static 3-of-4 schedule needs every
// workgroup of a sub-launch resident in ONE round: a single slot held by somebody else — a scenario-generator workgroup of the
// asynchronous fill that runs for milliseconds — sends a step workgroup to a second round and doubles the sub-launch.  Here the
// grid is a set of PERSISTENT workgroups (at most the resident slots) that take (env, visit)
// items from a device queue: a call of n steps is V = Params::dyn_visits visits of n / V steps per env, queued visit-major (all
// envs' visit 0, then all visit 1, ...), so the chip is busy for V B / G "rounds" of n / V steps whatever G is, with no launch
// boundary (and no slowest-wave tail) in between; short visits (~56 steps) also even out the envs that sit in a jam.  An env's
// visits must run in order: a workgroup that takes visit k of an env waits until dyn_queue[1 + env] == k (release / acquire at agent scope around the env's state in HBM).  No deadlock: an
// item's predecessor was dequeued earlier, i.e. is held by a workgroup that is running.
constexpr int kSchedDynamic = 100;
template <int MAXL, bool UNI, bool HEADLINE = false, bool KD = false>
__global__ __launch_bounds__(kMaxBlock, (MAXL == 10 && HEADLINE ? kGeom20Waves : 1)) void rollout_kernel(Params P_in, const StateView* Sd, const int* ring_filled_in,
                                                            RolloutView R, int n_steps, const double* ext_action) {
    // The ~20 state pointers are needed before and after the step loop and when an episode ends, never inside a step: they
    // are re-read from the engine's device copy of the StateView there (scalar loads) instead of holding 40 SGPRs — spilled
    // into VGPR lanes, and at 10 half-planes pushing vector registers into scratch — across the loop.  ring_filled_in is the
    // one pointer the host swaps between launches (fill_ring_if_needed), hence a direct argument.
    // HEADLINE: the geometry of BASELINE configs[1] (5 humans + robot, 2 envs per 64-lane workgroup) as compile-time
    // constants — the pair loops become single passes, the 5-candidate rank loop unrolls, divisions by A / NC fold:
    // 706 -> 740 M env-steps/s.  Every other geometry runs the generic instantiation.
    Params P = P_in;
    if (HEADLINE && MAXL == 5) {
        P.A = 6, P.NC = 5, P.E = 2, P.nA = 12, P.pairs = 60, P.threads = 64;
    }
    if (HEADLINE && MAXL == 10) {  // BASELINE configs[3]'s shard: 20 humans + robot, one env per 64-lane workgroup, 10 neighbours
        // kept of 20 candidates — every LDS offset an immediate, no scalar registers for the layout (the generic instantiation
        // spills 200+ SGPRs into VGPR lanes)
        P.A = 21, P.NC = 20, P.E = 1, P.nA = 21, P.pairs = 420, P.threads = 64, P.kd = KD ? 1 : 0;
        P.orca.max_neighbors = 10;
        P.kdl = kd_layout(21, 21, 1);
    }
    constexpr bool COMPACT = HEADLINE && MAXL == 10;  // the shard kernel's LDS layout (carve)
    // the float64 parameters of a step as VALU operands live in VGPRs (rollout_fused.h: in_vgpr): as SGPR kernel arguments the
    // 16-dword block was spilled into VGPR lanes and re-read with v_readlane several times per step (COMPACT: in LDS instead)
    if (!COMPACT) {
        auto pin = [](double& x) { asm volatile("" : "+v"(x)); };
        pin(P.dt), pin(P.time_limit), pin(P.success_reward), pin(P.collision_penalty), pin(P.discomfort_dist);
        pin(P.discomfort_factor), pin(P.human_safety);
    }
    // The shard's kernel is compiled for three resident waves per SIMD: 3072 workgroups fill the chip, 4096 envs would run as
    // a full round and a third of one.  launch_rollout therefore splits a call of 3 q steps into FOUR launches of q steps over
    // 3 B / 4 workgroups: sub-launch k leaves out env 3 - k of every group of four (ABC, ABD, ACD, BCD), so every env makes
    // its 3 q steps, in order, and every launch is exactly one round of the chip.
    int env_block = -1, extra_env = -1;
    if (HEADLINE && MAXL == 10 && P.sched >= 0) {
        const int g = (int)blockIdx.x / 3, rr = (int)blockIdx.x - 3 * g, skip = 3 - P.sched;
        env_block = 4 * g + rr + (rr >= skip ? 1 : 0);
        if (P.sched == 3 && rr == 0) extra_env = 4 * g;  // the last sub-launch reports for the env it leaves out as well
    }
    const Smem s = carve<MAXL, COMPACT>(P);
    if (COMPACT && threadIdx.x == 0) {
        s.disc[kParDt] = P.dt, s.disc[kParLimit] = P.time_limit, s.disc[kParSuccess] = P.success_reward;
        s.disc[kParCollision] = P.collision_penalty, s.disc[kParDDist] = P.discomfort_dist;
        s.disc[kParDFactor] = P.discomfort_factor, s.disc[kParHSafety] = P.human_safety;
    }
    constexpr bool kDyn = HEADLINE && MAXL == 10;
    const bool dynamic = kDyn && P.sched == kSchedDynamic;
    if (kDyn && threadIdx.x == 0) reinterpret_cast<EpisodeLds*>(s.disc + 8)->dyn_total = 0u;
    const int n_steps_call = n_steps;
    for (int visit_iter = 0; dynamic || visit_iter == 0; ++visit_iter) {
    int dyn_env = 0, dyn_k = 0;
    if (kDyn && dynamic) {
        __syncthreads();  // (the previous visit's last LDS reads are done)
        if (threadIdx.x == 0) {
            int* const queue = Sd->dyn_queue;
            const int v = atomicAdd(queue, 1);
            if (v < P.dyn_visits * P.B) {  // wait for the env's previous visit: its state is in memory once the flag says so
                const int env = v % P.B, k = v / P.B;
                int spins = 0;
                while (__hip_atomic_load(queue + 1 + env, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < k && spins < (1 << 22)) {
                    __builtin_amdgcn_s_sleep(16);
                    ++spins;
                }
                // ~2 s: never in a healthy run; cn_sync reports it.  The visit is then NOT run on state its predecessor may still
                // be writing: it is handed on as if complete, so that the env's later visits do not wait in turn —
                // the env loses those steps, the error bit says so
                if (spins == (1 << 22)) {
                    atomicOr(Sd->error, 4);
                    __hip_atomic_store(queue + 1 + env, k + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    s.flag[2] = 1;
                } else {
                    s.flag[2] = 0;
                }
            }
            s.flag[1] = v;
        }
        __syncthreads();
        const int v = s.flag[1];
        if (v >= P.dyn_visits * P.B) break;
        if (s.flag[2] != 0) continue;  // (the wait gave up)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // every wave: what the previous visit's workgroup wrote is visible
        dyn_env = v % P.B, dyn_k = v / P.B;
        env_block = dyn_env;
        const int q = n_steps_call / P.dyn_visits, rem = n_steps_call - q * P.dyn_visits;
        n_steps = q + (dyn_k < rem ? 1 : 0);
    }
    const Lane L = lane_of(P, env_block);
    AgentRegs r = {};
    float robot_max_speed = 0.0f;
    const bool robot = L.valid && L.a == 0;
    double theta = 0.0;  // heading of a unicycle robot (external actions only)
    double gtime = 0.0, cur_return = 0.0, cur_dsum = 0.0;
    int cur_steps = 0, cur_danger = 0, ep_count = 0, ring_filled = 0, state = kRetired;
    {
        const StateView S = *Sd;
        if (L.valid) load_agent(S, L.gi, r);
        if (P.robot_orca) load_robot_view(P, S, s, L, r, robot_max_speed);
        if (KD) kd_load(P, S, s, L);
        if (robot) theta = S.theta[L.env];
    }
    if (visit_iter == 0) build_pairs<COMPACT>(P, s, dynamic);
    if (robot) {
        const StateView S = *Sd;
        const cn_rollout_io io = *R.io;
        gtime = S.gtime[L.env];
        state = io.active[L.env];
        ep_count = io.ep_count[L.env];
        cur_steps = io.cur_steps[L.env];
        cur_return = io.cur_return[L.env];
        if (io.cur_danger) cur_danger = io.cur_danger[L.env];
        if (io.cur_danger_dmin_sum) cur_dsum = io.cur_danger_dmin_sum[L.env];
        ring_filled = ring_filled_in[L.env];
        int f = state == kRunning ? 1 : 0;
        if (state == kWaitingScenario && scenario_ready(P, S, L.env, ep_count, ring_filled)) {  // produced since
            f = 2 + ep_count % P.ring_depth;
            state = kRunning;
            gtime = 0.0;
        }
        s.flag[L.lane] = f;
    }
    __syncthreads();
    if (L.valid && s.flag[L.ebase] >= 2) {
        load_from_ring(P, *Sd, L, s.flag[L.ebase] - 2, r);
        if (KD) kd_new_episode(P, s, L);
        theta = 1.5707963267948966;
    }
    // COMPACT: what an agent keeps for a whole episode goes to LDS here and whenever the env loads its next scenario
    const auto stage_constants = [&]() {
        if (L.lane < P.nA) {
            s.goal2[L.lane] = make_double2(r.gx, r.gy);
            s.vpref[L.lane] = r.vpref;
            s.rad[L.lane] = r.rad;
            s.hview[L.lane] = (float)(r.rad + 0.01 + P.human_safety);
        }
    };
    if (COMPACT) stage_constants();
    unsigned int transitions = 0;
    if (!COMPACT)
        for (int t = threadIdx.x; t < kMaxDiscount; t += blockDim.x) s.disc[t] = t < R.discount_len ? R.discount[t] : 0.0;
    // COMPACT: the robot lane's episode bookkeeping waits in LDS while a transition is computed (12 VGPRs the step loop
    // does not hold across the pair and solve phases)
    EpisodeLds* const eps = reinterpret_cast<EpisodeLds*>(s.disc + 8);
    if (COMPACT && robot) {
        eps->gtime = gtime, eps->cur_return = cur_return, eps->cur_dsum = cur_dsum;
        eps->cur_steps = cur_steps, eps->cur_danger = cur_danger, eps->ep_count = ep_count;
        eps->ring_filled = ring_filled, eps->state = state, eps->transitions = 0u;
    }
    __syncthreads();
    const int disc_len = R.discount_len < kMaxDiscount ? R.discount_len : kMaxDiscount;

#ifdef CN_PHASE_TIMING
    PhaseClock clock = {};
    PhaseClock* clk = &clock;
    clock.last = __builtin_readcyclecounter();
#else
    PhaseClock* clk = nullptr;
#endif
    for (int step = 0; step < n_steps; ++step) {
        // The lane ids of this step are opaque to the compiler: everything derived from them — LDS addresses of every phase —
        // is recomputed in the step instead of being hoisted out of the step loop and held in VGPRs across it (step_kernel,
        // the same code without the loop, needs 92 VGPRs; this kernel needed 182 before).
        Lane Ls = L;
        asm volatile("" : "+v"(Ls.lane), "+v"(Ls.a), "+v"(Ls.ebase));
        Ls.valid = L.valid && s.flag[L.ebase] != 0;  // env is running

        StepResult res;
        double nvx, nvy;
        if constexpr (COMPACT) {
            // no LDS copy of the discount table: this step's factor is requested here, a whole transition before its use
            double disc_now = 0.0, gt = 0.0;
            if (robot) {
                const int cs = eps->cur_steps;
                gt = eps->gtime;
                if (cs < disc_len) disc_now = R.discount[cs];
            }
            step_core<MAXL, UNI, KD, true>(P, s, Ls, r, gt, robot_max_speed, ext_action, 1, res, nvx, nvy, &theta, clk);
            if (robot && eps->state == kRunning) {
                int next_flag = 1;
                int e_steps = eps->cur_steps, e_danger = eps->cur_danger;
                double e_return = eps->cur_return, e_dsum = eps->cur_dsum;
                eps->transitions += 1u;
                e_return = e_return + disc_now * res.reward;  // python sum(): left to right
                ++e_steps;
                if (res.info == CN_DANGER) {
                    ++e_danger;
                    e_dsum += res.dmin;
                }
                if (res.done) {
                    int e_count = eps->ep_count, e_state = kRunning;
                    next_flag = finish_episode(P, *Sd, R, L.env, P.ring_depth, eps->ring_filled, s.disc[kParLimit], res.info, gt,
                                               e_count, e_steps, e_return, e_danger, e_dsum, e_state);
                    eps->ep_count = e_count, eps->state = e_state;
                    e_steps = 0, e_return = 0.0, e_danger = 0, e_dsum = 0.0;
                    gt = 0.0;
                }
                eps->gtime = gt;
                eps->cur_steps = e_steps, eps->cur_danger = e_danger, eps->cur_return = e_return, eps->cur_dsum = e_dsum;
                s.flag[L.lane] = next_flag;
            }
        } else {
            step_core<MAXL, UNI, KD>(P, s, Ls, r, gtime, robot_max_speed, ext_action, 1, res, nvx, nvy, &theta, clk);
            if (robot && state == kRunning) {
                int next_flag = 1;
                ++transitions;
                const double disc = cur_steps < kMaxDiscount ? s.disc[cur_steps] : 0.0;
                cur_return = cur_return + disc * res.reward;  // python sum(): left to right
                ++cur_steps;
                if (res.info == CN_DANGER) {
                    ++cur_danger;
                    cur_dsum += res.dmin;
                }
                if (res.done) {
                    next_flag = finish_episode(P, *Sd, R, L.env, P.ring_depth, ring_filled, P.time_limit, res.info, gtime, ep_count,
                                               cur_steps, cur_return, cur_danger, cur_dsum, state);
                    cur_steps = 0, cur_return = 0.0, cur_danger = 0, cur_dsum = 0.0;
                    gtime = 0.0;
                }
                s.flag[L.lane] = next_flag;
            }
        }
        __syncthreads();
        if (L.valid && s.flag[L.ebase] >= 2) {
            load_from_ring(P, *Sd, L, s.flag[L.ebase] - 2, r);
            if (KD) kd_new_episode(P, s, L);
            theta = 1.5707963267948966;  // robot.set(..., np.pi / 2)
            if (COMPACT) stage_constants();
        }
        CN_TICK(clk, 7);
    }
    if (COMPACT && L.lane < P.nA) {
        const double2 g = s.goal2[L.lane];
        r.gx = g.x, r.gy = g.y, r.vpref = s.vpref[L.lane], r.rad = s.rad[L.lane];
    }
    if (COMPACT && robot) {
        gtime = eps->gtime, cur_return = eps->cur_return, cur_dsum = eps->cur_dsum;
        cur_steps = eps->cur_steps, cur_danger = eps->cur_danger, ep_count = eps->ep_count;
        state = eps->state, transitions = eps->transitions;
    }
#ifdef CN_PHASE_TIMING
    if ((threadIdx.x & (kWave - 1)) == 0) {
        for (int k = 0; k < 10; ++k) atomicAdd(&cn_phase_cycles[k], clock.acc[k]);
        atomicAdd(&cn_phase_cycles[15], 1ull);  // waves
    }
#endif

    const StateView S = *Sd;
    if (KD) kd_store(P, S, s, L);
    if (L.valid) {
        S.pos[L.gi] = make_double2(r.px, r.py);
        S.vel[L.gi] = make_double2(r.vx, r.vy);
        S.goal[L.gi] = make_double2(r.gx, r.gy);
        S.rv[L.gi] = make_double2(r.rad, r.vpref);
    }
    if (robot) {
        const cn_rollout_io io = *R.io;
        S.gtime[L.env] = gtime;
        S.theta[L.env] = theta;
        if (P.robot_orca) S.rsim_valid[L.env] = 1;
        io.active[L.env] = (uint8_t)state;
        io.ep_count[L.env] = ep_count;
        io.cur_steps[L.env] = cur_steps;
        io.cur_return[L.env] = cur_return;
        if (io.cur_danger) io.cur_danger[L.env] = cur_danger;
        if (io.cur_danger_dmin_sum) io.cur_danger_dmin_sum[L.env] = cur_dsum;
        S.ep_word[L.env] = (ep_count << 2) | state;
    }
    if (kDyn && dynamic) {
        if (robot) {
            const cn_rollout_io io = *R.io;
            if (io.env_transitions) io.env_transitions[L.env] += (uint64_t)transitions;
            eps->dyn_total += transitions;
        }
        // release: this env's state, bookkeeping and kd rows are in memory before its next visit may start anywhere
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (threadIdx.x == 0)
            __hip_atomic_store(Sd->dyn_queue + 1 + dyn_env, dyn_k + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        continue;
    }
    rollout_epilogue(P, S, *R.io, L, robot, transitions, ep_count, reinterpret_cast<double*>(s.lines), extra_env);
    }  // visits
    if (kDyn && dynamic) {
        // the job-wide transitions counter, if the caller keeps one (per-env counters were added visit by visit; the in-kernel
        // summary / record blocks are not offered under the dynamic schedule: launch_rollout falls back to the static one)
        __syncthreads();
        cn_rollout_io io = *R.io;
        io.blocks = nullptr, io.summary = nullptr, io.env_transitions = nullptr;
        Lane Le;
        Le.lane = threadIdx.x, Le.env = (int)blockIdx.x, Le.a = threadIdx.x == 0 ? 0 : 1, Le.ebase = 0, Le.valid = true, Le.gi = 0;
        const unsigned int dyn_transitions = threadIdx.x == 0 ? reinterpret_cast<EpisodeLds*>(s.disc + 8)->dyn_total : 0u;
        rollout_epilogue(P, *Sd, io, Le, threadIdx.x == 0, dyn_transitions, 0, reinterpret_cast<double*>(s.lines));
    }
}

#endif  // CN_SARL_TU

}  // namespace cn
