// SARL robot decision on device — replaces, batched over B envs x K candidate actions,
//   MultiHumanRL.predict                     /root/reference crowd_nav/policy/multi_human_rl.py:11-63
//   CrowdSim.onestep_lookahead (reward part) crowd_sim/envs/crowd_sim.py:314-389 (via step_core-compatible code)
//   CADRL.propagate / rotate                 crowd_nav/policy/cadrl.py:104-129, 187-222
//   MultiHumanRL.build_occupancy_maps        multi_human_rl.py:109-163
//   sarl.ValueNetwork.forward                crowd_nav/policy/sarl.py:28-65 (mlp(): cadrl.py:11-19)
//
// Kernels (launched back to back on the engine's stream by cn_sarl_select):
//   orca_kernel            (step_kernels.h) the humans' next velocities — computed ONCE per env: they do not
//                          depend on the candidate action
//   sarl_lookahead_kernel  lane = (env, human): next human observable states (float64) and their occupancy maps — only with_om
//                          or under LSTM-RL's re-ordering; otherwise the feature kernel derives the next states itself
//   sarl_feature_kernel    lane = (env, action, human): float32 rotated 13-vector (+48 map values) -> X
//   sarl_mlp_kernel        workgroup = 16 (env, action) groups x H humans: the whole value network on FP32 MFMA
//                          (v_mfma_f32_16x16x4_f32: exact f32, k-ordered fma chain), activations in LDS
//   sarl_select_kernel     wave = env: float64 reward of onestep_lookahead(action) (sarl_reward_of) + gamma^(dt v_pref) * V for
//                          every action, first strict maximum
//
// Row order inside an MLP tile is HUMAN-MAJOR: row = h * 16 + g (g = group within the tile).  A 16-row MFMA
// tile then holds human h of 16 different groups, so the per-group reductions of the network (mean over
// humans, masked softmax over humans, weighted feature sum) combine values at the same offset of different row
// tiles — no cross-lane traffic.
//
// Activations live in LDS in MFMA A-FRAGMENT ORDER: element (row tile rt, row r, feature n) of a buffer with
// `ks` k-steps per row tile sits at ((rt * ks + n / 4) * 64 + (n % 4) * 16 + r).  A wave then fetches the A operand
// of k-step s with lane l reading word (rt * ks + s) * 64 + l (conflict-free ds_read_b32), and the 4 accumulator
// values a lane owns (rows quad*4 .. quad*4+3 of one column) are 4 consecutive words: one ds_write_b128.
#pragma once
#include <hip/hip_runtime.h>

#include "orca_device.h"     // wave_lds_sync
#include "scenario_device.h"  // norm2

namespace cn {

// CN_PHASE_TIMING (profiling builds only): per-layer shader-clock ticks of sarl_mlp_kernel as wave 0 sees them
// (barrier to barrier), summed over tiles into cn_sarl_cycles[k]; [15] = tiles.  scripts/sarl_phase_probe.py
#ifdef CN_PHASE_TIMING
static __device__ unsigned long long cn_sarl_cycles[16];
#define CN_SARL_CLOCK_BEGIN() unsigned long long sclk_last_ = __builtin_readcyclecounter(), sclk_acc_[15] = {}
#define CN_SARL_TICK(k)                                                  \
    do {                                                                 \
        const unsigned long long now_ = __builtin_readcyclecounter();    \
        sclk_acc_[k] += now_ - sclk_last_;                               \
        sclk_last_ = now_;                                               \
    } while (0)
#define CN_SARL_CLOCK_END_N(n_)                                                          \
    do {                                                                                 \
        if (threadIdx.x == 0) {                                                          \
            for (int k_ = 0; k_ < 15; ++k_) atomicAdd(&cn_sarl_cycles[k_], sclk_acc_[k_]); \
            atomicAdd(&cn_sarl_cycles[15], (unsigned long long)(n_));                    \
        }                                                                                \
    } while (0)
#define CN_SARL_CLOCK_END() CN_SARL_CLOCK_END_N(1)
#else
#define CN_SARL_CLOCK_BEGIN() \
    do {                      \
    } while (0)
#define CN_SARL_TICK(k) \
    do {                \
    } while (0)
#define CN_SARL_CLOCK_END() \
    do {                    \
    } while (0)
#define CN_SARL_CLOCK_END_N(n_) \
    do {                        \
    } while (0)
#endif

constexpr int kWaveSize = 64;        // gfx950 wavefront
constexpr int kSarlGroups = 16;      // (env, action) groups per MLP tile = MFMA tile height
constexpr int kSarlMaxHumans = 8;    // register arrays of the occupancy map / LSTM-RL ordering; more humans: SARL without maps
                                     // (sarl_mlp_chunked_kernel streams them; one-tile kernels hold up to 5 at the shipped widths)
constexpr int kSarlThreads = 1024;   // 16 waves per MLP workgroup (4 per SIMD: one wave's LDS/L2 waits hide behind the others' MFMAs)
constexpr int kSarlKChunk = 5;       // k-steps per trip of the MFMA loop (= B fragments prefetched at a time): K = 100 is 25 k-steps
constexpr int kSarlLayers = 12;
constexpr int kSarlChunk5 = 5;       // row tiles per chunk of the streamed (any number of humans) kernels      // packed linear layers (attention.0 is split into its two K halves)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// One packed linear layer: B-operand fragments of v_mfma_f32_16x16x4_f32, fragment (ct, ks) at
// w[(ct * kpad + ks) * 64 + lane] = W[n = ct*16 + (lane & 15)][k = ks*4 + (lane >> 4)] (0 outside), so a wave
// fetches a fragment with one coalesced 256-byte load; bias padded to ctiles * 16.
struct PackedLinear {
    const float* w;
    const float* bias;
    int K, N;       // true sizes
    int ksteps;     // ceil(K / 4)
    int kpad;       // ksteps rounded up to kSarlKChunk (zero fragments): the B prefetch never runs off the end
    int ctiles;     // ceil(N / 16)
};

// k-steps per row tile of an LDS activation buffer that holds n features: whole 16-column tiles of its producer, and
// at least the consumer's k loop (ksteps rounded up to kSarlKChunk) so that the straight-line loop stays inside the row
// tile.  The kernels zero their LDS once per tile: k-steps the producer never writes are finite (zero or stale
// activations) and meet zero weights.
__host__ __device__ inline int sarl_ks(int n) {
    const int tiles = (n + 15) / 16 * 4;
    const int kpad = ((n + 3) / 4 + kSarlKChunk - 1) / kSarlKChunk * kSarlKChunk;
    return tiles > kpad ? tiles : kpad;
}

enum {
    kL_mlp1_0, kL_mlp1_2, kL_mlp2_0, kL_mlp2_2, kL_att0_local, kL_att0_global, kL_att_2, kL_att_4,
    kL_mlp3_0, kL_mlp3_2, kL_mlp3_4, kL_mlp3_6
};

struct SarlNet {
    PackedLinear L[kSarlLayers];
    int in_dim;        // 13 or 13 + cell_num^2 * om_channel_size
    int with_global;   // sarl.py:17-21
    int H;
    // k-steps per row tile of the LDS buffers (fragment order): X input, wide hidden (A), mlp1 output (B),
    // per-human feature (C), scores (S)
    int ks_x, ks_a, ks_b, ks_c, ks_s;
};

// The same network as 12 x 3 dwords (offsets into ONE weight arena) for the persistent kernel, whose loop keeps every
// descriptor live in SGPRs: 12 PackedLinear structs (pointers, sizes) do not fit and spill through VGPRs into scratch.
struct LayerRef {
    uint32_t w, b;   // float offsets of the B fragments / the bias from SarlNetRef::base
    uint32_t dims;   // kpad | ctiles << 8 | ksteps << 16
};
struct SarlNetRef {
    const float* base;
    LayerRef L[kSarlLayers];
    int nf;            // mlp2 output width (per-human feature)
    int with_global;
    int ks_x, ks_a, ks_b, ks_c, ks_s;
};
__device__ __forceinline__ PackedLinear layer_of(const SarlNetRef& n, int l) {
    const LayerRef r = n.L[l];
    PackedLinear P;
    P.w = n.base + r.w, P.bias = n.base + r.b;
    P.K = 0, P.N = 0;
    P.kpad = (int)(r.dims & 0xffu), P.ctiles = (int)((r.dims >> 8) & 0xffu), P.ksteps = (int)(r.dims >> 16);
    return P;
}

struct SarlCfg {
    int B, H, n_actions;
    int with_om, cell_num, om_channels;
    int unicycle;  // actions are ActionRot(v, r): cadrl.py:119-125, crowd_sim.py:339-341
    int cadrl;           // cadrl.ValueNetwork (the row MLP + minimum over humans) instead of sarl.ValueNetwork: sarl_narrow_kernel's branch
    int const_vel;       // query_env = false (multi_human_rl.py:39-42): humans keep their velocity, reward = compute_reward
    int sort_lookahead;  // ... and the joint state LstmRL.predict sorted by decreasing distance feeds the network (lstm_rl.py:96-103)
    double cell_size;
    double dt, time_limit, success_reward, collision_penalty, discomfort_dist, discomfort_factor;
    double gamma_bar;  // pow(gamma, time_step * v_pref), computed on the host like multi_human_rl.py:52
};

// torch.nn.Linear weight [N][K] (row-major) -> MFMA B fragments
__global__ void sarl_pack_kernel(const float* W, const float* bias, int N, int K, int k_offset, int k_count,
                                 int kpad, int ctiles, float* wp, float* bp) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = ctiles * kpad * 64;
    if (idx < total) {
        const int lane = idx & 63;
        const int frag = idx >> 6;
        const int ct = frag / kpad, ks = frag - ct * kpad;
        const int n = ct * 16 + (lane & 15);
        const int k = ks * 4 + (lane >> 4);
        wp[idx] = (n < N && k < k_count) ? W[(size_t)n * K + k_offset + k] : 0.0f;
    }
    if (idx < ctiles * 16) bp[idx] = (bias && idx < N) ? bias[idx] : 0.0f;
}

// All layers of a network in ONE launch (cn_sarl_set_weights runs once per sampled episode in the RL phase: eleven launches of
// the kernel above were ~50 us of host and device time each time).  Block b belongs to the job whose block range holds it.
constexpr int kPackJobs = 16;
struct PackJob {
    const float* W;
    const float* bias;
    float* wp;
    float* bp;
    int N, K, k_offset, k_count, kpad, ctiles, first_block;
};
struct PackJobs {
    PackJob job[kPackJobs];
    int n;
};
__global__ void sarl_pack_many_kernel(PackJobs jobs) {
    int j = 0;
    for (int i = 1; i < jobs.n; ++i) j = (int)blockIdx.x >= jobs.job[i].first_block ? i : j;
    const PackJob J = jobs.job[j];
    const int idx = ((int)blockIdx.x - J.first_block) * blockDim.x + threadIdx.x;
    const int total = J.ctiles * J.kpad * 64;
    if (idx < total) {
        const int lane = idx & 63;
        const int frag = idx >> 6;
        const int ct = frag / J.kpad, ks = frag - ct * J.kpad;
        const int n = ct * 16 + (lane & 15);
        const int k = ks * 4 + (lane >> 4);
        J.wp[idx] = (n < J.N && k < J.k_count) ? J.W[(size_t)n * J.K + J.k_offset + k] : 0.0f;
    }
    if (idx < J.ctiles * 16) J.bp[idx] = (J.bias && idx < J.N) ? J.bias[idx] : 0.0f;
}

// ------------------------------------------------------------------------------------ lookahead / reward
// Occupancy map human i sees among the H humans of one env (multi_human_rl.py:109-163; the robot is not in it).
// state_of(j, px, py, vx, vy) yields human j's state; m receives cells * channels float32 values.
// CAP = kSarlMaxHumans: the per-human slots are registers (loops fully unrolled); CAP = kSarlAnyHumans: any crowd the
// engine holds (<= 63 humans), slots in private memory, runtime trip counts.
constexpr int kSarlAnyHumans = 64;
template <int CAP, class StateOf>
__device__ __forceinline__ void occupancy_map_cap(const SarlCfg& C, int i, StateOf state_of, float* m) {
    const int cells = C.cell_num * C.cell_num;
    const int ch = C.om_channels;
    const int nh = CAP == kSarlMaxHumans ? kSarlMaxHumans : C.H;
    constexpr int kOmUnroll = CAP == kSarlMaxHumans ? kSarlMaxHumans : 1;
    double px, py, vx, vy;
    state_of(i, px, py, vx, vy);
    const double my_angle = atan2(vy, vx);
    // every other human's cell and rotated velocity once ...
    int cell_of[CAP];
    double rvx[CAP], rvy[CAP];
#pragma unroll kOmUnroll
    for (int j = 0; j < nh; ++j) {
        cell_of[j] = -1;
        rvx[j] = rvy[j] = 0.0;
        if (j >= C.H || j == i) continue;
        double qx, qy, wx, wy;
        state_of(j, qx, qy, wx, wy);
        const double ox = qx - px, oy = qy - py;
        const double rotation = atan2(oy, ox) - my_angle;
        const double dist = sqrt(ox * ox + oy * oy);
        const double rx = cos(rotation) * dist, ry = sin(rotation) * dist;
        const double xi = floor(rx / C.cell_size + C.cell_num / 2.0);
        const double yi = floor(ry / C.cell_size + C.cell_num / 2.0);
        if (xi < 0 || xi >= C.cell_num || yi < 0 || yi >= C.cell_num) continue;
        cell_of[j] = (int)(C.cell_num * yi + xi);
        const double vrot = atan2(wy, wx) - my_angle;
        const double speed = sqrt(wx * wx + wy * wy);
        rvx[j] = cos(vrot) * speed;
        rvy[j] = sin(vrot) * speed;
    }
    // ... then per cell: count, sum vx', sum vy' in visit order (python sum(): 0 + x1 + x2 ...)
    for (int cell = 0; cell < cells; ++cell) {
        double cnt = 0.0, svx = 0.0, svy = 0.0;
#pragma unroll kOmUnroll
        for (int j = 0; j < nh; ++j) {
            if (cell_of[j] != cell) continue;
            cnt += 1.0;
            svx += rvx[j];
            svy += rvy[j];
        }
        if (ch == 1) {
            m[cell] = cnt > 0.0 ? 1.0f : 0.0f;
        } else if (ch == 2) {
            m[2 * cell] = cnt > 0.0 ? (float)(svx / cnt) : 0.0f;
            m[2 * cell + 1] = cnt > 0.0 ? (float)(svy / cnt) : 0.0f;
        } else {
            m[3 * cell] = cnt > 0.0 ? (float)(cnt / cnt) : 0.0f;
            m[3 * cell + 1] = cnt > 0.0 ? (float)(svx / cnt) : 0.0f;
            m[3 * cell + 2] = cnt > 0.0 ? (float)(svy / cnt) : 0.0f;
        }
    }
}

template <class StateOf>
__device__ __forceinline__ void occupancy_map(const SarlCfg& C, int i, StateOf state_of, float* m) {
    if (C.H <= kSarlMaxHumans)
        occupancy_map_cap<kSarlMaxHumans>(C, i, state_of, m);
    else
        occupancy_map_cap<kSarlAnyHumans>(C, i, state_of, m);
}

// The same maps for the envs of ONE workgroup, spread over its lanes (sarl_decide_step_kernel: the maps of the next decision sit
// on a one-env sampling step's critical path, and a human's lane alone needs 1 + 4 x 2 float64 atan2 and 4 x 2 sincos in a
// row: +28 us).  The float64 operations of occupancy_map_cap, each exactly once and in the same expressions — only spread out:
//   phase 1  one item per human (its heading), per ordered pair (the other's bearing and distance) and per ordered pair again
//            (the other's velocity direction and speed): every atan2 / sqrt of the maps at once
//   phase 2  two items per ordered pair: cos / sin of the relative bearing -> cell; of the relative velocity direction -> (vx', vy')
//   phase 3  one item per (human, cell): count and velocity sums over the others IN INDEX ORDER (python's sum())
// so a map costs one atan2 and one sincos in a row.  state_of(e, j, px, py, vx, vy): human j of the workgroup's env e.
// scratch: (8 H + 52 H (H - 1)) bytes per env (H <= kSarlMaxHumans); sync(): the workgroup's barrier.
__host__ __device__ inline size_t occupancy_coop_bytes(int H) { return (size_t)8 * H + (size_t)52 * H * (H - 1); }
// map_of(e, i): where human i of env e's map (cells x channels floats) goes.
template <class StateOf, class Sync, class MapOf>
__device__ __forceinline__ void occupancy_maps_cooperative(const SarlCfg& C, int E, int tid, int threads, char* scratch, StateOf state_of,
                                                           Sync sync, MapOf map_of) {
    const int H = C.H, P = H * (H - 1);
    const size_t per_env = occupancy_coop_bytes(H);
    const auto ang_of = [&](int e) { return reinterpret_cast<double*>(scratch + e * per_env); };
    const auto pa_of = [&](int e) { return reinterpret_cast<double2*>(scratch + e * per_env + 8 * H); };           // (bearing, velocity direction)
    const auto pd_of = [&](int e) { return reinterpret_cast<double2*>(scratch + e * per_env + 8 * H + 16 * P); };  // (distance, speed)
    const auto pr_of = [&](int e) { return reinterpret_cast<double2*>(scratch + e * per_env + 8 * H + 32 * P); };  // (vx', vy')
    const auto pc_of = [&](int e) { return reinterpret_cast<int*>(scratch + e * per_env + 8 * H + 48 * P); };      // cell or -1
    const int n1 = H + 2 * P;
    for (int it = tid; it < E * n1; it += threads) {
        const int e = it / n1, k = it - e * n1;
        double px, py, vx, vy;
        if (k < H) {
            state_of(e, k, px, py, vx, vy);
            ang_of(e)[k] = atan2(vy, vx);
        } else {
            const int q = k - H, p = q < P ? q : q - P;
            const int i = p / (H - 1), jj = p - i * (H - 1), j = jj < i ? jj : jj + 1;
            double qx, qy, wx, wy;
            state_of(e, j, qx, qy, wx, wy);
            if (q < P) {
                state_of(e, i, px, py, vx, vy);
                const double ox = qx - px, oy = qy - py;
                pa_of(e)[p].x = atan2(oy, ox);
                pd_of(e)[p].x = sqrt(ox * ox + oy * oy);
            } else {
                pa_of(e)[p].y = atan2(wy, wx);
                pd_of(e)[p].y = sqrt(wx * wx + wy * wy);
            }
        }
    }
    sync();
    for (int it = tid; it < E * 2 * P; it += threads) {
        const int e = it / (2 * P), q = it - e * 2 * P, p = q < P ? q : q - P;
        const int i = p / (H - 1);
        const double my_angle = ang_of(e)[i];
        if (q < P) {
            const double rotation = pa_of(e)[p].x - my_angle, dist = pd_of(e)[p].x;
            const double rx = cos(rotation) * dist, ry = sin(rotation) * dist;
            const double xi = floor(rx / C.cell_size + C.cell_num / 2.0);
            const double yi = floor(ry / C.cell_size + C.cell_num / 2.0);
            pc_of(e)[p] = (xi < 0 || xi >= C.cell_num || yi < 0 || yi >= C.cell_num) ? -1 : (int)(C.cell_num * yi + xi);
        } else {
            const double vrot = pa_of(e)[p].y - my_angle, speed = pd_of(e)[p].y;
            pr_of(e)[p] = make_double2(cos(vrot) * speed, sin(vrot) * speed);
        }
    }
    sync();
    const int cells = C.cell_num * C.cell_num, ch = C.om_channels;
    for (int it = tid; it < E * H * cells; it += threads) {
        const int e = it / (H * cells), r = it - e * H * cells, i = r / cells, cell = r - i * cells;
        double cnt = 0.0, svx = 0.0, svy = 0.0;
        for (int jj = 0; jj < H - 1; ++jj) {  // the others in index order
            const int p = i * (H - 1) + jj;
            if (pc_of(e)[p] != cell) continue;
            const double2 rv = pr_of(e)[p];
            cnt += 1.0;
            svx += rv.x;
            svy += rv.y;
        }
        float* m = map_of(e, i);
        if (ch == 1) {
            m[cell] = cnt > 0.0 ? 1.0f : 0.0f;
        } else if (ch == 2) {
            m[2 * cell] = cnt > 0.0 ? (float)(svx / cnt) : 0.0f;
            m[2 * cell + 1] = cnt > 0.0 ? (float)(svy / cnt) : 0.0f;
        } else {
            m[3 * cell] = cnt > 0.0 ? (float)(cnt / cnt) : 0.0f;
            m[3 * cell + 1] = cnt > 0.0 ? (float)(svx / cnt) : 0.0f;
            m[3 * cell + 2] = cnt > 0.0 ? (float)(svy / cnt) : 0.0f;
        }
    }
}
__device__ __forceinline__ bool occupancy_coop_ok(const SarlCfg& C, size_t scratch_bytes, int envs) {
    return C.H >= 2 && C.H <= kSarlMaxHumans && (size_t)envs * occupancy_coop_bytes(C.H) <= scratch_bytes;
}

// LstmRL.predict re-orders the humans of the joint state by DEcreasing distance to the robot (lstm_rl.py:96-103; python's
// sorted(..., reverse=True) is stable: equal distances keep their original order).  Returns the human that sits at
// position p of the sorted joint state: the one whose stable rank equals p.  Distances are numpy's 2-vector norm of
// (human.position - robot.position).
__device__ inline int human_by_decreasing_distance(const double2* pos, size_t g0, int H, int p) {
    double d[kSarlAnyHumans];  // a parked (absent) human sorts behind every present one
    for (int j = 0; j < H; ++j)
        d[j] = is_parked(pos[g0 + 1 + j]) ? -1.0 - j : norm2(pos[g0 + 1 + j].x - pos[g0].x, pos[g0 + 1 + j].y - pos[g0].y);
    int who = p;
    for (int j = 0; j < H; ++j) {
        int rank = 0;
        for (int k = 0; k < H; ++k) rank += (d[k] > d[j] || (d[k] == d[j] && k < j)) ? 1 : 0;
        who = rank == p ? j : who;
    }
    return who;
}

// Next observable state of every human (get_next_observable_state, agent.py:63-74) from the ORCA velocities,
// and (with_om) the occupancy map each human would see among those next states.
__global__ void sarl_lookahead_kernel(SarlCfg C, const double2* pos, const double2* vel, const double2* rv,
                                      const float* orca_vel, double* next_obs /*[B][H][5]*/,
                                      float* om /*[B][H][cells*ch]*/) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= C.B * C.H) return;
    const int b = idx / C.H, i = idx - b * C.H;
    const int A = C.H + 1;
    const size_t g0 = (size_t)b * A;
    // query_env = true: the env's lookahead (ORCA velocities, env order).  query_env = false (multi_human_rl.py:39-40):
    // propagate(human_state, ActionXY(human_state.vx, human_state.vy)) — each human keeps its observed velocity — over
    // state.human_states, which LstmRL.predict has sorted by decreasing distance.
    auto next_of = [&](int j, double& px, double& py, double& vx, double& vy) {
        const int src = C.sort_lookahead ? human_by_decreasing_distance(pos, g0, C.H, j) : j;
        const size_t gj = g0 + 1 + src;
        if (C.const_vel) {
            vx = vel[gj].x, vy = vel[gj].y;
        } else {
            vx = orca_vel[2 * gj], vy = orca_vel[2 * gj + 1];
        }
        px = pos[gj].x + vx * C.dt, py = pos[gj].y + vy * C.dt;
    };
    double px, py, vx, vy;
    next_of(i, px, py, vx, vy);
    const int me = C.sort_lookahead ? human_by_decreasing_distance(pos, g0, C.H, i) : i;
    double* o = next_obs + (size_t)idx * 5;
    o[0] = px, o[1] = py, o[2] = vx, o[3] = vy, o[4] = rv[g0 + 1 + me].x;
    if (C.with_om) occupancy_map(C, i, next_of, om + (size_t)idx * C.cell_num * C.cell_num * C.om_channels);
}

// Reward of onestep_lookahead(action) for every (env, action) (crowd_sim.py:331-389, update = False).
// (a device function: sarl_select_kernel evaluates it where it combines reward and value — one kernel and one stream boundary
// less per decision, which is 10 % of a single-env decision)
__device__ inline double sarl_reward_of(const SarlCfg& C, const double2* pos, const double2* vel, const double2* goal,
                                        const double2* rv, const double* gtime, const double* theta,
                                        const double* actions /*[K][2]*/, int b, int a) {
    const int A = C.H + 1;
    const size_t g0 = (size_t)b * A;
    double ax = actions[2 * a], ay = actions[2 * a + 1];
    const double rot_v = ax, rot_r = ay;
    if (C.unicycle) {
        ax = rot_v * cos(rot_r + theta[b]);
        ay = rot_v * sin(rot_r + theta[b]);
    }
    const double2 rp = pos[g0];
    const double rrad = rv[g0].x;
    double dmin = __builtin_inf();
    bool collision = false;
    if (C.const_vel) {
        // MultiHumanRL.compute_reward(next_self_state, next_human_states) (multi_human_rl.py:65-88): end-point distances
        // only, its own hard-coded constants (-0.25, 1, 0.2, 0.5), no time limit
        double nx = rp.x + ax * C.dt, ny = rp.y + ay * C.dt;
        if (C.unicycle) {
            const double th = theta[b] + rot_r;
            nx = rp.x + rot_v * cos(th) * C.dt;  // cadrl.py:120-124: next_vx = v cos(next_theta); px + next_vx * dt
            ny = rp.y + rot_v * sin(th) * C.dt;
        }
        for (int i = 1; i < A; ++i) {
            const double2 hp = pos[g0 + i], hv = vel[g0 + i];
            const double hx = hp.x + hv.x * C.dt, hy = hp.y + hv.y * C.dt;
            const double dist = norm2(nx - hx, ny - hy) - rrad - rv[g0 + i].x;
            if (dist < 0.0) {
                collision = true;
                break;
            }
            if (dist < dmin) dmin = dist;
        }
        const double2 gl = goal[g0];
        const bool reaching = norm2(nx - gl.x, ny - gl.y) < rrad;
        double r;
        if (collision) {
            r = -0.25;
        } else if (reaching) {
            r = 1.0;
        } else if (dmin < 0.2) {
            r = (dmin - 0.2) * 0.5 * C.dt;
        } else {
            r = 0.0;
        }
        return r;
    }
    for (int i = 1; i < A; ++i) {
        const double2 hp = pos[g0 + i], hv = vel[g0 + i];
        const double x1 = hp.x - rp.x, y1 = hp.y - rp.y;
        const double wx = hv.x - ax, wy = hv.y - ay;
        const double x2 = x1 + wx * C.dt, y2 = y1 + wy * C.dt;
        const double sx = x2 - x1, sy = y2 - y1;
        double d;
        if (sx == 0.0 && sy == 0.0) {
            d = norm2(0.0 - x1, 0.0 - y1);
        } else {
            double u = ((0.0 - x1) * sx + (0.0 - y1) * sy) / (sx * sx + sy * sy);
            u = (u > 1.0) ? 1.0 : ((u < 0.0) ? 0.0 : u);
            d = norm2((x1 + u * sx) - 0.0, (y1 + u * sy) - 0.0);
        }
        const double c = d - rv[g0 + i].x - rrad;
        if (c < 0.0) {
            collision = true;
            break;
        } else if (c < dmin) {
            dmin = c;
        }
    }
    double endx = rp.x + ax * C.dt, endy = rp.y + ay * C.dt;
    if (C.unicycle) {
        const double th = theta[b] + rot_r;
        endx = rp.x + cos(th) * rot_v * C.dt;
        endy = rp.y + sin(th) * rot_v * C.dt;
    }
    const double2 gl = goal[g0];
    const bool reaching = norm2(endx - gl.x, endy - gl.y) < rrad;
    double r;
    if (gtime[b] >= C.time_limit - 1.0) {
        r = 0.0;
    } else if (collision) {
        r = C.collision_penalty;
    } else if (reaching) {
        r = C.success_reward;
    } else if (dmin < C.discomfort_dist) {
        r = (dmin - C.discomfort_dist) * C.discomfort_factor * C.dt;
    } else {
        r = 0.0;
    }
    return r;
}

// ------------------------------------------------------------------------------------ features
// CADRL.rotate (cadrl.py:187-222) of one float32 joint row [self (9) | human (5)]: one rounding per torch op.
__device__ __forceinline__ void rotate_row(float px, float py, float vx, float vy, float radius, float gx, float gy,
                                           float v_pref, float theta, int unicycle, float px1, float py1, float vx1,
                                           float vy1, float radius1, float* f) {
    const float dx = gx - px, dy = gy - py;
    const float rot = atan2f(dy, dx);
    const float dg = sqrtf(dx * dx + dy * dy);
    const float c = cosf(rot), s = sinf(rot);
    f[0] = dg;
    f[1] = v_pref;
    f[2] = unicycle ? theta - rot : 0.0f;  // cadrl.py:207-211
    f[3] = radius;
    f[4] = vx * c + vy * s;
    f[5] = vy * c - vx * s;
    f[6] = (px1 - px) * c + (py1 - py) * s;
    f[7] = (py1 - py) * c - (px1 - px) * s;
    f[8] = vx1 * c + vy1 * s;
    f[9] = vy1 * c - vx1 * s;
    f[10] = radius1;
    const float ex = px - px1, ey = py - py1;
    f[11] = sqrtf(ex * ex + ey * ey);
    f[12] = radius + radius1;
}

// The occupancy-map columns of X (features 13.. and the zero padding) for every (group, human) row: what
// sarl_feature_kernel leaves out with om_cols = 0.  Only cn_sarl_export uses it, so that an exported X is the whole input
// matrix of the value network whichever way the kernel read it.
__global__ void sarl_om_columns_kernel(SarlCfg C, int in_dim, int ks_x, const float* om, float* X, size_t n_tiles) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_tiles * C.H * kSarlGroups) return;
    const int g = (int)(idx % kSarlGroups);
    const int h = (int)((idx / kSarlGroups) % C.H);
    const size_t tile = idx / ((size_t)kSarlGroups * C.H);
    const size_t G = tile * kSarlGroups + g;
    if (G >= (size_t)C.B * C.n_actions) return;  // padding groups are zero already
    float* x = X + ((tile * C.H + h) * ks_x) * 64 + g;
    const int extra = in_dim - 13;
    const float* m = om + ((G / C.n_actions) * C.H + h) * (size_t)extra;
    for (int k = 0; k < extra; ++k) x[((13 + k) >> 2) * 64 + ((13 + k) & 3) * 16] = m[k];
    for (int n = in_dim; n < ks_x * 4; ++n) x[(n >> 2) * 64 + (n & 3) * 16] = 0.0f;
}

// The 13 rotated features of X row (env b, action a, human h): CADRL.rotate of the float32 joint row
// [propagate(self, action) (9) | next human state (5)] — shared by sarl_feature_kernel (X in HBM) and sarl_narrow_kernel (X
// built in LDS by the kernel that consumes it).
__device__ __forceinline__ void sarl_feature_row(const SarlCfg& C, int b, int a, int h, const double2* pos, const double2* goal,
                                                 const double2* rv, const double* theta, const double* actions,
                                                 double* next_obs, const double2* vel, const float* orca_vel, float* f) {
    const size_t g0 = (size_t)b * (C.H + 1);
    // propagate(self_state, action) in float64 (cadrl.py:113-118), then torch.Tensor([...]) narrows to float32
    double ax = actions[2 * a], ay = actions[2 * a + 1];
    float theta_f = 0.0f;
    if (C.unicycle) {  // cadrl.py:119-125: next_theta = theta + r, velocity v (cos, sin)(next_theta)
        const double th = theta[b] + ay, v = ax;
        ax = v * cos(th);
        ay = v * sin(th);
        theta_f = (float)th;
    }
    const float px = (float)(pos[g0].x + ax * C.dt), py = (float)(pos[g0].y + ay * C.dt);
    const float vx = (float)ax, vy = (float)ay;
    const float radius = (float)rv[g0].x, v_pref = (float)rv[g0].y;
    const float gx = (float)goal[g0].x, gy = (float)goal[g0].y;
    // the human's next observable state (get_next_observable_state, agent.py:63-74): from sarl_lookahead_kernel when it ran
    // (occupancy maps, LSTM-RL's re-ordering); otherwise computed here — the same float64 expressions — and written out by the
    // lanes of the env's first action (cn_sarl_export, the step's observation)
    double o[5];
    if (orca_vel != nullptr) {
        const size_t gj = g0 + 1 + h;
        const double vxn = C.const_vel ? vel[gj].x : (double)orca_vel[2 * gj], vyn = C.const_vel ? vel[gj].y : (double)orca_vel[2 * gj + 1];
        o[0] = pos[gj].x + vxn * C.dt, o[1] = pos[gj].y + vyn * C.dt, o[2] = vxn, o[3] = vyn, o[4] = rv[gj].x;
        if (a == 0) {
            double* dst = next_obs + ((size_t)b * C.H + h) * 5;
#pragma unroll
            for (int k = 0; k < 5; ++k) dst[k] = o[k];
        }
    } else {
        const double* src = next_obs + ((size_t)b * C.H + h) * 5;
#pragma unroll
        for (int k = 0; k < 5; ++k) o[k] = src[k];
    }
    const float px1 = (float)o[0], py1 = (float)o[1], vx1 = (float)o[2], vy1 = (float)o[3], radius1 = (float)o[4];
    rotate_row(px, py, vx, vy, radius, gx, gy, v_pref, theta_f, C.unicycle, px1, py1, vx1, vy1, radius1, f);
}

// X row of (env b, action a, human h): CADRL.rotate of the float32 joint row
// [propagate(self, action) (9) | next human state (5)] (+ the human's occupancy map), written straight in the MLP
// kernel's LDS order: group G = b * K + a -> tile G / 16, g = G % 16, row tile = h;
// X[((tile * H + h) * ks_x + n / 4) * 64 + (n % 4) * 16 + g] = feature n.  Lanes run over g fastest, so every
// store instruction writes 16 consecutive words per (tile, h).
__global__ void sarl_feature_kernel(SarlCfg C, int in_dim, int ks_x, const double2* pos, const double2* goal,
                                    const double2* rv, const double* theta, const double* actions,
                                    double* next_obs, const float* om, float* X, size_t n_tiles,
                                    int* hcount /*[n_tiles * 16] humans present per group*/,
                                    int om_cols = 1 /* 0: the consumer reads the occupancy maps from `om` itself (they do
                                    not depend on the action: written into X they are 81 copies, 48 of every 61 floats);
                                    only k-steps 0..3 — the 13 rotated features and map values 0..2 — are written */,
                                    const double2* vel = nullptr, const float* orca_vel = nullptr /* not null: no lookahead kernel ran */) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_tiles * C.H * kSarlGroups) return;
    const int g = (int)(idx % kSarlGroups);
    const int h = (int)((idx / kSarlGroups) % C.H);
    const size_t tile = idx / ((size_t)kSarlGroups * C.H);
    const size_t G = tile * kSarlGroups + g;
    float* x = X + ((tile * C.H + h) * ks_x) * 64 + g;
    if (G >= (size_t)C.B * C.n_actions) {  // padding groups of the last tile: finite zeros
        for (int n = 0; n < ks_x * 4; ++n) x[(n >> 2) * 64 + (n & 3) * 16] = 0.0f;
        if (h == 0) hcount[tile * kSarlGroups + g] = C.H;
        return;
    }
    if (h == 0) {  // len(state.human_states): under the `mixed` rule the env's absent humans are parked behind the present ones
        const size_t e0 = (G / C.n_actions) * (size_t)(C.H + 1);
        int present = 0;
        for (int j = 0; j < C.H; ++j) present += is_parked(pos[e0 + 1 + j]) ? 0 : 1;
        hcount[tile * kSarlGroups + g] = present;
    }
    const int a = (int)(G % C.n_actions);
    const int b = (int)(G / C.n_actions);
    float f[13];
    sarl_feature_row(C, b, a, h, pos, goal, rv, theta, actions, next_obs, vel, orca_vel, f);
#pragma unroll
    for (int k = 0; k < 13; ++k) x[(k >> 2) * 64 + (k & 3) * 16] = f[k];
    const int extra = in_dim - 13;
    const float* m = om + ((size_t)b * C.H + h) * (extra > 0 ? extra : 0);
    const int n_om = om_cols ? extra : (extra < 3 ? extra : 3);
    for (int k = 0; k < n_om; ++k) x[((13 + k) >> 2) * 64 + ((13 + k) & 3) * 16] = m[k];
    // (features in_dim .. 4 ks_x - 1 are never written by anyone: X is zero from its allocation — 7 of 20 floats per row at 13
    // inputs, and the kernel is bound by its HBM writes)
}

// ------------------------------------------------------------------------------------ replay-memory side
// MultiHumanRL.transform (multi_human_rl.py:90-104; CADRL.transform cadrl.py:171-185 when H = 1) of the CURRENT joint
// state of every env: rotate(float32 [self_state (9) | human h (5)]) (+ human h's occupancy map among the current
// human states) -> out[b][h][0..in_dim): the state a train-phase predict() leaves in policy.last_state and
// Explorer.update_memory pushes into the replay memory.  lane = (env, position in the joint state).
__device__ __forceinline__ void sarl_transform_row(const SarlCfg& C, int in_dim, int sort_humans, const double2* pos,
                                                   const double2* vel, const double2* goal, const double2* rv,
                                                   const double* theta, float* out, int64_t env_stride, int b, int h,
                                                   bool maps = true) {
    const size_t g0 = (size_t)b * (C.H + 1);
    // perm[p] = human at position p of the joint state.  LSTM-RL (sort_humans): LstmRL.predict re-orders the humans by
    // DEcreasing distance to the robot before MultiHumanRL.predict runs (lstm_rl.py:96-103; python's sorted(...,
    // reverse=True) is stable: equal distances keep their original order), so that is the order of last_state.
    int perm[kSarlMaxHumans];
#pragma unroll
    for (int p = 0; p < kSarlMaxHumans; ++p) perm[p] = p;
    if (sort_humans && C.H <= kSarlMaxHumans) {
        double d[kSarlMaxHumans];
#pragma unroll
        for (int j = 0; j < kSarlMaxHumans; ++j)
            d[j] = (j < C.H && !is_parked(pos[g0 + 1 + j])) ? norm2(pos[g0 + 1 + j].x - pos[g0].x, pos[g0 + 1 + j].y - pos[g0].y)
                                                            : -1.0 - j;
#pragma unroll
        for (int j = 0; j < kSarlMaxHumans; ++j) {
            int rank = 0;
#pragma unroll
            for (int k = 0; k < kSarlMaxHumans; ++k) rank += (k < C.H && (d[k] > d[j] || (d[k] == d[j] && k < j))) ? 1 : 0;
#pragma unroll
            for (int p = 0; p < kSarlMaxHumans; ++p) perm[p] = (j < C.H && rank == p) ? j : perm[p];
        }
    }
    const bool big = C.H > kSarlMaxHumans;  // the register permutation holds kSarlMaxHumans; larger crowds rank on demand
    int me = h;  // env order unless sorted
    if (sort_humans && !big) {
#pragma unroll
        for (int p = 0; p < kSarlMaxHumans; ++p) me = (p == h) ? perm[p] : me;
    } else if (sort_humans) {
        me = human_by_decreasing_distance(pos, g0, C.H, h);
    }
    const size_t g1 = g0 + 1 + me;
    float f[13];
    rotate_row((float)pos[g0].x, (float)pos[g0].y, (float)vel[g0].x, (float)vel[g0].y, (float)rv[g0].x,
               (float)goal[g0].x, (float)goal[g0].y, (float)rv[g0].y, (float)theta[b], C.unicycle, (float)pos[g1].x,
               (float)pos[g1].y, (float)vel[g1].x, (float)vel[g1].y, (float)rv[g1].x, f);
    float* x = out + (size_t)b * env_stride + (size_t)h * in_dim;
#pragma unroll
    for (int k = 0; k < 13; ++k) x[k] = f[k];
    if (C.with_om && maps) {  // (maps = false: the caller's lanes share them, occupancy_maps_cooperative)
        auto state_of = [&](int j, double& px, double& py, double& vx, double& vy) {
            int oj = j;
            if (!big) {
#pragma unroll
                for (int p = 0; p < kSarlMaxHumans; ++p) oj = (p == j) ? perm[p] : oj;
            } else if (sort_humans) {
                oj = human_by_decreasing_distance(pos, g0, C.H, j);
            }
            const size_t gj = g0 + 1 + oj;
            px = pos[gj].x, py = pos[gj].y, vx = vel[gj].x, vy = vel[gj].y;
        };
        occupancy_map(C, h, state_of, x + 13);
    }
}
__global__ void sarl_transform_kernel(SarlCfg C, int in_dim, int sort_humans, const double2* pos, const double2* vel,
                                      const double2* goal, const double2* rv, const double* theta,
                                      float* out /*[B][H][in_dim]*/, int64_t env_stride /*floats between envs*/) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= C.B * C.H) return;
    const int b = idx / C.H;
    sarl_transform_row(C, in_dim, sort_humans, pos, vel, goal, rv, theta, out, env_stride, b, idx - b * C.H);
}

// The epsilon-greedy branch of MultiHumanRL.predict (multi_human_rl.py:28-31), on each env's OWN numpy stream — the
// one cn_reset seeded (np.random.seed in CrowdSim.reset, crowd_sim.py:272-276), continued after the scenario draws:
//   probability = np.random.random();  if probability < epsilon: action_space[np.random.choice(K)]
// np.random.choice(K) of the legacy RandomState is randint(0, K): 32-bit draws masked to the next power of two minus
// one, rejected while > K - 1.  An env already at its goal (best == -1) returned before the draw (:22-23).
// lane = env.  explored (optional) receives 1 where the random action replaced the greedy one.
// the draw itself: returns the random action's index, or -1 when the greedy action stands (or the env has no stream)
__device__ __forceinline__ int sarl_explore_draw(int B, int K, double epsilon, uint32_t* mt_key, int* mt_pos, int* error, int b) {
    if (mt_pos[b] < 0) {  // the env was not (re)started by cn_reset: there is no stream to continue
        atomicOr(error, 2);
        return -1;
    }
    Mt19937 rng{mt_key + b, B, mt_pos[b]};
    const double probability = rng.random();
    int picked = -1;
    if (probability < epsilon) {
        uint32_t bits = (uint32_t)(K - 1);
        bits |= bits >> 1, bits |= bits >> 2, bits |= bits >> 4, bits |= bits >> 8, bits |= bits >> 16;
        uint32_t k = 0;
        if (K > 1) do k = rng.next32() & bits; while (k > (uint32_t)(K - 1));  // randint(0, 1) draws nothing
        picked = (int)k;
    }
    mt_pos[b] = rng.pos;
    return picked;
}
__device__ __forceinline__ void sarl_explore_env(int B, int K, double epsilon, uint32_t* mt_key, int* mt_pos,
                                                 const double* actions, bool masked_out, int32_t* best, double* action,
                                                 uint8_t* explored, int* error, int b) {
    if (explored) explored[b] = 0;
    if (masked_out) return;
    if (best[b] == -1) return;
    const int k = sarl_explore_draw(B, K, epsilon, mt_key, mt_pos, error, b);
    if (k >= 0) {
        best[b] = (int32_t)k;
        action[2 * b] = actions[2 * k];
        action[2 * b + 1] = actions[2 * k + 1];
        if (explored) explored[b] = 1;
    }
}
__global__ void sarl_explore_kernel(int B, int K, double epsilon, uint32_t* mt_key, int* mt_pos, const double* actions,
                                    const uint8_t* mask, int32_t* best, double* action, uint8_t* explored,
                                    int* error) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    sarl_explore_env(B, K, epsilon, mt_key, mt_pos, actions, mask && !mask[b], best, action, explored, error, b);
}

// ------------------------------------------------------------------------------------ value network
// Dense layer on MFMA: out[r][n] = act(bias[n] + extra[g][n] + sum_k in[r][k] W[n][k]) for r in [0, RT*16), buffers in
// fragment order (ks_in / ks_out k-steps per row tile).  A wave owns whole column tiles (ct = wave, wave + 16, ..)
// and all RT row tiles of them: per k-step it reads RT A fragments from LDS (conflict-free) and issues RT
// independent MFMAs; the B fragments of the next trip are already in flight from L2 (register double buffer).
// Workgroup barrier for data exchanged through LDS only: wait for this wave's LDS traffic, not for its global loads — so
// the B fragments of the NEXT layer, requested before the barrier (dense_prefetch), stay in flight across it
// (__syncthreads() drains vmcnt as well and would expose one L2 round trip per layer: 11 per tile).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Weights are read through GLOBAL-address-space pointers: a generic pointer whose provenance the compiler cannot see (the
// persistent kernel rebuilds them from an arena base) turns into flat_load, which counts on lgkmcnt as well as vmcnt — and
// lds_barrier's s_waitcnt lgkmcnt(0) would then wait for exactly the prefetch it is meant to leave in flight.
typedef const float __attribute__((address_space(1))) * gfloat_p;
__device__ __forceinline__ gfloat_p as_global(const float* p) { return (gfloat_p)p; }

// First trip of a wave's first column tile of a layer: kSarlKChunk B fragments + the bias, requested ahead of time.
struct BFrag {
    float b[kSarlKChunk];
    float bias;
};
__device__ __forceinline__ BFrag dense_prefetch(const PackedLinear& P, int wave, int lane) {
    BFrag f;
    const int ct = wave < P.ctiles ? wave : 0;  // waves without a column tile fetch tile 0 (no divergence, L2 hit)
    const gfloat_p wfrag = as_global(P.w) + (size_t)ct * P.kpad * 64 + lane;
#pragma unroll
    for (int j = 0; j < kSarlKChunk; ++j) f.b[j] = wfrag[j * 64];
    f.bias = as_global(P.bias)[ct * 16 + (lane & 15)];
    return f;
}

// wave0 / nwaves: the waves [wave0, wave0 + nwaves) share the layer's column tiles (default: the whole workgroup); the others
// return at once — the pipelined kernel runs two layers of different tiles side by side on disjoint wave ranges.
template <int RT, bool PRE = false>
__device__ __forceinline__ void dense_mfma(const PackedLinear& P, const float* in, int ks_in, float* out, int ks_out,
                                           bool relu, const float* extra, int wave, int lane, const BFrag* pre = nullptr,
                                           int wave0 = 0, int nwaves = kSarlThreads / 64) {
    const int col = lane & 15, quad = lane >> 4;
    if (wave < wave0 || wave >= wave0 + nwaves) return;
    for (int ct = wave - wave0; ct < P.ctiles; ct += nwaves) {
        // this lane's 4 accumulator rows of column n = ct*16 + col sit at 4 consecutive words of the out buffer
        const int frag_off = ((ct * 4 + (col >> 2)) * 64) + (col & 3) * 16 + quad * 4;
        // accumulators start at zero; bias (L2) and the per-group extra term (LDS) are requested now and added in the
        // epilogue, so their latency hides behind the k loop instead of opening the column tile
        f32x4 acc[RT];
        const bool first = PRE && ct == wave - wave0;  // this tile's first trip was prefetched before the barrier
        const float b0 = first ? pre->bias : as_global(P.bias)[ct * 16 + col];
        f32x4 addend = {b0, b0, b0, b0};
        if (extra) addend += *reinterpret_cast<const f32x4*>(extra + frag_off);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

        // Straight-line k loop, kSarlKChunk k-steps per trip, no conditionals (kpad <= ks_in by construction: every buffer
        // holds whole column tiles of its producer, zero beyond the true width, and the packed weights are zero
        // there too).  The B fragments of the next trip are requested before this trip's MFMAs issue; the
        // packed buffer carries one spare chunk so the last request stays in bounds.
        const gfloat_p wfrag = as_global(P.w) + (size_t)ct * P.kpad * 64 + lane;
        const float* afrag = in + lane;
        // One trip = kSarlKChunk k-steps x RT row tiles of MFMAs.  Its B fragments were requested from L2 a whole trip
        // earlier (bnxt; measured: a second trip of lead changes nothing).  Its A fragments come from LDS in two groups so
        // that no trip opens with a wait: the HEAD (first kHead k-steps) was requested during the previous trip, the REST
        // is requested now and arrives while the head's MFMAs run; then the next trip's head is requested while the
        // rest's MFMAs run.  (A whole second register set for A costs 25 VGPRs more and spills at 16 waves per CU.)
        // The head request after the last trip reads past the row tile into whatever follows it in LDS (always inside
        // the allocation: every MFMA input buffer is followed by another buffer) and is unused.
        constexpr int kHead = 2;
        float bcur[kSarlKChunk], bnxt[kSarlKChunk];
        float ah[kHead][RT], ar[kSarlKChunk - kHead][RT];
#pragma unroll
        for (int j = 0; j < kSarlKChunk; ++j) bcur[j] = first ? pre->b[j] : wfrag[j * 64];
#pragma unroll
        for (int j = 0; j < kHead; ++j)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) ah[j][rt] = afrag[(rt * ks_in + j) * 64];
        for (int k0 = 0; k0 < P.kpad; k0 += kSarlKChunk) {
#pragma unroll
            for (int j = 0; j < kSarlKChunk; ++j) bnxt[j] = wfrag[(k0 + kSarlKChunk + j) * 64];
#pragma unroll
            for (int j = kHead; j < kSarlKChunk; ++j)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) ar[j - kHead][rt] = afrag[(rt * ks_in + k0 + j) * 64];
            __builtin_amdgcn_sched_barrier(0);  // requests first
#pragma unroll
            for (int j = 0; j < kHead; ++j)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[j][rt], bcur[j], acc[rt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < kHead; ++j)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) ah[j][rt] = afrag[(rt * ks_in + k0 + kSarlKChunk + j) * 64];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = kHead; j < kSarlKChunk; ++j)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[j - kHead][rt], bcur[j], acc[rt], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < kSarlKChunk; ++j) bcur[j] = bnxt[j];
            __builtin_amdgcn_sched_barrier(0);
        }
        // columns >= N of the last column tile are exact zeros (zero weights, zero bias): they are the consumer's
        // k padding, so they are stored too
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            f32x4 v = acc[rt] + addend;
            if (relu) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = v[i] > 0.0f ? v[i] : 0.0f;
            }
            *reinterpret_cast<f32x4*>(out + rt * ks_out * 64 + frag_off) = v;
        }
    }
}

// Whole sarl.ValueNetwork.forward for one tile of 16 groups x H humans; X in fragment order, V[group] out.
// A layer with ONE output (attention.4, mlp3.6) as a dot product on the vector ALUs: as an MFMA it would occupy one
// wave of one SIMD with 15 of 16 columns wasted (7 % of the tile time at 100 -> 1 over 5 row tiles).  Thread = (row,
// k slice); partial sums meet in `scratch` (kSarlThreads floats).  Contains one workgroup barrier.
template <int RT>
__device__ __forceinline__ void dense_vec1(const PackedLinear& P, const float* in, int ks_in, float* out, int ks_out,
                                           float* scratch, int tid) {
    constexpr int kRows = RT * 16, kSlices = kSarlThreads / kRows;
    const int row = tid % kRows, slice = tid / kRows;
    if (slice < kSlices) {
        const float* a = in + ((row >> 4) * ks_in) * 64 + (row & 15);
        float sum = 0.0f;
        for (int s = slice; s < P.ksteps; s += kSlices) {  // fragment (0, s): w[s * 64 + 16 j] = W[0][4 s + j]
            const gfloat_p w = as_global(P.w) + s * 64;
            const float* x = a + s * 64;
            sum += (x[0] * w[0] + x[16] * w[16]) + (x[32] * w[32] + x[48] * w[48]);
        }
        scratch[slice * kRows + row] = sum;
    }
    __syncthreads();
    if (tid < kRows) {
        float v = as_global(P.bias)[0];
#pragma unroll
        for (int s = 0; s < kSlices; ++s) v += scratch[s * kRows + tid];
        out[(tid >> 4) * ks_out * 64 + (tid & 15)] = v;
    }
}

__device__ __forceinline__ void zero_lds(float* lds, size_t words, int tid) {
    for (size_t i = tid; i < words; i += kSarlThreads) lds[i] = 0.0f;
}

// Persistent form: a workgroup (one per CU: the tile's activations fill the LDS) strides over the tiles; LDS is zeroed once
// per launch instead of once per tile (every k-padding word the MFMA loops read is either written by the producing layer —
// whole column tiles — or zeroed explicitly below); layers are separated by lds_barrier, and every layer's first B
// fragments are requested BEFORE the barrier that precedes it (dense_prefetch), while the previous layer's epilogue and the
// barrier wait are still in progress.  (The plain form of this kernel — every wave on every layer, the value head in line:
// what profiles/r02_sarl_ablation.txt measured — and the one-tile-per-workgroup kernel of round 1 were removed in round 4;
// no default route reached them.)
// The value head of tile t - 1 (mlp3: three 16-row layers + the single-output layer, 13 k of a tile's 84 k ticks when run
// on its own: 16 rows cannot fill the workgroup) runs on the waves that idle during tile t's 7-column-tile layers:
//   slot of tile t            main waves 0..6 (0..9)      side waves
//   mlp1.2                    h2                          7..15: mlp3.0 (t - 1)   jbuf -> mbuf
//   mean + mlp2.0                                         7..13: mlp3.2 (t - 1)   mbuf -> jbuf
//   mlp2.2 + att0 global      features, global term       7..13: mlp3.4 (t - 1)   jbuf -> mbuf
//   att0 local                                            15:    mlp3.6 (t - 1)   mbuf -> V   (one wave, shuffles)
// The side chain has its own pong buffer (mbuf: kbuf carries tile t's global attention term in the same slots) and the
// joint state of tile t is written — self features from registers, weighted sum, zero padding — only in tile t's last slot,
// after the side chain has consumed the previous one.  The last tile's head runs after the loop on all waves.
__device__ __forceinline__ void value_head_on_one_wave(const PackedLinear& P, const float* in, float* V, size_t tile,
                                                       int n_groups, int lane, int groups_per_tile = kSarlGroups) {
    const int row = lane & 15, slice = lane >> 4;  // 4 k slices of the 16 rows
    float sum = 0.0f;
    for (int s = slice; s < P.ksteps; s += 4) {
        const gfloat_p w = as_global(P.w) + s * 64;
        const float* x = in + s * 64 + row;
        sum += (x[0] * w[0] + x[16] * w[16]) + (x[32] * w[32] + x[48] * w[48]);
    }
    float v = as_global(P.bias)[0];
#pragma unroll
    for (int j = 0; j < 4; ++j) v += __shfl(sum, row + 16 * j);
    const size_t G = tile * groups_per_tile + row;
    if (slice == 0 && row < groups_per_tile && G < (size_t)n_groups) V[G] = v;
}

template <int H>
__global__ __launch_bounds__(kSarlThreads) void sarl_mlp_pipe_kernel(SarlNetRef net, const float* X, float* V, int n_groups,
                                                                     int n_tiles, const int* hcount) {
    extern __shared__ float lds[];
    float* bufA = lds;                            // [H][ks_a][64]  wide hidden layers
    float* bufB = bufA + H * net.ks_a * 64;       // [H][ks_b][64]  X staging, then mlp1 output (h2), then attention.2
    float* bufC = bufB + H * net.ks_b * 64;       // [H][ks_c][64]  mlp2 output (per-human feature)
    float* gbuf = bufC + H * net.ks_c * 64;       // [ks_b][64]     mean over humans of h2
    float* jbuf = gbuf + net.ks_b * 64;           // [ks_a][64]     joint state / value-head ping
    float* kbuf = jbuf + net.ks_a * 64;           // [ks_a][64]     global attention term
    float* sbuf = kbuf + net.ks_a * 64;           // [H][ks_s][64]  attention scores -> weights
    float* vbuf = sbuf + H * net.ks_s * 64;       // [kSarlThreads] partial sums of attention.4
    int* hc = reinterpret_cast<int*>(vbuf + kSarlThreads);  // [16]
    float* mbuf = reinterpret_cast<float*>(hc + 16);        // [ks_a][64]     value-head pong (side chain)

    int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    zero_lds(lds, (size_t)(mbuf + net.ks_a * 64 - lds), tid);
    const int nf = net.nf;
    const int x_words = H * net.ks_x * 64;
    float* xs = bufB;
    BFrag pre = dense_prefetch(layer_of(net, kL_mlp1_0), wave, lane);
    lds_barrier();
    int prev_tile = -1;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        asm volatile("" : "+v"(tid), "+v"(lane));
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        SarlNetRef nn = net;
        asm volatile("" : "+s"(nn.base));
        const SarlNetRef* n = &nn;
        const bool side = prev_tile >= 0;  // a previous tile's value head is pending
        const float* xg = X + (size_t)tile * x_words;
        for (int i = tid; i < x_words; i += kSarlThreads) xs[i] = xg[i];
        if (tid < kSarlGroups) hc[tid] = hcount[(size_t)tile * kSarlGroups + tid];
        __syncthreads();  // X came from global memory
        // self_state = state[:, 0, :6] (sarl.py:36): kept in a register until the joint state of this tile is assembled
        float self_val = 0.0f;
        if (tid < kSarlGroups * 6) {
            const int g = tid & 15, f = tid >> 4;
            self_val = xs[(f >> 2) * 64 + (f & 3) * 16 + g];
        }
        dense_mfma<H, true>(layer_of(*n, kL_mlp1_0), xs, n->ks_x, bufA, n->ks_a, true, nullptr, wave, lane, &pre);
        pre = dense_prefetch(layer_of(*n, kL_mlp1_2), wave, lane);
        lds_barrier();
        dense_mfma<H, true>(layer_of(*n, kL_mlp1_2), bufA, n->ks_a, bufB, n->ks_b, true, nullptr, wave, lane, &pre);  // h2
        if (side) dense_mfma<1>(layer_of(*n, kL_mlp3_0), jbuf, n->ks_a, mbuf, n->ks_a, true, nullptr, wave, lane, nullptr, 7, 9);
        pre = dense_prefetch(layer_of(*n, kL_mlp2_0), wave, lane);
        lds_barrier();
        if (n->with_global) {
            for (int i = tid; i < n->ks_b * 64; i += kSarlThreads) {
                const int cnt = hc[i & 15];
                float sum = 0.0f;
#pragma unroll
                for (int h = 0; h < H; ++h) sum += h < cnt ? bufB[h * n->ks_b * 64 + i] : 0.0f;
                gbuf[i] = sum / (float)cnt;
            }
        }
        dense_mfma<H, true>(layer_of(*n, kL_mlp2_0), bufB, n->ks_b, bufA, n->ks_a, true, nullptr, wave, lane, &pre);
        if (side) dense_mfma<1>(layer_of(*n, kL_mlp3_2), mbuf, n->ks_a, jbuf, n->ks_a, true, nullptr, wave, lane, nullptr, 7, 7);
        pre = dense_prefetch(layer_of(*n, kL_mlp2_2), wave, lane);
        lds_barrier();
        dense_mfma<H, true>(layer_of(*n, kL_mlp2_2), bufA, n->ks_a, bufC, n->ks_c, false, nullptr, wave, lane, &pre);  // features
        if (n->with_global) dense_mfma<1>(layer_of(*n, kL_att0_global), gbuf, n->ks_b, kbuf, n->ks_a, false, nullptr, wave, lane);
        if (side) dense_mfma<1>(layer_of(*n, kL_mlp3_4), jbuf, n->ks_a, mbuf, n->ks_a, true, nullptr, wave, lane, nullptr, 7, 7);
        pre = dense_prefetch(layer_of(*n, kL_att0_local), wave, lane);
        lds_barrier();
        dense_mfma<H, true>(layer_of(*n, kL_att0_local), bufB, n->ks_b, bufA, n->ks_a, true, n->with_global ? kbuf : nullptr,
                            wave, lane, &pre);
        if (side && wave == 15) value_head_on_one_wave(layer_of(*n, kL_mlp3_6), mbuf, V, (size_t)prev_tile, n_groups, lane);
        pre = dense_prefetch(layer_of(*n, kL_att_2), wave, lane);
        lds_barrier();
        dense_mfma<H, true>(layer_of(*n, kL_att_2), bufA, n->ks_a, bufB, n->ks_b, true, nullptr, wave, lane, &pre);
        lds_barrier();
        dense_vec1<H>(layer_of(*n, kL_att_4), bufB, n->ks_b, sbuf, n->ks_s, vbuf, tid);  // score (h, g) at h*ks_s*64 + g
        pre = dense_prefetch(layer_of(*n, kL_mlp1_0), wave, lane);  // the next tile's first layer
        lds_barrier();
        // masked softmax without max subtraction (sarl.py:52-53)
        if (tid < kSarlGroups) {
            float e[H], total = 0.0f;
            const int cnt = hc[tid];
#pragma unroll
            for (int h = 0; h < H; ++h) {
                const float sc = sbuf[h * n->ks_s * 64 + tid];
                e[h] = h < cnt ? expf(sc) * (sc != 0.0f ? 1.0f : 0.0f) : 0.0f;
                total += e[h];
            }
#pragma unroll
            for (int h = 0; h < H; ++h) sbuf[h * n->ks_s * 64 + tid] = e[h] / total;
        }
        lds_barrier();
        // the joint state of this tile: self features, weighted feature sum (sarl.py:60), zero k padding
        if (tid < kSarlGroups * 6) {
            const int g = tid & 15, f = tid >> 4;
            jbuf[(f >> 2) * 64 + (f & 3) * 16 + g] = self_val;
        }
        for (int i = tid; i < kSarlGroups * nf; i += kSarlThreads) {
            const int g = i & 15, c = i >> 4;
            const int src = (c >> 2) * 64 + (c & 3) * 16 + g;
            float sum = 0.0f;
#pragma unroll
            for (int h = 0; h < H; ++h) sum += sbuf[h * n->ks_s * 64 + g] * bufC[h * n->ks_c * 64 + src];
            const int f = 6 + c;
            jbuf[(f >> 2) * 64 + (f & 3) * 16 + g] = sum;
        }
        for (int i = tid; i < kSarlGroups * (layer_of(*n, kL_mlp3_0).kpad * 4 - 6 - nf); i += kSarlThreads) {
            const int g = i & 15, f = 6 + nf + (i >> 4);
            jbuf[(f >> 2) * 64 + (f & 3) * 16 + g] = 0.0f;
        }
        lds_barrier();
        prev_tile = tile;
    }
    if (prev_tile >= 0) {  // the last tile's value head, on the whole workgroup
        dense_mfma<1>(layer_of(net, kL_mlp3_0), jbuf, net.ks_a, mbuf, net.ks_a, true, nullptr, wave, lane);
        lds_barrier();
        dense_mfma<1>(layer_of(net, kL_mlp3_2), mbuf, net.ks_a, jbuf, net.ks_a, true, nullptr, wave, lane);
        lds_barrier();
        dense_mfma<1>(layer_of(net, kL_mlp3_4), jbuf, net.ks_a, mbuf, net.ks_a, true, nullptr, wave, lane);
        lds_barrier();
        if (wave == 15) value_head_on_one_wave(layer_of(net, kL_mlp3_6), mbuf, V, (size_t)prev_tile, n_groups, lane);
    }
}
__host__ inline size_t sarl_mlp_pipe_extra_lds_bytes(const SarlNet& net) { return sizeof(float) * 64 * (size_t)net.ks_a; }


// cadrl.ValueNetwork (cadrl.py:22-29): the same MLP for every (robot, human) row, then the minimum over the humans
// of a group (cadrl.py:162-163).  Layers live in L[kL_mlp3_0 .. kL_mlp3_6]; buffers as in the SARL kernel.
// sarl.ValueNetwork for MORE humans than one tile's LDS holds (H > kSarlMaxHumans; e.g. the 20-human crowds of
// BASELINE configs[3]): the humans of a tile's 16 groups stream through in chunks of HC row tiles.
//   pass 1  mlp1 of every chunk, summed over the humans -> the global state (mean) and its attention term
//   pass 2  mlp1 again (cheaper than parking [H][112] floats per group row in LDS), mlp2, attention; exp(score) and
//           exp(score) * feature accumulate per group in human order, the division by the total comes last
//           (the reference divides first: w_h = e_h / total, then sums w_h f_h — same value to rounding)
// then the value head.  Rows of a partial last chunk are computed on zero inputs and masked out of every sum.
template <int HC>
__global__ __launch_bounds__(kSarlThreads) void sarl_mlp_chunked_kernel(SarlNet net, const float* X, float* V,
                                                                        int n_groups) {
    extern __shared__ float lds[];
    const int H = net.H;
    float* bufA = lds;                            // [HC][ks_a][64]
    float* bufB = bufA + HC * net.ks_a * 64;      // [HC][ks_b][64]
    float* bufC = bufB + HC * net.ks_b * 64;      // [HC][ks_c][64]
    float* gbuf = bufC + HC * net.ks_c * 64;      // [ks_b][64]  sum, then mean, over humans of h2
    float* jbuf = gbuf + net.ks_b * 64;           // [ks_a][64]
    float* kbuf = jbuf + net.ks_a * 64;           // [ks_a][64]
    float* sbuf = kbuf + net.ks_a * 64;           // [HC][ks_s][64]
    float* wsum = sbuf + HC * net.ks_s * 64;      // [ks_c][64]  sum_h exp(score_h) * feature_h
    float* den = wsum + net.ks_c * 64;            // [64]        sum_h exp(score_h) (16 groups used)
    float* vbuf = den + 64;                       // [kSarlThreads]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const size_t tile = blockIdx.x;
    const float* xg = X + tile * H * net.ks_x * 64;
    zero_lds(lds, (size_t)(vbuf - lds), tid);
    __syncthreads();
    if (tid < kSarlGroups * 6) {  // self_state = state[:, 0, :6]
        const int g = tid & 15, n = tid >> 4;
        jbuf[(n >> 2) * 64 + (n & 3) * 16 + g] = xg[(n >> 2) * 64 + (n & 3) * 16 + g];
    }
    auto stage = [&](int h0, int nh) {  // X rows of humans [h0, h0 + nh) -> bufB, zeros beyond
        for (int i = tid; i < HC * net.ks_x * 64; i += kSarlThreads)
            bufB[i] = (i / (net.ks_x * 64) < nh) ? xg[(size_t)h0 * net.ks_x * 64 + i] : 0.0f;
    };
    for (int h0 = 0; h0 < H; h0 += HC) {
        const int nh = H - h0 < HC ? H - h0 : HC;
        stage(h0, nh);
        __syncthreads();
        dense_mfma<HC>(net.L[kL_mlp1_0], bufB, net.ks_x, bufA, net.ks_a, true, nullptr, wave, lane);
        __syncthreads();
        dense_mfma<HC>(net.L[kL_mlp1_2], bufA, net.ks_a, bufB, net.ks_b, true, nullptr, wave, lane);
        __syncthreads();
        for (int i = tid; i < net.ks_b * 64; i += kSarlThreads) {
            float sum = gbuf[i];
            for (int rt = 0; rt < nh; ++rt) sum += bufB[rt * net.ks_b * 64 + i];
            gbuf[i] = sum;
        }
        __syncthreads();
    }
    if (net.with_global) {
        for (int i = tid; i < net.ks_b * 64; i += kSarlThreads) gbuf[i] = gbuf[i] / (float)H;
        __syncthreads();
        dense_mfma<1>(net.L[kL_att0_global], gbuf, net.ks_b, kbuf, net.ks_a, false, nullptr, wave, lane);
        __syncthreads();
    }
    const int nf = net.L[kL_mlp2_2].N;
    for (int h0 = 0; h0 < H; h0 += HC) {
        const int nh = H - h0 < HC ? H - h0 : HC;
        stage(h0, nh);
        __syncthreads();
        dense_mfma<HC>(net.L[kL_mlp1_0], bufB, net.ks_x, bufA, net.ks_a, true, nullptr, wave, lane);
        __syncthreads();
        dense_mfma<HC>(net.L[kL_mlp1_2], bufA, net.ks_a, bufB, net.ks_b, true, nullptr, wave, lane);
        __syncthreads();
        dense_mfma<HC>(net.L[kL_mlp2_0], bufB, net.ks_b, bufA, net.ks_a, true, nullptr, wave, lane);
        __syncthreads();
        dense_mfma<HC>(net.L[kL_mlp2_2], bufA, net.ks_a, bufC, net.ks_c, false, nullptr, wave, lane);
        __syncthreads();
        dense_mfma<HC>(net.L[kL_att0_local], bufB, net.ks_b, bufA, net.ks_a, true, net.with_global ? kbuf : nullptr,
                       wave, lane);
        __syncthreads();
        dense_mfma<HC>(net.L[kL_att_2], bufA, net.ks_a, bufB, net.ks_b, true, nullptr, wave, lane);
        __syncthreads();
        dense_vec1<HC>(net.L[kL_att_4], bufB, net.ks_b, sbuf, net.ks_s, vbuf, tid);
        __syncthreads();
        if (tid < kSarlGroups) {  // masked exp without max subtraction (sarl.py:52-53), humans in order
            float total = den[tid];
            for (int rt = 0; rt < nh; ++rt) {
                const float sc = sbuf[rt * net.ks_s * 64 + tid];
                const float e = expf(sc) * (sc != 0.0f ? 1.0f : 0.0f);
                sbuf[rt * net.ks_s * 64 + tid] = e;
                total += e;
            }
            den[tid] = total;
        }
        __syncthreads();
        for (int i = tid; i < kSarlGroups * nf; i += kSarlThreads) {
            const int g = i & 15, c = i >> 4;
            const int src = (c >> 2) * 64 + (c & 3) * 16 + g;
            float sum = wsum[src];
            for (int rt = 0; rt < nh; ++rt) sum += sbuf[rt * net.ks_s * 64 + g] * bufC[rt * net.ks_c * 64 + src];
            wsum[src] = sum;
        }
        __syncthreads();
    }
    for (int i = tid; i < kSarlGroups * nf; i += kSarlThreads) {
        const int g = i & 15, c = i >> 4, n = 6 + c;
        jbuf[(n >> 2) * 64 + (n & 3) * 16 + g] = wsum[(c >> 2) * 64 + (c & 3) * 16 + g] / den[g];
    }
    __syncthreads();
    dense_mfma<1>(net.L[kL_mlp3_0], jbuf, net.ks_a, kbuf, net.ks_a, true, nullptr, wave, lane);
    __syncthreads();
    dense_mfma<1>(net.L[kL_mlp3_2], kbuf, net.ks_a, jbuf, net.ks_a, true, nullptr, wave, lane);
    __syncthreads();
    dense_mfma<1>(net.L[kL_mlp3_4], jbuf, net.ks_a, kbuf, net.ks_a, true, nullptr, wave, lane);
    __syncthreads();
    dense_vec1<1>(net.L[kL_mlp3_6], kbuf, net.ks_a, sbuf, net.ks_s, vbuf, tid);
    __syncthreads();
    if (tid < kSarlGroups) {
        const size_t G = tile * kSarlGroups + tid;
        if (G < (size_t)n_groups) V[G] = sbuf[tid];
    }
}

constexpr int kSarlChunk = 5;  // row tiles per chunk of sarl_mlp_chunked_kernel
__host__ inline size_t sarl_mlp_chunked_lds_bytes(const SarlNet& net) {
    return sizeof(float) * (64 * ((size_t)kSarlChunk * (net.ks_a + net.ks_b + net.ks_c + net.ks_s) + net.ks_b +
                                  2 * net.ks_a + net.ks_c + 1) + kSarlThreads);
}

template <int H>
__global__ __launch_bounds__(kSarlThreads) void cadrl_mlp_kernel(SarlNet net, const float* X, float* V, int n_groups,
                                                                 const int* hcount) {
    extern __shared__ float lds[];
    float* bufA = lds;
    float* bufB = bufA + H * net.ks_a * 64;
    float* sbuf = bufB + H * net.ks_b * 64;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const size_t tile = blockIdx.x;
    zero_lds(lds, (size_t)H * (net.ks_a + net.ks_b + net.ks_s) * 64, tid);
    __syncthreads();
    const float* xg = X + tile * H * net.ks_x * 64;
    for (int i = tid; i < H * net.ks_x * 64; i += kSarlThreads) bufB[i] = xg[i];
    __syncthreads();
    dense_mfma<H>(net.L[kL_mlp3_0], bufB, net.ks_x, bufA, net.ks_a, true, nullptr, wave, lane);
    __syncthreads();
    dense_mfma<H>(net.L[kL_mlp3_2], bufA, net.ks_a, bufB, net.ks_b, true, nullptr, wave, lane);
    __syncthreads();
    dense_mfma<H>(net.L[kL_mlp3_4], bufB, net.ks_b, bufA, net.ks_a, true, nullptr, wave, lane);
    __syncthreads();
    dense_mfma<H>(net.L[kL_mlp3_6], bufA, net.ks_a, sbuf, net.ks_s, false, nullptr, wave, lane);
    __syncthreads();
    if (tid < kSarlGroups) {
        const int cnt = hcount[tile * kSarlGroups + tid];  // humans present (H unless the `mixed` rule)
        float m = sbuf[tid];
#pragma unroll
        for (int h = 1; h < H; ++h) {
            const float v = sbuf[h * net.ks_s * 64 + tid];
            m = (h < cnt && v < m) ? v : m;  // torch.min over dim 0: the first minimum's value
        }
        const size_t G = tile * kSarlGroups + tid;
        if (G < (size_t)n_groups) V[G] = m;
    }
}

// cadrl.ValueNetwork for MORE humans than the one-tile kernel holds (H > kSarlMaxHumans): the humans of a tile's 16
// groups stream through in chunks of HC row tiles, the per-group minimum (cadrl.py:162-163) accumulates in LDS.  Rows of a
// partial last chunk run on zero inputs and are left out of the minimum.
template <int HC>
__global__ __launch_bounds__(kSarlThreads) void cadrl_mlp_chunked_kernel(SarlNet net, const float* X, float* V,
                                                                         int n_groups) {
    extern __shared__ float lds[];
    const int H = net.H;
    float* bufA = lds;                             // [HC][ks_a][64]
    float* bufB = bufA + HC * net.ks_a * 64;       // [HC][ks_b][64]
    float* sbuf = bufB + HC * net.ks_b * 64;       // [HC][ks_s][64]
    float* vmin = sbuf + HC * net.ks_s * 64;       // [16]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const size_t tile = blockIdx.x;
    const float* xg = X + tile * H * net.ks_x * 64;
    zero_lds(lds, (size_t)HC * (net.ks_a + net.ks_b + net.ks_s) * 64 + 16, tid);
    __syncthreads();
    for (int h0 = 0; h0 < H; h0 += HC) {
        const int nh = H - h0 < HC ? H - h0 : HC;
        for (int i = tid; i < HC * net.ks_x * 64; i += kSarlThreads)
            bufB[i] = (i / (net.ks_x * 64) < nh) ? xg[(size_t)h0 * net.ks_x * 64 + i] : 0.0f;
        __syncthreads();
        dense_mfma<HC>(net.L[kL_mlp3_0], bufB, net.ks_x, bufA, net.ks_a, true, nullptr, wave, lane);
        __syncthreads();
        dense_mfma<HC>(net.L[kL_mlp3_2], bufA, net.ks_a, bufB, net.ks_b, true, nullptr, wave, lane);
        __syncthreads();
        dense_mfma<HC>(net.L[kL_mlp3_4], bufB, net.ks_b, bufA, net.ks_a, true, nullptr, wave, lane);
        __syncthreads();
        dense_mfma<HC>(net.L[kL_mlp3_6], bufA, net.ks_a, sbuf, net.ks_s, false, nullptr, wave, lane);
        __syncthreads();
        if (tid < kSarlGroups) {
            float m = h0 == 0 ? sbuf[tid] : vmin[tid];
            for (int rt = (h0 == 0 ? 1 : 0); rt < nh; ++rt) {
                const float v = sbuf[rt * net.ks_s * 64 + tid];
                m = v < m ? v : m;  // torch.min over dim 0: the first minimum's value
            }
            vmin[tid] = m;
        }
        __syncthreads();
    }
    if (tid < kSarlGroups) {
        const size_t G = tile * kSarlGroups + tid;
        if (G < (size_t)n_groups) V[G] = vmin[tid];
    }
}
__host__ inline size_t cadrl_mlp_chunked_lds_bytes(const SarlNet& net) {
    return sizeof(float) * ((size_t)kSarlChunk5 * (net.ks_a + net.ks_b + net.ks_s) * 64 + 16);
}

// lstm_rl.ValueNetwork1 / ValueNetwork2 (lstm_rl.py:9-66): an LSTM over the humans of a group (in the order the lookahead returns
// them), its final hidden state joined with the robot's 6 self features into the value head.  Layers: L[kL_mlp1_0] =
// weight_ih / bias_ih, L[kL_mlp1_2] = weight_hh / bias_hh (torch gate order i, f, g, o), L[kL_mlp3_*] = the head.
// Row tile t of X is human t of the 16 groups = LSTM time step t, so each step is a 16-row product.
template <int H>
__global__ __launch_bounds__(kSarlThreads) void lstm_mlp_kernel(SarlNet net, const float* X, float* V, int n_groups,
                                                                const int* hcount) {
    extern __shared__ float lds[];
    const int hid = net.L[kL_mlp1_2].K;                   // hidden width (50)
    const int ks_h = sarl_ks(hid), ks_g = net.L[kL_mlp1_0].ctiles * 4;
    float* xs = lds;                                       // [H][ks_x][64]
    float* pbuf = xs + H * net.ks_x * 64;                  // [H][ks_b][64] ValueNetwork2.mlp1 ping (ks_b = 0 otherwise)
    float* qbuf = pbuf + H * net.ks_b * 64;                // [H][ks_c][64] ... pong: the LSTM input when pairwise
    float* gates = qbuf + H * net.ks_c * 64;               // [ks_g][64]   i | f | g | o pre-activations
    float* hbuf = gates + ks_g * 64;                       // [ks_h][64]   hidden state (A operand of the next step)
    float* cbuf = hbuf + ks_h * 64;                        // [hid][16]    cell state
    float* jbuf = cbuf + hid * kSarlGroups;                // [ks_a][64]
    float* kbuf = jbuf + net.ks_a * 64;                    // [ks_a][64]
    float* sbuf = kbuf + net.ks_a * 64;                    // [ks_s][64]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const size_t tile = blockIdx.x;
    zero_lds(lds, (size_t)((sbuf + net.ks_s * 64) - lds), tid);
    __syncthreads();
    const float* xg = X + tile * H * net.ks_x * 64;
    for (int i = tid; i < H * net.ks_x * 64; i += kSarlThreads) xs[i] = xg[i];
    for (int i = tid; i < ks_h * 64; i += kSarlThreads) hbuf[i] = 0.0f;   // h0 = 0
    for (int i = tid; i < hid * kSarlGroups; i += kSarlThreads) cbuf[i] = 0.0f;  // c0 = 0
    for (int i = tid; i < net.ks_a * 64; i += kSarlThreads) jbuf[i] = 0.0f;
    __syncthreads();
    if (tid < kSarlGroups * 6) {
        const int g = tid & 15, n = tid >> 4;
        jbuf[(n >> 2) * 64 + (n & 3) * 16 + g] = xs[(n >> 2) * 64 + (n & 3) * 16 + g];  // self_state = state[:, 0, :6]
    }
    // lstm_rl.ValueNetwork2 (lstm_rl.py:36-66): mlp1 on every human's row first (ReLU between its 4 layers, none after)
    const bool pairwise = net.L[kL_mlp2_0].w != nullptr;
    const float* lstm_in = xs;
    int ks_in = net.ks_x;
    if (pairwise) {
        dense_mfma<H>(net.L[kL_mlp2_0], xs, net.ks_x, pbuf, net.ks_b, true, nullptr, wave, lane);
        __syncthreads();
        dense_mfma<H>(net.L[kL_mlp2_2], pbuf, net.ks_b, qbuf, net.ks_c, true, nullptr, wave, lane);
        __syncthreads();
        dense_mfma<H>(net.L[kL_att_2], qbuf, net.ks_c, pbuf, net.ks_b, true, nullptr, wave, lane);
        __syncthreads();
        dense_mfma<H>(net.L[kL_att_4], pbuf, net.ks_b, qbuf, net.ks_c, false, nullptr, wave, lane);
        __syncthreads();
        lstm_in = qbuf;
        ks_in = net.ks_c;
    }
    for (int t = 0; t < H; ++t) {
        dense_mfma<1>(net.L[kL_mlp1_0], lstm_in + t * ks_in * 64, ks_in, gates, ks_g, false, nullptr, wave, lane);
        __syncthreads();
        dense_mfma<1>(net.L[kL_mlp1_2], hbuf, ks_h, gates, ks_g, false, gates, wave, lane);  // + (W_hh h + b_hh)
        __syncthreads();
        for (int i = tid; i < hid * kSarlGroups; i += kSarlThreads) {
            const int g = i & 15, j = i >> 4;
            if (t >= hcount[tile * kSarlGroups + g]) continue;  // `mixed` rule: this group's episode has fewer humans
            auto at = [&](int n) { return gates[(n >> 2) * 64 + (n & 3) * 16 + g]; };
            const float ig = 1.0f / (1.0f + expf(-at(j)));
            const float fg = 1.0f / (1.0f + expf(-at(hid + j)));
            const float gg = tanhf(at(2 * hid + j));
            const float og = 1.0f / (1.0f + expf(-at(3 * hid + j)));
            const float c = fg * cbuf[i] + ig * gg;
            cbuf[i] = c;
            hbuf[(j >> 2) * 64 + (j & 3) * 16 + g] = og * tanhf(c);
        }
        __syncthreads();
    }
    for (int i = tid; i < hid * kSarlGroups; i += kSarlThreads) {
        const int g = i & 15, j = i >> 4, n = 6 + j;
        jbuf[(n >> 2) * 64 + (n & 3) * 16 + g] = hbuf[(j >> 2) * 64 + (j & 3) * 16 + g];
    }
    __syncthreads();
    dense_mfma<1>(net.L[kL_mlp3_0], jbuf, net.ks_a, kbuf, net.ks_a, true, nullptr, wave, lane);
    __syncthreads();
    dense_mfma<1>(net.L[kL_mlp3_2], kbuf, net.ks_a, jbuf, net.ks_a, true, nullptr, wave, lane);
    __syncthreads();
    dense_mfma<1>(net.L[kL_mlp3_4], jbuf, net.ks_a, kbuf, net.ks_a, true, nullptr, wave, lane);
    __syncthreads();
    dense_mfma<1>(net.L[kL_mlp3_6], kbuf, net.ks_a, sbuf, net.ks_s, false, nullptr, wave, lane);
    __syncthreads();
    if (tid < kSarlGroups) {
        const size_t G = tile * kSarlGroups + tid;
        if (G < (size_t)n_groups) V[G] = sbuf[tid];
    }
}

// lstm_rl.ValueNetwork1 / ValueNetwork2 for ANY number of humans (H > kSarlMaxHumans): the LSTM is sequential over the
// humans anyway, so human t's input row tile is staged (and, with the interaction module, passed through mlp1 as 16-row
// products) right before LSTM step t; nothing is sized by H.
__global__ __launch_bounds__(kSarlThreads) void lstm_mlp_anyh_kernel(SarlNet net, const float* X, float* V, int n_groups) {
    extern __shared__ float lds[];
    const int H = net.H;
    const int hid = net.L[kL_mlp1_2].K;
    const int ks_h = sarl_ks(hid), ks_g = net.L[kL_mlp1_0].ctiles * 4;
    float* xs = lds;                                       // [ks_x][64]   this step's input row tile
    float* pbuf = xs + net.ks_x * 64;                      // [ks_b][64]   ValueNetwork2.mlp1 ping
    float* qbuf = pbuf + net.ks_b * 64;                    // [ks_c][64]   ... pong
    float* gates = qbuf + net.ks_c * 64;                   // [ks_g][64]
    float* hbuf = gates + ks_g * 64;                       // [ks_h][64]
    float* cbuf = hbuf + ks_h * 64;                        // [hid][16]
    float* jbuf = cbuf + hid * kSarlGroups;                // [ks_a][64]
    float* kbuf = jbuf + net.ks_a * 64;                    // [ks_a][64]
    float* sbuf = kbuf + net.ks_a * 64;                    // [ks_s][64]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const size_t tile = blockIdx.x;
    zero_lds(lds, (size_t)((sbuf + net.ks_s * 64) - lds), tid);
    __syncthreads();
    const float* xg = X + tile * H * net.ks_x * 64;
    const bool pairwise = net.L[kL_mlp2_0].w != nullptr;
    for (int t = 0; t < H; ++t) {
        for (int i = tid; i < net.ks_x * 64; i += kSarlThreads) xs[i] = xg[(size_t)t * net.ks_x * 64 + i];
        __syncthreads();
        if (t == 0 && tid < kSarlGroups * 6) {  // self_state = state[:, 0, :6]
            const int g = tid & 15, n = tid >> 4;
            jbuf[(n >> 2) * 64 + (n & 3) * 16 + g] = xs[(n >> 2) * 64 + (n & 3) * 16 + g];
        }
        const float* lstm_in = xs;
        int ks_in = net.ks_x;
        if (pairwise) {
            dense_mfma<1>(net.L[kL_mlp2_0], xs, net.ks_x, pbuf, net.ks_b, true, nullptr, wave, lane);
            __syncthreads();
            dense_mfma<1>(net.L[kL_mlp2_2], pbuf, net.ks_b, qbuf, net.ks_c, true, nullptr, wave, lane);
            __syncthreads();
            dense_mfma<1>(net.L[kL_att_2], qbuf, net.ks_c, pbuf, net.ks_b, true, nullptr, wave, lane);
            __syncthreads();
            dense_mfma<1>(net.L[kL_att_4], pbuf, net.ks_b, qbuf, net.ks_c, false, nullptr, wave, lane);
            __syncthreads();
            lstm_in = qbuf;
            ks_in = net.ks_c;
        }
        dense_mfma<1>(net.L[kL_mlp1_0], lstm_in, ks_in, gates, ks_g, false, nullptr, wave, lane);
        __syncthreads();
        dense_mfma<1>(net.L[kL_mlp1_2], hbuf, ks_h, gates, ks_g, false, gates, wave, lane);
        __syncthreads();
        for (int i = tid; i < hid * kSarlGroups; i += kSarlThreads) {
            const int g = i & 15, j = i >> 4;
            auto at = [&](int n) { return gates[(n >> 2) * 64 + (n & 3) * 16 + g]; };
            const float ig = 1.0f / (1.0f + expf(-at(j)));
            const float fg = 1.0f / (1.0f + expf(-at(hid + j)));
            const float gg = tanhf(at(2 * hid + j));
            const float og = 1.0f / (1.0f + expf(-at(3 * hid + j)));
            const float c = fg * cbuf[i] + ig * gg;
            cbuf[i] = c;
            hbuf[(j >> 2) * 64 + (j & 3) * 16 + g] = og * tanhf(c);
        }
        __syncthreads();
    }
    for (int i = tid; i < hid * kSarlGroups; i += kSarlThreads) {
        const int g = i & 15, j = i >> 4, n = 6 + j;
        jbuf[(n >> 2) * 64 + (n & 3) * 16 + g] = hbuf[(j >> 2) * 64 + (j & 3) * 16 + g];
    }
    __syncthreads();
    dense_mfma<1>(net.L[kL_mlp3_0], jbuf, net.ks_a, kbuf, net.ks_a, true, nullptr, wave, lane);
    __syncthreads();
    dense_mfma<1>(net.L[kL_mlp3_2], kbuf, net.ks_a, jbuf, net.ks_a, true, nullptr, wave, lane);
    __syncthreads();
    dense_mfma<1>(net.L[kL_mlp3_4], jbuf, net.ks_a, kbuf, net.ks_a, true, nullptr, wave, lane);
    __syncthreads();
    dense_mfma<1>(net.L[kL_mlp3_6], kbuf, net.ks_a, sbuf, net.ks_s, false, nullptr, wave, lane);
    __syncthreads();
    if (tid < kSarlGroups) {
        const size_t G = tile * kSarlGroups + tid;
        if (G < (size_t)n_groups) V[G] = sbuf[tid];
    }
}

__host__ inline size_t sarl_mlp_lds_bytes(const SarlNet& net) {
    const size_t H = (size_t)net.H;
    return sizeof(float) * (64 * (H * (net.ks_a + net.ks_b + net.ks_c + net.ks_s) + net.ks_b + 2 * net.ks_a) + kSarlThreads + 16);
}

// ------------------------------------------------------------------------------------ action selection
// value = reward + pow(gamma, time_step * v_pref) * V (multi_human_rl.py:52); the first strict maximum wins (:54);
// a robot already at its goal stops (:22-23, policy.py:43-49).  best = -1 encodes that stop action.
// The arg-max of env b -> best / action_out (one lane): a robot already at its goal stops (:22-23, policy.py:43-49)
__device__ __forceinline__ void sarl_pick_tail(const SarlCfg& C, const double2* pos, const double2* goal, const double2* rv,
                                               const double* actions, int* best, double* action_out, int b, int bi) {
    const size_t g0 = (size_t)b * (C.H + 1);
    int arg = bi;
    const double dy = pos[g0].y - goal[g0].y, dx = pos[g0].x - goal[g0].x;
    const bool arrived = norm2(dy, dx) < rv[g0].x;  // np.linalg.norm((py - gy, px - gx))
    if (arrived) arg = -1;
    best[b] = (arrived || arg < 0) ? (arrived ? -1 : -2) : arg;  // -2: every value was NaN / -inf (:57-58)
    action_out[2 * b] = arg >= 0 ? actions[2 * arg] : 0.0;
    action_out[2 * b + 1] = arg >= 0 ? actions[2 * arg + 1] : 0.0;
}
// The wave's best (value, action) -> best / action_out of env b: a butterfly keeps the largest value, lowest index on ties
// (= the first strict maximum of the reference's loop; NaN and -inf never win: `value > max_value` is false)
__device__ __forceinline__ void sarl_pick_env(const SarlCfg& C, const double2* pos, const double2* goal, const double2* rv,
                                              const double* actions, int* best, double* action_out, int b, int lane, double bv,
                                              int bi) {
#pragma unroll
    for (int off = kWaveSize / 2; off > 0; off >>= 1) {
        const double ov = __shfl_xor(bv, off);
        const int oi = __shfl_xor(bi, off);
        const bool take = oi >= 0 && (bi < 0 || ov > bv || (ov == bv && oi < bi));
        bv = take ? ov : bv;
        bi = take ? oi : bi;
    }
    if (lane != 0) return;
    sarl_pick_tail(C, pos, goal, rv, actions, best, action_out, b, bi);
}
__device__ __forceinline__ void sarl_select_env(const SarlCfg& C, const double2* pos, const double2* vel, const double2* goal,
                                                const double2* rv, const double* gtime, const double* theta,
                                                const double* actions, double* reward, const float* V, double* values,
                                                int* best, double* action_out, int b, int lane) {
    // one wave per env: lanes stride over the actions
    double bv = -__builtin_inf();
    int bi = -1;
    for (int a = lane; a < C.n_actions; a += kWaveSize) {
        const double r = sarl_reward_of(C, pos, vel, goal, rv, gtime, theta, actions, b, a);  // onestep_lookahead's reward
        reward[(size_t)b * C.n_actions + a] = r;                                              // (kept for cn_sarl_export)
        const double v = r + C.gamma_bar * (double)V[(size_t)b * C.n_actions + a];
        if (values) values[(size_t)b * C.n_actions + a] = v;
        if (v > bv) {
            bv = v;
            bi = a;
        }
    }
    sarl_pick_env(C, pos, goal, rv, actions, best, action_out, b, lane, bv, bi);
}
__global__ void sarl_select_kernel(SarlCfg C, const double2* pos, const double2* vel, const double2* goal, const double2* rv,
                                   const double* gtime, const double* theta, const double* actions, double* reward,
                                   const float* V, double* values, int* best, double* action_out) {
    const int b = blockIdx.x * (blockDim.x / kWaveSize) + (threadIdx.x / kWaveSize);
    if (b >= C.B) return;
    sarl_select_env(C, pos, vel, goal, rv, gtime, theta, actions, reward, V, values, best, action_out, b,
                    threadIdx.x & (kWaveSize - 1));
}


// cn_sarl_sample_step outside the narrow route: the previous step's episode ends leave the set of sampling envs
__global__ void sarl_alive_kernel(int B, uint8_t* alive, const uint8_t* done) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B && done[b]) alive[b] = 0;
}

// What cn_sarl_sample_step adds to the network kernel: value != nullptr -> every tile adds the lookahead reward of its groups and
// stores reward + gamma V, tile b writes env b's replay state; counter != nullptr -> the last workgroup decides as well (otherwise
// sarl_decide_step_kernel does, in front of the transition).
struct SarlDecide {
    int* counter;        // workgroups that have written their V (zero between launches)
    double epsilon;
    uint8_t* alive;      // [B] in/out
    const uint8_t* done; // [B] the previous step's episode-end flags
    int32_t* best;       // [B]
    double* action;      // [B][2]
    float* state_out;    // [B][H][in_dim] at env_stride floats between envs (may be null)
    int64_t env_stride;
    int sort_humans, in_dim;
    double* reward;      // [B][K]
    double* value;       // [B][K] reward + gamma^(dt v_pref) V, written by the tile that computed V
    const double* gtime; // [B]
    uint32_t* mt_key;
    int* mt_pos;
    int* error;
    // occupancy maps (round 6): what sarl_lookahead_kernel computes per (env, human) for the NEXT decision, written by
    // sarl_decide_step_kernel behind its ORCA pass (null without maps)
    double* next_obs_out;  // [B][H][5]
    float* om_out;         // [B][H][cells * channels]
    int side_wg;           // sarl_narrow_kernel: the LAST workgroup of the grid is not a tile — it writes the replay-memory states
    // cn_sarl_values (ABI v11): the rows come from the caller — joint states [ext_groups][H][13] float32 as cn_sarl_transform /
    // cn_sarl_sample_step wrote them (a replay memory's states) — instead of being built from the envs: V of each, nothing else
    const float* x_rows;
    int ext_groups;
};
// The same network for a FEW decisions (the single-episode sampling of train.py:156-170: one env, 81 groups = 6 tiles): the
// one-tile kernels above put a decision on 6 of 256 CUs for ~40 us.  Here a tile is ONE 16-row MFMA tile holding
// 16 / H whole groups — row r = (group r / H, human r % H), 3 groups x 5 humans + 1 idle row at the shipped size — so one
// decision spreads over 27 workgroups, each with a fifth of the matrix work behind the same chain of layers.  X is built in
// LDS by the workgroup that consumes it (sarl_feature_row: no feature kernel, nothing materialised); the mean over a group's
// humans, the softmax and the weighted feature sum run over ROWS of the tile instead of over row tiles, in the order and
// with the partial-sum slicing of sarl_mlp_pipe_kernel / dense_vec1<H>, so V is bit-identical to that kernel's.
// With one row tile a k-step is ONE MFMA, so a layer is bound by the latency of its weights, not by the matrix pipe: a wave
// holds the B fragments of a whole column tile in registers (up to kNarrowK k-steps, 8 waves x 256 VGPRs) and requests the
// NEXT layer's before it starts this layer's MFMAs — one L2 round trip per layer, hidden behind the previous layer, instead
// of one per five k-steps.  No `mixed` rule (every group has H humans), no occupancy maps, H <= 8.
constexpr int kNarrowWaves = 8, kNarrowThreads = kNarrowWaves * 64;
constexpr int kNarrowK = 40;  // k-steps of a column tile held in registers (K = 150 is 38 -> kpad 40); longer layers loop on
struct BTile {
    float b[kNarrowK];
    float bias;
};
__device__ __forceinline__ BTile narrow_fetch(const PackedLinear& P, int ct, int lane) {
    BTile t;
    const bool mine = ct < P.ctiles;  // (wave-uniform: a wave without a column tile of this layer requests nothing)
    const int c = mine ? ct : 0;
    const gfloat_p w = as_global(P.w) + (size_t)c * P.kpad * 64 + lane;
#pragma unroll
    for (int g = 0; g < kNarrowK / kSarlKChunk; ++g) {
        if (mine && g * kSarlKChunk < P.kpad) {
#pragma unroll
            for (int j = 0; j < kSarlKChunk; ++j) t.b[g * kSarlKChunk + j] = w[(g * kSarlKChunk + j) * 64];
        } else {
#pragma unroll
            for (int j = 0; j < kSarlKChunk; ++j) t.b[g * kSarlKChunk + j] = 0.0f;
        }
    }
    t.bias = mine ? as_global(P.bias)[c * 16 + (lane & 15)] : 0.0f;
    return t;
}
// out[r][n] = act(bias[n] + extra[r][n] + sum_k in[r][k] W[n][k]) for the 16 rows of the tile: dense_mfma<1>'s arithmetic
// (k-steps in order into one accumulator from zero, bias + extra added last).  `first` = the fragments of column tile `wave`.
// The k loop is straight-line code of G x 5 k-steps, G = 3 / 5 / 8 by the layer's length (fragments past kpad are zero in
// registers and meet finite LDS words: + 0.0f), so that the A reads from LDS and the MFMAs pipeline without a branch between.
template <int G>
__device__ __forceinline__ f32x4 narrow_k_loop(const float* afrag, const BTile& t) {
    float a[G * kSarlKChunk];
#pragma unroll
    for (int k = 0; k < G * kSarlKChunk; ++k) a[k] = afrag[k * 64];
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < G * kSarlKChunk; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k], t.b[k], acc, 0, 0, 0);
    return acc;
}
__device__ __forceinline__ void dense_narrow(const PackedLinear& P, const float* in, float* out, bool relu, const float* extra,
                                             int wave, int lane, const BTile& first) {
    const int col = lane & 15, quad = lane >> 4;
    BTile t = first;
    for (int ct = wave; ct < P.ctiles; ct += kNarrowWaves) {
        const bool more = ct + kNarrowWaves < P.ctiles;  // (150-wide layers: ten column tiles on eight waves)
        BTile t2;
        if (more) t2 = narrow_fetch(P, ct + kNarrowWaves, lane);
        const int frag_off = ((ct * 4 + (col >> 2)) * 64) + (col & 3) * 16 + quad * 4;
        f32x4 addend = {t.bias, t.bias, t.bias, t.bias};
        if (extra) addend += *reinterpret_cast<const f32x4*>(extra + frag_off);
        const float* afrag = in + lane;
        f32x4 acc;
        if (P.kpad <= 3 * kSarlKChunk) acc = narrow_k_loop<3>(afrag, t);
        else if (P.kpad <= 4 * kSarlKChunk) acc = narrow_k_loop<4>(afrag, t);  // (61 inputs: 16 k-steps)
        else if (P.kpad <= 5 * kSarlKChunk) acc = narrow_k_loop<5>(afrag, t);
        else acc = narrow_k_loop<kNarrowK / kSarlKChunk>(afrag, t);
        if (P.kpad > kNarrowK) {  // wider than the shipped layers: the rest of the k loop straight from L2
            const gfloat_p w = as_global(P.w) + (size_t)ct * P.kpad * 64 + lane;
            for (int k = kNarrowK; k < P.kpad; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[k * 64], w[k * 64], acc, 0, 0, 0);
        }
        f32x4 v = acc + addend;
        if (relu) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = v[i] > 0.0f ? v[i] : 0.0f;
        }
        *reinterpret_cast<f32x4*>(out + frag_off) = v;
        if (more) t = t2;
    }
}

// (the four float64 helpers below are CALLED: inlined — tried in round 6 — the 2 KB / lane of scratch stays, it is the libm's
// private arrays, and 105 VGPRs of the network spill; the reservation costs nothing at launch: profiles/r06_scratch_launch.txt)
#define CN_NARROW_CALL __noinline__
// The decision behind the network (cn_sarl_sample_step), by the last workgroup of sarl_narrow_kernel: one wave per env.  Not
// inlined: its float64 reward / rotation code (registers, the libm's private arrays) stays out of the network's allocation.
__device__ CN_NARROW_CALL void narrow_decide(const SarlCfg& C, const SarlDecide& D, const double2* pos, const double2* goal,
                                           const double2* rv, int wave, int lane, const double* actions) {
    for (int b = wave; b < C.B; b += kNarrowWaves) {
        double bv = -__builtin_inf();
        int bi = -1;
        for (int a = lane; a < C.n_actions; a += kWaveSize) {
            const double v = __hip_atomic_load(&D.value[(size_t)b * C.n_actions + a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v > bv) {
                bv = v;
                bi = a;
            }
        }
        sarl_pick_env(C, pos, goal, rv, actions, D.best, D.action, b, lane, bv, bi);
        if (lane == 0) {
            // alive: envs still sampling.  The episode-end flags of the PREVIOUS step are folded in here (explorer.py:56-65's
            // `while not done` per env) rather than by a kernel of their own behind cn_step.
            const bool keep = D.alive[b] && !(D.done && D.done[b]);
            D.alive[b] = keep ? 1 : 0;
            sarl_explore_env(C.B, C.n_actions, D.epsilon, D.mt_key, D.mt_pos, actions, !keep, D.best, D.action, nullptr, D.error, b);
        }
    }
}
// The joint state of env b for the replay memory (sarl_transform_row), on the idle wave of tile b
__device__ CN_NARROW_CALL void narrow_transform(const SarlCfg& C, const SarlDecide& D, const double2* pos, const double2* vel,
                                              const double2* goal, const double2* rv, const double* theta, int b, int h, bool maps) {
    sarl_transform_row(C, D.in_dim, D.sort_humans, pos, vel, goal, rv, theta, D.state_out, D.env_stride, b, h, maps);
}
// ... and its occupancy maps (the CURRENT human states, multi_human_rl.py:96-105) shared by the 64 lanes of that wave: a human's
// lane alone needs 30 us of float64 trigonometry for its map — longer than the whole network beside it
__device__ CN_NARROW_CALL void narrow_transform_maps(const SarlCfg& C, const SarlDecide& D, const double2* pos, const double2* vel, int b,
                                                   int lane, char* scratch) {
    const size_t g0 = (size_t)b * (C.H + 1);
    occupancy_maps_cooperative(
        C, 1, lane, kWaveSize, scratch,
        [&](int, int j, double& px, double& py, double& vx, double& vy) {
            px = pos[g0 + 1 + j].x, py = pos[g0 + 1 + j].y, vx = vel[g0 + 1 + j].x, vy = vel[g0 + 1 + j].y;
        },
        [] { wave_lds_sync(); },
        [&](int, int i) { return D.state_out + (size_t)b * D.env_stride + (size_t)i * D.in_dim + 13; });
}
// onestep_lookahead's reward of one (env, action) group, for the tile that holds it (not inlined: float64, the libm's arrays)
__device__ CN_NARROW_CALL double narrow_reward(const SarlCfg& C, const double2* pos, const double2* vel, const double2* goal,
                                             const double2* rv, const double* gtime, const double* theta, const double* actions,
                                             int b, int a) {
    return sarl_reward_of(C, pos, vel, goal, rv, gtime, theta, actions, b, a);
}

// LSTM (compile time): lstm_rl.ValueNetwork1 (lstm_rl.py:9-33) instead — the tile's rows are its 16 / H GROUPS, the humans are
// the LSTM's steps: X of step t is a row tile of its own (xs[t]), the input half of the gates of every step (W_ih x_t + b_ih)
// is computed up front with each wave holding its column tile of W_ih across the steps, the recurrent half with W_hh held in
// registers across them; then the joint MLP on [self_state | h].  dense_mfma<1>'s arithmetic layer by layer and the gate
// expressions of lstm_mlp_kernel: V is bit-identical to that kernel's.
template <bool LSTM = false>
__global__ __launch_bounds__(kNarrowThreads) void sarl_narrow_kernel(SarlNetRef net, SarlCfg C, const double2* pos, const double2* vel,
                                                                     const double2* goal, const double2* rv, const double* theta,
                                                                     const double* actions, const float* orca_vel, double* next_obs,
                                                                     float* V, SarlDecide D, const float* om) {
    extern __shared__ float lds[];
    float* xs = lds;                          // [ks_x][64]  X of the tile
    float* bufA = xs + net.ks_x * 64;         // [ks_a][64]  wide hidden layers
    float* bufB = bufA + net.ks_a * 64;       // [ks_b][64]  mlp1 output (h2), then attention.2
    float* bufC = bufB + net.ks_b * 64;       // [ks_c][64]  mlp2 output (per-human feature)
    float* gbuf = bufC + net.ks_c * 64;       // [ks_b][64]  per row: the mean of h2 over the humans of the row's group
    float* jbuf = gbuf + net.ks_b * 64;       // [ks_a][64]  joint state (row = group) / value-head ping
    float* kbuf = jbuf + net.ks_a * 64;       // [ks_a][64]  global attention term
    float* mbuf = kbuf + net.ks_a * 64;       // [ks_a][64]  value-head pong
    float* sbuf = mbuf + net.ks_a * 64;       // [64]        attention scores -> weights (row r at word r)
    float* vbuf = sbuf + 64;                  // [kSarlThreads] partial sums of attention.4
    // LSTM: xs [H][ks_x][64] | gx [H][ks_g][64] input half of every step's gates | gates [ks_g][64] | hbuf [ks_h][64] |
    // cbuf [hid][16] | jbuf, kbuf [ks_a][64] | sbuf | vbuf   (net.nf = the hidden width; sarl_narrow_lds_bytes)
    const int lstm_ks_g = LSTM ? (int)((net.L[kL_mlp1_0].dims >> 8) & 0xffu) * 4 : 0, lstm_hid = LSTM ? net.nf : 0;
    float* const gx = xs + C.H * net.ks_x * 64;
    float* const gates = gx + C.H * lstm_ks_g * 64;
    float* const hbuf = gates + lstm_ks_g * 64;
    float* const cbuf = hbuf + sarl_ks(lstm_hid) * 64;
    if (LSTM) {
        jbuf = cbuf + lstm_hid * kSarlGroups, kbuf = jbuf + net.ks_a * 64, mbuf = kbuf;
        sbuf = kbuf + net.ks_a * 64, vbuf = sbuf + 64;
    }
    int* hc = reinterpret_cast<int*>(vbuf + kSarlThreads);  // [16] humans present in the tile's groups (H unless the `mixed` rule)
    int* const hl = hc + kSarlGroups;                       // [16] cn_sarl_sample_step: the group's env is still sampling
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int H = C.H, GT = kSarlGroups / H, rows = GT * H;
    // cn_sarl_sample_step on the two-launch route: an env whose episode is over (alive[b] && !done[b] is false: the flags as the
    // PREVIOUS call left them) needs no decision — a tile none of whose groups samples returns behind its prologue, the replay
    // state of such an env is not written, sarl_decide_step_kernel skips its transition.  A caller that streams calls past the
    // end of an episode (it cannot know the end without a round trip) pays two near-empty launches per dead step.
    const bool skip_dead = D.value != nullptr && D.counter == nullptr && D.alive != nullptr;
    const auto sampling = [&](int b) { return !skip_dead || (D.alive[b] != 0 && !(D.done != nullptr && D.done[b] != 0)); };
    // where row r = (group r / H, human r % H) of the tile keeps its features: its own row of the one X tile, or (LSTM) row
    // `group` of its human's X tile
    const auto xrow = [&](int r) { return LSTM ? (r % H) * net.ks_x * 64 + r / H : r; };
    const int n_groups = D.x_rows != nullptr ? D.ext_groups : C.B * C.n_actions;
    const size_t tile = blockIdx.x;
    const unsigned n_tiles = gridDim.x - (unsigned)D.side_wg;
    const SarlNetRef* n = &net;
    if (D.side_wg && blockIdx.x == n_tiles) {
        // cn_sarl_sample_step: the CURRENT joint state of every env for the replay memory (sarl_transform_row; nothing of it
        // depends on the network) by a workgroup of its own, beside the tiles on another CU — with occupancy maps a state is
        // ~5 us of float64 trigonometry even when a wave's lanes share it.  One wave per env.
        const bool coop = C.with_om && !D.sort_humans && occupancy_coop_ok(C, occupancy_coop_bytes(H), 1);
        char* scratch = reinterpret_cast<char*>(lds) + (size_t)wave * ((occupancy_coop_bytes(H) + 15) & ~(size_t)15);
        for (int b = wave; b < C.B; b += kNarrowWaves) {
            if (!sampling(b)) continue;
            if (lane < H) narrow_transform(C, D, pos, vel, goal, rv, theta, b, lane, !coop);
            if (coop) narrow_transform_maps(C, D, pos, vel, b, lane, scratch);
        }
        return;
    }
    CN_SARL_CLOCK_BEGIN();
    // (cadrl.ValueNetwork: its four layers live in the mlp3 slots)
    BTile cur = narrow_fetch(layer_of(*n, C.cadrl ? kL_mlp3_0 : kL_mlp1_0), wave, lane);
    // every word of LDS starts finite (k padding meets zero weights); meanwhile the tile's rows of X in registers
    {
        f32x4* z = reinterpret_cast<f32x4*>(lds);
        for (int i = tid; i < (int)(vbuf - lds) / 4; i += kNarrowThreads) z[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    double my_reward = 0.0;
    const bool head_lane = wave == kNarrowWaves - 1 && lane < GT && tile * GT + lane < (size_t)n_groups;
    float f[13];
    bool row_valid = false;
    if (tid < rows) {
        const int g = tid / H, h = tid - g * H;
        const size_t G = tile * GT + g;
        row_valid = G < (size_t)n_groups;
        if (row_valid && D.x_rows != nullptr) {
            const float* xr = D.x_rows + (G * H + h) * 13;
#pragma unroll
            for (int k = 0; k < 13; ++k) f[k] = xr[k];
        } else if (row_valid)
            sarl_feature_row(C, (int)(G / C.n_actions), (int)(G % C.n_actions), h, pos, goal, rv, theta, actions, next_obs, vel,
                             orca_vel, f);
        if (h == 0) {  // len(state.human_states): under the `mixed` rule the env's absent humans are parked behind the present ones
            int present = H;
            if (row_valid && D.x_rows == nullptr) {
                const size_t e0 = (G / C.n_actions) * (size_t)(H + 1);
                present = 0;
                for (int j = 0; j < H; ++j) present += is_parked(pos[e0 + 1 + j]) ? 0 : 1;
            }
            hc[g] = present;
            hl[g] = (row_valid && (D.x_rows != nullptr || sampling((int)(G / C.n_actions)))) ? 1 : 0;
        }
        // (occupancy maps) where this row's map starts in `om`; vbuf is not part of the zeroed region
        if (om != nullptr) reinterpret_cast<int*>(vbuf)[tid] = row_valid ? (int)(((G / C.n_actions) * H + h) * (size_t)(D.in_dim - 13)) : -1;
    }
    lds_barrier();
    CN_SARL_TICK(0);
    if (skip_dead) {
        int live = 0;
        for (int g = 0; g < GT; ++g) live |= hl[g];
        if (live == 0) return;  // (uniform: every thread reads the same words)
    }
    if (row_valid) {
        float* const x = xs + xrow(tid);
#pragma unroll
        for (int k = 0; k < 13; ++k) x[(k >> 2) * 64 + (k & 3) * 16] = f[k];
    }
    if (om != nullptr) {
        // occupancy maps (multi_human_rl.py:46-49): columns 13.. of a row are its human's map among the humans' NEXT states —
        // the same for every action of the env (sarl_lookahead_kernel or the previous call's sarl_decide_step_kernel wrote
        // them).  Four consecutive cells per thread (one 16-byte load), the row's offset into `om` from its own thread (vbuf)
        const int extra = D.in_dim - 13, quads = extra >> 2;  // (cells x channels: 16 x 3 at the shipped size)
        const int* row_om = reinterpret_cast<const int*>(vbuf);
        for (int i = tid; i < rows * quads; i += kNarrowThreads) {
            const int r = i / quads, q = i - r * quads;
            const int base = row_om[r];
            if (base >= 0) {
                const f32x4 m = *reinterpret_cast<const f32x4*>(om + base + 4 * q);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int kk = 13 + 4 * q + j;
                    xs[(kk >> 2) * 64 + (kk & 3) * 16 + xrow(r)] = m[j];
                }
            }
        }
        for (int i = tid; i < rows * (extra & 3); i += kNarrowThreads) {  // (a cell count that is not a multiple of four)
            const int r = i / (extra & 3), k = 4 * quads + i - r * (extra & 3);
            const int base = row_om[r], kk = 13 + k;
            if (base >= 0) xs[(kk >> 2) * 64 + (kk & 3) * 16 + xrow(r)] = om[base + k];
        }
    }
    BTile nxt = narrow_fetch(layer_of(*n, C.cadrl ? kL_mlp3_2 : kL_mlp1_2), wave, lane);
    lds_barrier();
    CN_SARL_TICK(1);
    // What every tile does with the V of its groups (on the value head's wave), and what follows it under cn_sarl_sample_step
    const auto finish = [&](float v) {
        int arrived = 0;
        if (wave == kNarrowWaves - 1) {
            if (head_lane) {
                const size_t G = tile * GT + lane;
                V[G] = v;
                // multi_human_rl.py:52, as sarl_select_env.  An agent-scope atomic store: written through to where every XCD's
                // agent-scope load finds it — no write-back of this XCD's whole L2 (a release fence) for 3 doubles
                if (D.value)
                    __hip_atomic_store(&D.value[G], my_reward + C.gamma_bar * (double)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (D.counter) {
                __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): the stores above have been acknowledged before the tile counts as arrived
                if (lane == 0) arrived = atomicAdd(D.counter, 1) + 1;
            }
        }
        if (!D.counter) return;  // cn_sarl_select, or the decision is sarl_decide_step_kernel's: the network only
        // ---- the workgroup that finishes LAST decides for every env, one wave per env — arg-max of reward + gamma V, the
        // epsilon-greedy draw on the env's own stream (sarl_explore_env) — instead of three more launches behind this one (the
        // joint state for the replay memory was written by tile b meanwhile).
        int* last = reinterpret_cast<int*>(sbuf);
        if (wave == kNarrowWaves - 1 && lane == 0) {
            *last = arrived == (int)n_tiles ? 1 : 0;
            if (*last) atomicExch(D.counter, 0);  // ready for the next launch
        }
        __syncthreads();
        if (!*last) return;
        narrow_decide(C, D, pos, goal, rv, wave, lane, actions);
    };
    const auto reward_of_my_group = [&]() {
        // cn_sarl_sample_step: the reward of the tile's groups on the lanes that will hold their V — the value head's wave, which has
        // no column tile of the 100-wide layers: this float64 chain runs beside a 100-wide layer's MFMAs.  The decision behind the
        // network then only compares reward + gamma V.
        if (D.value && head_lane) {
            const size_t G = tile * GT + lane;
            my_reward = narrow_reward(C, pos, vel, goal, rv, D.gtime, theta, actions, (int)(G / C.n_actions), (int)(G % C.n_actions));
            D.reward[G] = my_reward;
        }
    };
    const auto replay_state_of_my_env = [&]() {
        // ... and (without the side workgroup) the CURRENT joint state of env b for the replay memory, by tile b's idle wave.
        // A small action table has fewer tiles than envs (one human, 13 actions, 6 envs: 5 tiles): the tiles stride over the envs.
        if (D.value && D.state_out && !D.side_wg && wave == kNarrowWaves - 1) {
            // (occupancy maps: the wave's lanes share them; mbuf — the value head's pong buffer — is idle until mlp3.0)
            const bool coop = C.with_om && !D.sort_humans && occupancy_coop_ok(C, sizeof(float) * 64 * (size_t)net.ks_a, 1);
            for (size_t b = tile; b < (size_t)C.B; b += n_tiles) {
                if (!sampling((int)b)) continue;
                if (lane < H) narrow_transform(C, D, pos, vel, goal, rv, theta, (int)b, lane, !coop);
                if (coop) narrow_transform_maps(C, D, pos, vel, (int)b, lane, reinterpret_cast<char*>(mbuf));
            }
        }
    };
    if constexpr (LSTM) {
        const int hid = lstm_hid, ks_g = lstm_ks_g;
        const PackedLinear Pi = layer_of(*n, kL_mlp1_0), Ph = layer_of(*n, kL_mlp1_2);  // weight_ih_l0 + bias_ih, weight_hh_l0 + bias_hh
        const int col = lane & 15, quad = lane >> 4;
        // W_hh: 4 hid / 16 column tiles on eight waves — both of a wave's tiles stay in registers across the steps (`nxt`: tile `wave`)
        BTile hh2 = narrow_fetch(Ph, wave + kNarrowWaves, lane);
        {   // the input half of the gates of EVERY step: gx[t] = W_ih x_t + b_ih (dense_mfma<1> with no extra term)
            BTile t = cur;
            for (int ct = wave; ct < Pi.ctiles; ct += kNarrowWaves) {
                const bool more = ct + kNarrowWaves < Pi.ctiles;
                BTile t2;
                if (more) t2 = narrow_fetch(Pi, ct + kNarrowWaves, lane);
                const int frag_off = ((ct * 4 + (col >> 2)) * 64) + (col & 3) * 16 + quad * 4;
                const f32x4 addend = {t.bias, t.bias, t.bias, t.bias};
                for (int tt = 0; tt < H; ++tt) {
                    const float* afrag = xs + tt * net.ks_x * 64 + lane;
                    const f32x4 acc = Pi.kpad <= 3 * kSarlKChunk ? narrow_k_loop<3>(afrag, t) : narrow_k_loop<4>(afrag, t);
                    *reinterpret_cast<f32x4*>(gx + tt * ks_g * 64 + frag_off) = acc + addend;
                }
                if (more) t = t2;
            }
        }
        reward_of_my_group();      // (the value head's wave has one column tile of W_ih where waves 0..4 have two)
        replay_state_of_my_env();
        cur = narrow_fetch(layer_of(*n, kL_mlp3_0), wave, lane);
        lds_barrier();
        for (int t = 0; t < H; ++t) {
            // gates = (W_hh h + b_hh) + gx[t]: h = 0 at the first step, multiplied out like every other (lstm_mlp_kernel does)
#pragma unroll
            for (int ci = 0; ci < 2; ++ci) {
                const int ct = wave + ci * kNarrowWaves;
                if (ct < Ph.ctiles) {
                    const BTile& w = ci ? hh2 : nxt;
                    const int frag_off = ((ct * 4 + (col >> 2)) * 64) + (col & 3) * 16 + quad * 4;
                    f32x4 addend = {w.bias, w.bias, w.bias, w.bias};
                    addend += *reinterpret_cast<const f32x4*>(gx + t * ks_g * 64 + frag_off);
                    const f32x4 acc = narrow_k_loop<3>(hbuf + lane, w);
                    *reinterpret_cast<f32x4*>(gates + frag_off) = acc + addend;
                }
            }
            lds_barrier();
            for (int i = tid; i < hid * kSarlGroups; i += kNarrowThreads) {
                const int g = i & 15, j = i >> 4;
                if (g >= GT || t >= hc[g]) continue;  // (`mixed` rule: this group's episode has fewer humans)
                auto at = [&](int k) { return gates[(k >> 2) * 64 + (k & 3) * 16 + g]; };
                const float ig = 1.0f / (1.0f + expf(-at(j)));
                const float fg = 1.0f / (1.0f + expf(-at(hid + j)));
                const float gg = tanhf(at(2 * hid + j));
                const float og = 1.0f / (1.0f + expf(-at(3 * hid + j)));
                const float c = fg * cbuf[i] + ig * gg;
                cbuf[i] = c;
                hbuf[(j >> 2) * 64 + (j & 3) * 16 + g] = og * tanhf(c);
            }
            lds_barrier();
        }
        nxt = narrow_fetch(layer_of(*n, kL_mlp3_2), wave, lane);
        // joint state [self_state = state[:, 0, :6] | h_n], row = group (lstm_rl.py:29-31)
        if (tid < kSarlGroups * 6 && (tid & 15) < GT) {
            const int g = tid & 15, f6 = tid >> 4;
            jbuf[(f6 >> 2) * 64 + (f6 & 3) * 16 + g] = xs[(f6 >> 2) * 64 + (f6 & 3) * 16 + g];
        }
        for (int i = tid; i < hid * kSarlGroups; i += kNarrowThreads) {
            const int g = i & 15, j = i >> 4, f = 6 + j;
            if (g < GT) jbuf[(f >> 2) * 64 + (f & 3) * 16 + g] = hbuf[(j >> 2) * 64 + (j & 3) * 16 + g];
        }
        lds_barrier();
        dense_narrow(layer_of(*n, kL_mlp3_0), jbuf, kbuf, true, nullptr, wave, lane, cur);
        cur = narrow_fetch(layer_of(*n, kL_mlp3_4), wave, lane);
        lds_barrier();
        dense_narrow(layer_of(*n, kL_mlp3_2), kbuf, jbuf, true, nullptr, wave, lane, nxt);
        nxt = narrow_fetch(layer_of(*n, kL_mlp3_6), wave, lane);
        lds_barrier();
        dense_narrow(layer_of(*n, kL_mlp3_4), jbuf, kbuf, true, nullptr, wave, lane, cur);
        lds_barrier();
        dense_narrow(layer_of(*n, kL_mlp3_6), kbuf, jbuf, false, nullptr, wave, lane, nxt);  // column 0 of the tile: row r at word r
        lds_barrier();
        const float v = (wave == kNarrowWaves - 1 && lane < GT) ? jbuf[lane] : 0.0f;
        CN_SARL_CLOCK_END();
        finish(v);
        return;
    }
    if (C.cadrl) {
        // cadrl.ValueNetwork (cadrl.py:22-29): the same MLP for every (robot, human) row — cadrl_mlp_kernel's four layers on the
        // tile's 16 rows — then the minimum over the humans of a group (cadrl.py:162-163: the first minimum's value)
        dense_narrow(layer_of(*n, kL_mlp3_0), xs, bufA, true, nullptr, wave, lane, cur);
        cur = narrow_fetch(layer_of(*n, kL_mlp3_4), wave, lane);
        lds_barrier();
        dense_narrow(layer_of(*n, kL_mlp3_2), bufA, bufB, true, nullptr, wave, lane, nxt);
        reward_of_my_group();
        nxt = narrow_fetch(layer_of(*n, kL_mlp3_6), wave, lane);
        lds_barrier();
        dense_narrow(layer_of(*n, kL_mlp3_4), bufB, bufA, true, nullptr, wave, lane, cur);
        replay_state_of_my_env();
        lds_barrier();
        dense_narrow(layer_of(*n, kL_mlp3_6), bufA, kbuf, false, nullptr, wave, lane, nxt);  // column 0 of the tile: row r at word r
        lds_barrier();
        float m = 0.0f;
        if (wave == kNarrowWaves - 1 && lane < GT) {
            m = kbuf[lane * H];
            const int cnt = hc[lane];
            for (int h = 1; h < H; ++h) {
                const float v = kbuf[lane * H + h];
                m = (h < cnt && v < m) ? v : m;
            }
        }
        CN_SARL_CLOCK_END();
        finish(m);
        return;
    }
    // self_state = state[:, 0, :6] (sarl.py:36): the first human's row of the group
    float self_val = 0.0f;
    const int sg = tid & 15, sf = tid >> 4;
    if (tid < kSarlGroups * 6 && sg < GT) self_val = xs[(sf >> 2) * 64 + (sf & 3) * 16 + sg * H];
    dense_narrow(layer_of(*n, kL_mlp1_0), xs, bufA, true, nullptr, wave, lane, cur);
    cur = narrow_fetch(layer_of(*n, kL_mlp2_0), wave, lane);
    lds_barrier();
    CN_SARL_TICK(2);
    dense_narrow(layer_of(*n, kL_mlp1_2), bufA, bufB, true, nullptr, wave, lane, nxt);  // h2
    reward_of_my_group();
    nxt = narrow_fetch(layer_of(*n, kL_mlp2_2), wave, lane);
    lds_barrier();
    CN_SARL_TICK(3);
    if (n->with_global) {
        // the mean of h2 over a group's humans, once per (feature word, group) and copied to the group's H rows (it was summed
        // again for every row: five times the loads and divisions; rows beyond the tile's stay zero from the start)
        for (int i = tid; i < n->ks_b * 4 * GT; i += kNarrowThreads) {
            const int g = i % GT, first = (i / GT) * 16 + g * H;
            const int cnt = hc[g];
            float sum = 0.0f;
            for (int h = 0; h < H; ++h) sum += h < cnt ? bufB[first + h] : 0.0f;  // (as sarl_mlp_pipe_kernel masks a `mixed` episode)
            const float mean = sum / (float)cnt;
            for (int h = 0; h < H; ++h) gbuf[first + h] = mean;
        }
    }
    dense_narrow(layer_of(*n, kL_mlp2_0), bufB, bufA, true, nullptr, wave, lane, cur);
    replay_state_of_my_env();
    cur = narrow_fetch(layer_of(*n, kL_att0_global), wave, lane);
    lds_barrier();
    CN_SARL_TICK(4);
    dense_narrow(layer_of(*n, kL_mlp2_2), bufA, bufC, false, nullptr, wave, lane, nxt);  // features
    nxt = narrow_fetch(layer_of(*n, kL_att0_local), wave, lane);
    if (n->with_global) dense_narrow(layer_of(*n, kL_att0_global), gbuf, kbuf, false, nullptr, wave, lane, cur);
    cur = narrow_fetch(layer_of(*n, kL_att_2), wave, lane);
    lds_barrier();
    CN_SARL_TICK(5);
    dense_narrow(layer_of(*n, kL_att0_local), bufB, bufA, true, n->with_global ? kbuf : nullptr, wave, lane, nxt);
    nxt = narrow_fetch(layer_of(*n, kL_mlp3_0), wave, lane);
    lds_barrier();
    CN_SARL_TICK(6);
    dense_narrow(layer_of(*n, kL_att_2), bufA, bufB, true, nullptr, wave, lane, cur);
    cur = narrow_fetch(layer_of(*n, kL_mlp3_2), wave, lane);
    lds_barrier();
    CN_SARL_TICK(7);
    float* const wrow = reinterpret_cast<float*>(hl + kSarlGroups) + wave * 16;  // this wave's copy of the attention weights
    {   // attention.4 (one output) as dense_vec1<H> slices it: kSarlThreads / (16 H) k slices per row, summed in slice order
        const PackedLinear P = layer_of(*n, kL_att_4);
        const int slices = kSarlThreads / (H * 16);
        for (int i = tid; i < slices * 16; i += kNarrowThreads) {
            const int row = i & 15, slice = i >> 4;
            float sum = 0.0f;
            for (int s = slice; s < P.ksteps; s += slices) {
                const gfloat_p w = as_global(P.w) + s * 64;
                const float* x = bufB + s * 64 + row;
                sum += (x[0] * w[0] + x[16] * w[16]) + (x[32] * w[32] + x[48] * w[48]);
            }
            vbuf[slice * 16 + row] = sum;
        }
        lds_barrier();
        CN_SARL_TICK(8);
        // EVERY wave: lane = row (four copies per wave): its score, then the softmax without max subtraction over the group's
        // humans (sarl.py:52-53) — into the wave's OWN copy of the weights, so that the joint state below follows without another
        // workgroup barrier (round 6: wave 0 alone computed them and seven waves waited at a barrier of their own)
        {
            const int r = lane & 15;
            float v = as_global(P.bias)[0];
            for (int s = 0; s < slices; ++s) v += vbuf[s * 16 + r];
            const int g0 = (r / H) * H;
            const bool present = r < rows ? (r - g0) < hc[r / H] : true;
            const float e = present ? expf(v) * (v != 0.0f ? 1.0f : 0.0f) : 0.0f;
            float total = 0.0f;
            for (int h = 0; h < H; ++h) total += __shfl(e, (g0 + h) & 15);  // (row 15 of 3 x 5 wraps: unused)
            if (lane < 16) wrow[lane] = e / total;
        }
    }
    wave_lds_sync();
    CN_SARL_TICK(9);
    // the joint state, row = group: self features, weighted feature sum (sarl.py:60); everything else of jbuf is zero
    if (tid < kSarlGroups * 6 && sg < GT) jbuf[(sf >> 2) * 64 + (sf & 3) * 16 + sg] = self_val;
    const int nf = n->nf;
    for (int i = tid; i < GT * nf; i += kNarrowThreads) {  // (group, feature): the groups the tile really holds
        const int g = i % GT, c = i / GT;
        const int src = (c >> 2) * 64 + (c & 3) * 16 + g * H;
        float sum = 0.0f;
        for (int h = 0; h < H; ++h) sum += wrow[g * H + h] * bufC[src + h];
        const int f = 6 + c;
        jbuf[(f >> 2) * 64 + (f & 3) * 16 + g] = sum;
    }
    lds_barrier();
    CN_SARL_TICK(10);
    dense_narrow(layer_of(*n, kL_mlp3_0), jbuf, mbuf, true, nullptr, wave, lane, nxt);
    nxt = narrow_fetch(layer_of(*n, kL_mlp3_4), wave, lane);
    lds_barrier();
    CN_SARL_TICK(11);
    dense_narrow(layer_of(*n, kL_mlp3_2), mbuf, jbuf, true, nullptr, wave, lane, cur);
    lds_barrier();
    CN_SARL_TICK(12);
    dense_narrow(layer_of(*n, kL_mlp3_4), jbuf, mbuf, true, nullptr, wave, lane, nxt);
    lds_barrier();
    CN_SARL_TICK(13);
    float v = 0.0f;
    if (wave == kNarrowWaves - 1) {  // mlp3.6 as value_head_on_one_wave: 4 k slices of the 16 rows, summed in slice order
        const PackedLinear P = layer_of(*n, kL_mlp3_6);
        const int row = lane & 15, slice = lane >> 4;
        float sum = 0.0f;
        for (int s = slice; s < P.ksteps; s += 4) {
            const gfloat_p w = as_global(P.w) + s * 64;
            const float* x = mbuf + s * 64 + row;
            sum += (x[0] * w[0] + x[16] * w[16]) + (x[32] * w[32] + x[48] * w[48]);
        }
        v = as_global(P.bias)[0];
#pragma unroll
        for (int j = 0; j < 4; ++j) v += __shfl(sum, row + 16 * j);
    }
    CN_SARL_TICK(14);
    CN_SARL_CLOCK_END();
    finish(v);
}
__host__ inline size_t sarl_narrow_lds_bytes(const SarlNet& net, bool lstm = false) {
    if (lstm) {  // sarl_narrow_kernel<true>'s carve: xs, gx, gates, hbuf, cbuf, jbuf, kbuf, sbuf, vbuf, hc
        const size_t H = (size_t)net.H, ks_g = (size_t)net.L[kL_mlp1_0].ctiles * 4, hid = (size_t)net.L[kL_mlp1_2].K;
        return sizeof(float) * (64 * (H * net.ks_x + H * ks_g + ks_g + (size_t)sarl_ks((int)hid) + 2 * (size_t)net.ks_a + 1) +
                                hid * kSarlGroups + kSarlThreads + (2 + kNarrowWaves) * kSarlGroups);
    }
    return sizeof(float) * (64 * (size_t)(net.ks_x + 4 * net.ks_a + 2 * net.ks_b + net.ks_c + 1) + kSarlThreads + (2 + kNarrowWaves) * kSarlGroups);
}

}  // namespace cn
