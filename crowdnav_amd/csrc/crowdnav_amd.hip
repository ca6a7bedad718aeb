// libcrowdnav_amd.so — the C ABI declared in include/crowdnav_amd.h over the HIP kernels of step_kernels.h.
//
// Hot path replaced (reference file:line, /root/reference):
//   CrowdSim.step / onestep_lookahead          crowd_sim/envs/crowd_sim.py:314-420
//   Human.act / Robot.act -> ORCA.predict      crowd_sim/envs/utils/human.py:9-17, policy/orca.py:82-132
//   rvo2 doStep (un-vendored RVO2 library)     orca.py:128
//   Agent.compute_position / step              crowd_sim/envs/utils/agent.py:110-135
//   point_to_segment_dist                      crowd_sim/envs/utils/utils.py:4-26
//   CrowdSim.reset + scenario rules            crowd_sim.py:155-207, 251-312
//   Explorer.run_k_episodes inner loop         crowd_nav/utils/explorer.py:35-72
//
// Built with -ffp-contract=off: float32 ORCA and float64 env arithmetic must round exactly like the CPU
// reference; the only fused operation is the explicit fma in norm2().
#include "engine_host.h"
#include "rollout_fused.h"
#include "records_kernels.h"

thread_local char cn_g_err[512] = "";

// every agent's ORCA velocity on the engine's stream (cn_orca; the first kernel of cn_sarl_select in sarl_abi.hip)
void cn_launch_orca(cn_engine* e, float* out_vel) { CN_LAUNCH_MAXL(e, orca_kernel, grid_envs(e), e->P, e->S, out_vel); }

extern "C" {

const char* cn_last_error(void) { return cn_g_err; }
int cn_abi_version(void) { return 11; }

int cn_create(const cn_config* c, cn_engine** out) {
    if (!c || !out) return fail(CN_ERR_INVALID, "cn_create: NULL argument");
    *out = nullptr;
    if (c->num_envs < 1) return fail(CN_ERR_INVALID, "num_envs must be >= 1 (got %d)", c->num_envs);
    if (c->num_humans < 1 || c->num_humans > 63)
        return fail(CN_ERR_UNSUPPORTED, "num_humans must be in 1..63 (got %d)", c->num_humans);
    if (c->max_neighbors < 0 || c->max_neighbors > cn::kMaxNb)
        return fail(CN_ERR_UNSUPPORTED, "max_neighbors must be in 0..%d (got %d)", cn::kMaxNb, c->max_neighbors);
    if (!(c->time_step > 0.0)) return fail(CN_ERR_INVALID, "time_step must be > 0");
    if (c->scenario_rule != CN_CIRCLE_CROSSING && c->scenario_rule != CN_SQUARE_CROSSING && c->scenario_rule != CN_MIXED)
        return fail(CN_ERR_UNSUPPORTED, "scenario_rule %d not supported", c->scenario_rule);
    if (c->scenario_rule == CN_MIXED) {
        if (c->num_humans < 5 || c->num_humans > 8)
            return fail(CN_ERR_UNSUPPORTED, "scenario_rule mixed draws up to 5 humans per episode: num_humans must be in 5..8 (got %d)",
                        c->num_humans);
        if (c->randomize_attributes && c->robot_policy == CN_ROBOT_ORCA)
            return fail(CN_ERR_UNSUPPORTED,
                        "mixed + randomize_attributes + the ORCA robot: the reference rebuilds the robot's rvo2 simulator "
                        "whenever the number of humans changes (orca.py:95-98), which the per-env radius capture does not model");
    }
    if (c->robot_policy != CN_ROBOT_EXTERNAL && c->robot_policy != CN_ROBOT_ORCA)
        return fail(CN_ERR_INVALID, "robot_policy %d unknown", c->robot_policy);
    if (c->robot_kinematics != CN_HOLONOMIC && c->robot_kinematics != CN_UNICYCLE)
        return fail(CN_ERR_INVALID, "robot_kinematics %d unknown", c->robot_kinematics);
    if (c->robot_kinematics == CN_UNICYCLE && c->robot_policy == CN_ROBOT_ORCA)
        return fail(CN_ERR_INVALID, "the ORCA robot policy is holonomic (orca.py:59); a unicycle robot needs CN_ROBOT_EXTERNAL");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(CN_ERR_NO_DEVICE, "no HIP device visible: the MI355X engine has no CPU fallback");
    if (c->device < 0 || c->device >= ndev) return fail(CN_ERR_INVALID, "device %d out of range", c->device);
    CN_HIP(hipSetDevice(c->device));

    cn_engine* e = new (std::nothrow) cn_engine();
    if (!e) return fail(CN_ERR_INVALID, "out of host memory");
    e->cfg = *c;
    e->stream = nullptr;
    e->io_valid = false;
    e->steps_since_fill = -1;
    e->sarl = nullptr;
    e->orca_fresh = false;
    e->async_fill = false;
    e->rollout_done = nullptr;
    e->next_fill_stream = 0;
    for (int i = 0; i < cn_engine::kFillStreams; ++i) e->fill_streams[i] = nullptr;
    for (uint64_t& c : e->launch_counts) c = 0;
    cn::Params& P = e->P;
    P.B = c->num_envs;
    P.A = c->num_humans + 1;
    P.NC = P.A - 1;
    // Workgroup geometry.  E envs per workgroup (agents = lanes of wave 0), W waves sharing the per-pair phases.
    // Defaults from the MI355X sweep in DESIGN.md; CROWDNAV_AMD_ENVS_PER_WAVE / CROWDNAV_AMD_WAVES_PER_BLOCK
    // override them for tuning.
    const int e_max = cn::kWave / P.A;
    // <= 5 half-planes per agent: 2048 workgroups (2 waves per SIMD) were fastest; the 10-half-plane kernels hold
    // 187 VGPRs (2 resident waves per SIMD) and loop over many more pairs, so they get one env per wave up to
    // 4096 workgroups (H = 20: E = 1 48 M, E = 2 31 M env-steps/s)
    const bool small_lp = (P.NC < c->max_neighbors ? P.NC : c->max_neighbors) <= 5;
    // (r02, candidate-form programs: their (agent, half-plane) items fill one 64-lane pass at 2 envs x 6 agents; E = 3..8
    // take 2-3 passes and lose 25-45 % at 8192-32768 envs, so the small programs never pack more than 2 envs per wave)
    int e_want = env_int("CROWDNAV_AMD_ENVS_PER_WAVE", small_lp ? (P.B > 2048 ? 2 : 1) : (P.B + 4095) / 4096);
    P.E = e_want < 1 ? 1 : (e_want > e_max ? e_max : e_want);
    int w_want = env_int("CROWDNAV_AMD_WAVES_PER_BLOCK", 1);
    const int w_useful = (P.E * P.A * P.NC + cn::kWave - 1) / cn::kWave;  // more waves than pair passes is waste
    if (w_want > w_useful) w_want = w_useful;
    if (w_want > cn::kMaxBlock / cn::kWave) w_want = cn::kMaxBlock / cn::kWave;
    P.threads = cn::kWave * (w_want < 1 ? 1 : w_want);
    P.nA = P.E * P.A;
    P.pairs = P.nA * P.NC;
    P.ring_depth = env_int("CROWDNAV_AMD_RING_DEPTH", 48);
    if (P.ring_depth < 1) P.ring_depth = 1;
    e->maxl = ((P.NC < c->max_neighbors ? P.NC : c->max_neighbors) <= 5) ? 5 : 10;
    P.kd = P.A > cn::kKdLeaf ? 1 : 0;  // a simulator of more than 10 agents splits its kd-tree: visiting order matters at ties
    P.sched = -1;
    e->sched_min = env_int("CROWDNAV_AMD_SCHED_MIN_STEPS", 24);  // shortest call that runs under a schedule (static: at least 48)
    e->sched_force = env_int("CROWDNAV_AMD_SCHED_FORCE", 0) != 0;
    e->sched_reserve = env_int("CROWDNAV_AMD_DYN_RESERVE", 0);  // slots left free beside a dynamic launch under the asynchronous fill
    e->sched_dynamic = env_int("CROWDNAV_AMD_SCHED_DYNAMIC", 1) != 0;
    e->scenario_cache = env_int("CROWDNAV_AMD_SCENARIO_CACHE", 1) != 0;
    e->dyn_visits = env_int("CROWDNAV_AMD_DYN_VISITS", 0);  // 0: by call length (launch_rollout)
    P.dyn_visits = 3;
    {
        hipDeviceProp_t prop;
        const bool ok = hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0;
        e->sched_slots = 12 * (ok ? prop.multiProcessorCount : 256);
    }
    P.kdl = cn::kd_layout(P.nA, P.A, P.E);
    e->smem = cn::smem_bytes(P.nA, P.pairs, e->maxl, P.A, P.E);
    e->gen_wave = env_int("CROWDNAV_AMD_WAVE_SCENARIOS", c->num_humans > 8 ? 1 : 0) != 0;
    P.robot_visible = c->robot_visible ? 1 : 0;
    P.robot_orca = c->robot_policy == CN_ROBOT_ORCA;
    P.robot_unicycle = c->robot_kinematics == CN_UNICYCLE;
    e->async_fill = (c->flags & CN_FLAG_ASYNC_SCENARIO_FILL) != 0 && e->gen_wave;
    P.async_fill = e->async_fill ? 1 : 0;
    // Asynchronous fill: a slot is refilled by the fill launch AFTER the call that consumed it, behind up to one call's worth of
    // other scenarios, and a hard scenario of the reference geometry (20 humans on the 4 m circle: up to 3 M attempts) is tens
    // of milliseconds of one wave — so the ring is the latency buffer.  Measured (r06, 4096 x 20, 999-step calls, every
    // scenario generated afresh): depth 48 / 96 / 144 = 16 % / 3 % / 0 % of the env-steps paused.  0.6 GB of the 288.
    if (e->async_fill && !getenv("CROWDNAV_AMD_RING_DEPTH")) P.ring_depth = 144;
    e->fill_queue_wgs = env_int("CROWDNAV_AMD_FILL_QUEUE_WGS", 1024);  // 0: one workgroup per (env, slot) (rounds 2-5)
    P.dt = c->time_step;
    P.time_limit = c->time_limit;
    P.success_reward = c->success_reward;
    P.collision_penalty = c->collision_penalty;
    P.discomfort_dist = c->discomfort_dist;
    P.discomfort_factor = c->discomfort_penalty_factor;
    P.robot_safety = c->robot_safety_space;
    P.human_safety = c->human_safety_space;
    P.orca.neighbor_dist = (float)c->neighbor_dist;
    P.orca.inv_time_horizon = 1.0f / (float)c->time_horizon;
    P.orca.inv_time_step = 1.0f / (float)c->time_step;
    P.orca.max_neighbors = c->max_neighbors;
    e->C.num_agents = P.A;
    e->C.rule = c->scenario_rule;
    e->C.randomize = c->randomize_attributes ? 1 : 0;
    e->C.circle_radius = c->circle_radius;
    e->C.square_width = c->square_width;
    e->C.discomfort_dist = c->discomfort_dist;
    e->C.human_radius = c->human_radius;
    e->C.human_v_pref = c->human_v_pref;
    e->C.robot_radius = c->robot_radius;
    e->C.robot_v_pref = c->robot_v_pref;
    // give-up threshold of the rejection sampling: ~seconds of GPU time in either generator family
    int cap_log2 = env_int("CROWDNAV_AMD_MAX_ATTEMPTS_LOG2", c->num_humans > 8 ? 23 : 20);
    if (cap_log2 < 6) cap_log2 = 6;
    if (cap_log2 > 40) cap_log2 = 40;
    e->C.max_attempts = 1ull << cap_log2;
    e->C.error = nullptr;

    const size_t n = (size_t)P.B * P.A;
    int rc = CN_OK;
    cn::StateView& S = e->S;
    if ((rc = dev_alloc(e, &S.pos, n)) || (rc = dev_alloc(e, &S.vel, n)) || (rc = dev_alloc(e, &S.goal, n)) ||
        (rc = dev_alloc(e, &S.rv, n)) || (rc = dev_alloc(e, &S.gtime, (size_t)P.B)) ||
        (rc = dev_alloc(e, &S.theta, (size_t)P.B)) ||
        (rc = dev_alloc(e, &S.rsim_radius, n)) || (rc = dev_alloc(e, &S.rsim_max_speed, (size_t)P.B)) ||
        (rc = dev_alloc(e, &S.rsim_valid, (size_t)P.B)) || (rc = dev_alloc(e, &S.mt_key, (size_t)624 * P.B)) ||
        (rc = dev_alloc(e, &S.mt_pos, (size_t)P.B)) || (rc = dev_alloc(e, &e->probe_key, (size_t)624)) ||
        (rc = dev_alloc(e, &S.ring_pos, n * P.ring_depth)) || (rc = dev_alloc(e, &S.ring_goal, n * P.ring_depth)) ||
        (rc = dev_alloc(e, &S.ring_rv, n * P.ring_depth)) ||
        (rc = dev_alloc(e, &S.ring_mt_key, e->gen_wave ? (size_t)64 : (size_t)624 * cn::kRedoLanes)) ||
        (rc = dev_alloc(e, &S.redo_list, e->gen_wave ? (size_t)1 : (size_t)P.B * P.ring_depth)) ||
        (rc = dev_alloc(e, &S.redo_count, (size_t)1)) ||
        (rc = dev_alloc(e, &e->summary_scratch, (size_t)cn::kSummaryBlocks * cn::kSummaryFields + 1)) ||
        (rc = dev_alloc(e, &S.ring_filled_in, (size_t)P.B)) || (rc = dev_alloc(e, &S.ring_filled_out, (size_t)P.B)) ||
        (rc = dev_alloc(e, &S.ring_ready, e->async_fill ? (size_t)P.B * P.ring_depth : (size_t)1)) ||
        (rc = dev_alloc(e, &S.ring_claim, e->async_fill ? (size_t)P.B * P.ring_depth : (size_t)1)) ||
        (rc = dev_alloc(e, &S.ep_word, (size_t)P.B)) || (rc = dev_alloc(e, &S.dyn_queue, (size_t)P.B + 1)) ||
        (rc = dev_alloc(e, &e->fill_list, e->async_fill ? (size_t)cn_engine::kFillStreams * (2 + 2 * (size_t)P.B * P.ring_depth) : (size_t)2)) ||
        (rc = dev_alloc(e, &S.cache_pos, e->gen_wave ? (size_t)cn::kScenarioCacheMax * P.A : (size_t)1)) ||
        (rc = dev_alloc(e, &S.cache_goal, e->gen_wave ? (size_t)cn::kScenarioCacheMax * P.A : (size_t)1)) ||
        (rc = dev_alloc(e, &S.cache_rv, e->gen_wave ? (size_t)cn::kScenarioCacheMax * P.A : (size_t)1)) ||
        (rc = dev_alloc(e, &S.cache_state, (size_t)cn::kScenarioCacheMax)) ||
        (rc = dev_alloc(e, &S.kd_order, P.kd ? n * cn::kd_row_bytes(P.A) : (size_t)4)) ||
        (rc = dev_alloc(e, &S.kd_valid, P.kd ? n : (size_t)4)) ||
        (rc = dev_alloc(e, &S.wg_partial, (size_t)P.B * (CN_SUMMARY_FIELDS + 1))) ||
        (rc = dev_alloc(e, &S.group_partial, (size_t)cn::kEpilogueGroups * (CN_SUMMARY_FIELDS + 1))) ||
        (rc = dev_alloc(e, &S.tickets, (size_t)(cn::kEpilogueGroups + 1) * cn::kTicketStride)) ||
        (rc = dev_alloc(e, &e->io_dev, (size_t)1)) || (rc = dev_alloc(e, &e->C.error, (size_t)1)) ||
        (rc = dev_alloc(e, &e->S_dev, (size_t)1))) {
        cn_destroy(e);
        return rc;
    }
    e->S.error = e->C.error;
    if (hipMemcpy(e->S_dev, &e->S, sizeof(cn::StateView), hipMemcpyHostToDevice) != hipSuccess) {
        cn_destroy(e);
        return fail(CN_ERR_HIP, "cn_create: state view upload failed");
    }
    if (e->async_fill) {
        bool ok = hipEventCreateWithFlags(&e->rollout_done, hipEventDisableTiming) == hipSuccess;
        for (int i = 0; ok && i < cn_engine::kFillStreams; ++i)
            ok = hipStreamCreateWithFlags(&e->fill_streams[i], hipStreamNonBlocking) == hipSuccess;
        if (!ok) {
            cn_destroy(e);
            return fail(CN_ERR_HIP, "cn_create: side streams for the asynchronous scenario fill");
        }
    }
    e->discount = nullptr;
    e->discount_len = 0;
    *out = e;
    rc = cn_set_gamma(e, 0.9);
    if (rc) {
        cn_destroy(e);
        *out = nullptr;
    }
    return rc;
}

int cn_destroy(cn_engine* e) {
    if (!e) return CN_OK;
    (void)hipSetDevice(e->cfg.device);
    (void)hipStreamSynchronize(e->stream);
    for (int i = 0; i < cn_engine::kFillStreams; ++i)
        if (e->fill_streams[i]) {
            (void)hipStreamSynchronize(e->fill_streams[i]);
            (void)hipStreamDestroy(e->fill_streams[i]);
        }
    if (e->rollout_done) (void)hipEventDestroy(e->rollout_done);
    e->alloc_rollback(cn_engine::AllocMark{});
    if (e->discount) (void)hipFree(e->discount);
    cn_sarl_release(e);
    delete e;
    return CN_OK;
}

int cn_set_stream(cn_engine* e, void* hip_stream) {
    if (!e) return fail(CN_ERR_INVALID, "engine is NULL");
    e->stream = static_cast<hipStream_t>(hip_stream);
    e->orca_fresh = false;
    return CN_OK;
}

int cn_sync(cn_engine* e) {
    int rc = bind(e);
    if (rc) return rc;
    CN_HIP(hipStreamSynchronize(e->stream));
    for (int i = 0; i < cn_engine::kFillStreams; ++i)
        if (e->fill_streams[i]) CN_HIP(hipStreamSynchronize(e->fill_streams[i]));
    int gen_error = 0;
    CN_HIP(hipMemcpy(&gen_error, e->C.error, sizeof(int), hipMemcpyDeviceToHost));
    if (gen_error) {
        CN_HIP(hipMemset(e->C.error, 0, sizeof(int)));
        if (gen_error & 4)
            return fail(CN_ERR_HIP,
                        "the shard kernel's dynamic schedule: a workgroup waited ~2 s for an env's previous visit and went on "
                        "without it (results of this rollout are not to be trusted); CROWDNAV_AMD_SCHED_DYNAMIC=0 selects the "
                        "static schedule");
        if (gen_error & 2)
            return fail(CN_ERR_INVALID,
                        "cn_sarl_explore needs each env's numpy stream: (re)start the episodes with cn_reset (the scenario "
                        "ring of cn_rollout_* and the wave generator used for more than 8 humans do not keep it)");
        return fail(CN_ERR_INVALID,
                    "scenario generation gave up on a human after %llu rejected placements (the reference's rejection "
                    "sampling would not have terminated either: too many humans for this circle/square); the affected "
                    "scenario holds an overlapping placement",
                    e->C.max_attempts);
    }
    return CN_OK;
}

namespace {

__global__ void pack_state_kernel(int n, const double* state8, cn::StateView S) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* s = state8 + (size_t)i * 8;
    S.pos[i] = make_double2(s[0], s[1]);
    S.vel[i] = make_double2(s[2], s[3]);
    S.goal[i] = make_double2(s[4], s[5]);
    S.rv[i] = make_double2(s[6], s[7]);
}

__global__ void unpack_state_kernel(int n, double* state8, cn::StateView S) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double* s = state8 + (size_t)i * 8;
    const double2 p = S.pos[i], v = S.vel[i], g = S.goal[i], q = S.rv[i];
    s[0] = p.x, s[1] = p.y, s[2] = v.x, s[3] = v.y, s[4] = g.x, s[5] = g.y, s[6] = q.x, s[7] = q.y;
}

}  // namespace

int cn_set_state(cn_engine* e, const double* state8, const double* global_time) {
    int rc = bind(e);
    if (rc) return rc;
    const int n = e->P.B * e->P.A;
    if (state8) hipLaunchKernelGGL(pack_state_kernel, dim3((n + 255) / 256), dim3(256), 0, e->stream, n, state8, e->S);
    if (global_time)
        CN_HIP(hipMemcpyAsync(e->S.gtime, global_time, sizeof(double) * e->P.B, hipMemcpyDeviceToDevice, e->stream));
    CN_HIP(hipGetLastError());
    return CN_OK;
}

int cn_get_state(cn_engine* e, double* state8, double* global_time) {
    int rc = bind(e);
    if (rc) return rc;
    const int n = e->P.B * e->P.A;
    if (state8) hipLaunchKernelGGL(unpack_state_kernel, dim3((n + 255) / 256), dim3(256), 0, e->stream, n, state8, e->S);
    if (global_time)
        CN_HIP(hipMemcpyAsync(global_time, e->S.gtime, sizeof(double) * e->P.B, hipMemcpyDeviceToDevice, e->stream));
    CN_HIP(hipGetLastError());
    return CN_OK;
}

int cn_set_theta(cn_engine* e, const double* theta) {
    int rc = bind(e);
    if (rc) return rc;
    if (!theta) return fail(CN_ERR_INVALID, "cn_set_theta: NULL");
    CN_HIP(hipMemcpyAsync(e->S.theta, theta, sizeof(double) * e->P.B, hipMemcpyDeviceToDevice, e->stream));
    return CN_OK;
}

int cn_get_theta(cn_engine* e, double* theta) {
    int rc = bind(e);
    if (rc) return rc;
    if (!theta) return fail(CN_ERR_INVALID, "cn_get_theta: NULL");
    CN_HIP(hipMemcpyAsync(theta, e->S.theta, sizeof(double) * e->P.B, hipMemcpyDeviceToDevice, e->stream));
    return CN_OK;
}

namespace {
__global__ void human_count_kernel(int B, int A, const double2* pos, int32_t* out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int n = 0;
    for (int i = 1; i < A; ++i) n += cn::is_parked(pos[(size_t)b * A + i]) ? 0 : 1;
    out[b] = n;
}
}  // namespace

int cn_get_human_count(cn_engine* e, int32_t* count) {
    int rc = bind(e);
    if (rc) return rc;
    if (!count) return fail(CN_ERR_INVALID, "cn_get_human_count: NULL");
    hipLaunchKernelGGL(human_count_kernel, dim3((e->P.B + 255) / 256), dim3(256), 0, e->stream, e->P.B, e->P.A, e->S.pos, count);
    CN_HIP(hipGetLastError());
    return CN_OK;
}

namespace {
__global__ void set_robot_sim_kernel(int B, int A, const float* radii, float max_speed, cn::StateView S) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * A) return;
    S.rsim_radius[i] = radii[i % A];
    if (i % A == 0) {
        S.rsim_max_speed[i / A] = max_speed;
        S.rsim_valid[i / A] = 1;
    }
}
}  // namespace

int cn_set_robot_sim(cn_engine* e, const float* radii_host, float max_speed) {
    int rc = bind(e);
    if (rc) return rc;
    if (!radii_host) return fail(CN_ERR_INVALID, "cn_set_robot_sim: NULL");
    float* staged = reinterpret_cast<float*>(e->probe_key);  // 624 words of scratch, A <= 64
    CN_HIP(hipMemcpyAsync(staged, radii_host, sizeof(float) * e->P.A, hipMemcpyHostToDevice, e->stream));
    CN_HIP(hipStreamSynchronize(e->stream));  // radii_host may be a temporary
    hipLaunchKernelGGL(set_robot_sim_kernel, dim3((e->P.B * e->P.A + 255) / 256), dim3(256), 0, e->stream, e->P.B, e->P.A,
                       staged, max_speed, e->S);
    CN_HIP(hipGetLastError());
    return CN_OK;
}

int cn_drop_robot_sim(cn_engine* e) {
    int rc = bind(e);
    if (rc) return rc;
    CN_HIP(hipMemsetAsync(e->S.rsim_valid, 0, (size_t)e->P.B, e->stream));
    if (e->P.kd)  // ... and its simulator's kd-tree order: byte 0 of every env's A flags
        CN_HIP(hipMemset2DAsync(e->S.kd_valid, (size_t)e->P.A, 0, 1, (size_t)e->P.B, e->stream));
    return CN_OK;
}

int cn_drop_sims(cn_engine* e) {
    int rc = cn_drop_robot_sim(e);
    if (rc) return rc;
    if (e->P.kd) CN_HIP(hipMemsetAsync(e->S.kd_valid, 0, (size_t)e->P.B * e->P.A, e->stream));
    return CN_OK;
}

int cn_reset(cn_engine* e, const uint32_t* seeds, const uint8_t* mask, uint64_t* draws) {
    int rc = bind(e);
    if (rc) return rc;
    if (!seeds) return fail(CN_ERR_INVALID, "cn_reset: seeds is NULL");
    if (e->gen_wave)
        hipLaunchKernelGGL(cn::reset_wave_kernel, dim3(e->P.B), dim3(cn::kWave), 0, e->stream, e->P, e->C, e->S, seeds, mask,
                           draws);
    else
        hipLaunchKernelGGL(cn::reset_kernel, dim3(grid_lanes(e)), dim3(cn::kWave), 0, e->stream, e->P, e->C, e->S, seeds, mask,
                           draws);
    CN_HIP(hipGetLastError());
    return CN_OK;
}

int cn_orca(cn_engine* e, float* out_vel) {
    int rc = bind(e);
    if (rc) return rc;
    if (!out_vel) return fail(CN_ERR_INVALID, "cn_orca: out_vel is NULL");
    cn_launch_orca(e, out_vel);
    CN_HIP(hipGetLastError());
    return CN_OK;
}

int cn_step(cn_engine* e, const double* action, int update, double* reward, uint8_t* done, uint8_t* info,
            double* dmin, double* action_out, float* orca_vel, double* obs) {
    int rc = bind(e);
    if (rc) return rc;
    if (!reward || !done || !info) return fail(CN_ERR_INVALID, "cn_step: reward/done/info must not be NULL");
    if (e->P.robot_orca && action)
        return fail(CN_ERR_INVALID, "cn_step: action must be NULL when robot_policy == CN_ROBOT_ORCA");
    if (!e->P.robot_orca && !action)
        return fail(CN_ERR_INVALID, "cn_step: action is required when robot_policy == CN_ROBOT_EXTERNAL");
    cn::StepIo io{action, reward, done, info, dmin, action_out, orca_vel, obs, update ? 1 : 0};
    CN_LAUNCH_MAXL_UNI(e, step_kernel, grid_envs(e), e->P, e->S, io);
    CN_HIP(hipGetLastError());
    return CN_OK;
}

int cn_set_gamma(cn_engine* e, double gamma) {
    int rc = bind(e);
    if (rc) return rc;
    const int len = (int)std::ceil(e->cfg.time_limit / e->cfg.time_step) + 8;
    if (len > cn::kMaxDiscount)
        return fail(CN_ERR_UNSUPPORTED, "time_limit / time_step = %d steps per episode exceeds the %d the rollout kernel tabulates",
                    len - 8, cn::kMaxDiscount - 8);
    std::vector<double> table((size_t)len);
    for (int t = 0; t < len; ++t) table[(size_t)t] = std::pow(gamma, t * e->cfg.time_step * e->cfg.robot_v_pref);
    CN_HIP(hipStreamSynchronize(e->stream));
    if (e->discount) CN_HIP(hipFree(e->discount));
    e->discount = nullptr;
    CN_HIP(hipMalloc(reinterpret_cast<void**>(&e->discount), sizeof(double) * len));
    CN_HIP(hipMemcpy(e->discount, table.data(), sizeof(double) * len, hipMemcpyHostToDevice));
    e->discount_len = len;
    return CN_OK;
}

// CN_FLAG_ASYNC_SCENARIO_FILL: fill kernels still in flight on the side streams read the device copy of the io block, the
// caller's buffers behind it and the claim / ready flags.  Before any of those change hands (a new io block, a restart of the
// bookkeeping) the host waits for them: a straggler of the previous rollout must not publish a scenario of the old seed
// numbering into the new one's ring, nor read buffers the caller is about to free.
static int drain_fill_streams(cn_engine* e) {
    if (!e->async_fill) return CN_OK;
    for (int i = 0; i < cn_engine::kFillStreams; ++i)
        if (e->fill_streams[i]) CN_HIP(hipStreamSynchronize(e->fill_streams[i]));
    return CN_OK;
}

// Upload the caller's io struct (ordered on the engine's stream) if it differs from the device copy.
static int upload_io(cn_engine* e, const cn_rollout_io* io) {
    if (e->io_valid && std::memcmp(&e->io_host, io, sizeof(*io)) == 0) return CN_OK;
    // the seed numbering belongs to cn_rollout_begin: the scenario cache was sized by seed_mod and filled for seed_base (a
    // larger seed_mod would index past it, another seed_base would be served the old seeds' scenarios)
    if (e->rollout_begun && (io->seed_base != e->begin_seed_base || io->seed_mod != e->begin_seed_mod))
        return fail(CN_ERR_INVALID, "rollout io: seed_base / seed_mod (%u / %u) differ from cn_rollout_begin's (%u / %u); call "
                    "cn_rollout_begin to renumber the episodes", io->seed_base, io->seed_mod, e->begin_seed_base, e->begin_seed_mod);
    int rc = drain_fill_streams(e);
    if (rc) return rc;
    e->io_host = *io;
    CN_HIP(hipMemcpyAsync(e->io_dev, &e->io_host, sizeof(*io), hipMemcpyHostToDevice, e->stream));
    e->io_valid = true;
    return CN_OK;
}

static int check_io(const cn_engine* e, const cn_rollout_io* io) {
    (void)e;
    if (!io) return fail(CN_ERR_INVALID, "rollout io is NULL");
    if (io->seed_mod == 0) return fail(CN_ERR_INVALID, "seed_mod must be >= 1");
    if (!io->ep_count || !io->cur_steps || !io->cur_return || !io->active)
        return fail(CN_ERR_INVALID, "rollout io: ep_count, cur_steps, cur_return and active are required");
    if (io->record_capacity < 0) return fail(CN_ERR_INVALID, "record_capacity must be >= 0");
    if (io->blocks && io->blocks_records < 1) return fail(CN_ERR_INVALID, "rollout io: blocks needs blocks_records >= 1");
    if (io->env_offset < 0 || io->env_stride < io->env_offset + e->P.B)
        return fail(CN_ERR_INVALID, "rollout io: need env_offset >= 0 and env_stride >= env_offset + num_envs");
    return CN_OK;
}

int cn_rollout_begin(cn_engine* e, const cn_rollout_io* io) {
    int rc = bind(e);
    if (rc) return rc;
    if ((rc = check_io(e, io)) || (rc = drain_fill_streams(e))) return rc;
    e->rollout_begun = false;
    if ((rc = upload_io(e, io))) return rc;
    e->rollout_begun = true, e->begin_seed_base = io->seed_base, e->begin_seed_mod = io->seed_mod;
    cn::RolloutView R{e->io_dev, e->discount, e->discount_len};
    // scenario cache (step_kernels.h: cached_scenario_wave): on for the wave generators when the episode seeds come from a small
    // set; a new rollout may number its seeds differently, so it starts empty
    e->S.cache_n = (e->gen_wave && e->scenario_cache && io->seed_mod <= (uint32_t)cn::kScenarioCacheMax) ? (int)io->seed_mod : 0;
    if (e->gen_wave) CN_HIP(hipMemsetAsync(e->S.cache_state, 0, sizeof(int) * cn::kScenarioCacheMax, e->stream));
    if (e->gen_wave)
        hipLaunchKernelGGL(cn::rollout_begin_wave_kernel, dim3(e->P.B), dim3(cn::kWave), 0, e->stream, e->P, e->C, e->S, R);
    else
        hipLaunchKernelGGL(cn::rollout_begin_kernel, dim3(grid_lanes(e)), dim3(cn::kWave), 0, e->stream, e->P, e->C, e->S, R);
    CN_HIP(hipGetLastError());
    e->steps_since_fill = -1;
    return CN_OK;
}

// Top the scenario ring up to ring_depth episodes ahead of every env.  An env consumes at most one ring scenario per
// transition, so after a fill the next ring_depth transitions cannot run it dry: launches inside that budget skip the
// fill kernels altogether (a 20-step cn_rollout call used to spend more time here than in its transitions).
static int fill_ring_if_needed(cn_engine* e, const cn::RolloutView& R, int n_steps) {
    if (e->async_fill) {
        // a fill launch per transition launch, on the next side stream, ordered after the PREVIOUS transition kernel (whose
        // episode counters it reads) and beside the one about to be launched; nobody waits for it
        const int this_stream = e->next_fill_stream;
        hipStream_t fs = e->fill_streams[this_stream];
        e->next_fill_stream = (e->next_fill_stream + 1) % cn_engine::kFillStreams;
        CN_HIP(hipEventRecord(e->rollout_done, e->stream));
        CN_HIP(hipStreamWaitEvent(fs, e->rollout_done, 0));
        // (tried and dropped: 4 workgroups per env walking its slots in order — 4.0 M instead of 20.8 M env-steps/s at the
        // reference geometry, a workgroup stuck on a hard scenario delays its env's later slots.  Round 2's claim-then-work-list
        // pair of kernels "hung on the GPU box"; round 6 found why — a generator loop of the shape `for (;;) { lane 0 pops with
        // atomicAdd; j = readfirstlane; if (j >= jobs) break; generate }` never terminates as compiled (the popping loop below is
        // written around a ballot instead) — and the pair is now the default: CROWDNAV_AMD_FILL_QUEUE_WGS generator workgroups)
        if (e->fill_queue_wgs > 0) {  // scan + persistent generator workgroups on a job list (step_kernels.h); the default
            int* list = e->fill_list + (size_t)this_stream * (2 + 2 * (size_t)e->P.B * e->P.ring_depth);
            CN_HIP(hipMemsetAsync(list, 0, 2 * sizeof(int), fs));
            const int items = e->P.B * e->P.ring_depth;
            hipLaunchKernelGGL(cn::ring_fill_scan_kernel, dim3((items + 255) / 256), dim3(256), 0, fs, e->P, e->S, R, list);
            hipLaunchKernelGGL(cn::ring_fill_jobs_kernel, dim3(e->fill_queue_wgs), dim3(cn::kWave), 0, fs, e->P, e->C, e->S, R, list);
        } else {
            hipLaunchKernelGGL(cn::ring_fill_wave_async_kernel, dim3(e->P.B * e->P.ring_depth), dim3(cn::kWave), 0, fs, e->P, e->C,
                               e->S, R);
        }
        e->launch_counts[CN_COUNT_ASYNC_FILLS] += 1;
        e->steps_since_fill = 0;
        return CN_OK;
    }
    if (e->steps_since_fill >= 0 && e->steps_since_fill + n_steps <= e->P.ring_depth) {
        e->steps_since_fill += n_steps;
        return CN_OK;
    }
    const int fill_lanes = e->P.B * e->P.ring_depth;
    const dim3 fill_grid((fill_lanes + cn::kWave - 1) / cn::kWave);
    if (e->gen_wave) {
        hipLaunchKernelGGL(cn::ring_fill_wave_kernel, dim3(fill_lanes), dim3(cn::kWave), 0, e->stream, e->P, e->C, e->S, R);
    } else {
        CN_HIP(hipMemsetAsync(e->S.redo_count, 0, sizeof(int), e->stream));
        hipLaunchKernelGGL(cn::ring_fill_kernel, fill_grid, dim3(cn::kWave), 0, e->stream, e->P, e->C, e->S, R);
        hipLaunchKernelGGL(cn::ring_redo_kernel, dim3(cn::kRedoLanes / cn::kWave), dim3(cn::kWave), 0, e->stream, e->P, e->C,
                           e->S);
    }
    std::swap(e->S.ring_filled_in, e->S.ring_filled_out);
    e->launch_counts[CN_COUNT_RING_FILLS] += 1;
    e->steps_since_fill = n_steps;
    return CN_OK;
}

// The transitions themselves: the four-barrier fused kernel (rollout_fused.h) for the small-crowd geometries it covers,
// with BASELINE configs[1]'s geometry folded in as compile-time constants; the general phase kernel otherwise.
static void launch_rollout(cn_engine* e, const cn::RolloutView& R, int n_steps, const double* action) {
    const cn::Params& P = e->P;
    const uint64_t kernels_before = e->launch_counts[CN_COUNT_ROLLOUT_KERNELS];
    static const bool use_fused = env_int("CROWDNAV_AMD_FUSED", 1) != 0;
    static const bool use_geom20 = env_int("CROWDNAV_AMD_GEOM20", 1) != 0;  // the compile-time geometry of configs[3]'s shard
    const bool headline = P.A == 6 && P.NC == 5 && P.E == 2 && P.nA == 12 && P.pairs == 60 && P.threads == 64;
    // (small crowds the fused kernel does not take — unicycle robot, asynchronous fill, several waves per workgroup — run the
    // generic rollout_kernel<5, ..>; its own compile-time-geometry instantiation for configs[1] went when the fused kernel
    // became the headline path)
    // (the fused kernel reads the launch-time fill level only: never with the asynchronous fill, whose slots are published
    // one by one — CROWDNAV_AMD_WAVE_SCENARIOS=1 can switch that on for a small crowd)
    if (use_fused && !e->async_fill && e->maxl == 5 && !P.robot_unicycle && P.NC <= cn::kFusedMaxNC && P.pairs <= cn::kWave &&
        P.nA * 5 <= cn::kWave && P.threads == cn::kWave) {
        if (headline)
            hipLaunchKernelGGL((cn::rollout_fused_kernel<true>), dim3(grid_envs(e)), dim3(64), e->smem, e->stream, e->P,
                               (const cn::StateView*)e->S_dev, (const int*)e->S.ring_filled_in, R, n_steps, action);
        else
            hipLaunchKernelGGL((cn::rollout_fused_kernel<false>), dim3(grid_envs(e)), dim3(64), e->smem, e->stream, e->P,
                               (const cn::StateView*)e->S_dev, (const int*)e->S.ring_filled_in, R, n_steps, action);
    } else if (use_geom20 && e->maxl == 10 && !P.robot_unicycle && P.A == 21 && P.NC == 20 && P.E == 1 && P.threads == 64 &&
               P.orca.max_neighbors == 10 && P.kd) {
        const size_t smem20 = cn::smem_bytes_compact(P.nA, P.pairs, P.A, P.E);
        // Three resident waves per SIMD (step_kernels.h: kGeom20Waves): 3072 one-wave workgroups fill the chip, so B = 4096 envs would run as
        // a full round plus a third of one.  A call of 3 q + r steps becomes one launch of r steps over all envs (if r > 0) and
        // FOUR launches of q steps over 3 B / 4 workgroups each, sub-launch k leaving out env 3 - k of every group of four
        // (step_kernels.h: Params::sched): every env makes its steps in order, every launch is one round.
        // Worth it when it saves rounds: with S = 12 workgroups x CUs resident, a plain launch takes ceil(B / S) rounds of
        // n steps, the schedule 4 ceil(0.75 B / S) rounds of n / 3 (B = 4096 on 256 CUs: 2 vs 1.33; B = 3072: 1 vs 1.33 - plain).
        // CROWDNAV_AMD_SCHED_MIN_STEPS / CROWDNAV_AMD_SCHED_FORCE (read by cn_create): shortest call that is split; split
        // whatever the round count (the parity tests run the schedule on a handful of envs).
        cn::Params Pk = e->P;
        Pk.sched = -1;
        const int slots = e->sched_slots;
        // Dynamic schedule (step_kernels.h: kSchedDynamic): ONE launch of persistent workgroups taking (env, visit) items from a
        // device queue — wherever thirds of a call balance the chip better than whole calls (ceil(3 B / G) < 3 ceil(B / G)), and
        // always beside the asynchronous scenario fill, whose generator workgroups must find room WITHOUT sending a step workgroup
        // to a second round: the grid then leaves `sched_reserve` of the resident slots free (CROWDNAV_AMD_DYN_RESERVE).  Callers that
        // want the in-kernel summary / record blocks get the static 3-of-4 schedule below.
        {
            const bool use_dynamic = e->sched_dynamic;
            const int reserve = e->async_fill ? e->sched_reserve : 0;
            const int G = P.B < slots - reserve ? P.B : slots - reserve;
            // visits per env and call: ~56 steps each (measured at 4096 envs on the 12 m circle: 999-step calls 129 / 137 / 141 /
            // 143 / 143 / 136 / 110 M env-steps/s at 3 / 6 / 9 / 18 / 27 / 54 / 108 visits, 500-step calls 125 / 137 / 136 / 126 M
            // at 3 / 9 / 18 / 36 — shorter visits balance better until the per-visit prologue and the release / acquire of the
            // env's state show), at least three
            int visits = e->dyn_visits > 0 ? e->dyn_visits : (n_steps + 28) / 56;
            if (visits < 3) visits = 3;
            if (visits > n_steps) visits = n_steps;
            Pk.dyn_visits = visits;
            // work-conserving: worth it whenever the envs do not all fit at once (a plain launch then runs ceil(B / G) rounds)
            const bool helps = e->sched_force || e->async_fill || P.B > G;
            if (use_dynamic && G > 0 && n_steps >= e->sched_min && action == nullptr && helps && !e->io_host.summary &&
                !e->io_host.blocks) {
                (void)hipMemsetAsync(e->S.dyn_queue, 0, sizeof(int) * ((size_t)P.B + 1), e->stream);
                Pk.sched = cn::kSchedDynamic;
                e->launch_counts[CN_COUNT_ROLLOUT_KERNELS] += 1;
                e->launch_counts[CN_COUNT_SCHEDULED_KERNELS] += 1;
                hipLaunchKernelGGL((cn::rollout_kernel<10, false, true, true>), dim3(G), dim3(64), smem20, e->stream, Pk,
                                   (const cn::StateView*)e->S_dev, (const int*)e->S.ring_filled_in, R, n_steps, action);
                return;
            }
        }
        const int rounds_plain = (P.B + slots - 1) / slots, rounds_sched = (P.B / 4 * 3 + slots - 1) / slots;
        const bool sched = P.B % 4 == 0 && n_steps >= (e->sched_min > 48 ? e->sched_min : 48) && action == nullptr &&
                           (e->sched_force || 4 * rounds_sched < 3 * rounds_plain);
        const int q = sched ? n_steps / 3 : 0, rest = n_steps - 3 * q;
        e->launch_counts[CN_COUNT_ROLLOUT_KERNELS] += (rest > 0 ? 1 : 0) + (q > 0 ? 4 : 0);
        e->launch_counts[CN_COUNT_SCHEDULED_KERNELS] += q > 0 ? 4 : 0;
        if (rest > 0)
            hipLaunchKernelGGL((cn::rollout_kernel<10, false, true, true>), dim3(grid_envs(e)), dim3(64), smem20, e->stream, Pk,
                               (const cn::StateView*)e->S_dev, (const int*)e->S.ring_filled_in, R, rest, action);
        for (int k = 0; k < 4 && q > 0; ++k) {
            Pk.sched = k;
            hipLaunchKernelGGL((cn::rollout_kernel<10, false, true, true>), dim3(P.B / 4 * 3), dim3(64), smem20, e->stream, Pk,
                               (const cn::StateView*)e->S_dev, (const int*)e->S.ring_filled_in, R, q, action);
        }
    } else {
        CN_LAUNCH_ROLLOUT(e, grid_envs(e), e->P, (const cn::StateView*)e->S_dev, (const int*)e->S.ring_filled_in, R, n_steps, action);
    }
    if (e->launch_counts[CN_COUNT_ROLLOUT_KERNELS] == kernels_before) e->launch_counts[CN_COUNT_ROLLOUT_KERNELS] += 1;
}

int cn_rollout(cn_engine* e, const cn_rollout_io* io, int n_steps) {
    int rc = bind(e);
    if (rc) return rc;
    if ((rc = check_io(e, io))) return rc;
    if (n_steps < 0) return fail(CN_ERR_INVALID, "n_steps must be >= 0");
    if (n_steps == 0) return CN_OK;
    if (!e->P.robot_orca)
        return fail(CN_ERR_UNSUPPORTED, "cn_rollout needs an on-device robot policy (robot_policy == CN_ROBOT_ORCA); with "
                                        "CN_ROBOT_EXTERNAL use cn_rollout_step(action)");
    if ((rc = upload_io(e, io))) return rc;
    cn::RolloutView R{e->io_dev, e->discount, e->discount_len};
    if ((rc = fill_ring_if_needed(e, R, n_steps))) return rc;  // then the fused transitions
    launch_rollout(e, R, n_steps, nullptr);
    CN_HIP(hipGetLastError());
    return CN_OK;
}

int cn_rollout_step(cn_engine* e, const cn_rollout_io* io, const double* action) {
    int rc = bind(e);
    if (rc) return rc;
    if ((rc = check_io(e, io))) return rc;
    if (e->P.robot_orca) return fail(CN_ERR_INVALID, "cn_rollout_step is for CN_ROBOT_EXTERNAL engines (use cn_rollout)");
    if (!action) return fail(CN_ERR_INVALID, "cn_rollout_step: action is NULL");
    if ((rc = upload_io(e, io))) return rc;
    cn::RolloutView R{e->io_dev, e->discount, e->discount_len};
    if ((rc = fill_ring_if_needed(e, R, 1))) return rc;
    launch_rollout(e, R, 1, action);
    CN_HIP(hipGetLastError());
    return CN_OK;
}

int cn_launch_counts(cn_engine* e, uint64_t* counts_host) {
    if (!e || !counts_host) return fail(CN_ERR_INVALID, "cn_launch_counts: NULL argument");
    for (int i = 0; i < CN_LAUNCH_COUNTERS; ++i) counts_host[i] = e->launch_counts[i];
    return CN_OK;
}

int cn_rollout_records(cn_engine* e, const cn_rollout_io* io, int max_records, double* blocks) {
    int rc = bind(e);
    if (rc) return rc;
    if ((rc = check_io(e, io))) return rc;
    if (max_records < 1 || !blocks) return fail(CN_ERR_INVALID, "cn_rollout_records: need max_records >= 1 and blocks");
    if ((rc = upload_io(e, io))) return rc;
    const int n = e->P.B * max_records;
    hipLaunchKernelGGL(cn::records_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, e->stream, e->P.B, max_records, e->io_dev,
                       blocks);
    CN_HIP(hipGetLastError());
    return CN_OK;
}

int cn_records_summary(cn_engine* e, int64_t n_envs, int max_records, int record_capacity, const double* blocks,
                       double* summary) {
    int rc = bind(e);
    if (rc) return rc;
    if (n_envs < 0 || max_records < 1 || record_capacity < 0 || !blocks || !summary)
        return fail(CN_ERR_INVALID, "cn_records_summary: bad arguments");
    const cn::BlockRecords src{blocks, cn::record_block_doubles(max_records)};
    if (n_envs * max_records <= cn::kSummarySmallItems)
        hipLaunchKernelGGL(cn::records_summary_small_kernel<cn::BlockRecords>, dim3(1), dim3(cn::kSummarySmallThreads), 0, e->stream,
                           n_envs, max_records, record_capacity, src, summary);
    else
        hipLaunchKernelGGL(cn::records_summary_kernel<cn::BlockRecords>, dim3(cn::kSummaryBlocks), dim3(cn::kSummaryThreads), 0,
                           e->stream, n_envs, max_records, record_capacity, src, summary, e->summary_scratch);
    CN_HIP(hipGetLastError());
    return CN_OK;
}

int cn_rollout_summary(cn_engine* e, const cn_rollout_io* io, double* summary) {
    int rc = bind(e);
    if (rc) return rc;
    if ((rc = check_io(e, io))) return rc;
    if (!summary) return fail(CN_ERR_INVALID, "cn_rollout_summary: summary is NULL");
    if (io->record_capacity < 1) return fail(CN_ERR_INVALID, "cn_rollout_summary: the rollout keeps no records (record_capacity 0)");
    if ((rc = upload_io(e, io))) return rc;
    if ((int64_t)e->P.B * io->record_capacity <= cn::kSummarySmallItems)
        hipLaunchKernelGGL(cn::records_summary_small_kernel<cn::RingRecords>, dim3(1), dim3(cn::kSummarySmallThreads), 0, e->stream,
                           (int64_t)e->P.B, io->record_capacity, io->record_capacity, cn::RingRecords{e->io_dev}, summary);
    else
        hipLaunchKernelGGL(cn::records_summary_kernel<cn::RingRecords>, dim3(cn::kSummaryBlocks), dim3(cn::kSummaryThreads), 0,
                           e->stream, (int64_t)e->P.B, io->record_capacity, io->record_capacity, cn::RingRecords{e->io_dev}, summary,
                           e->summary_scratch);
    CN_HIP(hipGetLastError());
    return CN_OK;
}

// RCCL is bound at first use (dlopen of librccl.so.1: in a PyTorch-ROCm process that is the copy torch already loaded),
// so the library itself does not depend on it.
namespace {
struct RcclApi {
    int (*all_gather)(const void*, void*, size_t, int, void*, hipStream_t);
    const char* (*error_string)(int);
    bool ok;
};
const RcclApi* rccl_api() {
    static RcclApi api = [] {
        RcclApi a{};
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return a;
        a.all_gather = reinterpret_cast<int (*)(const void*, void*, size_t, int, void*, hipStream_t)>(dlsym(h, "ncclAllGather"));
        a.error_string = reinterpret_cast<const char* (*)(int)>(dlsym(h, "ncclGetErrorString"));
        a.ok = a.all_gather && a.error_string;
        return a;
    }();
    return &api;
}
constexpr int kNcclFloat64 = 8;  // ncclDataType_t (rccl.h)
}  // namespace

int cn_gather_records(cn_engine* e, void* rccl_comm, int n_ranks, int max_records, const double* blocks,
                      double* blocks_all) {
    int rc = bind(e);
    if (rc) return rc;
    if (!rccl_comm || n_ranks < 1 || max_records < 1 || !blocks || !blocks_all)
        return fail(CN_ERR_INVALID, "cn_gather_records: bad arguments");
    const RcclApi* api = rccl_api();
    if (!api->ok) return fail(CN_ERR_UNSUPPORTED, "cn_gather_records: librccl.so.1 could not be loaded");
    const size_t n = (size_t)e->P.B * cn::record_block_doubles(max_records);
    const int st = api->all_gather(blocks, blocks_all, n, kNcclFloat64, rccl_comm, e->stream);
    if (st) return fail(CN_ERR_HIP, "cn_gather_records: RCCL error %d (%s)", st, api->error_string(st));
    return CN_OK;
}

int cn_mt_random(cn_engine* e, uint32_t seed, int n, double* out) {
    int rc = bind(e);
    if (rc) return rc;
    if (n < 0 || !out) return fail(CN_ERR_INVALID, "cn_mt_random: bad arguments");
    hipLaunchKernelGGL(cn::mt_probe_kernel, dim3(1), dim3(1), 0, e->stream, e->probe_key, seed, n, out);
    CN_HIP(hipGetLastError());
    return CN_OK;
}

}  // extern "C"

#ifdef CN_PHASE_TIMING
// profiling builds only (scripts/phase_probe.py): accumulated shader-clock cycles per rollout phase, [8] = waves
extern "C" int cn_debug_phase_cycles(unsigned long long* out16, int reset) {
    if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(cn::cn_phase_cycles), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long zero[16] = {};
        zero[13] = ~0ull;  // the fastest wave's total (atomicMin)
        if (hipMemcpyToSymbol(HIP_SYMBOL(cn::cn_phase_cycles), zero, sizeof(zero)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
#ifdef CN_PHASE_TIMING
extern "C" int cn_debug_lazy_counts(unsigned long long* out8, int reset) {
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(cn::cn_lazy_counts), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long zero[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(cn::cn_lazy_counts), zero, sizeof(zero)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
#ifdef CN_WAVE_TRACE
extern "C" int cn_debug_wave_trace(unsigned long long* out, int waves) {
    if (waves > 8192) waves = 8192;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(cn::cn_wave_trace), (size_t)waves * 6 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
