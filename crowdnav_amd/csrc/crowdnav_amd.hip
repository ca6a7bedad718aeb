// libcrowdnav_amd.so — HIP kernels (gfx950) + the C ABI declared in include/crowdnav_amd.h.
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
// Kernel design: one wave64 workgroup owns floor(64 / A) whole envs, one lane per
// (env, agent).  Agent state is double2 SoA planes in HBM (pos, vel, goal, {radius, v_pref}); a step stages the
// float32 view of it in LDS, every lane solves its own ORCA program, human lanes do the float64 swept
// collision test against the robot, the robot lane reduces them to reward / done / info, and all lanes
// integrate.  The rollout kernel keeps the state in VGPRs for n_steps transitions and auto-resets finished
// envs in-kernel, so HBM sees one read and one write of the state per launch plus the episode records.
//
// Built with -ffp-contract=off: float32 ORCA and float64 env arithmetic must round exactly like the CPU
// reference; the only fused operation is the explicit fma in norm2().
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <limits>
#include <new>
#include <vector>

#include "../../include/crowdnav_amd.h"
#include "orca_device.h"
#include "scenario_device.h"

namespace cn {

struct Params {
    int B, A;  // envs, agents per env
    int envs_per_block;
    int robot_visible, robot_orca;
    double dt, time_limit, success_reward, collision_penalty, discomfort_dist, discomfort_factor;
    double robot_safety, human_safety;
    OrcaParams orca;
    ScenarioCfg scen;
};

struct StateView {
    double2* pos;
    double2* vel;
    double2* goal;
    double2* rv;  // (radius, v_pref)
    double* gtime;
    float* rsim_radius;  // [B*A] radii captured by the robot's persistent ORCA policy (orca.py:98-104)
    float* rsim_max_speed;  // [B]
    uint8_t* rsim_valid;    // [B]
    uint32_t* mt_key;       // [624][B]
    int* mt_pos;            // [B]
};

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

struct StepLds {
    OrcaLds orca;
    float4 kin[kWave];       // float32(px, py, vx, vy) as ORCA sees the agents (rvo2 boundary cast)
    double rad[kWave];       // float64 radius
    float robot_view_radius[kWave];  // radius + 0.01 + robot_safety as captured by the robot's policy
    double2 posd[kWave];     // float64 position (collision test)
    double2 veld[kWave];     // float64 velocity
    double2 act[kWave];      // per env (indexed by the robot's lane): applied robot action
    double closest[kWave];   // per human lane: closest boundary distance over the step
    int flag[kWave];         // per env (robot lane): done
};

struct Lane {
    int lane, env, a, ebase;  // ebase = lane of this env's robot
    bool valid;
    size_t gi;  // env * A + a
};

__device__ __forceinline__ Lane lane_of(const Params& P) {
    Lane L;
    L.lane = threadIdx.x;
    const int el = L.lane / P.A;
    L.a = L.lane - el * P.A;
    L.env = blockIdx.x * P.envs_per_block + el;
    L.valid = (el < P.envs_per_block) && (L.env < P.B);
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

// float32 staging of what every rvo2 simulator of this env is told (orca.py:100-110)
__device__ __forceinline__ void stage(StepLds& s, const Lane& L, const AgentRegs& r) {
    s.kin[L.lane] = make_float4((float)r.px, (float)r.py, (float)r.vx, (float)r.vy);
    s.rad[L.lane] = r.rad;
    s.posd[L.lane] = make_double2(r.px, r.py);
    s.veld[L.lane] = make_double2(r.vx, r.vy);
}

// The robot's ORCA policy object outlives episodes and keeps the radii / max speed it saw when its rvo2
// simulator was first built (orca.py:95-104).  Load or capture them.
__device__ __forceinline__ void load_robot_view(const Params& P, const StateView& S, StepLds& s,
                                                const Lane& L, const AgentRegs& r, float& robot_max_speed) {
    robot_max_speed = 0.0f;
    if (!L.valid) return;
    const bool have = S.rsim_valid[L.env] != 0;
    float rr;
    if (have) {
        rr = S.rsim_radius[L.gi];
    } else {
        rr = (float)(r.rad + 0.01 + P.robot_safety);
        S.rsim_radius[L.gi] = rr;
    }
    s.robot_view_radius[L.lane] = rr;
    if (L.a == 0) {
        if (have) {
            robot_max_speed = S.rsim_max_speed[L.env];
        } else {
            robot_max_speed = (float)r.vpref;
            S.rsim_max_speed[L.env] = robot_max_speed;
        }
    }
}

// ORCA.predict for this lane's agent (orca.py:82-132): neighbours = other humans in index order, then
// the robot if it is visible (crowd_sim.py:325-327); the robot itself sees every human.
__device__ inline void orca_predict(const Params& P, StepLds& s, const Lane& L, const AgentRegs& r,
                                    float robot_max_speed, float& out_vx, float& out_vy) {
    const int lane = L.lane;
    const bool is_robot = (L.a == 0);
    const double safety = is_robot ? P.robot_safety : P.human_safety;
    const float4 self = s.kin[lane];
    const float self_radius = is_robot ? s.robot_view_radius[lane] : (float)(r.rad + 0.01 + safety);
    const float max_speed = is_robot ? robot_max_speed : (float)r.vpref;

    // preferred velocity: towards the goal, unit length once farther than 1 m (orca.py:113-115)
    const double gdx = r.gx - r.px, gdy = r.gy - r.py;
    const double speed = norm2(gdx, gdy);
    const float pref_x = (float)(speed > 1.0 ? gdx / speed : gdx);
    const float pref_y = (float)(speed > 1.0 ? gdy / speed : gdy);

    // neighbour selection (Appendix A.2); a single kd-tree leaf for <= 10 agents means index order
    int count = 0;
    float range_sq = P.orca.neighbor_dist * P.orca.neighbor_dist;
    const int n_cand = is_robot ? P.A - 1 : (P.robot_visible ? P.A - 1 : P.A - 2);
    for (int c = 0; c < n_cand; ++c) {
        int j;  // agent index inside the env
        if (is_robot) {
            j = c + 1;
        } else {
            j = c + 1 + (c + 1 >= L.a ? 1 : 0);
            if (j >= P.A) j = 0;  // the visible robot comes last
        }
        const float4 o = s.kin[L.ebase + j];
        const float ddx = self.x - o.x, ddy = self.y - o.y;
        offer_neighbor(s.orca, lane, L.ebase + j, ddx * ddx + ddy * ddy, P.orca.max_neighbors, count, range_sq);
    }

    const Planes lines = planes_of(s.orca, 0, lane);
    for (int k = 0; k < count; ++k) {
        const int ol = s.orca.nb_lane[k][lane];
        const float4 o = s.kin[ol];
        const float other_radius = is_robot ? s.robot_view_radius[ol] : (float)(s.rad[ol] + 0.01 + safety);
        float lpx, lpy, ldx, ldy;
        make_half_plane(P.orca, self.x, self.y, self.z, self.w, o.x, o.y, o.z, o.w, self_radius + other_radius,
                        lpx, lpy, ldx, ldy);
        lines.px[k * kWave] = lpx;
        lines.py[k * kWave] = lpy;
        lines.dx[k * kWave] = ldx;
        lines.dy[k * kWave] = ldy;
    }

    float rx, ry;
    const int fail = lp_planar(lines, count, max_speed, pref_x, pref_y, false, rx, ry);
    if (fail < count) lp_relaxed(lines, planes_of(s.orca, 1, lane), count, fail, max_speed, rx, ry);
    out_vx = rx;
    out_vy = ry;
}

// crowd_sim/envs/utils/utils.py:4-26 with (x3, y3) = (0, 0)
__device__ __forceinline__ double point_to_segment_origin(double x1, double y1, double x2, double y2) {
    const double sx = x2 - x1, sy = y2 - y1;
    if (sx == 0.0 && sy == 0.0) return norm2(0.0 - x1, 0.0 - y1);
    double u = ((0.0 - x1) * sx + (0.0 - y1) * sy) / (sx * sx + sy * sy);
    if (u > 1.0) {
        u = 1.0;
    } else if (u < 0.0) {
        u = 0.0;
    }
    const double x = x1 + u * sx, y = y1 + u * sy;
    return norm2(x - 0.0, y - 0.0);
}

struct StepResult {  // meaningful on the robot lane
    double reward, dmin, ax, ay;
    uint8_t done, info;
};

// One transition for the lane's agent (crowd_sim.py:317-420).  `r` is updated in place when update != 0.
// new_vx/new_vy: the velocity this lane's agent chose (float32 for ORCA agents, the action for the robot).
// Must be called by all 64 lanes (contains workgroup barriers).
__device__ inline void step_core(const Params& P, StepLds& s, const Lane& L, AgentRegs& r, double& gtime,
                                 float robot_max_speed, const double* ext_action, int update,
                                 StepResult& res, double& new_vx, double& new_vy) {
    stage(s, L, r);
    __syncthreads();

    float ovx = 0.0f, ovy = 0.0f;
    if (L.valid && (L.a > 0 || P.robot_orca)) orca_predict(P, s, L, r, robot_max_speed, ovx, ovy);
    new_vx = ovx;
    new_vy = ovy;
    if (L.valid && L.a == 0) {
        if (!P.robot_orca) {
            new_vx = ext_action[2 * (size_t)L.env];
            new_vy = ext_action[2 * (size_t)L.env + 1];
        }
        s.act[L.lane] = make_double2(new_vx, new_vy);
    }
    __syncthreads();

    // swept robot-human collision over the step: human's CURRENT velocity vs the robot's NEW action
    // (crowd_sim.py:331-351)
    if (L.valid && L.a > 0) {
        const double2 rp = s.posd[L.ebase];
        const double2 act = s.act[L.ebase];
        const double rx = r.px - rp.x, ry = r.py - rp.y;
        const double wx = r.vx - act.x, wy = r.vy - act.y;
        const double ex = rx + wx * P.dt, ey = ry + wy * P.dt;
        s.closest[L.lane] = point_to_segment_origin(rx, ry, ex, ey) - r.rad - s.rad[L.ebase];
    }
    __syncthreads();

    res.done = 0;
    if (L.valid && L.a == 0) {
        double dmin = std::numeric_limits<double>::infinity();
        bool collision = false;
        for (int i = 1; i < P.A; ++i) {
            const double c = s.closest[L.lane + i];
            if (c < 0.0) {
                collision = true;
                break;
            } else if (c < dmin) {
                dmin = c;
            }
        }
        const double endx = r.px + new_vx * P.dt, endy = r.py + new_vy * P.dt;
        const bool reaching = norm2(endx - r.gx, endy - r.gy) < r.rad;
        if (gtime >= P.time_limit - 1.0) {
            res.reward = 0.0, res.done = 1, res.info = CN_TIMEOUT;
        } else if (collision) {
            res.reward = P.collision_penalty, res.done = 1, res.info = CN_COLLISION;
        } else if (reaching) {
            res.reward = P.success_reward, res.done = 1, res.info = CN_REACH_GOAL;
        } else if (dmin < P.discomfort_dist) {
            res.reward = (dmin - P.discomfort_dist) * P.discomfort_factor * P.dt;
            res.done = 0, res.info = CN_DANGER;
        } else {
            res.reward = 0.0, res.done = 0, res.info = CN_NOTHING;
        }
        res.dmin = dmin;
        res.ax = new_vx, res.ay = new_vy;
        if (update) gtime += P.dt;
    }
    if (update && L.valid) {  // Agent.step (agent.py:127-135)
        r.px = r.px + new_vx * P.dt;
        r.py = r.py + new_vy * P.dt;
        r.vx = new_vx;
        r.vy = new_vy;
    }
}

// ---------------------------------------------------------------------------------------------- kernels

__global__ __launch_bounds__(kWave) void orca_kernel(Params P, StateView S, float* out_vel) {
    __shared__ StepLds s;
    const Lane L = lane_of(P);
    AgentRegs r = {};
    if (L.valid) load_agent(S, L.gi, r);
    float robot_max_speed;
    load_robot_view(P, S, s, L, r, robot_max_speed);
    stage(s, L, r);
    __syncthreads();
    if (L.valid) {
        float vx, vy;
        orca_predict(P, s, L, r, robot_max_speed, vx, vy);
        out_vel[2 * L.gi] = vx;
        out_vel[2 * L.gi + 1] = vy;
        if (L.a == 0) S.rsim_valid[L.env] = 1;
    }
}

__global__ __launch_bounds__(kWave) void step_kernel(Params P, StateView S, StepIo io) {
    __shared__ StepLds s;
    const Lane L = lane_of(P);
    AgentRegs r = {};
    if (L.valid) load_agent(S, L.gi, r);
    float robot_max_speed = 0.0f;
    if (P.robot_orca) load_robot_view(P, S, s, L, r, robot_max_speed);
    double gtime = (L.valid && L.a == 0) ? S.gtime[L.env] : 0.0;
    const AgentRegs before = r;

    StepResult res;
    double nvx, nvy;
    step_core(P, s, L, r, gtime, robot_max_speed, io.action, io.update, res, nvx, nvy);
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
        if (io.update) S.gtime[L.env] = gtime;
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

// np.random.seed(seed) + scenario of one env per lane (lane = env)
__global__ __launch_bounds__(kWave) void reset_kernel(Params P, StateView S, const uint32_t* seeds,
                                                     const uint8_t* mask, uint64_t* draws) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= P.B) return;
    if (mask && !mask[b]) return;
    Mt19937 rng{S.mt_key + b, P.B, 0};
    const uint64_t n = generate_scenario(P.scen, rng, seeds[b], (size_t)b * P.A, S.pos, S.vel, S.goal, S.rv);
    S.mt_pos[b] = rng.pos;
    S.gtime[b] = 0.0;
    if (draws) draws[b] = n;
}

__global__ void mt_probe_kernel(uint32_t* key, uint32_t seed, int n, double* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Mt19937 rng{key, 1, 0};
    rng.seed(seed);
    for (int i = 0; i < n; ++i) out[i] = rng.random();
}

struct RolloutView {
    cn_rollout_io io;
    const double* discount;  // [discount_len]
    int discount_len;
};

__device__ __forceinline__ uint32_t episode_seed(const cn_rollout_io& io, int64_t c) {
    return io.seed_base + (uint32_t)((uint64_t)c % io.seed_mod);
}

// (re)start bookkeeping: env b begins its episode j = 0 (global id c = b)
__global__ __launch_bounds__(kWave) void rollout_begin_kernel(Params P, StateView S, RolloutView R) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= P.B) return;
    const cn_rollout_io& io = R.io;
    const int64_t c0 = io.env_offset + b;
    const bool on = io.episode_limit < 0 || c0 < io.episode_limit;
    io.active[b] = on ? 1 : 0;
    io.ep_count[b] = 0;
    io.cur_steps[b] = 0;
    io.cur_return[b] = 0.0;
    if (io.cur_danger) io.cur_danger[b] = 0;
    if (io.cur_danger_dmin_sum) io.cur_danger_dmin_sum[b] = 0.0;
    if (!on) return;
    Mt19937 rng{S.mt_key + b, P.B, 0};
    generate_scenario(P.scen, rng, episode_seed(io, c0), (size_t)b * P.A, S.pos, S.vel, S.goal, S.rv);
    S.mt_pos[b] = rng.pos;
    S.gtime[b] = 0.0;
}

// n_steps transitions per active env in one launch; state lives in VGPRs between steps.
__global__ __launch_bounds__(kWave) void rollout_kernel(Params P, StateView S, RolloutView R, int n_steps) {
    __shared__ StepLds s;
    const cn_rollout_io& io = R.io;
    const Lane L = lane_of(P);
    AgentRegs r = {};
    if (L.valid) load_agent(S, L.gi, r);
    float robot_max_speed = 0.0f;
    load_robot_view(P, S, s, L, r, robot_max_speed);

    const bool robot = L.valid && L.a == 0;
    double gtime = 0.0, cur_return = 0.0, cur_dsum = 0.0;
    int cur_steps = 0, cur_danger = 0, ep_count = 0;
    bool active = false;
    if (robot) {
        gtime = S.gtime[L.env];
        active = io.active[L.env] != 0;
        ep_count = io.ep_count[L.env];
        cur_steps = io.cur_steps[L.env];
        cur_return = io.cur_return[L.env];
        if (io.cur_danger) cur_danger = io.cur_danger[L.env];
        if (io.cur_danger_dmin_sum) cur_dsum = io.cur_danger_dmin_sum[L.env];
    }
    unsigned long long transitions = 0;

    for (int step = 0; step < n_steps; ++step) {
        if (robot) s.flag[L.lane] = active ? 1 : 0;
        __syncthreads();
        const bool env_on = L.valid && s.flag[L.ebase] != 0;
        Lane Ls = L;
        Ls.valid = env_on;
        __syncthreads();

        StepResult res;
        double nvx, nvy;
        step_core(P, s, Ls, r, gtime, robot_max_speed, nullptr, 1, res, nvx, nvy);

        bool reload = false;
        if (robot && active) {
            ++transitions;
            const double disc = cur_steps < R.discount_len ? R.discount[cur_steps] : 0.0;
            cur_return = cur_return + disc * res.reward;  // python sum(): left to right
            ++cur_steps;
            if (res.info == CN_DANGER) {
                ++cur_danger;
                cur_dsum += res.dmin;
            }
            if (res.done) {
                if (io.record_capacity > 0) {
                    const size_t k = (size_t)L.env * io.record_capacity + (ep_count % io.record_capacity);
                    if (io.ep_outcome) io.ep_outcome[k] = res.info;
                    if (io.ep_steps) io.ep_steps[k] = cur_steps;
                    if (io.ep_return) io.ep_return[k] = cur_return;
                    if (io.ep_time) io.ep_time[k] = (res.info == CN_TIMEOUT) ? P.time_limit : gtime;
                    if (io.ep_danger) io.ep_danger[k] = cur_danger;
                    if (io.ep_danger_dmin_sum) io.ep_danger_dmin_sum[k] = cur_dsum;
                }
                ++ep_count;
                cur_steps = 0, cur_return = 0.0, cur_danger = 0, cur_dsum = 0.0;
                const int64_t c = io.env_offset + L.env + (int64_t)ep_count * io.env_stride;
                if (io.episode_limit >= 0 && c >= io.episode_limit) {
                    active = false;
                } else {
                    Mt19937 rng{S.mt_key + L.env, P.B, 0};
                    generate_scenario(P.scen, rng, episode_seed(io, c), (size_t)L.env * P.A, S.pos, S.vel,
                                      S.goal, S.rv);
                    S.mt_pos[L.env] = rng.pos;
                    gtime = 0.0;
                    reload = true;
                }
            }
        }
        if (robot) s.flag[L.lane] = reload ? 1 : 0;
        __syncthreads();
        if (L.valid && s.flag[L.ebase] != 0) load_agent(S, L.gi, r);  // new scenario written by the robot lane
        __syncthreads();
    }

    if (L.valid) {
        S.pos[L.gi] = make_double2(r.px, r.py);
        S.vel[L.gi] = make_double2(r.vx, r.vy);
        S.goal[L.gi] = make_double2(r.gx, r.gy);
        S.rv[L.gi] = make_double2(r.rad, r.vpref);
    }
    if (robot) {
        S.gtime[L.env] = gtime;
        S.rsim_valid[L.env] = 1;
        io.active[L.env] = active ? 1 : 0;
        io.ep_count[L.env] = ep_count;
        io.cur_steps[L.env] = cur_steps;
        io.cur_return[L.env] = cur_return;
        if (io.cur_danger) io.cur_danger[L.env] = cur_danger;
        if (io.cur_danger_dmin_sum) io.cur_danger_dmin_sum[L.env] = cur_dsum;
        if (io.transitions && transitions) atomicAdd((unsigned long long*)io.transitions, transitions);
    }
}

}  // namespace cn

// ------------------------------------------------------------------------------------------------ C ABI

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define CN_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t err__ = (call);                                                                \
        if (err__ != hipSuccess)                                                                  \
            return fail(CN_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(err__), __FILE__, \
                        __LINE__);                                                                \
    } while (0)

}  // namespace

struct cn_engine {
    cn_config cfg;
    cn::Params P;
    cn::StateView S;
    hipStream_t stream;
    double* discount;
    int discount_len;
    uint32_t* probe_key;
    std::vector<void*> allocs;
};

namespace {

template <typename T>
int dev_alloc(cn_engine* e, T** out, size_t n) {
    void* p = nullptr;
    CN_HIP(hipMalloc(&p, n * sizeof(T)));
    CN_HIP(hipMemset(p, 0, n * sizeof(T)));
    e->allocs.push_back(p);
    *out = static_cast<T*>(p);
    return CN_OK;
}

int bind(cn_engine* e) {
    if (!e) return fail(CN_ERR_INVALID, "engine is NULL");
    CN_HIP(hipSetDevice(e->cfg.device));
    return CN_OK;
}

inline int grid_envs(const cn_engine* e) {
    return (e->P.B + e->P.envs_per_block - 1) / e->P.envs_per_block;
}
inline int grid_lanes(const cn_engine* e) { return (e->P.B + cn::kWave - 1) / cn::kWave; }

}  // namespace

extern "C" {

const char* cn_last_error(void) { return g_err; }
int cn_abi_version(void) { return 1; }

int cn_create(const cn_config* c, cn_engine** out) {
    if (!c || !out) return fail(CN_ERR_INVALID, "cn_create: NULL argument");
    *out = nullptr;
    if (c->num_envs < 1) return fail(CN_ERR_INVALID, "num_envs must be >= 1 (got %d)", c->num_envs);
    if (c->num_humans < 1 || c->num_humans > 63)
        return fail(CN_ERR_UNSUPPORTED, "num_humans must be in 1..63 (got %d)", c->num_humans);
    if (c->max_neighbors < 0 || c->max_neighbors > cn::kMaxNb)
        return fail(CN_ERR_UNSUPPORTED, "max_neighbors must be in 0..%d (got %d)", cn::kMaxNb, c->max_neighbors);
    if (!(c->time_step > 0.0)) return fail(CN_ERR_INVALID, "time_step must be > 0");
    if (c->scenario_rule != CN_CIRCLE_CROSSING && c->scenario_rule != CN_SQUARE_CROSSING)
        return fail(CN_ERR_UNSUPPORTED, "scenario_rule %d not supported", c->scenario_rule);
    if (c->robot_policy != CN_ROBOT_EXTERNAL && c->robot_policy != CN_ROBOT_ORCA)
        return fail(CN_ERR_INVALID, "robot_policy %d unknown", c->robot_policy);

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(CN_ERR_NO_DEVICE, "no HIP device visible: the MI355X engine has no CPU fallback");
    if (c->device < 0 || c->device >= ndev) return fail(CN_ERR_INVALID, "device %d out of range", c->device);
    CN_HIP(hipSetDevice(c->device));

    cn_engine* e = new (std::nothrow) cn_engine();
    if (!e) return fail(CN_ERR_INVALID, "out of host memory");
    e->cfg = *c;
    e->stream = nullptr;
    cn::Params& P = e->P;
    P.B = c->num_envs;
    P.A = c->num_humans + 1;
    P.envs_per_block = cn::kWave / P.A;
    P.robot_visible = c->robot_visible ? 1 : 0;
    P.robot_orca = c->robot_policy == CN_ROBOT_ORCA;
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
    P.scen.num_agents = P.A;
    P.scen.rule = c->scenario_rule;
    P.scen.randomize = c->randomize_attributes ? 1 : 0;
    P.scen.circle_radius = c->circle_radius;
    P.scen.square_width = c->square_width;
    P.scen.discomfort_dist = c->discomfort_dist;
    P.scen.human_radius = c->human_radius;
    P.scen.human_v_pref = c->human_v_pref;
    P.scen.robot_radius = c->robot_radius;
    P.scen.robot_v_pref = c->robot_v_pref;

    const size_t n = (size_t)P.B * P.A;
    int rc = CN_OK;
    cn::StateView& S = e->S;
    if ((rc = dev_alloc(e, &S.pos, n)) || (rc = dev_alloc(e, &S.vel, n)) || (rc = dev_alloc(e, &S.goal, n)) ||
        (rc = dev_alloc(e, &S.rv, n)) || (rc = dev_alloc(e, &S.gtime, (size_t)P.B)) ||
        (rc = dev_alloc(e, &S.rsim_radius, n)) || (rc = dev_alloc(e, &S.rsim_max_speed, (size_t)P.B)) ||
        (rc = dev_alloc(e, &S.rsim_valid, (size_t)P.B)) || (rc = dev_alloc(e, &S.mt_key, (size_t)624 * P.B)) ||
        (rc = dev_alloc(e, &S.mt_pos, (size_t)P.B)) || (rc = dev_alloc(e, &e->probe_key, (size_t)624))) {
        cn_destroy(e);
        return rc;
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
    for (void* p : e->allocs) (void)hipFree(p);
    if (e->discount) (void)hipFree(e->discount);
    delete e;
    return CN_OK;
}

int cn_set_stream(cn_engine* e, void* hip_stream) {
    if (!e) return fail(CN_ERR_INVALID, "engine is NULL");
    e->stream = static_cast<hipStream_t>(hip_stream);
    return CN_OK;
}

int cn_sync(cn_engine* e) {
    int rc = bind(e);
    if (rc) return rc;
    CN_HIP(hipStreamSynchronize(e->stream));
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

int cn_drop_robot_sim(cn_engine* e) {
    int rc = bind(e);
    if (rc) return rc;
    CN_HIP(hipMemsetAsync(e->S.rsim_valid, 0, (size_t)e->P.B, e->stream));
    return CN_OK;
}

int cn_reset(cn_engine* e, const uint32_t* seeds, const uint8_t* mask, uint64_t* draws) {
    int rc = bind(e);
    if (rc) return rc;
    if (!seeds) return fail(CN_ERR_INVALID, "cn_reset: seeds is NULL");
    hipLaunchKernelGGL(cn::reset_kernel, dim3(grid_lanes(e)), dim3(cn::kWave), 0, e->stream, e->P, e->S, seeds, mask,
                       draws);
    CN_HIP(hipGetLastError());
    return CN_OK;
}

int cn_orca(cn_engine* e, float* out_vel) {
    int rc = bind(e);
    if (rc) return rc;
    if (!out_vel) return fail(CN_ERR_INVALID, "cn_orca: out_vel is NULL");
    hipLaunchKernelGGL(cn::orca_kernel, dim3(grid_envs(e)), dim3(cn::kWave), 0, e->stream, e->P, e->S, out_vel);
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
    hipLaunchKernelGGL(cn::step_kernel, dim3(grid_envs(e)), dim3(cn::kWave), 0, e->stream, e->P, e->S, io);
    CN_HIP(hipGetLastError());
    return CN_OK;
}

int cn_set_gamma(cn_engine* e, double gamma) {
    int rc = bind(e);
    if (rc) return rc;
    const int len = (int)std::ceil(e->cfg.time_limit / e->cfg.time_step) + 8;
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

static int check_io(const cn_engine* e, const cn_rollout_io* io) {
    if (!io) return fail(CN_ERR_INVALID, "rollout io is NULL");
    if (!e->P.robot_orca)
        return fail(CN_ERR_UNSUPPORTED, "cn_rollout needs an on-device robot policy (robot_policy == CN_ROBOT_ORCA)");
    if (io->seed_mod == 0) return fail(CN_ERR_INVALID, "seed_mod must be >= 1");
    if (!io->ep_count || !io->cur_steps || !io->cur_return || !io->active)
        return fail(CN_ERR_INVALID, "rollout io: ep_count, cur_steps, cur_return and active are required");
    if (io->record_capacity < 0) return fail(CN_ERR_INVALID, "record_capacity must be >= 0");
    if (io->env_offset < 0 || io->env_stride < io->env_offset + e->P.B)
        return fail(CN_ERR_INVALID, "rollout io: need env_offset >= 0 and env_stride >= env_offset + num_envs");
    return CN_OK;
}

int cn_rollout_begin(cn_engine* e, const cn_rollout_io* io) {
    int rc = bind(e);
    if (rc) return rc;
    if ((rc = check_io(e, io))) return rc;
    cn::RolloutView R{*io, e->discount, e->discount_len};
    hipLaunchKernelGGL(cn::rollout_begin_kernel, dim3(grid_lanes(e)), dim3(cn::kWave), 0, e->stream, e->P, e->S, R);
    CN_HIP(hipGetLastError());
    return CN_OK;
}

int cn_rollout(cn_engine* e, const cn_rollout_io* io, int n_steps) {
    int rc = bind(e);
    if (rc) return rc;
    if ((rc = check_io(e, io))) return rc;
    if (n_steps < 0) return fail(CN_ERR_INVALID, "n_steps must be >= 0");
    if (n_steps == 0) return CN_OK;
    cn::RolloutView R{*io, e->discount, e->discount_len};
    hipLaunchKernelGGL(cn::rollout_kernel, dim3(grid_envs(e)), dim3(cn::kWave), 0, e->stream, e->P, e->S, R, n_steps);
    CN_HIP(hipGetLastError());
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
