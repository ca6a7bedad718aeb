// cn_sarl_sample_step for a few envs, second half: ONE kernel for the decision, the transition and the humans' ORCA velocities of
// the next decision —
//   arg-max of reward + gamma V over the env's actions (the values sarl_narrow_kernel's tiles stored), the robot already at its
//   goal, the previous step's episode ends folded into `alive`, the epsilon-greedy draw          multi_human_rl.py:11-63
//   step_kernel's body (step_kernels.h) with that action                                           crowd_sim.py:317-420
//   orca_kernel's body on the state the transition has just written: what the NEXT call's network kernel reads as the humans'
//   next velocities (cn_engine::orca_fresh), so that a streamed sampling loop is two launches per step
//   (occupancy maps) sarl_lookahead_kernel's body on those velocities: the next decision's map columns
// instead of three kernels (orca, last-workgroup decision inside the network kernel, step).  One-wave workgroups without kd
// bookkeeping only (up to 10 agents per simulator): an env's lanes are lanes of one wave, so its arg-max is folded with shuffles.
#pragma once
#include "sarl_kernels.h"

namespace cn {

template <int MAXL, bool UNI>
__global__ __launch_bounds__(kMaxBlock) void sarl_decide_step_kernel(Params P, StateView S, StepIo io, SarlCfg C, SarlDecide D,
                                                                     const double* actions, float* next_orca_vel) {
    const Smem s = carve<MAXL>(P);
    const Lane L = lane_of(P);
    AgentRegs r = {};
    if (L.valid) load_agent(S, L.gi, r);  // (requested in front of the decision's own loads: one round trip for both)
    // Everything else this kernel will need from global memory that does not depend on the decision is requested HERE, beside
    // the values: one wave runs the whole kernel, and every load it meets on the way is a round trip nothing hides (round 6) —
    //   the robot lane's flags and its position in the env's numpy stream, then the five generator words one
    //   np.random.random() call reads (key[i], key[i + 1], key[i + 2], key[i + 397], key[i + 398], indices mod 624);
    //   every human lane's ORCA velocity for this transition; the captured robot-simulator view of the ORCA pass behind it.
    const bool robot_lane = L.valid && L.a == 0;
    uint8_t pre_alive = 0, pre_done = 0;
    int pre_pos = -1;
    uint32_t pw[5] = {0u, 0u, 0u, 0u, 0u};
    int pi0 = 0, pi1 = 0, pi2 = 0;
    if (robot_lane) {
        pre_alive = D.alive[L.env];
        pre_done = D.done ? D.done[L.env] : 0;
        pre_pos = D.mt_pos[L.env];
        if (pre_pos >= 0) {
            const uint32_t* key = D.mt_key + L.env;
            const auto wrap = [](int i) { return i >= 624 ? i - 624 : i; };
            pi0 = pre_pos, pi1 = wrap(pre_pos + 1), pi2 = wrap(pre_pos + 2);
            pw[0] = key[(size_t)pi0 * C.B], pw[1] = key[(size_t)pi1 * C.B], pw[2] = key[(size_t)pi2 * C.B];
            pw[3] = key[(size_t)wrap(pre_pos + 397) * C.B], pw[4] = key[(size_t)wrap(pre_pos + 398) * C.B];
        }
    }
    float lane_vel[2] = {0.0f, 0.0f};
    if (next_orca_vel != nullptr && L.valid && L.a > 0) lane_vel[0] = next_orca_vel[2 * L.gi], lane_vel[1] = next_orca_vel[2 * L.gi + 1];
    uint8_t view_have = 0;
    float view_rr = 0.0f, view_ms = 0.0f;
    if (next_orca_vel != nullptr && L.valid) {
        view_have = S.rsim_valid[L.env];
        view_rr = S.rsim_radius[L.gi];
        if (L.a == 0) view_ms = S.rsim_max_speed[L.env];
    }
    // ---- the decision: arg-max of reward + gamma V over the env's actions (the largest value, the lowest index on ties = the
    // first strict maximum of the reference's loop; NaN and -inf never win)
    bool samples = false;  // (robot lanes) this env's episode is still running
    double robot_action[2] = {0.0, 0.0};  // (robot lanes) the action of this step
    {
        // Round 6: ALL 64 lanes of the (one-wave) workgroup fetch an env's K values — two independent loads per lane at 81
        // actions — and a shuffle butterfly folds (value, index) pairs under the reference's order (the largest value, the lowest
        // index on ties: a total order, so every lane ends with the same winner).  The env's own A lanes striding over its
        // actions were ceil(K / A) = 14 DEPENDENT global loads of values other XCDs had just written: half of this kernel.
        double best_v = -__builtin_inf();
        int best_i = -1;
        const int wl = (int)threadIdx.x & (kWave - 1);
        const int my_el = L.valid ? L.ebase / P.A : -1;
        for (int el = 0; el < P.E; ++el) {  // (uniform)
            const int env = (int)blockIdx.x * P.E + el;
            if (env >= C.B) break;
            const double* val = D.value + (size_t)env * C.n_actions;
            double bv = -__builtin_inf();
            int bi = -1;
            for (int a = wl; a < C.n_actions; a += 2 * kWave) {
                const int a2 = a + kWave;
                const double v1 = val[a];
                const double v2 = a2 < C.n_actions ? val[a2] : -__builtin_inf();
                if (v1 > bv) bv = v1, bi = a;
                if (v2 > bv) bv = v2, bi = a2;
            }
#pragma unroll
            for (int off = kWave / 2; off > 0; off >>= 1) {
                const double ov = __shfl_xor(bv, off);
                const int oi = __shfl_xor(bi, off);
                const bool take = oi >= 0 && (bi < 0 || ov > bv || (ov == bv && oi < bi));
                bv = take ? ov : bv;
                bi = take ? oi : bi;
            }
            if (el == my_el) best_v = bv, best_i = bi;
        }
        (void)best_v;
        if (L.valid && L.a == 0) {
            const int b = L.env;
            // sarl_pick_tail on the robot's own registers (the same doubles): a robot already at its goal stops (:22-23)
            const bool arrived = norm2(r.py - r.gy, r.px - r.gx) < r.rad;
            const int arg = arrived ? -1 : best_i;
            int picked = (arrived || arg < 0) ? (arrived ? -1 : -2) : arg;  // -2: every value was NaN / -inf (:57-58)
            const bool keep = pre_alive && !pre_done;  // (the previous call's flags: this call's are written below)
            D.alive[b] = keep ? 1 : 0;
            samples = keep;
            // sarl_explore_env on registers: best / action are stored once and the transition below takes the action from the
            // registers (written to memory and read back by the same lane they were two global round trips of this kernel);
            // np.random.random() from the five prefetched words (Mt19937::next32 twice: words i and i + 1 are regenerated and
            // stored, the position moves on by two), np.random.choice through the memory-backed generator behind it
            int act_i = arg;
            if (keep && picked != -1) {
                if (pre_pos < 0) {  // the env was not (re)started by cn_reset: there is no stream to continue
                    atomicOr(D.error, 2);
                } else {
                    uint32_t* key = D.mt_key + b;
                    const auto twist = [](uint32_t hi, uint32_t lo, uint32_t m) {
                        const uint32_t y = (hi & 0x80000000u) | (lo & 0x7fffffffu);
                        return m ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
                    };
                    const auto temper = [](uint32_t v) {
                        v ^= (v >> 11);
                        v ^= (v << 7) & 0x9d2c5680u;
                        v ^= (v << 15) & 0xefc60000u;
                        v ^= (v >> 18);
                        return v;
                    };
                    const uint32_t v0 = twist(pw[0], pw[1], pw[3]), v1 = twist(pw[1], pw[2], pw[4]);
                    key[(size_t)pi0 * C.B] = v0;
                    key[(size_t)pi1 * C.B] = v1;
                    const double probability = ((double)(temper(v0) >> 5) * 67108864.0 + (double)(temper(v1) >> 6)) / 9007199254740992.0;
                    int pos = pi2;
                    if (probability < D.epsilon) {
                        Mt19937 rng{key, C.B, pos};
                        uint32_t bits = (uint32_t)(C.n_actions - 1);
                        bits |= bits >> 1, bits |= bits >> 2, bits |= bits >> 4, bits |= bits >> 8, bits |= bits >> 16;
                        uint32_t k = 0;
                        if (C.n_actions > 1) do k = rng.next32() & bits; while (k > (uint32_t)(C.n_actions - 1));  // randint(0, 1) draws nothing
                        picked = (int)k, act_i = (int)k;
                        pos = rng.pos;
                    }
                    D.mt_pos[b] = pos;
                }
            }
            robot_action[0] = act_i >= 0 ? actions[2 * act_i] : 0.0;
            robot_action[1] = act_i >= 0 ? actions[2 * act_i + 1] : 0.0;
            D.best[b] = picked;
            D.action[2 * b] = robot_action[0];
            D.action[2 * b + 1] = robot_action[1];
        }
    }
    // every env of this (one-wave) workgroup has finished its episode: nothing to step — a caller that streams calls past an
    // episode's end pays an almost empty launch; the env's state, done flag and histories stay as its last step left them
    if (blockDim.x == kWave && __ballot(samples) == 0ull) return;
    // ---- the transition: step_kernel's body (the robot's lane reads the action it has just written)
    float robot_max_speed = 0.0f;
    build_pairs(P, s);
    double gtime = (L.valid && L.a == 0) ? S.gtime[L.env] : 0.0;
    double theta = (L.valid && L.a == 0) ? S.theta[L.env] : 0.0;
    StepResult res;
    double nvx, nvy;
    // (the humans' velocities for this transition are the ones the previous call — or cn_launch_orca — left for the decision's
    // lookahead: one ORCA pass per step, the one behind the transition, instead of two)
    step_core<MAXL, UNI, false>(P, s, L, r, gtime, robot_max_speed, io.action, io.update, res, nvx, nvy, &theta, nullptr, next_orca_vel,
                                robot_action, lane_vel);
    if (L.valid) {
        if (L.a == 0) {
            // (info LAST: reward / dmin / info — and the action further up — may live in pinned host memory, where a caller that
            // watches info arrive takes the step's other outputs as written)
            io.reward[L.env] = res.reward;
            if (io.dmin) io.dmin[L.env] = res.dmin;
            io.done[L.env] = res.done;
            // (an episode's LAST step: its outputs have been acknowledged before the end code goes out — a system-scope release,
            // once per episode; earlier steps are ordered by the kernel boundaries between them)
            if (res.done) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            io.info[L.env] = res.info;
            if (io.update) {
                S.gtime[L.env] = gtime;
                S.theta[L.env] = theta;
            }
        }
        if (io.update) {
            S.pos[L.gi] = make_double2(r.px, r.py);
            S.vel[L.gi] = make_double2(r.vx, r.vy);
        }
    }
    // ---- orca_kernel's body on the new state (r holds what was just stored)
    if (next_orca_vel == nullptr) return;
    block_sync(P);
    load_robot_view(P, S, s, L, r, robot_max_speed, true, view_have != 0, view_rr, view_ms);
    float vx, vy;
    orca_phases<MAXL, false>(P, s, L, r, robot_max_speed, L.valid, vx, vy);
    if (L.valid) {
        next_orca_vel[2 * L.gi] = vx;
        next_orca_vel[2 * L.gi + 1] = vy;
        if (L.a == 0) S.rsim_valid[L.env] = 1;
    }
    // ---- sarl_lookahead_kernel's body for the next decision (occupancy maps only): the humans' next observable states from the
    // velocities just computed — staged in LDS (posd / act are dead by now) — and the map each human sees among them, by the
    // human's own lane with the same functions, so the same bits
    if (D.om_out == nullptr) return;
    block_sync(P);
    if (L.valid && L.a > 0) {
        const double nvx = (double)vx, nvy = (double)vy;
        s.posd[L.lane] = make_double2(r.px + nvx * C.dt, r.py + nvy * C.dt);
        s.act[L.lane] = make_double2(nvx, nvy);
    }
    block_sync(P);
    if (L.valid && L.a > 0) {
        const double2 p = s.posd[L.lane], w = s.act[L.lane];
        double* o = D.next_obs_out + ((size_t)L.env * C.H + (L.a - 1)) * 5;
        o[0] = p.x, o[1] = p.y, o[2] = w.x, o[3] = w.y, o[4] = r.rad;
    }
    const int env0 = (int)blockIdx.x * P.E;
    const int envs_here = C.B - env0 < P.E ? C.B - env0 : P.E;
    float* om0 = D.om_out + (size_t)env0 * C.H * C.cell_num * C.cell_num * C.om_channels;
    const auto state_of = [&](int e, int j, double& px, double& py, double& wx, double& wy) {
        const double2 p = s.posd[e * P.A + 1 + j], w = s.act[e * P.A + 1 + j];
        px = p.x, py = p.y, wx = w.x, wy = w.y;
    };
    // the workgroup's lanes share the maps' float64 trigonometry (sarl_kernels.h): `lines` and `proj` are dead by now
    if (occupancy_coop_ok(C, (size_t)P.nA * 2 * kLineStride * 16, envs_here)) {
        occupancy_maps_cooperative(C, envs_here, (int)threadIdx.x, (int)blockDim.x, reinterpret_cast<char*>(s.lines), state_of,
                                   [&]() { block_sync(P); },
                                   [&](int e, int i) { return om0 + ((size_t)e * C.H + i) * C.cell_num * C.cell_num * C.om_channels; });
    } else if (L.valid && L.a > 0) {
        const int e = L.env - env0;
        occupancy_map(C, L.a - 1, [&](int j, double& px, double& py, double& wx, double& wy) { state_of(e, j, px, py, wx, wy); },
                      D.om_out + ((size_t)L.env * C.H + (L.a - 1)) * C.cell_num * C.cell_num * C.om_channels);
    }
}

}  // namespace cn
