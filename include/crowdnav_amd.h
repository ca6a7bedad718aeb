/*
 * crowdnav_amd.h — C ABI of libcrowdnav_amd.so, the MI355X (gfx950) batched crowd-navigation engine.
 *
 * The reference (vita-epfl/CrowdNav) has NO C ABI for this path: its boundary is the Python plugin API
 *   crowd_sim/envs/crowd_sim.py:51-79,251-420   CrowdSim.configure / reset / step / onestep_lookahead
 *   crowd_sim/envs/policy/orca.py:82-132        ORCA.predict  (-> rvo2.PyRVOSimulator, the only native code)
 *   crowd_nav/utils/explorer.py:21-90           Explorer.run_k_episodes
 * Each entry point below names the reference interface it replaces.  The Python host side
 * (crowdnav_amd/) binds these with ctypes and mirrors the reference's classes on top; INTEGRATION.md shows
 * the stub a reference maintainer would add.
 *
 * Conventions
 *   - every function returns 0 (CN_OK) or a negative cn_status; cn_last_error() gives the message of the
 *     last failure on the calling thread; nothing throws across the ABI.
 *   - all data pointers are DEVICE pointers (hipMalloc'd, e.g. a torch tensor's data_ptr()) unless the
 *     parameter name ends in _host.  The caller owns them and keeps them alive until the stream is synced.
 *   - calls are asynchronous on the engine's stream (cn_set_stream; default = the null stream) unless
 *     documented otherwise.  An engine is bound to one device and is not thread-safe.
 *   - agent 0 of every env is the robot, agents 1..H are the humans; A = H + 1.
 *   - batched state layout ("state8"): double [B][A][8] = px, py, vx, vy, gx, gy, radius, v_pref
 *     (crowd_sim/envs/utils/state.py:1-50 minus theta: the robot's heading is a separate [B] plane,
 *     cn_set_theta / cn_get_theta; only a unicycle robot changes it).
 */
#ifndef CROWDNAV_AMD_H
#define CROWDNAV_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum cn_status {
    CN_OK = 0,
    CN_ERR_INVALID = -1,     /* bad argument / config            */
    CN_ERR_UNSUPPORTED = -2, /* valid in the reference, not built here (e.g. a value network under the mixed rule) */
    CN_ERR_HIP = -3,         /* a HIP runtime call failed         */
    CN_ERR_NO_DEVICE = -4    /* no gfx950 device visible          */
} cn_status;

/* info codes returned by step (crowd_sim/envs/utils/info.py:1-38) */
enum { CN_NOTHING = 0, CN_DANGER = 1, CN_REACH_GOAL = 2, CN_COLLISION = 3, CN_TIMEOUT = 4 };
/* robot_policy */
enum { CN_ROBOT_EXTERNAL = 0, CN_ROBOT_ORCA = 1 };
/* robot_kinematics */
enum { CN_HOLONOMIC = 0, CN_UNICYCLE = 1 };
/* scenario_rule (crowd_sim.py:84-153) */
enum { CN_CIRCLE_CROSSING = 0, CN_SQUARE_CROSSING = 1, CN_MIXED = 2 };

/* Everything CrowdSim.configure (crowd_sim.py:51-79), the [humans]/[robot] sections read by Agent.__init__
 * (crowd_sim/envs/utils/agent.py:10-31) and the hard-coded ORCA parameters (orca.py:60-66) provide. */
typedef struct cn_config {
    int32_t num_envs;   /* B */
    int32_t num_humans; /* H, 1..63 */
    double time_step;
    double time_limit;
    double success_reward;
    double collision_penalty;
    double discomfort_dist;
    double discomfort_penalty_factor;
    int32_t robot_visible; /* humans' ORCA sees the robot (crowd_sim.py:326-327) */
    int32_t robot_policy;  /* CN_ROBOT_EXTERNAL: caller supplies the action; CN_ROBOT_ORCA: solved on device */
    double robot_safety_space;
    double human_safety_space;
    double neighbor_dist;  /* 10 */
    int32_t max_neighbors; /* 10 (<= 10 supported) */
    int32_t scenario_rule;
    double time_horizon;      /* 5 */
    double time_horizon_obst; /* 5, unused: the reference never adds obstacles */
    double circle_radius;
    double square_width;
    double human_radius;
    double human_v_pref;
    double robot_radius;
    double robot_v_pref;
    int32_t randomize_attributes; /* agent.py:39-45 */
    int32_t device;               /* HIP device ordinal */
    int32_t robot_kinematics;     /* CN_HOLONOMIC: actions are ActionXY(vx, vy); CN_UNICYCLE: ActionRot(v, r)
                                     (agent.py:115-135; CN_ROBOT_EXTERNAL only - the ORCA policy is holonomic) */
    int32_t flags;                /* CN_FLAG_* */
} cn_config;

/* cn_config.flags */
/* Generate the scenarios of cn_rollout / cn_rollout_step's auto-resets ASYNCHRONOUSLY: the fill kernels run on the engine's
 * own side streams next to the transition kernels and publish every scenario on its own (per-slot ready flag, release /
 * acquire at device scope), so a launch never waits for the hardest scenario of the batch — only the env whose NEXT
 * scenario is not ready yet pauses (it steps again in a later launch).  Trajectories and episode records are unchanged;
 * how many transitions a launch executes becomes timing-dependent.  For crowds whose rejection sampling
 * (crowd_sim.py:155-176) is heavy-tailed: 20 humans on the 4 m circle need 28 k random() calls per scenario on average and
 * 10+ M for the worst.  More than 8 humans only (the wave-cooperative generators). */
#define CN_FLAG_ASYNC_SCENARIO_FILL 1

typedef struct cn_engine cn_engine;

/* message of the last failed call on this thread ("" if none) */
const char* cn_last_error(void);
/* ABI version of this header (bumped on any change of a signature or of what a call does).  v10 (round 6): no new symbol —
 * cn_sarl_sample_step skips the envs outside `alive` on its two-launch route, takes CN_MODEL_LSTM_RL there and accepts `info` in
 * pinned host memory; cn_rollout / cn_rollout_step / the boundary calls refuse an io whose seed_base / seed_mod differ from
 * cn_rollout_begin's.  v11 (round 6): + cn_sarl_values. */
int cn_abi_version(void);

/* replaces gym.make('CrowdSim-v0') + CrowdSim.configure + set_robot (crowd_sim.py:13-82): allocates the
 * SoA state of B envs in HBM.  Synchronous. */
int cn_create(const cn_config* cfg, cn_engine** out);
int cn_destroy(cn_engine* e);
/* hipStream_t the engine launches on (pass torch.cuda.current_stream().cuda_stream); NULL = null stream */
int cn_set_stream(cn_engine* e, void* hip_stream);
/* blocks until everything queued on the engine's stream is done */
int cn_sync(cn_engine* e);

/* teacher forcing / inspection: copy state8 [B][A][8] f64 and global_time [B] f64 (either may be NULL).
 * replaces Agent.set / get_full_state (agent.py:47-108). */
int cn_set_state(cn_engine* e, const double* state8, const double* global_time);
int cn_get_state(cn_engine* e, double* state8, double* global_time);
/* robot heading theta, double [B] (FullState.theta; reset sets pi/2, crowd_sim.py:274; only a unicycle robot changes it) */
int cn_set_theta(cn_engine* e, const double* theta);
int cn_get_theta(cn_engine* e, double* theta);
/* len(env.humans) of every env, int32 [B].  Only the `mixed` rule (crowd_sim.py:103-151) draws the number of humans
 * per episode (0..5 static obstacles — 0 leaves the reference's single placeholder human at (0, -10) — or 1..5 moving
 * humans: two circle-crossing, the rest square-crossing).  The engine keeps num_humans slots per env; absent humans are
 * parked at rest at x >= 1e6, out of every neighbour range, and sit AFTER the present ones in cn_get_state / obs. */
int cn_get_human_count(cn_engine* e, int32_t* count);
/* forget the robot ORCA policy's captured radii and its rvo2 simulator's kd-tree order (a new policy object;
 * orca.py:95-104) */
int cn_drop_robot_sim(cn_engine* e);
/* (ABI v5) forget EVERY agent's rvo2 simulator — the robot's as above and each human's (a human's ORCA policy object and
 * its simulator are rebuilt with the Human at every CrowdSim.reset, crowd_sim.py:155-207; cn_reset and the rollout's
 * auto-reset do this themselves).  What a simulator carries from step to step besides the radii is the permutation its
 * kd-tree partitions in place (RVO2 KdTree::buildAgentTree; only simulators of more than 10 agents ever split): it decides
 * the visiting order of candidates at EXACTLY equal squared distance.  For a caller that teleports the agents with
 * cn_set_state and wants the behaviour of freshly built simulators. */
int cn_drop_sims(cn_engine* e);
/* The opposite: give every env the SAME captured simulator — radii host float32 [num_humans + 1] (robot, humans) as
 * the robot's rvo2 simulator holds them (radius + 0.01 + safety_space), max_speed its maxSpeed.  This is what one
 * persistent ORCA policy object means for a batch: the reference builds the simulator at the policy's first predict and
 * keeps those radii for every later episode (orca.py:95-110), so k episodes run as a batch must all see the radii of
 * the policy's first episode (only observable with randomize_attributes). */
int cn_set_robot_sim(cn_engine* e, const float* radii_host, float max_speed);

/* replaces CrowdSim.reset (crowd_sim.py:251-312) for the envs with mask[b] != 0 (mask NULL = all):
 * np.random.seed(seeds[b]) + generate_random_human_position, numpy-MT19937-compatible, on device.
 * draws (optional, uint64 [B]) receives the number of np.random.random() calls consumed. */
int cn_reset(cn_engine* e, const uint32_t* seeds, const uint8_t* mask, uint64_t* draws);

/* replaces Human.act/Robot.act -> ORCA.predict -> rvo2 doStep -> getAgentVelocity(0) for EVERY agent of
 * every env from the current state (orca.py:82-132): out_vel float [B][A][2].  Agent 0 is solved with the
 * robot's ORCA parameters whatever robot_policy is. */
int cn_orca(cn_engine* e, float* out_vel);

/* replaces CrowdSim.step(action, update) / onestep_lookahead (crowd_sim.py:314-420).
 *   action      double [B][2] (ActionXY vx, vy - or ActionRot v, r for a CN_UNICYCLE robot), NULL iff robot_policy ==
 *               CN_ROBOT_ORCA
 *   reward      double [B]; done uint8 [B]; info uint8 [B] (CN_*); dmin double [B] (+inf if no human checked)
 *   action_out  double [B][2] action actually applied (optional)
 *   orca_vel    float [B][A][2] velocity chosen by every agent (optional)
 *   obs         double [B][H][5] humans' ObservableState px,py,vx,vy,radius AFTER the step (update=1) or
 *               their next observable state (update=0, agent.py:63-74) (optional) */
int cn_step(cn_engine* e, const double* action, int update, double* reward, uint8_t* done, uint8_t* info,
            double* dmin, double* action_out, float* orca_vel, double* obs);

/* Episode bookkeeping of Explorer.run_k_episodes (explorer.py:35-72), kept per env on device. */
typedef struct cn_rollout_io {
    /* episode numbering: local env b has global env id g = env_offset + b and runs the global episode ids
     * c = g + j*env_stride (j = 0,1,..) while c < episode_limit; episode c is seeded
     * np.random.seed(seed_base + c % seed_mod) (crowd_sim.py:272-283).  A single engine uses env_offset 0,
     * env_stride B; rank r of W uses env_offset r*B, env_stride W*B, so the set of episodes and every
     * trajectory is independent of how the env axis is sharded. */
    uint32_t seed_base;
    uint32_t seed_mod;
    int64_t episode_limit; /* <0: unbounded */
    int64_t env_offset;
    int64_t env_stride;    /* >= env_offset + B */
    int32_t record_capacity; /* records kept per env: a RING — episode j of the env lands in slot j % record_capacity, so
                                once an env has finished more than record_capacity episodes slot j holds its most recent
                                episode with ordinal congruent to j (older ones are overwritten) */
    /* per-episode records, [B][record_capacity]; all optional */
    uint8_t* ep_outcome;   /* CN_REACH_GOAL / CN_COLLISION / CN_TIMEOUT */
    int32_t* ep_steps;
    double* ep_return;     /* sum_t gamma^(t*dt*v_pref) * r_t, python left-to-right order (explorer.py:71-72) */
    double* ep_time;       /* env.global_time at the end (explorer.py:52-62) */
    int32_t* ep_danger;    /* number of Danger steps (explorer.py:46-48) */
    double* ep_danger_dmin_sum;
    /* per-env counters, [B], in/out; the caller zeroes them before the first launch */
    int32_t* ep_count;     /* episodes finished so far by env b */
    int32_t* cur_steps;    /* steps into the running episode */
    double* cur_return;
    int32_t* cur_danger;
    double* cur_danger_dmin_sum;
    uint8_t* active;       /* 0 once env b ran out of episodes (c >= episode_limit) */
    uint64_t* transitions; /* [1] device counter: += number of step() transitions executed */
    /* (ABI v5) shard-boundary outputs produced by the LAST workgroup of every cn_rollout / cn_rollout_step launch itself
     * (arrival tickets, fixed summation order: bitwise reproducible), so that a single-GPU run needs no boundary kernel
     * and a sharded run only its all-gather.  Both optional (NULL = off): */
    double* summary;       /* [CN_SUMMARY_FIELDS] the numbers of cn_records_summary over THIS engine's record rings */
    double* blocks;        /* [B][CN_RECORD_BLOCK_DOUBLES(blocks_records)] what cn_rollout_records(blocks_records) packs */
    int32_t blocks_records; /* >= 1 when blocks != NULL */
    /* (ABI v6) per-env transition counters, [B], optional: env b's entry grows by the transitions env b made in the call.
     * With `transitions` and `summary` both NULL a launch ends WITHOUT any hand-off between workgroups (no arrival tickets: ~9 us
     * of the last wave's tail): whoever wants a job-wide number sums this array, and computes the statistics once per run with
     * cn_rollout_records + cn_records_summary — which is when explorer.py:74-90 computes them. */
    uint64_t* env_transitions;
} cn_rollout_io;

/* discount table used for ep_return: gamma^(t * time_step * robot_v_pref), computed on the host with libm
 * pow exactly as explorer.py:71 does.  Synchronous. */
int cn_set_gamma(cn_engine* e, double gamma);

/* (re)start episode bookkeeping: every env b with b < episode_limit is reset to its episode j=0 scenario
 * and the per-env counters of io are zeroed.  Crowds of more than 8 humans (the wave-cooperative generators): when io.seed_mod <=
 * 4096 — the episode seeds come from a small set, as in the reference's 'val' / 'test' phases (crowd_sim.py:272-283) — the rollout
 * keeps every scenario it has generated and copies it when the same seed comes round again (a seeded scenario is a pure function
 * of its seed; CROWDNAV_AMD_SCENARIO_CACHE=0 switches this off).  Same scenarios, same episodes. */
int cn_rollout_begin(cn_engine* e, const cn_rollout_io* io);
/* replaces the `while not done: action = robot.act(ob); env.step(action)` loop of
 * Explorer.run_k_episodes (explorer.py:41-48) for an on-device robot policy (robot_policy ==
 * CN_ROBOT_ORCA): n_steps transitions per active env in ONE call, with in-kernel auto-reset.  A call is one kernel launch
 * over all envs, except on the 20-human shard geometry (20 humans + robot, max_neighbors 10, holonomic), whose kernel keeps S =
 * 12 x CUs one-wave workgroups resident: when B > S (or beside the asynchronous scenario fill, or with
 * CROWDNAV_AMD_SCHED_FORCE=1) a call of n >= 24 steps (CROWDNAV_AMD_SCHED_MIN_STEPS; the static form: 48) runs under a SCHEDULE:
 *   - dynamic (default): ONE launch of min(B, S) persistent workgroups that take (env, visit) items from a device queue — a call
 *     is max(3, n / 56) visits per env (CROWDNAV_AMD_DYN_VISITS), an env's visits in order (release / acquire on a per-env word):
 *     the chip stays full whatever B / S is, and nothing waits at a launch boundary for the slowest wave of a round;
 *   - static (CROWDNAV_AMD_SCHED_DYNAMIC=0, and whenever io.summary or io.blocks is set; needs B % 4 == 0 and
 *     4 ceil(0.75 B / S) < 3 ceil(B / S)): one launch of n % 3 steps over all envs and four launches of n / 3 steps over three
 *     envs of every four (the in-kernel statistics cover every env: the last sub-launch reports for the env it leaves out).
 * Every env makes its n transitions in order either way: per-env state, per-env counters and episode records are bit-identical
 * to an unscheduled call; the float64 SUMS of io.summary (nav time, return) are accumulated in a different workgroup order under
 * the static schedule and may differ from an unscheduled call's, and from cn_records_summary's, in the last bits (counts are
 * exact).  cn_launch_counts reports what a call launched. */
int cn_rollout(cn_engine* e, const cn_rollout_io* io, int n_steps);

/* (ABI v7) what the HOST has enqueued for this engine since cn_create — counts_host: HOST uint64 [CN_LAUNCH_COUNTERS], indexed
 * by CN_COUNT_*; synchronous, no device work.  Lets a caller (bench.py, the parity tests) state which launches a timed region
 * contained — e.g. whether a cn_rollout call carried a scenario fill (the ring budget rule: a fill only when the steps since the
 * last one exceed the ring depth) and whether the 20-human shard's 3-of-4 env schedule was taken — instead of inferring it. */
enum {
    CN_COUNT_ROLLOUT_KERNELS = 0,   /* transition kernels launched by cn_rollout / cn_rollout_step */
    CN_COUNT_SCHEDULED_KERNELS = 1, /* ... of which launches of the shard kernel's schedules: one per call under the dynamic
                                       schedule, four per call (sub-launches) under the static 3-of-4 one */
    CN_COUNT_RING_FILLS = 2,        /* synchronous scenario-ring fills (one per cn_rollout call that needed one) */
    CN_COUNT_ASYNC_FILLS = 3,       /* fill launches on the side streams (CN_FLAG_ASYNC_SCENARIO_FILL: one per call) */
    CN_COUNT_SARL_NARROW = 4,       /* (ABI v9) value-network launches on the narrow tiles (cn_sarl_select / cn_sarl_sample_step
                                       of a few envs): which route a decision took */
    CN_COUNT_SARL_DECIDE_STEPS = 5  /* (ABI v9) cn_sarl_sample_step calls that ran decision + transition + next ORCA as ONE kernel */
};
#define CN_LAUNCH_COUNTERS 6
int cn_launch_counts(cn_engine* e, uint64_t* counts_host);

/* ------------------------------------------------------------------------------------------------------
 * SARL robot decision (crowd_nav/policy/sarl.py:9-86 on top of multi_human_rl.py:11-63, cadrl.py:82-222).
 * Needs robot_policy == CN_ROBOT_EXTERNAL: the chosen action is then applied with cn_step(action). */
/* value network behind cn_sarl_select:
 *   CN_MODEL_SARL   sarl.ValueNetwork (attention over humans, sarl.py:9-65).  Every model takes any num_humans the engine
 *                   holds (<= 63): crowds beyond one tile's LDS (6+ humans at the shipped SARL widths, 9+ for CADRL /
 *                   LSTM-RL) stream through the tile in chunks; occupancy maps and LSTM-RL's distance ordering likewise.
 *   CN_MODEL_CADRL  cadrl.ValueNetwork: one MLP per (robot, human) pair, value = min over humans
 *                   (crowd_nav/policy/cadrl.py:22-29, 156-168); only mlp3_dims (= [cadrl] mlp_dims), n_actions, gamma
 *                   are read; cn_sarl_set_weights then takes 8 pointers (value_network.{0,2,4,6}.{weight,bias})
 *   CN_MODEL_LSTM_RL lstm_rl.ValueNetwork1 (LSTM over the humans + value head, lstm_rl.py:9-33; the variant the shipped
 *                   policy.config selects, with_interaction_module = false): mlp1_dims[0] = [lstm_rl] global_state_dim,
 *                   mlp3_dims = [lstm_rl] mlp2_dims, with_om as configured; cn_sarl_set_weights takes 12 pointers
 *                   (mlp.{0,2,4,6}.{weight,bias}, lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0).
 *                   interaction_dims[0] > 0 selects lstm_rl.ValueNetwork2 (with_interaction_module = true,
 *                   lstm_rl.py:36-66): every human's input row first passes mlp1 = [lstm_rl] mlp1_dims (4 layers, ReLU
 *                   between them) and the LSTM runs over mlp1's outputs; cn_sarl_set_weights then takes 20 pointers:
 *                   mlp1.{0,2,4,6}.{weight,bias} followed by the 12 above (the module's state_dict order) */
enum { CN_MODEL_SARL = 0, CN_MODEL_CADRL = 1, CN_MODEL_LSTM_RL = 2 };

typedef struct cn_sarl_config {
    int32_t n_actions;          /* 81 = speed_samples * rotation_samples + 1 (cadrl.py:82-102) */
    int32_t with_om;            /* [sarl] with_om: append the occupancy map (multi_human_rl.py:109-163) */
    int32_t cell_num;           /* [om] cell_num */
    int32_t om_channel_size;    /* [om] om_channel_size: 1, 2 or 3 */
    double cell_size;           /* [om] cell_size */
    double gamma;               /* [rl] gamma */
    int32_t with_global_state;  /* [sarl] with_global_state */
    int32_t mlp1_dims[2];       /* [sarl] mlp1_dims      (150, 100) */
    int32_t mlp2_dims[2];       /* [sarl] mlp2_dims      (100, 50) */
    int32_t attention_dims[3];  /* [sarl] attention_dims (100, 100, 1) */
    int32_t mlp3_dims[4];       /* [sarl] mlp3_dims      (150, 100, 100, 1); CN_MODEL_CADRL: [cadrl] mlp_dims */
    int32_t model;              /* CN_MODEL_SARL (0), CN_MODEL_CADRL (1) or CN_MODEL_LSTM_RL (2) */
    int32_t interaction_dims[4];/* CN_MODEL_LSTM_RL only: [lstm_rl] mlp1_dims (150, 100, 100, 50) when
                                   with_interaction_module, else all 0 */
    int32_t constant_velocity_model; /* 0: [action_space] query_env = true — next human states and reward from the env's
                                   onestep_lookahead (multi_human_rl.py:37-38).  1: query_env = false — every human keeps
                                   its observed velocity for one step and the reward is MultiHumanRL.compute_reward
                                   (multi_human_rl.py:39-42, 65-88: end-point distances, constants -0.25 / 1 / 0.2 / 0.5, no
                                   time limit); for CN_MODEL_LSTM_RL the humans then enter the network in the order
                                   LstmRL.predict sorted them (decreasing distance to the robot, lstm_rl.py:96-103).
                                   SARL / LSTM-RL only: CADRL.predict always queries the env. */
    int32_t reserved;
} cn_sarl_config;

/* replaces SARL.configure + CADRL.build_action_space: actions_host = double [n_actions][2] (ActionXY table, HOST
 * pointer, computed by the caller exactly as cadrl.py:86-99 does).  Synchronous; once per engine.
 * Layer widths are free (the shipped ones have register-resident kernels).  A sarl.ValueNetwork whose tile — one (env, action)
 * group's activations for all humans plus the pipelined side buffer, 4 x 64 x (H (ks_a + ks_b + ks_c + 4) + ks_b + 3 ks_a) bytes,
 * ks = the widest layer of each buffer in 4-column steps — exceeds the 160 KiB of LDS streams the humans through in chunks
 * instead (at 5 humans: a first mlp1 / mlp3 layer wider than 160); under the `mixed` rule, whose one-tile kernel masks absent
 * humans, such a network is CN_ERR_UNSUPPORTED. */
int cn_sarl_configure(cn_engine* e, const cn_sarl_config* cfg, const double* actions_host);
/* replaces model.load_state_dict: params_host_array = HOST array of 22 DEVICE pointers to the float32 tensors of
 * sarl.ValueNetwork.state_dict() in its own order (mlp1.0.weight, mlp1.0.bias, mlp1.2.*, mlp2.0.*, mlp2.2.*,
 * attention.0.*, attention.2.*, attention.4.*, mlp3.0.*, mlp3.2.*, mlp3.4.*, mlp3.6.*); weights are [out][in]
 * row-major as torch stores them.  Repacked into MFMA operand order on device; call again after every optimizer
 * step whose result the rollout should see. */
int cn_sarl_set_weights(cn_engine* e, const float* const* params_host_array);
/* replaces the greedy branch of MultiHumanRL.predict (multi_human_rl.py:32-58) for every env:
 *   values  double [B][n_actions] (optional) reward(lookahead) + gamma^(dt*v_pref) * V(next state)
 *   best    int32  [B] index of the first strict maximum; -1 = robot already at its goal -> stop action (:22-23);
 *                  -2 = no finite value (the reference raises ValueError, :57-58)
 *   action  double [B][2] the chosen ActionXY */
int cn_sarl_select(cn_engine* e, double* values, int32_t* best, double* action);
/* replaces the epsilon-greedy branch of MultiHumanRL.predict in the train phase (multi_human_rl.py:28-31), applied to
 * the best/action a cn_sarl_select just produced: per env (mask == NULL or mask[b] != 0, and not already at its goal)
 *   probability = np.random.random(); if probability < epsilon: action_space[np.random.choice(n_actions)]
 * drawn from the env's OWN numpy stream — the one cn_reset seeded (np.random.seed, crowd_sim.py:272-276) continued
 * after the scenario draws, exactly as the sequential reference consumes it.  Needs episodes started by cn_reset
 * (otherwise the next cn_sync fails); both scenario generators (one lane / one wave per scenario) leave the env's
 * generator where the scenario's last draw left it.  explored (optional) uint8 [B]: 1 where the random action was taken. */
int cn_sarl_explore(cn_engine* e, double epsilon, const uint8_t* mask, int32_t* best, double* action,
                    uint8_t* explored);
/* replaces MultiHumanRL.transform (multi_human_rl.py:90-104; CADRL.transform cadrl.py:171-185 for one human) for the
 * CURRENT joint state of every env — the state a train-phase predict() leaves in policy.last_state and
 * Explorer.update_memory (explorer.py:92-125) pushes into the replay memory:
 *   out float32, env b's [H][13 (+ cell_num^2 * om_channel_size)] block at out + b * env_stride (floats;
 *   0 = densely packed [B][H][D]; a larger stride writes step t of a [B][T][H][D] trajectory tensor in place).
 * sort_humans != 0 (meaningful for CN_MODEL_LSTM_RL): the humans appear by decreasing distance to the robot, as
 * LstmRL.predict re-orders them before it stores last_state (lstm_rl.py:96-103; stable for equal distances) — the RL
 * phase; 0 = env order, which is what imitation learning stores (explorer.py:99 transforms the ORCA robot's own state). */
int cn_sarl_transform(cn_engine* e, float* out, int64_t env_stride, int sort_humans);
/* (ABI v8) replaces ONE iteration of the train-phase sampling loop, Explorer.run_k_episodes' `while not done: action =
 * robot.act(ob); ob, reward, done, info = env.step(action)` (explorer.py:56-65) with MultiHumanRL.predict behind robot.act
 * (multi_human_rl.py:11-63) — for every env, in this order and with these results:
 *   alive[b] &= !done[b]                              (the previous call's episode ends: an env samples until its episode is over;
 *                                                      the caller zeroes done before an episode's first call)
 *   cn_sarl_select(e, NULL, best, action)
 *   cn_sarl_explore(e, epsilon, alive, best, action, NULL)
 *   cn_sarl_transform(e, state_out, env_stride, sort_humans)      (state_out == NULL: skipped)
 *   cn_step(e, action, 1, reward, done, info, dmin, NULL, NULL, NULL)
 * as one call.  For a FEW envs (CN_MODEL_SARL with or without occupancy maps, or CN_MODEL_CADRL; up to 8 humans, not the `mixed` rule, at most one
 * workgroup per CU: 9 envs of 5 humans x 81 actions — BASELINE configs[4]'s one episode at a time, train.py:156-170) a streamed
 * loop of these calls is TWO launches per step instead of eight: the value network on tiles of 16 / num_humans whole
 * (env, action) groups, one per workgroup — a decision spread over 27 CUs instead of 6, its input rows built in LDS, each
 * tile adding the lookahead reward of its own groups and writing env b's replay-memory state on an idle wave; then ONE kernel
 * for the arg-max, the epsilon-greedy draw, the transition and the humans' ORCA velocities (with occupancy maps also their next
 * states and the maps) of the NEXT decision.  Those
 * velocities are trusted by the next call only if no other entry point of this engine ran in between (any of them may change
 * the state they belong to); otherwise, and on the first call, ORCA is a launch of its own in front.  Same bits as the five
 * calls above in every case (tests/test_rl_pipeline.py).  cn_sarl_select takes the same network kernel at these sizes (two
 * launches less).  CROWDNAV_AMD_SARL_NARROW=0 keeps the one-tile kernels, 2 takes the narrow tiles at any size the
 * configuration allows; CROWDNAV_AMD_SARL_FUSED_STEP=0 (and workgroups of several waves / simulators of more than 10
 * agents): ORCA, the network with the decision by its last workgroup, the transition — three launches.
 * Round 6: CN_MODEL_LSTM_RL (lstm_rl.ValueNetwork1, the environment queried) takes the two launches as well; the transition
 * kernel reuses the humans' ORCA velocities the decision's lookahead was given (one ORCA pass per step); and on the two-launch
 * route an env OUTSIDE alive (its episode is over) is SKIPPED: no decision, no replay-memory state, no transition — its
 * state, its done flag and its entries of reward / info / dmin / best / state_out stay as its last sampled step left them
 * (the other routes keep stepping such an env, as cn_step does; either way those outputs mean nothing).  A caller that
 * cannot know an episode's end without a round trip may therefore stream a few calls past it at the price of two
 * near-empty launches each; info — and reward / dmin / best, which that route only writes, a step's before its info code and
 * an episode's last step's acknowledged before it — (written, never read, by the kernels) may point into pinned host memory so that the host
 * sees the episode-end codes arrive without synchronising (compat.Explorer._run_batched_rl). */
int cn_sarl_sample_step(cn_engine* e, double epsilon, uint8_t* alive, int32_t* best, double* action, float* state_out,
                        int64_t env_stride, int sort_humans, double* reward, uint8_t* done, uint8_t* info, double* dmin);
/* (ABI v11) V(state) of n joint states the CALLER holds — float32 [n][num_humans][13] rows in the layout cn_sarl_transform /
 * cn_sarl_sample_step write (a replay memory's states) — under the weights last given to cn_sarl_set_weights: what
 * `target_model(next_states)` is to the TD targets of Explorer.update_memory (crowd_nav/utils/explorer.py:113-116), on the
 * narrow tiles (sarl_narrow_kernel: ONE launch, the rows read where they lie) instead of the ~35 library kernels of a
 * framework forward on a few dozen rows.  out: float32 [n].  The env state of the engine is neither read nor written — an engine
 * that only serves a target network needs no reset.  CN_MODEL_SARL / CN_MODEL_LSTM_RL (ValueNetwork1) without occupancy maps
 * at a size that takes the narrow tiles (cn_sarl_sample_step's conditions), 1 <= n <= num_envs x n_actions; CN_ERR_UNSUPPORTED
 * otherwise.  Same arithmetic as the decision's network kernel (tests/test_sarl.py: <= 2e-6 of the framework's forward). */
int cn_sarl_values(cn_engine* e, const float* states, int64_t n, float* out);
/* test/inspection: copy an internal buffer of the last cn_sarl_select to dst (device pointer):
 *   0 reward f64 [B][K] · 1 V f32 [B*K] · 2 next human states f64 [B][H][5] · 3 occupancy maps f32 [B][H][cells*ch]
 *   4 X f32 in MLP tile order (see sarl_kernels.h) */
int cn_sarl_export(cn_engine* e, int which, void* dst, uint64_t bytes);

/* The same episode loop for a robot whose policy lives OUTSIDE the engine (robot_policy == CN_ROBOT_EXTERNAL, e.g. the
 * SARL decision of cn_sarl_select): ONE transition of every running env with action double [B][2], plus all of
 * cn_rollout's bookkeeping and seeded auto-reset from the scenario ring — `action = robot.act(ob); env.step(action)`
 * of explorer.py:41-48 without leaving the device. */
int cn_rollout_step(cn_engine* e, const cn_rollout_io* io, const double* action);

/* ------------------------------------------------------------------------------------------------------
 * Shard boundary: the statistics block of Explorer.run_k_episodes (explorer.py:50-62, 71-90) needs every finished
 * episode on one rank.  The env axis is sharded over GPUs with NO collective on the step path (cn_rollout_io.env_offset /
 * env_stride); the only exchange is one all-gather of fixed-size per-env record blocks when a run ends. */
#define CN_RECORD_FIELDS 6  /* outcome (CN_*), steps, discounted return, nav time, danger steps, danger dmin sum */
#define CN_SUMMARY_FIELDS 8
/* doubles per env in a record block holding up to K records: { episodes finished, K x CN_RECORD_FIELDS } */
#define CN_RECORD_BLOCK_DOUBLES(K) (1 + (K) * CN_RECORD_FIELDS)
/* pack the record rings of io into ONE self-contained float64 block per env: blocks double
 * [B][CN_RECORD_BLOCK_DOUBLES(max_records)]; block b = { number of episodes env b has finished (unclamped), then record j
 * = ring slot j for j < min(count, record_capacity, max_records), zeros beyond }.  Slot j is the env's j-th finished episode
 * while count <= record_capacity; after the ring has wrapped it is the most recent episode with ordinal = j mod
 * record_capacity (size the rings for the episodes a run can finish when per-episode identities matter).  One kernel. */
int cn_rollout_records(cn_engine* e, const cn_rollout_io* io, int max_records, double* blocks);
/* all-gather such blocks over the ranks of an RCCL communicator (rccl_comm = the caller's ncclComm_t, one rank per GPU;
 * librccl.so.1 is bound at first use): blocks_all double [n_ranks * B][CN_RECORD_BLOCK_DOUBLES(max_records)], rank-major
 * = ordered by global env id, identical on every rank.  ONE ncclAllGather on the engine's stream (200 KiB per rank at
 * 4096 envs, max_records 1: latency-bound).  A reference process that owns one engine per GPU calls this where
 * explorer.py:74 starts. */
int cn_gather_records(cn_engine* e, void* rccl_comm, int n_ranks, int max_records, const double* blocks,
                      double* blocks_all);
/* explorer.py:74-90 on record blocks (a shard's own or the gathered ones): summary double [CN_SUMMARY_FIELDS] =
 * episodes finished, records held, ReachGoal / Collision / Timeout among them, sum of the successful nav times, sum of
 * the discounted returns, sum of the Danger steps.  record_capacity = the capacity of the rings the blocks were packed
 * from (records beyond it are not counted).  One kernel, fixed summation order (bitwise reproducible; rates and averages
 * are quotients of these). */
int cn_records_summary(cn_engine* e, int64_t n_envs, int max_records, int record_capacity, const double* blocks,
                       double* summary);
/* (ABI v6) the same numbers for THIS engine's own record rings, without packing blocks first: what a single-engine run calls
 * where explorer.py:74 starts (bitwise equal to cn_records_summary over cn_rollout_records(record_capacity)). */
int cn_rollout_summary(cn_engine* e, const cn_rollout_io* io, double* summary);

/* numpy legacy RNG probe (np.random.seed(seed); n × np.random.random()): out double [n].  For tests. */
int cn_mt_random(cn_engine* e, uint32_t seed, int n, double* out);

#ifdef __cplusplus
}
#endif
#endif /* CROWDNAV_AMD_H */
