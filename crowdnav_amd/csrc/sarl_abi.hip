// C ABI of the SARL robot decision (declared in include/crowdnav_amd.h): cn_sarl_* and the value-network kernels, a
// translation unit of its own (engine_host.h has what it shares with crowdnav_amd.hip).
#define CN_SARL_TU  // step_kernels.h: the ORCA / step kernels and the engine types, not the rollout and scenario kernels
#include "engine_host.h"
#include "sarl_kernels.h"
#include "sarl_reg_kernel.h"
#include "sarl_step_fused.h"

struct cn_sarl {
    cn_sarl_config cfg;
    cn::SarlCfg C;
    cn::SarlNet net;
    cn::SarlNetRef ref;    // the same layers as offsets into `arena` (persistent value-network kernel)
    float* arena;          // one allocation for every layer's packed weights and biases
    size_t arena_used;
    double* actions;    // [K][2] device copy of the action table
    float* orca_vel;    // [B][A][2]
    double* next_obs;   // [B][H][5]
    float* om;          // [B][H][cells*channels]
    double* reward;     // [B][K]
    float* X;           // [tiles][H][ks_x][64] (MFMA A-fragment order)
    float* V;           // [B*K]
    int* hcount;        // [tiles * 16] humans present per (env, action) group (num_humans unless the `mixed` rule parks some)
    size_t n_groups, n_tiles;
    size_t lds_bytes;
    bool weights_set;
    bool chunked;       // sarl_mlp_chunked_kernel: the humans do not fit one tile's LDS
    bool reg_mlp;       // sarl_reg_kernel: activations in registers (shipped widths, 5 humans, more than 512 tiles); CROWDNAV_AMD_SARL_REG=0 / 2: never / always
    int reg_xks;        // its network key: k-steps of the SARL input, 4 (13 features) or 16 (+ 48 occupancy-map features); kRegCadrl
    float* reg_stream;  // its weight stream: reg_total_quads(reg_xks) quads of 256 floats
    float* reg_stream2; // LSTM-RL: the value head's stream (reg_stream is the gate layer's)
    float* reg_stream3; // sarl_reg_chunk_kernel (6+ humans): streams A, G, B = reg_stream, reg_stream2, reg_stream3
    float* reg_scratch; // ... and its per-wave parking space for mlp1's output
    int chunk_nt, n_chunks;
    int cadrl_nt, cadrl_chunks;  // cadrl_reg_kernel: humans per chunk (1..5), chunks per tile (1 up to 5 humans)
    float* om_w;        // kRegSarlPre: mlp1.0's occupancy-map columns [48][160 slots] and its bias [160] (sarl_om_weights_kernel)
    float* om_term;     // kRegSarlPre: b + W[:, 13:61] om per (env, human), [B * H][160] in accumulator order
    int n_cus;
    // sarl_narrow_kernel: a few decisions (train.py's single-episode sampling) on tiles of 16 / H groups, one per workgroup
    bool narrow;
    size_t narrow_tiles, narrow_lds;
    bool fused_step;      // cn_sarl_sample_step on the narrow route: decision + transition + next ORCA as one kernel (CROWDNAV_AMD_SARL_FUSED_STEP)
    int* narrow_counter;  // cn_sarl_sample_step: workgroups of sarl_narrow_kernel that have written their V
    double* narrow_value; // ... and reward + gamma V per (env, action), each written by the tile that computed V
    cn::PackJobs pack_jobs = {};  // cn_sarl_set_weights: the layers to repack, run as one launch (sarl_pack_flush)
    int pack_blocks = 0;
};

void cn_sarl_release(cn_engine* e) {
    delete e->sarl;  // device buffers are owned by the engine's slabs
    e->sarl = nullptr;
}

namespace {

// the packing jobs of one cn_sarl_set_weights call are collected and run as ONE launch (sarl_pack_flush)
int sarl_pack_flush(cn_engine* e) {
    cn::PackJobs& jobs = e->sarl->pack_jobs;
    if (jobs.n > 0) {
        hipLaunchKernelGGL(cn::sarl_pack_many_kernel, dim3((unsigned)e->sarl->pack_blocks), dim3(256), 0, e->stream, jobs);
        jobs.n = 0, e->sarl->pack_blocks = 0;
        CN_HIP(hipGetLastError());
    }
    return CN_OK;
}
int sarl_pack(cn_engine* e, cn::PackedLinear& L, const float* W, const float* bias, int N, int K, int k_off, int k_cnt) {
    (void)k_cnt;
    cn::PackJobs& jobs = e->sarl->pack_jobs;
    if (jobs.n == cn::kPackJobs) {
        const int rc = sarl_pack_flush(e);
        if (rc) return rc;
    }
    const int total = L.ctiles * L.kpad * 64;
    cn::PackJob& J = jobs.job[jobs.n++];
    J.W = W, J.bias = bias, J.wp = const_cast<float*>(L.w), J.bp = const_cast<float*>(L.bias);
    J.N = N, J.K = K, J.k_offset = k_off, J.k_count = L.K, J.kpad = L.kpad, J.ctiles = L.ctiles, J.first_block = e->sarl->pack_blocks;
    e->sarl->pack_blocks += (total + 255) / 256;
    return CN_OK;
}

constexpr size_t kSarlArenaFloats = (size_t)4 << 20;  // 16 MiB: the shipped networks pack into ~0.5 MiB

int sarl_alloc_layer(cn_engine* e, struct cn_sarl* s, cn::PackedLinear& L, int N, int K) {
    (void)e;
    L.K = K, L.N = N, L.ksteps = (K + 3) / 4, L.ctiles = (N + 15) / 16;
    L.kpad = (L.ksteps + cn::kSarlKChunk - 1) / cn::kSarlKChunk * cn::kSarlKChunk;
    if (L.kpad > 255 || L.ctiles > 255) return fail(CN_ERR_UNSUPPORTED, "layer %d -> %d is wider than the packed descriptors hold", K, N);
    const size_t nw = (size_t)(L.ctiles * L.kpad + 2 * cn::kSarlKChunk) * 64, nb = ((size_t)L.ctiles * 16 + 63) / 64 * 64;
    if (s->arena_used + nw + nb > kSarlArenaFloats) return fail(CN_ERR_UNSUPPORTED, "value network does not fit the weight arena");
    L.w = s->arena + s->arena_used;
    L.bias = s->arena + s->arena_used + nw;
    s->arena_used += nw + nb;
    return CN_OK;
}

}  // namespace

extern "C" {

int cn_sarl_configure(cn_engine* e, const cn_sarl_config* c, const double* actions_host) {
    int rc = bind(e);
    if (rc) return rc;
    if (!c || !actions_host) return fail(CN_ERR_INVALID, "cn_sarl_configure: NULL argument");
    if (e->sarl) return fail(CN_ERR_INVALID, "cn_sarl_configure: already configured for this engine");
    const int H = e->cfg.num_humans;
    if (c->n_actions < 1) return fail(CN_ERR_INVALID, "n_actions must be >= 1");
    const int extra = c->with_om ? c->cell_num * c->cell_num * c->om_channel_size : 0;
    if (c->with_om && (c->cell_num < 1 || c->om_channel_size < 1 || c->om_channel_size > 3 || !(c->cell_size > 0)))
        return fail(CN_ERR_INVALID, "bad occupancy-map parameters");
    if (c->with_om && H < 2) return fail(CN_ERR_INVALID, "occupancy maps need at least 2 humans (multi_human_rl.py:117)");
    const int in_dim = 13 + extra;
    const bool cadrl = c->model == CN_MODEL_CADRL, lstm = c->model == CN_MODEL_LSTM_RL;
    if (c->model != CN_MODEL_SARL && !cadrl && !lstm)
        return fail(CN_ERR_INVALID, "unknown value-network model %d", c->model);
    if (cadrl && c->with_om) return fail(CN_ERR_INVALID, "CADRL has no occupancy-map input");
    if (cadrl && c->constant_velocity_model)
        return fail(CN_ERR_INVALID, "CADRL.predict always queries the env (cadrl.py:150): constant_velocity_model is for SARL / LSTM-RL");
    if (lstm && c->mlp1_dims[0] < 1) return fail(CN_ERR_INVALID, "LSTM-RL: mlp1_dims[0] must hold the hidden width");
    const bool pairwise = lstm && c->interaction_dims[0] > 0;  // lstm_rl.ValueNetwork2
    if (pairwise)
        for (int i = 0; i < 4; ++i)
            if (c->interaction_dims[i] < 1) return fail(CN_ERR_INVALID, "LSTM-RL: interaction_dims needs 4 positive widths");
    if (!cadrl && !lstm) {
        for (int i = 0; i < 2; ++i)
            if (c->mlp1_dims[i] < 1 || c->mlp2_dims[i] < 1) return fail(CN_ERR_INVALID, "bad mlp dims");
        if (c->attention_dims[2] != 1) return fail(CN_ERR_UNSUPPORTED, "attention must end in a single output");
    }
    for (int i = 0; i < 3; ++i)
        if (c->mlp3_dims[i] < 1) return fail(CN_ERR_INVALID, "bad mlp dims");
    if (c->mlp3_dims[3] != 1) return fail(CN_ERR_UNSUPPORTED, "the value head must end in a single output");

    if (!cadrl && !lstm && (c->attention_dims[0] < 1 || c->attention_dims[1] < 1))
        return fail(CN_ERR_INVALID, "bad attention dims");
    // built in a local object and handed to the engine only when everything below succeeded: a failed configure leaves
    // the engine unconfigured
    // ... and the device buffers allocated on the way are freed again: a host that retries configurations (say, falling
    // back from a chunked network under the mixed rule) must not leak the 16 MiB weight arena per attempt
    struct Guard {
        cn_sarl* p;
        cn_engine* e;
        cn_engine::AllocMark mark;
        ~Guard() {
            if (!p) return;  // success: ownership went to the engine
            e->alloc_rollback(mark);
            delete p;
        }
    } guard{new (std::nothrow) cn_sarl(), e, e->alloc_mark()};
    cn_sarl* s = guard.p;
    if (!s) return fail(CN_ERR_INVALID, "out of host memory");
    s->cfg = *c;
    s->weights_set = false;
    s->chunked = false;
    s->reg_mlp = false, s->reg_xks = 0, s->reg_stream = s->reg_stream2 = s->reg_stream3 = s->reg_scratch = s->om_w = s->om_term = nullptr;
    s->chunk_nt = s->n_chunks = s->cadrl_nt = s->cadrl_chunks = 0;
    {
        hipDeviceProp_t prop;
        CN_HIP(hipGetDeviceProperties(&prop, e->cfg.device));
        s->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    cn::SarlCfg& C = s->C;
    C.B = e->P.B, C.H = H, C.n_actions = c->n_actions;
    C.with_om = c->with_om ? 1 : 0, C.cell_num = c->cell_num, C.om_channels = c->om_channel_size, C.cell_size = c->cell_size;
    C.dt = e->P.dt, C.time_limit = e->P.time_limit, C.success_reward = e->P.success_reward;
    C.collision_penalty = e->P.collision_penalty, C.discomfort_dist = e->P.discomfort_dist;
    C.discomfort_factor = e->P.discomfort_factor;
    C.gamma_bar = std::pow(c->gamma, e->cfg.time_step * e->cfg.robot_v_pref);
    C.unicycle = e->P.robot_unicycle;
    C.const_vel = c->constant_velocity_model ? 1 : 0;
    C.cadrl = cadrl ? 1 : 0;
    C.sort_lookahead = (C.const_vel && lstm) ? 1 : 0;

    s->arena = nullptr, s->arena_used = 0;
    if ((rc = dev_alloc(e, &s->arena, kSarlArenaFloats))) return rc;
    cn::SarlNet& net = s->net;
    net.in_dim = in_dim, net.with_global = c->with_global_state ? 1 : 0, net.H = H;
    const int m1a = c->mlp1_dims[0], m1b = c->mlp1_dims[1], m2a = c->mlp2_dims[0], m2b = c->mlp2_dims[1];
    const int a0 = c->attention_dims[0], a1 = c->attention_dims[1];
    const int j0 = c->mlp3_dims[0], j1 = c->mlp3_dims[1], j2 = c->mlp3_dims[2];
    const int dims[cn::kSarlLayers][2] = {  // {N, K}
        {m1a, in_dim}, {m1b, m1a}, {m2a, m1b}, {m2b, m2a}, {a0, m1b}, {a0, m1b}, {a1, a0}, {1, a1},
        {j0, 6 + m2b}, {j1, j0}, {j2, j1}, {1, j2}};
    const int hid = c->mlp1_dims[0];  // LSTM-RL: hidden width
    for (int l = 0; l < cn::kSarlLayers; ++l) {
        net.L[l] = cn::PackedLinear{};
        int N = dims[l][0], K = dims[l][1];
        if (cadrl) {
            if (l < cn::kL_mlp3_0) continue;  // CADRL uses the value head only
            if (l == cn::kL_mlp3_0) K = in_dim;
        } else if (lstm) {
            const int* id = c->interaction_dims;
            if (l == cn::kL_mlp1_0) {
                N = 4 * hid, K = pairwise ? id[3] : in_dim;  // weight_ih_l0
            } else if (l == cn::kL_mlp1_2) {
                N = 4 * hid, K = hid;     // weight_hh_l0
            } else if (pairwise && l == cn::kL_mlp2_0) {  // ValueNetwork2.mlp1: 4 layers in the otherwise unused slots
                N = id[0], K = in_dim;
            } else if (pairwise && l == cn::kL_mlp2_2) {
                N = id[1], K = id[0];
            } else if (pairwise && l == cn::kL_att_2) {
                N = id[2], K = id[1];
            } else if (pairwise && l == cn::kL_att_4) {
                N = id[3], K = id[2];
            } else if (l < cn::kL_mlp3_0) {
                continue;
            } else if (l == cn::kL_mlp3_0) {
                K = 6 + hid;
            }
        }
        if ((rc = sarl_alloc_layer(e, s, net.L[l], N, K))) return rc;
    }
    auto max2 = [](int a, int b) { return a > b ? a : b; };
    // k-steps per row tile of each LDS buffer = whole column tiles of the widest layer written into it
    auto ks_of = [](int n) { return cn::sarl_ks(n); };
    net.ks_x = ks_of(in_dim);
    net.ks_a = ks_of(max2(max2(max2(m1a, m2a), max2(a0, 6 + m2b)), max2(max2(j0, j1), j2)));
    net.ks_b = max2(ks_of(max2(m1b, a1)), net.ks_x);
    net.ks_c = ks_of(m2b);
    net.ks_s = 4;
    if (lstm) {
        net.ks_a = ks_of(max2(max2(j0, j1), max2(j2, 6 + hid)));
        net.ks_b = net.ks_c = 0;
        const size_t ks_g = (size_t)net.L[cn::kL_mlp1_0].ctiles * 4, ks_h = (size_t)cn::sarl_ks(hid);
        if (pairwise) {  // ping-pong buffers of ValueNetwork2.mlp1, all H row tiles: widths d0, d2 -> ks_b; d1, d3 -> ks_c
            const int* id = c->interaction_dims;
            net.ks_b = ks_of(max2(id[0], id[2]));
            net.ks_c = ks_of(max2(id[1], id[3]));
        }
        // more than kSarlMaxHumans humans: lstm_mlp_anyh_kernel stages one human's row tile per LSTM step (nothing sized by H)
        // (... and whenever H row tiles do not fit: ValueNetwork2's ping-pong buffers from 6 humans on)
        const auto lds_for = [&](size_t row_tiles) {
            return sizeof(float) * (64 * (row_tiles * (net.ks_x + net.ks_b + net.ks_c) + ks_g + ks_h + 2 * net.ks_a + net.ks_s) +
                                    (size_t)hid * cn::kSarlGroups);
        };
        s->chunked = H > cn::kSarlMaxHumans || lds_for((size_t)H) > 160 * 1024;
        s->lds_bytes = lds_for(s->chunked ? 1 : (size_t)H);
    } else if (cadrl) {  // ping-pong between A (first / third hidden layer) and B (X staging, second hidden layer)
        net.ks_a = ks_of(max2(j0, j2));
        net.ks_b = max2(ks_of(j1), net.ks_x);
        net.ks_c = 0;
        s->lds_bytes = sizeof(float) * 64 * (size_t)H * (net.ks_a + net.ks_b + net.ks_s);
        s->chunked = H > cn::kSarlMaxHumans;  // cadrl_mlp_chunked_kernel streams the humans in chunks of 5
        if (s->chunked) s->lds_bytes = cn::cadrl_mlp_chunked_lds_bytes(net);
    } else {
        // one tile's activations + the side chain's pong buffer (sarl_mlp_pipe_kernel) in LDS, or the humans stream through
        // in chunks (6+ humans at the shipped widths)
        s->lds_bytes = cn::sarl_mlp_lds_bytes(net) + cn::sarl_mlp_pipe_extra_lds_bytes(net);
        s->chunked = H > cn::kSarlMaxHumans || s->lds_bytes > 160 * 1024;
        if (s->chunked) s->lds_bytes = cn::sarl_mlp_chunked_lds_bytes(net);
    }
    // The register-resident kernel is compiled for the shipped network (policy.config [sarl]) on 1..5 humans.  It is a THROUGHPUT
    // kernel: one wave carries a tile through the whole network in ~86 us, 1024 of them at a time; the LDS kernel puts a whole
    // workgroup on a tile (37 us, 256 at a time).  Up to 512 tiles (~100 envs x 81 actions: the single-episode sampling of
    // train.py) the LDS kernel finishes first.  CROWDNAV_AMD_SARL_REG: 0 never, 1 (default) by size, 2 always.
    const int reg_mode = env_int("CROWDNAV_AMD_SARL_REG", 1);
    const size_t tiles_here = ((size_t)C.B * C.n_actions + cn::kSarlGroups - 1) / cn::kSarlGroups;
    if (!cadrl && !lstm && H >= 1 && H <= cn::kRegHumans && c->with_global_state && (in_dim == 13 || in_dim == 61) &&
        m1a == 150 && m1b == 100 && m2a == 100 && m2b == 50 && a0 == 100 && a1 == 100 && j0 == 150 && j1 == 100 && j2 == 100 &&
        (reg_mode == 2 || (reg_mode == 1 && tiles_here > 512))) {
        // 61 inputs: the occupancy-map half of mlp1.0 is hoisted out of the action loop (sarl_om_term_kernel, kRegSarlPre)
        s->reg_mlp = true, s->reg_xks = in_dim == 13 ? 4 : cn::kRegSarlPre;
        if ((rc = dev_alloc(e, &s->reg_stream, (size_t)cn::reg_total_quads(s->reg_xks) * 256))) return rc;
        if (s->reg_xks == cn::kRegSarlPre &&
            ((rc = dev_alloc(e, &s->om_w, (size_t)160 * 49)) || (rc = dev_alloc(e, &s->om_term, (size_t)C.B * H * 160))))
            return rc;
    }
    if (!cadrl && !lstm && H > cn::kRegHumans && c->with_global_state && (in_dim == 13 || in_dim == 61) && m1a == 150 &&
        m1b == 100 && m2a == 100 && m2b == 50 && a0 == 100 && a1 == 100 && j0 == 150 && j1 == 100 && j2 == 100 &&
        (reg_mode == 2 || (reg_mode == 1 && tiles_here > 512))) {  // sarl_reg_chunk_kernel: chunks of 3 or 4 humans
        s->n_chunks = (H + 3) / 4, s->chunk_nt = (H + s->n_chunks - 1) / s->n_chunks;
        s->reg_mlp = true, s->reg_xks = in_dim == 13 ? cn::kRegChunkA : cn::kRegChunkAPre;
        const size_t waves = (size_t)s->n_cus * cn::kRegWaves;
        if ((rc = dev_alloc(e, &s->reg_stream, (size_t)cn::reg_total_quads(s->reg_xks) * 256)) ||
            (rc = dev_alloc(e, &s->reg_stream2, (size_t)cn::reg_total_quads(cn::kRegChunkG) * 256)) ||
            (rc = dev_alloc(e, &s->reg_stream3, (size_t)cn::reg_total_quads(cn::kRegChunkB) * 256)) ||
            (rc = dev_alloc(e, &s->reg_scratch, waves * s->n_chunks * s->chunk_nt * 7 * 256)))
            return rc;
        if (s->reg_xks == cn::kRegChunkAPre &&
            ((rc = dev_alloc(e, &s->om_w, (size_t)160 * 49)) || (rc = dev_alloc(e, &s->om_term, (size_t)C.B * H * 160))))
            return rc;
    }
    if (cadrl && H >= 1 && in_dim == 13 && j0 == 150 && j1 == 100 && j2 == 100 &&
        (reg_mode == 2 || (reg_mode == 1 && tiles_here > 512))) {  // cadrl_reg_kernel: [cadrl] mlp_dims = 150, 100, 100, 1
        s->reg_mlp = true, s->reg_xks = cn::kRegCadrl;
        s->cadrl_chunks = (H + cn::kRegHumans - 1) / cn::kRegHumans, s->cadrl_nt = (H + s->cadrl_chunks - 1) / s->cadrl_chunks;
        if ((rc = dev_alloc(e, &s->reg_stream, (size_t)cn::reg_total_quads(s->reg_xks) * 256))) return rc;
    }
    if (lstm && !pairwise && hid == cn::kRegLstmHid && (in_dim == 13 || in_dim == 61) && j0 == 150 && j1 == 100 &&
        j2 == 100 && (reg_mode == 2 || (reg_mode == 1 && tiles_here > 512))) {  // lstm_reg_kernel: any number of humans
        s->reg_mlp = true, s->reg_xks = cn::kRegLstmGates + (in_dim == 13 ? 4 : 16);
        if ((rc = dev_alloc(e, &s->reg_stream, (size_t)cn::reg_total_quads(s->reg_xks) * 256)) ||
            (rc = dev_alloc(e, &s->reg_stream2, (size_t)cn::reg_total_quads(cn::kRegLstmHead) * 256)))
            return rc;
    }
    if (pairwise && hid == cn::kRegLstmHid && (in_dim == 13 || in_dim == 61) && c->interaction_dims[0] == 150 &&
        c->interaction_dims[1] == 100 && c->interaction_dims[2] == 100 && c->interaction_dims[3] == cn::kRegLstmHid && j0 == 150 &&
        j1 == 100 && j2 == 100 && (reg_mode == 2 || (reg_mode == 1 && tiles_here > 512))) {  // lstm2_reg_kernel (ValueNetwork2)
        s->reg_mlp = true, s->reg_xks = cn::kRegLstmMlp1 + (in_dim == 13 ? 4 : 16);
        if ((rc = dev_alloc(e, &s->reg_stream, (size_t)cn::reg_total_quads(s->reg_xks) * 256)) ||
            (rc = dev_alloc(e, &s->reg_stream2, (size_t)cn::reg_total_quads(cn::kRegLstmHead) * 256)) ||
            (rc = dev_alloc(e, &s->reg_stream3, (size_t)cn::reg_total_quads(cn::kRegLstmGates + cn::kRegLstmKs) * 256)))
            return rc;
    }
    if (s->lds_bytes > 160 * 1024)
        return fail(CN_ERR_UNSUPPORTED, "SARL network of this size needs %zu bytes of LDS per tile (> 160 KiB)", s->lds_bytes);

    s->n_groups = (size_t)C.B * C.n_actions;
    s->n_tiles = (s->n_groups + cn::kSarlGroups - 1) / cn::kSarlGroups;
    // Few decisions: 16-row tiles of whole groups, one per workgroup, X built in the kernel (sarl_narrow_kernel) — while the
    // whole launch is at most one workgroup per CU (measured, a sampled step of 5 humans x 81 actions: 8 envs 53 us against
    // 73 us on the one-tile kernels, 16 envs 90 against 74: 16-group tiles do ~1.5 x less matrix work per group).
    // CROWDNAV_AMD_SARL_NARROW: 0 never, 1 (default) by size, 2 whenever the configuration allows it.
    {
        const int narrow_mode = env_int("CROWDNAV_AMD_SARL_NARROW", 1);
        const size_t per_tile = (size_t)(cn::kSarlGroups / (H < 1 ? 1 : H));
        s->narrow_tiles = per_tile ? (s->n_groups + per_tile - 1) / per_tile : 0;
        s->narrow_lds = cn::sarl_narrow_lds_bytes(net, lstm);
        s->fused_step = env_int("CROWDNAV_AMD_SARL_FUSED_STEP", 1) != 0 && e->P.threads == 64 && !e->P.kd;
        // LSTM-RL (round 6): lstm_rl.ValueNetwork1 with the environment queried (the joint state LstmRL.predict sorted feeds the
        // network only under the constant-velocity model); its straight-line k loops hold W_ih rows of up to 80 inputs and W_hh
        // of up to 60 hidden units
        const bool lstm_ok = !lstm || (!pairwise && net.L[cn::kL_mlp1_0].kpad <= 4 * cn::kSarlKChunk &&
                                       net.L[cn::kL_mlp1_2].kpad <= 3 * cn::kSarlKChunk && net.L[cn::kL_mlp3_0].kpad <= cn::kNarrowK);
        s->narrow = lstm_ok && !s->chunked && !s->reg_mlp && (in_dim == 13 || (C.with_om && !cadrl)) && !C.sort_lookahead && H >= 1 &&
                    H <= cn::kSarlMaxHumans && s->narrow_lds <= 160 * 1024 &&
                    (narrow_mode == 2 || (narrow_mode == 1 && s->narrow_tiles <= (size_t)s->n_cus));
    }
    const size_t nA = (size_t)C.B * (H + 1);
    if ((rc = dev_alloc(e, &s->actions, (size_t)2 * C.n_actions)) || (rc = dev_alloc(e, &s->orca_vel, 2 * nA)) ||
        (rc = dev_alloc(e, &s->next_obs, (size_t)C.B * H * 5)) ||
        (rc = dev_alloc(e, &s->om, (size_t)C.B * H * (extra > 0 ? extra : 1))) ||
        (rc = dev_alloc(e, &s->reward, s->n_groups)) || (rc = dev_alloc(e, &s->V, s->n_tiles * cn::kSarlGroups)) ||
        (rc = dev_alloc(e, &s->X, s->n_tiles * H * net.ks_x * 64)) ||
        (rc = dev_alloc(e, &s->hcount, s->n_tiles * cn::kSarlGroups)) || (rc = dev_alloc(e, &s->narrow_counter, 1)) ||
        (rc = dev_alloc(e, &s->narrow_value, s->n_groups)))
        return rc;
    CN_HIP(hipMemset(s->narrow_counter, 0, sizeof(int)));
    if (e->cfg.scenario_rule == CN_MIXED && s->chunked) {
        if (H <= cn::kSarlMaxHumans && !cadrl && !lstm)  // the network, not the crowd, is too large for one tile
            return fail(CN_ERR_UNSUPPORTED,
                        "value networks under the mixed rule run the one-tile kernel (it masks an episode's absent humans), whose "
                        "tile — activations %zu B + the pipelined side buffer %zu B — must fit the 160 KiB of LDS; these layer "
                        "widths do not (narrower mlp1 / mlp3 layers do: the shipped 150-wide network needs 150.5 KiB)",
                        cn::sarl_mlp_lds_bytes(net), cn::sarl_mlp_pipe_extra_lds_bytes(net));
        return fail(CN_ERR_UNSUPPORTED,
                    "value networks under the mixed rule run the one-tile kernels (they mask an episode's absent humans): "
                    "num_humans must be 5 (the rule never draws more)");
    }
    CN_HIP(hipMemcpy(s->actions, actions_host, sizeof(double) * 2 * C.n_actions, hipMemcpyHostToDevice));
    s->ref = cn::SarlNetRef{};
    s->ref.base = s->arena;
    for (int l = 0; l < cn::kSarlLayers; ++l) {
        const cn::PackedLinear& L = net.L[l];
        if (!L.w) continue;
        s->ref.L[l] = cn::LayerRef{(uint32_t)(L.w - s->arena), (uint32_t)(L.bias - s->arena),
                                   (uint32_t)L.kpad | ((uint32_t)L.ctiles << 8) | ((uint32_t)L.ksteps << 16)};
    }
    s->ref.nf = lstm ? hid : net.L[cn::kL_mlp2_2].N, s->ref.with_global = net.with_global;  // (LSTM-RL: the hidden width)
    s->ref.ks_x = net.ks_x, s->ref.ks_a = net.ks_a, s->ref.ks_b = net.ks_b, s->ref.ks_c = net.ks_c, s->ref.ks_s = net.ks_s;
    e->sarl = s;
    guard.p = nullptr;
    return CN_OK;
}

int cn_sarl_set_weights(cn_engine* e, const float* const* params_host_array) {
    int rc = bind(e);
    if (rc) return rc;
    cn_sarl* s = e->sarl;
    if (!s) return fail(CN_ERR_INVALID, "cn_sarl_set_weights: call cn_sarl_configure first");
    if (!params_host_array) return fail(CN_ERR_INVALID, "cn_sarl_set_weights: NULL");
    const bool cadrl = s->cfg.model == CN_MODEL_CADRL, lstm = s->cfg.model == CN_MODEL_LSTM_RL;
    const bool pairwise = lstm && s->cfg.interaction_dims[0] > 0;
    s->pack_jobs.n = 0, s->pack_blocks = 0;  // (a call that failed half-way leaves nothing queued)
    for (int i = 0; i < (cadrl ? 8 : (lstm ? (pairwise ? 20 : 12) : 22)); ++i)
        if (!params_host_array[i]) return fail(CN_ERR_INVALID, "cn_sarl_set_weights: parameter %d is NULL", i);
    const float* const* p = params_host_array;  // W0, b0, W1, b1, ... in state_dict order
    cn::SarlNet& net = s->net;
    if (pairwise) {  // ValueNetwork2: mlp1.{0,2,4,6} come first in the state_dict
        const int pre[4] = {cn::kL_mlp2_0, cn::kL_mlp2_2, cn::kL_att_2, cn::kL_att_4};
        for (int i = 0; i < 4; ++i) {
            cn::PackedLinear& L = net.L[pre[i]];
            if ((rc = sarl_pack(e, L, p[2 * i], p[2 * i + 1], L.N, L.K, 0, L.K))) return rc;
        }
        p += 8;
    }
    if (cadrl || lstm) {
        const int head[4] = {cn::kL_mlp3_0, cn::kL_mlp3_2, cn::kL_mlp3_4, cn::kL_mlp3_6};
        for (int i = 0; i < 4; ++i) {
            cn::PackedLinear& L = net.L[head[i]];
            if ((rc = sarl_pack(e, L, p[2 * i], p[2 * i + 1], L.N, L.K, 0, L.K))) return rc;
        }
        if (cadrl && s->reg_mlp) {  // the same parameters as the weight stream of cadrl_reg_kernel
            cn::RegPackPlan plan{};
            plan.xks = s->reg_xks;
            for (int l = 0; l < 4; ++l) {
                const cn::PackedLinear& L = net.L[head[l]];
                cn::RegPackLayer& R = plan.L[l];
                R.W = p[2 * l], R.b = p[2 * l + 1], R.N = L.N, R.K = L.K, R.ldw = L.K, R.k_off = 0, R.replicate = l == 3 ? 1 : 0;
            }
            const int total = cn::reg_total_quads(s->reg_xks) * 256;
            hipLaunchKernelGGL(cn::sarl_reg_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, e->stream, plan, s->reg_stream);
            CN_HIP(hipGetLastError());
        }
        if (lstm) {  // lstm.weight_ih_l0, weight_hh_l0, bias_ih_l0, bias_hh_l0
            cn::PackedLinear& Li = net.L[cn::kL_mlp1_0];
            cn::PackedLinear& Lh = net.L[cn::kL_mlp1_2];
            if ((rc = sarl_pack(e, Li, p[8], p[10], Li.N, Li.K, 0, Li.K)) ||
                (rc = sarl_pack(e, Lh, p[9], p[11], Lh.N, Lh.K, 0, Lh.K)))
                return rc;
        }
        if (lstm && s->reg_mlp) {  // the same parameters as the two weight streams of lstm_reg_kernel
            cn::RegPackPlan gates{}, headp{};
            gates.xks = s->reg_xks, headp.xks = cn::kRegLstmHead;
            cn::RegPackLayer& G = gates.L[0];
            G.W = p[8], G.W2 = p[9], G.b = p[10], G.b2 = p[11];
            G.N = 4 * cn::kRegLstmHid, G.K = net.in_dim, G.ldw = net.in_dim, G.k_split = s->reg_xks - cn::kRegLstmGates;
            float* gate_stream = s->reg_stream;
            if (pairwise) {  // lstm2_reg_kernel: mlp1's stream (the parameters in front of the head's), gates on mlp1's 50 outputs
                cn::RegPackPlan m1{};
                m1.xks = s->reg_xks;
                const int dims[5] = {net.in_dim, s->cfg.interaction_dims[0], s->cfg.interaction_dims[1], s->cfg.interaction_dims[2],
                                     s->cfg.interaction_dims[3]};
                for (int l = 0; l < 4; ++l) {
                    cn::RegPackLayer& R = m1.L[l];
                    R.W = p[2 * l - 8], R.b = p[2 * l - 7], R.N = dims[l + 1], R.K = dims[l], R.ldw = dims[l], R.k_off = 0, R.replicate = 0;
                }
                const int tm = cn::reg_total_quads(s->reg_xks) * 256;
                hipLaunchKernelGGL(cn::sarl_reg_pack_kernel, dim3((tm + 255) / 256), dim3(256), 0, e->stream, m1, s->reg_stream);
                gates.xks = cn::kRegLstmGates + cn::kRegLstmKs;
                G.K = cn::kRegLstmHid, G.ldw = cn::kRegLstmHid, G.k_split = cn::kRegLstmKs;
                gate_stream = s->reg_stream3;
            }
            for (int l = 0; l < 4; ++l) {
                const cn::PackedLinear& L = net.L[head[l]];
                cn::RegPackLayer& R = headp.L[l];
                R.W = p[2 * l], R.b = p[2 * l + 1], R.N = L.N, R.K = L.K, R.ldw = L.K, R.k_off = 0, R.replicate = l == 3 ? 1 : 0;
            }
            const int tg = cn::reg_total_quads(gates.xks) * 256, th = cn::reg_total_quads(cn::kRegLstmHead) * 256;
            hipLaunchKernelGGL(cn::sarl_reg_pack_kernel, dim3((tg + 255) / 256), dim3(256), 0, e->stream, gates, gate_stream);
            hipLaunchKernelGGL(cn::sarl_reg_pack_kernel, dim3((th + 255) / 256), dim3(256), 0, e->stream, headp, s->reg_stream2);
            CN_HIP(hipGetLastError());
        }
        if ((rc = sarl_pack_flush(e))) return rc;
        s->weights_set = true;
        return CN_OK;
    }
    // state_dict layer i -> packed layer(s)
    const int map[11] = {cn::kL_mlp1_0, cn::kL_mlp1_2, cn::kL_mlp2_0, cn::kL_mlp2_2, cn::kL_att0_local, cn::kL_att_2,
                         cn::kL_att_4,  cn::kL_mlp3_0, cn::kL_mlp3_2, cn::kL_mlp3_4, cn::kL_mlp3_6};
    for (int i = 0; i < 11; ++i) {
        cn::PackedLinear& L = net.L[map[i]];
        if (map[i] == cn::kL_att0_local) {
            const int half = L.K;  // mlp1 output width
            const int Ktot = net.with_global ? 2 * half : half;
            if ((rc = sarl_pack(e, L, p[2 * i], p[2 * i + 1], L.N, Ktot, 0, half))) return rc;
            if (net.with_global &&
                (rc = sarl_pack(e, net.L[cn::kL_att0_global], p[2 * i], nullptr, L.N, Ktot, half, half)))
                return rc;
        } else {
            if ((rc = sarl_pack(e, L, p[2 * i], p[2 * i + 1], L.N, L.K, 0, L.K))) return rc;
        }
    }
    if (s->reg_mlp && s->n_chunks) {  // the three weight streams of sarl_reg_chunk_kernel
        // state_dict layer (0..10: mlp1.0, mlp1.2, mlp2.0, mlp2.2, attention.0, .2, .4, mlp3.0, .2, .4, .6) and packed layer of
        // each stream layer
        const auto fill = [&](cn::RegPackLayer& R, int sd, int kl) {
            const cn::PackedLinear& L = net.L[kl];
            R.W = p[2 * sd], R.b = p[2 * sd + 1], R.N = L.N, R.K = L.K, R.ldw = L.K, R.k_off = 0, R.replicate = 0;
        };
        cn::RegPackPlan A{}, B{}, G{};
        A.xks = s->reg_xks, B.xks = cn::kRegChunkB, G.xks = cn::kRegChunkG;
        fill(A.L[0], 0, cn::kL_mlp1_0), fill(A.L[1], 1, cn::kL_mlp1_2);
        if (s->reg_xks == cn::kRegChunkAPre) A.L[0].K = 13;  // the map columns live in om_w
        fill(B.L[0], 2, cn::kL_mlp2_0), fill(B.L[1], 3, cn::kL_mlp2_2), fill(B.L[2], 4, cn::kL_att0_local);
        B.L[2].b = nullptr, B.L[2].ldw = 2 * B.L[2].K;  // attention.0 sees [h2 | mean]
        fill(B.L[3], 5, cn::kL_att_2), fill(B.L[4], 6, cn::kL_att_4), B.L[4].replicate = 1;
        fill(G.L[0], 4, cn::kL_att0_global), G.L[0].ldw = 2 * G.L[0].K, G.L[0].k_off = G.L[0].K;
        fill(G.L[1], 7, cn::kL_mlp3_0), fill(G.L[2], 8, cn::kL_mlp3_2), fill(G.L[3], 9, cn::kL_mlp3_4), fill(G.L[4], 10, cn::kL_mlp3_6);
        G.L[4].replicate = 1;
        const struct { const cn::RegPackPlan* plan; float* dst; } jobs[3] = {{&A, s->reg_stream}, {&G, s->reg_stream2}, {&B, s->reg_stream3}};
        for (const auto& j : jobs) {
            const int total = cn::reg_total_quads(j.plan->xks) * 256;
            hipLaunchKernelGGL(cn::sarl_reg_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, e->stream, *j.plan, j.dst);
        }
        if (s->reg_xks == cn::kRegChunkAPre)
            hipLaunchKernelGGL(cn::sarl_om_weights_kernel, dim3((160 * 49 + 255) / 256), dim3(256), 0, e->stream, p[0], p[1], s->om_w);
        CN_HIP(hipGetLastError());
    } else if (s->reg_mlp) {  // the same parameters as the weight stream of sarl_reg_kernel
        cn::RegPackPlan plan{};
        plan.xks = s->reg_xks;
        const int sd[cn::kRegLayers] = {0, 1, 2, 3, 4, 4, 5, 6, 7, 8, 9, 10};  // state_dict layer of each stream layer
        for (int l = 0; l < cn::kRegLayers; ++l) {
            const int kl[cn::kRegLayers] = {cn::kL_mlp1_0, cn::kL_mlp1_2, cn::kL_mlp2_0, cn::kL_mlp2_2, cn::kL_att0_global,
                                            cn::kL_att0_local, cn::kL_att_2, cn::kL_att_4, cn::kL_mlp3_0, cn::kL_mlp3_2,
                                            cn::kL_mlp3_4, cn::kL_mlp3_6};
            const cn::PackedLinear& L = net.L[kl[l]];
            cn::RegPackLayer& R = plan.L[l];
            R.W = p[2 * sd[l]], R.b = l == cn::kR_att0_local ? nullptr : p[2 * sd[l] + 1];
            R.N = L.N, R.K = L.K, R.ldw = L.K, R.k_off = 0, R.replicate = l == cn::kR_att_4 ? 1 : 0;
            if (l == cn::kR_mlp1_0 && s->reg_xks == cn::kRegSarlPre) R.K = 13;  // the map columns live in om_w
            if (l == cn::kR_att0_local || l == cn::kR_att0_global) R.ldw = 2 * L.K;   // attention.0 sees [h2 | mean]
            if (l == cn::kR_att0_global) R.k_off = L.K;
        }
        const int total = cn::reg_total_quads(s->reg_xks) * 256;
        hipLaunchKernelGGL(cn::sarl_reg_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, e->stream, plan, s->reg_stream);
        if (s->reg_xks == cn::kRegSarlPre)
            hipLaunchKernelGGL(cn::sarl_om_weights_kernel, dim3((160 * 49 + 255) / 256), dim3(256), 0, e->stream, p[0], p[1], s->om_w);
        CN_HIP(hipGetLastError());
    }
    if ((rc = sarl_pack_flush(e))) return rc;
    s->weights_set = true;
    return CN_OK;
}

// The register-resident kernel for 61 inputs (13 rotated features + 48 map cells) reads the occupancy maps from
// `om` itself; X then holds k-steps 0..3 only (sarl_reg_kernel.h).
static bool sarl_om_direct(const cn_sarl* s) {
    return s->reg_mlp && (s->reg_xks == cn::kRegSarlPre || s->reg_xks == cn::kRegChunkAPre) && s->net.in_dim == 61;
}

int cn_sarl_select(cn_engine* e, double* values, int32_t* best, double* action) {
    int rc = bind(e);
    if (rc) return rc;
    cn_sarl* s = e->sarl;
    if (!s || !s->weights_set) return fail(CN_ERR_INVALID, "cn_sarl_select: configure and set weights first");
    if (!best || !action) return fail(CN_ERR_INVALID, "cn_sarl_select: best/action must not be NULL");
    const cn::SarlCfg& C = s->C;
    const int H = C.H;
    // humans' next velocities, once per env (query_env = false: they keep their current ones, no ORCA pass)
    if (!C.const_vel) cn_launch_orca(e, s->orca_vel);
    // the humans' next observable states: their own kernel only where something is built on them per (env, human) — occupancy
    // maps, LSTM-RL's re-ordering; otherwise the feature kernel derives them itself.  The reward of every (env, action) is
    // evaluated inside sarl_select_kernel.  (Each small kernel less is ~7 us of a 70 us single-env decision.)
    if (s->narrow) {  // X never leaves the network kernel's LDS
        if (C.with_om)  // the humans' next states and the map each of them sees: once per (env, human)
            hipLaunchKernelGGL(cn::sarl_lookahead_kernel, dim3((C.B * H + 255) / 256), dim3(256), 0, e->stream, C, e->S.pos,
                               e->S.vel, e->S.rv, s->orca_vel, s->next_obs, s->om);
        cn::SarlDecide D0{};
        D0.in_dim = s->net.in_dim;
        const auto narrow_kernel = s->cfg.model == CN_MODEL_LSTM_RL ? cn::sarl_narrow_kernel<true> : cn::sarl_narrow_kernel<false>;
        hipLaunchKernelGGL(narrow_kernel, dim3((unsigned)s->narrow_tiles), dim3(cn::kNarrowThreads), s->narrow_lds,
                           e->stream, s->ref, C, e->S.pos, e->S.vel, e->S.goal, e->S.rv, e->S.theta, s->actions, s->orca_vel,
                           s->next_obs, s->V, D0, C.with_om ? (const float*)s->om : (const float*)nullptr);
        e->launch_counts[CN_COUNT_SARL_NARROW] += 1;
        hipLaunchKernelGGL(cn::sarl_select_kernel, dim3((C.B + 3) / 4), dim3(256), 0, e->stream, C, e->S.pos, e->S.vel,
                           e->S.goal, e->S.rv, e->S.gtime, e->S.theta, s->actions, s->reward, s->V, values, best, action);
        CN_HIP(hipGetLastError());
        return CN_OK;
    }
    const bool lookahead = C.with_om || C.sort_lookahead;
    if (lookahead)
        hipLaunchKernelGGL(cn::sarl_lookahead_kernel, dim3((C.B * H + 255) / 256), dim3(256), 0, e->stream, C, e->S.pos,
                           e->S.vel, e->S.rv, s->orca_vel, s->next_obs, s->om);
    const size_t rows = s->n_tiles * cn::kSarlGroups * H;
    hipLaunchKernelGGL(cn::sarl_feature_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, e->stream, C,
                       s->net.in_dim, s->net.ks_x, e->S.pos, e->S.goal, e->S.rv, e->S.theta, s->actions, s->next_obs, s->om, s->X,
                       s->n_tiles, s->hcount, sarl_om_direct(s) ? 0 : 1, e->S.vel,
                       lookahead ? (const float*)nullptr : (const float*)s->orca_vel);
    const dim3 grid((unsigned)s->n_tiles), block(cn::kSarlThreads);
    const dim3 pgrid((unsigned)(s->n_tiles < (size_t)s->n_cus ? s->n_tiles : (size_t)s->n_cus));  // persistent: one workgroup per CU
    const int ng = (int)s->n_groups;
#define CN_SARL_MLP(HH)                                                                                              \
    case HH:                                                                                                         \
        if (s->cfg.model == CN_MODEL_CADRL)                                                                          \
            hipLaunchKernelGGL(cn::cadrl_mlp_kernel<HH>, grid, block, s->lds_bytes, e->stream, s->net, s->X, s->V, ng,  \
                               s->hcount);                                                                          \
        else if (s->cfg.model == CN_MODEL_LSTM_RL)                                                                   \
            hipLaunchKernelGGL(cn::lstm_mlp_kernel<HH>, grid, block, s->lds_bytes, e->stream, s->net, s->X, s->V, ng,   \
                               s->hcount);                                                                          \
        else                                                                                                         \
            hipLaunchKernelGGL(cn::sarl_mlp_pipe_kernel<HH>, pgrid, block, s->lds_bytes, e->stream, s->ref, s->X, s->V, \
                               ng, (int)s->n_tiles, s->hcount);                                                      \
        break;
    if (s->reg_mlp) {
        const unsigned wgs = (unsigned)((s->n_tiles + cn::kRegWaves - 1) / cn::kRegWaves);
        // waves per SIMD by the kernels' register counts (of 512; scripts/kernel_resources.py): SARL 1 / 2 humans 153 / 215 ->
        // 3 / 2, which also covers the dependent MFMA chain of the 1-human kernel; CADRL keeps fewer activations alive
        const bool reg_cadrl = s->reg_xks == cn::kRegCadrl;
        const int NTK = reg_cadrl ? s->cadrl_nt : H;  // N tiles per wave of the kernel that runs
        const unsigned per_simd = reg_cadrl ? (NTK == 1 ? 4u : NTK == 2 ? 2u : 1u) : (H == 1 ? 3u : H == 2 ? 2u : 1u);
        const unsigned resident = (unsigned)s->n_cus * per_simd;
        const dim3 rgrid(wgs < resident ? wgs : resident), rblock(cn::kRegWaves * 64);
        const float* om_direct = sarl_om_direct(s) ? s->om : (const float*)nullptr;
        if (s->reg_xks == cn::kRegSarlPre || s->reg_xks == cn::kRegChunkAPre) {
            const int rows = C.B * H;
            hipLaunchKernelGGL(cn::sarl_om_term_kernel, dim3((rows + cn::kOmTermRows - 1) / cn::kOmTermRows),
                               dim3(cn::kOmTermThreads), 0, e->stream, s->om_w, s->om, s->om_term, rows);
            om_direct = s->om_term;
        }
        if (s->n_chunks) {
            const dim3 cgrid(wgs < (unsigned)s->n_cus ? wgs : (unsigned)s->n_cus);
#define CN_SARL_CHUNK(NT, PRE)                                                                                               \
    hipLaunchKernelGGL((cn::sarl_reg_chunk_kernel<NT, PRE>), cgrid, rblock, 0, e->stream, s->reg_stream, s->reg_stream3, s->reg_stream2, \
                       s->X, s->V, s->reg_scratch, ng, (int)s->n_tiles, H, s->n_chunks, s->net.ks_x, s->hcount, om_direct, C.n_actions)
            const bool pre = s->reg_xks == cn::kRegChunkAPre;
            if (s->chunk_nt == 3) {
                if (pre) CN_SARL_CHUNK(3, true);
                else CN_SARL_CHUNK(3, false);
            } else {
                if (pre) CN_SARL_CHUNK(4, true);
                else CN_SARL_CHUNK(4, false);
            }
#undef CN_SARL_CHUNK
        } else if (cn::reg_is_lstm_mlp1(s->reg_xks)) {
            const dim3 lgrid(wgs < (unsigned)s->n_cus ? wgs : (unsigned)s->n_cus);
            if (s->reg_xks == cn::kRegLstmMlp1 + 4)
                hipLaunchKernelGGL(cn::lstm2_reg_kernel<4>, lgrid, rblock, 0, e->stream, s->reg_stream, s->reg_stream3, s->reg_stream2, s->X,
                                   s->V, ng, (int)s->n_tiles, H, s->net.ks_x, s->hcount);
            else
                hipLaunchKernelGGL(cn::lstm2_reg_kernel<16>, lgrid, rblock, 0, e->stream, s->reg_stream, s->reg_stream3, s->reg_stream2, s->X,
                                   s->V, ng, (int)s->n_tiles, H, s->net.ks_x, s->hcount);
        } else if (cn::reg_is_gates(s->reg_xks)) {
            const dim3 lgrid(wgs < (unsigned)s->n_cus ? wgs : (unsigned)s->n_cus);
            if (s->reg_xks == cn::kRegLstmGates + 4)
                hipLaunchKernelGGL(cn::lstm_reg_kernel<4>, lgrid, rblock, 0, e->stream, s->reg_stream, s->reg_stream2, s->X, s->V, ng,
                                   (int)s->n_tiles, H, s->net.ks_x, s->hcount);
            else
                hipLaunchKernelGGL(cn::lstm_reg_kernel<16>, lgrid, rblock, 0, e->stream, s->reg_stream, s->reg_stream2, s->X, s->V, ng,
                                   (int)s->n_tiles, H, s->net.ks_x, s->hcount);
        } else
#define CN_SARL_REG(KERNEL, ...) \
    hipLaunchKernelGGL(KERNEL, rgrid, rblock, 0, e->stream, s->reg_stream, s->X, s->V, ng, (int)s->n_tiles, s->net.ks_x, \
                       s->hcount, ##__VA_ARGS__)
#define CN_SARL_REG_NT(NT)                                                                        \
    case NT:                                                                                      \
        if (reg_cadrl) CN_SARL_REG(cn::cadrl_reg_kernel<NT>, H, s->cadrl_chunks);                 \
        else if (s->reg_xks == 4) CN_SARL_REG((cn::sarl_reg_kernel<4, NT>), om_direct, C.n_actions); \
        else CN_SARL_REG((cn::sarl_reg_kernel<4, NT, true>), om_direct, C.n_actions);             \
        break;
        switch (NTK) {
            CN_SARL_REG_NT(1)
            CN_SARL_REG_NT(2)
            CN_SARL_REG_NT(3)
            CN_SARL_REG_NT(4)
            CN_SARL_REG_NT(5)
        }
#undef CN_SARL_REG_NT
#undef CN_SARL_REG
    } else if (s->chunked && s->cfg.model == CN_MODEL_CADRL) {
        hipLaunchKernelGGL(cn::cadrl_mlp_chunked_kernel<cn::kSarlChunk5>, grid, block, s->lds_bytes, e->stream, s->net, s->X,
                           s->V, ng);
    } else if (s->chunked && s->cfg.model == CN_MODEL_LSTM_RL) {
        hipLaunchKernelGGL(cn::lstm_mlp_anyh_kernel, grid, block, s->lds_bytes, e->stream, s->net, s->X, s->V, ng);
    } else if (s->chunked) {
        hipLaunchKernelGGL(cn::sarl_mlp_chunked_kernel<cn::kSarlChunk>, grid, block, s->lds_bytes, e->stream, s->net, s->X,
                           s->V, ng);
    } else
    switch (H) {
        CN_SARL_MLP(1) CN_SARL_MLP(2) CN_SARL_MLP(3) CN_SARL_MLP(4) CN_SARL_MLP(5) CN_SARL_MLP(6) CN_SARL_MLP(7) CN_SARL_MLP(8)
        default: return fail(CN_ERR_UNSUPPORTED, "value network on device: %d humans", H);
    }
#undef CN_SARL_MLP
    hipLaunchKernelGGL(cn::sarl_select_kernel, dim3((C.B + 3) / 4), dim3(256), 0, e->stream, C, e->S.pos, e->S.vel, e->S.goal,
                       e->S.rv, e->S.gtime, e->S.theta, s->actions, s->reward, s->V, values, best, action);
    CN_HIP(hipGetLastError());
    return CN_OK;
}

int cn_sarl_explore(cn_engine* e, double epsilon, const uint8_t* mask, int32_t* best, double* action,
                    uint8_t* explored) {
    int rc = bind(e);
    if (rc) return rc;
    cn_sarl* s = e->sarl;
    if (!s) return fail(CN_ERR_INVALID, "cn_sarl_explore: cn_sarl_configure first");
    if (!best || !action) return fail(CN_ERR_INVALID, "cn_sarl_explore: best/action must not be NULL");
    if (!(epsilon >= 0.0 && epsilon <= 1.0)) return fail(CN_ERR_INVALID, "cn_sarl_explore: epsilon must be in [0, 1]");
    const cn::SarlCfg& C = s->C;
    hipLaunchKernelGGL(cn::sarl_explore_kernel, dim3((C.B + 63) / 64), dim3(64), 0, e->stream, C.B, C.n_actions, epsilon,
                       e->S.mt_key, e->S.mt_pos, s->actions, mask, best, action, explored, e->C.error);
    CN_HIP(hipGetLastError());
    return CN_OK;
}

int cn_sarl_transform(cn_engine* e, float* out, int64_t env_stride, int sort_humans) {
    int rc = bind(e);
    if (rc) return rc;
    cn_sarl* s = e->sarl;
    if (!s) return fail(CN_ERR_INVALID, "cn_sarl_transform: cn_sarl_configure first");
    if (!out) return fail(CN_ERR_INVALID, "cn_sarl_transform: out is NULL");
    const cn::SarlCfg& C = s->C;
    const int64_t row = (int64_t)C.H * s->net.in_dim;
    if (env_stride == 0) env_stride = row;
    if (env_stride < row) return fail(CN_ERR_INVALID, "cn_sarl_transform: env_stride %lld < %lld", (long long)env_stride, (long long)row);
    hipLaunchKernelGGL(cn::sarl_transform_kernel, dim3((C.B * C.H + 255) / 256), dim3(256), 0, e->stream, C, s->net.in_dim,
                       sort_humans ? 1 : 0, e->S.pos, e->S.vel, e->S.goal, e->S.rv, e->S.theta, out, env_stride);
    CN_HIP(hipGetLastError());
    return CN_OK;
}

int cn_sarl_sample_step(cn_engine* e, double epsilon, uint8_t* alive, int32_t* best, double* action, float* state_out,
                        int64_t env_stride, int sort_humans, double* reward, uint8_t* done, uint8_t* info, double* dmin) {
    const bool was_fresh = e && e->orca_fresh;  // bind() clears it: every entry point but this one invalidates the velocities
    int rc = bind(e);
    if (rc) return rc;
    cn_sarl* s = e->sarl;
    if (!s || !s->weights_set) return fail(CN_ERR_INVALID, "cn_sarl_sample_step: configure and set weights first");
    if (!alive || !best || !action || !reward || !done || !info)
        return fail(CN_ERR_INVALID, "cn_sarl_sample_step: alive, best, action, reward, done and info must not be NULL");
    if (!(epsilon >= 0.0 && epsilon <= 1.0)) return fail(CN_ERR_INVALID, "cn_sarl_sample_step: epsilon must be in [0, 1]");
    if (e->P.robot_orca) return fail(CN_ERR_INVALID, "cn_sarl_sample_step: the robot must be CN_ROBOT_EXTERNAL");
    const cn::SarlCfg& C = s->C;
    const int64_t row = (int64_t)C.H * s->net.in_dim;
    if (env_stride == 0) env_stride = row;
    if (state_out && env_stride < row)
        return fail(CN_ERR_INVALID, "cn_sarl_sample_step: env_stride %lld < %lld", (long long)env_stride, (long long)row);
    const bool fresh = was_fresh;
    if (s->narrow) {
        // Two launches per step in a streamed loop: the value network (its tiles add the lookahead reward and write the replay
        // state), then decision + transition + the humans' ORCA velocities of the NEXT decision (sarl_decide_step_kernel); the
        // first call after anything else touched the engine computes those velocities with a launch of its own.  Workgroup
        // geometries that kernel does not cover (several waves per workgroup, kd bookkeeping) and
        // CROWDNAV_AMD_SARL_FUSED_STEP=0: ORCA, the network with the decision by its last workgroup, the transition.
        const bool fused = s->fused_step;
        if (!C.const_vel && !(fused && fresh)) cn_launch_orca(e, s->orca_vel);
        // occupancy maps: the previous call's sarl_decide_step_kernel left next_obs / om behind its ORCA pass (fresh); otherwise
        // sarl_lookahead_kernel, a launch of its own like ORCA
        if (C.with_om && !(fused && fresh && !C.const_vel))
            hipLaunchKernelGGL(cn::sarl_lookahead_kernel, dim3((C.B * C.H + 255) / 256), dim3(256), 0, e->stream, C, e->S.pos,
                               e->S.vel, e->S.rv, s->orca_vel, s->next_obs, s->om);
        cn::SarlDecide D{};
        D.counter = fused ? nullptr : s->narrow_counter;
        D.epsilon = epsilon, D.alive = alive, D.done = done, D.best = best, D.action = action;
        D.state_out = state_out, D.env_stride = env_stride, D.sort_humans = sort_humans ? 1 : 0, D.in_dim = s->net.in_dim;
        D.reward = s->reward, D.value = s->narrow_value, D.gtime = e->S.gtime, D.mt_key = e->S.mt_key, D.mt_pos = e->S.mt_pos, D.error = e->C.error;
        if (C.with_om) D.next_obs_out = s->next_obs, D.om_out = s->om;
        // the replay-memory states on a workgroup of their own beside the tiles (CROWDNAV_AMD_SARL_SIDE_WG=0: on tile b's idle wave)
        static const bool side = env_int("CROWDNAV_AMD_SARL_SIDE_WG", 1) != 0;
        D.side_wg = (side && state_out) ? 1 : 0;
        const auto narrow_kernel = s->cfg.model == CN_MODEL_LSTM_RL ? cn::sarl_narrow_kernel<true> : cn::sarl_narrow_kernel<false>;
        hipLaunchKernelGGL(narrow_kernel, dim3((unsigned)s->narrow_tiles + (unsigned)D.side_wg), dim3(cn::kNarrowThreads), s->narrow_lds,
                           e->stream, s->ref, C, e->S.pos, e->S.vel, e->S.goal, e->S.rv, e->S.theta, s->actions, s->orca_vel,
                           s->next_obs, s->V, D, C.with_om ? (const float*)s->om : (const float*)nullptr);
        e->launch_counts[CN_COUNT_SARL_NARROW] += 1;
        CN_HIP(hipGetLastError());
        if (fused) {
            cn::StepIo io{action, reward, done, info, dmin, nullptr, nullptr, nullptr, 1};
            float* next_vel = C.const_vel ? nullptr : s->orca_vel;
            const dim3 grid(grid_envs(e)), block(e->P.threads);
#define CN_DECIDE_STEP(MAXL, UNI)                                                                                      \
    hipLaunchKernelGGL((cn::sarl_decide_step_kernel<MAXL, UNI>), grid, block, e->smem, e->stream, e->P, e->S, io, C, D, \
                       s->actions, next_vel)
            if (e->maxl == 5) {
                if (e->P.robot_unicycle) CN_DECIDE_STEP(5, true);
                else CN_DECIDE_STEP(5, false);
            } else {
                if (e->P.robot_unicycle) CN_DECIDE_STEP(10, true);
                else CN_DECIDE_STEP(10, false);
            }
#undef CN_DECIDE_STEP
            CN_HIP(hipGetLastError());
            e->launch_counts[CN_COUNT_SARL_DECIDE_STEPS] += 1;
            e->orca_fresh = next_vel != nullptr;
            return CN_OK;
        }
    } else {
        hipLaunchKernelGGL(cn::sarl_alive_kernel, dim3((C.B + 255) / 256), dim3(256), 0, e->stream, C.B, alive, done);
        if ((rc = cn_sarl_select(e, nullptr, best, action)) || (rc = cn_sarl_explore(e, epsilon, alive, best, action, nullptr)))
            return rc;
        if (state_out && (rc = cn_sarl_transform(e, state_out, env_stride, sort_humans))) return rc;
    }
    return cn_step(e, action, 1, reward, done, info, dmin, nullptr, nullptr, nullptr);
}

int cn_sarl_values(cn_engine* e, const float* states, int64_t n, float* out) {
    int rc = bind(e);
    if (rc) return rc;
    cn_sarl* s = e->sarl;
    if (!s || !s->weights_set) return fail(CN_ERR_INVALID, "cn_sarl_values: configure and set weights first");
    if (!states || !out) return fail(CN_ERR_INVALID, "cn_sarl_values: NULL argument");
    if (n < 1 || (uint64_t)n > (uint64_t)s->n_groups)
        return fail(CN_ERR_INVALID, "cn_sarl_values: %lld states, the engine's tiles hold 1 .. %zu (envs x actions)", (long long)n,
                    s->n_groups);
    const cn::SarlCfg& C = s->C;
    if (!s->narrow || s->net.in_dim != 13 || s->cfg.model == CN_MODEL_CADRL)
        return fail(CN_ERR_UNSUPPORTED, "cn_sarl_values: SARL / LSTM-RL on 13-wide rows at a size that takes the narrow tiles only");
    cn::SarlDecide D{};
    D.x_rows = states, D.ext_groups = (int)n, D.in_dim = s->net.in_dim;
    const int GT = cn::kSarlGroups / C.H;
    const unsigned tiles = (unsigned)((n + GT - 1) / GT);
    const auto narrow_kernel = s->cfg.model == CN_MODEL_LSTM_RL ? cn::sarl_narrow_kernel<true> : cn::sarl_narrow_kernel<false>;
    hipLaunchKernelGGL(narrow_kernel, dim3(tiles), dim3(cn::kNarrowThreads), s->narrow_lds, e->stream, s->ref, C, e->S.pos, e->S.vel,
                       e->S.goal, e->S.rv, e->S.theta, s->actions, s->orca_vel, s->next_obs, out, D, (const float*)nullptr);
    e->launch_counts[CN_COUNT_SARL_NARROW] += 1;
    CN_HIP(hipGetLastError());
    return CN_OK;
}

int cn_sarl_export(cn_engine* e, int which, void* dst, uint64_t bytes) {
    int rc = bind(e);
    if (rc) return rc;
    cn_sarl* s = e->sarl;
    if (!s || !dst) return fail(CN_ERR_INVALID, "cn_sarl_export: not configured or NULL dst");
    const cn::SarlCfg& C = s->C;
    const void* src = nullptr;
    size_t have = 0;
    switch (which) {
        case 0: src = s->reward, have = sizeof(double) * s->n_groups; break;
        case 1: src = s->V, have = sizeof(float) * s->n_groups; break;
        case 2: src = s->next_obs, have = sizeof(double) * C.B * C.H * 5; break;
        case 3: src = s->om, have = sizeof(float) * C.B * C.H * (s->net.in_dim - 13); break;
        case 4:
            if (s->narrow) {  // the network kernel built X in LDS: the same rows, for the caller who asks to see them
                const size_t rows = s->n_tiles * cn::kSarlGroups * C.H;
                hipLaunchKernelGGL(cn::sarl_feature_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, e->stream, C,
                                   s->net.in_dim, s->net.ks_x, e->S.pos, e->S.goal, e->S.rv, e->S.theta, s->actions, s->next_obs,
                                   s->om, s->X, s->n_tiles, s->hcount, 1, e->S.vel, (const float*)s->orca_vel);
                CN_HIP(hipGetLastError());
            }
            if (sarl_om_direct(s)) {
                const size_t rows = s->n_tiles * cn::kSarlGroups * C.H;
                hipLaunchKernelGGL(cn::sarl_om_columns_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, e->stream, C,
                                   s->net.in_dim, s->net.ks_x, s->om, s->X, s->n_tiles);
                CN_HIP(hipGetLastError());
            }
            src = s->X, have = sizeof(float) * s->n_tiles * C.H * s->net.ks_x * 64;
            break;
        default: return fail(CN_ERR_INVALID, "cn_sarl_export: unknown buffer %d", which);
    }
    if (bytes > have) return fail(CN_ERR_INVALID, "cn_sarl_export: buffer %d holds %zu bytes, %llu requested", which, have,
                                 (unsigned long long)bytes);
    CN_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, e->stream));
    return CN_OK;
}

}  // extern "C"

#ifdef CN_PHASE_TIMING
// profiling builds only (scripts/sarl_phase_probe.py)
extern "C" int cn_debug_sarl_cycles(unsigned long long* out16, int reset) {
    if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(cn::cn_sarl_cycles), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long zero[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(cn::cn_sarl_cycles), zero, sizeof(zero)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
