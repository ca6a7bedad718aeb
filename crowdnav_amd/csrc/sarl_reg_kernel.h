// sarl.ValueNetwork.forward (crowd_nav/policy/sarl.py:28-65) with the ACTIVATIONS IN REGISTERS: the shipped widths
// (mlp1 150-100, mlp2 100-50, attention 100-100-1 with the global state, mlp3 150-100-100-1) and 5 humans.
//
// The LDS kernels (sarl_kernels.h) compute Y = X W^T with the activations as the MFMA A operand: a layer's output has to
// travel through LDS to become the next layer's input, every layer ends in a workgroup barrier, and 7 column tiles on 16
// waves leave the MFMA pipes 54 % busy.  Here a wave computes Y^T = W X^T for ITS OWN 16 (env, action) groups x 5 humans
// (5 "N tiles" of 16 rows) through the whole network:
//
//   * A operand = weights (16 output features x 4 inputs), B operand = activations (4 inputs x 16 rows).  The accumulator
//     of v_mfma_f32_16x16x4_f32 holds, in lane l, rows 4*(l>>4) .. +3 of column l&15 — four OUTPUT FEATURES of one ROW —
//     and the B operand of a k-step wants, in lane l, input feature 4*ks + (l>>4) of row l&15.  With the output features of
//     a 16-feature tile dealt to the accumulator rows as  slot m = 4*lg + s  <->  feature 16*t + 4*s + lg  (a 4 x 4
//     transpose, done once in the weight packing), register s of output tile t IS the B operand of k-step 4*t + s of the
//     next layer.  No LDS, no barrier, no data movement between layers: bias is the accumulator's initial value, ReLU one
//     v_max per register.
//   * The reductions over a group's humans (mean for the global state, masked softmax, weighted feature sum) combine the
//     same register of the wave's 5 N tiles: lane-local.
//   * Weights stream from L2 (L1-shared by the 4 waves of a workgroup, which run the same sequence) in the exact order of
//     use as ONE linear array of 1-KiB "quads" (lane l: 4 consecutive k-steps of one output tile, or the bias of a tile),
//     kRegDepth quads ahead, across layer and tile boundaries — a wave never waits for a layer to start.
//   * One wave per SIMD (up to 512 registers: at most 310 activations live — the 80 registers of per-human features wait in
//     wave-private LDS while the attention layers run), 20 independent MFMAs per weight quad.
//
// Arithmetic: each output is the fma chain  bias, then inputs in ascending k  (the MFMA's own order) — the same products as
// the LDS kernels, which add the bias last; both within 1e-6 of torch (tests/test_sarl.py).
#pragma once

namespace cn {

constexpr int kRegDepth = 4;   // weight quads in flight per wave (RegStreamT<6> in the 5-human SARL kernels: round 3, below)
constexpr int kRegPad = 12;    // a stream is a whole number of 4- AND 6-quad groups: one packing serves both depths
constexpr int kRegHumans = 5;  // N tiles per wave, at most
constexpr int kRegWaves = 4;   // waves per workgroup = SIMDs per CU

enum {
    kR_mlp1_0, kR_mlp1_2, kR_mlp2_0, kR_mlp2_2, kR_att0_global, kR_att0_local, kR_att_2, kR_att_4,
    kR_mlp3_0, kR_mlp3_2, kR_mlp3_4, kR_mlp3_6, kRegLayers
};

struct RegShape {
    int ks;      // k-steps (inputs / 4, rounded up)
    int mt;      // output tiles of 16 features
    int bias;    // 1: a bias quad opens every output tile (0: the accumulator starts from another layer's result)
    int paired;  // 1: one N tile only — two output tiles advance together so that consecutive MFMAs are independent
};

__host__ __device__ constexpr int reg_cdiv(int a, int b) { return (a + b - 1) / b; }

// The functions below describe a weight stream by a NETWORK KEY: for sarl.ValueNetwork the k-steps of its input, 4 (13
// features) or 16 (13 + 48 occupancy-map features); kRegCadrl for cadrl.ValueNetwork (cadrl.py:22-29: 13 -> 150 -> 100 ->
// 100 -> 1 on every (robot, human) row; layers 0..3).
// kRegSarlPre: sarl.ValueNetwork on 61 inputs with the occupancy-map half of mlp1.0 hoisted out of the action loop — the 48 map
// features of (env, human) are the same for all 81 actions, so  b + W[:, 13:61] om  is computed once per (env, human)
// (sarl_om_term_kernel) and mlp1.0's accumulators START from it: 4 k-steps instead of 16, no bias quad.
constexpr int kRegSarlPre = 5;
constexpr int kRegCadrl = 1004;
// lstm_rl.ValueNetwork1 (lstm_rl.py:9-33) is two streams: the LSTM cell's gate layer (one layer, re-read for every human:
// kRegLstmGates + k-steps of the input) and the value head on [self (6) | h_n (50)] (4 layers).
constexpr int kRegLstmGates = 2000, kRegLstmHead = 2100, kRegLstmHid = 50, kRegLstmKs = 13;  // 13 k-steps / output tiles of 50 units
__host__ __device__ constexpr bool reg_is_gates(int key) { return key > kRegLstmGates && key < kRegLstmHead; }
// lstm_rl.ValueNetwork2 (lstm_rl.py:36-66, with_interaction_module = true at the shipped widths mlp1_dims = 150, 100, 100, 50):
// every human's row passes mlp1 in front of the LSTM — a third stream (kRegLstmMlp1 + k-steps of the input: 4 layers, the last
// without ReLU, re-read for every human like the gate layer), and the gate layer's input is mlp1's 50 outputs (13 k-steps).
constexpr int kRegLstmMlp1 = 2200;
__host__ __device__ constexpr bool reg_is_lstm_mlp1(int key) { return key > kRegLstmMlp1 && key < kRegLstmMlp1 + 100; }
// sarl.ValueNetwork for MORE than 5 humans (sarl_reg_chunk_kernel): the humans pass in chunks of NT, so the network is three
// streams — A: mlp1.0, mlp1.2 (re-read per chunk, first pass; APre: mlp1.0 starts from the hoisted occupancy-map term),
// G: attention.0's global half, then — after the second pass — the value head (read once per tile, in this order),
// B: mlp2.0, mlp2.2, attention.0 local, attention.2, attention.4 (re-read per chunk, second pass).
constexpr int kRegChunkA = 3001, kRegChunkAPre = 3002, kRegChunkB = 3003, kRegChunkG = 3004;
__host__ __device__ constexpr int reg_layers(int key) {
    return key == kRegCadrl || key == kRegLstmHead || reg_is_lstm_mlp1(key) ? 4
           : reg_is_gates(key)                       ? 1
           : key == kRegChunkA || key == kRegChunkAPre ? 2
           : key == kRegChunkB || key == kRegChunkG    ? 5
                                                       : (int)kRegLayers;
}

__host__ __device__ constexpr RegShape reg_shape(int l, int xks) {
    if (reg_is_gates(xks)) return {xks - kRegLstmGates + kRegLstmKs, kRegLstmKs, 1, 0};  // [x_t | h_(t-1)] -> i, f, g, o of 4 units per tile
    if (reg_is_lstm_mlp1(xks)) {
        switch (l) {
            case 0: return {xks - kRegLstmMlp1, 10, 1, 0};
            case 1: return {38, 7, 1, 0};
            case 2: return {25, 7, 1, 0};
            default: return {25, 4, 1, 0};  // 50 outputs: k-steps 0..12 of the gate layer (units 50..63 of the last tile are zero)
        }
    }
    if (xks == kRegChunkA || xks == kRegChunkAPre)
        return l == 0 ? (xks == kRegChunkAPre ? RegShape{4, 10, 0, 0} : RegShape{4, 10, 1, 0}) : RegShape{38, 7, 1, 0};
    if (xks == kRegChunkB) {
        switch (l) {
            case 0: return {25, 7, 1, 0};   // mlp2.0
            case 1: return {25, 4, 1, 0};   // mlp2.2
            case 2: return {25, 7, 0, 0};   // attention.0, local half: starts from the global term
            case 3: return {25, 7, 1, 0};   // attention.2
            default: return {25, 1, 1, 0};  // attention.4
        }
    }
    if (xks == kRegChunkG) {
        switch (l) {
            case 0: return {25, 7, 1, 1};   // attention.0, global half (+ the layer's bias)
            case 1: return {15, 10, 1, 1};  // mlp3.0 on [weighted feature | self], as kR_mlp3_0
            case 2: return {38, 7, 1, 1};
            case 3: return {25, 7, 1, 1};
            default: return {25, 1, 1, 1};
        }
    }
    if (xks == kRegCadrl || xks == kRegLstmHead) {
        switch (l) {
            case 0: return {xks == kRegCadrl ? 4 : 15, 10, 1, 0};
            case 1: return {38, 7, 1, 0};
            case 2: return {25, 7, 1, 0};
            default: return {25, 1, 1, 0};
        }
    }
    switch (l) {
        case kR_mlp1_0: return xks == kRegSarlPre ? RegShape{4, 10, 0, 0} : RegShape{xks, 10, 1, 0};
        case kR_mlp1_2: return {38, 7, 1, 0};
        case kR_mlp2_0: return {25, 7, 1, 0};
        case kR_mlp2_2: return {25, 4, 1, 0};
        case kR_att0_global: return {25, 7, 1, 1};
        case kR_att0_local: return {25, 7, 0, 0};
        case kR_att_2: return {25, 7, 1, 0};
        case kR_att_4: return {25, 1, 1, 0};
        case kR_mlp3_0: return {15, 10, 1, 1};  // 12.5 k-steps of weighted feature + the self state from the X registers
        case kR_mlp3_2: return {38, 7, 1, 1};
        case kR_mlp3_4: return {25, 7, 1, 1};
        default: return {25, 1, 1, 1};          // kR_mlp3_6
    }
}
__host__ __device__ constexpr int reg_tile_quads(int l, int xks) { return reg_shape(l, xks).bias + reg_cdiv(reg_shape(l, xks).ks, 4); }
__host__ __device__ constexpr int reg_layer_quads(int l, int xks) { return reg_shape(l, xks).mt * reg_tile_quads(l, xks); }
__host__ __device__ constexpr int reg_qbase(int l, int xks) {
    int q = 0;
    for (int i = 0; i < l; ++i) q += reg_layer_quads(i, xks);
    return q;
}
// the stream is padded to a multiple of the queue depth (of every depth in use: kRegPad) so that quad I always lives in
// register slot I % depth
__host__ __device__ constexpr int reg_total_quads(int xks) {
    return reg_cdiv(reg_qbase(reg_layers(xks), xks), kRegPad) * kRegPad;
}
// position of item j (0 = bias if the layer has one, then the weight quads) of output tile mt inside its layer
__host__ __device__ constexpr int reg_qpos(int l, int xks, int mt, int j) {
    const RegShape s = reg_shape(l, xks);
    const int S = reg_tile_quads(l, xks);
    if (!s.paired || mt >= (s.mt / 2) * 2) return mt * S + j;
    return (mt / 2) * 2 * S + 2 * j + (mt & 1);
}
// input column of W that k-step ks, lane group lg multiplies (-1: none).  mlp3.0 reads joint = [self (6) | weighted feature]
// (sarl.py:61): k-steps 0..11 and lanes 0..31 of k-step 12 are the weighted feature's registers; lanes 32..63 of k-step 12
// take self features 2, 3 from X k-step 0, k-step 13 takes 4, 5 from X k-step 1 and k-step 14 takes 0, 1 from X k-step 0 —
// the registers that already hold them in those lanes.
// The LSTM-RL head reads joint = [self (6) | h_n (50)] (lstm_rl.py:30-31): k-steps 0..12 are the hidden state's registers
// (unit 4 ks + lg), k-step 13 is X k-step 0 of the first human's row (self features 0..3), k-step 14 its k-step 1 (4, 5).
__host__ __device__ constexpr int reg_kcol(int key, int l, int K, int k_off, int ks, int lg) {
    if (key == kRegLstmHead && l == 0)
        return ks < kRegLstmKs ? (4 * ks + lg < kRegLstmHid ? 6 + 4 * ks + lg : -1) : ks == 13 ? lg : lg < 2 ? 4 + lg : -1;
    if ((key >= 1000 && !(key == kRegChunkG && l == 1)) || (key < 1000 && l != kR_mlp3_0))
        return 4 * ks + lg < K ? k_off + 4 * ks + lg : -1;
    if (ks < 12) return 6 + 4 * ks + lg;
    if (ks == 12) return lg < 2 ? 6 + 48 + lg : lg;
    if (ks == 13) return lg < 2 ? 4 + lg : -1;
    return lg < 2 ? lg : -1;
}

struct RegPackLayer {
    const float* W;     // torch.nn.Linear weight [N][ldw]
    const float* b;     // bias or nullptr
    int N, ldw, K, k_off, replicate;  // replicate: the single output feature fills all 16 slots (attention score: every lane gets it)
    // LSTM gate layer (W = weight_ih [4 hid][K], b = bias_ih): the recurrent half; k-steps from k_split on multiply h
    const float* W2;    // weight_hh [4 hid][hid]
    const float* b2;    // bias_hh
    int k_split;
};
struct RegPackPlan {
    RegPackLayer L[kRegLayers];
    int xks;  // the network key
};

__global__ void sarl_reg_pack_kernel(RegPackPlan plan, float* stream) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int xks = plan.xks;
    const int total = reg_total_quads(xks);
    if (idx >= total * 256) return;
    const int quad = idx >> 8, lane = (idx >> 2) & 63, kk = idx & 3;
    const int lg = lane >> 4, m = lane & 15;
    float v = 0.0f;
    int l = 0, base = 0;
    const int n_layers = reg_layers(xks);
    while (l < n_layers && quad >= base + reg_layer_quads(l, xks)) base += reg_layer_quads(l++, xks);
    if (l < n_layers) {
        const RegShape s = reg_shape(l, xks);
        const int S = reg_tile_quads(l, xks);
        const RegPackLayer& L = plan.L[l];
        const int r = quad - base;
        int mt, j;
        const int pairs = s.paired ? s.mt / 2 : 0;
        if (r < pairs * 2 * S) {
            const int rr = r % (2 * S);
            mt = 2 * (r / (2 * S)) + (rr & 1), j = rr >> 1;
        } else {
            const int r2 = r - pairs * 2 * S;
            mt = 2 * pairs + r2 / S, j = r2 % S;
        }
        if (reg_is_gates(xks)) {
            // output tile mt = units 4 mt .. 4 mt + 3: accumulator register kk of lane group lg = gate kk (torch order i, f, g, o)
            // of unit 4 mt + lg, so that a tile's f32x4 is everything the cell update of its 4 units needs and the new hidden
            // state lands in the B-operand layout of k-step mt
            if (j == 0) {
                const int u = 4 * mt + lg;
                v = u < kRegLstmHid ? L.b[kk * kRegLstmHid + u] + L.b2[kk * kRegLstmHid + u] : 0.0f;
            } else {
                const int ks = 4 * (j - 1) + kk, u = 4 * mt + (m >> 2), n = (m & 3) * kRegLstmHid + u;
                if (u < kRegLstmHid && ks < s.ks) {
                    if (ks < L.k_split) v = 4 * ks + lg < L.K ? L.W[(size_t)n * L.ldw + 4 * ks + lg] : 0.0f;
                    else v = 4 * (ks - L.k_split) + lg < kRegLstmHid ? L.W2[(size_t)n * kRegLstmHid + 4 * (ks - L.k_split) + lg] : 0.0f;
                }
            }
        } else if (s.bias && j == 0) {  // accumulator register kk of lane group lg = feature 16 mt + 4 kk + lg
            const int f = L.replicate ? 0 : 16 * mt + 4 * kk + lg;
            v = (L.b && f < L.N) ? L.b[f] : 0.0f;
        } else {
            const int ks = 4 * (j - s.bias) + kk;
            const int n = L.replicate ? 0 : 16 * mt + 4 * (m & 3) + (m >> 2);
            const int col = ks < s.ks ? reg_kcol(xks, l, L.K, L.k_off, ks, lg) : -1;
            v = (n < L.N && col >= 0) ? L.W[(size_t)n * L.ldw + col] : 0.0f;
        }
    }
    stream[idx] = v;
}

typedef const f32x4 __attribute__((address_space(1))) * gf32x4_p;

// Quad J is read as  buffer_load_dwordx4 v, voff, s[rsrc], soffset = J KiB offen : a buffer resource (4 SGPRs) over the
// stream, the lane offset in ONE VGPR that never changes, the quad's offset a scalar constant.  No per-quad address in
// vector registers: with global_load the 546 addresses were either hoisted out of the tile loop (474 of them spilled to
// scratch in the first build) or rebuilt as 64-bit vector adds whose destination the register allocator, out of VGPRs, put
// on top of loads still in flight (s_waitcnt vmcnt(0): the whole prefetch queue drained, several times per tile).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// DEPTH quads in flight.  4 everywhere but in the 5-human SARL kernels, which have the registers for 6 (round 3: the one-N-tile
// layers — attention.0's global half, the value head — consume a quad every 128 cycles, and 4 in flight cover 512 cycles of L2
// latency: cn_sarl_select 1.868 -> 1.838 ms, with occupancy maps 1.922 -> 1.907, 8 no better than 6).
template <int DEPTH>
struct RegStreamT {
    static constexpr int kDepth = DEPTH;
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t voff;     // lane * 16
    f32x4 q[DEPTH];    // quads I .. I + DEPTH - 1 of the running position (quad J in slot J % DEPTH)
};
typedef RegStreamT<kRegDepth> RegStream;
template <class WS>
__device__ __forceinline__ f32x4 reg_quad(const WS& s, int J) {
    // 4 quads share one scalar offset (the other 2 address bits go into the instruction's 12-bit immediate)
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(s.rsrc, s.voff + (J & 3) * 1024, (J >> 2) * 4096, 0));
}
template <int QT, class WS>
__device__ __forceinline__ f32x4 reg_take(WS& s, int I) {
    static_assert(QT % WS::kDepth == 0, "slot I % depth must survive the wrap of the stream");
    const f32x4 v = s.q[I % WS::kDepth];
    s.q[I % WS::kDepth] = reg_quad(s, (I + WS::kDepth) % QT);
    return v;
}

__device__ __forceinline__ f32x4 reg_relu(f32x4 v) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // max(x, 0) on the bit pattern: negative floats are negative integers (-0.0 too) — ONE
        const float f = v[i];                      // v_max_i32; fmaxf would quiet signalling NaNs first (a second v_max_f32)
        const int b = __float_as_int(f);           // (bit_cast straight on the vector element reads element 0 for every i)
        v[i] = __int_as_float(b > 0 ? b : 0);
    }
    return v;
}

// emit(nt, mt, act(W x + b)) for NT N tiles and every output tile mt; in(nt, ks) = the B operand of k-step ks; init(mt) =
// accumulator start when the layer has no bias quad
template <int XKS, int L, int NT, bool RELU, class WS, class In, class Init, class Emit>
__device__ __forceinline__ void reg_dense(WS& ws, In in, Init init, Emit emit) {
    constexpr RegShape S = reg_shape(L, XKS);
    constexpr int QT = reg_total_quads(XKS), QB = reg_qbase(L, XKS), KQ = reg_cdiv(S.ks, 4);
    static_assert(!S.paired, "use reg_dense1");
    // Output tile mt - 1 leaves the accumulators (AGPR -> VGPR, ReLU, or the LDS store) after the first MFMAs of tile mt have
    // issued, so that it never waits for the matrix pipe to drain (two accumulator sets).
    f32x4 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < S.mt; ++mt) {
        f32x4 c0;
        if constexpr (S.bias != 0) c0 = reg_take<QT>(ws, QB + reg_qpos(L, XKS, mt, 0));
        else c0 = init(mt);
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            const f32x4 a = reg_take<QT>(ws, QB + reg_qpos(L, XKS, mt, S.bias + q));
            __builtin_amdgcn_sched_barrier(0);  // the request for quad I + kRegDepth is issued HERE, not sunk to its use
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int ks = 4 * q + kk;
                if (ks < S.ks) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt & 1][nt] =
                            __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk], in(nt, ks), ks == 0 ? c0 : acc[mt & 1][nt], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (q == 0 && mt > 0) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) emit(nt, mt - 1, RELU ? reg_relu(acc[(mt - 1) & 1][nt]) : acc[(mt - 1) & 1][nt]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
        emit(nt, S.mt - 1, RELU ? reg_relu(acc[(S.mt - 1) & 1][nt]) : acc[(S.mt - 1) & 1][nt]);
    __builtin_amdgcn_sched_barrier(0);
}

// A layer with ONE output (a value head, an attention score) on the vector ALUs: as a 16-wide MFMA tile it executes 16 x the
// multiply-adds it needs — 25 k-steps x NT matrix instructions of 32 cycles each for 100 inputs — and the FP32 MFMA runs at the
// vector rate, so nothing is gained by keeping it on the matrix side.  The stream's quads are the same (`replicate` packing: lane
// l's A fragment of k-step ks IS the weight of the input its own B fragment holds — feature 4 ks + (l >> 4) of row l & 15): every
// lane multiplies its 25 inputs by its 25 weights (v_fma_f32), the four lane groups of a row are summed through two cross-lane
// exchanges, the bias goes on last.  Every lane of a row ends with the row's value, as with the replicated tile.  Not the MFMA's
// summation order: ~1e-7 of it (tests: 1e-6 of the reference).
template <int XKS, int L, int NT, class WS, class In, class Emit>
__device__ __forceinline__ void reg_dense_valu1(WS& ws, In in, Emit emit) {
    constexpr RegShape S = reg_shape(L, XKS);
    constexpr int QT = reg_total_quads(XKS), QB = reg_qbase(L, XKS), KQ = reg_cdiv(S.ks, 4);
    static_assert(S.mt == 1 && S.bias && !S.paired, "one replicated output tile with a bias quad");
    const f32x4 c0 = reg_take<QT>(ws, QB + reg_qpos(L, XKS, 0, 0));
    float acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = 0.0f;
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
        const f32x4 a = reg_take<QT>(ws, QB + reg_qpos(L, XKS, 0, 1 + q));
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int ks = 4 * q + kk;
            if (ks < S.ks) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_fmaf(a[kk], in(nt, ks), acc[nt]);
            }
        }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        float v = acc[nt];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        v += c0[0];
        emit(nt, 0, f32x4{v, v, v, v});
    }
}

// ReLU of one value; AG: the result goes to an AGPR (the asm's "=a" operand is what keeps the array in the accumulator half
// of the register file, where the next layer's MFMAs read it as their B operand directly — arrays the compiler moves there
// on its own are copied back with v_accvgpr_read before every use).
template <bool AG>
__device__ __forceinline__ float reg_relu1(float x) {
    const int b = __float_as_int(x);
    x = __int_as_float(b > 0 ? b : 0);  // v_max_i32: negative floats are negative integers
    if constexpr (AG) {
        float r;
        asm("v_accvgpr_write_b32 %0, %1" : "=a"(r) : "v"(x));
        return r;
    }
    return x;
}

// out[nt][mt] = relu(W x + b).  Every instruction a wave issues between two MFMAs delays the second one by ~5 cycles (one wave
// per SIMD: nothing else hides it; measured: tile time = 32 cycles x MFMAs + 5.4 x everything else — and WHERE it sits does
// not matter: round 3 spread the ReLU burst of tile mt - 1 one value per MFMA gap over tile mt's first 20 MFMAs with
// sched_group_barrier(MFMA 1, VALU 3) — gap histogram 62-instruction bursts -> 3..7 per gap — and cn_sarl_select went
// 1.834 -> 1.839 ms; -mllvm -amdgpu-mfma-vgpr-form removes the 802 v_accvgpr_read per tile and gains 0.8 %), so the layer is written
// for instruction count: a VGPR array (AG = false) is accumulated in place and rectified with one v_max_i32 per value; an
// AGPR array is accumulated in VGPRs, rectified there and moved with one v_accvgpr_write (2 per value; read - max - write on
// an AGPR accumulator would be 3).  The ReLU of tile mt - 1 follows the first MFMAs of tile mt, so that it never waits for
// the matrix pipe to drain.
template <int XKS, int L, int NT, bool AG, class WS, class In, class Init, int MT = reg_shape(L, XKS).mt>
__device__ __forceinline__ void reg_dense_arr(WS& ws, In in, Init init, f32x4 (&out)[NT][MT]) {
    constexpr RegShape S = reg_shape(L, XKS);
    constexpr int QT = reg_total_quads(XKS), QB = reg_qbase(L, XKS), KQ = reg_cdiv(S.ks, 4);
    static_assert(!S.paired && MT == S.mt && S.ks >= 4, "use reg_dense1");
    constexpr bool kPerTileInit = std::is_invocable_v<Init, int, int>;  // init(nt, mt): a start value per N tile
    f32x4 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        f32x4 c0[kPerTileInit ? NT : 1];
        if constexpr (S.bias != 0) c0[0] = reg_take<QT>(ws, QB + reg_qpos(L, XKS, mt, 0));
        else if constexpr (kPerTileInit) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) c0[nt] = init(nt, mt);
        } else c0[0] = init(mt);
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            const f32x4 a = reg_take<QT>(ws, QB + reg_qpos(L, XKS, mt, S.bias + q));
            __builtin_amdgcn_sched_barrier(0);  // the request for quad I + kRegDepth is issued HERE, not sunk to its use
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int ks = 4 * q + kk;
                if (ks < S.ks) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt & 1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                            a[kk], in(nt, ks), ks == 0 ? c0[kPerTileInit ? nt : 0] : acc[mt & 1][nt], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (q == 0 && mt > 0) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) out[nt][mt - 1][i] = reg_relu1<AG>(acc[(mt - 1) & 1][nt][i]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) out[nt][MT - 1][i] = reg_relu1<AG>(acc[(MT - 1) & 1][nt][i]);
    __builtin_amdgcn_sched_barrier(0);
}

// one N tile: output tiles two at a time (their MFMAs alternate: a dependent v_mfma_f32_16x16x4_f32 issues after 40 cycles,
// an independent one after 32)
template <int XKS, int L, bool RELU, class WS, class In, int MT = reg_shape(L, XKS).mt>
__device__ __forceinline__ void reg_dense1(WS& ws, In in, f32x4 (&out)[MT]) {
    constexpr RegShape S = reg_shape(L, XKS);
    constexpr int QT = reg_total_quads(XKS), QB = reg_qbase(L, XKS), KQ = reg_cdiv(S.ks, 4);
    static_assert(S.paired && S.bias && MT == S.mt, "use reg_dense");
#pragma unroll
    for (int p = 0; p < MT / 2; ++p) {
        const f32x4 c0 = reg_take<QT>(ws, QB + reg_qpos(L, XKS, 2 * p, 0));
        const f32x4 c1 = reg_take<QT>(ws, QB + reg_qpos(L, XKS, 2 * p + 1, 0));
        f32x4 acc0, acc1;
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            const f32x4 a0 = reg_take<QT>(ws, QB + reg_qpos(L, XKS, 2 * p, 1 + q));
            const f32x4 a1 = reg_take<QT>(ws, QB + reg_qpos(L, XKS, 2 * p + 1, 1 + q));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int ks = 4 * q + kk;
                if (ks < S.ks) {
                    const float x = in(ks);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[kk], x, ks == 0 ? c0 : acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[kk], x, ks == 0 ? c1 : acc1, 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        out[2 * p] = RELU ? reg_relu(acc0) : acc0, out[2 * p + 1] = RELU ? reg_relu(acc1) : acc1;
        __builtin_amdgcn_sched_barrier(0);
    }
    if (MT & 1) {
        constexpr int mt = MT - 1;
        const f32x4 c0 = reg_take<QT>(ws, QB + reg_qpos(L, XKS, mt, 0));
        f32x4 acc;
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            const f32x4 a = reg_take<QT>(ws, QB + reg_qpos(L, XKS, mt, 1 + q));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int ks = 4 * q + kk;
                if (ks < S.ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk], in(ks), ks == 0 ? c0 : acc, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        out[mt] = RELU ? reg_relu(acc) : acc;
        __builtin_amdgcn_sched_barrier(0);
    }
}

// X: the feature kernel's fragment order, X[((tile * 5 + h) * ks_x + ks) * 64 + lane] = feature 4 ks + (lane >> 4) of human h
// of group lane & 15 — exactly the B operand of k-step ks.  V[group] out.  Persistent: wave w of the grid takes tiles
// w, w + waves, ...; the next tile's X is requested while the value head of the current one runs.
// om (XKS = 16 only; nullptr: everything comes from X): the occupancy maps [env][human][48] of the lookahead kernel.  They do
// not depend on the action, so instead of 81 copies inside X (425 MB written by the feature kernel and read back here per
// decision, 48 of every 61 floats) k-steps 4..15 are read from the maps themselves: lane l wants feature 4 ks + (l >> 4) of
// group l & 15, i.e. map value 4 ks + (l >> 4) - 13 of the group's env — 16 consecutive groups are one or two envs, so the
// loads of a k-step hit one or two cache lines.
// NT = humans of the crowd (N tiles per wave), 1..5.  With NT = 1 the MFMAs of a k-step chain are dependent (40 instead of 32
// cycles each); from 2 humans on consecutive MFMAs alternate between accumulators.
// PRE (XKS = 4): `om` is the term of sarl_om_term_kernel, [env][human][10 tiles][4 lane groups][4 registers] — mlp1.0's
// accumulator for (row, output tile) in the lane's own order, one 16-byte load each, requested one output tile ahead.
template <int XKS, int NT, bool PRE = false>
__global__ __launch_bounds__(kRegWaves * 64) void sarl_reg_kernel(const float* stream, const float* X, float* V, int n_groups,
                                                                  int n_tiles, int ks_x, const int* hcount,
                                                                  const float* om = nullptr, int n_actions = 1) {
    static_assert(NT >= 1 && NT <= kRegHumans, "the activations of at most 5 humans fit the register file");
    static_assert(!PRE || XKS == 4, "the hoisted occupancy-map term replaces k-steps 4..15");
    constexpr int KEY = PRE ? kRegSarlPre : XKS;
    constexpr int QT = reg_total_quads(KEY);
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * kRegWaves + (threadIdx.x >> 6), nw = gridDim.x * kRegWaves;
    RegStreamT<(NT == kRegHumans ? 6 : kRegDepth)> ws;
    ws.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(stream), 0, QT * 1024, 0x00020000);  // raw, 32-bit elements
    ws.voff = (uint32_t)lane * 16u;
#pragma unroll
    for (int i = 0; i < ws.kDepth; ++i) ws.q[i] = reg_quad(ws, i);
    if (wid >= n_tiles) return;
    const gfloat_p Xg = as_global(X) + lane;
    float x[NT][XKS];
    int cnt;
    const bool om_direct = XKS == 16 && om != nullptr;
    const auto load_x = [&](int t) {
        const gfloat_p xt = Xg + (size_t)t * NT * ks_x * 64;
        gfloat_p ob = as_global(om);
        if (om_direct) {
            const long long G = (long long)t * kSarlGroups + (lane & 15);
            const int env = (int)((G < n_groups ? G : (long long)n_groups - 1) / n_actions);
            ob += (size_t)env * NT * 48 + (lane >> 4);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int ks = 0; ks < XKS; ++ks) {
                if (om_direct && ks >= 4)  // map value 4 ks + (lane >> 4) - 13; 61..63 do not exist (k-step 15, lanes 16..63)
                    x[nt][ks] = (ks < 15 || lane < 16) ? ob[nt * 48 + 4 * ks - 13] : 0.0f;
                else
                    x[nt][ks] = xt[(nt * ks_x + ks) * 64];
            }
    };
    load_x(wid < n_tiles ? wid : 0);
    cnt = hcount[(size_t)(wid < n_tiles ? wid : 0) * kSarlGroups + (lane & 15)];
    // PRE: mlp1.0's start values, output tile mt of N tile nt in pre[mt][nt] — all 10 x NT of the NEXT group tile are requested
    // with its X, when the value head starts and 200 registers fall free (one output tile ahead was too late: 20 MFMAs are 640
    // cycles, a loaded L2 answers later: 1.964 ms instead of 1.88)
    // Buffer loads as for the weight stream: ONE lane offset (the env's rows, the lane group's 16 bytes), the (nt, mt) part in the
    // instruction's immediate — with global_load the 50 64-bit addresses were 350 VALU instructions per tile.
    f32x4 pre[PRE ? 10 : 1][PRE ? NT : 1];
    __amdgpu_buffer_rsrc_t prsrc = ws.rsrc;
    if constexpr (PRE)
        prsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(om), 0, (int)(((long long)n_groups / n_actions) * NT * 640), 0x00020000);
    const auto load_pre = [&](int t) {
        const long long G = (long long)t * kSarlGroups + (lane & 15);
        const int env = (int)((G < n_groups ? G : (long long)n_groups - 1) / n_actions);
        const uint32_t voff = (uint32_t)env * (NT * 640u) + (uint32_t)(lane >> 4) * 16u;
#pragma unroll
        for (int mt = 0; mt < 10; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                pre[mt][nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prsrc, voff + (nt * 40 + mt * 4) * 16, 0, 0));
    };
    if constexpr (PRE) load_pre(wid);
    // per-human features (mlp2 output, 80 registers) wait in LDS while the attention layers run: wave-private, no barrier
    __shared__ f32x4 park[kRegWaves][NT * 4][64];
    f32x4(*const mypark)[64] = park[threadIdx.x >> 6];
    const auto none = [](int) { return f32x4{0.0f, 0.0f, 0.0f, 0.0f}; };
    CN_SARL_CLOCK_BEGIN();
    for (int tile = wid; tile < n_tiles; tile += nw) {
        const float self0 = x[0][0], self1 = x[0][1];  // features 0..3 / 4..7 of human 0's row: the self state lives in 0..5
        f32x4 att[NT][7];
        {
            f32x4 h2[NT][7];
            {
                f32x4 h1[NT][10];
                if constexpr (PRE)
                    reg_dense_arr<KEY, kR_mlp1_0, NT, true>(
                        ws, [&](int nt, int ks) { return x[nt][ks]; },
                        [&](int nt, int mt) { return pre[mt][nt]; },
                        h1);
                else
                    reg_dense_arr<KEY, kR_mlp1_0, NT, true>(ws, [&](int nt, int ks) { return x[nt][ks]; }, none, h1);
                CN_SARL_TICK(1);
                reg_dense_arr<KEY, kR_mlp1_2, NT, false>(ws, [&](int nt, int ks) { return h1[nt][ks >> 2][ks & 3]; }, none, h2);
                CN_SARL_TICK(2);
            }
            {
                f32x4 t1[NT][7];
                reg_dense_arr<KEY, kR_mlp2_0, NT, true>(ws, [&](int nt, int ks) { return h2[nt][ks >> 2][ks & 3]; }, none, t1);
                CN_SARL_TICK(3);
                reg_dense<KEY, kR_mlp2_2, NT, false>(ws, [&](int nt, int ks) { return t1[nt][ks >> 2][ks & 3]; }, none,
                                                     [&](int nt, int mt, f32x4 v) { mypark[nt * 4 + mt][lane] = v; });
                CN_SARL_TICK(4);
            }
            // global state: mean over the humans present (sarl.py:42), elementwise over the 5 N tiles
            f32x4 gterm[7];
            {
                f32x4 gm[7];
                const float fc = (float)cnt;
#pragma unroll
                for (int t = 0; t < 7; ++t) {
                    f32x4 sum = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int i = 0; i < 4; ++i) sum[i] += nt < cnt ? h2[nt][t][i] : 0.0f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) gm[t][i] = sum[i] / fc;
                }
                // attention.0 on [h2 | mean]: the global half (+ the layer's bias) is one N tile, shared by the 5 humans
                reg_dense1<KEY, kR_att0_global, false>(ws, [&](int ks) { return gm[ks >> 2][ks & 3]; }, gterm);
                CN_SARL_TICK(5);
            }
            f32x4 a0[NT][7];
            reg_dense_arr<KEY, kR_att0_local, NT, true>(ws, [&](int nt, int ks) { return h2[nt][ks >> 2][ks & 3]; },
                                                        [&](int mt) { return gterm[mt]; }, a0);
            CN_SARL_TICK(6);
            reg_dense_arr<KEY, kR_att_2, NT, false>(ws, [&](int nt, int ks) { return a0[nt][ks >> 2][ks & 3]; }, none, att);
            CN_SARL_TICK(7);
        }
        f32x4 wf[4];
        {
            float sc[NT];  // attention.4: the score of (human, group) in every register of the group's lanes
            // (stays on the matrix side although it is one output: the reference masks a score that is EXACTLY zero, sarl.py:52, and
            // a sum in another order lands on that discontinuity for other rows — one of 1.66 M at the benchmark size, measured)
            reg_dense<KEY, kR_att_4, NT, false>(ws, [&](int nt, int ks) { return att[nt][ks >> 2][ks & 3]; }, none,
                                                [&](int nt, int, f32x4 v) { sc[nt] = v[0]; });
            CN_SARL_TICK(8);
            // masked softmax without max subtraction (sarl.py:52-53); an absent human carries no weight
            float e[NT], total = 0.0f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float s = sc[nt];
                e[nt] = nt < cnt ? expf(s) * (s != 0.0f ? 1.0f : 0.0f) : 0.0f;
                total += e[nt];
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) e[nt] = e[nt] / total;
            // weighted feature sum (sarl.py:60)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                f32x4 sum = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const f32x4 f = mypark[nt * 4 + t][lane];
#pragma unroll
                    for (int i = 0; i < 4; ++i) sum[i] += e[nt] * f[i];
                }
                wf[t] = sum;
            }
        }
        CN_SARL_TICK(9);
        // the next tile's input (and this tile's last use of x is behind us)
        const int next = tile + nw < n_tiles ? tile + nw : tile;
        load_x(next);
        if constexpr (PRE) load_pre(next);
        const int cnt_next = hcount[(size_t)next * kSarlGroups + (lane & 15)];
        // value head on joint = [self | weighted feature] (sarl.py:61-62)
        f32x4 j1[10], j2[7], j3[7], val[1];
        const float mix = lane < 32 ? wf[3][0] : self0;
        reg_dense1<KEY, kR_mlp3_0, true>(
            ws, [&](int ks) { return ks < 12 ? wf[ks >> 2][ks & 3] : ks == 12 ? mix : ks == 13 ? self1 : self0; }, j1);
        CN_SARL_TICK(10);
        reg_dense1<KEY, kR_mlp3_2, true>(ws, [&](int ks) { return j1[ks >> 2][ks & 3]; }, j2);
        CN_SARL_TICK(11);
        reg_dense1<KEY, kR_mlp3_4, true>(ws, [&](int ks) { return j2[ks >> 2][ks & 3]; }, j3);
        CN_SARL_TICK(12);
        reg_dense1<KEY, kR_mlp3_6, false>(ws, [&](int ks) { return j3[ks >> 2][ks & 3]; }, val);
        if (lane < kSarlGroups) {
            const size_t G = (size_t)tile * kSarlGroups + lane;
            if (G < (size_t)n_groups) V[G] = val[0][0];
        }
        cnt = cnt_next;
        // the stream position wraps to quad 0 here: the padding quads are consumed so that slot I % kRegDepth stays aligned
#pragma unroll
        for (int i = reg_qbase(reg_layers(KEY), KEY); i < QT; ++i) (void)reg_take<QT>(ws, i);
        CN_SARL_TICK(13);
    }
    CN_SARL_CLOCK_END_N((n_tiles - wid + nw - 1) / nw);
}

// b + W[:, 13:61] om for every (env, human) and mlp1.0 output feature, in sarl_reg_kernel<4, NT, true>'s accumulator order:
// term[((row * 10 + mt) * 4 + lg) * 4 + kk] = feature 16 mt + 4 kk + lg of row = env * H + human (0 beyond the 150 features).
// Plain fma chain, bias first, map values in ascending order.  Thread = accumulator slot: its 48 weights and its bias stay in
// registers for the block's rows, a row's 48 map values are scalar loads (uniform address).  wom = sarl_om_weights_kernel's
// [48][160] (map value, slot) then the 160 biases — already in slot order, zero beyond the 150 features.
constexpr int kOmTermRows = 8, kOmTermThreads = 192;  // (80 rows per block: 36 us — 640 waves, each waiting for its scalar loads row by row; 8: many waves per SIMD)
__global__ __launch_bounds__(kOmTermThreads) void sarl_om_term_kernel(const float* __restrict__ wom, const float* __restrict__ om, float* __restrict__ term, int rows) {
    const int tid = threadIdx.x;
    if (tid >= 160) return;
    float w[48];
#pragma unroll
    for (int k = 0; k < 48; ++k) w[k] = wom[k * 160 + tid];
    const float bias = wom[48 * 160 + tid];
    const int r0 = blockIdx.x * kOmTermRows, r1 = r0 + kOmTermRows < rows ? r0 + kOmTermRows : rows;
    for (int row = r0; row < r1; ++row) {
        const float* m = om + (size_t)row * 48;
        float v = bias;
#pragma unroll
        for (int k = 0; k < 48; ++k) v = __builtin_fmaf(w[k], m[k], v);
        term[(size_t)row * 160 + tid] = v;
    }
}
__global__ void sarl_om_weights_kernel(const float* W /*[150][61]*/, const float* b, float* wom) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 49 * 160) return;
    const int k = idx / 160, slot = idx % 160;
    const int mt = slot >> 4, lg = (slot >> 2) & 3, kk = slot & 3, f = 16 * mt + 4 * kk + lg;  // slot (mt, lg, kk) <- feature
    wom[idx] = f < 150 ? (k < 48 ? W[f * 61 + 13 + k] : b[f]) : 0.0f;
}

// cadrl.ValueNetwork (cadrl.py:22-29) with the activations in registers: the same MLP for every (group, human) row — NT N tiles
// of 16 rows per wave, layers chained through the accumulator layout as above — then the minimum over the humans present
// (cadrl.py:162-163: torch.min over dim 0, the first minimum's value).  X, hcount, V as for sarl_reg_kernel.
template <int NT>
__global__ __launch_bounds__(kRegWaves * 64) void cadrl_reg_kernel(const float* stream, const float* X, float* V, int n_groups,
                                                                   int n_tiles, int ks_x, const int* hcount, int H, int n_chunks) {
    static_assert(NT >= 1 && NT <= kRegHumans, "the activations of at most 5 humans fit the register file");
    constexpr int NK = kRegCadrl, QT = reg_total_quads(NK);
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * kRegWaves + (threadIdx.x >> 6), nw = gridDim.x * kRegWaves;
    RegStream ws;
    ws.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(stream), 0, QT * 1024, 0x00020000);
    ws.voff = (uint32_t)lane * 16u;
#pragma unroll
    for (int i = 0; i < kRegDepth; ++i) ws.q[i] = reg_quad(ws, i);
    if (wid >= n_tiles) return;
    const gfloat_p Xg = as_global(X) + lane;
    float x[NT][4];
    // chunk c of a tile = humans c NT .. c NT + NT - 1 (more than 5 humans: n_chunks > 1; indices beyond the crowd repeat its last
    // human and never win the minimum)
    const auto load_x = [&](int t, int c) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int h = c * NT + nt < H ? c * NT + nt : H - 1;
            const gfloat_p xt = Xg + ((size_t)t * H + h) * ks_x * 64;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) x[nt][ks] = xt[ks * 64];
        }
    };
    load_x(wid, 0);
    const auto none = [](int) { return f32x4{0.0f, 0.0f, 0.0f, 0.0f}; };
    for (int tile = wid; tile < n_tiles; tile += nw) {
        const int cnt = hcount[(size_t)tile * kSarlGroups + (lane & 15)];
        float m = 0.0f;
#pragma unroll 1
        for (int c = 0; c < n_chunks; ++c) {
            f32x4 h3[NT][7];
            {
                f32x4 h2[NT][7];
                {
                    f32x4 h1[NT][10];
                    reg_dense_arr<NK, 0, NT, true>(ws, [&](int nt, int ks) { return x[nt][ks]; }, none, h1);
                    // x is dead: the next chunk's (or the next tile's first chunk's) rows travel while this one computes
                    if (c + 1 < n_chunks) load_x(tile, c + 1);
                    else load_x(tile + nw < n_tiles ? tile + nw : tile, 0);
                    reg_dense_arr<NK, 1, NT, false>(ws, [&](int nt, int ks) { return h1[nt][ks >> 2][ks & 3]; }, none, h2);
                }
                reg_dense_arr<NK, 2, NT, true>(ws, [&](int nt, int ks) { return h2[nt][ks >> 2][ks & 3]; }, none, h3);
            }
            float sc[NT];
            reg_dense_valu1<NK, 3, NT>(ws, [&](int nt, int ks) { return h3[nt][ks >> 2][ks & 3]; },
                                       [&](int nt, int, f32x4 v) { sc[nt] = v[0]; });
#pragma unroll
            for (int i = reg_qbase(reg_layers(NK), NK); i < QT; ++i) (void)reg_take<QT>(ws, i);
            if (c == 0) m = sc[0];  // torch.min over dim 0: the first minimum's value
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) m = (c * NT + nt < cnt && sc[nt] < m) ? sc[nt] : m;
        }
        if (lane < kSarlGroups) {
            const size_t G = (size_t)tile * kSarlGroups + lane;
            if (G < (size_t)n_groups) V[G] = m;
        }
    }
}

// ---- lstm_rl.ValueNetwork1 (lstm_rl.py:9-33) with the state in registers ------------------------------------------------
// The recurrence is sequential over the humans of a group, so a wave carries NT TILES (16 groups each) through it side by
// side: per human one gate layer [x_t | h] -> 4 x 50 pre-activations as 13 output tiles x (XKS + 13) k-steps x NT independent
// MFMAs; the cell update of tile mt (sigmoid / tanh on its f32x4, c and h are one register per k-step) runs behind the first
// MFMAs of tile mt + 1.  Any number of humans: nothing is sized by H.  sigmoid(x) = 1 / (1 + 2^(-x log2 e)) and tanh(x) =
// 1 - 2 / (1 + 2^(2 x log2 e)) on v_exp_f32 / v_rcp_f32 (1 ulp each): absolute error ~1e-7 per activation, against expf /
// tanhf's ~20 instructions each on a wave that has nothing to hide them behind.
__device__ __forceinline__ float reg_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.442695040888963f * x));
}
__device__ __forceinline__ float reg_tanh(float x) {
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.885390081777927f * x));
}

// (m all ones: a, zero: b) as ONE v_bfi_b32 on values that are computed either way.  `present ? new : old` let hipcc sink the
// whole cell update under a divergent branch — s_and_saveexec / s_cbranch_execz / s_or_b64 around each of the 65 (tile, output
// tile) updates of a step, every one a scheduling barrier between the MFMAs of neighbouring output tiles.
__device__ __forceinline__ float reg_select(uint32_t m, float a, float b) {
    return __uint_as_float((__float_as_uint(a) & m) | (__float_as_uint(b) & ~m));
}

template <int XKS, int NT>
__device__ __forceinline__ void lstm_reg_bundle(RegStream& wg, RegStream& wh, const float* X, float* V, int n_groups, int H, int ks_x,
                                                const int* hcount, int base, int lane) {
    constexpr int GK = kRegLstmGates + XKS, HK = kRegLstmHead, KS = kRegLstmKs;
    constexpr int QG = reg_total_quads(GK), QH = reg_total_quads(HK);
    const gfloat_p Xg = as_global(X) + lane;
    float h[NT][KS], c[NT][KS], x[NT][XKS], self0[NT], self1[NT];
    int cnt[NT];
    const auto load_x = [&](int t, float (&dst)[NT][XKS]) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const gfloat_p xt = Xg + ((size_t)(base + nt) * H + t) * ks_x * 64;
#pragma unroll
            for (int ks = 0; ks < XKS; ++ks) dst[nt][ks] = xt[ks * 64];
        }
    };
    load_x(0, x);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        cnt[nt] = hcount[(size_t)(base + nt) * kSarlGroups + (lane & 15)];
        self0[nt] = x[nt][0], self1[nt] = x[nt][1];  // self_state = state[:, 0, :6]: the first row's k-steps 0, 1
#pragma unroll
        for (int j = 0; j < KS; ++j) h[nt][j] = 0.0f, c[nt][j] = 0.0f;
    }
    const auto none = [](int) { return f32x4{0.0f, 0.0f, 0.0f, 0.0f}; };
#pragma unroll 1
    for (int t = 0; t < H; ++t) {
        float xn[NT][XKS], hn[NT][KS];
        load_x(t + 1 < H ? t + 1 : t, xn);  // the next human's rows travel while this one computes
        uint32_t present[NT];  // `mixed` rule: a group whose episode has fewer humans keeps its state from here on
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) present[nt] = t < cnt[nt] ? 0xffffffffu : 0u;
        reg_dense<GK, 0, NT, false>(
            wg, [&](int nt, int ks) { return ks < XKS ? x[nt][ks] : h[nt][ks - XKS]; }, none,
            [&](int nt, int mt, f32x4 v) {
                const float cn_ = reg_sigmoid(v[1]) * c[nt][mt] + reg_sigmoid(v[0]) * reg_tanh(v[2]);
                const float hn_ = reg_sigmoid(v[3]) * reg_tanh(cn_);
                c[nt][mt] = reg_select(present[nt], cn_, c[nt][mt]);
                hn[nt][mt] = reg_select(present[nt], hn_, h[nt][mt]);
            });
#pragma unroll
        for (int i = reg_qbase(1, GK); i < QG; ++i) (void)reg_take<QG>(wg, i);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int j = 0; j < KS; ++j) h[nt][j] = hn[nt][j];
#pragma unroll
            for (int ks = 0; ks < XKS; ++ks) x[nt][ks] = xn[nt][ks];
        }
    }
    float val[NT];
    {
        f32x4 j3[NT][7];
        {
            f32x4 j2[NT][7];
            {
                f32x4 j1[NT][10];
                reg_dense_arr<HK, 0, NT, true>(
                    wh, [&](int nt, int ks) { return ks < KS ? h[nt][ks] : ks == KS ? self0[nt] : self1[nt]; }, none, j1);
                reg_dense_arr<HK, 1, NT, false>(wh, [&](int nt, int ks) { return j1[nt][ks >> 2][ks & 3]; }, none, j2);
            }
            reg_dense_arr<HK, 2, NT, true>(wh, [&](int nt, int ks) { return j2[nt][ks >> 2][ks & 3]; }, none, j3);
        }
        reg_dense<HK, 3, NT, false>(wh, [&](int nt, int ks) { return j3[nt][ks >> 2][ks & 3]; }, none,
                                    [&](int nt, int, f32x4 v) { val[nt] = v[0]; });
    }
#pragma unroll
    for (int i = reg_qbase(4, HK); i < QH; ++i) (void)reg_take<QH>(wh, i);
    if (lane < kSarlGroups) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const size_t G = (size_t)(base + nt) * kSarlGroups + lane;
            if (G < (size_t)n_groups) V[G] = val[nt];
        }
    }
}

// Persistent: full rounds of 5-tile bundles over all waves, then the remainder — as one more round of bundles when that is
// shorter than its single tiles one per wave (a single tile's MFMAs are one dependent chain: ~0.4 of a bundle's time).
template <int XKS>
__global__ __launch_bounds__(kRegWaves * 64) void lstm_reg_kernel(const float* gates, const float* head, const float* X, float* V,
                                                                  int n_groups, int n_tiles, int H, int ks_x, const int* hcount) {
    constexpr int GK = kRegLstmGates + XKS, NT = kRegHumans;
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * kRegWaves + (threadIdx.x >> 6), nw = gridDim.x * kRegWaves;
    RegStream wg, wh;
    wg.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gates), 0, reg_total_quads(GK) * 1024, 0x00020000);
    wh.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(head), 0, reg_total_quads(kRegLstmHead) * 1024, 0x00020000);
    wg.voff = wh.voff = (uint32_t)lane * 16u;
#pragma unroll
    for (int i = 0; i < kRegDepth; ++i) wg.q[i] = reg_quad(wg, i), wh.q[i] = reg_quad(wh, i);
    const int rounds = n_tiles / (NT * nw);
    int done = 0;
    for (int k = 0; k < rounds; ++k)
        lstm_reg_bundle<XKS, NT>(wg, wh, X, V, n_groups, H, ks_x, hcount, (k * nw + wid) * NT, lane);
    done = rounds * NT * nw;
    if (n_tiles - done > 2 * nw) {
        const int nb = (n_tiles - done) / NT;
        if (wid < nb) lstm_reg_bundle<XKS, NT>(wg, wh, X, V, n_groups, H, ks_x, hcount, done + wid * NT, lane);
        done += nb * NT;
    }
    for (int tile = done + wid; tile < n_tiles; tile += nw)
        lstm_reg_bundle<XKS, 1>(wg, wh, X, V, n_groups, H, ks_x, hcount, tile, lane);
}

// ---- lstm_rl.ValueNetwork2 (lstm_rl.py:36-66): mlp1 on every human's row in front of the cell ----------------------------
// lstm_reg_bundle with one more stage per human: x_t -> mlp1 (150 - 100 - 100 - 50, ReLU between) as the value head's layers
// are run — accumulators of a layer are the next layer's B operands, AGPR / VGPR arrays alternating — and the gate layer reads
// [mlp1(x_t) | h] (13 + 13 k-steps).  NT tiles per wave: 3 (mlp1's 150-wide layer keeps 40 registers per tile alive beside the
// 100-wide one's 28 and the cell's 26).
template <int XKS, int NT>
__device__ __forceinline__ void lstm2_reg_bundle(RegStream& wm, RegStream& wg, RegStream& wh, const float* X, float* V, int n_groups,
                                                 int H, int ks_x, const int* hcount, int base, int lane) {
    constexpr int MK = kRegLstmMlp1 + XKS, GK = kRegLstmGates + kRegLstmKs, HK = kRegLstmHead, KS = kRegLstmKs;
    constexpr int QM = reg_total_quads(MK), QG = reg_total_quads(GK), QH = reg_total_quads(HK);
    const gfloat_p Xg = as_global(X) + lane;
    float h[NT][KS], c[NT][KS], x[NT][XKS], self0[NT], self1[NT];
    int cnt[NT];
    const auto load_x = [&](int t, float (&dst)[NT][XKS]) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const gfloat_p xt = Xg + ((size_t)(base + nt) * H + t) * ks_x * 64;
#pragma unroll
            for (int ks = 0; ks < XKS; ++ks) dst[nt][ks] = xt[ks * 64];
        }
    };
    load_x(0, x);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        cnt[nt] = hcount[(size_t)(base + nt) * kSarlGroups + (lane & 15)];
        self0[nt] = x[nt][0], self1[nt] = x[nt][1];  // self_state = state[:, 0, :6]: the first row's k-steps 0, 1
#pragma unroll
        for (int j = 0; j < KS; ++j) h[nt][j] = 0.0f, c[nt][j] = 0.0f;
    }
    const auto none = [](int) { return f32x4{0.0f, 0.0f, 0.0f, 0.0f}; };
#pragma unroll 1
    for (int t = 0; t < H; ++t) {
        float xn[NT][XKS], hn[NT][KS];
        load_x(t + 1 < H ? t + 1 : t, xn);  // the next human's rows travel while this one computes
        uint32_t present[NT];  // `mixed` rule: a group whose episode has fewer humans keeps its state from here on
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) present[nt] = t < cnt[nt] ? 0xffffffffu : 0u;
        f32x4 m4[NT][4];
        {
            f32x4 m3[NT][7];
            {
                f32x4 m2[NT][7];
                {
                    f32x4 m1[NT][10];
                    reg_dense_arr<MK, 0, NT, true>(wm, [&](int nt, int ks) { return x[nt][ks]; }, none, m1);
                    reg_dense_arr<MK, 1, NT, false>(wm, [&](int nt, int ks) { return m1[nt][ks >> 2][ks & 3]; }, none, m2);
                }
                reg_dense_arr<MK, 2, NT, true>(wm, [&](int nt, int ks) { return m2[nt][ks >> 2][ks & 3]; }, none, m3);
            }
            reg_dense<MK, 3, NT, false>(wm, [&](int nt, int ks) { return m3[nt][ks >> 2][ks & 3]; }, none,
                                        [&](int nt, int mt, f32x4 v) { m4[nt][mt] = v; });  // (cadrl.mlp: no ReLU behind the last layer)
        }
#pragma unroll
        for (int i = reg_qbase(4, MK); i < QM; ++i) (void)reg_take<QM>(wm, i);
        reg_dense<GK, 0, NT, false>(
            wg, [&](int nt, int ks) { return ks < KS ? m4[nt][ks >> 2][ks & 3] : h[nt][ks - KS]; }, none,
            [&](int nt, int mt, f32x4 v) {
                const float cn_ = reg_sigmoid(v[1]) * c[nt][mt] + reg_sigmoid(v[0]) * reg_tanh(v[2]);
                const float hn_ = reg_sigmoid(v[3]) * reg_tanh(cn_);
                c[nt][mt] = reg_select(present[nt], cn_, c[nt][mt]);
                hn[nt][mt] = reg_select(present[nt], hn_, h[nt][mt]);
            });
#pragma unroll
        for (int i = reg_qbase(1, GK); i < QG; ++i) (void)reg_take<QG>(wg, i);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int j = 0; j < KS; ++j) h[nt][j] = hn[nt][j];
#pragma unroll
            for (int ks = 0; ks < XKS; ++ks) x[nt][ks] = xn[nt][ks];
        }
    }
    float val[NT];
    {
        f32x4 j3[NT][7];
        {
            f32x4 j2[NT][7];
            {
                f32x4 j1[NT][10];
                reg_dense_arr<HK, 0, NT, true>(
                    wh, [&](int nt, int ks) { return ks < KS ? h[nt][ks] : ks == KS ? self0[nt] : self1[nt]; }, none, j1);
                reg_dense_arr<HK, 1, NT, false>(wh, [&](int nt, int ks) { return j1[nt][ks >> 2][ks & 3]; }, none, j2);
            }
            reg_dense_arr<HK, 2, NT, true>(wh, [&](int nt, int ks) { return j2[nt][ks >> 2][ks & 3]; }, none, j3);
        }
        reg_dense<HK, 3, NT, false>(wh, [&](int nt, int ks) { return j3[nt][ks >> 2][ks & 3]; }, none,
                                    [&](int nt, int, f32x4 v) { val[nt] = v[0]; });
    }
#pragma unroll
    for (int i = reg_qbase(4, HK); i < QH; ++i) (void)reg_take<QH>(wh, i);
    if (lane < kSarlGroups) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const size_t G = (size_t)(base + nt) * kSarlGroups + lane;
            if (G < (size_t)n_groups) V[G] = val[nt];
        }
    }
}

constexpr int kLstm2Tiles = 3;  // (measured at 4096 x 81 x 5: 2 / 3 / 4 tiles per wave 1.797 / 1.765 / 1.768 ms; 4 spill with 61 inputs)
template <int XKS>
__global__ __launch_bounds__(kRegWaves * 64) void lstm2_reg_kernel(const float* mlp1, const float* gates, const float* head, const float* X,
                                                                   float* V, int n_groups, int n_tiles, int H, int ks_x,
                                                                   const int* hcount) {
    constexpr int NT = kLstm2Tiles;
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * kRegWaves + (threadIdx.x >> 6), nw = gridDim.x * kRegWaves;
    RegStream wm, wg, wh;
    wm.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(mlp1), 0, reg_total_quads(kRegLstmMlp1 + XKS) * 1024, 0x00020000);
    wg.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gates), 0, reg_total_quads(kRegLstmGates + kRegLstmKs) * 1024, 0x00020000);
    wh.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(head), 0, reg_total_quads(kRegLstmHead) * 1024, 0x00020000);
    wm.voff = wg.voff = wh.voff = (uint32_t)lane * 16u;
#pragma unroll
    for (int i = 0; i < kRegDepth; ++i) wm.q[i] = reg_quad(wm, i), wg.q[i] = reg_quad(wg, i), wh.q[i] = reg_quad(wh, i);
    const int rounds = n_tiles / (NT * nw);
    int done = 0;
    for (int k = 0; k < rounds; ++k)
        lstm2_reg_bundle<XKS, NT>(wm, wg, wh, X, V, n_groups, H, ks_x, hcount, (k * nw + wid) * NT, lane);
    done = rounds * NT * nw;
    if (n_tiles - done > 2 * nw) {
        const int nb = (n_tiles - done) / NT;
        if (wid < nb) lstm2_reg_bundle<XKS, NT>(wm, wg, wh, X, V, n_groups, H, ks_x, hcount, done + wid * NT, lane);
        done += nb * NT;
    }
    for (int tile = done + wid; tile < n_tiles; tile += nw)
        lstm2_reg_bundle<XKS, 1>(wm, wg, wh, X, V, n_groups, H, ks_x, hcount, tile, lane);
}

// ---- sarl.ValueNetwork (sarl.py:28-65) for more than 5 humans, activations in registers ----------------------------------
// The humans of a tile's 16 groups pass in n_chunks chunks of NT (3 or 4: human h = chunk * NT + nt; indices beyond the crowd
// repeat its last human and are masked like the absent humans of the `mixed` rule).  Two passes, because attention.0 needs
// the mean of mlp1's output over ALL humans (sarl.py:42-46) before any score exists:
//   pass 1  per chunk: mlp1 -> h2 (NT x 7 registers of f32x4), summed over the humans present, and parked in a per-wave
//           global scratch (7 KiB per human: written and read back by the same lane, loads bypass L1);
//   then    the mean, attention.0's global term (one N tile);
//   pass 2  per chunk: h2 back from the scratch, mlp2 -> features (wave-private LDS), attention -> score, and — as the chunked
//           LDS kernel does — exp(score) and exp(score) x feature accumulate per group; the division by the total comes last
//           (the reference divides first: w_h = e_h / total, then sums w_h f_h — same value to rounding);
//   then    the value head.
// PRE as in sarl_reg_kernel: mlp1.0 starts from the hoisted occupancy-map term instead of its bias.
template <int NT, bool PRE>
__global__ __launch_bounds__(kRegWaves * 64) void sarl_reg_chunk_kernel(const float* sa, const float* sb, const float* sg, const float* X,
                                                                        float* V, float* scratch, int n_groups, int n_tiles, int H,
                                                                        int n_chunks, int ks_x, const int* hcount, const float* term,
                                                                        int n_actions) {
    static_assert(NT == 3 || NT == 4, "two accumulator sets, mlp1's 17 tiles and three weight queues fit the register file up to 4 N tiles");
    constexpr int KA = PRE ? kRegChunkAPre : kRegChunkA, KB = kRegChunkB, KG = kRegChunkG;
    constexpr int QA = reg_total_quads(KA), QB = reg_total_quads(KB), QG = reg_total_quads(KG);
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * kRegWaves + (threadIdx.x >> 6), nw = gridDim.x * kRegWaves;
    RegStream wa, wb, wg;
    wa.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sa), 0, QA * 1024, 0x00020000);
    wb.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sb), 0, QB * 1024, 0x00020000);
    wg.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sg), 0, QG * 1024, 0x00020000);
    wa.voff = wb.voff = wg.voff = (uint32_t)lane * 16u;
#pragma unroll
    for (int i = 0; i < kRegDepth; ++i) wa.q[i] = reg_quad(wa, i), wb.q[i] = reg_quad(wb, i), wg.q[i] = reg_quad(wg, i);
    if (wid >= n_tiles) return;
    // the wave's scratch: [chunk][nt][7 quads][64 lanes] of f32x4
    const int wave_bytes = n_chunks * NT * 7 * 1024;
    const __amdgpu_buffer_rsrc_t hrsrc =
        __builtin_amdgcn_make_buffer_rsrc(scratch + (size_t)wid * (wave_bytes / 4), 0, wave_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t prsrc = wa.rsrc;
    if constexpr (PRE)
        prsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(term), 0, (int)(((long long)n_groups / n_actions) * H * 640), 0x00020000);
    const gfloat_p Xg = as_global(X) + lane;
    __shared__ f32x4 park[kRegWaves][NT * 4][64];
    f32x4(*const mypark)[64] = park[threadIdx.x >> 6];
    const auto none = [](int) { return f32x4{0.0f, 0.0f, 0.0f, 0.0f}; };
    const auto human = [&](int c, int nt) { return c * NT + nt < H ? c * NT + nt : H - 1; };
    for (int tile = wid; tile < n_tiles; tile += nw) {
        const int cnt = hcount[(size_t)tile * kSarlGroups + (lane & 15)];
        const long long G = (long long)tile * kSarlGroups + (lane & 15);
        const uint32_t env_off = (uint32_t)((G < n_groups ? G : (long long)n_groups - 1) / n_actions) * (uint32_t)(H * 640) +
                                 (uint32_t)(lane >> 4) * 16u;
        float x[NT][4];
        // PRE: the term of output tile mt in pre[mt & 1][nt]; tile 0 travels with the chunk's X, tile mt + 1 is requested when tile mt
        // is consumed (all ten at once do not fit beside mlp1's 17 tiles of activations here)
        f32x4 pre[2][PRE ? NT : 1];
        uint32_t prow[NT];
        const auto load_chunk = [&](int c) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int h = human(c, nt);
                const gfloat_p xt = Xg + ((size_t)tile * H + h) * ks_x * 64;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) x[nt][ks] = xt[ks * 64];
                if constexpr (PRE) {
                    prow[nt] = env_off + (uint32_t)h * 640u;
                    pre[0][nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prsrc, prow[nt], 0, 0));
                }
            }
        };
        // ---- pass 1
        load_chunk(0);
        const float self0 = x[0][0], self1 = x[0][1];  // self state = features 0..5 of any row (sarl.py:37)
        f32x4 gsum[7];
#pragma unroll
        for (int t = 0; t < 7; ++t) gsum[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 1
        for (int c = 0; c < n_chunks; ++c) {
            f32x4 h2[NT][7];
            {
                f32x4 h1[NT][10];
                if constexpr (PRE)
                    reg_dense_arr<KA, 0, NT, true>(
                        wa, [&](int nt, int ks) { return x[nt][ks]; },
                        [&](int nt, int mt) {
                            const f32x4 v = pre[mt & 1][nt];
                            if (mt + 1 < 10)
                                pre[(mt + 1) & 1][nt] =
                                    __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prsrc, prow[nt] + (mt + 1) * 64, 0, 0));
                            return v;
                        },
                        h1);
                else
                    reg_dense_arr<KA, 0, NT, true>(wa, [&](int nt, int ks) { return x[nt][ks]; }, none, h1);
                load_chunk(c + 1 < n_chunks ? c + 1 : c);  // x / pre are dead: the next chunk's travel while this one computes
                reg_dense_arr<KA, 1, NT, false>(wa, [&](int nt, int ks) { return h1[nt][ks >> 2][ks & 3]; }, none, h2);
            }
#pragma unroll
            for (int i = reg_qbase(2, KA); i < QA; ++i) (void)reg_take<QA>(wa, i);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const bool present = c * NT + nt < cnt;
#pragma unroll
                for (int t = 0; t < 7; ++t) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) gsum[t][i] += present ? h2[nt][t][i] : 0.0f;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, h2[nt][t]), hrsrc, (uint32_t)lane * 16u + (nt * 7 + t) * 1024,
                                                           c * (NT * 7 * 1024), 0);
                }
            }
        }
        // ---- the global state and its attention term
        f32x4 gterm[7];
        {
            const float fc = (float)cnt;
#pragma unroll
            for (int t = 0; t < 7; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) gsum[t][i] = gsum[t][i] / fc;
            reg_dense1<KG, 0, false>(wg, [&](int ks) { return gsum[ks >> 2][ks & 3]; }, gterm);
        }
        // ---- pass 2
        f32x4 wf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) wf[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        float den = 0.0f;
        f32x4 h2[NT][7];
        const auto load_h2 = [&](int c) {  // glc: the lines were written by this lane a moment ago and may sit stale in L1
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int t = 0; t < 7; ++t)
                    h2[nt][t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(hrsrc, (uint32_t)lane * 16u + (nt * 7 + t) * 1024,
                                                                                               c * (NT * 7 * 1024), 1));
        };
        load_h2(0);
#pragma unroll 1
        for (int c = 0; c < n_chunks; ++c) {
            f32x4 att[NT][7];
            {
                {
                    f32x4 t1[NT][7];
                    reg_dense_arr<KB, 0, NT, true>(wb, [&](int nt, int ks) { return h2[nt][ks >> 2][ks & 3]; }, none, t1);
                    reg_dense<KB, 1, NT, false>(wb, [&](int nt, int ks) { return t1[nt][ks >> 2][ks & 3]; }, none,
                                                [&](int nt, int mt, f32x4 v) { mypark[nt * 4 + mt][lane] = v; });
                }
                f32x4 a0[NT][7];
                reg_dense_arr<KB, 2, NT, true>(wb, [&](int nt, int ks) { return h2[nt][ks >> 2][ks & 3]; },
                                               [&](int mt) { return gterm[mt]; }, a0);
                load_h2(c + 1 < n_chunks ? c + 1 : c);  // h2 is dead: the next chunk's comes back during attention.2 / .4
                reg_dense_arr<KB, 3, NT, false>(wb, [&](int nt, int ks) { return a0[nt][ks >> 2][ks & 3]; }, none, att);
            }
            float sc[NT];
            reg_dense<KB, 4, NT, false>(wb, [&](int nt, int ks) { return att[nt][ks >> 2][ks & 3]; }, none,
                                        [&](int nt, int, f32x4 v) { sc[nt] = v[0]; });
#pragma unroll
            for (int i = reg_qbase(5, KB); i < QB; ++i) (void)reg_take<QB>(wb, i);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {  // masked softmax without max subtraction (sarl.py:52-53), unnormalised
                const float s_ = sc[nt];
                const float e = c * NT + nt < cnt ? expf(s_) * (s_ != 0.0f ? 1.0f : 0.0f) : 0.0f;
                den += e;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const f32x4 f = mypark[nt * 4 + t][lane];
#pragma unroll
                    for (int i = 0; i < 4; ++i) wf[t][i] += e * f[i];
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) wf[t][i] = wf[t][i] / den;
        // ---- value head on joint = [self | weighted feature] (sarl.py:61-62)
        f32x4 j1[10], j2[7], j3[7], val[1];
        const float mix = lane < 32 ? wf[3][0] : self0;
        reg_dense1<KG, 1, true>(
            wg, [&](int ks) { return ks < 12 ? wf[ks >> 2][ks & 3] : ks == 12 ? mix : ks == 13 ? self1 : self0; }, j1);
        reg_dense1<KG, 2, true>(wg, [&](int ks) { return j1[ks >> 2][ks & 3]; }, j2);
        reg_dense1<KG, 3, true>(wg, [&](int ks) { return j2[ks >> 2][ks & 3]; }, j3);
        reg_dense1<KG, 4, false>(wg, [&](int ks) { return j3[ks >> 2][ks & 3]; }, val);
#pragma unroll
        for (int i = reg_qbase(5, KG); i < QG; ++i) (void)reg_take<QG>(wg, i);
        if (lane < kSarlGroups && G < (long long)n_groups) V[G] = val[0][0];
    }
}

}  // namespace cn
