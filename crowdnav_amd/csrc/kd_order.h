// RVO2's kd-tree visiting order for simulators of more than 10 agents.
//
// Replaces, for the neighbour selection of `sim.doStep()` (/root/reference crowd_sim/envs/policy/orca.py:128, with
// max_neighbors = 10 at orca.py:62 and one simulator per agent, orca.py:95-110), what RVO2's KdTree::buildAgentTree /
// queryAgentTreeRecursive and Agent::insertAgentNeighbor decide when two candidates are at EXACTLY the same float32 squared
// distance (the published RVO2 v2.0 KdTree.cpp / Agent.cpp).  Below 11 agents per simulator the tree is
// one leaf and candidates are visited in insertion order — the brute-force stable rank of orca_phases.  Above, RVO2 visits
// them in the order of a nearer-child-first traversal of a tree whose leaves hold a permutation that the builder partitions
// IN PLACE and that persists from step to step with the simulator (a human's simulator lives for one episode, the robot's for
// the life of its policy object), so the order at a tie depends on the history of the scene.
//
// The device does not run RVO2's recursion per simulator; it uses three facts (pinned on the CPU by
// tests/test_kd_order_emulation.py against a transcription of RVO2's code, and on the GPU by the forced-tie tests):
//   1. the tree STRUCTURE — which position ranges split, at which count, with which agents on the lower side — depends on the
//      point SET only (bounding boxes are min / max), so it is built once per env, breadth first on agent sets with the env's
//      lanes cooperating (kd_build_tree), and shared by the env's simulators, whose permutations differ;
//   2. RVO2's two-pointer partition equals "the i-th misplaced element of the lower zone, from the left, swaps with the i-th
//      misplaced element of the upper zone, from the right": position masks and a few bit operations per simulator lane
//      (kd_partition); a node whose record equals last step's, below unchanged ancestors, needs nothing at all;
//   3. the neighbour list RVO2 ends with is the first maxNeighbors candidates of the stable order by (distSq, position in the
//      traversal of the WHOLE tree): a pruned subtree only holds candidates that would have been rejected.  So the traversal
//      is needed only when a tie is detected among the kept neighbours or at the cut (rare: kd_visit_order runs serially
//      on the lanes of the affected simulators, then those simulators' candidates are ranked again).
// Degenerate splits (no coordinate below the split value: more than 10 agents within an ulp of each other on both axes) make
// RVO2's children depend on the permutation; such a node is kept as a leaf here.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cn {

constexpr int kKdLeaf = 10;  // RVO2 MAX_LEAF_SIZE

struct KdNode {      // a node that splits, of one (env, tree)
    uint32_t meta;   // begin | end << 8 | n_left << 16 | degenerate << 24
    uint32_t pad0;
    uint64_t left;   // env-local agents on the lower side of the split
    uint64_t set;    // env-local agents of the node
    uint64_t pad;
};

__host__ __device__ inline int kd_row_bytes(int A) { return (A + 3) & ~3; }          // a simulator's permutation, dword-padded
__host__ __device__ inline int kd_max_nodes(int A) { return A > kKdLeaf ? A - kKdLeaf : 1; }
// trees per env: 0 = every agent (the robot's simulator; the humans' when they see the robot), 1 = the humans only
__host__ __device__ inline size_t kd_lds_bytes(int nA, int A, int E) {
    const size_t row = (size_t)kd_row_bytes(A), mn = (size_t)kd_max_nodes(A);
    return (size_t)nA * row * 2                 // ord, visit
           + (size_t)nA * (mn + 1) * 2          // traversal stacks
           + (size_t)E * 2 * mn * sizeof(KdNode) * 2  // node lists: two generations (this step's, last step's)
           + (size_t)E * 2 * 3 * 4              // their lengths; first node that differs from last step's
           + (size_t)E * mn * 16                // bounding-box accumulators
           + (size_t)nA * 8 + 16;               // next-nearest distance, tie flag; generation
}

struct KdSmem {
    uint8_t* ord;     // [nA][row] env-local agent at each position of the simulator's permutation
    uint8_t* visit;   // [nA][row] traversal position of each env-local agent (valid for simulators with a tie)
    uint16_t* stack;  // [nA][mn + 1]
    KdNode* nodes;    // [2 generations][E][2 trees][mn] (a step whose tree differs from the last one builds into the other generation)
    int* count;       // [2 generations][E][2 trees]
    int* dirty;       // [E][2 trees] first node of this step's list that differs from last step's (= count: none does)
    uint32_t* bb;     // [E][mn][4] min x, max x, min y, max y as order-preserving integers
    float* dnext;     // [nA] (as int) slots below maxNeighbors claimed by the agent's candidates in the first sweep of the pair phase
    int* tie;         // [nA] this simulator has an exact tie that the visiting order decides
    int* gen;         // [1] the generation that holds the most recent node lists
    int row, mn;
};

// Byte offsets of the region's arrays, computed ONCE on the host (cn_create) and carried in the kernel arguments: derived on
// the device they were ~40 scalar instructions of 64-bit index arithmetic per use — eight uses per step, a third of the whole
// kd bookkeeping — and held in registers across the step loop they cost the rollout kernel its last VGPRs.  Kernel arguments
// are re-loadable (s_load from the kernarg segment), so they need not stay live.
struct KdLayout {
    uint32_t nodes, bb, count, dirty, dnext, tie, gen, ord, visit, stack;
    int32_t row, mn;
};
__host__ __device__ inline KdLayout kd_layout(int nA, int A, int E) {
    KdLayout l;
    l.row = kd_row_bytes(A), l.mn = kd_max_nodes(A);
    uint32_t p = 0;
    l.nodes = p, p += (uint32_t)(E * 2 * l.mn * sizeof(KdNode) * 2);
    l.bb = p, p += (uint32_t)(E * l.mn * 16);
    l.count = p, p += (uint32_t)(E * 2 * 2 * 4);
    l.dirty = p, p += (uint32_t)(E * 2 * 4);
    l.dnext = p, p += (uint32_t)(nA * 4);
    l.tie = p, p += (uint32_t)(nA * 4);
    l.gen = p, p += 16;
    l.ord = p, p += (uint32_t)(nA * l.row);
    l.visit = p, p += (uint32_t)(nA * l.row);
    l.stack = p;
    return l;
}
__device__ __forceinline__ KdSmem kd_carve(char* p, const KdLayout& l) {
    KdSmem k;
    k.row = l.row, k.mn = l.mn;
    k.nodes = reinterpret_cast<KdNode*>(p + l.nodes);
    k.bb = reinterpret_cast<uint32_t*>(p + l.bb);
    k.count = reinterpret_cast<int*>(p + l.count);
    k.dirty = reinterpret_cast<int*>(p + l.dirty);
    k.dnext = reinterpret_cast<float*>(p + l.dnext);
    k.tie = reinterpret_cast<int*>(p + l.tie);
    k.gen = reinterpret_cast<int*>(p + l.gen);
    k.ord = reinterpret_cast<uint8_t*>(p + l.ord);
    k.visit = reinterpret_cast<uint8_t*>(p + l.visit);
    k.stack = reinterpret_cast<uint16_t*>(p + l.stack);
    return k;
}

// float -> unsigned with the same order (for LDS min / max)
__device__ __forceinline__ uint32_t kd_key(float v) {
    const uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float kd_unkey(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// agents in simulator (tree) t of an env, and which tree agent a's own simulator uses
__device__ __forceinline__ int kd_tree_size(int A, int t) { return t == 0 ? A : A - 1; }
__device__ __forceinline__ int kd_tree_of(int a, int robot_visible) { return (a > 0 && !robot_visible) ? 1 : 0; }
__device__ __forceinline__ bool kd_tree_on(int A, int t, int robot_visible) {
    return kd_tree_size(A, t) > kKdLeaf && (t == 0 || !robot_visible);
}

// A freshly built simulator: itself first, then the others in the order ORCA.predict adds them (orca.py:99-104: the other
// humans by index, then the robot if it is visible; the robot's own simulator: the humans by index).
__device__ __forceinline__ void kd_identity_row(uint8_t* row, int row_bytes, int A, int a, int robot_visible) {
    for (int i = 0; i < row_bytes; ++i) row[i] = 0;
    int n = 0;
    row[n++] = (uint8_t)a;
    for (int j = 1; j < A; ++j)
        if (j != a) row[n++] = (uint8_t)j;
    if (a > 0 && robot_visible) row[n++] = 0;
}

// RVO2's partition of positions [begin, end) of one permutation into "lower side" (n_left elements) and the rest.
__device__ __forceinline__ void kd_partition(uint8_t* row, int begin, int end, int n_left, uint64_t left) {
    const uint32_t* row32 = reinterpret_cast<const uint32_t*>(row);
    uint64_t is_left = 0ull;
    for (int w = begin >> 2; w <= (end - 1) >> 2; ++w) {
        const uint32_t v = row32[w];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const uint32_t a = (v >> (8 * b)) & 0x3fu;  // (padding bytes are zero; masked by `range` below)
            is_left |= ((left >> a) & 1ull) << (4 * w + b);
        }
    }
    const uint64_t below_end = end >= 64 ? ~0ull : ((1ull << end) - 1ull);
    const uint64_t range = below_end & ~((1ull << begin) - 1ull);
    const uint64_t zone = ((1ull << (begin + n_left)) - 1ull) & ~((1ull << begin) - 1ull);  // begin + n_left < end <= 64
    uint64_t bad_lo = ~is_left & zone, bad_hi = is_left & range & ~zone;
    while (bad_lo != 0ull && bad_hi != 0ull) {
        const int p = __ffsll((long long)bad_lo) - 1, q = 63 - __clzll((long long)bad_hi);
        const uint8_t tp = row[p];
        row[p] = row[q];
        row[q] = tp;
        bad_lo &= bad_lo - 1ull;
        bad_hi &= ~(1ull << q);
    }
}

// The same with 32-bit masks (simulators of at most 32 agents: half the instructions).
__device__ __forceinline__ void kd_partition32(uint8_t* row, int begin, int end, int n_left, uint32_t left) {
    const uint32_t* row32 = reinterpret_cast<const uint32_t*>(row);
    uint32_t is_left = 0u;
    for (int w = begin >> 2; w <= (end - 1) >> 2; ++w) {
        const uint32_t v = row32[w];
#pragma unroll
        for (int b = 0; b < 4; ++b) is_left |= ((left >> ((v >> (8 * b)) & 0x1fu)) & 1u) << (4 * w + b);
    }
    const uint32_t below_end = end >= 32 ? ~0u : ((1u << end) - 1u);
    const uint32_t range = below_end & ~((1u << begin) - 1u);
    const uint32_t zone = ((1u << (begin + n_left)) - 1u) & ~((1u << begin) - 1u);
    uint32_t bad_lo = ~is_left & zone, bad_hi = is_left & range & ~zone;
    while (bad_lo != 0u && bad_hi != 0u) {
        const int p = __ffs((int)bad_lo) - 1, q = 31 - __clz((int)bad_hi);
        const uint8_t tp = row[p];
        row[p] = row[q];
        row[q] = tp;
        bad_lo &= bad_lo - 1u;
        bad_hi &= ~(1u << q);
    }
}

// Bounding box of a set of lanes: min and max of two unsigned keys over the 64 lanes of the wave (non-members pass the
// identities), every lane's register ending with the total in lane 63: quad swaps, row rotations, then lane 15 / lane 31
// broadcasts into the following rows.  Hand-placed DPP: hipcc emits v_mov + v_mov_dpp + v_min + s_nop per step; here each
// step is ONE v_min/max_u32_dpp, and the four independent reductions are interleaved so that a register is read by a DPP
// operand three instructions after it was written (the hazard needs two wait states): 24 instructions instead of 96.
__device__ __forceinline__ void kd_wave_bbox(uint32_t& min_x, uint32_t& max_x, uint32_t& min_y, uint32_t& max_y) {
#define CN_KD_STEP(ctrl)                            \
    "v_min_u32_dpp %0, %0, %0 " ctrl "\n"           \
    "v_max_u32_dpp %1, %1, %1 " ctrl "\n"           \
    "v_min_u32_dpp %2, %2, %2 " ctrl "\n"           \
    "v_max_u32_dpp %3, %3, %3 " ctrl "\n"
    asm volatile("s_nop 4\n"  // (a VALU write of EXEC or of these registers just before: worst case 5 wait states)
                 CN_KD_STEP("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                 CN_KD_STEP("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                 CN_KD_STEP("row_ror:4 row_mask:0xf bank_mask:0xf")
                 CN_KD_STEP("row_ror:8 row_mask:0xf bank_mask:0xf")
                 CN_KD_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 CN_KD_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 "s_nop 1\n"
                 : "+v"(min_x), "+v"(max_x), "+v"(min_y), "+v"(max_y));
#undef CN_KD_STEP
    min_x = (uint32_t)__builtin_amdgcn_readlane((int)min_x, 63);
    max_x = (uint32_t)__builtin_amdgcn_readlane((int)max_x, 63);
    min_y = (uint32_t)__builtin_amdgcn_readlane((int)min_y, 63);
    max_y = (uint32_t)__builtin_amdgcn_readlane((int)max_y, 63);
}

}  // namespace cn
