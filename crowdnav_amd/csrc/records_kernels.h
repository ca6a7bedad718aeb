// Shard-boundary kernels: what Explorer.run_k_episodes does with the finished episodes AFTER its loop
//   /root/reference crowd_nav/utils/explorer.py:50-62 (per-episode outcome / nav time), :71-72 (discounted return),
//   :74-90 (success / collision rate, average nav time, average return, danger frequency)
// on the per-env record rings cn_rollout / cn_rollout_step filled.  One pack kernel turns the six typed record arrays
// into ONE fixed-size float64 block per env (the unit the RCCL all-gather moves between shards), one single-workgroup
// kernel reduces any such block — a shard's own or the gathered one — to the eight numbers of the log line in a fixed
// order (bitwise reproducible, no atomics).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/crowdnav_amd.h"
#include "orca_device.h"

namespace cn {

constexpr int kRecordFields = CN_RECORD_FIELDS;  // outcome, steps, discounted return, nav time, danger steps, danger dmin sum

__host__ __device__ inline size_t record_block_doubles(int K) { return 1 + (size_t)K * kRecordFields; }

// blocks [B][1 + K * 6] f64: block b = { episodes env b has FINISHED (unclamped), then K records }; record j is slot j of
// the env's record RING (the rollout kernels write episode e to slot e % record_capacity): the env's j-th episode while
// count <= record_capacity, afterwards its most recent episode with ordinal = j mod record_capacity; valid while
// j < min(count, record_capacity, K), zeros otherwise.  (The rollout kernels' own epilogue writes the same blocks when
// cn_rollout_io.blocks is set: step_kernels.h, rollout_epilogue.)
__global__ void records_pack_kernel(int B, int K, const cn_rollout_io* io_dev, double* blocks) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * K) return;
    const cn_rollout_io* io = io_dev;
    const int b = idx / K, j = idx - b * K;
    const int cap = io->record_capacity;
    const int n = io->ep_count[b];
    double* blk = blocks + (size_t)b * record_block_doubles(K);
    if (j == 0) blk[0] = (double)n;
    double* r = blk + 1 + (size_t)j * kRecordFields;
    if (j < n && j < cap) {
        const size_t k = (size_t)b * cap + j;
        r[0] = io->ep_outcome ? (double)io->ep_outcome[k] : 0.0;
        r[1] = io->ep_steps ? (double)io->ep_steps[k] : 0.0;
        r[2] = io->ep_return ? io->ep_return[k] : 0.0;
        r[3] = io->ep_time ? io->ep_time[k] : 0.0;
        r[4] = io->ep_danger ? (double)io->ep_danger[k] : 0.0;
        r[5] = io->ep_danger_dmin_sum ? io->ep_danger_dmin_sum[k] : 0.0;
    } else {
        for (int f = 0; f < kRecordFields; ++f) r[f] = 0.0;
    }
}

constexpr int kSummaryThreads = 256;   // threads per workgroup
constexpr int kSummaryBlocks = 64;     // workgroups of one summary launch
constexpr int kSummaryFields = CN_SUMMARY_FIELDS;

// summary[8] = episodes finished, records held, ReachGoal / Collision / Timeout among them, sum of the successful nav
// times, sum of the discounted returns, sum of the Danger steps.  Thread = one (env, record) item per grid stride (all
// loads of an item are independent: one memory round trip instead of the dependent chain a per-env loop makes), fixed tree
// per workgroup, per-workgroup partials to `scratch`, and the LAST workgroup to arrive (ticket) adds the partials in
// workgroup order: the same bits for the same input on every run and every rank, without a second launch.
//   scratch: double [kSummaryBlocks][8] followed by one unsigned ticket counter (zero before the first launch; the
//   last workgroup leaves it zero again)
// Where a record comes from: packed blocks (a shard's own or the gathered ones) ...
struct BlockRecords {
    const double* blocks;
    size_t stride;
    __device__ void get(int64_t b, int j, double& n, double& outcome, double& ret, double& time, double& danger) const {
        const double* blk = blocks + (size_t)b * stride;
        const double* r = blk + 1 + (size_t)j * kRecordFields;
        n = blk[0], outcome = r[0], ret = r[2], time = r[3], danger = r[4];
    }
};
// ... or the engine's own record rings, straight from the rollout io block (cn_rollout_summary: a single engine needs no blocks)
struct RingRecords {
    const cn_rollout_io* io;
    __device__ void get(int64_t b, int j, double& n, double& outcome, double& ret, double& time, double& danger) const {
        const size_t k = (size_t)b * io->record_capacity + j;
        n = (double)io->ep_count[b];
        outcome = io->ep_outcome ? (double)io->ep_outcome[k] : 0.0;
        ret = io->ep_return ? io->ep_return[k] : 0.0;
        time = io->ep_time ? io->ep_time[k] : 0.0;
        danger = io->ep_danger ? (double)io->ep_danger[k] : 0.0;
    }
};

template <class Records>
__global__ __launch_bounds__(kSummaryThreads) void records_summary_kernel(int64_t n_envs, int K, int capacity, Records src,
                                                                          double* summary, double* scratch) {
    __shared__ double part[kSummaryFields][kSummaryThreads];
    __shared__ unsigned ticket;
    double acc[kSummaryFields] = {};
    const int64_t items = n_envs * K;
    for (int64_t it = (int64_t)blockIdx.x * kSummaryThreads + threadIdx.x; it < items;
         it += (int64_t)kSummaryBlocks * kSummaryThreads) {
        const int64_t b = it / K;
        const int j = (int)(it - b * K);
        double n, r0, r2, r3, r4;
        src.get(b, j, n, r0, r2, r3, r4);
        const bool held = (double)j < n && j < capacity;
        const int outcome = (int)r0;
        if (j == 0) acc[0] += n;
        acc[1] += held ? 1.0 : 0.0;
        acc[2] += (held && outcome == CN_REACH_GOAL) ? 1.0 : 0.0;
        acc[3] += (held && outcome == CN_COLLISION) ? 1.0 : 0.0;
        acc[4] += (held && outcome == CN_TIMEOUT) ? 1.0 : 0.0;
        acc[5] += (held && outcome == CN_REACH_GOAL) ? r3 : 0.0;
        acc[6] += held ? r2 : 0.0;
        acc[7] += held ? r4 : 0.0;
    }
    for (int f = 0; f < kSummaryFields; ++f) part[f][threadIdx.x] = acc[f];
    __syncthreads();
    for (int half = kSummaryThreads / 2; half > 0; half >>= 1) {
        if ((int)threadIdx.x < half)
            for (int f = 0; f < kSummaryFields; ++f) part[f][threadIdx.x] += part[f][threadIdx.x + half];
        __syncthreads();
    }
    unsigned* counter = reinterpret_cast<unsigned*>(scratch + kSummaryBlocks * kSummaryFields);
    if (threadIdx.x < kSummaryFields) scratch[blockIdx.x * kSummaryFields + threadIdx.x] = part[threadIdx.x][0];
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) ticket = atomicAdd(counter, 1u);
    __syncthreads();
    if (ticket != kSummaryBlocks - 1) return;
    __threadfence();
    if (threadIdx.x < kSummaryFields) {
        double total = 0.0;
        for (int w = 0; w < kSummaryBlocks; ++w)
            total += __builtin_nontemporal_load(scratch + w * kSummaryFields + threadIdx.x);
        summary[threadIdx.x] = total;
    }
    if (threadIdx.x == 0) *counter = 0u;
}

// The same reduction for SMALL inputs (a single engine's rings at the shard boundary of a short run: 4096 envs x 1 record) in
// ONE workgroup of 1024 threads: no partials in global memory, no ticket, no fence — the two-level kernel above spends 8 us on
// 4096 items, most of it in the hand-off between its 64 workgroups; this one is a strided pass and a fixed LDS tree.  The
// summation order is a function of (n_envs, K) only: the same bits on every run, for rings and for packed blocks alike
// (cn_rollout_summary == cn_records_summary over cn_rollout_records, as tested).
constexpr int kSummarySmallThreads = 1024;
constexpr int64_t kSummarySmallItems = 16 * kSummarySmallThreads;  // launch_records_summary: up to 16 items per thread
template <class Records>
__global__ __launch_bounds__(kSummarySmallThreads) void records_summary_small_kernel(int64_t n_envs, int K, int capacity, Records src,
                                                                                     double* summary) {
    __shared__ double part[kSummaryFields][kSummarySmallThreads / kWave];
    double acc[kSummaryFields] = {};
    const int64_t items = n_envs * K;
    // four items of a thread at a time, every load of the four requested before the first sum: behind a rollout launch the
    // records sit in memory (written through other XCDs' L2s), and one round trip per item was the larger part of this
    // kernel at the shard boundary.  The items are still added in increasing order: the same bits.
    constexpr int kBatch = 4;
    for (int64_t it0 = threadIdx.x; it0 < items; it0 += (int64_t)kBatch * kSummarySmallThreads) {
        double n[kBatch], r0[kBatch], r2[kBatch], r3[kBatch], r4[kBatch];
        int jj[kBatch];
        bool have[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const int64_t it = it0 + (int64_t)u * kSummarySmallThreads;
            have[u] = it < items;
            const int64_t itc = have[u] ? it : it0;
            const int64_t b = itc / K;
            jj[u] = (int)(itc - b * K);
            src.get(b, jj[u], n[u], r0[u], r2[u], r3[u], r4[u]);
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            if (!have[u]) continue;
            const int j = jj[u];
            const bool held = (double)j < n[u] && j < capacity;
            const int outcome = (int)r0[u];
            if (j == 0) acc[0] += n[u];
            acc[1] += held ? 1.0 : 0.0;
            acc[2] += (held && outcome == CN_REACH_GOAL) ? 1.0 : 0.0;
            acc[3] += (held && outcome == CN_COLLISION) ? 1.0 : 0.0;
            acc[4] += (held && outcome == CN_TIMEOUT) ? 1.0 : 0.0;
            acc[5] += (held && outcome == CN_REACH_GOAL) ? r3[u] : 0.0;
            acc[6] += held ? r2[u] : 0.0;
            acc[7] += held ? r4[u] : 0.0;
        }
    }
    // lanes of a wave by a fixed shuffle tree, the 16 waves in index order
#pragma unroll
    for (int f = 0; f < kSummaryFields; ++f) {
        double v = acc[f];
#pragma unroll
        for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off);
        if ((threadIdx.x & (kWave - 1)) == 0) part[f][threadIdx.x / kWave] = v;
    }
    __syncthreads();
    if (threadIdx.x < kSummaryFields) {
        double total = 0.0;
        for (int w = 0; w < kSummarySmallThreads / kWave; ++w) total += part[threadIdx.x][w];
        summary[threadIdx.x] = total;
    }
}

}  // namespace cn
