// Host side shared by the translation units of libcrowdnav_amd.so: the engine object behind the opaque cn_engine handle, error
// text, the slab allocator and the launch helpers.  crowdnav_amd.hip holds the env / rollout entry points and their kernels,
// sarl_abi.hip the value-network decision (cn_sarl_*) and its kernels: two code objects, compiled in parallel, no device
// symbol shared between them.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <new>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../include/crowdnav_amd.h"
#include "step_kernels.h"


// ------------------------------------------------------------------------------------------------ C ABI

// thread-local text of the last failure (cn_last_error); one definition, in crowdnav_amd.hip
extern thread_local char cn_g_err[512];

namespace {

[[maybe_unused]] int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(cn_g_err, sizeof(cn_g_err), fmt, ap);
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
    cn::ScenarioCfg C;
    cn::StateView S;
    hipStream_t stream;
    cn_rollout_io io_host;   // last cn_rollout_io uploaded to io_dev
    cn_rollout_io* io_dev;   // device copy the rollout kernels read through
    cn::StateView* S_dev;    // device copy of S (the fused rollout kernel re-reads state pointers instead of holding them)
    // CN_FLAG_ASYNC_SCENARIO_FILL: fill kernels go round-robin over side streams (a launch stuck on a hard scenario must
    // not hold back the next one) and start once the previous transition kernel has written its episode counters
    static constexpr int kFillStreams = 8;
    bool async_fill;
    hipStream_t fill_streams[kFillStreams];
    int* fill_list = nullptr;    // [kFillStreams][2 + 2 B D] job lists of ring_fill_scan_kernel / ring_fill_jobs_kernel, one per side stream
    int fill_queue_wgs = 0;      // persistent generator workgroups of an asynchronous fill launch (0: one workgroup per slot)
    hipEvent_t rollout_done;
    int next_fill_stream;
    bool io_valid;
    bool rollout_begun = false;          // cn_rollout_begin has run: the seed numbering below is fixed until the next one
    uint32_t begin_seed_base = 0, begin_seed_mod = 0;  // (the scenario cache is sized and keyed by them)
    int steps_since_fill;    // transitions launched since the scenario ring was last topped up; < 0 = never filled
    struct cn_sarl* sarl;    // SARL decision state (sarl_abi.hip), NULL until cn_sarl_configure
    bool orca_fresh;         // cn_sarl_sample_step: the humans' ORCA velocities of the CURRENT state are in sarl->orca_vel (left by the
                             // previous call's transition kernel); cleared by every other entry point (bind)
    double* discount;
    int discount_len;
    uint32_t* probe_key;
    double* summary_scratch;  // records_summary_kernel: per-workgroup partials + ticket counter
    int maxl;          // half-planes held in VGPRs by the solve phase: 5 or 10
    bool gen_wave;     // wave-per-scenario generators (64 rejection attempts at a time): long chains, H > 8
    size_t smem;       // dynamic LDS bytes per workgroup
    int sched_min, sched_slots, sched_reserve, dyn_visits;  // the 20-human shard kernel's schedules (launch_rollout): shortest call split, resident workgroups, slots left free beside the asynchronous fill
    bool sched_force, sched_dynamic;
    bool scenario_cache;  // wave generators keep the scenarios of a small seed set (CROWDNAV_AMD_SCENARIO_CACHE)
    uint64_t launch_counts[CN_LAUNCH_COUNTERS];  // cn_launch_counts: what the host enqueued since cn_create
    // Device memory comes from a few large slabs, not one hipMalloc per buffer: an engine has ~60 device buffers, most of them a
    // few KiB; one 32 MiB slab (plus one per buffer larger than that) is 2-4 mappings to create and - each hipFree being a device
    // synchronisation - 2-4 to tear down, and cn_sarl_configure can roll a failed configuration back to a mark.
    struct Slab {
        char* base;
        size_t size, used;
    };
    std::vector<Slab> slabs;
    struct AllocMark {  // fill level of every slab that existed at the mark
        std::vector<size_t> used;
    };
    AllocMark alloc_mark() const {
        AllocMark m;
        for (const Slab& s : slabs) m.used.push_back(s.used);
        return m;
    }
    void alloc_rollback(const AllocMark& m) {  // frees everything allocated since alloc_mark()
        while (slabs.size() > m.used.size()) {
            (void)hipFree(slabs.back().base);
            slabs.pop_back();
        }
        for (size_t i = 0; i < slabs.size(); ++i) slabs[i].used = m.used[i];
    }
};

namespace {

constexpr size_t kSlabBytes = (size_t)32 << 20, kSlabAlign = 4096;

template <typename T>
int dev_alloc(cn_engine* e, T** out, size_t n) {
    const size_t bytes = (n * sizeof(T) + kSlabAlign - 1) / kSlabAlign * kSlabAlign;
    // first fit over ALL slabs: a buffer larger than a slab gets one of its own, exactly sized (so it is full and never
    // chosen again), and the small buffers that follow keep filling the shared slab they were filling before — looking at
    // the newest slab only stranded up to 32 MiB at every large / small alternation of cn_sarl_configure
    size_t k = 0;
    while (k < e->slabs.size() && e->slabs[k].used + bytes > e->slabs[k].size) ++k;
    if (k == e->slabs.size()) {
        const size_t size = bytes > kSlabBytes ? (bytes + ((size_t)2 << 20) - 1) >> 21 << 21 : kSlabBytes;
        void* p = nullptr;
        CN_HIP(hipMalloc(&p, size));
        e->slabs.push_back({static_cast<char*>(p), size, 0});
    }
    cn_engine::Slab& sl = e->slabs[k];
    void* p = sl.base + sl.used;
    sl.used += bytes;
    CN_HIP(hipMemset(p, 0, bytes));
    *out = static_cast<T*>(p);
    return CN_OK;
}

[[maybe_unused]] int bind(cn_engine* e) {
    if (!e) return fail(CN_ERR_INVALID, "engine is NULL");
    e->orca_fresh = false;  // whatever this call is, it may change the state those velocities belong to
    CN_HIP(hipSetDevice(e->cfg.device));
    return CN_OK;
}

inline int grid_envs(const cn_engine* e) { return (e->P.B + e->P.E - 1) / e->P.E; }

// launch a kernel template instantiated for the engine's half-plane capacity (and with / without the kd-tree bookkeeping of
// simulators with more than 10 agents)
#define CN_LAUNCH_MAXL(e, kernel, grid, ...)                                                                        \
    do {                                                                                                            \
        const dim3 g__(grid), b__((e)->P.threads);                                                                  \
        if ((e)->maxl == 5 && !(e)->P.kd)                                                                           \
            hipLaunchKernelGGL((cn::kernel<5, false>), g__, b__, (e)->smem, (e)->stream, __VA_ARGS__);               \
        else if ((e)->maxl == 5)                                                                                    \
            hipLaunchKernelGGL((cn::kernel<5, true>), g__, b__, (e)->smem, (e)->stream, __VA_ARGS__);                \
        else if (!(e)->P.kd)                                                                                        \
            hipLaunchKernelGGL((cn::kernel<10, false>), g__, b__, (e)->smem, (e)->stream, __VA_ARGS__);              \
        else                                                                                                        \
            hipLaunchKernelGGL((cn::kernel<10, true>), g__, b__, (e)->smem, (e)->stream, __VA_ARGS__);               \
    } while (0)

// ... and for the robot kinematics (the unicycle code only exists in the <.., true, ..> instantiations).  K: the kernel
// template takes <MAXL, UNI, KD> (step_kernel) — rollout_kernel has HEADLINE in between, see CN_LAUNCH_ROLLOUT
#define CN_LAUNCH_MAXL_UNI(e, kernel, grid, ...)                                                                    \
    do {                                                                                                            \
        const dim3 g__(grid), b__((e)->P.threads);                                                                  \
        const int v__ = ((e)->maxl == 5 ? 0 : 4) | ((e)->P.robot_unicycle ? 2 : 0) | ((e)->P.kd ? 1 : 0);           \
        switch (v__) {                                                                                              \
            case 0: hipLaunchKernelGGL((cn::kernel<5, false, false>), g__, b__, (e)->smem, (e)->stream, __VA_ARGS__); break;  \
            case 1: hipLaunchKernelGGL((cn::kernel<5, false, true>), g__, b__, (e)->smem, (e)->stream, __VA_ARGS__); break;   \
            case 2: hipLaunchKernelGGL((cn::kernel<5, true, false>), g__, b__, (e)->smem, (e)->stream, __VA_ARGS__); break;   \
            case 3: hipLaunchKernelGGL((cn::kernel<5, true, true>), g__, b__, (e)->smem, (e)->stream, __VA_ARGS__); break;    \
            case 4: hipLaunchKernelGGL((cn::kernel<10, false, false>), g__, b__, (e)->smem, (e)->stream, __VA_ARGS__); break; \
            case 5: hipLaunchKernelGGL((cn::kernel<10, false, true>), g__, b__, (e)->smem, (e)->stream, __VA_ARGS__); break;  \
            case 6: hipLaunchKernelGGL((cn::kernel<10, true, false>), g__, b__, (e)->smem, (e)->stream, __VA_ARGS__); break;  \
            default: hipLaunchKernelGGL((cn::kernel<10, true, true>), g__, b__, (e)->smem, (e)->stream, __VA_ARGS__); break;  \
        }                                                                                                           \
    } while (0)

#define CN_LAUNCH_ROLLOUT(e, grid, ...)                                                                             \
    do {                                                                                                            \
        const dim3 g__(grid), b__((e)->P.threads);                                                                  \
        const int v__ = ((e)->maxl == 5 ? 0 : 4) | ((e)->P.robot_unicycle ? 2 : 0) | ((e)->P.kd ? 1 : 0);           \
        switch (v__) {                                                                                              \
            case 0: hipLaunchKernelGGL((cn::rollout_kernel<5, false, false, false>), g__, b__, (e)->smem, (e)->stream, __VA_ARGS__); break;  \
            case 1: hipLaunchKernelGGL((cn::rollout_kernel<5, false, false, true>), g__, b__, (e)->smem, (e)->stream, __VA_ARGS__); break;   \
            case 2: hipLaunchKernelGGL((cn::rollout_kernel<5, true, false, false>), g__, b__, (e)->smem, (e)->stream, __VA_ARGS__); break;   \
            case 3: hipLaunchKernelGGL((cn::rollout_kernel<5, true, false, true>), g__, b__, (e)->smem, (e)->stream, __VA_ARGS__); break;    \
            case 4: hipLaunchKernelGGL((cn::rollout_kernel<10, false, false, false>), g__, b__, (e)->smem, (e)->stream, __VA_ARGS__); break; \
            case 5: hipLaunchKernelGGL((cn::rollout_kernel<10, false, false, true>), g__, b__, (e)->smem, (e)->stream, __VA_ARGS__); break;  \
            case 6: hipLaunchKernelGGL((cn::rollout_kernel<10, true, false, false>), g__, b__, (e)->smem, (e)->stream, __VA_ARGS__); break;  \
            default: hipLaunchKernelGGL((cn::rollout_kernel<10, true, false, true>), g__, b__, (e)->smem, (e)->stream, __VA_ARGS__); break;  \
        }                                                                                                           \
    } while (0)

[[maybe_unused]] int env_int(const char* name, int fallback) {
    const char* v = std::getenv(name);
    return (v && *v) ? std::atoi(v) : fallback;
}
inline int grid_lanes(const cn_engine* e) { return (e->P.B + cn::kWave - 1) / cn::kWave; }

}  // namespace


// sarl_abi.hip: frees the host part of the SARL state (device buffers are the engine's slabs)
void cn_sarl_release(cn_engine* e);
// crowdnav_amd.hip: launches orca_kernel (the kernels of step_kernels.h are instantiated in that translation unit only)
void cn_launch_orca(cn_engine* e, float* out_vel);
