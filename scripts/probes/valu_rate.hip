// VALU issue-rate probe (gfx950): cycles per wave64 instruction of one wave, and SIMD throughput with 1..4 resident waves per
// SIMD, for plain and packed f32 ops, compares, selects and f64 FMAs.  Settles whether a wave64 VALU op occupies its SIMD for
// 2 or 4 cycles — i.e. what "VALU issue peak" the rollout kernels are to be priced against.
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/valu_rate.hip -o build/exp/valu_rate && build/exp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP8(x) x x x x x x x x
#define ITER 20000
template <int KIND>
__global__ __launch_bounds__(64) void k(float* out, unsigned long long* ticks) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b0 = 1.0f, b1 = 1.5f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6}, q = {1.0f, 1.5f};
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7, e = 1.25;
    int r0 = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITER; ++i) {
        if (KIND == 0)  // 8 independent v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));
        if (KIND == 1)  // 8 independent v_pk_fma_f32
            asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                         "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(q));
        if (KIND == 2)  // 8 independent v_mul_f32
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));
        if (KIND == 3)  // 8 independent v_pk_mul_f32
            asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                         "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(q));
        if (KIND == 4)  // 4 x (v_cmp_lt_f32 + v_addc_co_u32): the rank loop's pair
            asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n v_cmp_lt_f32 vcc, %3, %2\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n"
                         "v_cmp_lt_f32 vcc, %4, %2\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n v_cmp_lt_f32 vcc, %5, %2\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n"
                         : "+v"(r0) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4) : "vcc");
        if (KIND == 5)  // 8 independent v_fma_f64
            asm volatile("v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8\n v_fma_f64 %2, %2, %8, %8\n v_fma_f64 %3, %3, %8, %8\n"
                         "v_fma_f64 %4, %4, %8, %8\n v_fma_f64 %5, %5, %8, %8\n v_fma_f64 %6, %6, %8, %8\n v_fma_f64 %7, %7, %8, %8\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(e));
        if (KIND == 6)  // 8 independent v_pk_add_f32
            asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                         "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(q));
        if (KIND == 7)  // 8 dependent v_fma_f32 (latency of a dependent chain)
            asm volatile(REP8("v_fma_f32 %0, %0, %1, %2\n") : "+v"(a0) : "v"(b0), "v"(b1));
        if (KIND == 8)  // 8 independent v_rcp_f32 (transcendental rate)
            asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                         "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        if (KIND == 9)  // 8 s_and_b64 (scalar ALU beside nothing)
            asm volatile(REP8("s_and_b64 s[20:21], s[20:21], s[22:23]\n") ::: "s20", "s21", "s22", "s23", "scc");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.x + p6.x + p7.x + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + r0;
    if (threadIdx.x == 0 && blockIdx.x == 0) *ticks = t1 - t0;
}
template <int KIND>
void run(const char* name, int per_loop) {
    float* out; unsigned long long* ticks;
    hipMalloc(&out, 256 * 4 * 8 * 64 * 4); hipMalloc(&ticks, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-34s", name);
    for (int w : {1, 2, 3, 4, 8}) {
        const int grid = 256 * 4 * w;
        k<KIND><<<grid, 64>>>(out, ticks); hipDeviceSynchronize();
        hipEventRecord(e0); k<KIND><<<grid, 64>>>(out, ticks); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
        // per SIMD: w waves x ITER x per_loop instructions in ms
        const double inst_per_simd = (double)w * ITER * per_loop;
        printf("  w=%d: %.2f ticks/inst/wave, %.2f ns/inst/SIMD", w, (double)t / (ITER * per_loop), ms * 1e6 / inst_per_simd);
    }
    printf("\n");
}
int main() {
    run<0>("v_fma_f32 x8 indep", 8); run<1>("v_pk_fma_f32 x8 indep", 8); run<2>("v_mul_f32 x8 indep", 8);
    run<3>("v_pk_mul_f32 x8 indep", 8); run<6>("v_pk_add_f32 x8 indep", 8); run<4>("v_cmp_lt_f32+v_addc x4", 8);
    run<5>("v_fma_f64 x8 indep", 8); run<7>("v_fma_f32 x8 dependent", 8); run<8>("v_rcp_f32 x8 indep", 8); run<9>("s_and_b64 x8", 8);
    return 0;
}
