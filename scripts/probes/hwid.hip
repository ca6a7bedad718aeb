// Which SIMD does wave w of a 1024-thread workgroup run on?  (HW_REG_HW_ID: [3:0] wave slot, [5:4] SIMD, [11:8] CU)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(1024) void probe(unsigned* out) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = id;
}
int main() {
    unsigned* d;
    const int blocks = 512;
    hipMalloc(&d, blocks * 16 * sizeof(unsigned));
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(1024), 100 * 1024, 0, d);  // 100 KB LDS: one workgroup per CU
    hipDeviceSynchronize();
    unsigned h[blocks * 16];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int pattern_mod = 0, pattern_div = 0, other = 0;
    for (int b = 0; b < blocks; ++b) {
        bool mod = true, div = true;
        for (int w = 0; w < 16; ++w) {
            const unsigned simd = (h[b * 16 + w] >> 4) & 3;
            mod = mod && simd == (unsigned)((w + ((h[b * 16] >> 4) & 3)) % 4);
            div = div && simd == (unsigned)(w / 4);
        }
        pattern_mod += mod, pattern_div += div, other += (!mod && !div);
    }
    printf("blocks %d: simd = (w + simd0) %% 4 in %d, simd = w / 4 in %d, other %d\n", blocks, pattern_mod, pattern_div, other);
    for (int b = 0; b < 3; ++b) {
        printf("block %d simd of waves 0..15:", b);
        for (int w = 0; w < 16; ++w) printf(" %u", (h[b * 16 + w] >> 4) & 3);
        printf("  cu %u\n", (h[b * 16] >> 8) & 15);
    }
    return 0;
}
