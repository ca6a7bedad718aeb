// How many one-wave workgroups with N bytes of dynamic LDS fit a CU (gfx950: 160 KB)?  Prints the occupancy the runtime
// computes for a range of sizes: the step where it drops gives the LDS allocation granule.
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/lds_granule.hip -o build/exp/lds_granule
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void k(float* out) {
    extern __shared__ float sm[];
    sm[threadIdx.x] = threadIdx.x;
    __syncthreads();
    out[threadIdx.x] = sm[63 - threadIdx.x];
}
int main() {
    int last = -1;
    for (int bytes = 9216; bytes <= 21504; bytes += 128) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 64, bytes) != hipSuccess) return 1;
        if (n != last) printf("dynamic LDS %6d B -> %d workgroups per CU\n", bytes, n);
        last = n;
    }
    return 0;
}
