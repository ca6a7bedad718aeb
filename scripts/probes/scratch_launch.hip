// Does a scratch reservation cost launch time?  (sarl_narrow_kernel reserves 2 KB per lane for the float64 libm's
// private arrays.)  Two kernels that do the same trivial work on 27 workgroups of 512 threads, one of them with a 2 KB private
// array it indexes dynamically (scratch), streamed 2000 times each, alone and alternating with a scratch-free kernel.
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/scratch_launch.hip -o build/exp/scratch_launch && build/exp/scratch_launch
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void plain(float* out, int n) {
    out[blockIdx.x * 512 + threadIdx.x] = (float)n;
}
__global__ __launch_bounds__(512) void with_scratch(float* out, int n) {
    float a[512];
    float s = 0.0f;
    if (n < 0) {  // never taken at run time, not provable at compile time: the array is reserved, the launch does not touch it
        for (int i = 0; i < 512; ++i) a[i] = (float)(i * n);
        for (int i = 0; i < 64; ++i) s += a[(threadIdx.x * 7 + i * n) & 511];
    }
    out[blockIdx.x * 512 + threadIdx.x] = (float)n + s;
}
int main() {
    float* out;
    (void)hipMalloc(&out, 27 * 512 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int N = 2000;
    auto run = [&](const char* name, int mode) {
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            for (int i = 0; i < N; ++i) {
                if (mode == 0) plain<<<27, 512>>>(out, i);
                if (mode == 1) with_scratch<<<27, 512>>>(out, i);
                if (mode == 2) { with_scratch<<<27, 512>>>(out, i); plain<<<1, 64>>>(out, i); }
                if (mode == 3) { plain<<<27, 512>>>(out, i); plain<<<1, 64>>>(out, i); }
            }
            (void)hipEventRecord(e1);
            (void)hipDeviceSynchronize();
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%-58s %.2f us per iteration\n", name, ms * 1e3 / N);
        }
    };
    run("plain kernel, streamed", 0);
    run("kernel with 2 KB/lane of scratch, streamed", 1);
    run("scratch kernel alternating with a one-wave plain kernel", 2);
    run("plain kernel alternating with a one-wave plain kernel", 3);
    return 0;
}
