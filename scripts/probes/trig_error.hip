// Exhaustive error of the hardware V_COS_F32 / V_SIN_F32 (input in revolutions) on the inputs the scenario generators'
// float32 stages feed them: f = k 2^-27, k = 0 .. 2^27 - 1 (crowdnav_amd/csrc/scenario_wave.h: kTrigAbsError).
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/trig_error.hip -o build/exp/trig_error && build/exp/trig_error
// prints the maximum of |cosf_hw(f) - cos(2 pi f)| and |sinf_hw(f) - sin(2 pi f)| (float64 reference on the device) and the
// bound the generator assumes; exit code 1 if the bound does not hold.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>

__global__ void probe(unsigned* max_bits) {
    const uint32_t stride = gridDim.x * blockDim.x;
    float worst = 0.0f;
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < (1u << 27); k += stride) {
        const float f = (float)k * 0x1p-27f;
        const double a = (double)k * 0x1p-27 * 6.283185307179586476925286766559;
        const float ec = fabsf((float)((double)__builtin_amdgcn_cosf(f) - cos(a)));
        const float es = fabsf((float)((double)__builtin_amdgcn_sinf(f) - sin(a)));
        worst = fmaxf(worst, fmaxf(ec, es));
    }
    atomicMax(max_bits, __float_as_uint(worst));  // non-negative floats order like their bit patterns
}

int main() {
    const float bound = 1.0e-5f;  // = cn::kTrigAbsError
    unsigned* d = nullptr;
    unsigned h = 0;
    if (hipMalloc(&d, sizeof(unsigned)) != hipSuccess) return 2;
    (void)hipMemcpy(d, &h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(4096), dim3(256), 0, 0, d);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    (void)hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost);
    float worst;
    memcpy(&worst, &h, sizeof(worst));
    printf("max abs error of V_COS_F32 / V_SIN_F32 over 2^27 fractions: %.3e   assumed bound (kTrigAbsError): %.1e   %s\n", worst, bound,
           worst <= bound ? "ok" : "VIOLATED");
    return worst <= bound ? 0 : 1;
}
