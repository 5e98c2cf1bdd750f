// Attainable fp32 matrix-core rate: every wave issues independent v_mfma_f32_32x32x2_f32 back to back
// (4 accumulators, no memory traffic) for a fixed count; reports TFLOP/s at 1..4 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o tools/micro/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void spin(int iters, float *out)
{
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    if (s == 12345.f) out[0] = s;
}
int main()
{
    float *out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs_per_cu = 1; wgs_per_cu <= 4; ++wgs_per_cu)
        for (int iters : {2000, 20000}) {
            const int grid = 256 * wgs_per_cu;
            hipLaunchKernelGGL(spin, dim3(grid), dim3(256), 0, 0, 100, out);
            hipEventRecord(e0);
            hipLaunchKernelGGL(spin, dim3(grid), dim3(256), 0, 0, iters, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)grid * 4 * iters * 4 * 4096.0;
            printf("%d waves/SIMD, %6d iters: %8.1f us  %6.1f TFLOP/s\n", wgs_per_cu, iters, ms * 1e3, flop / ms / 1e9);
        }
    return 0;
}
