// Issue-rate probe for v_mfma_f32_32x32x2_f32: N resident waves per SIMD, each running a bare loop of MFMAs on 4 independent
// accumulators (no memory traffic).  Prints TFLOP/s per occupancy, with random and with zero operands.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void probe(float* out, int iters, float a0, float b0) {
    v16f acc[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a = a0 * (threadIdx.x % 7 + 1), b = b0 * (threadIdx.x % 5 + 1);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * 256 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    for (int zero = 0; zero < 2; ++zero)
        for (int wg_per_cu = 1; wg_per_cu <= 4; ++wg_per_cu) {
            const int grid = 256 * wg_per_cu;
            const float a0 = zero ? 0.f : 1.0001f, b0 = zero ? 0.f : 0.9999f;
            hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 0, 0, out, 100, a0, b0);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 0, 0, out, iters, a0, b0);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double flops = (double)grid * 4 * iters * 32 * 4096.0;
            printf("%s operands, %d waves/SIMD: %.3f ms  %.1f TFLOP/s\n", zero ? "zero  " : "random", wg_per_cu, ms, flops / ms / 1e9);
        }
    return 0;
}
