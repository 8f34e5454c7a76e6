// What does it cost to FEED v_mfma_f32_32x32x2_f32 from LDS?  Each wave loops over "blocks": NB MFMAs on four independent accumulators
// whose A / B operands were read from LDS (ds_read_b128) one block earlier.  Variants: operands from registers only (no LDS traffic),
// A from LDS, A and B from LDS; NB = 16 / 32 / 48; 1, 2, 4 waves per SIMD.  Prints TFLOP/s (peak fp32 MFMA: 157.3).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_feed_probe.hip -o /tmp/mfma_fp && /tmp/mfma_fp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));

template <int NB, int FEED, int ACC = 0>   // FEED 0: registers, 1: A from LDS, 2: A and B from LDS; ACC 1: accumulators in AGPRs (inline asm)
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[4 * 64 * 20 * 2 + 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * 64 * 20 * 2; i += 256) lds[i] = 1.0f + (i & 7) * 0.125f;
    __syncthreads();
    const float* base = lds + wave * (64 * 20 * 2) + (lane & 31) * 20 + (lane >> 5) * 4;
    v16f acc[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float4 a[2], b[2];
    a[0] = *reinterpret_cast<const float4*>(base);
    a[1] = *reinterpret_cast<const float4*>(base + 32 * 20);
    b[0] = *reinterpret_cast<const float4*>(base + 64 * 20);
    b[1] = *reinterpret_cast<const float4*>(base + 96 * 20);
    for (int it = 0; it < iters; ++it) {
        float4 an[2] = {a[0], a[1]}, bn[2] = {b[0], b[1]};
        if (FEED >= 1) {
            an[0] = *reinterpret_cast<const float4*>(base + (it % 5) * 8);
            an[1] = *reinterpret_cast<const float4*>(base + 32 * 20 + (it % 5) * 8);
        }
        if (FEED >= 2) {
            bn[0] = *reinterpret_cast<const float4*>(base + 64 * 20 + (it % 5) * 8);
            bn[1] = *reinterpret_cast<const float4*>(base + 96 * 20 + (it % 5) * 8);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < NB / 16; ++r) {
#define MF(C, A, B) \
            if constexpr (ACC == 0) C = __builtin_amdgcn_mfma_f32_32x32x2f32(A, B, C, 0, 0, 0); \
            else if constexpr (ACC == 1) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(C) : "v"(A), "v"(B)); \
            else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(C) : "v"(A), "v"(B));
#define STEP(S)                                                                          \
            MF(acc[0], a[0].S, b[0].S) MF(acc[1], a[0].S, b[1].S) MF(acc[2], a[1].S, b[0].S) MF(acc[3], a[1].S, b[1].S)
            STEP(x) STEP(y) STEP(z) STEP(w)
#undef STEP
#undef MF
        }
        __builtin_amdgcn_sched_barrier(0);
        a[0] = an[0]; a[1] = an[1]; b[0] = bn[0]; b[1] = bn[1];
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NB, int FEED, int ACC = 0>
static void run(const char* what, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000 * 16 / NB;
    for (int wg = 1; wg <= 4; wg *= 2) {
        const int grid = 256 * wg;
        hipLaunchKernelGGL((probe<NB, FEED, ACC>), dim3(grid), dim3(256), 0, 0, out, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<NB, FEED, ACC>), dim3(grid), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)grid * 4 * iters * NB * 4096.0;
        printf("%-22s %2d MFMAs per block, %d wave(s)/SIMD: %7.1f TFLOP/s\n", what, NB, wg, flops / ms / 1e9);
    }
}

int main() {
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * 256 * 8);
    run<16, 0>("operands in registers", out);
    run<16, 1>("A from LDS", out);
    run<16, 2>("A and B from LDS", out);
    run<32, 1>("A from LDS", out);
    run<32, 2>("A and B from LDS", out);
    run<48, 2>("A and B from LDS", out);
    run<16, 0, 1>("registers, AGPR acc", out);
    run<16, 2, 1>("A and B from LDS, AGPR acc", out);
    run<32, 2, 1>("A and B from LDS, AGPR acc", out);
    run<16, 0, 2>("registers, VGPR acc", out);
    run<16, 2, 2>("A and B from LDS, VGPR acc", out);
    run<32, 2, 2>("A and B from LDS, VGPR acc", out);
    return 0;
}
