// What the write path sustains for the correlation volume's address pattern, by store width (round 6).
// 1024 workgroups x 4 waves; a wave owns 64 (96) rows of a [pairs][6144][6144] fp32 volume (rows 24 KB apart) and walks them
// column block by column block (128 bytes per row and block), as corr_vol_split_kernel does.  Per block and wave:
//   x4: 8 (12)  buffer_store_dwordx4, 8 lanes per row  -> 8 whole lines per instruction
//   x2: 16 (24) buffer_store_dwordx2, 16 lanes per row -> 4 whole lines per instruction
//   x1: 32 (48) buffer_store_dword,   32 lanes per row -> 2 whole lines per instruction
// No arithmetic, no LDS: the time is the store path's alone.  9.66 GB per launch (level 0 of 64 pairs).
#include <hip/hip_runtime.h>
#include <cstdio>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));

template <int WIDTH, int ROWS>
__global__ __launch_bounds__(256) void store_kernel(float* out, int N, int groups_per_pair) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int z = blockIdx.x / groups_per_pair, g = blockIdx.x % groups_per_pair;
    const int row0 = g * (4 * ROWS) + wave * ROWS;
    if (row0 >= N) return;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(out + (long)z * N * N), (short)0, (int)((long)N * N * 4), 0x00020000);
    constexpr int LPR = 32 / WIDTH;                 // lanes per row (128 bytes)
    constexpr int RPI = 64 / LPR;                   // rows per instruction
    const int vo = ((row0 + lane / LPR) * N + (lane % LPR) * WIDTH) * 4;
    const int step = RPI * N * 4;
    for (int cb = 0; cb < N / 32; ++cb) {
#pragma unroll
        for (int i = 0; i < ROWS / RPI; ++i) {
            if constexpr (WIDTH == 4) __builtin_amdgcn_raw_buffer_store_b128(v4i{cb, i, lane, 0}, rs, vo, cb * 128 + i * step, 0);
            else if constexpr (WIDTH == 2) __builtin_amdgcn_raw_buffer_store_b64(v2i{cb, i}, rs, vo, cb * 128 + i * step, 0);
            else __builtin_amdgcn_raw_buffer_store_b32(cb, rs, vo, cb * 128 + i * step, 0);
        }
    }
}

int main() {
    const int N = 6144, pairs = 64;
    float* buf = nullptr;
    if (hipMalloc(&buf, (size_t)pairs * N * N * 4) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    auto timed = [&](auto launch, const char* name) {
        launch();
        (void)hipEventRecord(e0, 0);
        for (int i = 0; i < 3; ++i) launch();
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        ms /= 3;
        printf("%-44s %7.3f ms  %5.2f TB/s\n", name, ms, (double)pairs * N * N * 4 / (ms * 1e-3) / 1e12);
    };
    timed([&] { hipLaunchKernelGGL((store_kernel<4, 64>), dim3(pairs * 24), dim3(256), 0, 0, buf, N, 24); }, "dwordx4, 64 rows per wave (8 lines / instr)");
    timed([&] { hipLaunchKernelGGL((store_kernel<2, 64>), dim3(pairs * 24), dim3(256), 0, 0, buf, N, 24); }, "dwordx2, 64 rows per wave (4 lines / instr)");
    timed([&] { hipLaunchKernelGGL((store_kernel<1, 64>), dim3(pairs * 24), dim3(256), 0, 0, buf, N, 24); }, "dword,   64 rows per wave (2 lines / instr)");
    timed([&] { hipLaunchKernelGGL((store_kernel<4, 96>), dim3(pairs * 16), dim3(256), 0, 0, buf, N, 16); }, "dwordx4, 96 rows per wave");
    timed([&] { hipLaunchKernelGGL((store_kernel<1, 96>), dim3(pairs * 16), dim3(256), 0, 0, buf, N, 16); }, "dword,   96 rows per wave");
    return 0;
}
