// Are HBM reads on gfx950 served in 128-byte lines or in smaller sectors?  (round-5 verdict, item 4: a 4x4 pyramid block = 64 B would
// save 23 % of the lookup's read bytes IF a 64-byte request only moves 64 bytes.)
// Every kernel walks the same 4 GiB span (16x the Infinity Cache) line by line and reads only the first PART bytes of every 128-byte
// line with 16-byte lane loads: PART = 128 (whole lines), 64, 32.  If memory hands over whole lines, the three take the same time and
// FETCH_SIZE / TCC_EA0_RDREQ report the same traffic; if 64-byte (32-byte) sectors exist, the partial walks are faster and the
// 32-byte request counter moves.  Run through tools/sector_probe.sh.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef float f4v __attribute__((ext_vector_type(4)));
template <int PART>
__global__ void part_line_kernel(const f4v* __restrict__ p, size_t n_lines, float* sink) {
    constexpr int LPL = PART / 16;                         // lanes per line
    const size_t gl = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = ((size_t)gridDim.x * blockDim.x) / LPL;
    float acc = 0.f;
    for (size_t line = gl / LPL; line < n_lines; line += stride) {
        const f4v v = __builtin_nontemporal_load(p + line * 8 + (gl % LPL));
        acc += v.x;
    }
    if (acc == -1.f) *sink = acc;
}

int main() {
    const size_t bytes = 4ull << 30;
    void* buf = nullptr;
    float* sink = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
    (void)hipMemset(buf, 0, bytes);
    const size_t n_lines = bytes / 128;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    auto timed = [&](auto launch, const char* name, size_t useful) {
        launch();
        (void)hipEventRecord(e0, 0);
        for (int i = 0; i < 3; ++i) launch();
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        ms /= 3;
        printf("%-28s %8.1f us   useful %5.2f GiB -> %6.2f TB/s useful, %6.2f TB/s if whole lines move\n", name, ms * 1e3, useful / 1073741824.0,
               useful / (ms * 1e-3) / 1e12, (double)bytes / (ms * 1e-3) / 1e12);
    };
    const int grid = 256 * 32, block = 256;
    timed([&] { hipLaunchKernelGGL(part_line_kernel<128>, dim3(grid), dim3(block), 0, 0, (const f4v*)buf, n_lines, sink); }, "128 of 128 bytes per line", bytes);
    timed([&] { hipLaunchKernelGGL(part_line_kernel<64>, dim3(grid), dim3(block), 0, 0, (const f4v*)buf, n_lines, sink); }, "64 of 128 bytes per line", bytes / 2);
    timed([&] { hipLaunchKernelGGL(part_line_kernel<32>, dim3(grid), dim3(block), 0, 0, (const f4v*)buf, n_lines, sink); }, "32 of 128 bytes per line", bytes / 4);
    return 0;
}
