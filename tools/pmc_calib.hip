// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 per ACCESS WIDTH (tools/pmc_calibrate.sh).
//
// MI355X_MICROARCH.md states that FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane) coalesced
// streaming read and that other widths and WRITE_SIZE are uncalibrated.  Each kernel below moves a KNOWN number of
// bytes (1 GiB, far past the 256 MiB Infinity Cache) with one access width; the ratio counter / known bytes is the
// correction to use for kernels with that access pattern (mask: 4 B/lane reads + 8 B stores; lookup: 4 B gathers of
// 40-byte rows + 4 B/lane stores; warp: 16 B/lane flow reads, byte gathers, 12 B stores).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <typename T>
__global__ void rd_kernel(const T* __restrict__ p, size_t n, float* sink) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (; i < n; i += stride) {
        T v = p[i];
        const unsigned char* b = reinterpret_cast<const unsigned char*>(&v);
        acc += (float)b[0];
    }
    if (acc == -1.f) *sink = acc;
}
// one block-iteration per row: lanes 0..9 of each 16-lane group read a 40-byte row at a pseudo-random row index
// of a [rows][6144] float matrix (the lookup's window rows)
__global__ void gather40_kernel(const float* __restrict__ p, size_t rows, int rowlen, size_t n_gathers, float* sink) {
    size_t g = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 4;
    const int l = threadIdx.x & 15;
    const size_t stride = ((size_t)gridDim.x * blockDim.x) >> 4;
    float acc = 0.f;
    for (; g < n_gathers; g += stride) {
        const size_t row = (g * 2654435761ull) % rows;
        const int col = (int)((g * 40503ull) % (unsigned)(rowlen - 10));
        if (l < 10) acc += p[row * rowlen + col + l];
    }
    if (acc == -1.f) *sink = acc;
}
template <typename T>
__global__ void wr_kernel(T* __restrict__ p, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    T v;
    unsigned char* b = reinterpret_cast<unsigned char*>(&v);
    for (unsigned k = 0; k < sizeof(T); ++k) b[k] = (unsigned char)(k + threadIdx.x);
    for (; i < n; i += stride) p[i] = v;
}
struct alignas(4) B12 { unsigned a, b, c; };

int main() {
    const size_t bytes = 1ull << 30;
    void* buf = nullptr;
    float* sink = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
    (void)hipMemset(buf, 1, bytes);
    const int grid = 256 * 16, block = 256;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(rd_kernel<uint8_t>, dim3(grid), dim3(block), 0, 0, (const uint8_t*)buf, bytes, sink);
        hipLaunchKernelGGL(rd_kernel<float>, dim3(grid), dim3(block), 0, 0, (const float*)buf, bytes / 4, sink);
        hipLaunchKernelGGL(rd_kernel<float2>, dim3(grid), dim3(block), 0, 0, (const float2*)buf, bytes / 8, sink);
        hipLaunchKernelGGL(rd_kernel<float4>, dim3(grid), dim3(block), 0, 0, (const float4*)buf, bytes / 16, sink);
        // 2^24 gathers of 40 B = 671 MB algorithmic; 64-B sectors touched: 1 or 2 per row
        hipLaunchKernelGGL(gather40_kernel, dim3(grid), dim3(block), 0, 0, (const float*)buf, bytes / 4 / 6144, 6144, (size_t)1 << 24, sink);
        hipLaunchKernelGGL(wr_kernel<uint8_t>, dim3(grid), dim3(block), 0, 0, (uint8_t*)buf, bytes);
        hipLaunchKernelGGL(wr_kernel<float>, dim3(grid), dim3(block), 0, 0, (float*)buf, bytes / 4);
        hipLaunchKernelGGL(wr_kernel<float2>, dim3(grid), dim3(block), 0, 0, (float2*)buf, bytes / 8);
        hipLaunchKernelGGL(wr_kernel<B12>, dim3(grid), dim3(block), 0, 0, (B12*)buf, bytes / 12);
        hipLaunchKernelGGL(wr_kernel<float4>, dim3(grid), dim3(block), 0, 0, (float4*)buf, bytes / 16);
    }
    (void)hipDeviceSynchronize();
    printf("known bytes per launch: rd/wr kernels %zu; gather40 %zu algorithmic (40 B x 2^24)\n", bytes, (size_t)40 << 24);
    return 0;
}
