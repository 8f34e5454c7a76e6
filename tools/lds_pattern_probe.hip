// What makes a ds_read_b128 / ds_write_b128 pattern conflict-free on gfx950: cycles per instruction for lane -> address tables.
//   hipcc -O3 --offload-arch=gfx950 tools/lds_pattern_probe.hip -o /tmp/lds_probe && /tmp/lds_probe
// One wave per workgroup, one workgroup; 64 back-to-back independent reads (writes) per timed block, s_memtime around it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
#include <functional>

__global__ __launch_bounds__(64) void probe(const int* __restrict__ addr, long long* __restrict__ out, float* sink, int write) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 16384; i += 64) lds[i] = (float)i;
    __syncthreads();
    const int a = addr[threadIdx.x];          // byte offset
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    long long best = 1LL << 60;
    for (int rep = 0; rep < 8; ++rep) {
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_s_barrier();
        const long long t0 = __builtin_readcyclecounter();
        if (!write) {
            f4 v[16];
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) {
#pragma unroll
                for (int u = 0; u < 16; ++u) asm volatile("ds_read_b128 %0, %1" : "=v"(v[u]) : "v"(a));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int u = 0; u < 16; ++u) acc += v[u];
            }
        } else {
#pragma unroll
            for (int u = 0; u < 64; ++u) asm volatile("ds_write_b128 %0, %1" ::"v"(a), "v"(acc) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        const long long t1 = __builtin_readcyclecounter();
        if (t1 - t0 < best) best = t1 - t0;
    }
    if (threadIdx.x == 0) out[0] = best;
    sink[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

int main() {
    int* d_addr; long long* d_out; float* d_sink;
    hipMalloc(&d_addr, 64 * 4); hipMalloc(&d_out, 8); hipMalloc(&d_sink, 64 * 4);
    struct P { std::string name; std::function<int(int)> f; };
    auto rows = [](int ldk, std::function<int(int)> rowof) { return [=](int l) { return (rowof(l & 31) * ldk + (l >> 5) * 4) * 4; }; };
    std::vector<P> ps = {
        {"consecutive rows, stride 20 floats (non-patch BK16)", rows(20, [](int r) { return r; })},
        {"patch PWH=18 (3x3, 16-wide), stride 20", rows(20, [](int r) { return (r / 16) * 18 + r % 16; })},
        {"patch PWH=20 (1x5), stride 20", rows(20, [](int r) { return (r / 16) * 20 + r % 16; })},
        {"patch PWH=16 (5x1), stride 20", rows(20, [](int r) { return (r / 16) * 16 + r % 16; })},
        {"patch PWH=18 shifted by tap (1,1), stride 20", rows(20, [](int r) { return 19 + (r / 16) * 18 + r % 16; })},
        {"consecutive rows, stride 36 floats (non-patch BK32)", rows(36, [](int r) { return r; })},
        {"8x8 patch PWH=10 (3x3), stride 36", rows(36, [](int r) { return (r / 8) * 10 + r % 8; })},
        {"8x8 patch PWH=12 (1x5), stride 36", rows(36, [](int r) { return (r / 8) * 12 + r % 8; })},
        {"8x8 patch PWH=8 (5x1), stride 36", rows(36, [](int r) { return (r / 8) * 8 + r % 8; })},
        {"all lanes same address (broadcast)", [](int) { return 0; }},
        {"lane * 16 B (linear)", [](int l) { return l * 16; }},
        {"lane * 64 B", [](int l) { return l * 64; }},
        {"lane * 256 B (worst)", [](int l) { return l * 256; }},
        // candidate fixes: patch rows whose starts are congruent to 16 tile rows (row stride in floats a multiple of 64 banks)
        {"fix A: patch row pitch 384 floats (PWH=18), stride 20", [](int l) { int r = l & 31; return ((r / 16) * 384 + (r % 16) * 20 + (l >> 5) * 4) * 4; }},
        {"fix A shifted by tap (1,1)", [](int l) { int r = l & 31; return (384 + 20 + (r / 16) * 384 + (r % 16) * 20 + (l >> 5) * 4) * 4; }},
        {"fix B: 8x8 patch, patch row pitch 8*36+? = 320 floats", [](int l) { int r = l & 31; return ((r / 8) * 320 + (r % 8) * 36 + (l >> 5) * 4) * 4; }},
        {"8x8 patch, pitch 10*36 = 360 (today), rows 0..7 only x4", [](int l) { int r = l & 7; return (r * 36 + (l >> 5) * 4) * 4; }},
    };
    std::vector<int> pitches = {288, 296, 304, 312, 320, 328, 336, 344, 352, 360, 368, 376, 384, 392, 400, 416, 432, 448};
    for (int pch : pitches)
        ps.push_back({"8x8 patch, stride 36, patch row pitch " + std::to_string(pch) + " floats",
                      [pch](int l) { int r = l & 31; return ((r / 8) * pch + (r % 8) * 36 + (l >> 5) * 4) * 4; }});
    std::vector<int> pitches16 = {320, 336, 352, 360, 368, 384, 400, 416, 432, 448};
    for (int pch : pitches16)
        ps.push_back({"8x16 patch, stride 20, patch row pitch " + std::to_string(pch) + " floats",
                      [pch](int l) { int r = l & 31; return ((r / 16) * pch + (r % 16) * 20 + (l >> 5) * 4) * 4; }});
    for (int w = 0; w < 2; ++w) {
        printf("---- %s: cycles per instruction (64 lanes x 16 B)\n", w ? "ds_write_b128" : "ds_read_b128");
        for (auto& p : ps) {
            int h[64];
            bool ok = true;
            for (int l = 0; l < 64; ++l) { h[l] = p.f(l); ok = ok && h[l] + 16 <= 65536 && h[l] % 16 == 0; }
            if (!ok) { printf("%-70s skipped\n", p.name.c_str()); continue; }
            hipMemcpy(d_addr, h, sizeof(h), hipMemcpyHostToDevice);
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 65536, 0, d_addr, d_out, d_sink, w);
            long long c = 0;
            hipMemcpy(&c, d_out, 8, hipMemcpyDeviceToHost);
            printf("%-70s %7.2f\n", p.name.c_str(), (double)c / 64.0);
        }
    }
    return 0;
}
