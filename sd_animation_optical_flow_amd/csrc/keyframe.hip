// Key-frame detector (SURVEY section 8 "next" row f4): the step in front of the flow path,
// ofgen_keyframe_inpaint.py:143-192,327-370 -- dilate(Canny(V channel, median-derived thresholds), k x k ones) per
// frame and the mean absolute difference of two edge maps.  The reference does this with OpenCV on the host; here
// the decoded frames stay on the device.  All arithmetic is integer (OpenCV's canny.cpp restated: 3x3 Sobel with
// replicated borders, L1 magnitude, the 15-bit fixed-point tan(22.5) direction test, hysteresis); the thresholds
// come from the exact median through a 256-bin histogram.  PARITY UNPINNED (no OpenCV in the reference tree or in
// this image): the kernels are held bit-exactly to oracle/keyframe_oracle.py.
#include "ofx_internal.h"

namespace {

constexpr int kTG22 = 13573;   // (int)(0.4142135623730950488016887242097 * (1 << 15) + 0.5)

// V = max(B, G, R) + per-image histogram
__global__ __launch_bounds__(256) void kf_value_hist_kernel(const uint8_t* __restrict__ bgr, uint8_t* __restrict__ lum,
                                                            unsigned* __restrict__ hist, long HW) {
    __shared__ unsigned h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const long b = blockIdx.y;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < HW; i += (long)gridDim.x * 256) {
        const uint8_t* p = bgr + (b * HW + i) * 3;
        const unsigned v = max(max((unsigned)p[0], (unsigned)p[1]), (unsigned)p[2]);
        lum[b * HW + i] = (uint8_t)v;
        atomicAdd(&h[v], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[b * 256 + threadIdx.x], h[threadIdx.x]);
}

// np.median -> low = int(max(0, (1 - sigma) * median)), high = int(min(255, (1 + sigma) * median)), sigma = 1/3 (f64)
__global__ void kf_thresholds_kernel(const unsigned* __restrict__ hist, int* __restrict__ thr, long HW) {
    const int b = blockIdx.x;
    if (threadIdx.x != 0) return;
    const unsigned* h = hist + (long)b * 256;
    const long k_hi = HW / 2, k_lo = (HW & 1) ? k_hi : k_hi - 1;   // 0-based ranks of the middle element(s)
    long cum = 0;
    int v_lo = -1, v_hi = -1;
    for (int v = 0; v < 256; ++v) {
        cum += h[v];
        if (v_lo < 0 && cum > k_lo) v_lo = v;
        if (v_hi < 0 && cum > k_hi) { v_hi = v; break; }
    }
    const double median = ((double)v_lo + (double)v_hi) / 2.0;
    const double sigma = 1.0 / 3.0;
    const double lo = (1.0 - sigma) * median, hi = (1.0 + sigma) * median;
    int low = (int)(lo < 0.0 ? 0.0 : lo), high = (int)(hi > 255.0 ? 255.0 : hi);
    if (low > high) { const int t = low; low = high; high = t; }   // cv::Canny swaps
    thr[2 * b] = low;
    thr[2 * b + 1] = high;
}

// Sobel + L1 magnitude into an LDS tile with a one-pixel apron (zero outside the image), then non-maximum
// suppression: map = 2 strong, 0 candidate, 1 none.
constexpr int kCW = 32, kCH = 8;
__global__ __launch_bounds__(256) void kf_canny_map_kernel(const uint8_t* __restrict__ lum, const int* __restrict__ thr,
                                                           uint8_t* __restrict__ map, int H, int W) {
    __shared__ int smag[kCH + 2][kCW + 2];
    __shared__ short sdx[kCH][kCW], sdy[kCH][kCW];
    const long b = blockIdx.z;
    const uint8_t* L = lum + b * (long)H * W;
    const int x0 = blockIdx.x * kCW, y0 = blockIdx.y * kCH;
    for (int i = threadIdx.x; i < (kCH + 2) * (kCW + 2); i += 256) {
        const int ly = i / (kCW + 2), lx = i - ly * (kCW + 2);
        const int y = y0 + ly - 1, x = x0 + lx - 1;
        int m = 0;
        if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
            const int ym = max(y - 1, 0), yp = min(y + 1, H - 1), xm = max(x - 1, 0), xp = min(x + 1, W - 1);   // BORDER_REPLICATE
            const int a00 = L[(long)ym * W + xm], a01 = L[(long)ym * W + x], a02 = L[(long)ym * W + xp];
            const int a10 = L[(long)y * W + xm], a12 = L[(long)y * W + xp];
            const int a20 = L[(long)yp * W + xm], a21 = L[(long)yp * W + x], a22 = L[(long)yp * W + xp];
            const int dx = (a02 + 2 * a12 + a22) - (a00 + 2 * a10 + a20);
            const int dy = (a20 + 2 * a21 + a22) - (a00 + 2 * a01 + a02);
            m = abs(dx) + abs(dy);
            if (ly >= 1 && ly <= kCH && lx >= 1 && lx <= kCW) {
                sdx[ly - 1][lx - 1] = (short)dx;
                sdy[ly - 1][lx - 1] = (short)dy;
            }
        }
        smag[ly][lx] = m;
    }
    __syncthreads();
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int x = x0 + tx, y = y0 + ty;
    if (x >= W || y >= H) return;
    const int low = thr[2 * b], high = thr[2 * b + 1];
    const int m = smag[ty + 1][tx + 1];
    uint8_t out = 1;
    if (m > low) {
        const int xs = sdx[ty][tx], ys = sdy[ty][tx];
        const int ax = abs(xs), ay = abs(ys) << 15;
        const int tg22x = ax * kTG22;
        bool is_max;
        if (ay < tg22x) {
            is_max = m > smag[ty + 1][tx] && m >= smag[ty + 1][tx + 2];
        } else {
            const int tg67x = tg22x + (ax << 16);
            if (ay > tg67x) {
                is_max = m > smag[ty][tx + 1] && m >= smag[ty + 2][tx + 1];
            } else {
                const int s = (xs ^ ys) < 0 ? -1 : 1;
                is_max = m > smag[ty][tx + 1 - s] && m > smag[ty + 2][tx + 1 + s];
            }
        }
        if (is_max) out = m > high ? 2 : 0;
    }
    map[(b * H + y) * (long)W + x] = out;
}

// One hysteresis sweep: inside a 32x32 tile (with apron) candidates touching a strong pixel become strong until the
// tile is stable; `changed` is raised when anything moved.  Sweeps are enqueued in batches: a sweep first looks at the flag
// its predecessor left (`prev`, null for the first) and retires at once when that one changed nothing, so the host checks
// for convergence once per batch instead of once per sweep.
constexpr int kHT = 32;
constexpr int kSweepBatch = 8;
__global__ __launch_bounds__(256) void kf_hysteresis_kernel(uint8_t* __restrict__ map, int H, int W, const int* __restrict__ prev,
                                                            int* __restrict__ changed) {
    __shared__ uint8_t t[kHT + 2][kHT + 2];
    __shared__ int moved, any;
    if (prev != nullptr && __hip_atomic_load(prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;   // already converged
    const long b = blockIdx.z;
    uint8_t* M = map + b * (long)H * W;
    const int x0 = blockIdx.x * kHT, y0 = blockIdx.y * kHT;
    for (int i = threadIdx.x; i < (kHT + 2) * (kHT + 2); i += 256) {
        const int ly = i / (kHT + 2), lx = i - ly * (kHT + 2);
        const int y = y0 + ly - 1, x = x0 + lx - 1;
        t[ly][lx] = ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) ? M[(long)y * W + x] : (uint8_t)1;
    }
    if (threadIdx.x == 0) any = 0;
    __syncthreads();
    for (;;) {
        if (threadIdx.x == 0) moved = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < kHT * kHT; i += 256) {
            const int ly = i / kHT + 1, lx = i % kHT + 1;
            if (t[ly][lx] == 0) {
                const bool strong = t[ly - 1][lx - 1] == 2 || t[ly - 1][lx] == 2 || t[ly - 1][lx + 1] == 2 || t[ly][lx - 1] == 2 ||
                                    t[ly][lx + 1] == 2 || t[ly + 1][lx - 1] == 2 || t[ly + 1][lx] == 2 || t[ly + 1][lx + 1] == 2;
                if (strong) {
                    t[ly][lx] = 2;       // racy reads of neighbours only ever see 0 -> 2 transitions: monotone, converges
                    moved = 1;
                }
            }
        }
        __syncthreads();
        if (!moved) break;
        if (threadIdx.x == 0) any = 1;
        __syncthreads();
    }
    if (any) {
        for (int i = threadIdx.x; i < kHT * kHT; i += 256) {
            const int ly = i / kHT, lx = i % kHT;
            const int y = y0 + ly, x = x0 + lx;
            if (y < H && x < W) M[(long)y * W + x] = t[ly + 1][lx + 1];
        }
        if (threadIdx.x == 0) atomicOr(changed, 1);
    }
}

// dilate(255 * (map == 2), ones(k, k)): separable OR, outside pixels ignored
__global__ __launch_bounds__(256) void kf_dilate_rows_kernel(const uint8_t* __restrict__ map, uint8_t* __restrict__ tmp, int H, int W,
                                                             int r, long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int x = (int)(i % W);
        const long row = i - x;
        bool on = false;
        for (int d = max(x - r, 0); d <= min(x + r, W - 1); ++d) on |= map[row + d] == 2;
        tmp[i] = on ? 255 : 0;
    }
}
__global__ __launch_bounds__(256) void kf_dilate_cols_kernel(const uint8_t* __restrict__ tmp, uint8_t* __restrict__ out, int H, int W,
                                                             int r, long total) {
    const long HW = (long)H * W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long b = i / HW;
        const long p = i - b * HW;
        const int y = (int)(p / W), x = (int)(p - (long)y * W);
        uint8_t v = 0;
        for (int d = max(y - r, 0); d <= min(y + r, H - 1); ++d) v |= tmp[b * HW + (long)d * W + x];
        out[i] = v;
    }
}

__global__ __launch_bounds__(256) void kf_absdiff_sum_kernel(const uint8_t* __restrict__ a, long abs_, const uint8_t* __restrict__ bb,
                                                             long bbs, unsigned long long* __restrict__ sums, long n) {
    const long img = blockIdx.y;
    const uint8_t* pa = a + img * abs_;
    const uint8_t* pb = bb + img * bbs;
    unsigned long long acc = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int d = (int)pa[i] - (int)pb[i];
        acc += (unsigned)(d < 0 ? -d : d);
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    __shared__ unsigned long long part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&sums[img], part[0] + part[1] + part[2] + part[3]);
}

struct KfScratch {
    uint8_t *lum, *map, *tmp;
    unsigned* hist;
    int* thr;
    int* flag;
    size_t bytes;
};
KfScratch kf_carve(void* base, int B, int H, int W) {
    KfScratch s{};
    const size_t n = ((size_t)B * H * W + 255) / 256 * 256;
    char* p = (char*)base;
    size_t off = 0;
    s.lum = (uint8_t*)(p + off); off += n;
    s.map = (uint8_t*)(p + off); off += n;
    s.tmp = (uint8_t*)(p + off); off += n;
    s.hist = (unsigned*)(p + off); off += (size_t)B * 256 * sizeof(unsigned);
    s.thr = (int*)(p + off); off += ((size_t)B * 2 * sizeof(int) + 255) / 256 * 256;
    s.flag = (int*)(p + off); off += 256;
    s.bytes = off;
    return s;
}

}  // namespace

extern "C" {

size_t ofx_detect_edges_scratch_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return kf_carve(nullptr, B, H, W).bytes;
}

int ofx_detect_edges(const uint8_t* frames_bgr, uint8_t* edges, void* scratch, size_t scratch_bytes, int B, int H, int W, int ksize,
                     void* stream) {
    OFX_REQUIRE(frames_bgr && edges && scratch && B > 0 && H > 0 && W > 0, OFX_EINVAL);
    OFX_REQUIRE(ksize >= 1 && (ksize & 1) && B <= 65535, OFX_EINVAL);
    OFX_REQUIRE((((uintptr_t)scratch) & 255u) == 0, OFX_EALIGN);
    KfScratch s = kf_carve(scratch, B, H, W);
    OFX_REQUIRE(s.bytes <= scratch_bytes, OFX_ENOMEM);
    hipStream_t st = (hipStream_t)stream;
    const long HW = (long)H * W, total = (long)B * HW;
    OFX_HIP_CHECK(hipMemsetAsync(s.hist, 0, (size_t)B * 256 * sizeof(unsigned), st));
    {
        OfxProfScope prof("keyframe_edges", st);
        const int gx = (int)std::min<long>((HW + 255) / 256, 1024);
        hipLaunchKernelGGL(kf_value_hist_kernel, dim3(gx, B), dim3(256), 0, st, frames_bgr, s.lum, s.hist, HW);
        hipLaunchKernelGGL(kf_thresholds_kernel, dim3(B), dim3(64), 0, st, s.hist, s.thr, HW);
        hipLaunchKernelGGL(kf_canny_map_kernel, dim3(ofx_cdiv(W, kCW), ofx_cdiv(H, kCH), B), dim3(256), 0, st, s.lum, s.thr, s.map, H, W);
    }
    // hysteresis: sweep until nothing changes.  kSweepBatch sweeps per host check (a 4-byte copy + one stream
    // synchronisation per batch); flags[k] = "sweep k of the batch moved something", a converged sweep leaves 0 and
    // every later sweep of the batch retires at once
    for (int it = 0; it < 4 * (H + W); it += kSweepBatch) {
        OFX_HIP_CHECK(hipMemsetAsync(s.flag, 0, kSweepBatch * sizeof(int), st));
        for (int k = 0; k < kSweepBatch; ++k)
            hipLaunchKernelGGL(kf_hysteresis_kernel, dim3(ofx_cdiv(W, kHT), ofx_cdiv(H, kHT), B), dim3(256), 0, st, s.map, H, W,
                               k == 0 ? (const int*)nullptr : (const int*)(s.flag + k - 1), s.flag + k);
        int host_flag = 0;
        OFX_HIP_CHECK(hipMemcpyAsync(&host_flag, s.flag + kSweepBatch - 1, sizeof(int), hipMemcpyDeviceToHost, st));
        OFX_HIP_CHECK(hipStreamSynchronize(st));
        if (!host_flag) break;
    }
    const int r = ksize / 2;
    const int g = (int)std::min<long>((total + 255) / 256, 256L * 32);
    hipLaunchKernelGGL(kf_dilate_rows_kernel, dim3(g), dim3(256), 0, st, s.map, s.tmp, H, W, r, total);
    hipLaunchKernelGGL(kf_dilate_cols_kernel, dim3(g), dim3(256), 0, st, s.tmp, edges, H, W, r, total);
    return ofx_launch_status();
}

int ofx_abs_diff_sum_u8(const uint8_t* a, long a_bstride, const uint8_t* b, long b_bstride, unsigned long long* sums, int B, long n,
                        void* stream) {
    OFX_REQUIRE(a && b && sums && B > 0 && B <= 65535 && n > 0, OFX_EINVAL);
    hipStream_t st = (hipStream_t)stream;
    OFX_HIP_CHECK(hipMemsetAsync(sums, 0, (size_t)B * sizeof(unsigned long long), st));
    OfxProfScope prof("abs_diff_sum_u8", st);
    const int gx = (int)std::min<long>((n + 255) / 256, 512);
    hipLaunchKernelGGL(kf_absdiff_sum_kernel, dim3(gx, B), dim3(256), 0, st, a, a_bstride, b, b_bstride, sums, n);
    return ofx_launch_status();
}

}  // extern "C"
