// Backward warp, inpaint-mask morphology and the small per-pixel compositing steps around them.
//
// Reference behaviour being reproduced (file:line in the reference repo):
//   warp_frame              pdcnet_of.py:34-42 ; ofgen_keyframe_inpaint.py:92-98   (cv2.remap cubic)
//   generate_mask           ofgen_keyframe_inpaint.py:317-322      (threshold + 7x7 ellipse dilate)
//   confidence_to_mask      ofgen_keyframe_inpaint.py:237-248
//   expand_mask             ofgen_keyframe_inpaint.py:968-973
//   of_calc distance map    ofgen_keyframe_inpaint.py:118-126
//   merge_images / mix      ofgen_keyframe_inpaint.py:676-681, :306-315
//
// All of these are HBM-bound byte/float streaming kernels (warp: 5.5 MB, mask: 2 MB per 512x768
// frame): one pass over the data, coalesced row-major access, the dilation window served from an LDS
// tile.  Integer paths (cv2 fixed-point cubic, masks, merges) are bit-exact by construction.
#include "ofx_internal.h"

#include <cmath>
#include <mutex>
#include <vector>

// Bit-exact paths (cv2 float cubic, distance map, frame mix) need every multiply and add rounded
// separately, like the numpy/OpenCV reference: HIP's __fmul_rn/__fadd_rn are plain operators that the
// default -ffp-contract=fast would fuse into FMAs.  These kernels are HBM-bound; no cost.
#pragma clang fp contract(off)

namespace {

// ------------------------------------------------------------------------------------------
// OpenCV remap tables for INTER_CUBIC (imgwarp.cpp initInterTab2D): 32x32 fractions x 16 taps
// ------------------------------------------------------------------------------------------
constexpr int kTabSize = 32;
constexpr int kCoefBits = 15;
constexpr int kCoefScale = 1 << kCoefBits;

struct Cv2Tables {
    float* f = nullptr;   // [1024][16]
    short* i = nullptr;   // [1024][16]
    int status = 0;
};
// One table set per HIP device: the tables live in that device's memory, so a process that moves its algorithm
// to another GPU (`PDCNetPlus.to(device)`) must not hand device 0's pointers to a kernel running on device 1.
constexpr int kMaxDevices = 64;
Cv2Tables g_tabs_dev[kMaxDevices];
bool g_tabs_built[kMaxDevices] = {};
std::mutex g_tabs_mu;

void host_cubic(float x, float* c) {
    const float A = -0.75f;
    volatile float t1 = x + 1.0f;   // volatile: keep separate roundings (no contraction)
    volatile float a0 = A * t1;
    volatile float a1 = a0 - 5.0f * A;
    volatile float a2 = a1 * t1;
    volatile float a3 = a2 + 8.0f * A;
    volatile float a4 = a3 * t1;
    c[0] = a4 - 4.0f * A;
    volatile float b0 = (A + 2.0f) * x;
    volatile float b1 = b0 - (A + 3.0f);
    volatile float b2 = b1 * x;
    volatile float b3 = b2 * x;
    c[1] = b3 + 1.0f;
    volatile float u = 1.0f - x;
    volatile float d0 = (A + 2.0f) * u;
    volatile float d1 = d0 - (A + 3.0f);
    volatile float d2 = d1 * u;
    volatile float d3 = d2 * u;
    c[2] = d3 + 1.0f;
    volatile float e0 = 1.0f - c[0];
    volatile float e1 = e0 - c[1];
    c[3] = e1 - c[2];
}

void build_tables(Cv2Tables& tabs) {
    std::vector<float> c1(kTabSize * 4);
    for (int i = 0; i < kTabSize; ++i) host_cubic((float)i * (1.0f / kTabSize), &c1[i * 4]);
    std::vector<float> tf(kTabSize * kTabSize * 16);
    std::vector<short> ti(kTabSize * kTabSize * 16);
    for (int i = 0; i < kTabSize; ++i)
        for (int j = 0; j < kTabSize; ++j) {
            float* f = &tf[(i * kTabSize + j) * 16];
            short* it = &ti[(i * kTabSize + j) * 16];
            int isum = 0;
            for (int k1 = 0; k1 < 4; ++k1)
                for (int k2 = 0; k2 < 4; ++k2) {
                    volatile float v = c1[i * 4 + k1] * c1[j * 4 + k2];
                    f[k1 * 4 + k2] = v;
                    volatile float sv = v * (float)kCoefScale;
                    long r = lrintf(sv);   // round-half-even, = cv::saturate_cast<short>(float)
                    if (r > 32767) r = 32767;
                    if (r < -32768) r = -32768;
                    it[k1 * 4 + k2] = (short)r;
                    isum += (int)r;
                }
            if (isum != kCoefScale) {
                const int diff = isum - kCoefScale;
                int mk = 2 * 4 + 2, Mk = 2 * 4 + 2;
                for (int k1 = 2; k1 < 4; ++k1)
                    for (int k2 = 2; k2 < 4; ++k2) {
                        const int k = k1 * 4 + k2;
                        if (it[k] < it[mk]) mk = k;
                        else if (it[k] > it[Mk]) Mk = k;
                    }
                if (diff < 0) it[Mk] = (short)(it[Mk] - diff);
                else it[mk] = (short)(it[mk] - diff);
            }
        }
    hipError_t e = hipMalloc(&tabs.f, tf.size() * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&tabs.i, ti.size() * sizeof(short));
    if (e == hipSuccess) e = hipMemcpy(tabs.f, tf.data(), tf.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(tabs.i, ti.data(), ti.size() * sizeof(short), hipMemcpyHostToDevice);
    tabs.status = (int)e;
}

// Tables of the CURRENT device (the one the caller's stream and pointers belong to), built on first use.
int ensure_tables(const Cv2Tables** out) {
    int dev = 0;
    OFX_HIP_CHECK(hipGetDevice(&dev));
    OFX_REQUIRE(dev >= 0 && dev < kMaxDevices, OFX_ENODEV);
    std::lock_guard<std::mutex> lk(g_tabs_mu);
    if (!g_tabs_built[dev]) {
        build_tables(g_tabs_dev[dev]);
        g_tabs_built[dev] = g_tabs_dev[dev].status == 0;
    }
    *out = &g_tabs_dev[dev];
    return g_tabs_dev[dev].status;
}

// ------------------------------------------------------------------------------------------
// warp
// ------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ float ld_px(const T* p) { return (float)(*p); }

template <typename T>
__device__ __forceinline__ T st_px(float v);
template <>
__device__ __forceinline__ float st_px<float>(float v) { return v; }
template <>
__device__ __forceinline__ uint8_t st_px<uint8_t>(float v) {
    return (uint8_t)fminf(fmaxf(rintf(v), 0.f), 255.f);
}

// sample position, reproducing (X + disp).astype(float32) with X from an f64 linspace
__device__ __forceinline__ float map_coord(int x, float f, float sign) {
    return (float)((double)x + (double)sign * (double)f);
}

template <typename T, int C, int MODE>
__device__ __forceinline__ void warp_pixel(const T* __restrict__ src, int H, int W, float mx, float my,
                                            const short* __restrict__ tabi, const float* __restrict__ tabf,
                                            T* __restrict__ dst) {
    if (MODE == OFX_WARP_BILINEAR) {
        const float x0f = floorf(mx), y0f = floorf(my);
        const float fx = mx - x0f, fy = my - y0f;
        // guard the int conversion against absurd flows (everything is out of range then)
        const int x0 = (int)fminf(fmaxf(x0f, -1.0e6f), 1.0e6f);
        const int y0 = (int)fminf(fmaxf(y0f, -1.0e6f), 1.0e6f);
        const float w00 = (1.f - fx) * (1.f - fy), w01 = fx * (1.f - fy);
        const float w10 = (1.f - fx) * fy, w11 = fx * fy;
        const bool xa = (unsigned)x0 < (unsigned)W, xb = (unsigned)(x0 + 1) < (unsigned)W;
        const bool ya = (unsigned)y0 < (unsigned)H, yb = (unsigned)(y0 + 1) < (unsigned)H;
        const T* p00 = src + ((long)y0 * W + x0) * C;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float v00 = (xa && ya) ? ld_px(p00 + c) : 0.f;
            const float v01 = (xb && ya) ? ld_px(p00 + C + c) : 0.f;
            const float v10 = (xa && yb) ? ld_px(p00 + (long)W * C + c) : 0.f;
            const float v11 = (xb && yb) ? ld_px(p00 + (long)W * C + C + c) : 0.f;
            float acc = v00 * w00;
            acc = acc + v01 * w01;
            acc = acc + v10 * w10;
            acc = acc + v11 * w11;
            dst[c] = st_px<T>(acc);
        }
    } else if (MODE == OFX_WARP_BICUBIC) {
        const float x0f = floorf(mx), y0f = floorf(my);
        float wx[4], wy[4];
        ofx_cubic_coeffs(mx - x0f, wx);
        ofx_cubic_coeffs(my - y0f, wy);
        const int x0 = (int)fminf(fmaxf(x0f, -1.0e6f), 1.0e6f) - 1;
        const int y0 = (int)fminf(fmaxf(y0f, -1.0e6f), 1.0e6f) - 1;
        float acc[C];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = 0.f;
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1) {
            const int yy = y0 + k1;
            const bool yok = (unsigned)yy < (unsigned)H;
            float row[C];
#pragma unroll
            for (int c = 0; c < C; ++c) row[c] = 0.f;
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) {
                const int xx = x0 + k2;
                const bool ok = yok && (unsigned)xx < (unsigned)W;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float v = ok ? ld_px(src + ((long)yy * W + xx) * C + c) : 0.f;
                    row[c] = row[c] + v * wx[k2];
                }
            }
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = acc[c] + row[c] * wy[k1];
        }
#pragma unroll
        for (int c = 0; c < C; ++c) dst[c] = st_px<T>(acc[c]);
    } else {   // OFX_WARP_CV2_CUBIC
        const float lim = 1.0e9f;
        const int sx = (int)rintf(fminf(fmaxf(mx * (float)kTabSize, -lim), lim));
        const int sy = (int)rintf(fminf(fmaxf(my * (float)kTabSize, -lim), lim));
        int ix = sx >> 5, iy = sy >> 5;
        ix = min(max(ix, -32768), 32767);
        iy = min(max(iy, -32768), 32767);
        const int t = (sy & (kTabSize - 1)) * kTabSize + (sx & (kTabSize - 1));
        const int x0 = ix - 1, y0 = iy - 1;
        if (sizeof(T) == 1) {
            const short* w = tabi + t * 16;
            int acc[C];
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = 0;
#pragma unroll
            for (int k1 = 0; k1 < 4; ++k1) {
                const int yy = y0 + k1;
                const bool yok = (unsigned)yy < (unsigned)H;
#pragma unroll
                for (int k2 = 0; k2 < 4; ++k2) {
                    const int xx = x0 + k2;
                    if (yok && (unsigned)xx < (unsigned)W) {
                        const int wv = w[k1 * 4 + k2];
#pragma unroll
                        for (int c = 0; c < C; ++c) acc[c] += (int)src[((long)yy * W + xx) * C + c] * wv;
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int v = (acc[c] + (1 << (kCoefBits - 1))) >> kCoefBits;
                dst[c] = (T)min(max(v, 0), 255);
            }
        } else {
            const float* w = tabf + t * 16;
            float acc[C];
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = 0.f;
#pragma unroll
            for (int k1 = 0; k1 < 4; ++k1) {
                const int yy = y0 + k1;
                const bool yok = (unsigned)yy < (unsigned)H;
#pragma unroll
                for (int k2 = 0; k2 < 4; ++k2) {
                    const int xx = x0 + k2;
                    const bool ok = yok && (unsigned)xx < (unsigned)W;
                    const float wv = w[k1 * 4 + k2];
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        const float v = ok ? ld_px(src + ((long)yy * W + xx) * C + c) : 0.f;
                        acc[c] = __fadd_rn(acc[c], __fmul_rn(v, wv));   // no FMA: matches the oracle bit for bit
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < C; ++c) dst[c] = (T)acc[c];
        }
    }
}

template <typename T, int C, int MODE>
__global__ __launch_bounds__(256) void warp_kernel(const T* __restrict__ frame, long fbs,
                                                   const float* __restrict__ flow, T* __restrict__ out,
                                                   int H, int W, long total, float sign,
                                                   const short* __restrict__ tabi,
                                                   const float* __restrict__ tabf) {
    const long hw = (long)H * W;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long b = idx / hw;
        const int rem = (int)(idx - b * hw);
        const int y = rem / W;
        const int x = rem - y * W;
        const float2 f = reinterpret_cast<const float2*>(flow)[idx];
        const float mx = map_coord(x, f.x, sign);
        const float my = map_coord(y, f.y, sign);
        T px[C];
        warp_pixel<T, C, MODE>(frame + b * fbs, H, W, mx, my, tabi, tabf, px);
#pragma unroll
        for (int c = 0; c < C; ++c) out[idx * C + c] = px[c];
    }
}

// ------------------------------------------------------------------------------------------
// fast path: uint8, 3 channels (the AI key frame), 4 consecutive output pixels per thread.
// The generic kernel issues 12 byte-loads + 3 byte-stores per pixel; here a row of taps is ONE unaligned
// 8- or 12-byte load (the taps of a row are adjacent in memory) and the 4 results are one 12-byte store,
// so the kernel is limited by HBM bytes instead of load/store instruction issue.
// ------------------------------------------------------------------------------------------
struct __attribute__((packed, aligned(1))) PackedU2 { uint32_t a, b; };
struct __attribute__((packed, aligned(1))) PackedU3 { uint32_t a, b, c; };

// N adjacent pixels (3 bytes each) of row yy starting at column xx; zeros outside the image
template <int N>
__device__ __forceinline__ void fetch_run_u8c3(const uint8_t* __restrict__ img, long img_bytes, int H, int W, int yy, int xx,
                                               uint8_t (&px)[N][3]) {
    static_assert(N == 2 || N == 4, "runs of 2 (bilinear) or 4 (cubic) pixels");
#pragma unroll
    for (int i = 0; i < N; ++i) px[i][0] = px[i][1] = px[i][2] = 0;
    if ((unsigned)yy >= (unsigned)H) return;
    const long off = ((long)yy * W + xx) * 3;
    constexpr int kLoadBytes = N == 2 ? 8 : 12;
    if (xx >= 0 && xx + N <= W && off + kLoadBytes <= img_bytes) {
        uint32_t w[3];
        if (N == 2) {
            const PackedU2 v = *reinterpret_cast<const PackedU2*>(img + off);
            w[0] = v.a; w[1] = v.b; w[2] = 0;
        } else {
            const PackedU3 v = *reinterpret_cast<const PackedU3*>(img + off);
            w[0] = v.a; w[1] = v.b; w[2] = v.c;
        }
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int byte = i * 3 + c;
                px[i][c] = (uint8_t)((w[byte >> 2] >> ((byte & 3) * 8)) & 0xFFu);
            }
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int x = xx + i;
            if ((unsigned)x < (unsigned)W) {
                const uint8_t* q = img + ((long)yy * W + x) * 3;
                px[i][0] = q[0]; px[i][1] = q[1]; px[i][2] = q[2];
            }
        }
    }
}

template <int MODE>
__device__ __forceinline__ void warp_pixel_u8c3(const uint8_t* __restrict__ img, long img_bytes, int H, int W, float mx, float my,
                                                const short* __restrict__ tabi, uint8_t (&dst)[3]) {
    if (MODE == OFX_WARP_BILINEAR) {
        const float x0f = floorf(mx), y0f = floorf(my);
        const float fx = mx - x0f, fy = my - y0f;
        const int x0 = (int)fminf(fmaxf(x0f, -1.0e6f), 1.0e6f);
        const int y0 = (int)fminf(fmaxf(y0f, -1.0e6f), 1.0e6f);
        const float w00 = (1.f - fx) * (1.f - fy), w01 = fx * (1.f - fy);
        const float w10 = (1.f - fx) * fy, w11 = fx * fy;
        // Branch-free taps: ONE unconditional 8-byte load per row from a position clamped into the frame
        // (so all 8 row loads of a thread's 4 pixels are in flight together), then the two pixels are
        // extracted with shifts and zeroed where the true tap lies outside the image.
        uint8_t r0[2][3], r1[2][3];
        const int xc = min(max(x0, 0), W - 2);
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int yy = y0 + rr;
            const bool yok = (unsigned)yy < (unsigned)H;
            const long off = ((long)min(max(yy, 0), H - 1) * W + xc) * 3;
            const long start = off + 8 <= img_bytes ? off : img_bytes - 8;      // last pixel pair of the frame
            const PackedU2 v = *reinterpret_cast<const PackedU2*>(img + start);
            const unsigned long long bits = ((((unsigned long long)v.b) << 32) | v.a) >> ((int)(off - start) * 8);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int col = x0 + i;
                const bool ok = yok && (unsigned)col < (unsigned)W;
                const int sh = ok ? 24 * (col - xc) : 0;                         // col - xc is 0 or 1 when ok
                const unsigned px3 = ok ? (unsigned)(bits >> sh) & 0xFFFFFFu : 0u;
#pragma unroll
                for (int c = 0; c < 3; ++c) (rr == 0 ? r0 : r1)[i][c] = (uint8_t)((px3 >> (8 * c)) & 0xFFu);
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float acc = (float)r0[0][c] * w00;       // same evaluation order as the generic kernel
            acc = acc + (float)r0[1][c] * w01;
            acc = acc + (float)r1[0][c] * w10;
            acc = acc + (float)r1[1][c] * w11;
            dst[c] = st_px<uint8_t>(acc);
        }
    } else {   // OFX_WARP_CV2_CUBIC
        const float lim = 1.0e9f;
        const int sx = (int)rintf(fminf(fmaxf(mx * (float)kTabSize, -lim), lim));
        const int sy = (int)rintf(fminf(fmaxf(my * (float)kTabSize, -lim), lim));
        int ix = sx >> 5, iy = sy >> 5;
        ix = min(max(ix, -32768), 32767);
        iy = min(max(iy, -32768), 32767);
        const short* w = tabi + ((sy & (kTabSize - 1)) * kTabSize + (sx & (kTabSize - 1))) * 16;
        int acc[3] = {0, 0, 0};
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1) {
            uint8_t row[4][3];
            fetch_run_u8c3<4>(img, img_bytes, H, W, iy - 1 + k1, ix - 1, row);
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) {
                const int wv = w[k1 * 4 + k2];
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[c] += (int)row[k2][c] * wv;
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int v = (acc[c] + (1 << (kCoefBits - 1))) >> kCoefBits;
            dst[c] = (uint8_t)min(max(v, 0), 255);
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void warp_u8c3_x4_kernel(const uint8_t* __restrict__ frame, long fbs, const float* __restrict__ flow,
                                                           uint8_t* __restrict__ out, int H, int W, long ngroups, float sign,
                                                           const short* __restrict__ tabi) {
    const int W4 = (W + 3) >> 2;
    const long img_bytes = (long)H * W * 3;
    for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += (long)gridDim.x * blockDim.x) {
        const int xg = (int)(g % W4);
        const long row = g / W4;            // b*H + y
        const int y = (int)(row % H);
        const long b = row / H;
        const int x = xg * 4;
        const int npx = min(4, W - x);
        const long p0 = row * W + x;
        const uint8_t* img = frame + b * fbs;
        uint8_t res[4][3];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < npx) {
                const float2 f = reinterpret_cast<const float2*>(flow)[p0 + j];
                warp_pixel_u8c3<MODE>(img, img_bytes, H, W, map_coord(x + j, f.x, sign), map_coord(y, f.y, sign), tabi, res[j]);
            } else {
                res[j][0] = res[j][1] = res[j][2] = 0;
            }
        }
        uint8_t* o = out + p0 * 3;
        if (npx == 4 && (((uintptr_t)o) & 3u) == 0) {
            uint32_t w[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                uint32_t v = 0;
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    const int byte = k * 4 + bb;
                    v |= (uint32_t)res[byte / 3][byte % 3] << (bb * 8);
                }
                w[k] = v;
            }
            PackedU3 st{w[0], w[1], w[2]};
            *reinterpret_cast<PackedU3*>(o) = st;
        } else {
            for (int j = 0; j < npx; ++j) {
                o[j * 3 + 0] = res[j][0]; o[j * 3 + 1] = res[j][1]; o[j * 3 + 2] = res[j][2];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// bilinear fast path (the bench / FrameSynthesizer default): uint8 x 3 channels, W % 4 == 0, everything
// addressable with 32 bits.  Same arithmetic as warp_pixel (weights, products and sums in the same order,
// contraction off) with the instruction count roughly halved -- the x4 kernel above spends ~200 VALU
// instructions per pixel and is VALU-bound, not HBM-bound:
//   * 32-bit index math, the (row, x-group) split by magic-number division;
//   * out-of-image taps get a ZERO WEIGHT instead of a zeroed pixel (u8 * 0 = +0 either way), so the
//     loaded bytes never need masking;
//   * a row of two taps is one unaligned 8-byte load whose bytes are consumed directly by
//     v_cvt_f32_ubyteN; channel pairs are blended with packed fp32 math.
// ------------------------------------------------------------------------------------------
struct BilinArgs {
    const uint8_t* frame;
    const float* flow;
    uint8_t* out;
    int H, W, W4;
    unsigned fbs;            // bytes between key frames (0 = shared)
    unsigned img_bytes;
    unsigned ngroups;
    unsigned magic_w4, magic_h;   // ceil(2^32 / d); exact for the operand ranges checked by the launcher
    float sign;
};

typedef float ofx_v2f __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void warp_bilinear_u8c3_kernel(const BilinArgs a) {
    // one group of 4 pixels per thread (the launcher sizes the grid to cover all groups): the kernel is bound by
    // the two dependent memory latencies of a group (flow, then taps) times the generations of resident
    // wavefronts, so more independent threads beat longer per-thread loops (a software-pipelined 4-groups-per-
    // thread variant measured 23 % slower)
    for (unsigned g = blockIdx.x * 256u + threadIdx.x; g < a.ngroups; g += gridDim.x * 256u) {
        const float4 fa = reinterpret_cast<const float4*>(a.flow)[2 * (size_t)g];       // output pixel 4g -> float4 2g, 2g+1
        const float4 fb = reinterpret_cast<const float4*>(a.flow)[2 * (size_t)g + 1];
        const unsigned row = a.W4 == 1 ? g : __umulhi(g, a.magic_w4);      // b*H + y
        const int x = (int)(g - row * (unsigned)a.W4) * 4;
        const unsigned b = a.H == 1 ? row : __umulhi(row, a.magic_h);
        const int y = (int)(row - b * (unsigned)a.H);
        const unsigned p0 = row * (unsigned)a.W + (unsigned)x;               // first output pixel
        const unsigned fbase = b * a.fbs;
        const unsigned last8 = fbase + a.img_bytes - 8u;                      // last 8 readable bytes of this frame
        const float fxs[4] = {fa.x, fa.z, fb.x, fb.z}, fys[4] = {fa.y, fa.w, fb.y, fb.w};
        unsigned lo[4][2], hi[4][2];
        unsigned eqx[4];
        float w[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float mx = map_coord(x + j, fxs[j], a.sign), my = map_coord(y, fys[j], a.sign);
            const float x0f = floorf(mx), y0f = floorf(my);
            const float fx = mx - x0f, fy = my - y0f;
            const int x0 = (int)fminf(fmaxf(x0f, -1.0e6f), 1.0e6f);
            const int y0 = (int)fminf(fmaxf(y0f, -1.0e6f), 1.0e6f);
            const bool xa = (unsigned)x0 < (unsigned)a.W, xb = (unsigned)(x0 + 1) < (unsigned)a.W;
            const bool ya = (unsigned)y0 < (unsigned)a.H, yb = (unsigned)(y0 + 1) < (unsigned)a.H;
            const float gx = 1.f - fx, gy = 1.f - fy;
            w[j][0] = (xa && ya) ? gx * gy : 0.f;
            w[j][1] = (xb && ya) ? fx * gy : 0.f;
            w[j][2] = (xa && yb) ? gx * fy : 0.f;
            w[j][3] = (xb && yb) ? fx * fy : 0.f;
            const int xc = min(max(x0, 0), a.W - 2);
            eqx[j] = x0 == xc ? 1u : 0u;                                      // taps sit at bytes 0..5 in order
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int yc = min(max(y0 + rr, 0), a.H - 1);
                const unsigned off = fbase + (unsigned)(yc * a.W + xc) * 3u;
                const unsigned start = min(off, last8);                       // the frame's last pixel pair: back up 2 bytes
                const PackedU2 v = *reinterpret_cast<const PackedU2*>(a.frame + start);
                const unsigned sh = off - start;                              // 0 or 2 bytes
                lo[j][rr] = __builtin_amdgcn_alignbyte(v.b, v.a, sh);
                hi[j][rr] = v.b >> (8u * sh);
            }
        }
        unsigned res[4][3];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned t[4];                                                    // packed pixels of the 4 taps (low 3 bytes)
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const unsigned pa = lo[j][rr];
                const unsigned pb = __builtin_amdgcn_alignbyte(hi[j][rr], lo[j][rr], 3u);
                t[rr * 2 + 0] = eqx[j] ? pa : pb;
                t[rr * 2 + 1] = eqx[j] ? pb : pa;
            }
            ofx_v2f acc01;
            float acc2;
            {
                const ofx_v2f p = {(float)(t[0] & 0xFFu), (float)((t[0] >> 8) & 0xFFu)};
                acc01 = p * w[j][0];
                acc2 = (float)((t[0] >> 16) & 0xFFu) * w[j][0];
            }
#pragma unroll
            for (int k = 1; k < 4; ++k) {
                const ofx_v2f p = {(float)(t[k] & 0xFFu), (float)((t[k] >> 8) & 0xFFu)};
                acc01 = acc01 + p * w[j][k];
                acc2 = acc2 + (float)((t[k] >> 16) & 0xFFu) * w[j][k];
            }
            res[j][0] = (unsigned)fminf(fmaxf(rintf(acc01.x), 0.f), 255.f);
            res[j][1] = (unsigned)fminf(fmaxf(rintf(acc01.y), 0.f), 255.f);
            res[j][2] = (unsigned)fminf(fmaxf(rintf(acc2), 0.f), 255.f);
        }
        uint3 o;
        o.x = res[0][0] | (res[0][1] << 8) | (res[0][2] << 16) | (res[1][0] << 24);
        o.y = res[1][1] | (res[1][2] << 8) | (res[2][0] << 16) | (res[2][1] << 24);
        o.z = res[2][2] | (res[3][0] << 8) | (res[3][1] << 16) | (res[3][2] << 24);
        *reinterpret_cast<uint3*>(a.out + (size_t)p0 * 3) = o;                // p0 % 4 == 0 -> 4-byte aligned
    }
}

template <typename T, int C>
int launch_warp_c(const T* frame, long fbs, const float* flow, T* out, int B, int H, int W, int mode,
                  float sign, const short* ti, const float* tf, hipStream_t s) {
    const long total = (long)B * H * W;
    const int grid = (int)std::min<long>((total + 255) / 256, 256L * 32);
    OfxProfScope prof(sizeof(T) == 1 ? "warp_u8" : "warp_f32", s);
    switch (mode) {
        case OFX_WARP_BILINEAR:
            hipLaunchKernelGGL((warp_kernel<T, C, OFX_WARP_BILINEAR>), dim3(grid), dim3(256), 0, s, frame, fbs, flow, out, H, W, total, sign, ti, tf);
            break;
        case OFX_WARP_BICUBIC:
            hipLaunchKernelGGL((warp_kernel<T, C, OFX_WARP_BICUBIC>), dim3(grid), dim3(256), 0, s, frame, fbs, flow, out, H, W, total, sign, ti, tf);
            break;
        case OFX_WARP_CV2_CUBIC:
            hipLaunchKernelGGL((warp_kernel<T, C, OFX_WARP_CV2_CUBIC>), dim3(grid), dim3(256), 0, s, frame, fbs, flow, out, H, W, total, sign, ti, tf);
            break;
        default: return OFX_EINVAL;
    }
    return ofx_launch_status();
}

template <typename T>
int launch_warp(const T* frame, long fbs, const float* flow, T* out, int B, int H, int W, int C, int mode,
                float sign, void* stream) {
    OFX_REQUIRE(frame && flow && out, OFX_EINVAL);
    OFX_REQUIRE(B > 0 && H > 0 && W > 0 && C >= 1 && C <= 4, OFX_EINVAL);
    OFX_REQUIRE(mode >= 0 && mode <= 2, OFX_EINVAL);
    OFX_REQUIRE((((uintptr_t)flow) & 7u) == 0, OFX_EALIGN);
    const short* tabs_i = nullptr;
    const float* tabs_f = nullptr;
    if (mode == OFX_WARP_CV2_CUBIC) {
        const Cv2Tables* t = nullptr;
        int st = ensure_tables(&t);
        if (st) return st;
        tabs_i = t->i;
        tabs_f = t->f;
    }
    hipStream_t s = (hipStream_t)stream;
    const long npix = (long)B * H * W;
    if (sizeof(T) == 1 && C == 3 && mode == OFX_WARP_BILINEAR && fbs == 0 && B >= 4) {
        // one key frame shared by the batch (the hot path's case): zero-bordered RGBX staging + lean kernel (warp_fast.hip)
        const int st = ofx_warp_bilinear_shared_launch((const uint8_t*)frame, flow, (uint8_t*)out, B, H, W, sign, s);
        if (st != OFX_EINVAL) return st;
    }
    if (sizeof(T) == 1 && C == 3 && mode == OFX_WARP_BILINEAR && (W & 3) == 0 && H >= 2 && npix * 8 < (1L << 32) &&
        (fbs == 0 || fbs == (long)H * W * 3) && (long)H * W * 3 >= 8 && (((uintptr_t)flow) & 15u) == 0 && (((uintptr_t)out) & 3u) == 0) {
        BilinArgs a;
        a.frame = (const uint8_t*)frame; a.flow = flow; a.out = (uint8_t*)out;
        a.H = H; a.W = W; a.W4 = W >> 2;
        a.fbs = (unsigned)fbs; a.img_bytes = (unsigned)((long)H * W * 3);
        a.ngroups = (unsigned)(npix >> 2);
        // umulhi(n, ceil(2^32 / d)) == n / d for n * d < 2^32 (n < ngroups <= 2^29 here, d <= 2^16 in practice;
        // the bound is checked)
        auto magic = [](unsigned d) { return (unsigned)(((1ull << 32) + d - 1) / d); };
        a.magic_w4 = magic((unsigned)a.W4); a.magic_h = magic((unsigned)H);
        a.sign = sign;
        if ((unsigned long long)a.ngroups * (unsigned)a.W4 < (1ull << 32) && (unsigned long long)B * H * (unsigned)H < (1ull << 32)) {
            const int grid = (int)std::min<long>(((long)a.ngroups + 255) / 256, 1L << 22);   // one group per thread
            OfxProfScope prof("warp_u8", s);
            hipLaunchKernelGGL(warp_bilinear_u8c3_kernel, dim3(grid), dim3(256), 0, s, a);
            return ofx_launch_status();
        }
    }
    if (sizeof(T) == 1 && C == 3 && mode != OFX_WARP_BICUBIC && W >= 4 && H >= 2) {
        const long ngroups = (long)B * H * ((W + 3) / 4);
        const int grid = (int)std::min<long>((ngroups + 255) / 256, 256L * 64);
        OfxProfScope prof("warp_u8", s);
        if (mode == OFX_WARP_BILINEAR)
            hipLaunchKernelGGL((warp_u8c3_x4_kernel<OFX_WARP_BILINEAR>), dim3(grid), dim3(256), 0, s, (const uint8_t*)frame, fbs, flow,
                               (uint8_t*)out, H, W, ngroups, sign, tabs_i);
        else
            hipLaunchKernelGGL((warp_u8c3_x4_kernel<OFX_WARP_CV2_CUBIC>), dim3(grid), dim3(256), 0, s, (const uint8_t*)frame, fbs, flow,
                               (uint8_t*)out, H, W, ngroups, sign, tabs_i);
        return ofx_launch_status();
    }
    switch (C) {
        case 1: return launch_warp_c<T, 1>(frame, fbs, flow, out, B, H, W, mode, sign, tabs_i, tabs_f, s);
        case 2: return launch_warp_c<T, 2>(frame, fbs, flow, out, B, H, W, mode, sign, tabs_i, tabs_f, s);
        case 3: return launch_warp_c<T, 3>(frame, fbs, flow, out, B, H, W, mode, sign, tabs_i, tabs_f, s);
        default: return launch_warp_c<T, 4>(frame, fbs, flow, out, B, H, W, mode, sign, tabs_i, tabs_f, s);
    }
}

// ------------------------------------------------------------------------------------------
// cv2.resize INTER_CUBIC for float HWC images
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resize_cubic_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                           int Hs, int Ws, int Hd, int Wd, int C, long total) {
    const double sy = (double)Hs / (double)Hd, sx = (double)Ws / (double)Wd;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        long t = idx / C;
        const int x = (int)(t % Wd);
        t /= Wd;
        const int y = (int)(t % Hd);
        const long b = t / Hd;
        const double fx = ((double)x + 0.5) * sx - 0.5, fy = ((double)y + 0.5) * sy - 0.5;
        const double x0 = floor(fx), y0 = floor(fy);
        float wx[4], wy[4];
        ofx_cubic_coeffs((float)(fx - x0), wx);
        ofx_cubic_coeffs((float)(fy - y0), wy);
        const float* img = src + b * (long)Hs * Ws * C;
        float acc = 0.f;
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1) {
            const int yy = min(max((int)y0 - 1 + k1, 0), Hs - 1);
            float row = 0.f;
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) {
                const int xx = min(max((int)x0 - 1 + k2, 0), Ws - 1);
                row = row + img[((long)yy * Ws + xx) * C + c] * wx[k2];
            }
            acc = acc + row * wy[k1];
        }
        dst[idx] = acc;
    }
}

// ------------------------------------------------------------------------------------------
// elliptical dilation from an LDS tile
// ------------------------------------------------------------------------------------------
constexpr int kMaxK = 31;
struct Ellipse {
    int r;
    signed char hw[kMaxK];   // half width per row dy=-r..r
};

Ellipse make_ellipse(int ksize) {
    // cv::getStructuringElement(MORPH_ELLIPSE): r=c=ksize/2, dx = cvRound(c*sqrt((r^2-dy^2)/r^2))
    Ellipse e;
    e.r = ksize / 2;
    const int r = e.r;
    const double inv_r2 = r ? 1.0 / ((double)r * r) : 0.0;
    for (int i = 0; i < kMaxK; ++i) e.hw[i] = -1;
    for (int i = 0; i < ksize; ++i) {
        const int dy = i - r;
        int dx = (int)lrint(r * std::sqrt((r * r - dy * dy) * inv_r2));
        e.hw[i] = (signed char)dx;
    }
    return e;
}

constexpr int kTileW = 64, kTileH = 16;
constexpr int kMaxR = kMaxK / 2;

// Grey-level dilation (cv2.dilate on an arbitrary uint8 image, ofgen_pdcnetplus.py:181); the binary
// masks of generate_mask / expand_mask run on bit planes instead (mask_bits.hip).
struct DilateArgs {
    const uint8_t* in_u8;
    uint8_t* out;
    int H, W;
    Ellipse el;
};

__global__ __launch_bounds__(256) void dilate_kernel(const DilateArgs a) {
    __shared__ uint8_t tile[(kTileH + 2 * kMaxR) * (kTileW + 2 * kMaxR)];
    const int r = a.el.r;
    const int tw = kTileW + 2 * r, th = kTileH + 2 * r;
    const int x0 = blockIdx.x * kTileW, y0 = blockIdx.y * kTileH;
    const long b = blockIdx.z;
    for (int i = threadIdx.x; i < tw * th; i += 256) {
        const int ly = i / tw, lx = i - ly * tw;
        const int gy = y0 - r + ly, gx = x0 - r + lx;
        uint8_t v = 0;
        if ((unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W) v = a.in_u8[(b * a.H + gy) * (long)a.W + gx];
        tile[i] = v;
    }
    __syncthreads();
    const int ty = threadIdx.x >> 4;
    const int tx = (threadIdx.x & 15) * 4;
    const int gy = y0 + ty;
    if (gy >= a.H) return;
    const long rowbase = (b * a.H + gy) * (long)a.W;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int m = 0;
        for (int dy = -r; dy <= r; ++dy) {
            const int hw = a.el.hw[dy + r];
            const uint8_t* row = &tile[(ty + r + dy) * tw + tx + j + r];
            for (int dx = -hw; dx <= hw; ++dx) m = max(m, (int)row[dx]);
        }
        const int gx = x0 + tx + j;
        if (gx < a.W) a.out[rowbase + gx] = (uint8_t)m;
    }
}

// ------------------------------------------------------------------------------------------
// small elementwise kernels
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void travel_distance_kernel(const float* __restrict__ flow, const float* __restrict__ conf,
                                                              float* __restrict__ out, int H, int W, long total, float floor_) {
    const long hw = (long)H * W;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int rem = (int)(idx % hw);
        const int y = rem / W, x = rem - y * W;
        const float2 f = reinterpret_cast<const float2*>(flow)[idx];
        // (X + disp).astype(f32) - arange: the f64->f32->subtract round trip of of_calc
        const float mx = (float)((double)map_coord(x, f.x, 1.f) - (double)x);
        const float my = (float)((double)map_coord(y, f.y, 1.f) - (double)y);
// f32 sqrt on this toolchain is 1-ulp approximate (measured: 83 % exact); the f64 sqrt rounded to f32
        // is the correctly rounded f32 result (53 >= 2*24+2 bits), matching np.sqrt bit for bit.
        // Contraction is off for this file, so the sum is mul, mul, add like numpy.
        float v = (float)sqrt((double)(mx * mx + my * my));
        if (conf[idx] < floor_) v = 0.f;
        out[idx] = v;
    }
}

// |flow| of a bare flow field (the RAFT-variant of_calc, reference ofgen.py:45-49): mul, mul, add in f32 (contraction is off for this
// file), square root through f64 = the correctly rounded f32 result, like np.sqrt
__global__ __launch_bounds__(256) void flow_magnitude_kernel(const float* __restrict__ flow, float* __restrict__ out, long total) {
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const float2 f = reinterpret_cast<const float2*>(flow)[idx];
        out[idx] = (float)sqrt((double)(f.x * f.x + f.y * f.y));
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void travel_mask_kernel(const float* __restrict__ conf, const float* __restrict__ flow,
                                                          const float* __restrict__ dist, const float* __restrict__ tin,
                                                          float* __restrict__ tout, uint8_t* __restrict__ raw, int H, int W,
                                                          long total, float thres, const short* tabi, const float* tabf) {
    const long hw = (long)H * W;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long b = idx / hw;
        const int rem = (int)(idx - b * hw);
        const int y = rem / W, x = rem - y * W;
        const float2 f = reinterpret_cast<const float2*>(flow)[idx];
        float w;
        warp_pixel<float, 1, MODE>(tin + b * hw, H, W, map_coord(x, f.x, 1.f), map_coord(y, f.y, 1.f), tabi, tabf, &w);
        float t = __fadd_rn(w, dist[idx]);
        const bool low = conf[idx] < 0.9f;
        if (low) t = 0.f;
        const bool far_ = t > thres;
        raw[idx] = (low || far_) ? 255 : 0;
        if (far_) t = 0.f;
        tout[idx] = t;
    }
}

__global__ __launch_bounds__(256) void merge_kernel(const uint8_t* __restrict__ base, const uint8_t* __restrict__ second,
                                                    const uint8_t* __restrict__ mask, uint8_t* __restrict__ out, int C, long npix) {
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < npix; idx += (long)gridDim.x * blockDim.x) {
        const bool m = mask[idx] == 255;
        for (int c = 0; c < C; ++c) out[idx * C + c] = m ? second[idx * C + c] : base[idx * C + c];
    }
}

__global__ __launch_bounds__(256) void mix_kernel(const uint8_t* __restrict__ rawf, const uint8_t* __restrict__ warped,
                                                  const uint8_t* __restrict__ mask, uint8_t* __restrict__ out, int C, long npix,
                                                  float w_lo, float w_hi) {
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < npix; idx += (long)gridDim.x * blockDim.x) {
        const float w = mask[idx] <= 127 ? w_lo : w_hi;
        const float iw = __fsub_rn(1.0f, w);
        for (int c = 0; c < C; ++c) {
            const float v = __fadd_rn(__fmul_rn((float)rawf[idx * C + c], iw), __fmul_rn((float)warped[idx * C + c], w));
            out[idx * C + c] = (uint8_t)fminf(fmaxf(v, 0.f), 255.f);   // clip then truncate (astype(uint8))
        }
    }
}

__global__ __launch_bounds__(256) void conf_sum_kernel(const float* __restrict__ x, double* __restrict__ sums, long HW,
                                                       int nchan, int chan) {
    // one workgroup per (n, slice); f64 atomics keep the result independent of slice count to ~1e-16
    const int n = blockIdx.y;
    const float* p = x + (long)n * HW * nchan + chan;
    double acc = 0.0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (long)gridDim.x * blockDim.x)
        acc += (double)p[i * nchan];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&sums[n], part[0] + part[1] + part[2] + part[3]);
}

// forward-backward consistency confidence (labelled extension: RAFT emits no confidence):
//   e = f_fw(p) + f_bw(p + f_fw(p))  (f_bw sampled bilinearly, zeros outside),
//   log_conf = -|e|^2 / (2 sigma^2), conf = exp(log_conf)
__global__ __launch_bounds__(256) void fb_conf_kernel(const float* __restrict__ fw, const float* __restrict__ bw,
                                                      float* __restrict__ conf, float* __restrict__ logc, int H, int W,
                                                      long total, float inv2s2) {
    const long hw = (long)H * W;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long b = idx / hw;
        const int rem = (int)(idx - b * hw);
        const int y = rem / W, x = rem - y * W;
        const float2 f = reinterpret_cast<const float2*>(fw)[idx];
        float s[2];
        warp_pixel<float, 2, OFX_WARP_BILINEAR>(bw + b * hw * 2, H, W, (float)x + f.x, (float)y + f.y, nullptr, nullptr, s);
        const float ex = f.x + s[0], ey = f.y + s[1];
        const float lc = -(ex * ex + ey * ey) * inv2s2;
        logc[idx] = lc;
        conf[idx] = expf(lc);
    }
}

int grid_for(long total) { return (int)std::min<long>((total + 255) / 256, 256L * 32); }

}  // namespace

// ==========================================================================================
extern "C" {

int ofx_warp_u8(const uint8_t* frame, long fbs, const float* flow, uint8_t* out, int B, int H, int W, int C,
                int mode, float sign, void* stream) {
    return launch_warp<uint8_t>(frame, fbs, flow, out, B, H, W, C, mode, sign, stream);
}

int ofx_warp_f32(const float* frame, long fbs, const float* flow, float* out, int B, int H, int W, int C,
                 int mode, float sign, void* stream) {
    return launch_warp<float>(frame, fbs, flow, out, B, H, W, C, mode, sign, stream);
}

int ofx_resize_cubic_f32(const float* src, float* dst, int B, int Hs, int Ws, int Hd, int Wd, int C, void* stream) {
    OFX_REQUIRE(src && dst && B > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0 && C > 0, OFX_EINVAL);
    const long total = (long)B * Hd * Wd * C;
    hipStream_t s = (hipStream_t)stream;
    OfxProfScope prof("resize_cubic", s);
    hipLaunchKernelGGL(resize_cubic_kernel, dim3(grid_for(total)), dim3(256), 0, s, src, dst, Hs, Ws, Hd, Wd, C, total);
    return ofx_launch_status();
}

int ofx_generate_mask(const float* conf, float* log_conf, uint8_t* mask, int B, int H, int W, float thres,
                      int ksize, int cmp_gt, void* stream) {
    OFX_REQUIRE(conf && mask && B > 0 && H > 0 && W > 0, OFX_EINVAL);
    OFX_REQUIRE(ksize >= 1 && ksize <= kMaxK && (ksize & 1), OFX_EINVAL);
    // binary morphology runs on bit planes (mask_bits.hip)
    const Ellipse el = make_ellipse(ksize);
    return ofx_mask_bits_launch(cmp_gt ? OFX_MSRC_CONF_NGT : OFX_MSRC_CONF_LT, conf, log_conf, nullptr, nullptr, mask, B, H, W, thres, 0,
                                el.r, el.hw, "generate_mask", (hipStream_t)stream);
}

int ofx_dilate_u8(const uint8_t* in, uint8_t* out, int B, int H, int W, int ksize, void* stream) {
    OFX_REQUIRE(in && out && in != out && B > 0 && H > 0 && W > 0, OFX_EINVAL);
    OFX_REQUIRE(ksize >= 1 && ksize <= kMaxK && (ksize & 1), OFX_EINVAL);
    DilateArgs a{};
    a.in_u8 = in; a.out = out; a.H = H; a.W = W; a.el = make_ellipse(ksize);
    hipStream_t s = (hipStream_t)stream;
    OfxProfScope prof("dilate_u8", s);
    hipLaunchKernelGGL(dilate_kernel, dim3(ofx_cdiv(W, kTileW), ofx_cdiv(H, kTileH), B), dim3(256), 0, s, a);
    return ofx_launch_status();
}

int ofx_expand_mask(const uint8_t* mask, const uint8_t* image_bgr, uint8_t* out, uint8_t* scratch, int B, int H,
                    int W, int edge_thres, int ksize, void* stream) {
    (void)scratch;
    OFX_REQUIRE(mask && image_bgr && out && B > 0 && H > 0 && W > 0, OFX_EINVAL);
    OFX_REQUIRE(ksize >= 1 && ksize <= kMaxK && (ksize & 1), OFX_EINVAL);
    const Ellipse el = make_ellipse(ksize);
    return ofx_mask_bits_launch(OFX_MSRC_EDGES, nullptr, nullptr, image_bgr, mask, out, B, H, W, 0.f, edge_thres, el.r, el.hw,
                                "expand_mask", (hipStream_t)stream);
}

int ofx_travel_distance(const float* flow, const float* conf, float* out, int B, int H, int W, float conf_floor,
                        void* stream) {
    OFX_REQUIRE(flow && conf && out && B > 0 && H > 0 && W > 0, OFX_EINVAL);
    const long total = (long)B * H * W;
    hipStream_t s = (hipStream_t)stream;
    OfxProfScope prof("travel_distance", s);
    hipLaunchKernelGGL(travel_distance_kernel, dim3(grid_for(total)), dim3(256), 0, s, flow, conf, out, H, W, total, conf_floor);
    return ofx_launch_status();
}

int ofx_flow_magnitude(const float* flow, float* out, long n, void* stream) {
    OFX_REQUIRE(flow && out && n > 0, OFX_EINVAL);
    OFX_REQUIRE((((uintptr_t)flow) & 7u) == 0, OFX_EALIGN);
    hipStream_t s = (hipStream_t)stream;
    OfxProfScope prof("flow_magnitude", s);
    hipLaunchKernelGGL(flow_magnitude_kernel, dim3(grid_for(n)), dim3(256), 0, s, flow, out, n);
    return ofx_launch_status();
}

int ofx_travel_mask(const float* conf, const float* flow, const float* dist, const float* travel_in,
                    float* travel_out, uint8_t* raw, int B, int H, int W, float thres, int warp_mode, void* stream) {
    OFX_REQUIRE(conf && flow && dist && travel_in && travel_out && raw && travel_in != travel_out, OFX_EINVAL);
    OFX_REQUIRE(B > 0 && H > 0 && W > 0 && warp_mode >= 0 && warp_mode <= 2, OFX_EINVAL);
    const short* tabs_i = nullptr;
    const float* tabs_f = nullptr;
    if (warp_mode == OFX_WARP_CV2_CUBIC) {
        const Cv2Tables* t = nullptr;
        int st = ensure_tables(&t);
        if (st) return st;
        tabs_i = t->i;
        tabs_f = t->f;
    }
    const long total = (long)B * H * W;
    hipStream_t s = (hipStream_t)stream;
    const int g = grid_for(total);
    OfxProfScope prof("travel_mask", s);
    if (warp_mode == OFX_WARP_BILINEAR)
        hipLaunchKernelGGL((travel_mask_kernel<OFX_WARP_BILINEAR>), dim3(g), dim3(256), 0, s, conf, flow, dist, travel_in, travel_out, raw, H, W, total, thres, tabs_i, tabs_f);
    else if (warp_mode == OFX_WARP_BICUBIC)
        hipLaunchKernelGGL((travel_mask_kernel<OFX_WARP_BICUBIC>), dim3(g), dim3(256), 0, s, conf, flow, dist, travel_in, travel_out, raw, H, W, total, thres, tabs_i, tabs_f);
    else
        hipLaunchKernelGGL((travel_mask_kernel<OFX_WARP_CV2_CUBIC>), dim3(g), dim3(256), 0, s, conf, flow, dist, travel_in, travel_out, raw, H, W, total, thres, tabs_i, tabs_f);
    return ofx_launch_status();
}

int ofx_merge_images(const uint8_t* base, const uint8_t* second, const uint8_t* mask, uint8_t* out, int B, int H,
                     int W, int C, void* stream) {
    OFX_REQUIRE(base && second && mask && out && B > 0 && H > 0 && W > 0 && C > 0, OFX_EINVAL);
    const long npix = (long)B * H * W;
    hipStream_t s = (hipStream_t)stream;
    OfxProfScope prof("merge_images", s);
    hipLaunchKernelGGL(merge_kernel, dim3(grid_for(npix)), dim3(256), 0, s, base, second, mask, out, C, npix);
    return ofx_launch_status();
}

int ofx_mix_frames(const uint8_t* raw, const uint8_t* warped, const uint8_t* mask, uint8_t* out, int B, int H, int W,
                   int C, float ppw, void* stream) {
    OFX_REQUIRE(raw && warped && mask && out && B > 0 && H > 0 && W > 0 && C > 0, OFX_EINVAL);
    const long npix = (long)B * H * W;
    hipStream_t s = (hipStream_t)stream;
    const float w_lo = ppw;
    const float w_hi = (float)(1.0 - (double)ppw);
    OfxProfScope prof("mix_frames", s);
    hipLaunchKernelGGL(mix_kernel, dim3(grid_for(npix)), dim3(256), 0, s, raw, warped, mask, out, C, npix, w_lo, w_hi);
    return ofx_launch_status();
}

int ofx_conf_sum(const float* x, double* sums, int N, long HW, int nchan, int chan, void* stream) {
    OFX_REQUIRE(x && sums && N > 0 && HW > 0 && nchan > 0 && chan >= 0 && chan < nchan, OFX_EINVAL);
    hipStream_t s = (hipStream_t)stream;
    OFX_HIP_CHECK(hipMemsetAsync(sums, 0, sizeof(double) * N, s));
    const int slices = (int)std::min<long>((HW + 256L * 16 - 1) / (256L * 16), 64);
    OfxProfScope prof("conf_sum", s);
    hipLaunchKernelGGL(conf_sum_kernel, dim3(slices, N), dim3(256), 0, s, x, sums, HW, nchan, chan);
    return ofx_launch_status();
}

int ofx_fb_confidence(const float* flow_fw, const float* flow_bw, float* conf, float* log_conf, int B, int H, int W,
                      float sigma, void* stream) {
    OFX_REQUIRE(flow_fw && flow_bw && conf && log_conf && B > 0 && H > 0 && W > 0 && sigma > 0.f, OFX_EINVAL);
    const long total = (long)B * H * W;
    hipStream_t s = (hipStream_t)stream;
    OfxProfScope prof("fb_confidence", s);
    hipLaunchKernelGGL(fb_conf_kernel, dim3(grid_for(total)), dim3(256), 0, s, flow_fw, flow_bw, conf, log_conf, H, W, total,
                       1.0f / (2.0f * sigma * sigma));
    return ofx_launch_status();
}

int ofx_warp_and_mask(const uint8_t* frame, long fbs, const float* flow, const float* conf, uint8_t* warped,
                      uint8_t* mask, int B, int H, int W, int C, int warp_mode, float sign, float thres, int ksize,
                      int cmp_gt, void* stream) {
    int st = ofx_warp_u8(frame, fbs, flow, warped, B, H, W, C, warp_mode, sign, stream);
    if (st) return st;
    return ofx_generate_mask(conf, nullptr, mask, B, H, W, thres, ksize, cmp_gt, stream);
}

}  // extern "C"
