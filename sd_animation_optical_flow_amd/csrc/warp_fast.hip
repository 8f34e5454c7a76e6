// Bilinear backward warp of ONE shared uint8 RGB key frame along B flow fields -- the warp of the hot path
// (pdcnet_of.py:34-42 semantics in its `north_star` bilinear mode: out(y,x) = frame(y + fy, x + fx), zeros outside,
// grid_sample(bilinear, zeros, align_corners=True) arithmetic; every frame of a shard warps the same AI key frame).
//
// The byte-triplet kernel in warp_mask.hip spends ~100 VALU instructions per pixel on border weights, unaligned
// 3-byte taps and tap swaps and is VALU-bound at 3.2 TB/s.  Here the key frame is first expanded ONCE per call into a
// zero-bordered RGBX image (4 bytes per pixel, 2 pixels of zeros all around; 1.6 MB at 512x768, stays in L2):
//   * a tap row is one aligned 8-byte load (two RGBX pixels), no byte shuffling;
//   * coordinates are clamped into the border with one v_med3 per axis -- out-of-image taps READ zeros, so there are
//     no validity compares, no zeroed weights and no tap swaps;
//   * blending is packed fp32 FMA; round-half-even + byte extraction is one add of 1.5 * 2^23 per channel and byte
//     permutes (values are convex combinations of bytes: no clamp needed).
// ~50 VALU instructions per pixel (was ~100).  A workgroup covers a 64 x 16 pixel tile so that the source rows its taps
// touch are shared inside one CU's L1.  Measured (B = 64, 512x768, tools/warp_bench.py): 64-69 us = 4.0-4.3 TB/s of
// algorithmic bytes against 85 us before.  What bounds it now (DESIGN.md has the table): removing any ONE of the three
// memory streams -- flow loads, taps, stores -- gives 46-48 us (the rate of the stream pair that remains), while
// restructurings that keep all three (row-pair taps in one 16-byte load, LDS-staged taps per workgroup or per
// wavefront, lane-contiguous taps, software-pipelined grid-stride loops, 8 pixels per thread, dense 16-byte stores,
// non-temporal flow loads) all measured 69-117 us: the three dependent round trips of a wavefront through one CU's
// in-order vector-memory path add up, and HBM writes (4.4 TB/s peak here vs 6.4 TB/s for reads) share the channel.
// This file is compiled with FMA contraction ON (u8 results may differ from the weights-form oracle by 1 LSB on
// rounding ties only; the tests bound that).
#include "ofx_internal.h"
#include "upsample_inl.h"

namespace {

constexpr int kPad = 2;

struct PadArgs {
    const uint8_t* frame;     // [H][W][3]
    uint32_t* pad;            // [(H + 4)][(W + 4)] RGBX
    int H, W, Wp;
    unsigned total;           // (H + 4) * (W + 4)
};

__global__ __launch_bounds__(256) void pad_rgbx_kernel(const PadArgs a) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= a.total) return;
    const int yp = (int)(i / (unsigned)a.Wp), xp = (int)(i - (unsigned)yp * (unsigned)a.Wp);
    const int y = yp - kPad, x = xp - kPad;
    uint32_t v = 0;
    if ((unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W) {
        const uint8_t* p = a.frame + ((size_t)y * a.W + x) * 3;
        v = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
    }
    a.pad[i] = v;
}

struct FastArgs {
    const uint32_t* pad;      // points at padded pixel (kPad, kPad)
    const float* flow;
    uint8_t* out;
    int H, W, Wp;
    float sign;
};

typedef float v2f __attribute__((ext_vector_type(2)));

// 4 consecutive output pixels per thread; x % 4 == 0; row = b * H + y
__device__ __forceinline__ void warp_group(const FastArgs& a, const unsigned row, const int x, const int y, const float4 fa,
                                           const float4 fb) {
    const float xf = (float)x, yf = (float)y, Wf = (float)a.W, Hf = (float)a.H;
    const float fxs[4] = {fa.x, fa.z, fb.x, fb.z}, fys[4] = {fa.y, fa.w, fb.y, fb.w};
    uint2 top[4], bot[4];
    float wx[4], wy[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // (float)(x + j) + sign * f: the exact sum rounded once (the reference adds in f64 and casts)
        const float mx = __builtin_fmaf(a.sign, fxs[j], xf + (float)j), my = __builtin_fmaf(a.sign, fys[j], yf);
        const float x0f = __builtin_floorf(mx), y0f = __builtin_floorf(my);
        wx[j] = mx - x0f;
        wy[j] = my - y0f;
        // clamp into the zero border in the float domain (one v_med3_f32 per axis; a NaN coordinate lands on the
        // border too), then the conversion is exact
        const int x0 = (int)__builtin_amdgcn_fmed3f(x0f, -(float)kPad, Wf), y0 = (int)__builtin_amdgcn_fmed3f(y0f, -(float)kPad, Hf);
        const int idx = __mul24(y0, a.Wp) + x0;
        top[j] = *reinterpret_cast<const uint2*>(a.pad + idx);
        bot[j] = *reinterpret_cast<const uint2*>(a.pad + idx + a.Wp);
    }
    unsigned r[4][3];
    const float kMagic = 12582912.0f;                                        // 1.5 * 2^23: (v + kMagic) holds rne(v) in its low byte
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float fx = wx[j], fy = wy[j], gx = 1.f - fx, gy = 1.f - fy;
        const float w00 = gx * gy, w01 = fx * gy, w10 = gx * fy, w11 = fx * fy;
        const unsigned p00 = top[j].x, p01 = top[j].y, p10 = bot[j].x, p11 = bot[j].y;
        v2f acc = v2f{(float)(p00 & 0xFFu), (float)((p00 >> 8) & 0xFFu)} * w00;
        float acc2 = (float)((p00 >> 16) & 0xFFu) * w00;
        acc += v2f{(float)(p01 & 0xFFu), (float)((p01 >> 8) & 0xFFu)} * w01;
        acc2 += (float)((p01 >> 16) & 0xFFu) * w01;
        acc += v2f{(float)(p10 & 0xFFu), (float)((p10 >> 8) & 0xFFu)} * w10;
        acc2 += (float)((p10 >> 16) & 0xFFu) * w10;
        acc += v2f{(float)(p11 & 0xFFu), (float)((p11 >> 8) & 0xFFu)} * w11;
        acc2 += (float)((p11 >> 16) & 0xFFu) * w11;
        r[j][0] = __float_as_uint(acc.x + kMagic);
        r[j][1] = __float_as_uint(acc.y + kMagic);
        r[j][2] = __float_as_uint(acc2 + kMagic);
    }
    // byte 0 of each r -> 12 packed bytes.  __builtin_amdgcn_perm(hi, lo, sel): selector byte k picks byte (0-3 from lo,
    // 4-7 from hi) for output byte k
    auto pair = [](unsigned hi, unsigned lo) { return __builtin_amdgcn_perm(hi, lo, 0x0c0c0400u); };   // [lo.b0, hi.b0, 0, 0]
    auto quad = [](unsigned hi2, unsigned lo2) { return __builtin_amdgcn_perm(hi2, lo2, 0x05040100u); }; // [lo2.b0, lo2.b1, hi2.b0, hi2.b1]
    uint3 o;
    o.x = quad(pair(r[1][0], r[0][2]), pair(r[0][1], r[0][0]));
    o.y = quad(pair(r[2][1], r[2][0]), pair(r[1][2], r[1][1]));
    o.z = quad(pair(r[3][2], r[3][1]), pair(r[3][0], r[2][2]));
    *reinterpret_cast<uint3*>(a.out + ((size_t)row * a.W + x) * 3) = o;      // (row * W + x) % 4 == 0: 4-byte aligned
}

constexpr int kTileW = 64, kTileH = 16;

__global__ __launch_bounds__(256) void warp_bilinear_shared_kernel(const FastArgs a, const int tiles_x, const int tiles_y) {
    const int txi = blockIdx.x % tiles_x;
    const int rest = blockIdx.x / tiles_x;
    const int tyi = rest % tiles_y;
    const int b = rest / tiles_y;
    const int x = txi * kTileW + ((threadIdx.x & 15) << 2), y = tyi * kTileH + (threadIdx.x >> 4);
    if (x >= a.W || y >= a.H) return;                        // W % 4 == 0: a group of 4 is inside or outside as a whole
    const unsigned row = (unsigned)b * (unsigned)a.H + (unsigned)y;
    const size_t p0 = (size_t)row * a.W + x;
    const float4 fa = reinterpret_cast<const float4*>(a.flow)[p0 >> 1];
    const float4 fb = reinterpret_cast<const float4*>(a.flow)[(p0 >> 1) + 1];
    warp_group(a, row, x, y, fa, fb);
}

// ---- convex upsample + warp in one pass: the flow of a lane's four fine pixels is still in registers when the upsample ends, so the
// AI key frame is sampled right there -- the 201 MB re-read of flow_up (64 frames at 512x768) and one of the warp's three dependent
// round trips disappear; flow_up itself is written only when the caller wants it.  Same arithmetic as upsample_kernel followed by
// warp_bilinear_shared_kernel (the two device functions are shared), so the results are bit-identical.
__global__ __launch_bounds__(256) void upsample_warp_kernel(const float* __restrict__ coords1, const float* __restrict__ mask,
                                                            float* __restrict__ flow_up, const FastArgs a, int h, int w, long M) {
    const OfxUpLane o = ofx_upsample_lane(coords1, mask, h, w, M);
    if (!o.valid) return;
    if (flow_up) ofx_upsample_store(o, flow_up, h, w);
    const int X = o.x * 8 + o.j, Y = o.y * 8 + o.i;
    const unsigned row = (unsigned)(o.b * a.H + Y);
    warp_group(a, row, X, Y, make_float4(o.ax.x, o.ay.x, o.ax.y, o.ay.y), make_float4(o.ax.z, o.ay.z, o.ax.w, o.ay.w));
}

}  // namespace

// zero-bordered RGBX copy of a key frame for the bilinear warp: `pad` holds (H + 4) * (W + 4) uint32
size_t ofx_warp_pad_bytes(int H, int W) { return (size_t)(W + 2 * kPad) * (H + 2 * kPad) * sizeof(uint32_t); }
bool ofx_upsample_warp_ok(int B, int H, int W) {
    return (W & 7) == 0 && (H & 7) == 0 && H >= 8 && (long)B * H * W * 3 < (1L << 40) && H + 2 * kPad < (1 << 15) && W + 2 * kPad < (1 << 15) &&
           (long)B * H < (1L << 31);
}
int ofx_warp_pad_launch(const uint8_t* frame, void* pad, int H, int W, hipStream_t s) {
    const int Wp = W + 2 * kPad, Hp = H + 2 * kPad;
    PadArgs pa{frame, (uint32_t*)pad, H, W, Wp, (unsigned)(Wp * Hp)};
    OfxProfScope prof("warp_pad_keyframe", s);
    hipLaunchKernelGGL(pad_rgbx_kernel, dim3(ofx_cdiv(pa.total, 256)), dim3(256), 0, s, pa);
    return ofx_launch_status();
}
// coords1 [B*h*w][2], mask [B*h*w][576] -> warped u8 [B, 8h, 8w, 3] (and flow_up f32 [B, 8h, 8w, 2] unless NULL); `pad` from ofx_warp_pad_launch
int ofx_upsample_warp_launch(const float* coords1, const float* mask, float* flow_up, const void* pad, uint8_t* warped, int B, int h, int w,
                             float sign, hipStream_t s) {
    const int H = h * 8, W = w * 8;
    OFX_REQUIRE(coords1 && mask && pad && warped && ofx_upsample_warp_ok(B, H, W), OFX_EINVAL);
    OFX_REQUIRE((((uintptr_t)coords1) & 7u) == 0 && ofx_aligned16(mask) && (!flow_up || ofx_aligned16(flow_up)) && (((uintptr_t)warped) & 3u) == 0,
                OFX_EALIGN);
    const int Wp = W + 2 * kPad;
    FastArgs a;
    a.pad = (const uint32_t*)pad + (size_t)kPad * Wp + kPad;
    a.flow = nullptr; a.out = warped;
    a.H = H; a.W = W; a.Wp = Wp;
    a.sign = sign;
    const long M = (long)B * h * w;
    const long groups = (M / w) * ((w + 3) / 4);               // 4 coarse pixels per wavefront, 4 wavefronts per workgroup
    OfxProfScope prof("upsample_warp", s);
    hipLaunchKernelGGL(upsample_warp_kernel, dim3((unsigned)((groups + 3) / 4)), dim3(256), 0, s, coords1, mask, flow_up, a, h, w, M);
    return ofx_launch_status();
}

extern "C" {

// RAFT.upsample_flow (raft.py:72-83) + the bilinear backward warp of ONE shared uint8 RGB frame (pdcnet_of.py:34-42 in its bilinear
// mode; sign = +1: out(y,x) = frame(y + fy, x + fx), -1: the RAFT-side convention of ofgen_keyframe_inpaint.py:92-98) in one kernel.
// coords1 [B*h*w][2] and mask [B*h*w][576] as ofx_upsample_flow takes them; frame u8 [8h][8w][3]; warped u8 [B][8h][8w][3];
// flow_up f32 [B][8h][8w][2] or NULL (not written).  Bit-identical to ofx_upsample_flow followed by ofx_warp_u8(bilinear).
int ofx_upsample_flow_warp(const float* coords1, const float* mask, float* flow_up, const uint8_t* frame, uint8_t* warped, int B, int h, int w,
                           float sign, void* stream) {
    OFX_REQUIRE(coords1 && mask && frame && warped && B > 0 && h > 0 && w > 0, OFX_EINVAL);
    OFX_REQUIRE(ofx_upsample_warp_ok(B, h * 8, w * 8), OFX_EINVAL);
    hipStream_t s = (hipStream_t)stream;
    void* pad = nullptr;
    OFX_HIP_CHECK(hipMallocAsync(&pad, ofx_warp_pad_bytes(h * 8, w * 8), s));
    int st = ofx_warp_pad_launch(frame, pad, h * 8, w * 8, s);
    if (!st) st = ofx_upsample_warp_launch(coords1, mask, flow_up, pad, warped, B, h, w, sign, s);
    const hipError_t e = hipFreeAsync(pad, s);
    return st ? st : (int)e;
}

}  // extern "C"

// Returns 0 when the launch was taken, OFX_EINVAL when the shape is outside this path's limits (the caller then uses
// the generic kernels), or a HIP error.  The padded key frame is a stream-ordered allocation (hipMallocAsync /
// hipFreeAsync on the caller's stream: no cross-stream sharing, nothing to synchronise).
int ofx_warp_bilinear_shared_launch(const uint8_t* frame, const float* flow, uint8_t* out, int B, int H, int W, float sign,
                                    hipStream_t s) {
    const long npix = (long)B * H * W;
    if ((W & 3) != 0 || H < 1 || npix * 3 >= (1L << 40) || H + 2 * kPad >= (1 << 15) || W + 2 * kPad >= (1 << 15)) return OFX_EINVAL;
    if ((long)B * H >= (1L << 31)) return OFX_EINVAL;
    if ((((uintptr_t)flow) & 15u) != 0 || (((uintptr_t)out) & 3u) != 0) return OFX_EINVAL;
    const int Wp = W + 2 * kPad, Hp = H + 2 * kPad;
    const int tiles_x = ofx_cdiv(W, kTileW), tiles_y = ofx_cdiv(H, kTileH);
    const long nwg = (long)tiles_x * tiles_y * B;
    if (nwg >= (1L << 31)) return OFX_EINVAL;
    const size_t pad_bytes = (size_t)Wp * Hp * sizeof(uint32_t);
    uint32_t* pad = nullptr;
    if (hipMallocAsync((void**)&pad, pad_bytes, s) != hipSuccess) {
        (void)hipGetLastError();
        return OFX_EINVAL;
    }
    PadArgs pa{frame, pad, H, W, Wp, (unsigned)(Wp * Hp)};
    FastArgs a;
    a.pad = pad + (size_t)kPad * Wp + kPad;
    a.flow = flow; a.out = out;
    a.H = H; a.W = W; a.Wp = Wp;
    a.sign = sign;
    {
        OfxProfScope prof("warp_pad_keyframe", s);
        hipLaunchKernelGGL(pad_rgbx_kernel, dim3(ofx_cdiv(pa.total, 256)), dim3(256), 0, s, pa);
    }
    {
        OfxProfScope prof("warp_u8", s);
        hipLaunchKernelGGL(warp_bilinear_shared_kernel, dim3((unsigned)nwg), dim3(256), 0, s, a, tiles_x, tiles_y);
    }
    int st = ofx_launch_status();
    hipError_t e = hipFreeAsync(pad, s);
    return st ? st : (int)e;
}
