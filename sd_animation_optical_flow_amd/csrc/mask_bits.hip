// Binary inpaint-mask morphology on bit planes.
//
// generate_mask (ofgen_keyframe_inpaint.py:317-322) and the edge half of expand_mask (:968-973) dilate a
// BINARY image with an elliptical structuring element.  A wavefront turns 64 pixels of a row into one
// 64-bit word with a single ballot, so a 64 x 32 pixel tile (+ halo) is 2 x (32+2r) machine words in LDS;
// the dilation is then shifts and ORs of 128-bit rows (the ellipse is one half-width per row), and the
// result is expanded back to 0/255 bytes with 8-byte stores.  Reads every confidence value ~1.3x (halo),
// writes every mask byte once: HBM-bound, bit-exact by construction.
//
// The byte-wise LDS kernel in warp_mask.hip remains the path for grey-value dilation (ofx_dilate_u8).
#include "ofx_internal.h"

namespace {

constexpr int kTW = 64;       // tile width  = one ballot
constexpr int kTH = 32;       // tile height
constexpr int kMaxR = 15;     // ksize <= 31


struct BitArgs {
    const float* conf;
    float* log_conf;
    const uint8_t* image;     // OFX_MSRC_EDGES: BGR image
    const uint8_t* or_mask;   // optional: out |= or_mask
    uint8_t* out;
    int H, W;
    float thres;
    int edge_thres;
    int r;
    signed char hw[2 * kMaxR + 1];
};

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

template <int SRC>
__device__ __forceinline__ bool src_bit(const BitArgs& a, long b, int y, int x) {
    if ((unsigned)y >= (unsigned)a.H || (unsigned)x >= (unsigned)a.W) return false;
    const long pix = (b * a.H + y) * (long)a.W + x;
    if (SRC == OFX_MSRC_CONF_LT) return a.conf[pix] < a.thres;
    if (SRC == OFX_MSRC_CONF_NGT) return !(a.conf[pix] > a.thres);
    // |laplacian| per channel wraps mod 256 (the reference's astype(uint8)), cv::cvtColor RGB2GRAY fixed
    // point applied to the BGR image, then > edge_thres
    const uint8_t* img = a.image + b * (long)a.H * a.W * 3;
    const int ym = reflect101(y - 1, a.H), yp = reflect101(y + 1, a.H);
    const int xm = reflect101(x - 1, a.W), xp = reflect101(x + 1, a.W);
    int g[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int lap = (int)img[((long)ym * a.W + x) * 3 + c] + (int)img[((long)yp * a.W + x) * 3 + c] +
                        (int)img[((long)y * a.W + xm) * 3 + c] + (int)img[((long)y * a.W + xp) * 3 + c] -
                        4 * (int)img[((long)y * a.W + x) * 3 + c];
        g[c] = abs(lap) & 255;
    }
    return ((g[0] * 9798 + g[1] * 19235 + g[2] * 3735 + (1 << 14)) >> 15) > a.edge_thres;
}

struct U128 {
    unsigned long long lo, hi;
};
__device__ __forceinline__ U128 shr(U128 v, int s) {   // 0 <= s < 64
    if (s == 0) return v;
    U128 o;
    o.lo = (v.lo >> s) | (v.hi << (64 - s));
    o.hi = v.hi >> s;
    return o;
}

template <int SRC>
__global__ __launch_bounds__(256) void mask_bits_kernel(const BitArgs a) {
    __shared__ unsigned long long rows[(kTH + 2 * kMaxR) * 2];
    __shared__ unsigned long long outbits[kTH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = a.r;
    const int x0 = blockIdx.x * kTW, y0 = blockIdx.y * kTH;
    const long b = blockIdx.z;
    const int nrows = kTH + 2 * r;
    // ---- phase 1: one ballot per 64 columns; bit p of a row <-> column x0 - r + p
    // all of a wave's row loads are issued before the first ballot consumes one (a ballot per load would
    // serialise ~10 HBM round trips per wave)
    constexpr int kIter = (kTH + 2 * kMaxR + 3) / 4;
    bool pa[kIter], pb[kIter];
#pragma unroll
    for (int it = 0; it < kIter; ++it) {
        const int ry = wave + 4 * it;
        const int y = y0 - r + ry;
        pa[it] = ry < nrows ? src_bit<SRC>(a, b, y, x0 - r + lane) : false;
        pb[it] = (ry < nrows && lane < 2 * r) ? src_bit<SRC>(a, b, y, x0 - r + 64 + lane) : false;
    }
#pragma unroll
    for (int it = 0; it < kIter; ++it) {
        const int ry = wave + 4 * it;
        if (ry >= nrows) break;
        const int y = y0 - r + ry;
        const int xa = x0 - r + lane;
        const bool ba = pa[it];
        const int xb = x0 - r + 64 + lane;
        const bool bb = pb[it];
        const unsigned long long ma = __ballot(ba), mb = __ballot(bb);
        if (lane == 0) {
            rows[ry * 2] = ma;
            rows[ry * 2 + 1] = mb;
        }
        if ((SRC == OFX_MSRC_CONF_LT || SRC == OFX_MSRC_CONF_NGT) && a.log_conf != nullptr && ry >= r && ry < r + kTH) {
            // generate_mask's side effect: log_confidence[low] = 0 on the tile's own pixels
            if (ba && lane >= r && xa < a.W) a.log_conf[(b * a.H + y) * (long)a.W + xa] = 0.f;
            if (bb && lane < r && xb < a.W) a.log_conf[(b * a.H + y) * (long)a.W + xb] = 0.f;
        }
    }
    __syncthreads();
    // ---- phase 2: out[row][c] = OR over dy of OR_{|d| <= hw[dy]} row[dy][c + d].  All 256 threads take part: thread
    // (row, part) handles the structuring-element rows dy = -r + part, -r + part + 8, ... and the partial words are
    // combined with LDS atomics (one 32-thread wave doing all rows serially was half of this kernel's time)
    if (threadIdx.x < kTH) outbits[threadIdx.x] = 0;
    __syncthreads();
    {
        const int t = threadIdx.x & (kTH - 1), part = threadIdx.x >> 5;
        unsigned long long acc = 0;
        for (int dy = -r + part; dy <= r; dy += 256 / kTH) {
            const int hw = a.hw[dy + r];
            if (hw < 0) continue;
            U128 v;
            v.lo = rows[(t + r + dy) * 2];
            v.hi = rows[(t + r + dy) * 2 + 1];
            U128 s = v;   // s[q] = OR_{e=0..2hw} v[q+e]
            for (int e = 1; e <= 2 * hw; ++e) {
                const U128 sh = shr(v, e);
                s.lo |= sh.lo;
                s.hi |= sh.hi;
            }
            acc |= shr(s, r - hw).lo;   // column c <-> bit c + r; dilated value = s[c + r - hw]
        }
        if (acc) atomicOr(&outbits[t], acc);
    }
    __syncthreads();
    // ---- phase 3: expand to bytes, 8 pixels per thread
    const int row = threadIdx.x >> 3, seg = threadIdx.x & 7;
    const int y = y0 + row;
    if (y >= a.H) return;
    const int xs = x0 + seg * 8;
    if (xs >= a.W) return;
    const unsigned bits = (unsigned)((outbits[row] >> (seg * 8)) & 0xFFu);
    const long base = (b * a.H + y) * (long)a.W + xs;
    if (xs + 8 <= a.W && ((base & 7) == 0)) {
        unsigned long long v = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if ((bits >> i) & 1u) v |= 0xFFull << (8 * i);
        if (a.or_mask) v |= *reinterpret_cast<const unsigned long long*>(a.or_mask + base);
        *reinterpret_cast<unsigned long long*>(a.out + base) = v;
    } else {
        for (int i = 0; i < 8 && xs + i < a.W; ++i) {
            uint8_t v = ((bits >> i) & 1u) ? 255 : 0;
            if (a.or_mask) v |= a.or_mask[base + i];
            a.out[base + i] = v;
        }
    }
}

// ---- narrow variant (r <= 4: the 7x7 and 9x9 structuring elements): the tile is 56 output columns wide so that
// tile + halo is exactly one 64-lane load / one ballot per row, and the dilation runs on 64-bit rows.
constexpr int kNW = 56, kNH = 4;      // output columns per tile, fixed halo (>= r)
template <int SRC>
__global__ __launch_bounds__(256) void mask_bits_narrow_kernel(const BitArgs a) {
    __shared__ unsigned long long rows[kTH + 2 * kNH];
    __shared__ unsigned long long outbits[kTH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = a.r;
    const int x0 = blockIdx.x * kNW, y0 = blockIdx.y * kTH;
    const long b = blockIdx.z;
    const int nrows = kTH + 2 * r;
    constexpr int kIter = (kTH + 2 * kNH + 3) / 4;
    bool pa[kIter];
    const int xa = x0 - kNH + lane;                       // bit p of a row <-> column x0 - kNH + p
#pragma unroll
    for (int it = 0; it < kIter; ++it) {
        const int ry = wave + 4 * it;
        pa[it] = ry < nrows ? src_bit<SRC>(a, b, y0 - r + ry, xa) : false;
    }
#pragma unroll
    for (int it = 0; it < kIter; ++it) {
        const int ry = wave + 4 * it;
        if (ry >= nrows) break;
        const unsigned long long ma = __ballot(pa[it]);
        if (lane == 0) rows[ry] = ma;
        if ((SRC == OFX_MSRC_CONF_LT || SRC == OFX_MSRC_CONF_NGT) && a.log_conf != nullptr && ry >= r && ry < r + kTH) {
            // generate_mask's side effect: log_confidence[low] = 0 on the tile's own pixels
            if (pa[it] && lane >= kNH && lane < kNH + kNW && xa < a.W) a.log_conf[(b * a.H + (y0 - r + ry)) * (long)a.W + xa] = 0.f;
        }
    }
    if (threadIdx.x < kTH) outbits[threadIdx.x] = 0;
    __syncthreads();
    {
        const int t = threadIdx.x & (kTH - 1), part = threadIdx.x >> 5;
        unsigned long long acc = 0;
        for (int dy = -r + part; dy <= r; dy += 256 / kTH) {
            const int hw = a.hw[dy + r];
            if (hw < 0) continue;
            const unsigned long long v = rows[t + r + dy];
            unsigned long long s = v;                     // s[q] = OR_{e=0..2hw} v[q+e]
            for (int e = 1; e <= 2 * hw; ++e) s |= v >> e;
            acc |= s >> (kNH - hw);                       // column c <-> bit c + kNH; dilated value = s[c + kNH - hw]
        }
        if (acc) atomicOr(&outbits[t], acc);
    }
    __syncthreads();
    const int row = threadIdx.x >> 3, seg = threadIdx.x & 7;
    const int y = y0 + row;
    if (seg >= kNW / 8 || y >= a.H) return;
    const int xs = x0 + seg * 8;
    if (xs >= a.W) return;
    const unsigned bits = (unsigned)((outbits[row] >> (seg * 8)) & 0xFFu);
    const long base = (b * a.H + y) * (long)a.W + xs;
    if (xs + 8 <= a.W && ((base & 7) == 0)) {
        unsigned long long v = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if ((bits >> i) & 1u) v |= 0xFFull << (8 * i);
        if (a.or_mask) v |= *reinterpret_cast<const unsigned long long*>(a.or_mask + base);
        *reinterpret_cast<unsigned long long*>(a.out + base) = v;
    } else {
        for (int i = 0; i < 8 && xs + i < a.W; ++i) {
            uint8_t v = ((bits >> i) & 1u) ? 255 : 0;
            if (a.or_mask) v |= a.or_mask[base + i];
            a.out[base + i] = v;
        }
    }
}

}  // namespace

// hw[]: ellipse half width per row dy = -r..r (-1 = empty row); see make_ellipse in warp_mask.hip
int ofx_mask_bits_launch(int src, const float* conf, float* log_conf, const uint8_t* image, const uint8_t* or_mask,
                         uint8_t* out, int B, int H, int W, float thres, int edge_thres, int r, const signed char* hw,
                         const char* name, hipStream_t s) {
    OFX_REQUIRE(r >= 0 && r <= kMaxR, OFX_EINVAL);
    BitArgs a{};
    a.conf = conf; a.log_conf = log_conf; a.image = image; a.or_mask = or_mask; a.out = out;
    a.H = H; a.W = W; a.thres = thres; a.edge_thres = edge_thres; a.r = r;
    for (int i = 0; i < 2 * kMaxR + 1; ++i) a.hw[i] = i < 2 * r + 1 ? hw[i] : (signed char)-1;
    OfxProfScope prof(name, s);
    if (r <= kNH) {
        dim3 gridn(ofx_cdiv(W, kNW), ofx_cdiv(H, kTH), B);
        switch (src) {
            case OFX_MSRC_CONF_LT: hipLaunchKernelGGL((mask_bits_narrow_kernel<OFX_MSRC_CONF_LT>), gridn, dim3(256), 0, s, a); break;
            case OFX_MSRC_CONF_NGT: hipLaunchKernelGGL((mask_bits_narrow_kernel<OFX_MSRC_CONF_NGT>), gridn, dim3(256), 0, s, a); break;
            case OFX_MSRC_EDGES: hipLaunchKernelGGL((mask_bits_narrow_kernel<OFX_MSRC_EDGES>), gridn, dim3(256), 0, s, a); break;
            default: return OFX_EINVAL;
        }
        return ofx_launch_status();
    }
    dim3 grid(ofx_cdiv(W, kTW), ofx_cdiv(H, kTH), B);
    switch (src) {
        case OFX_MSRC_CONF_LT: hipLaunchKernelGGL((mask_bits_kernel<OFX_MSRC_CONF_LT>), grid, dim3(256), 0, s, a); break;
        case OFX_MSRC_CONF_NGT: hipLaunchKernelGGL((mask_bits_kernel<OFX_MSRC_CONF_NGT>), grid, dim3(256), 0, s, a); break;
        case OFX_MSRC_EDGES: hipLaunchKernelGGL((mask_bits_kernel<OFX_MSRC_EDGES>), grid, dim3(256), 0, s, a); break;
        default: return OFX_EINVAL;
    }
    return ofx_launch_status();
}
