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
    // band kernel: the distinct half widths > 0 of the element ("planes", ascending) and, per element row, its plane
    // (-1: empty row, -2: half width 0 = the raw bitmap); nplanes < 0: too many distinct widths, direct evaluation
    signed char plane_hw[8];
    signed char plane_of[2 * kMaxR + 1];
    int nplanes;
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


// ---- full-row band variant (confidence sources, W % 4 == 0): the path of generate_mask on the hot path.
// A workgroup owns kBH output rows of ONE image at full width.  Every confidence value is fetched with 16-byte
// loads (a wavefront covers 256 pixels of a row per instruction; all of a wave's loads of a batch are in flight
// before the first compare), a lane's 4 threshold bits are merged into 32-bit words with three DPP row shifts, and
// the band's bitmap (+ r halo rows above and below, + one zero word either side of a row) lives in LDS: 70 bytes per
// row at W = 512.  The elliptical dilation is then an OR of log-step "spreads" of 64-bit windows, and the result is
// expanded to 0/255 bytes with 16-byte stores.  No horizontal halo exists (the row is whole), the vertical halo is
// r rows per band edge, and workgroup ids are laid out so that the bands of an image share an XCD (its L2 serves
// the halo rows).  Measured, B = 64 x 512x768 (tools/mask_bench.py): 7x7 27 us = 4.6 TB/s of algorithmic bytes (the
// 56-column tile kernel above: 69 us); 1x1 (no halo, no dilation) 21.5 us = 5.9 TB/s is this access pattern's floor
// (the 4:1 read:write mix of a float-in / byte-out kernel; a 1 GiB float4 read alone runs at 6.4 TB/s here, a write at
// 4.4).  PMC FETCH_SIZE says the halo rows are re-fetched past L2 (7x7: +9 % reads at 64-row bands).

template <int CTRL>
__device__ __forceinline__ unsigned dpp_or(unsigned x) {
    // x | (x of the lane CTRL-selected within the 16-lane row; 0 when that lane is outside the row)
    return x | (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xF, 0xF, true);
}

// s[q] = OR_{e = 0..n-1} v[q + e]  (right spread) in O(log n) steps
__device__ __forceinline__ unsigned long long spread_r(unsigned long long v, int n) {
    int cover = 1;
    while (cover < n) {
        const int step = min(cover, n - cover);
        v |= v >> step;
        cover += step;
    }
    return v;
}
__device__ __forceinline__ unsigned long long spread_l(unsigned long long v, int n) {
    int cover = 1;
    while (cover < n) {
        const int step = min(cover, n - cover);
        v |= v << step;
        cover += step;
    }
    return v;
}

__device__ __forceinline__ unsigned nibble_to_bytes(unsigned n) {   // 4 bits -> 4 bytes of 0x00 / 0xFF
    return (((n & 0xFu) * 0x00204081u) & 0x01010101u) * 0xFFu;
}

// NT threads per workgroup, kRowBatch (row, chunk) loads in flight per wave, kBH output rows per band
template <int SRC, int NT, int kRowBatch>
__global__ __launch_bounds__(NT) void mask_rows_kernel(const BitArgs a, const int nbands, const int B, const int xcd_map,
                                                       const int kBH) {
    extern __shared__ unsigned lds[];
    __shared__ signed char hw_s[2 * kMaxR + 2], plane_s[2 * kMaxR + 2], plane_hw_s[8];
    constexpr int NW = NT / 64;
    const int W = a.W, H = a.H, r = a.r;
    if (threadIdx.x < 2 * kMaxR + 1) {
        hw_s[threadIdx.x] = a.hw[threadIdx.x];
        plane_s[threadIdx.x] = a.plane_of[threadIdx.x];
        if (threadIdx.x < 8) plane_hw_s[threadIdx.x] = a.plane_hw[threadIdx.x];
    }
    const int nwords = (W + 31) >> 5;
    const int rs = nwords + 2;                               // row stride in words: [0 | words | 0]
    unsigned* in_bits = lds;                                 // (kBH + 2r) rows
    unsigned* out_bits = in_bits + (kBH + 2 * r) * rs;       // kBH rows, nwords each
    unsigned* planes = out_bits + kBH * nwords;              // nplanes x (kBH + 2r) rows x nwords
    // ---- which band: bands of one image stay on one XCD when there are enough images to fill all eight
    long img;
    int band;
    if (xcd_map) {
        const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
        img = (long)(k / nbands) * 8 + xcd;
        band = k % nbands;
        if (img >= B) return;
    } else {
        img = blockIdx.x / nbands;
        band = blockIdx.x % nbands;
    }
    const int y0 = band * kBH;
    const int nrows = min(kBH, H - y0) + 2 * r;              // bitmap rows of this band
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // zero the pad words (and everything else once: rows outside the image stay 0)
    for (int i = threadIdx.x; i < (kBH + 2 * r) * rs; i += NT) in_bits[i] = 0;
    __syncthreads();
    // ---- phase 1: threshold bits.  item = (row, chunk of 256 pixels); items are dealt round-robin to the waves
    const int nchunks = (W + 255) >> 8;
    const int nitems = nrows * nchunks;
    const float* cbase = a.conf + img * (long)H * W;
    float* lbase = a.log_conf ? a.log_conf + img * (long)H * W : nullptr;
    for (int it0 = wave; it0 < nitems; it0 += NW * kRowBatch) {
        float4 v[kRowBatch];
        bool ok[kRowBatch];
#pragma unroll
        for (int u = 0; u < kRowBatch; ++u) {
            const int it = it0 + NW * u;
            const int row = it / nchunks, ch = it - row * nchunks;
            const int y = y0 - r + row, x = (ch << 8) + (lane << 2);
            ok[u] = it < nitems && (unsigned)y < (unsigned)H && x < W;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok[u]) {   // the confidence map is read once: non-temporal
                typedef float f4v __attribute__((ext_vector_type(4)));
                const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(cbase + (long)y * W + x));
                v[u] = make_float4(t.x, t.y, t.z, t.w);
            }
        }
#pragma unroll
        for (int u = 0; u < kRowBatch; ++u) {
            const int it = it0 + NW * u;
            if (it >= nitems) break;                         // wave-uniform
            const int row = it / nchunks, ch = it - row * nchunks;
            unsigned n = 0;
            if (ok[u]) {
                if (SRC == OFX_MSRC_CONF_LT)
                    n = (v[u].x < a.thres ? 1u : 0u) | (v[u].y < a.thres ? 2u : 0u) | (v[u].z < a.thres ? 4u : 0u) | (v[u].w < a.thres ? 8u : 0u);
                else
                    n = (!(v[u].x > a.thres) ? 1u : 0u) | (!(v[u].y > a.thres) ? 2u : 0u) | (!(v[u].z > a.thres) ? 4u : 0u) |
                        (!(v[u].w > a.thres) ? 8u : 0u);
                if (lbase != nullptr && n != 0 && row >= r && row < nrows - r) {
                    // generate_mask's side effect on the band's own rows: log_confidence[low] = 0
                    float* lp = lbase + (long)(y0 - r + row) * W + (ch << 8) + (lane << 2);
                    if (n & 1u) lp[0] = 0.f;
                    if (n & 2u) lp[1] = 0.f;
                    if (n & 4u) lp[2] = 0.f;
                    if (n & 8u) lp[3] = 0.f;
                }
            }
            unsigned x32 = n << ((lane & 7) << 2);
            x32 = dpp_or<0x111>(x32);                        // row_shr:1
            x32 = dpp_or<0x112>(x32);                        // row_shr:2
            x32 = dpp_or<0x114>(x32);                        // row_shr:4  -> lanes 7, 15, 23, ... hold 32 pixels each
            const int word = (ch << 3) + (lane >> 3);
            if ((lane & 7) == 7 && word < nwords) in_bits[row * rs + 1 + word] = x32;
        }
    }
    __syncthreads();
    // ---- phase 2: dilation, one 32-pixel word per item
    const int orows = nrows - 2 * r;
    if (a.nplanes >= 0) {
        // (a) per bitmap row, the horizontal spread for every distinct half width of the element (incremental: the
        //     spread for width k extends the one for k - 1 by one funnel shift either side) ...
        const int np = a.nplanes;
        const int pstride = (kBH + 2 * r) * nwords;
        for (int i = threadIdx.x; i < nrows * nwords; i += NT) {
            const int row = i / nwords, j = i - row * nwords;
            const unsigned* p = in_bits + row * rs + 1 + j;
            const unsigned left = p[-1], mid = p[0], right = p[1];
            unsigned acc = mid;
            int e = 1;
            for (int k = 0; k < np; ++k) {
                const int hw = plane_hw_s[k];
                for (; e <= hw; ++e)
                    acc |= __builtin_amdgcn_alignbit(right, mid, e) | __builtin_amdgcn_alignbit(mid, left, 32 - e);
                planes[k * pstride + i] = acc;
            }
        }
        __syncthreads();
        // (b) ... then the vertical OR over the element's rows
        for (int i = threadIdx.x; i < orows * nwords; i += NT) {
            const int t = i / nwords, j = i - t * nwords;
            unsigned acc = 0;
            for (int dy = -r; dy <= r; ++dy) {
                const int pl = plane_s[dy + r];
                if (pl == -1) continue;
                acc |= pl == -2 ? in_bits[(t + r + dy) * rs + 1 + j] : planes[pl * pstride + (t + r + dy) * nwords + j];
            }
            out_bits[i] = acc;
        }
    } else {
        for (int i = threadIdx.x; i < orows * nwords; i += NT) {
            const int t = i / nwords, j = i - t * nwords;
            unsigned acc = 0;
            for (int dy = -r; dy <= r; ++dy) {
                const int hw = hw_s[dy + r];
                if (hw < 0) continue;
                const unsigned* p = in_bits + (t + r + dy) * rs + 1 + j;
                const unsigned left = p[-1], mid = p[0], right = p[1];
                const unsigned long long w1 = (unsigned long long)left | ((unsigned long long)mid << 32);    // pixels 32(j-1) ..
                const unsigned long long w2 = (unsigned long long)mid | ((unsigned long long)right << 32);   // pixels 32j ..
                acc |= (unsigned)spread_r(w2, hw + 1) | (unsigned)(spread_l(w1, hw + 1) >> 32);
            }
            out_bits[i] = acc;
        }
    }
    __syncthreads();
    // ---- phase 3: bits -> bytes
    uint8_t* obase = a.out + img * (long)H * W;
    const uint8_t* orb = a.or_mask ? a.or_mask + img * (long)H * W : nullptr;
    if ((W & 15) == 0) {
        const int per_row = W >> 4;                          // 16-pixel groups
        for (int i = threadIdx.x; i < orows * per_row; i += NT) {
            const int t = i / per_row, g = i - t * per_row;
            const unsigned bits = (out_bits[t * nwords + (g >> 1)] >> ((g & 1) << 4)) & 0xFFFFu;
            uint4 o;
            o.x = nibble_to_bytes(bits);
            o.y = nibble_to_bytes(bits >> 4);
            o.z = nibble_to_bytes(bits >> 8);
            o.w = nibble_to_bytes(bits >> 12);
            const long off = (long)(y0 + t) * W + (g << 4);
            if (orb) {
                const uint4 m = *reinterpret_cast<const uint4*>(orb + off);
                o.x |= m.x; o.y |= m.y; o.z |= m.z; o.w |= m.w;
            }
            *reinterpret_cast<uint4*>(obase + off) = o;
        }
    } else {
        const int per_row = W >> 2;                          // 4-pixel groups (W % 4 == 0)
        for (int i = threadIdx.x; i < orows * per_row; i += NT) {
            const int t = i / per_row, g = i - t * per_row;
            unsigned o = nibble_to_bytes(out_bits[t * nwords + (g >> 3)] >> ((g & 7) << 2));
            const long off = (long)(y0 + t) * W + (g << 2);
            if (orb) o |= *reinterpret_cast<const unsigned*>(orb + off);
            *reinterpret_cast<unsigned*>(obase + off) = o;
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
    if (src != OFX_MSRC_EDGES && (W & 3) == 0 && W <= 4096 && ofx_aligned16(conf) && ofx_aligned16(out) &&
        (or_mask == nullptr || ofx_aligned16(or_mask))) {
        // band height / workgroup shape, measured at B = 64, 512x768, 7x7 (tools/mask_bench.py): 64 rows x 512 threads
        // 26.7-27.5 us, 32 x 256: 28.1-29.4, 96 x 512: 28.4, 16 x 256: 29.9, 64 x 1024: 32.4 (the vertical halo is 2r
        // rows per band, so taller bands re-read less); small batches take shorter bands to keep every CU busy
        const int nwords = (W + 31) >> 5;
        int bh = 64;
        if ((long)B * ofx_cdiv(H, 64) < 512) bh = 32;
        if ((long)B * ofx_cdiv(H, 32) < 512) bh = 16;
        const int nbands = ofx_cdiv(H, bh);
        const int xcd_map = B >= 16 ? 1 : 0;
        const long nwg = xcd_map ? (long)ofx_cdiv(B, 8) * 8 * nbands : (long)B * nbands;
        OFX_REQUIRE(nwg < (1L << 31), OFX_EINVAL);
        // distinct half widths of the element -> planes
        a.nplanes = 0;
        for (int hwv = 1; hwv <= kMaxR && a.nplanes >= 0; ++hwv) {
            bool used = false;
            for (int i = 0; i < 2 * r + 1; ++i) used |= a.hw[i] == hwv;
            if (!used) continue;
            if (a.nplanes == 8) { a.nplanes = -1; break; }
            a.plane_hw[a.nplanes++] = (signed char)hwv;
        }
        for (int i = 0; i < 2 * kMaxR + 1; ++i) {
            a.plane_of[i] = -1;
            if (i < 2 * r + 1 && a.hw[i] == 0) a.plane_of[i] = -2;
            for (int k = 0; k < a.nplanes; ++k)
                if (i < 2 * r + 1 && a.hw[i] == a.plane_hw[k]) a.plane_of[i] = (signed char)k;
        }
        auto lds_words = [&](int np) { return (size_t)(bh + 2 * r) * (nwords + 2) + (size_t)bh * nwords + (size_t)np * (bh + 2 * r) * nwords; };
        if (a.nplanes > 0 && lds_words(a.nplanes) * sizeof(unsigned) > 48 * 1024) a.nplanes = -1;   // wide frames x many widths: direct
        const size_t lds = lds_words(a.nplanes > 0 ? a.nplanes : 0) * sizeof(unsigned);
        OFX_REQUIRE(lds <= 64 * 1024, OFX_EINVAL);
#define OFX_MASK_ROWS(NT, BATCH)                                                                                             \
    do {                                                                                                                     \
        if (src == OFX_MSRC_CONF_LT)                                                                                         \
            hipLaunchKernelGGL((mask_rows_kernel<OFX_MSRC_CONF_LT, NT, BATCH>), dim3((unsigned)nwg), dim3(NT), lds, s, a, nbands, B, \
                               xcd_map, bh);                                                                                 \
        else                                                                                                                 \
            hipLaunchKernelGGL((mask_rows_kernel<OFX_MSRC_CONF_NGT, NT, BATCH>), dim3((unsigned)nwg), dim3(NT), lds, s, a, nbands, B, \
                               xcd_map, bh);                                                                                 \
    } while (0)
        if (bh == 64)
            OFX_MASK_ROWS(512, 9);
        else
            OFX_MASK_ROWS(256, 8);
#undef OFX_MASK_ROWS
        return ofx_launch_status();
    }
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
