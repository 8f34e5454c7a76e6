// Correlation kernels of the flow path: pyramid pooling, the multi-scale bilinear lookup
// (CorrBlock.__call__) and the on-the-fly local correlation that replaces `alt_cuda_corr.forward`.
//
//   volume + pyramid   RAFT/core/corr.py:13-27,52-60  -- the GEMM itself runs on the fp32 matrix cores
//                      through the batched mode of the implicit-GEMM kernel (conv.hip); this file adds
//                      the 3-level average-pool pyramid, computed from one read of level 0 with the
//                      intermediate levels held in LDS.
//   layout             every pixel's slice of every level is stored in 4-row x 8-column BLOCKS of 128 bytes (one
//                      HBM line): element (y, x) of an h_l x w_l slice sits at
//                      ((y/4) * ceil(w_l/8) + x/8) * 32 + (y%4) * 8 + x%8, padding elements are zero.  Level 0 comes
//                      out of the GEMM in this order for free: the rows of fmap2 (the GEMM's B operand) are permuted
//                      once (ofx_corr_block_rows).
//   lookup             RAFT/core/corr.py:29-50 ; RAFT/core/utils/utils.py:57-71.  Per pixel, four 10x10 windows ->
//                      324 contiguous floats.  One wavefront per pixel: the blocks a window touches are fetched with
//                      dense 16-byte lane loads into LDS (all 81 samples of a level share one fractional offset), the
//                      324 outputs are written as one coalesced 1296-byte run.
//   local correlation  RAFT/alt_cuda_corr/correlation_kernel.cu:18-119 (forward only; the backward
//                      kernel is training-only and never reached by the reference's no_grad callers).
#include "ofx_internal.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace {

// ------------------------------------------------------------------------------------------
// pyramid levels 1..3 from level 0
// ------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ int blocked_index(int y, int x, int wb) {
    return (((y >> 2) * wb + (x >> 3)) << 5) + ((y & 3) << 3) + (x & 7);
}

struct PoolArgs {
    const float* l0;
    float* lv[4];             // lv[1..3]: outputs
    int h[4], w[4], wb[4];
    int slice[4];             // floats per pixel slice per level
    int levels;               // total pyramid levels (2..4)
    int from_l1;              // level 1 already exists (written by the volume GEMM's epilogue): start there
};

// LDS map (row-major, h x w) -> one blocked slice in HBM, padding elements written as zeros, 16-byte stores
__device__ __forceinline__ void store_blocked(const float* __restrict__ smap, float* __restrict__ dst, int h, int w, int wb, int slice) {
    for (int e = threadIdx.x * 4; e < slice; e += 256 * 4) {
        const int blk = e >> 5, off = e & 31;
        const int by = blk / wb, bx = blk - by * wb;
        const int y = (by << 2) + (off >> 3), x = (bx << 3) + (off & 7);   // x % 4 == 0: four consecutive columns
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (y < h) {
            const float* r = smap + y * w + x;
            if (x < w) v.x = r[0];
            if (x + 1 < w) v.y = r[1];
            if (x + 2 < w) v.z = r[2];
            if (x + 3 < w) v.w = r[3];
        }
        *reinterpret_cast<float4*>(dst + e) = v;
    }
}

__global__ __launch_bounds__(256) void pyramid_pool_kernel(const PoolArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* s1 = lds;                          // level 1 map, row-major
    float* s2 = s1 + a.h[1] * a.w[1];         // level 2
    float* s3 = s2 + a.h[2] * a.w[2];         // level 3
    const long p = blockIdx.x;                // source pixel (row of the volume)
    const int h1 = a.h[1], w1 = a.w[1], wb0 = a.wb[0];
    if (a.from_l1) {
        // level 1 came out of the GEMM (blocked): bring it into the row-major LDS map and go on from there
        const float* l1 = a.lv[1] + p * (long)a.slice[1];
        for (int e = threadIdx.x * 4; e < a.slice[1]; e += 256 * 4) {
            typedef float f4v __attribute__((ext_vector_type(4)));
            const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(l1 + e));   // level 1 is read once here
            const float4 v = make_float4(t.x, t.y, t.z, t.w);
            const int blk = e >> 5, off = e & 31;
            const int by = blk / a.wb[1], bx = blk - by * a.wb[1];
            const int y = (by << 2) + (off >> 3), x = (bx << 3) + (off & 7);
            if (y < h1) {
                float* r = s1 + y * w1 + x;
                if (x < w1) r[0] = v.x;
                if (x + 1 < w1) r[1] = v.y;
                if (x + 2 < w1) r[2] = v.z;
                if (x + 3 < w1) r[3] = v.w;
            }
        }
    }
    const float* src = a.l0 + p * (long)a.slice[0];
    // level 1 from ONE read of level 0: a 4x8 block pools into a 2x4 patch, so an item is (block, row pair, column half):
    // two 16-byte loads -> two outputs; four items per trip keep 8 x 16 B in flight per lane
    const int nitem = a.from_l1 ? 0 : (a.slice[0] >> 5) << 2;
    for (int i0 = 0; i0 < nitem; i0 += 4 * 256) {
        float4 t[4], u[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = min(i0 + q * 256 + (int)threadIdx.x, nitem - 1);
            const float* r = src + ((i >> 2) << 5) + (((i >> 1) & 1) << 4) + ((i & 1) << 2);
            t[q] = *reinterpret_cast<const float4*>(r);
            u[q] = *reinterpret_cast<const float4*>(r + 8);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + q * 256 + (int)threadIdx.x;
            if (i >= nitem) continue;
            const int b0 = i >> 2;
            const int by = b0 / wb0, bx = b0 - by * wb0;
            const int y1 = (by << 1) + ((i >> 1) & 1), x1 = (bx << 2) + ((i & 1) << 1);
            if (y1 >= h1) continue;
            if (x1 < w1) s1[y1 * w1 + x1] = (((t[q].x + t[q].y) + u[q].x) + u[q].y) * 0.25f;     // avg_pool2d: window sum, then /4
            if (x1 + 1 < w1) s1[y1 * w1 + x1 + 1] = (((t[q].z + t[q].w) + u[q].z) + u[q].w) * 0.25f;
        }
    }
    __syncthreads();
    if (!a.from_l1) store_blocked(s1, a.lv[1] + p * (long)a.slice[1], h1, w1, a.wb[1], a.slice[1]);
    if (a.levels < 3) return;
    const int h2 = a.h[2], w2 = a.w[2];
    for (int i = threadIdx.x; i < h2 * w2; i += 256) {
        const int y = i / w2, x = i - y * w2;
        const float* r0 = s1 + (2 * y) * w1 + 2 * x;
        const float* r1 = r0 + w1;
        s2[i] = (((r0[0] + r0[1]) + r1[0]) + r1[1]) * 0.25f;
    }
    __syncthreads();
    store_blocked(s2, a.lv[2] + p * (long)a.slice[2], h2, w2, a.wb[2], a.slice[2]);
    if (a.levels < 4) return;
    const int h3 = a.h[3], w3 = a.w[3];
    for (int i = threadIdx.x; i < h3 * w3; i += 256) {
        const int y = i / w3, x = i - y * w3;
        const float* r0 = s2 + (2 * y) * w2 + 2 * x;
        const float* r1 = r0 + w2;
        s3[i] = (((r0[0] + r0[1]) + r1[0]) + r1[1]) * 0.25f;
    }
    __syncthreads();
    store_blocked(s3, a.lv[3] + p * (long)a.slice[3], h3, w3, a.wb[3], a.slice[3]);
}

// Levels 2 and 3 from a blocked level 1, in registers (round 4).  A 4x8 block of level 1 is eight lanes' float4 (lane q: row q >> 1,
// columns (q & 1) * 4 ..): it pools into a 2x4 patch of level 2 -- two 16-byte pieces of one level-2 block -- and that patch into a
// 1x2 patch of level 3, with the partners two / one / four lanes away (row_shl DPP moves, no ds_bpermute).  No LDS, no workgroup barrier, one
// float4 per lane in, 16- and 8-byte stores out: a pure stream where the kernel above spends a workgroup with three barriers per
// source pixel.  Same additions in the same order (((a + b) + c) + d) * 0.25 as avg_pool2d's window sum.  Needs every level tiled
// by whole blocks that pool into whole patches: h1 % 16 == 0 and w1 % 32 == 0 (512x768 and 1024x1024 frames are).
struct PoolRegArgs {
    const float* l1;
    float* l2;
    float* l3;
    int wb1, wb2, wb3;        // blocks per slice row of levels 1, 2, 3
    int f4_per_slice;         // slice[1] / 4
    int slice2, slice3;
    long total;               // pixels * f4_per_slice
    unsigned wps, mag_wps;    // waves per slice (f4_per_slice / 64) and ceil(2^32 / wps)
    unsigned mag_wb1;         // ceil(2^32 / wb1)
    int pair;                 // two adjacent pieces per wave and trip
};

__global__ __launch_bounds__(256) void pyramid_pool_reg_kernel(const PoolRegArgs a) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    // a wave takes 64 consecutive float4 of one pixel's slice (a slice is a whole number of waves): the pixel and the wave's place in the
    // slice are scalar, the two small divisions go through precomputed reciprocals
    const int lane = threadIdx.x & 63;
    const unsigned nwaves = (unsigned)(a.total >> 6);
    auto body = [&](unsigned gw, const f4v v) __attribute__((always_inline)) {
        const unsigned p = __umulhi(gw, a.mag_wps);                                                   // gw / (waves per slice)
        const int e = (int)(gw - p * a.wps) * 64 + lane;
        const int b = e >> 3, q = e & 7;
        const int by = (int)__umulhi((unsigned)b, a.mag_wb1), bx = b - by * a.wb1;
        // the row below (lane + 2): rows 0 / 2 of the block pool with rows 1 / 3
        // (row_shl DPP moves: lane l reads lane l + n inside its row of 16 -- an 8-lane block never straddles a row; full-rate vector
        // moves where wave shuffles would be twelve LDS-crossbar operations per lane)
        auto up = [](float x, auto n) {
            return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x100 + decltype(n)::value, 0xF, 0xF, true));
        };
        using std::integral_constant;
        f4v d;
        d.x = up(v.x, integral_constant<int, 2>{}); d.y = up(v.y, integral_constant<int, 2>{});
        d.z = up(v.z, integral_constant<int, 2>{}); d.w = up(v.w, integral_constant<int, 2>{});
        const float s0 = (((v.x + v.y) + d.x) + d.y) * 0.25f;     // level 2, columns bx * 4 + (q & 1) * 2 + {0, 1}  (valid on lanes with an even row)
        const float s1 = (((v.z + v.w) + d.z) + d.w) * 0.25f;
        // the right half of the row pair (lane + 1): lanes q = 0 and q = 4 assemble the 2x4 patch's rows
        const float n0 = up(s0, integral_constant<int, 1>{}), n1 = up(s1, integral_constant<int, 1>{});
        // level 3 from the patch: its second row sits four lanes up
        const float u0 = up(s0, integral_constant<int, 4>{}), u1 = up(s1, integral_constant<int, 4>{});
        const float un0 = up(n0, integral_constant<int, 4>{}), un1 = up(n1, integral_constant<int, 4>{});
        if ((q & 3) == 0) {
            const int y2 = (by << 1) + (q >> 2), x2 = bx << 2;                       // level-2 position of this piece: 4 columns of one row
            const int i2 = ((((y2 >> 2) * a.wb2) + (x2 >> 3)) << 5) + ((y2 & 3) << 3) + (x2 & 7);
            *reinterpret_cast<float4*>(a.l2 + (long)p * a.slice2 + i2) = make_float4(s0, s1, n0, n1);
            if (q == 0 && a.l3) {
                const int y3 = by, x3 = bx << 1;
                const int i3 = ((((y3 >> 2) * a.wb3) + (x3 >> 3)) << 5) + ((y3 & 3) << 3) + (x3 & 7);
                const float t0 = (((s0 + s1) + u0) + u1) * 0.25f, t1 = (((n0 + n1) + un0) + un1) * 0.25f;
                *reinterpret_cast<float2*>(a.l3 + (long)p * a.slice3 + i3) = make_float2(t0, t1);
            }
        }
    };
    // one 1 KB load in flight per wave and trip (level 1 is read once, here: non-temporal); a second load a grid stride away measured
    // SLOWER (0.70 against 0.63 ms: two distant streams per wave instead of one)
    if (a.pair > 1) {
        // a.pair ADJACENT 1 KB pieces per wave and trip: that many times the bytes in flight per CU, still one stream per wave
        // (a second piece a grid stride away measured slower; adjacent ones 0.667 -> 0.608 ms)
        auto run = [&](auto n_tag) __attribute__((always_inline)) {
            constexpr int NPC = decltype(n_tag)::value;
            for (unsigned gw = NPC * __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)); gw < nwaves; gw += gridDim.x * 4 * NPC) {
                const f4v* p = reinterpret_cast<const f4v*>(a.l1) + ((long)gw * 64 + lane);
                f4v v[NPC];
#pragma unroll
                for (int i = 0; i < NPC; ++i) v[i] = (gw + i < nwaves) ? __builtin_nontemporal_load(p + 64 * i) : f4v{};
#pragma unroll
                for (int i = 0; i < NPC; ++i)
                    if (gw + i < nwaves) body(gw + i, v[i]);
            }
        };
        if (a.pair == 2) run(std::integral_constant<int, 2>{});
        else if (a.pair == 3) run(std::integral_constant<int, 3>{});
        else run(std::integral_constant<int, 4>{});
        return;
    }
    for (unsigned gw = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)); gw < nwaves; gw += gridDim.x * 4)
        body(gw, __builtin_nontemporal_load(reinterpret_cast<const f4v*>(a.l1) + ((long)gw * 64 + lane)));
}

// fmap rows (pixel order, [n][h*w][D]) -> blocked column order of the volume ([n][slice][D], zero rows for padding)
__global__ __launch_bounds__(256) void block_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int h, int w, int wb,
                                                         int slice, int D4, long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % D4);
        const long rowi = i / D4;
        const int e = (int)(rowi % slice);
        const long n = rowi / slice;
        const int blk = e >> 5, off = e & 31;
        const int by = blk / wb, bx = blk - by * wb;
        const int y = (by << 2) + (off >> 3), x = (bx << 3) + (off & 7);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (y < h && x < w) v = reinterpret_cast<const float4*>(src)[(n * h * w + (long)y * w + x) * D4 + c];
        reinterpret_cast<float4*>(dst)[i] = v;
    }
}

// ------------------------------------------------------------------------------------------
// lookup
// ------------------------------------------------------------------------------------------
constexpr int kMaxLevels = 4;
constexpr int kMaxWin = 10;   // 2r+2 with r <= 4

struct LookupArgs {
    const float* pyr[kMaxLevels];
    int hl[kMaxLevels], wl[kMaxLevels], wb[kMaxLevels];
    long slice[kMaxLevels];
    const float* coords;   // [M][2] (x, y)
    float* out;
    int ldo;
    long M;
    int levels, r;
};

// any (levels, radius): one tap per lane per trip.  The reference configuration (4 levels, radius 4) takes the block
// kernel below.
__global__ __launch_bounds__(256) void corr_lookup_generic_kernel(const LookupArgs a) {
    __shared__ float win[4][kMaxLevels * kMaxWin * kMaxWin];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long m = (long)blockIdx.x * 4 + wave;
    if (m >= a.M) return;                                       // no workgroup barrier below
    const int levels = a.levels, r = a.r;
    const int rd = 2 * r + 1, wn = rd + 1, wn2 = wn * wn;
    const float2 c = reinterpret_cast<const float2*>(a.coords)[m];
    float* s = win[wave];
    for (int t = lane; t < levels * wn2; t += 64) {
        const int l = t / wn2;
        const int rem = t - l * wn2;
        const int ty = rem / wn, tx = rem - ty * wn;
        const float inv = 1.0f / (float)(1 << l);               // exact: coords / 2**l
        const float xs = c.x * inv, ys = c.y * inv;
        float v = 0.f;
        if (fabsf(xs) < 1.0e7f && fabsf(ys) < 1.0e7f) {
            const int xx = (int)floorf(xs) - r + tx;
            const int yy = (int)floorf(ys) - r + ty;
            if ((unsigned)xx < (unsigned)a.wl[l] && (unsigned)yy < (unsigned)a.hl[l])
                v = a.pyr[l][m * a.slice[l] + blocked_index(yy, xx, a.wb[l])];
        }
        s[t] = v;
    }
    // each wavefront owns its LDS window (win[wave]): LDS operations of one wave complete in order
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float* o = a.out + m * (long)a.ldo;
    const int rd2 = rd * rd;
    for (int k = lane; k < levels * rd2; k += 64) {
        const int l = k / rd2;
        const int rem = k - l * rd2;
        const int i = rem / rd, j = rem - i * rd;               // i: x offset (slow), j: y offset (fast)
        const float inv = 1.0f / (float)(1 << l);
        const float xs = c.x * inv, ys = c.y * inv;
        const float fx = xs - floorf(xs), fy = ys - floorf(ys);
        const float* b = s + l * wn2 + j * wn + i;
        float acc = b[0] * ((1.f - fx) * (1.f - fy));
        acc = acc + b[1] * (fx * (1.f - fy));
        acc = acc + b[wn] * ((1.f - fx) * fy);
        acc = acc + b[wn + 1] * (fx * fy);
        o[k] = acc;
    }
}

// ------------------------------------------------------------------------------------------
// lookup on the BLOCKED pyramid.  HBM hands data over in 128-byte lines (profiles/r02_pmc_calibration.txt: a gather of
// 40-byte rows moves 4.1x its bytes), so a 10x10 window of a row-major slice costs ten lines per level for 400 useful
// bytes.  In the blocked layout a slice is stored as 4-row x 8-column blocks of 128 bytes = one line each; a window
// then touches (10+3)/4 x (10+7)/8 = 6.9 lines on average instead of 13, every one of them fetched with eight dense
// 16-byte lane loads, all 4 levels (48 block slots) in six rounds that are in flight together.
// ------------------------------------------------------------------------------------------
struct LookupBArgs {
    const float* pyr[kMaxLevels];
    int hb[kMaxLevels], wb[kMaxLevels];     // blocks per slice column / row
    long slice[kMaxLevels];                 // floats per pixel slice = hb * wb * 32
    const float* coords;
    float* out;
    int ldo;
    int pad;                                // zeros written behind the 324 features of a row (0 ... 47): convc1's whole chunks
    long M;
};

constexpr int kWinRows = 16, kWinCols = 24;   // 4 x 3 block slots per level

// The row-major kernel above turned out VALU-bound, not HBM-bound: ~490 vector instructions per pixel (35 of them
// quarter-rate 32-bit multiplies, 64-bit address arithmetic per tap) = 310 us of VALU time per 64-pair launch.  Here a
// wavefront walks pixels m, m + stride, ...; everything that does not depend on the pixel (which block slot and which
// 16-byte piece a lane fetches, where it lands in the LDS window, which output channel a lane owns) is computed once
// per wavefront, the pixel index and slice bases live in scalar registers, taps come through buffer descriptors whose
// range check turns an invalid block into zeros (one v_cndmask instead of a branch), and every round has a compile-time
// pyramid level: two rounds of block loads (12 slots x 8 pieces) and two rounds of outputs (81 channels) per level.
typedef int v4i __attribute__((ext_vector_type(4)));

template <int AUX>
__global__ __launch_bounds__(256) void corr_lookup_blocked_kernel(const LookupBArgs a) {
    __shared__ __attribute__((aligned(16))) float win[4][2 * kWinRows * kWinCols];
    constexpr int r = 4, rd = 9, kLvl = kWinRows * kWinCols;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* s = win[wave];
    // ---- per-lane constants
    const int part = lane & 7;
    int bj[2], bi[2], lds_off[2];
    bool slot_ok[2];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int slot = rr * 8 + (lane >> 3);                  // 0..15, 12 used: slot = bj * 3 + bi
        slot_ok[rr] = slot < 12;
        bj[rr] = (slot * 11) >> 5;                              // slot / 3 for slot < 16
        bi[rr] = slot - 3 * bj[rr];
        lds_off[rr] = (bj[rr] * 4 + (part >> 1)) * kWinCols + bi[rr] * 8 + ((part & 1) << 2);
    }
    int tap_off[2];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int k = min(rr * 64 + lane, rd * rd - 1);
        const int i = (k * 57) >> 9;                            // k / 9 for k < 81
        tap_off[rr] = (k - 9 * i) * kWinCols + i;               // i: x offset (slow), j: y offset (fast)
    }
    const long stride = (long)gridDim.x * 4;
    for (long mm = (long)blockIdx.x * 4 + wave; mm < a.M; mm += stride) {
        const unsigned m_lo = __builtin_amdgcn_readfirstlane((unsigned)mm), m_hi = __builtin_amdgcn_readfirstlane((unsigned)(mm >> 32));
        const long m = (long)(((unsigned long long)m_hi << 32) | m_lo);     // wave-uniform: addressing on the scalar unit
        const float2 c = reinterpret_cast<const float2*>(a.coords)[m];
        int wx[4], wy[4];
        float w00[4], w01[4], w10[4], w11[4];
        v4i v[8];
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const float inv = 1.0f / (float)(1 << l);           // exact: coords / 2**l
            const float xs = c.x * inv, ys = c.y * inv;
            const bool sane = fabsf(xs) < 1.0e7f && fabsf(ys) < 1.0e7f;
            const float xf = floorf(xs), yf = floorf(ys);
            const float fx = xs - xf, fy = ys - yf;
            w00[l] = (1.f - fx) * (1.f - fy);
            w01[l] = fx * (1.f - fy);
            w10[l] = (1.f - fx) * fy;
            w11[l] = fx * fy;
            wx[l] = sane ? (int)xf - r : -100000;
            wy[l] = sane ? (int)yf - r : -100000;
            const __amdgpu_buffer_rsrc_t rs =
                __builtin_amdgcn_make_buffer_rsrc((void*)(a.pyr[l] + m * a.slice[l]), (short)0, (int)(a.slice[l] * 4), 0x00020000);
            // last window row / column that carries weight: integer coordinates (the whole first iteration at level 0) need 9, not 10
            const int ex = fx > 0.f ? 9 : 8, ey = fy > 0.f ? 9 : 8;
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int by = (wy[l] >> 2) + bj[rr], bx = (wx[l] >> 3) + bi[rr];
                // only the blocks the 10x10 window overlaps: ((wy & 3) + 9) / 4 + 1 block rows, ((wx & 7) + 9) / 8 + 1 block columns
                // (3.25 x 2.125 of the 4 x 3 slots on average; round 3 fetched every in-range slot: 27.6 lines per pixel where 18.7 do).  The slots left out keep the zeros of the failed range check;
                // no tap reads them.
                const bool ok = slot_ok[rr] && bj[rr] <= (((wy[l] & 3) + ey) >> 2) && bi[rr] <= (((wx[l] & 7) + ex) >> 3) &&
                                (unsigned)by < (unsigned)a.hb[l] && (unsigned)bx < (unsigned)a.wb[l];
                const int voff = ((__mul24(by, a.wb[l]) + bx) << 7) + (part << 4);
                v[l * 2 + rr] = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? voff : -1, 0, AUX);   // out of range -> zeros
            }
        }
        float* o = a.out + m * (long)a.ldo;
        // all eight block loads are in flight; the window holds two levels at a time (3 KB per wavefront keeps eight
        // wavefronts per SIMD resident)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int l2 = 0; l2 < 2; ++l2)
#pragma unroll
                for (int rr = 0; rr < 2; ++rr)
                    if (slot_ok[rr]) *reinterpret_cast<v4i*>(s + l2 * kLvl + lds_off[rr]) = v[(half * 2 + l2) * 2 + rr];
            // each wavefront owns its LDS window: LDS operations of one wave complete in order
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int l2 = 0; l2 < 2; ++l2) {
                const int l = half * 2 + l2;
                const float* sl = s + l2 * kLvl + (wy[l] & 3) * kWinCols + (wx[l] & 7);
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    const float* b = sl + tap_off[rr];
                    const float v00 = b[0], v01 = b[1], v10 = b[kWinCols], v11 = b[kWinCols + 1];
                    float acc = v00 * w00[l];
                    acc = acc + v01 * w01[l];
                    acc = acc + v10 * w10[l];
                    acc = acc + v11 * w11[l];
                    // (the last round of the last level also writes the row's zero padding: ldo - 324 floats, convc1's whole chunks)
                    const int live = rd * rd - 64 + (l == 3 ? a.pad : 0);
                    if (rr == 0 || lane < live) o[l * (rd * rd) + rr * 64 + lane] = (rr == 0 || lane < rd * rd - 64) ? acc : 0.0f;
                }
            }
            __builtin_amdgcn_wave_barrier();                   // the window is rewritten next
        }
    }
}

// ------------------------------------------------------------------------------------------
// local correlation (alt_cuda_corr.forward semantics)
// ------------------------------------------------------------------------------------------
struct LocalArgs {
    const float* f1;      // [B,H1,W1,C]
    const float* f2;      // [B,H2,W2,C]
    const float* coords;  // [B,N,H1,W1,2]
    float* out;
    long sb, sn, sc, sp;  // output strides: batch, n, channel, pixel
    int B, H1, W1, H2, W2, C, N, r;
    float scale;
    float cscale;         // coords multiplier (1/2^level when called per pyramid level)
    long total;           // B*N*H1*W1
};

// One wavefront per output pixel.  Four taps are processed at a time: 16 lanes x (C/16) channels each,
// reduced with 4 xor-shuffles; the (2r+2)^2 tap dot products land in LDS and are then splatted
// bilinearly into the (2r+1)^2 output channels exactly as correlation_kernel.cu:92-114 does.
__global__ __launch_bounds__(256) void local_corr_kernel(const LocalArgs a) {
    __shared__ float dots[4][kMaxWin * kMaxWin];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long g = (long)blockIdx.x * 4 + wave;
    const bool live = g < a.total;
    const int rd = 2 * a.r + 1, wn = rd + 1;
    const long hw1 = (long)a.H1 * a.W1;
    long b = 0, n = 0;
    int pix = 0;
    float x = 0.f, y = 0.f;
    if (live) {
        b = g / (a.N * hw1);
        const long rem = g - b * a.N * hw1;
        n = rem / hw1;
        pix = (int)(rem - n * hw1);
        const float2 c = reinterpret_cast<const float2*>(a.coords)[g];
        x = c.x * a.cscale;
        y = c.y * a.cscale;
    }
    const bool sane = fabsf(x) < 1.0e7f && fabsf(y) < 1.0e7f;
    const int x0 = sane ? (int)floorf(x) : -100000, y0 = sane ? (int)floorf(y) : -100000;
    const int sub = lane >> 4;        // which of the 4 concurrent taps
    const int cl = lane & 15;         // channel slice
    if (live) {
        const float* f1 = a.f1 + (b * hw1 + pix) * (long)a.C;
        const float* f2b = a.f2 + b * (long)a.H2 * a.W2 * a.C;
        for (int t0 = 0; t0 < wn * wn; t0 += 4) {
            const int t = t0 + sub;
            float acc = 0.f;
            if (t < wn * wn) {
                const int iy = t / wn, ix = t - iy * wn;
                const int yy = y0 - a.r + iy, xx = x0 - a.r + ix;
                if ((unsigned)yy < (unsigned)a.H2 && (unsigned)xx < (unsigned)a.W2) {
                    const float* f2 = f2b + ((long)yy * a.W2 + xx) * a.C;
                    for (int c = cl * 4; c < a.C; c += 64) {
                        const float4 u = *reinterpret_cast<const float4*>(f1 + c);
                        const float4 v = *reinterpret_cast<const float4*>(f2 + c);
                        acc = fmaf(u.x, v.x, acc);
                        acc = fmaf(u.y, v.y, acc);
                        acc = fmaf(u.z, v.z, acc);
                        acc = fmaf(u.w, v.w, acc);
                    }
                }
            }
            acc += __shfl_xor(acc, 8, 64);
            acc += __shfl_xor(acc, 4, 64);
            acc += __shfl_xor(acc, 2, 64);
            acc += __shfl_xor(acc, 1, 64);
            if (cl == 0 && t < wn * wn) dots[wave][t] = acc;
        }
    }
    __syncthreads();
    if (!live) return;
    const float dx = sane ? x - floorf(x) : 0.f, dy = sane ? y - floorf(y) : 0.f;
    const float* s = dots[wave];
    float* o = a.out + b * a.sb + n * a.sn + (long)pix * a.sp;
    for (int k = lane; k < rd * rd; k += 64) {
        const int ox = k / rd, oy = k - ox * rd;      // channel = oy + rd*ox
        // contributions in the reference's accumulation order: se, sw, ne, nw
        float v = s[oy * wn + ox] * (1.f - dy) * (1.f - dx);
        v += s[oy * wn + ox + 1] * (1.f - dy) * dx;
        v += s[(oy + 1) * wn + ox] * dy * (1.f - dx);
        v += s[(oy + 1) * wn + ox + 1] * dy * dx;
        o[(long)k * a.sc] = v * a.scale;
    }
}

// ------------------------------------------------------------------------------------------
// local correlation, backward (alt_cuda_corr.backward, correlation_kernel.cu:122-256,288-324)
// ------------------------------------------------------------------------------------------
// One wavefront per (b, n, pixel): the adjoint of the bilinear splat turns the (2r+1)^2 output gradients into one
// weight g per integer tap (kept in LDS); then fmap1_grad[pixel] += sum_taps g * f2[tap] stays in registers
// (a lane owns C/64 channel quads; the wavefront owns the pixel row for n = 0..N-1 in sequence) and
// fmap2_grad[tap] += g * f1[pixel] goes out through float atomics, like the reference's atomicAdd.
struct LocalBwdArgs {
    const float* f1;      // [B,H1,W1,C]
    const float* f2;      // [B,H2,W2,C]
    const float* coords;  // [B,N,H1,W1,2]
    const float* gout;    // [B,N,(2r+1)^2,H1,W1]
    float* g1;            // [B,H1,W1,C]  (zero-initialised by the launcher)
    float* g2;            // [B,H2,W2,C]
    int B, H1, W1, H2, W2, C, N, r;
    long total;           // B*H1*W1
};

__global__ __launch_bounds__(256) void local_corr_bwd_kernel(const LocalBwdArgs a) {
    __shared__ float gw[4][kMaxWin * kMaxWin];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long gp = (long)blockIdx.x * 4 + wave;          // b*H1*W1 + pixel
    if (gp >= a.total) return;                             // no block-wide barriers below
    const int rd = 2 * a.r + 1, wn = rd + 1;
    const long hw1 = (long)a.H1 * a.W1;
    const long b = gp / hw1;
    const int pix = (int)(gp - b * hw1);
    const float* f1 = a.f1 + gp * (long)a.C;
    const float* f2b = a.f2 + b * (long)a.H2 * a.W2 * a.C;
    float* g2b = a.g2 + b * (long)a.H2 * a.W2 * a.C;
    float* g = gw[wave];
    for (int n = 0; n < a.N; ++n) {
        const float2 cc = reinterpret_cast<const float2*>(a.coords)[(b * a.N + n) * hw1 + pix];
        const bool sane = fabsf(cc.x) < 1.0e7f && fabsf(cc.y) < 1.0e7f;
        const int x0 = sane ? (int)floorf(cc.x) : -100000, y0 = sane ? (int)floorf(cc.y) : -100000;
        const float dx = sane ? cc.x - floorf(cc.x) : 0.f, dy = sane ? cc.y - floorf(cc.y) : 0.f;
        const float* go = a.gout + ((b * a.N + n) * (long)rd * rd) * hw1 + pix;
        // g[iy][ix]: the four splat targets of tap (iy, ix), cu:207-222 (channel = y + rd * x); all 64 lanes take part
        for (int t = lane; t < wn * wn; t += 64) {
            const int iy = t / wn, ix = t - iy * wn;
            float v = 0.f;
            if (iy > 0 && ix > 0) v += go[(long)((iy - 1) + rd * (ix - 1)) * hw1] * dy * dx;
            if (iy > 0 && ix < rd) v += go[(long)((iy - 1) + rd * ix) * hw1] * dy * (1.f - dx);
            if (iy < rd && ix > 0) v += go[(long)(iy + rd * (ix - 1)) * hw1] * (1.f - dy) * dx;
            if (iy < rd && ix < rd) v += go[(long)(iy + rd * ix) * hw1] * (1.f - dy) * (1.f - dx);
            g[t] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int c0 = lane * 4; c0 < a.C; c0 += 256) {     // channel quads owned by this lane
            const float4 u = *reinterpret_cast<const float4*>(f1 + c0);
            float4 acc1 = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int t = 0; t < wn * wn; ++t) {
                const int iy = t / wn, ix = t - iy * wn;
                const int yy = y0 - a.r + iy, xx = x0 - a.r + ix;
                if ((unsigned)yy >= (unsigned)a.H2 || (unsigned)xx >= (unsigned)a.W2) continue;   // wave-uniform
                const float gt = g[t];
                const long off = ((long)yy * a.W2 + xx) * a.C + c0;
                const float4 v = *reinterpret_cast<const float4*>(f2b + off);
                acc1.x = fmaf(gt, v.x, acc1.x); acc1.y = fmaf(gt, v.y, acc1.y);
                acc1.z = fmaf(gt, v.z, acc1.z); acc1.w = fmaf(gt, v.w, acc1.w);
                atomicAdd(g2b + off + 0, gt * u.x);
                atomicAdd(g2b + off + 1, gt * u.y);
                atomicAdd(g2b + off + 2, gt * u.z);
                atomicAdd(g2b + off + 3, gt * u.w);
            }
            // this wavefront is the only writer of its pixel's fmap1_grad row (zeroed by the launcher)
            float4* o = reinterpret_cast<float4*>(a.g1 + gp * (long)a.C + c0);
            const float4 prev = *o;
            *o = make_float4(prev.x + acc1.x, prev.y + acc1.y, prev.z + acc1.z, prev.w + acc1.w);
        }
        __builtin_amdgcn_wave_barrier();                   // g[] is rewritten by the next n
    }
}

__global__ __launch_bounds__(256) void avgpool2_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                            int H, int W, int C, long total4) {
    const int Ho = H / 2, Wo = W / 2, C4 = C / 4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total4; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C4);
        long t = idx / C4;
        const int x = (int)(t % Wo);
        t /= Wo;
        const int y = (int)(t % Ho);
        const long b = t / Ho;
        const float* p = in + ((b * H + 2 * y) * (long)W + 2 * x) * C + c * 4;
        const float4 a0 = *reinterpret_cast<const float4*>(p);
        const float4 a1 = *reinterpret_cast<const float4*>(p + C);
        const float4 a2 = *reinterpret_cast<const float4*>(p + (long)W * C);
        const float4 a3 = *reinterpret_cast<const float4*>(p + (long)W * C + C);
        float4 o;
        o.x = (((a0.x + a1.x) + a2.x) + a3.x) * 0.25f;
        o.y = (((a0.y + a1.y) + a2.y) + a3.y) * 0.25f;
        o.z = (((a0.z + a1.z) + a2.z) + a3.z) * 0.25f;
        o.w = (((a0.w + a1.w) + a2.w) + a3.w) * 0.25f;
        *reinterpret_cast<float4*>(out + idx * 4) = o;
    }
}

}  // namespace

// internal entry used by the RAFT engine (custom output layout + scale)
int ofx_local_corr_launch(const float* f1, const float* f2, const float* coords, float* out, long sb, long sn,
                          long sc, long sp, int B, int H1, int W1, int H2, int W2, int C, int N, int r, float scale,
                          float cscale, hipStream_t s) {
    LocalArgs a;
    a.f1 = f1; a.f2 = f2; a.coords = coords; a.out = out;
    a.sb = sb; a.sn = sn; a.sc = sc; a.sp = sp;
    a.B = B; a.H1 = H1; a.W1 = W1; a.H2 = H2; a.W2 = W2; a.C = C; a.N = N; a.r = r;
    a.scale = scale;
    a.cscale = cscale;
    a.total = (long)B * N * H1 * W1;
    OfxProfScope prof("local_corr", s);
    hipLaunchKernelGGL(local_corr_kernel, dim3((unsigned)((a.total + 3) / 4)), dim3(256), 0, s, a);
    return ofx_launch_status();
}

int ofx_corr_slice_floats_l(int hl, int wl) { return ((hl + 3) >> 2) * ((wl + 7) >> 3) * 32; }

// The volume GEMM can write level 1 itself when level 1 is tiled by whole 4x8 blocks and every level-0 block maps to a
// whole 2x4 patch of it
bool ofx_corr_volpool_ok(int h, int w) { return h % 8 == 0 && w % 16 == 0; }

int ofx_corr_pool_launch(const float* l0, float* l1, float* l2, float* l3, int B, int h, int w, int levels, hipStream_t s, bool from_l1) {
    PoolArgs a{};
    a.from_l1 = from_l1 ? 1 : 0;
    if (from_l1 && levels < 3) return 0;
    a.l0 = l0; a.lv[1] = l1; a.lv[2] = l2; a.lv[3] = l3;
    for (int l = 0; l < 4; ++l) {
        a.h[l] = h >> l; a.w[l] = w >> l;
        a.wb[l] = ((w >> l) + 7) >> 3;
        a.slice[l] = ofx_corr_slice_floats_l(h >> l, w >> l);
    }
    a.levels = levels;
    OFX_REQUIRE(a.h[1] > 0 && a.w[1] > 0, OFX_EINVAL);
    static const bool no_reg = getenv("OFX_POOL_LDS") != nullptr;       // diagnostic: always the LDS kernel
    // register kernel: whole blocks at every level, and wave / block-row counts inside the range in which its reciprocal
    // divisions are exact -- every other shape (tiny or very wide maps included) takes the LDS kernel below
    const long reg_f4 = a.slice[1] >> 2, reg_total = (long)B * h * w * reg_f4;
    const long reg_wps = reg_f4 >> 6;
    const bool reg_ok = from_l1 && !no_reg && a.h[1] % 16 == 0 && a.w[1] % 32 == 0 && levels >= 3 && l2 && (levels < 4 || l3) &&
                        (reg_total >> 6) < (1L << 26) && reg_wps >= 2 && reg_wps <= 64 && a.wb[1] >= 2 && a.wb[1] <= 64;
    if (reg_ok) {
        PoolRegArgs r{};
        r.l1 = l1; r.l2 = l2; r.l3 = levels >= 4 ? l3 : nullptr;
        r.wb1 = a.wb[1]; r.wb2 = a.wb[2]; r.wb3 = a.wb[3];
        r.f4_per_slice = a.slice[1] >> 2; r.slice2 = a.slice[2]; r.slice3 = a.slice[3];
        r.total = reg_total;
        r.wps = (unsigned)reg_wps;                                     // h1 % 16 == 0 and w1 % 32 == 0: the slice is a multiple of 512 floats
        r.mag_wps = r.wps <= 1 ? 0u : (unsigned)(((1ull << 32) + r.wps - 1) / r.wps);
        r.mag_wb1 = r.wb1 <= 1 ? 0u : (unsigned)(((1ull << 32) + r.wb1 - 1) / r.wb1);
        static const char* pp = getenv("OFX_POOL_PAIR");
        r.pair = pp ? atoi(pp) : 4;   // 1 / 2 / 3 / 4 pieces: 0.63 / 0.58 / 0.57 / 0.55 ms on the bench geometry
        static const char* pg = getenv("OFX_POOL_GRID");
        const long cap = pg ? atol(pg) : 256L * 64;
        OfxProfScope prof("corr_pyramid_pool", s);
        hipLaunchKernelGGL(pyramid_pool_reg_kernel, dim3((unsigned)std::min<long>((r.total + 255) / 256, cap)), dim3(256), 0, s, r);
        return ofx_launch_status();
    }
    const size_t lds = sizeof(float) * ((size_t)a.h[1] * a.w[1] + (size_t)a.h[2] * a.w[2] + (size_t)a.h[3] * a.w[3]);
    OFX_REQUIRE(lds <= 64 * 1024, OFX_EINVAL);
    OfxProfScope prof("corr_pyramid_pool", s);
    hipLaunchKernelGGL(pyramid_pool_kernel, dim3((unsigned)((long)B * h * w)), dim3(256), lds, s, a);
    return ofx_launch_status();
}

// src [n][h*w][D] (pixel order) -> dst [n][slice][D] (blocked order, zero rows for padding); D % 4 == 0
int ofx_corr_block_rows(const float* src, float* dst, int n, int h, int w, int D, hipStream_t s) {
    OFX_REQUIRE(src && dst && n > 0 && h > 0 && w > 0 && D > 0 && (D & 3) == 0, OFX_EINVAL);
    OFX_REQUIRE(ofx_aligned16(src) && ofx_aligned16(dst), OFX_EALIGN);
    const int slice = ofx_corr_slice_floats_l(h, w);
    const long total = (long)n * slice * (D >> 2);
    OfxProfScope prof("corr_block_rows", s);
    hipLaunchKernelGGL(block_rows_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 256L * 64)), dim3(256), 0, s, src, dst, h, w,
                       (w + 7) >> 3, slice, D >> 2, total);
    return ofx_launch_status();
}

extern "C" {

int ofx_corr_slice_floats(int h_l, int w_l) { return (h_l > 0 && w_l > 0) ? ofx_corr_slice_floats_l(h_l, w_l) : 0; }

int ofx_corr_volume(const float* f1, const float* f2, float* const* pyr, int B, int h, int w, int D, int levels,
                    void* stream) {
    OFX_REQUIRE(f1 && f2 && pyr && B > 0 && h > 0 && w > 0 && D > 0, OFX_EINVAL);
    OFX_REQUIRE(levels >= 1 && levels <= kMaxLevels, OFX_EINVAL);
    OFX_REQUIRE(D % 32 == 0, OFX_EALIGN);
    for (int l = 0; l < levels; ++l) OFX_REQUIRE(pyr[l] != nullptr, OFX_EINVAL);
    const long N = (long)h * w;
    const long Nb = ofx_corr_slice_floats_l(h, w);
    hipStream_t s = (hipStream_t)stream;
    const bool fused = levels >= 2 && ofx_corr_volpool_ok(h, w) && Nb % 128 == 0 && N * (long)ofx_corr_slice_floats_l(h >> 1, w >> 1) * 4 < (1L << 31) - 64;
    // the GEMM's B operand in blocked row order (stream-ordered scratch: freed behind the GEMM on the same stream)
    float* f2b = nullptr;
    OFX_HIP_CHECK(hipMallocAsync((void**)&f2b, (size_t)B * Nb * D * sizeof(float), s));
    int st = ofx_corr_block_rows(f2, f2b, B, h, w, D, s);
    if (!st) {
        // level 0: batched GEMM  vol[b] = f1[b] (N x D) * f2b[b]^T (D x Nb) / sqrt(D)
        ofx_conv_desc d{};
        d.in0 = f1; d.ld0 = D; d.c0 = D;
        d.w = f2b;
        d.out = pyr[0]; d.ldo = (int)Nb;
        d.nz = B; d.a_zs = N * D; d.w_zs = Nb * D; d.o_zs = N * Nb;
        d.B = 1; d.Hin = h; d.Win = w; d.Hout = h; d.Wout = w; d.Cout = (int)Nb;
        d.KH = 1; d.KW = 1; d.stride = 1; d.padH = 0; d.padW = 0;
        d.act = OFX_ACT_NONE; d.epi = OFX_EPI_PLAIN;
        if (fused)   // level 1 from the accumulators: level 0 is not read again
            st = ofx_conv2d_volpool(&d, 1.0f / sqrtf((float)D), pyr[1], N * ofx_corr_slice_floats_l(h >> 1, w >> 1), (w + 7) >> 3,
                                    ((w >> 1) + 7) >> 3, ofx_corr_slice_floats_l(h >> 1, w >> 1), stream);
        else
            st = ofx_conv2d_alpha(&d, 1.0f / sqrtf((float)D), stream);   // D = 256 -> exactly /16
    }
    const hipError_t fe = hipFreeAsync(f2b, s);
    if (st) return st;
    if (fe != hipSuccess) return (int)fe;
    if (levels == 1) return 0;
    return ofx_corr_pool_launch(pyr[0], pyr[1], levels > 2 ? pyr[2] : nullptr, levels > 3 ? pyr[3] : nullptr, B, h, w, levels, s, fused);
}

int ofx_corr_lookup(const float* const* pyr, const float* coords, float* out, int ldo, int B, int h, int w,
                    int levels, int radius, void* stream) {
    return ofx_corr_lookup_pad(pyr, coords, out, ldo, 0, B, h, w, levels, radius, stream);
}

}  // extern "C"

// internal (the RAFT executor): the same lookup that also writes `pad` zeros behind the features of every row -- the executor keeps
// rows of 336 floats so that convc1 walks whole 16-wide chunks
int ofx_corr_lookup_pad(const float* const* pyr, const float* coords, float* out, int ldo, int pad, int B, int h, int w,
                        int levels, int radius, void* stream) {
    OFX_REQUIRE(pyr && coords && out && B > 0 && h > 0 && w > 0, OFX_EINVAL);
    OFX_REQUIRE(pad >= 0 && pad <= 47 && (pad == 0 || (levels == 4 && radius == 4)) && ldo >= levels * (2 * radius + 1) * (2 * radius + 1) + pad, OFX_EINVAL);
    OFX_REQUIRE(levels >= 1 && levels <= kMaxLevels && radius >= 0 && 2 * radius + 2 <= kMaxWin, OFX_EINVAL);
    OFX_REQUIRE(ldo >= levels * (2 * radius + 1) * (2 * radius + 1), OFX_EINVAL);
    OFX_REQUIRE((((uintptr_t)coords) & 7u) == 0, OFX_EALIGN);
    for (int l = 0; l < levels; ++l) OFX_REQUIRE(pyr[l] != nullptr && ofx_aligned16(pyr[l]) && (h >> l) > 0 && (w >> l) > 0, OFX_EINVAL);
    hipStream_t s = (hipStream_t)stream;
    const long M = (long)B * h * w;
    OfxProfScope prof("corr_lookup", s);
    if (levels == 4 && radius == 4 && (long)ofx_corr_slice_floats_l(h, w) * 4 < (1L << 31)) {
        LookupBArgs a{};
        for (int l = 0; l < 4; ++l) {
            a.pyr[l] = pyr[l];
            a.hb[l] = ((h >> l) + 3) >> 2;
            a.wb[l] = ((w >> l) + 7) >> 3;
            a.slice[l] = (long)a.hb[l] * a.wb[l] * 32;
        }
        a.coords = coords; a.out = out; a.ldo = ldo; a.M = M; a.pad = pad;
        // window blocks are read once per iteration out of a 12.8 GB pyramid: non-temporal loads (aux bit 1 on gfx950) keep them from
        // displacing the row the kernel is writing for convc1 (tools/lookup_bench.py: 363 -> 353 us, lookup + convc1 889 -> 875 us)
        hipLaunchKernelGGL(corr_lookup_blocked_kernel<2>, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, a);
        return ofx_launch_status();
    }
    LookupArgs a{};
    for (int l = 0; l < levels; ++l) {
        a.pyr[l] = pyr[l];
        a.hl[l] = h >> l;
        a.wl[l] = w >> l;
        a.wb[l] = ((w >> l) + 7) >> 3;
        a.slice[l] = ofx_corr_slice_floats_l(h >> l, w >> l);
    }
    OFX_REQUIRE(pad == 0, OFX_EINVAL);
    a.coords = coords; a.out = out; a.ldo = ldo; a.M = M; a.levels = levels; a.r = radius;
    hipLaunchKernelGGL(corr_lookup_generic_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, a);
    return ofx_launch_status();
}

extern "C" {

int ofx_local_corr_fwd(const float* fmap1, const float* fmap2, const float* coords, float* corr, int B, int H1,
                       int W1, int H2, int W2, int C, int N, int r, void* stream) {
    OFX_REQUIRE(fmap1 && fmap2 && coords && corr, OFX_EINVAL);
    OFX_REQUIRE(B > 0 && H1 > 0 && W1 > 0 && H2 > 0 && W2 > 0 && C > 0 && N > 0, OFX_EINVAL);
    OFX_REQUIRE(r >= 0 && 2 * r + 2 <= kMaxWin, OFX_EINVAL);
    OFX_REQUIRE(C % 4 == 0 && ofx_aligned16(fmap1) && ofx_aligned16(fmap2), OFX_EALIGN);
    const long rd2 = (long)(2 * r + 1) * (2 * r + 1);
    const long hw = (long)H1 * W1;
    return ofx_local_corr_launch(fmap1, fmap2, coords, corr, (long)N * rd2 * hw, rd2 * hw, hw, 1, B, H1, W1, H2, W2, C,
                                 N, r, 1.0f, 1.0f, (hipStream_t)stream);
}

int ofx_local_corr_bwd(const float* fmap1, const float* fmap2, const float* coords, const float* corr_grad, float* fmap1_grad,
                       float* fmap2_grad, int B, int H1, int W1, int H2, int W2, int C, int N, int r, void* stream) {
    OFX_REQUIRE(fmap1 && fmap2 && coords && corr_grad && fmap1_grad && fmap2_grad, OFX_EINVAL);
    OFX_REQUIRE(B > 0 && H1 > 0 && W1 > 0 && H2 > 0 && W2 > 0 && C > 0 && N > 0, OFX_EINVAL);
    OFX_REQUIRE(r >= 0 && 2 * r + 2 <= kMaxWin, OFX_EINVAL);
    OFX_REQUIRE(C % 4 == 0 && ofx_aligned16(fmap1) && ofx_aligned16(fmap2) && ofx_aligned16(fmap1_grad) && ofx_aligned16(fmap2_grad),
                OFX_EALIGN);
    hipStream_t s = (hipStream_t)stream;
    LocalBwdArgs a;
    a.f1 = fmap1; a.f2 = fmap2; a.coords = coords; a.gout = corr_grad; a.g1 = fmap1_grad; a.g2 = fmap2_grad;
    a.B = B; a.H1 = H1; a.W1 = W1; a.H2 = H2; a.W2 = W2; a.C = C; a.N = N; a.r = r;
    a.total = (long)B * H1 * W1;
    // fresh zeros like the reference's torch::zeros (cu:303-305)
    OFX_HIP_CHECK(hipMemsetAsync(fmap1_grad, 0, (size_t)B * H1 * W1 * C * sizeof(float), s));
    OFX_HIP_CHECK(hipMemsetAsync(fmap2_grad, 0, (size_t)B * H2 * W2 * C * sizeof(float), s));
    OfxProfScope prof("local_corr_bwd", s);
    hipLaunchKernelGGL(local_corr_bwd_kernel, dim3((unsigned)((a.total + 3) / 4)), dim3(256), 0, s, a);
    return ofx_launch_status();
}

int ofx_avgpool2_nhwc(const float* in, float* out, int B, int H, int W, int C, void* stream) {
    OFX_REQUIRE(in && out && B > 0 && H > 1 && W > 1 && C > 0, OFX_EINVAL);
    OFX_REQUIRE(C % 4 == 0 && ofx_aligned16(in) && ofx_aligned16(out), OFX_EALIGN);
    const long total4 = (long)B * (H / 2) * (W / 2) * (C / 4);
    hipStream_t s = (hipStream_t)stream;
    OfxProfScope prof("avgpool2_nhwc", s);
    hipLaunchKernelGGL(avgpool2_nhwc_kernel, dim3((unsigned)std::min<long>((total4 + 255) / 256, 8192)), dim3(256), 0, s,
                       in, out, H, W, C, total4);
    return ofx_launch_status();
}

}  // extern "C"
