// The correlation lookup fused into the first convolution of the motion encoder:
//
//     corr  = CorrBlock.__call__(coords1)                         RAFT/core/corr.py:29-50     [M][324]
//     cor   = relu(convc1(corr))        1x1, 324 -> 256            RAFT/core/update.py:79-86   [M][256]
//
// Unfused, the 324-float row of every pixel is written by the lookup kernel (1296 B) and read back by the convolution in
// every one of the 20 refinement iterations: 1.02 GB of the two kernels' 2.8 GB of HBM traffic per 64-pair launch.  Here
// the row never exists in HBM: a workgroup owns 64 pixels and is split by ROLE,
//
//   waves 4..7  producers  per pyramid level, each wave gathers the 10x10 windows of its 16 pixels from the blocked pyramid
//                          (dense 16-byte lane loads through buffer descriptors, as corr_lookup_blocked_kernel does), blends
//                          the 81 bilinear taps and writes them as one K-chunk of the GEMM's A operand into LDS:
//                          As[buf][64 pixels][84] (81 taps + 3 zeros);
//   waves 0..3  consumers  multiply that chunk by the level's 84 x 256 slice of the weights on the fp32 matrix cores
//                          (v_mfma_f32_32x32x2_f32, each wave 64 pixels x 64 channels) while the producers fill the other
//                          buffer with the next level; bias + ReLU + the store of the 256-channel row end a tile.
//
// One s_barrier per level chunk orders the two roles (producers run exactly one chunk ahead).  The weight fragments do
// not pass through LDS: they are uploaded once in MFMA fragment order (`ofx_lookup_conv_pack`) and every consumer wave
// streams its own 64 columns from L2 straight into registers, two k-steps ahead of the multiply.  The kernel is
// persistent (one workgroup per CU walks the tiles), so a producer is already gathering the next tile's level 0 while
// the consumers finish level 3 and store.
//
// Bound: the GEMM (2 * 336 * 256 flop per pixel on the fp32 MFMA: >= 430 us per 64-pair launch), not HBM -- fusing can only hide the
// gather (265 us of HBM time) under it and drop the row's 1.02 GB round trip.
//
// MEASURED (tools/lookup_bench.py, B = 64 at 512x768): 950-990 us against 890-935 us on the same boxes for the two kernels it replaces
// (lookup 335-355 + convc1 530), results identical to 7e-7.  It does NOT win, so the RAFT executor keeps the two-kernel schedule by
// default (OFX_RAFT_FUSED_LOOKUP / OFX_FUSED_LOOKUP=1 select this kernel).  What the probes say (compile-time switches, OFX_LC_DBG):
//   * two role-split workgroups per CU (128 VGPRs per wave) cost the consumers dearly: alone they needed 710-750 us -- two MFMA waves of
//     DIFFERENT workgroups share a SIMD, progress unevenly and wait for each other at every chunk barrier.  ONE workgroup per CU (256
//     VGPRs per wave, this version): consumers alone 551 us = 118 TFLOP/s, on a par with the convolution kernel on this GEMM (123);
//   * with 256 VGPRs the producers can keep 16 items (32 block loads) per wave in flight: producers alone 362 us = the stand-alone
//     lookup's rate.  Shallower queues (4 items at 128 VGPRs) left them latency-bound at 620-800 us;
//   * together: ~950 us = the SUM of the two, not the maximum.  Producers that only gather and stage (no tap reads, no blend) are free
//     (549 us with the consumers running); the tap reads + blend + row writes alone, without any global load, cost +190 us; both
//     together +400 us.  Halving the producers' vector instructions (72 -> 32 per item: the per-tile item table below) changed nothing,
//     nor did wave priorities, accumulator order, or 16-byte stores through swapped MFMA operands (slower: scattered 16-byte pieces);
//   * i.e. on one SIMD the blend's LDS-read -> FMA -> LDS-write chains and the MFMA stream do not overlap the way two workgroups of the
//     convolution kernel overlap each other; what exactly they contend for (LDS return path, issue arbitration) the counters we can
//     reach do not separate.  A role-free variant (every wave gathers, then multiplies; four workgroups per CU) measured 1300 us, and
//     the un-fused overlap (lookup of chunk c + 1 on one stream beside convc1 of chunk c on another) nothing at all.
#include "ofx_internal.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));

constexpr int kLevels = 4, kR = 4, kTaps = 81;
constexpr int kBM = 64;                       // pixels per tile
constexpr int kLDA = 84;                      // floats per A row of one level chunk: 81 taps + 3 zeros; 84 = 20 (mod 32): the
                                              // consumers' b128 fragment reads of 8 consecutive rows cover all 32 banks once
constexpr int kN = 256;                       // output channels of convc1
constexpr int kWinRows = 16, kWinCols = 24;   // a window's 4 x 3 block slots
constexpr int kSteps = 11;                    // k-steps per level chunk: ten of 8 k, one of 4
constexpr int kWfPerWaveLevel = 10 * 2 * 64 * 4 + 2 * 64 * 2;   // fragment-ordered weights of one (consumer wave, level): 5376 floats

struct LcArgs {
    const float* pyr[kLevels];
    int hb[kLevels], wb[kLevels];
    long slice[kLevels];
    const float* coords;   // [M][2] (x, y)
    const float* wf;       // fragment-ordered weights (ofx_lookup_conv_pack)
    const float* bias;     // [256]
    float* out;            // [M][ldo]
    int ldo;
    long M;
    int ntiles;
};

// ---- producer ------------------------------------------------------------------------------------------------------------
// A wave's work on a tile is 64 items: (level 0..3) x (its 16 pixels).  The block loads of an item go out kDepth items ahead of the
// item being blended -- across level chunks and across tiles, so the gather never restarts cold behind a barrier: HBM latency under
// this access pattern is ~3 us and only the depth of that queue hides it.
#ifndef OFX_LC_DBG
#define OFX_LC_DBG 0   // compile-time probes: 4 = producers skip taps + blend, 8 = skip staging too, 16 = skip the global loads
#endif
constexpr int kDepth = 16;           // items in flight per producer wave (2 x 16-byte loads each: 128 VGPRs); divides 64
constexpr int kItems = 4 * 16;

struct ProdConst {                   // per-lane constants of the block gather (corr.hip: 12 slots x 8 pieces of 16 B in two rounds) and of the taps
    int part, bj[2], bi[2], lds_off[2], tap_off[2];
    bool slot_ok[2];
};

__device__ __forceinline__ ProdConst prod_const(int lane) {
    ProdConst c;
    c.part = lane & 7;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int slot = rr * 8 + (lane >> 3);                  // slot = bj * 3 + bi, 12 used
        c.slot_ok[rr] = slot < 12;
        c.bj[rr] = (slot * 11) >> 5;
        c.bi[rr] = slot - 3 * c.bj[rr];
        c.lds_off[rr] = (c.bj[rr] * 4 + (c.part >> 1)) * kWinCols + c.bi[rr] * 8 + ((c.part & 1) << 2);
        const int k = min(rr * 64 + lane, kTaps - 1);
        const int i = (k * 57) >> 9;                            // k / 9
        c.tap_off[rr] = (k - 9 * i) * kWinCols + i;             // i: x offset (slow), j: y offset (fast)  (corr.py:37-43)
    }
    return c;
}

// Vector work on a producer wave is paid in matrix-pipe time of the consumer wave that shares its SIMD (probes in the header), so the
// per-item arithmetic is kept to what genuinely differs per lane.  Everything that depends only on (pixel, level) -- window origin,
// block base offset, fractional weights -- is computed ONCE PER TILE, vectorised: lane L of a producer wave = item L of the tile
// (level L / 16, pixel L % 16: exactly 64 items), and an item's values are broadcast with v_readlane when its turn comes.
struct ItemTab {
    int sbase;     // byte offset of block (wy >> 2, wx >> 3) of the window in the pixel's slice of its level (may be negative / past the end)
    int wxb;       // wx >> 3: block column of the window origin (the column range is the one check the descriptor cannot do)
    int winoff;    // (wy & 3) * kWinCols + (wx & 7): where the window starts inside the staged 16 x 24 region, in floats
    float w00, w01, w10, w11;
};

__device__ __forceinline__ ItemTab item_table(const LcArgs& a, float2 cxy, int lane) {
    const int l = lane >> 4;
    const float inv = l == 0 ? 1.0f : l == 1 ? 0.5f : l == 2 ? 0.25f : 0.125f;     // exact: coords / 2**l
    const int wb = l == 0 ? a.wb[0] : l == 1 ? a.wb[1] : l == 2 ? a.wb[2] : a.wb[3];
    const float xs = cxy.x * inv, ys = cxy.y * inv;
    const bool sane = fabsf(xs) < 1.0e7f && fabsf(ys) < 1.0e7f;
    const float xf = floorf(xs), yf = floorf(ys);
    const float fx = xs - xf, fy = ys - yf;
    const int wx = sane ? (int)xf - kR : -100000, wy = sane ? (int)yf - kR : -100000;
    ItemTab t;
    t.wxb = wx >> 3;
    t.sbase = (__mul24(wy >> 2, wb) + t.wxb) << 7;
    t.winoff = (wy & 3) * kWinCols + (wx & 7);
    t.w00 = (1.f - fx) * (1.f - fy); t.w01 = fx * (1.f - fy); t.w10 = (1.f - fx) * fy; t.w11 = fx * fy;
    return t;
}

// per-lane, per-level constant of the block gather: byte offset of this lane's 16-byte piece relative to the window's first block, or
// far out of range for the lanes of round 1 that have no slot (the descriptor then returns zeros)
struct LaneConst { int off[kLevels][2]; };

__device__ __forceinline__ LaneConst lane_const(const LcArgs& a, const ProdConst& pc) {
    LaneConst c;
#pragma unroll
    for (int l = 0; l < kLevels; ++l)
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
            c.off[l][rr] = pc.slot_ok[rr] ? ((__mul24(pc.bj[rr], a.wb[l]) + pc.bi[rr]) << 7) + (pc.part << 4) : 0x40000000;
    return c;
}

__device__ __forceinline__ int bcast_i(int v, int q) { return __builtin_amdgcn_readlane(v, q); }
__device__ __forceinline__ float bcast_f(float v, int q) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), q)); }

// block loads of item q = (level l, pixel p) of the tile whose first pixel of this wave is m_w.  A block row outside the slice lands
// outside the descriptor's extent by itself (negative -> huge unsigned, too large -> past the end); only the block COLUMN needs a test.
__device__ __forceinline__ void issue_item(const LcArgs& a, const ProdConst& pc, const LaneConst& lc, const ItemTab& tab, int q, long m_w,
                                           v4i (&v)[2]) {
    const int l = q >> 4, p = q & 15;
    const long m = min(m_w + p, a.M - 1);
    const unsigned m_lo = __builtin_amdgcn_readfirstlane((unsigned)m), m_hi = __builtin_amdgcn_readfirstlane((unsigned)(m >> 32));
    const long mu = (long)(((unsigned long long)m_hi << 32) | m_lo);
    const int sbase = bcast_i(tab.sbase, q), wxb = bcast_i(tab.wxb, q);
    const int wb = a.wb[l];
    const long slice = a.slice[l];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.pyr[l] + mu * slice), (short)0, (int)(slice * 4), 0x00020000);
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const bool ok = (unsigned)(wxb + pc.bi[rr]) < (unsigned)wb;
        v[rr] = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? sbase + lc.off[l][rr] : -1, 0, 0);   // out of range -> zeros
    }
}

// window of an item (in registers) -> one of this wave's two LDS scratch windows.  LDS instructions of one wavefront execute in issue
// order, so the tap reads that follow in program order see these writes without waiting for them.
__device__ __forceinline__ void stage_window(const ProdConst& pc, const v4i (&v)[2], float* __restrict__ win) {
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
        if (pc.slot_ok[rr]) *reinterpret_cast<v4i*>(win + pc.lds_off[rr]) = v[rr];
}

struct Taps { float v[2][4]; };

// the item's 2 x 4 tap values per lane (81 bilinear footprints), read from its staged window
__device__ __forceinline__ Taps read_taps(const ProdConst& pc, const ItemTab& tab, int q, const float* __restrict__ win) {
    const float* sl = win + bcast_i(tab.winoff, q);
    Taps t;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const float* b = sl + pc.tap_off[rr];
        t.v[rr][0] = b[0]; t.v[rr][1] = b[1]; t.v[rr][2] = b[kWinCols]; t.v[rr][3] = b[kWinCols + 1];
    }
    return t;
}

// bilinear blend of the taps -> row of the A chunk
__device__ __forceinline__ void blend_taps(const ItemTab& tab, int q, const Taps& t, float* __restrict__ arow, int lane) {
    const float w00 = bcast_f(tab.w00, q), w01 = bcast_f(tab.w01, q), w10 = bcast_f(tab.w10, q), w11 = bcast_f(tab.w11, q);
    float ta[2];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        float acc = t.v[rr][0] * w00;
        acc = acc + t.v[rr][1] * w01;
        acc = acc + t.v[rr][2] * w10;
        acc = acc + t.v[rr][3] * w11;
        ta[rr] = acc;
    }
    arow[lane] = ta[0];
    if (lane < kTaps - 64) arow[64 + lane] = ta[1];
}

// ---- consumer: one level chunk (84 k) of a 64 x 64 output block per wave -----------------------------------------------
// B fragments come from global memory in fragment order, prefetched two k-steps ahead into a ring of four register sets; a
// tile has 4 x 11 = 44 steps, a multiple of four, so the ring position of every step is a compile-time constant.  A fragments are
// read from LDS one k-step ahead of their MFMAs.
struct BSet { float4 j0, j1; };

template <int L>
__device__ __forceinline__ void consume_level(const float* __restrict__ Achunk, const __amdgpu_buffer_rsrc_t wfr, f32x16 (&acc)[2][2],
                                              BSet (&ring)[4], int lane) {
    const int frow = lane & 31, fk = (lane >> 5) * 4;
    // `wfr`: buffer descriptor over this wave's weights of all four levels; level l starts l * kWfPerWaveLevel floats in.  Every load is
    // (descriptor, per-lane offset, compile-time scalar offset): no per-load address registers
    auto prefetch = [&](int gstep, BSet& dst) __attribute__((always_inline)) {
        // gstep counts k-steps from the start of THIS level; steps past its end belong to the next level (wrapping to level 0)
        const int lv = (L + gstep / kSteps) % kLevels, g = gstep % kSteps;
        const int so = lv * kWfPerWaveLevel * 4;
        if (g < 10) {
            const v4i t0 = __builtin_amdgcn_raw_buffer_load_b128(wfr, lane * 16, so + (g * 2 + 0) * 1024, 0);
            const v4i t1 = __builtin_amdgcn_raw_buffer_load_b128(wfr, lane * 16, so + (g * 2 + 1) * 1024, 0);
            dst.j0 = *reinterpret_cast<const float4*>(&t0);
            dst.j1 = *reinterpret_cast<const float4*>(&t1);
        } else {
            const v2i t0 = __builtin_amdgcn_raw_buffer_load_b64(wfr, lane * 8, so + 5120 * 4, 0);
            const v2i t1 = __builtin_amdgcn_raw_buffer_load_b64(wfr, lane * 8, so + 5120 * 4 + 512, 0);
            dst.j0 = make_float4(__int_as_float(t0.x), __int_as_float(t0.y), 0.f, 0.f);
            dst.j1 = make_float4(__int_as_float(t1.x), __int_as_float(t1.y), 0.f, 0.f);
        }
    };
    float4 fa[2][2];                                            // [step parity][row tile]
#pragma unroll
    for (int i = 0; i < 2; ++i) fa[0][i] = *reinterpret_cast<const float4*>(&Achunk[(i * 32 + frow) * kLDA + fk]);
#pragma unroll
    for (int g = 0; g < kSteps; ++g) {
        constexpr int base = (L * kSteps) % 4;
        prefetch(g + 2, ring[(base + g + 2) % 4]);
        const BSet& b = ring[(base + g) % 4];
        if (g + 1 < 10) {
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[(g + 1) & 1][i] = *reinterpret_cast<const float4*>(&Achunk[(i * 32 + frow) * kLDA + (g + 1) * 8 + fk]);
        } else if (g + 1 == 10) {
            // the last four k of the level (80 = the 81st tap, 81..83 = zeros): lane half h supplies k = 80 + 2h + s
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float2 t = *reinterpret_cast<const float2*>(&Achunk[(i * 32 + frow) * kLDA + 80 + (lane >> 5) * 2]);
                fa[0][i] = make_float4(t.x, t.y, 0.f, 0.f);
            }
        }
        // the scheduler must not sink the loads just issued down to their uses two / one steps later (left alone it does, to save
        // registers, and every step then waits out a full L2 round trip)
        __builtin_amdgcn_sched_barrier(0);
        const float4 (&f)[2] = fa[g & 1];
        // round-robin over the four accumulators: consecutive MFMAs never depend on each other
#define OFX_LC_MFMA(S)                                                                                      \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[0].S, b.j0.S, acc[0][0], 0, 0, 0);               \
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[0].S, b.j1.S, acc[0][1], 0, 0, 0);               \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[1].S, b.j0.S, acc[1][0], 0, 0, 0);               \
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[1].S, b.j1.S, acc[1][1], 0, 0, 0);
        OFX_LC_MFMA(x)
        OFX_LC_MFMA(y)
        if (g < 10) {
            OFX_LC_MFMA(z)
            OFX_LC_MFMA(w)
        }
#undef OFX_LC_MFMA
        __builtin_amdgcn_sched_barrier(0);
    }
}

__global__ __launch_bounds__(512, 2) void lookup_convc1_kernel(const LcArgs a) {
    __shared__ __attribute__((aligned(16))) float As[2][kBM * kLDA];            // 43 008 B
    __shared__ __attribute__((aligned(16))) float win[4][2][kWinRows * kWinCols];  // 12 288 B: two windows per producer wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the three padding columns of both chunks: zero once (the producers write taps 0..80 only)
    if (tid < 2 * kBM) {
        float* r = &As[tid >> 6][(tid & 63) * kLDA + kTaps];
        r[0] = 0.f; r[1] = 0.f; r[2] = 0.f;
    }
    __syncthreads();
    const int G = gridDim.x;
    if (wave >= 4) {
        // ------------------------------------------------------------------ producers
        const int pw = (wave - 4) * 16;                        // this wave's 16 pixels of the tile
        float* const w0 = win[wave - 4][0];
        float* const w1 = win[wave - 4][1];
        const ProdConst pc = prod_const(lane);
        const LaneConst lc = lane_const(a, pc);
        v4i v[kDepth][2];
        int t = blockIdx.x;                                    // (the launcher never starts more workgroups than tiles; a surplus one would
                                                               // find both roles' tile loops empty and meet no barrier)
        long m_w = (long)t * kBM + pw;
        // lane i holds the coordinates of pixel i % 16 of the wave (one load per tile, a tile ahead: a load per pixel would queue
        // behind the block loads just issued -- vector-memory results return in order -- and serialise the pipeline)
        ItemTab tab = item_table(a, reinterpret_cast<const float2*>(a.coords)[min(m_w + (lane & 15), a.M - 1)], lane);
#pragma unroll
        for (int q = 0; q < kDepth; ++q) issue_item(a, pc, lc, tab, q, m_w, v[q]);
        // software pipeline over items: [tap reads of item q] [window of item q + 1 staged] [loads of item q + 1 + kDepth issued]
        // [blend of item q].  Two LDS windows alternate, so staging never waits for the reads of the item before.
        stage_window(pc, v[0], w0);
        issue_item(a, pc, lc, tab, kDepth, m_w, v[0]);
        for (; t < a.ntiles; t += G) {
            const long m_next = (long)min(t + G, a.ntiles - 1) * kBM + pw;      // past the last tile: harmless re-reads of the last one
            const ItemTab tab_next = item_table(a, reinterpret_cast<const float2*>(a.coords)[min(m_next + (lane & 15), a.M - 1)], lane);
#pragma unroll
            for (int q = 0; q < kItems; ++q) {
                const int l = q >> 4, p = q & 15;
                Taps tp{};
                if (!(OFX_LC_DBG & 4)) tp = read_taps(pc, tab, q, (q & 1) ? w1 : w0);
                const int q1 = q + 1;                          // its window is staged now, behind the reads above
                if (!(OFX_LC_DBG & 8)) stage_window(pc, v[q1 % kDepth], (q1 & 1) ? w1 : w0);
                else asm volatile("" :: "v"(v[q1 % kDepth][0]), "v"(v[q1 % kDepth][1]));
                const int qn = q1 + kDepth;                    // the registers are free again: the item kDepth further goes out
                if (!(OFX_LC_DBG & 16)) {
                if (qn < kItems) issue_item(a, pc, lc, tab, qn, m_w, v[q1 % kDepth]);
                else issue_item(a, pc, lc, tab_next, qn - kItems, m_next, v[q1 % kDepth]);
                }
                if (!(OFX_LC_DBG & 4)) blend_taps(tab, q, tp, &As[l & 1][(pw + p) * kLDA], lane);
                if (p == 15) __syncthreads();                  // level chunk l is complete
            }
            m_w = m_next;
            tab = tab_next;
        }
    } else {
        // ------------------------------------------------------------------ consumers
        const float* wfw = a.wf + (long)__builtin_amdgcn_readfirstlane(wave) * kLevels * kWfPerWaveLevel;
        const __amdgpu_buffer_rsrc_t wfr = __builtin_amdgcn_make_buffer_rsrc((void*)wfw, (short)0, kLevels * kWfPerWaveLevel * 4, 0x00020000);
        BSet ring[4];
        {   // steps 0 and 1 of level 0 (the ring positions the first consume_level<0> expects)
            const v4i t0 = __builtin_amdgcn_raw_buffer_load_b128(wfr, lane * 16, 0, 0);
            const v4i t1 = __builtin_amdgcn_raw_buffer_load_b128(wfr, lane * 16, 1024, 0);
            const v4i t2 = __builtin_amdgcn_raw_buffer_load_b128(wfr, lane * 16, 2048, 0);
            const v4i t3 = __builtin_amdgcn_raw_buffer_load_b128(wfr, lane * 16, 3072, 0);
            ring[0].j0 = *reinterpret_cast<const float4*>(&t0);
            ring[0].j1 = *reinterpret_cast<const float4*>(&t1);
            ring[1].j0 = *reinterpret_cast<const float4*>(&t2);
            ring[1].j1 = *reinterpret_cast<const float4*>(&t3);
            ring[2] = ring[0]; ring[3] = ring[0];
        }
        const int col = lane & 31, hh = lane >> 5;
        float bias[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) bias[j] = a.bias[wave * 64 + j * 32 + col];
        for (int t = blockIdx.x; t < a.ntiles; t += G) {
            f32x16 acc[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
            __syncthreads();
            consume_level<0>(As[0], wfr, acc, ring, lane);
            __syncthreads();
            consume_level<1>(As[1], wfr, acc, ring, lane);
            __syncthreads();
            consume_level<2>(As[0], wfr, acc, ring, lane);
            __syncthreads();
            consume_level<3>(As[1], wfr, acc, ring, lane);
            // bias + ReLU + store: accumulator element e of tile (i, j) is row 32 i + 8 (e / 4) + 4 hh + e % 4, column 32 j + col
            const long m0 = (long)t * kBM;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const long m = m0 + i * 32 + (e >> 2) * 8 + hh * 4 + (e & 3);
                    if (m < a.M) {
                        float* o = a.out + m * (long)a.ldo + wave * 64 + col;
                        o[0] = fmaxf(acc[i][0][e] + bias[0], 0.f);
                        o[32] = fmaxf(acc[i][1][e] + bias[1], 0.f);
                    }
                }
        }
    }
}

}  // namespace

// number of floats of the fragment-ordered weight copy
long ofx_lookup_conv_pack_floats() { return (long)4 * kLevels * kWfPerWaveLevel; }   // 336 * 256

// w: [256][ldw] row-major (ldw >= 324), k = l * 81 + i * 9 + j as CorrBlock orders its channels -> fragment order:
// for consumer wave w (64 columns), level l:  ten k-steps of [2 column tiles][64 lanes][4 k]  +  one of [2][64][2]
int ofx_lookup_conv_pack(const float* w, int ldw, float* out) {
    OFX_REQUIRE(w && out && ldw >= kLevels * kTaps, OFX_EINVAL);
    for (int wv = 0; wv < 4; ++wv)
        for (int l = 0; l < kLevels; ++l) {
            float* base = out + ((long)wv * kLevels + l) * kWfPerWaveLevel;
            for (int g = 0; g < 10; ++g)
                for (int j = 0; j < 2; ++j)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int s = 0; s < 4; ++s) {
                            const int n = wv * 64 + j * 32 + (lane & 31);
                            const int kl = g * 8 + (lane >> 5) * 4 + s;                     // < 80
                            base[((g * 2 + j) * 64 + lane) * 4 + s] = w[(long)n * ldw + l * kTaps + kl];
                        }
            for (int j = 0; j < 2; ++j)
                for (int lane = 0; lane < 64; ++lane)
                    for (int s = 0; s < 2; ++s) {
                        const int n = wv * 64 + j * 32 + (lane & 31);
                        const int kl = 80 + (lane >> 5) * 2 + s;
                        base[5120 + (j * 64 + lane) * 2 + s] = kl < kTaps ? w[(long)n * ldw + l * kTaps + kl] : 0.f;
                    }
        }
    return 0;
}

bool ofx_lookup_conv_ok(int h, int w) {
    return (long)ofx_corr_slice_floats_l(h, w) * 4 < (1L << 31) && (h >> 3) > 0 && (w >> 3) > 0;
}

int ofx_lookup_conv_launch(const float* const* pyr, const float* coords, const float* wf, const float* bias, float* out, int ldo, int B,
                           int h, int w, hipStream_t s) {
    OFX_REQUIRE(pyr && coords && wf && bias && out && B > 0 && h > 0 && w > 0 && ldo >= kN, OFX_EINVAL);
    OFX_REQUIRE(ofx_lookup_conv_ok(h, w), OFX_EINVAL);
    OFX_REQUIRE((((uintptr_t)coords) & 7u) == 0 && ofx_aligned16(wf), OFX_EALIGN);
    LcArgs a{};
    for (int l = 0; l < kLevels; ++l) {
        OFX_REQUIRE(pyr[l] && ofx_aligned16(pyr[l]), OFX_EALIGN);
        a.pyr[l] = pyr[l];
        a.hb[l] = ((h >> l) + 3) >> 2;
        a.wb[l] = ((w >> l) + 7) >> 3;
        a.slice[l] = (long)a.hb[l] * a.wb[l] * 32;
    }
    a.coords = coords; a.wf = wf; a.bias = bias; a.out = out; a.ldo = ldo;
    a.M = (long)B * h * w;
    const long nt = (a.M + kBM - 1) / kBM;
    OFX_REQUIRE(nt < (1L << 31), OFX_EINVAL);
    a.ntiles = (int)nt;
    int cus = 256;
    {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            cus = prop.multiProcessorCount;
    }
    OfxProfScope prof("lookup_convc1", s);
    prof.flops(2.0 * (double)a.M * 336.0 * 256.0);
    const unsigned grid = (unsigned)std::min<long>(nt, (long)cus);      // persistent: ONE workgroup per CU (256 VGPRs per wave)
    hipLaunchKernelGGL(lookup_convc1_kernel, dim3(grid), dim3(512), 0, s, a);
    return ofx_launch_status();
}

extern "C" {

// host side: convc1's weight [256][324] -> the fragment order the kernel streams (out: ofx_corr_lookup_convc1_pack_floats() floats)
long ofx_corr_lookup_convc1_pack_floats(void) { return ofx_lookup_conv_pack_floats(); }
int ofx_corr_lookup_convc1_pack(const float* w_host, float* out_host) {
    OFX_REQUIRE(w_host && out_host, OFX_EINVAL);
    return ofx_lookup_conv_pack(w_host, kLevels * kTaps, out_host);
}

// CorrBlock.__call__ + relu(convc1(.)) in one kernel.  pyr: the 4-level blocked pyramid of ofx_corr_volume; wf_dev: the packed
// weights on the device; bias_dev [256]; out [B*h*w][ldo >= 256].
int ofx_corr_lookup_convc1(const float* const* pyr, const float* coords, const float* wf_dev, const float* bias_dev, float* out, int ldo,
                           int B, int h, int w, void* stream) {
    OFX_REQUIRE(pyr && coords && wf_dev && bias_dev && out, OFX_EINVAL);
    return ofx_lookup_conv_launch(pyr, coords, wf_dev, bias_dev, out, ldo, B, h, w, (hipStream_t)stream);
}

}  // extern "C"
