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
// persistent (two workgroups per CU walk the tiles), so a producer is already gathering the next tile's level 0 while
// the consumers finish level 3 and store.
//
// Bound: the GEMM (2 * 336 * 256 flop per pixel on the fp32 MFMA), not HBM -- the gather's traffic hides under it.
#include "ofx_internal.h"

#include <algorithm>
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

// ---- producer: level L of 16 pixels -> one A chunk ------------------------------------------------------------------
template <int L>
__device__ __forceinline__ void produce_level(const LcArgs& a, long m0, int pw, float* __restrict__ Achunk, float* __restrict__ win,
                                              int lane) {
    // per-lane constants of the block gather (corr.hip: 12 slots x 8 pieces of 16 B in two rounds) and of the 81 taps
    const int part = lane & 7;
    int bj[2], bi[2], lds_off[2];
    bool slot_ok[2];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int slot = rr * 8 + (lane >> 3);                  // slot = bj * 3 + bi, 12 used
        slot_ok[rr] = slot < 12;
        bj[rr] = (slot * 11) >> 5;
        bi[rr] = slot - 3 * bj[rr];
        lds_off[rr] = (bj[rr] * 4 + (part >> 1)) * kWinCols + bi[rr] * 8 + ((part & 1) << 2);
    }
    int tap_off[2];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int k = min(rr * 64 + lane, kTaps - 1);
        const int i = (k * 57) >> 9;                            // k / 9
        tap_off[rr] = (k - 9 * i) * kWinCols + i;               // i: x offset (slow), j: y offset (fast)  (corr.py:37-43)
    }
    constexpr float inv = 1.0f / (float)(1 << L);               // exact: coords / 2**l
    const int hb = a.hb[L], wb = a.wb[L];
    const long slice = a.slice[L];
    const float* __restrict__ base = a.pyr[L];

    constexpr int kDepth = 4;                                   // pixels whose block loads are in flight
    v4i v[kDepth][2];
    auto issue = [&](int p, int slot) __attribute__((always_inline)) {
        const long m = min(m0 + pw + p, a.M - 1);
        const unsigned m_lo = __builtin_amdgcn_readfirstlane((unsigned)m), m_hi = __builtin_amdgcn_readfirstlane((unsigned)(m >> 32));
        const long mu = (long)(((unsigned long long)m_hi << 32) | m_lo);
        const float2 c = reinterpret_cast<const float2*>(a.coords)[mu];
        const float xs = c.x * inv, ys = c.y * inv;
        const bool sane = fabsf(xs) < 1.0e7f && fabsf(ys) < 1.0e7f;
        const int wx = sane ? (int)floorf(xs) - kR : -100000;
        const int wy = sane ? (int)floorf(ys) - kR : -100000;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(base + mu * slice), (short)0, (int)(slice * 4), 0x00020000);
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int by = (wy >> 2) + bj[rr], bx = (wx >> 3) + bi[rr];
            const bool ok = slot_ok[rr] && (unsigned)by < (unsigned)hb && (unsigned)bx < (unsigned)wb;
            const int voff = ((__mul24(by, wb) + bx) << 7) + (part << 4);
            v[slot][rr] = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? voff : -1, 0, 0);   // out of range -> zeros
        }
    };
#pragma unroll
    for (int p = 0; p < kDepth; ++p) issue(p, p);
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        const int slot = p % kDepth;
        // the pixel's window into this wave's LDS scratch
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
            if (slot_ok[rr]) *reinterpret_cast<v4i*>(win + lds_off[rr]) = v[slot][rr];
        if (p + kDepth < 16) issue(p + kDepth, slot);           // the registers are free again: next pixel's loads go out now
        const long m = min(m0 + pw + p, a.M - 1);
        const float2 c = reinterpret_cast<const float2*>(a.coords)[m];
        const float xs = c.x * inv, ys = c.y * inv;
        const bool sane = fabsf(xs) < 1.0e7f && fabsf(ys) < 1.0e7f;
        const float xf = floorf(xs), yf = floorf(ys);
        const float fx = xs - xf, fy = ys - yf;
        const float w00 = (1.f - fx) * (1.f - fy), w01 = fx * (1.f - fy), w10 = (1.f - fx) * fy, w11 = fx * fy;
        const int wx = sane ? (int)xf - kR : -100000, wy = sane ? (int)yf - kR : -100000;
        // each wavefront owns its window: LDS operations of one wave complete in order
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const float* sl = win + (wy & 3) * kWinCols + (wx & 7);
        float* arow = Achunk + (pw + p) * kLDA;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const float* b = sl + tap_off[rr];
            const float v00 = b[0], v01 = b[1], v10 = b[kWinCols], v11 = b[kWinCols + 1];
            float acc = v00 * w00;
            acc = acc + v01 * w01;
            acc = acc + v10 * w10;
            acc = acc + v11 * w11;
            if (rr == 0 || lane < kTaps - 64) arow[rr * 64 + lane] = acc;
        }
        __builtin_amdgcn_wave_barrier();                        // the window is rewritten by the next pixel
    }
}

// ---- consumer: one level chunk (84 k) of a 64 x 64 output block per wave -----------------------------------------------
// B fragments come from global memory in fragment order, prefetched two k-steps ahead into a ring of four register sets; a
// tile has 4 x 11 = 44 steps, a multiple of four, so the ring position of every step is a compile-time constant.
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
#pragma unroll
    for (int g = 0; g < kSteps; ++g) {
        constexpr int base = (L * kSteps) % 4;
        prefetch(g + 2, ring[(base + g + 2) % 4]);
        const BSet& b = ring[(base + g) % 4];
        if (g < 10) {
            float4 fa[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const float4*>(&Achunk[(i * 32 + frow) * kLDA + g * 8 + fk]);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, b.j0.x, acc[i][0], 0, 0, 0);
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, b.j0.y, acc[i][0], 0, 0, 0);
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, b.j0.z, acc[i][0], 0, 0, 0);
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, b.j0.w, acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, b.j1.x, acc[i][1], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, b.j1.y, acc[i][1], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, b.j1.z, acc[i][1], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, b.j1.w, acc[i][1], 0, 0, 0);
            }
        } else {
            // the last four k of the level (80 = the 81st tap, 81..83 = zeros): lane half h supplies k = 80 + 2h + s
            float2 fa[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const float2*>(&Achunk[(i * 32 + frow) * kLDA + 80 + (lane >> 5) * 2]);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, b.j0.x, acc[i][0], 0, 0, 0);
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, b.j0.y, acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, b.j1.x, acc[i][1], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, b.j1.y, acc[i][1], 0, 0, 0);
            }
        }
    }
}

__global__ __launch_bounds__(512, 4) void lookup_convc1_kernel(const LcArgs a) {
    __shared__ __attribute__((aligned(16))) float As[2][kBM * kLDA];            // 43 008 B
    __shared__ __attribute__((aligned(16))) float win[4][kWinRows * kWinCols];  //  6 144 B
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
        float* w = win[wave - 4];
        for (int t = blockIdx.x; t < a.ntiles; t += G) {
            const long m0 = (long)t * kBM;
            produce_level<0>(a, m0, pw, As[0], w, lane);
            __syncthreads();
            produce_level<1>(a, m0, pw, As[1], w, lane);
            __syncthreads();
            produce_level<2>(a, m0, pw, As[0], w, lane);
            __syncthreads();
            produce_level<3>(a, m0, pw, As[1], w, lane);
            __syncthreads();
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
    const unsigned grid = (unsigned)std::min<long>(nt, 2L * cus);       // persistent: two workgroups per CU
    OfxProfScope prof("lookup_convc1", s);
    prof.flops(2.0 * (double)a.M * 336.0 * 256.0);
    hipLaunchKernelGGL(lookup_convc1_kernel, dim3(grid), dim3(512), 0, s, a);
    return ofx_launch_status();
}

extern "C" {

// CorrBlock.__call__ + relu(convc1(.)) in one kernel.  pyr: the 4-level blocked pyramid of ofx_corr_volume; w: convc1's weight
// [256][324] (OIHW with a 1x1 kernel); bias [256]; out [B*h*w][ldo >= 256].  Packs the weights on the host and uploads them for this
// call (stream-ordered): the RAFT executor keeps a packed copy instead.
int ofx_corr_lookup_convc1(const float* const* pyr, const float* coords, const float* w_host, const float* bias_dev, float* out, int ldo,
                           int B, int h, int w, void* stream) {
    OFX_REQUIRE(pyr && coords && w_host && bias_dev && out, OFX_EINVAL);
    hipStream_t s = (hipStream_t)stream;
    std::vector<float> packed((size_t)ofx_lookup_conv_pack_floats());
    int st = ofx_lookup_conv_pack(w_host, kLevels * kTaps, packed.data());
    if (st) return st;
    float* wf = nullptr;
    OFX_HIP_CHECK(hipMallocAsync((void**)&wf, packed.size() * sizeof(float), s));
    hipError_t e = hipMemcpyAsync(wf, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);          // `packed` is pageable host memory going out of scope
    if (e == hipSuccess) st = ofx_lookup_conv_launch(pyr, coords, wf, bias_dev, out, ldo, B, h, w, s);
    const hipError_t fe = hipFreeAsync(wf, s);
    if (e != hipSuccess) return (int)e;
    if (st) return st;
    return fe == hipSuccess ? 0 : (int)fe;
}

}  // extern "C"
