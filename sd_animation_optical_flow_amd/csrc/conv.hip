// Implicit-GEMM convolution / batched GEMM on the fp32 matrix cores of gfx950.
//
// One kernel family serves every dense contraction of the flow network:
//   * all convolutions of the RAFT encoders and update block (RAFT/core/extractor.py:118-192,
//     RAFT/core/update.py:6-136): M = B*Hout*Wout output pixels, N = Cout, K = KH*KW*Cin, with the
//     im2col gather done on the fly from NHWC activations (up to two channel segments, so that
//     torch.cat([h, x]) / cat([r*h, x]) never materialises);
//   * the all-pairs correlation volume (RAFT/core/corr.py:52-60) as a batched 1x1 "conv" whose
//     weights are the second feature map.
//
// Arithmetic (template PREC):
//   PREC = 0  exact fp32: v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate, bit-identical to an fmaf
//             chain, 157 TFLOP/s peak = the roofline of this kernel).  The reference runs fp32
//             (mixed_precision=False, ofgen_keyframe_inpaint.py:57); this is the default everywhere.
//   PREC = 3  opt-in "bf16x6": three bf16 pieces per operand (hi + mid + lo = the fp32 value exactly: 3 x 8 mantissa
//             bits), the six products of weight >= 2^-16 (hh, hm, mh, mm, hl, lh) on v_mfma_f32_32x32x16_bf16 with fp32
//             accumulation: what is dropped (ml, lm, ll) is below 2^-23 relative, i.e. fp32 rounding level, at 2.7x the
//             fp32 MFMA rate.  Not bit-identical to an fmaf chain; see DESIGN.md "Precision".
//   PREC = 4  the same arithmetic as PREC = 3 with the WEIGHTS arriving pre-split (ofx_split_conv_weight3: [hi x4 | mid x4] groups, then
//             the lo x4 groups): the B-side three-way split -- ~40 vector instructions per thread and tap -- leaves the kernel.
//   PREC = 1  opt-in "bf16x3": every operand is split at LDS-commit time into hi = bf16(x) and
//             lo = bf16(x - hi); hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16, f32 accumulate
//             (~16 mantissa bits per product; measured flow EPE ~1e-4 px after 20 iterations).
//
// Tiling (64-wide wavefronts): 256 threads = 4 waves per workgroup, a BMxBN output tile, BK = 16 or
// 32.  Both operands are staged k-contiguous in LDS with a row stride of BK+4 floats (20 / 36), which
// makes the per-lane ds_read_b128 fragment reads conflict-free; each b128 read feeds four fp32 MFMAs
// (lane half h supplies k = 8*ks + 4*h + s for s = 0..3 -- the k order inside a chunk is permuted
// identically for A and B, which leaves the dot product unchanged).  BK = 16 keeps a 128x128 tile at
// 40 KB of LDS: four workgroups (4 x 40 960 B = the CU's 160 KB, <= 128 VGPRs) share a CU; the 128x192 tile, three.
//
// Pipeline (one barrier per K-chunk, two LDS buffers, two chunks of global loads in flight):
//     MFMA block on buf[cur]  ->  commit registers (chunk kt+1) to buf[cur^1]  ->  barrier  ->
//     issue global loads of chunk kt+2
// The gather is branch-free: operands are read through buffer descriptors, conv padding / K-tail /
// M-tail lanes get an out-of-range offset and the hardware returns 0, so all of a thread's 16-byte
// loads are in flight together and have a full MFMA block to land; the steady-state loop is a single
// basic block.  The linear block id is remapped so that the N-tiles sharing an A tile run on the
// same XCD (private L2).
//
// Three A-side schedules (template MODE, chosen by the launcher): 0 = the general gather above; 1 = chunk coordinates in scalar
// registers when every chunk lies inside one tap and one segment; 2 = the halo patch -- stride-1 3x3 / 1x5 / 5x1 layers stage
// the input patch of a tile with its halo ONCE per 16-channel slab and every tap reads it at a shifted row (see the comment at
// the kernel).  MODE 2 carries 92 % of the step's FLOPs at 0.92 of the fp32 peak; 0 and 1 remain for 1x1 / strided layers.
//
// Epilogues fuse bias / folded BatchNorm, an optional per-element addend (the loop-invariant part of
// the GRU convolutions), activation, residual add, the GRU gate algebra (z, r*h, h = (1-z)h + z*q)
// and the flow/coords update, so none of those run as separate passes.
#include "ofx_internal.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace {

struct ConvK {
    const float* in0;
    const float* in1;
    const float* w;
    const float* scale;
    const float* shift;
    const float* addend;
    float* out;
    const float* res;
    const float* nmean;
    const float* nrstd;
    float* aux_z;
    float* aux_rh;
    float* aux_h;
    float* aux_coords;
    float* aux_flow4;
    long a_zs, w_zs, o_zs;
    int ld0, c0, ld1, c1, cin, ldo, ldres, ldh, ldadd;
    int Hin, Win, Hout, Wout, Cout, KW, stride, padH, padW;
    int K, Kpad, M, act;
    int mtiles, ntiles;
    int group_m;                    // > 1: tiles are ordered M-fastest inside groups of group_m M-tiles (wide-N GEMMs)
    int ksplit;                     // > 1: split-K over workgroups (small grids); blockIdx.x = tile * ksplit + split
    float* sk_part;                 // [tile][split][BM*BN] partial accumulators
    int* sk_count;                  // [tile] arrival counters (zero outside a launch)
    float alpha;
    unsigned magic_cin, magic_kw;   // ceil(2^32/d) for d = cin, KW (0 when d == 1): k/d = umulhi(k, magic)
    unsigned kw1_mask;              // all ones when KW == 1 (then tap / KW = tap), else 0
    int uk;                         // every BK chunk inside one tap and one segment: scalar chunk coordinates (MODE 1)
    int KH, patch;                  // patch: the layer qualifies for the halo-patch kernel (MODE 2)
    float* stats;                   // plain epilogue: per (M tile, wave row) and channel, (sum, sum of squares) of the stored values
    int bytes0, bytes1, bytesw;     // extents of the two input segments and of the weight matrix (per z)
    // kEpiVolPool (correlation volume): level 1 of the pyramid written from the accumulators (see the epilogue)
    float* pool_out; long pool_zs; int pool_wb0, pool_wb1, pool_slice1;
};

constexpr int kEpiPlainT = 4;   // internal: OFX_EPI_PLAIN with a sigmoid / tanh activation (own instantiation, own register budget)
constexpr int kEpiVolPool = 5;  // internal: plain store of the blocked correlation volume + its 2x2 average (pyramid level 1)
constexpr int kKAlign = 32;   // packed weights are zero-padded along K to this (a multiple of every BK)

// KS = 2 ("paired pipelines", small grids only): the workgroup has a second set of four waves that runs the
// same pipeline on its own LDS buffers over the odd K chunks while the first set takes the even ones; the
// two accumulators are added through LDS before the epilogue.  A grid of one workgroup per CU is bound by the
// latency of its two chunks in flight -- this doubles the loads in flight and the waves per SIMD without
// touching the tile shape or the epilogue.
// UK ("uniform K"): every BK-wide chunk lies inside one filter tap and one input segment (cin % BK == 0, c0 % BK == 0).  The tap,
// its (ky, kx), the segment and the channel base are then the same for the whole wave and live in scalar registers; what is left
// per staged row is two adds + two compares for the bounds test and one add for the offset -- a third of the vector instructions
// of the general path, whose issue time is paid in matrix-pipe time (conv experiments).
//
// MODE 2 ("patch"): stride-1 3x3 / 1x5 / 5x1 layers whose map is a whole number of 8x16-pixel patches.  The M tile is such a patch,
// and instead of gathering an im2col chunk per filter tap, the 16-channel slab of the patch PLUS ITS HALO (10x18 / 8x20 / 12x16
// pixels) is staged in LDS once and every tap reads its A fragments from it at a shifted row -- the same bytes feed 9 (5) chunks of
// MFMAs.  The A side then stages 2.8 float4 per thread per 9 chunks instead of 18, its global offsets are constants plus a scalar
// channel offset, and the input is fetched ~5x less often.  K is walked channel-slab-major (slab, tap) instead of tap-major: the
// same products in another summation order.  Why it matters: the rate of the general kernel follows the A bytes staged per MFMA
// (DESIGN.md section 4), which this cuts by the number of taps.
template <int BM, int BN, int WM, int WN, int EPI, bool NORM, int BK, int PREC, int KS = 1, bool SK = false, int MODE = 0>
__global__ __launch_bounds__(256 * KS, (BK == 16 && KS == 1) ? ((PREC == 3 || PREC == 4) ? 2 : (PREC == 0 && BM <= 128 && BN <= 128 && !NORM && EPI != OFX_EPI_FLOW) ? 4 : 3) : 1) void igemm_kernel(const ConvK p) {
    constexpr bool UK = MODE == 1, PATCH = MODE == 2;
    constexpr bool X6 = PREC == 3 || PREC == 4;           // three bf16 pieces per operand, six products
    // halo patch rows staged per channel slab, rounded up to whole groups of 16: 8x16 patches 12 x 16 / 10 x 18 / 8 x 20 -> 192,
    // 8x8 patches (the 64-row tile) 12 x 8 / 10 x 10 / 8 x 12 -> 112
    // (the 256-row tile: 16x16 patches, 20 x 16 / 18 x 18 / 16 x 20 -> 336)
    constexpr int kPatchRows = BM == 256 ? 336 : BM == 128 ? 192 : 112;
    constexpr int kPW = BM >= 128 ? 16 : 8;   // patch width in pixels
    constexpr int kPH = BM / kPW;             // patch height: 8, or 16 for the 256-row tile
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    constexpr int LDK = BK + 4;               // LDS row stride in floats (144 B / 80 B): conflict-free b128 fragment reads
    constexpr int QPR = BK / 4;               // float4 slots per staged row
    constexpr int RPG = 256 / QPR;            // rows staged per pass of the 256 threads
    constexpr int A_PER = BM / RPG, B_PER = (BN + RPG - 1) / RPG;
    constexpr int BN_ST = B_PER * RPG;        // B rows staged (> BN only for the 96-wide tile: the surplus rows are never read)
    static_assert(A_PER >= 1 && BM % RPG == 0, "tile too small for this BK");
    // PREC = 1 (bf16x3): every fp32 operand element is staged as two bf16 values hi = bf16(x), lo = bf16(x - hi)
    // (same 4 bytes per element); rows are BK bf16 + 16 B of padding (48 B / 80 B: conflict-free b128 reads)
    constexpr int ROWB = BK * 2 + 16;                       // bytes per staged bf16 row
    constexpr int NPC = X6 ? 3 : 2;                         // bf16 pieces per operand element
    constexpr int STAGE = PREC ? ((BM + BN_ST) * ROWB * NPC) / 4 : (BM + BN_ST) * LDK;   // floats per stage
    // BSWZ (the fp32 64-row patch tile, i.e. the small-grid schedule): the two weight stages are UNPADDED 128-byte rows whose eight
    // 16-byte slots are XOR-swizzled with (row >> 1) & 7 -- 16 consecutive rows of one slot then cover the 64 banks exactly (row parity
    // picks the half, the swizzle the quad), which is what the LDK padding buys elsewhere.  It takes 2 KB off the workgroup (34 564 ->
    // 32 516 bytes): FIVE workgroups share a CU's 160 KB instead of four, so that the 1152-workgroup layers of a single pair
    // (`convc2`: 288 tiles x 4 splits) run in one round instead of 1024 + a tail of 128 at one workgroup per CU.
#ifdef OFX_NO_BSWZ   // A/B build (tools/build_variant.sh nobswz -DOFX_NO_BSWZ): the padded weight stages, four workgroups per CU
    constexpr bool BSWZ = false;
#else
    constexpr bool BSWZ = PATCH && PREC == 0 && BM == 64 && BK == 32;
#endif
    static_assert((BM / WM) * (BN / WN) == 4, "4 waves per workgroup");
    static_assert(KS == 1 || KS == 2, "one or two pipelines");
    static_assert(KS == 1 || 2 * STAGE * KS >= 256 * TM * TN * 16, "accumulator exchange must fit the staging buffers");
    __shared__ __attribute__((aligned(16))) float smem_all[!PATCH ? 2 * STAGE * KS : BSWZ ? kPatchRows * LDK + 2 * BN_ST * BK : PREC == 0 ? (kPatchRows + 2 * BN_ST) * LDK : (kPatchRows + 2 * BN_ST) * (X6 ? 3 : 2) * (ROWB / 4)];
    const int grp = KS == 1 ? 0 : (int)(threadIdx.x >> 8);      // pipeline this thread belongs to
    float* const smem = smem_all + grp * (2 * STAGE);

    const int tid = KS == 1 ? (int)threadIdx.x : (int)(threadIdx.x & 255);
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N;
    const int wn = wave % WAVES_N;

    // XCD-aware bijective remap: hardware places block b on XCD b%8; give each XCD a contiguous
    // range of logical tiles so N-tiles sharing an A tile (and neighbouring M-tiles sharing halo
    // rows) hit the same private L2.
    // split-K: consecutive blocks are the splits of one tile (they land on one XCD through the remap below, so the
    // partial tiles they exchange stay in that XCD's L2)
    const int nblk = p.mtiles * p.ntiles * (SK ? p.ksplit : 1);
    const int bid = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    // Few N-tiles (every convolution): N fastest, the N-tiles of one A tile run back to back.  Many N-tiles (the
    // correlation volume, 48 x 48 tiles per pair): a sweep over all of N streams the whole B operand (6.3 MB)
    // through the 4 MB L2 once per M-tile, so tiles are walked M-fastest inside groups of group_m M-tiles --
    // the workgroups in flight then share a few A tiles and a few B tiles that all stay resident.
    int nt, mt;
    int split = 0;
    if (SK) {
        split = L % p.ksplit;
        const int tl = L / p.ksplit;
        nt = tl % p.ntiles;
        mt = tl / p.ntiles;
    } else if (p.group_m <= 1) {
        nt = L % p.ntiles;
        mt = L / p.ntiles;
    } else {
        const int per = p.group_m * p.ntiles;
        const int g = L / per;
        const int rem = L - g * per;
        const int gm = min(p.group_m, p.mtiles - g * p.group_m);   // the last group may be short
        nt = rem / gm;
        mt = g * p.group_m + (rem - nt * gm);
    }
    const int m0 = mt * BM;
    const int n0 = nt * BN;
    const int z = blockIdx.y;

    const float* __restrict__ in0 = p.in0 + (long)z * p.a_zs;
    const float* __restrict__ in1 = p.in1;
    const float* __restrict__ wgt = p.w + (long)z * p.w_zs;

    // ---- buffer descriptors: hardware range checking returns 0 for any offset past the extent, which
    // implements zero padding, the K tail and the ragged last N tile without address clamps or selects
    const float* in1s = in1 ? in1 : in0;
    const int bytes1s = in1 ? p.bytes1 : p.bytes0;
    const __amdgpu_buffer_rsrc_t rsrcw = __builtin_amdgcn_make_buffer_rsrc((void*)wgt, (short)0, p.bytesw, 0x00020000);
    // PREC = 4: the lo x4 groups of the pre-split weights sit behind the [hi x4 | mid x4] groups, 8 bytes per four k (half the offsets)
    const __amdgpu_buffer_rsrc_t rsrcw2 = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)wgt + (PREC == 4 ? p.bytesw : 0)), (short)0, p.bytesw >> 1, 0x00020000);
    constexpr int kOOB = 0x7FFFFFF0;

    // ---- per-thread gather coordinates for the A (im2col) tile
    const int kq = tid % QPR;   // float4 slot inside the BK-wide k chunk
    // row inside each RPG-row group.  fp32 staging: the 16 lanes that one ds_write_b128 pass serves must hit 16
    // distinct 16-byte bank groups.  With rows LDK floats apart, consecutive rows collide (row 3 of a 4-row group
    // wraps onto row 0: a third of all LDS cycles were bank conflicts, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE),
    // rows S apart with (LDK/4)*S = QPR (mod 16) do not: S = 4 for BK 16, 8 for BK 32.  The permutation only
    // changes which thread stages which row.
    constexpr int kG = 16 / QPR, kS = BK == 16 ? 4 : 8;
    const int j0 = tid / QPR;
    // bf16 staging (8-byte writes, 32 lanes per pass, rows ROWB = 2*BK + 16 bytes apart): the rows of one pass
    // must start 32 bytes apart modulo 256 -- same-parity rows for BK 16 (48-byte rows), every 4th row for BK 32
    const int r0 = PREC == 0 ? (j0 % kG) * kS + (j0 / kG) % kS + (j0 / (kG * kS)) * (kG * kS)
                   : BK == 16 ? 16 * (j0 / 16) + 2 * (j0 % 8) + ((j0 / 8) & 1)
                              : 16 * (j0 / 16) + 4 * (j0 % 4) + ((j0 / 4) & 3);
    const int HWo = p.Hout * p.Wout;
    int apix[A_PER], a_iy0[A_PER], a_ix0[A_PER], a_b[NORM ? A_PER : 1];
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
        const int m = min(m0 + r0 + RPG * i, p.M - 1);   // rows past M are computed but never stored
        const int b = m / HWo;
        const int rem = m - b * HWo;
        const int oy = rem / p.Wout;
        const int ox = rem - oy * p.Wout;
        if (NORM) a_b[i] = b;
        a_iy0[i] = oy * p.stride - p.padH;
        a_ix0[i] = ox * p.stride - p.padW;
        apix[i] = b * p.Hin * p.Win + a_iy0[i] * p.Win + a_ix0[i];   // may be negative (padding)
    }
    const int brow0 = BSWZ ? j0 : r0;   // BSWZ: a 16-lane write pass covers two whole consecutive rows (opposite parities)
    int browb[B_PER];
#pragma unroll
    for (int i = 0; i < B_PER; ++i) browb[i] = (n0 + brow0 + RPG * i) * (p.Kpad * 4) + kq * 16;   // rows past Cout fall off the extent
    // UK: byte offset of (row pixel, this thread's float4 slot) in each input segment
    int rowb0[UK ? A_PER : 1], rowb1[UK ? A_PER : 1];
    if constexpr (UK) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            rowb0[i] = apix[i] * (p.ld0 * 4) + kq * 16;
            rowb1[i] = apix[i] * (p.ld1 * 4) + kq * 16;
        }
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;


    // patch origin of this tile (MODE 2): mt enumerates the 8x16 patches of the batch in raster order
    int pt_b = 0, pt_y0 = 0, pt_x0 = 0;
    if constexpr (PATCH) {
        const int tpr = (p.Win + kPW - 1) / kPW;           // patches per image row / per image: the last ones may hang over the map
        const int tpi = ((p.Hin + kPH - 1) / kPH) * tpr;
        pt_b = mt / tpi;
        const int trem = mt - pt_b * tpi;
        const int ty = trem / tpr;
        pt_y0 = ty * kPH;
        pt_x0 = (trem - ty * tpr) * kPW;
    }

    if constexpr (PATCH) {
        static_assert(!PATCH || (PREC <= 4 && KS == 1 && ((BM == 128 && (WM == 64 || WM == 32) && BK == 16 && !SK) || (PREC == 0 && BM == 64 && WM == 32 && BK == 32) ||
                                                         (PREC == 0 && BM == 256 && WM == 64 && BK == 16 && !SK))),
                      "patch mode: 128-row tiles with 16-channel slabs (fp32 or bf16x3), or the fp32 64x64 small-grid tile with 32-channel slabs (split-K allowed)");
        static_assert(!PATCH || (EPI != OFX_EPI_FLOW && EPI != kEpiVolPool), "patch mode: plain / GRU epilogues");
        typedef int v4i __attribute__((ext_vector_type(4)));
        const int PWH = kPW + p.KW - 1;                      // halo patch width in pixels
        const int T = p.KH * p.KW;
        const int CBall = p.cin / BK;                        // channel slabs of BK channels
        // split-K: this workgroup takes the slabs [cb_lo, cb_hi) -- possibly none: it then parks a zero tile
        const int cb_per = SK ? (CBall + p.ksplit - 1) / p.ksplit : CBall;
        const int cb_lo = SK ? min(split * cb_per, CBall) : 0;
        const int CB = SK ? min(cb_lo + cb_per, CBall) : CBall;   // exclusive upper bound
        // fp32: [kPatchRows][LDK] floats, then two weight stages of [BN_ST][LDK].  bf16x3: the same rows as (hi, lo) bf16 pieces of
        // ROWB bytes each -- patch hi | patch lo | two stages of (weights hi | weights lo)
        constexpr int AROWF = PREC == 0 ? LDK : ROWB / 4;    // floats per staged row (of one piece)
        constexpr int APIECE = kPatchRows * AROWF;           // floats per A piece
        constexpr int BPIECE = BN_ST * (BSWZ ? BK : AROWF);
        float* const Apatch = smem_all;
        constexpr int NPIECE = PREC == 0 ? 1 : X6 ? 3 : 2;
        float* const Bst = smem_all + NPIECE * APIECE;
        constexpr int BSTAGE = NPIECE * BPIECE;
        // the (row, float4 slot) pairs this thread stages per slab; rows permuted like r0 (conflict-free ds_write_b128)
        constexpr int NSLOT = (kPatchRows * QPR + 255) / 256;
        constexpr int SROWS = 256 / QPR;                     // rows between a thread's consecutive slots
        // The 256-row tile keeps one pixel index per slot and forms the byte offset (pixel * row bytes of the slab's segment + the
        // thread's float4 slot) when a slab is issued -- a multiply per slot and slab instead of two resident offset tables: its six
        // slots x two segments were registers it spilled at four workgroups per CU (28 -> 12-20 bytes of scratch; since it measured level
        // at three workgroups per CU -- 131 registers, nothing spilled -- it now runs there: no kernel of the step carries scratch).  The 128-row tiles
        // keep both tables: they fit, and forming offsets late cost the GRU z|r kernel a spill of its own.
        constexpr bool LAZY_OFF = BM == 256;
        int apixs[LAZY_OFF ? NSLOT : 1];
        int avo0[LAZY_OFF ? 1 : NSLOT], avo1[LAZY_OFF ? 1 : NSLOT];
        int alds0 = 0;                                       // slot q sits q * SROWS rows below slot 0 (the row permutation keeps j / 16)
        unsigned aval = 0;
#pragma unroll
        for (int q = 0; q < NSLOT; ++q) {
            const int j = tid / QPR + SROWS * q;
            // fp32: 16-byte writes, rows 4 apart inside a group of 16; bf16 pieces: 8-byte writes, same-parity rows (see r0)
            const int row = PREC == 0 ? (j % kG) * kS + (j / kG) % kS + (j / (kG * kS)) * (kG * kS) : 16 * (j / 16) + 2 * (j % 8) + ((j / 8) & 1);
            const int hy = row / PWH, hx = row - hy * PWH;
            const int gy = pt_y0 - p.padH + hy, gx = pt_x0 - p.padW + hx;
            const bool ok = j < kPatchRows && row < (kPH + p.KH - 1) * PWH && (unsigned)gy < (unsigned)p.Hin && (unsigned)gx < (unsigned)p.Win;
            const int pix = (pt_b * p.Hin + gy) * p.Win + gx;
            if constexpr (LAZY_OFF) {
                apixs[q] = ok ? pix : 0;
            } else {
                avo0[q] = ok ? pix * (p.ld0 * 4) + kq * 16 : kOOB;
                avo1[q] = ok ? pix * (p.ld1 * 4) + kq * 16 : kOOB;
            }
            if (q == 0) alds0 = PREC == 0 ? row * LDK + kq * 4 : row * ROWB + kq * 8;   // floats / bytes
            aval |= (ok ? 1u : 0u) << q;
        }
        const int frow = lane & 31, fk = (lane >> 5) * 4;
        int afr[TM], bfr[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int r = wm * WM + i * 32 + frow;
            afr[i] = PREC == 0 ? ((r / kPW) * PWH + (r % kPW)) * LDK + fk : ((r / kPW) * PWH + (r % kPW)) * ROWB + (lane >> 5) * 16;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) bfr[j] = PREC == 0 ? (wn * WN + j * 32 + frow) * LDK + fk : (wn * WN + j * 32 + frow) * ROWB + (lane >> 5) * 16;
        // BSWZ: float offset of the fragment read of k-step ks = row * 32 + 4 * ((2 ks + half) ^ ((row >> 1) & 7))
        int bfz[BSWZ ? TN : 1][BSWZ ? BK / 8 : 1];
        if constexpr (BSWZ) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = wn * WN + j * 32 + frow;
#pragma unroll
                for (int ks = 0; ks < BK / 8; ++ks) bfz[j][ks] = n * BK + 4 * ((2 * ks + (lane >> 5)) ^ ((n >> 1) & 7));
            }
        }

        // fp32 -> (hi, lo) bf16 pair, round-to-nearest-even both times (as in the general path's commit)
        auto split_store = [&](char* hi_row, char* lo_row, float4 v) __attribute__((always_inline)) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            typedef __bf16 b4 __attribute__((ext_vector_type(4)));
            const f4 x = {v.x, v.y, v.z, v.w};
            const b4 hh = __builtin_convertvector(x, b4);
            const f4 rr = x - __builtin_convertvector(hh, f4);
            const b4 ll = __builtin_convertvector(rr, b4);
            *reinterpret_cast<b4*>(hi_row) = hh;
            *reinterpret_cast<b4*>(lo_row) = ll;
        };
        // fp32 -> (hi, mid, lo) bf16 triple (bf16x6)
        auto split_store3 = [&](char* hi_row, char* mid_row, char* lo_row, float4 v) __attribute__((always_inline)) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            typedef __bf16 b4 __attribute__((ext_vector_type(4)));
            const f4 x = {v.x, v.y, v.z, v.w};
            const b4 hh = __builtin_convertvector(x, b4);
            const f4 r1 = x - __builtin_convertvector(hh, f4);
            const b4 mm = __builtin_convertvector(r1, b4);
            const f4 r2 = r1 - __builtin_convertvector(mm, f4);
            const b4 ll = __builtin_convertvector(r2, b4);
            *reinterpret_cast<b4*>(hi_row) = hh;
            *reinterpret_cast<b4*>(mid_row) = mm;
            *reinterpret_cast<b4*>(lo_row) = ll;
        };
        float4 pa[NSLOT];
        float4 pmu = make_float4(0.f, 0.f, 0.f, 0.f), prs = pmu;
        float4 rb0 = make_float4(0.f, 0.f, 0.f, 0.f), rb1 = rb0, rb2 = rb0, rb3 = rb0;
        float2 rl0 = make_float2(0.f, 0.f), rl1 = rl0, rl2 = rl0, rl3 = rl0;      // PREC = 4: the lo pieces
        auto a_issue = [&](int cb) __attribute__((always_inline)) {
            const int c = cb * BK;
            const bool s0 = c < p.c0;                        // wave-uniform
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(s0 ? in0 : in1s), (short)0, s0 ? p.bytes0 : bytes1s, 0x00020000);
            const int so = (s0 ? c : c - p.c0) * 4;
            const int ldb = (s0 ? p.ld0 : p.ld1) * 4;        // scalar
#pragma unroll
            for (int q = 0; q < NSLOT; ++q) {
                int vo;
                if constexpr (LAZY_OFF) vo = ((aval >> q) & 1u) ? apixs[q] * ldb + kq * 16 : kOOB;
                else vo = s0 ? avo0[q] : avo1[q];
                v4i t = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, 0);
                pa[q] = *reinterpret_cast<float4*>(&t);
            }
            if (NORM) {
                pmu = *reinterpret_cast<const float4*>(p.nmean + (long)pt_b * p.c0 + c + kq * 4);
                prs = *reinterpret_cast<const float4*>(p.nrstd + (long)pt_b * p.c0 + c + kq * 4);
            }
        };
        auto a_commit = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < NSLOT; ++q) {
                if (NSLOT * 256 > kPatchRows * QPR && q == NSLOT - 1 && tid / QPR + SROWS * q >= kPatchRows) break;   // beyond the patch buffer
                float4 v = pa[q];
                if (NORM) {
                    v.x = fmaxf((v.x - pmu.x) * prs.x, 0.f);
                    v.y = fmaxf((v.y - pmu.y) * prs.y, 0.f);
                    v.z = fmaxf((v.z - pmu.z) * prs.z, 0.f);
                    v.w = fmaxf((v.w - pmu.w) * prs.w, 0.f);
                    if (!((aval >> q) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                }
                if constexpr (PREC == 0) {
                    *reinterpret_cast<float4*>(&Apatch[alds0 + q * SROWS * LDK]) = v;
                } else {
                    char* hi = reinterpret_cast<char*>(Apatch) + alds0 + q * SROWS * ROWB;
                    if constexpr (X6) split_store3(hi, hi + APIECE * 4, hi + 2 * APIECE * 4, v);
                    else split_store(hi, hi + APIECE * 4, v);
                }
            }
        };
        // weight chunks are issued two ahead of the one being multiplied: (itap, icb) is the next one to issue
        int itap = 0, icb = cb_lo;
        auto b_issue = [&]() __attribute__((always_inline)) {
            const int so = (itap * p.cin + icb * BK) * 4;
#define OFX_B_ISSUE(i) \
    if constexpr (B_PER > i) { v4i t = __builtin_amdgcn_raw_buffer_load_b128(rsrcw, browb[i], so, 0); rb##i = *reinterpret_cast<float4*>(&t); \
        if constexpr (PREC == 4) { typedef int v2i_ __attribute__((ext_vector_type(2))); v2i_ u = __builtin_amdgcn_raw_buffer_load_b64(rsrcw2, browb[i] >> 1, so >> 1, 0); rl##i = *reinterpret_cast<float2*>(&u); } }
            OFX_B_ISSUE(0) OFX_B_ISSUE(1) OFX_B_ISSUE(2) OFX_B_ISSUE(3)
#undef OFX_B_ISSUE
            if (icb + 1 < CB || itap + 1 < T) {              // past the last chunk: keep re-issuing it (never multiplied)
                if (++itap == T) { itap = 0; ++icb; }
            }
        };
        auto b_commit = [&](float* Bs) __attribute__((always_inline)) {
            if constexpr (PREC == 0) {
#define OFX_B_COMMIT(i) \
    if constexpr (B_PER > i) { \
        if constexpr (BSWZ) *reinterpret_cast<float4*>(&Bs[(brow0 + RPG * i) * BK + 4 * (kq ^ (((brow0 + RPG * i) >> 1) & 7))]) = rb##i; \
        else *reinterpret_cast<float4*>(&Bs[(r0 + RPG * i) * LDK + kq * 4]) = rb##i; \
    }
                OFX_B_COMMIT(0) OFX_B_COMMIT(1) OFX_B_COMMIT(2) OFX_B_COMMIT(3)
#undef OFX_B_COMMIT
            } else {
                char* b_hi = reinterpret_cast<char*>(Bs);
                char* b_lo = b_hi + BPIECE * 4;
                // PREC = 2: the weight matrix arrives pre-split ([hi x4 | lo x4] per 16 bytes)
#define OFX_B_COMMIT(i) \
    if constexpr (B_PER > i) { \
        const int o = (r0 + RPG * i) * ROWB + kq * 8; \
        if constexpr (PREC == 2) { \
            *reinterpret_cast<float2*>(b_hi + o) = make_float2(rb##i.x, rb##i.y); \
            *reinterpret_cast<float2*>(b_lo + o) = make_float2(rb##i.z, rb##i.w); \
        } else if constexpr (PREC == 4) { \
            *reinterpret_cast<float2*>(b_hi + o) = make_float2(rb##i.x, rb##i.y); \
            *reinterpret_cast<float2*>(b_lo + o) = make_float2(rb##i.z, rb##i.w); \
            *reinterpret_cast<float2*>(b_lo + BPIECE * 4 + o) = rl##i; \
        } else if constexpr (PREC == 3) { \
            split_store3(b_hi + o, b_lo + o, b_lo + BPIECE * 4 + o, rb##i); \
        } else { \
            split_store(b_hi + o, b_lo + o, rb##i); \
        } \
    }
                OFX_B_COMMIT(0) OFX_B_COMMIT(1) OFX_B_COMMIT(2) OFX_B_COMMIT(3)
#undef OFX_B_COMMIT
            }
        };

        auto step = [&](int c, int ky, int kx) __attribute__((always_inline)) {
            if constexpr (X6) {
                // bf16x6 on the patch: pieces (hi, mid, lo) in that order; the six products of weight >= 2^-16, smallest first
                typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
                const char* a0 = reinterpret_cast<const char*>(Apatch) + (ky * PWH + kx) * ROWB;
                const char* b0 = reinterpret_cast<const char*>(Bst + (c & 1) * BSTAGE);
                bf16x8 ah[TM], am[TM], al[TM], bh[TN], bm_[TN], bl[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    ah[i] = *reinterpret_cast<const bf16x8*>(a0 + afr[i]);
                    am[i] = *reinterpret_cast<const bf16x8*>(a0 + APIECE * 4 + afr[i]);
                    al[i] = *reinterpret_cast<const bf16x8*>(a0 + 2 * APIECE * 4 + afr[i]);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    bh[j] = *reinterpret_cast<const bf16x8*>(b0 + bfr[j]);
                    bm_[j] = *reinterpret_cast<const bf16x8*>(b0 + BPIECE * 4 + bfr[j]);
                    bl[j] = *reinterpret_cast<const bf16x8*>(b0 + 2 * BPIECE * 4 + bfr[j]);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bm_[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bm_[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    }
            } else if constexpr (PREC != 0) {
                // bf16x3 on the patch: acc += a_hi*b_hi + a_hi*b_lo + a_lo*b_hi, one 16-wide k-step per tap
                typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
                const char* a_hi = reinterpret_cast<const char*>(Apatch) + (ky * PWH + kx) * ROWB;
                const char* a_lo = a_hi + APIECE * 4;
                const char* b_hi = reinterpret_cast<const char*>(Bst + (c & 1) * BSTAGE);
                const char* b_lo = b_hi + BPIECE * 4;
                bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    ah[i] = *reinterpret_cast<const bf16x8*>(a_hi + afr[i]);
                    al[i] = *reinterpret_cast<const bf16x8*>(a_lo + afr[i]);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    bh[j] = *reinterpret_cast<const bf16x8*>(b_hi + bfr[j]);
                    bl[j] = *reinterpret_cast<const bf16x8*>(b_lo + bfr[j]);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    }
            } else {
            const float* As = Apatch + (ky * PWH + kx) * LDK;
            const float* Bs = Bst + (c & 1) * BSTAGE;
#pragma unroll
            for (int ks = 0; ks < BK / 8; ++ks) {
                float4 fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const float4*>(&As[afr[i] + ks * 8]);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const float4*>(&Bs[BSWZ ? bfz[BSWZ ? j : 0][BSWZ ? ks : 0] : bfr[j] + ks * 8]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
                    }
            }
            }
            b_commit(Bst + ((c + 1) & 1) * BSTAGE);          // chunk c + 1 has had this MFMA block to land
            __syncthreads();
            b_issue();                                       // chunk c + 2
            __builtin_amdgcn_sched_barrier(0);
        };

        if (cb_lo < CB) {
        a_issue(cb_lo);
        b_issue();
        a_commit();
        b_commit(Bst);
        __syncthreads();
        b_issue();
        // Per slab: T - 1 plain steps, then the last tap peeled -- the next slab's three loads per thread are issued in front of it
        // and committed behind it, so their registers are live for one step only (held across the whole slab they pushed the
        // 128x128 tile into scratch).  The barrier that ends the last step also says every wave is done reading the patch.
        int c = 0;
        for (int cb = cb_lo; cb < CB; ++cb) {
            int ky = 0, kx = 0;
            for (int tap = 0; tap < T - 1; ++tap, ++c) {
                step(c, ky, kx);
                if (++kx == p.KW) { kx = 0; ++ky; }
            }
            const bool more = cb + 1 < CB;                   // wave-uniform
            if (more) a_issue(cb + 1);
            step(c, ky, kx);
            ++c;
            if (more) {
                a_commit();
                __syncthreads();
            }
        }
        }   // cb_lo < CB
    } else {
    // chunks of this pipeline: grp, grp + KS, ...  (an index past the end addresses k >= K: the A operand
    // reads as zero there, so the surplus iteration of the odd pipeline adds nothing)
    // split-K: this workgroup takes the chunks [kbase, kbase + nk_s) of the K axis
    const int nk_all = (p.K + BK - 1) / BK;       // chunks that hold real k (Kpad rounds K up to 32: up to one all-zero chunk more)
    const int per_split = SK ? (nk_all + p.ksplit - 1) / p.ksplit : nk_all;
    const int kbase = split * per_split;
    const int nk_s = max(0, min(per_split, nk_all - kbase));
    const int nk = (nk_s + KS - 1) / KS;
    const int frag_row = lane & 31;
    const int frag_k = (lane >> 5) * 4;

    // ---- pipeline registers (chunk in flight between `issue` and `commit`)
    float4 ra[A_PER];
    float4 rb0 = make_float4(0.f, 0.f, 0.f, 0.f), rb1 = rb0, rb2 = rb0, rb3 = rb0;   // named scalars: see OFX_B_* below
    float2 rl0 = make_float2(0.f, 0.f), rl1 = rl0, rl2 = rl0, rl3 = rl0;             // PREC = 4: the lo pieces of the pre-split weights
    float4 rmu[NORM ? A_PER : 1], rrs[NORM ? A_PER : 1];
    unsigned okbits = 0;
    int voffa[A_PER], voffb[B_PER];
    int cch_next = 0;
    bool seg0_next = true;

    // ---- pipeline stages (force-inlined; every staging array is indexed with compile-time constants)
    auto offsets = [&](int chunk) __attribute__((always_inline)) {
        // byte offsets of chunk `chunk`: pure VALU, no memory access -> the scheduler interleaves it
        // with the MFMA block that follows it in the steady-state loop body
        if constexpr (UK) {
            // wave-uniform chunk coordinates: scalar arithmetic
            const int k0 = __builtin_amdgcn_readfirstlane(((kbase + chunk * KS) + grp) * BK);
            const int tap = (int)__umulhi((unsigned)k0, p.magic_cin);
            const int cch0 = k0 - tap * p.cin;
            const int ky = (int)(__umulhi((unsigned)tap, p.magic_kw) + ((unsigned)tap & p.kw1_mask));
            const int kx = tap - ky * p.KW;
            const bool seg0 = cch0 < p.c0;
            const int common = (ky * p.Win + kx) * ((seg0 ? p.ld0 : p.ld1) * 4) + (seg0 ? cch0 : cch0 - p.c0) * 4;
            const bool kval = k0 < p.K;
            unsigned bits = 0;
#pragma unroll
            for (int i = 0; i < A_PER; ++i) {
                const int iy = a_iy0[i] + ky;
                const int ix = a_ix0[i] + kx;
                const bool in = kval & ((unsigned)iy < (unsigned)p.Hin) & ((unsigned)ix < (unsigned)p.Win);
                bits |= (in ? 1u : 0u) << i;
                voffa[i] = ((seg0 ? rowb0[i] : rowb1[i]) + common) | (in ? 0 : kOOB);
            }
#pragma unroll
            for (int i = 0; i < B_PER; ++i) voffb[i] = browb[i] + k0 * 4;
            if (NORM) okbits = (okbits & 0xFFFFu) | (bits << 16);
            cch_next = cch0 + kq * 4;
            seg0_next = seg0;
            return;
        }
        const int k0 = ((kbase + chunk * KS) + grp) * BK;
        const int k = k0 + kq * 4;
        const int tap = (int)__umulhi((unsigned)k, p.magic_cin);
        const int cch = k - tap * p.cin;
        const int ky = (int)(__umulhi((unsigned)tap, p.magic_kw) + ((unsigned)tap & p.kw1_mask));   // branch-free tap / KW
        const int kx = tap - ky * p.KW;
        const bool seg0 = cch < p.c0;
        const int doff = ky * p.Win + kx;
        // one multiply per row instead of two per-segment row tables (a select between two arrays
        // would force both into scratch memory)
        const int ldb = (seg0 ? p.ld0 : p.ld1) * 4;
        const int common = doff * ldb + (seg0 ? cch : cch - p.c0) * 4;
        const bool kval = k < p.K;
        unsigned bits = 0;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int iy = a_iy0[i] + ky;
            const int ix = a_ix0[i] + kx;
            // branch-free: the offset is always computed, out-of-image taps get the OOB bits OR-ed in
            // (a ?: here makes the compiler emit exec-masked branches that split the MFMA basic block)
            const bool in = kval & ((unsigned)iy < (unsigned)p.Hin) & ((unsigned)ix < (unsigned)p.Win);
            bits |= (in ? 1u : 0u) << i;
            voffa[i] = (apix[i] * ldb + common) | (in ? 0 : kOOB);
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) voffb[i] = browb[i] + k0 * 4;
        if (NORM) okbits = (okbits & 0xFFFFu) | (bits << 16);   // [31:16] = chunk being addressed
        cch_next = cch;
        seg0_next = seg0;
    };

    auto issue = [&]() __attribute__((always_inline)) {
        // 1 VGPR offset per 16-byte load, all 8 in flight together
        typedef int v4i __attribute__((ext_vector_type(4)));
        // the segment is uniform per chunk (c0 % 32 == 0): pick the descriptor with scalar selects, no branch
        const bool seg_u = __builtin_amdgcn_readfirstlane(seg0_next ? 1 : 0) != 0;
        const __amdgpu_buffer_rsrc_t rsrca =
            __builtin_amdgcn_make_buffer_rsrc((void*)(seg_u ? in0 : in1s), (short)0, seg_u ? p.bytes0 : bytes1s, 0x00020000);
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            v4i t = __builtin_amdgcn_raw_buffer_load_b128(rsrca, voffa[i], 0, 0);
            ra[i] = *reinterpret_cast<float4*>(&t);
        }
        if (NORM) {
#pragma unroll
            for (int i = 0; i < A_PER; ++i) {
                rmu[i] = *reinterpret_cast<const float4*>(p.nmean + (long)a_b[i] * p.c0 + cch_next);
                rrs[i] = *reinterpret_cast<const float4*>(p.nrstd + (long)a_b[i] * p.c0 + cch_next);
            }
            okbits >>= 16;   // the chunk just issued becomes the one the next commit sees
        }
#define OFX_B_ISSUE(i) \
    if constexpr (B_PER > i) { v4i t = __builtin_amdgcn_raw_buffer_load_b128(rsrcw, voffb[i], 0, 0); rb##i = *reinterpret_cast<float4*>(&t); \
        if constexpr (PREC == 4) { typedef int v2i_ __attribute__((ext_vector_type(2))); v2i_ u = __builtin_amdgcn_raw_buffer_load_b64(rsrcw2, voffb[i] >> 1, 0, 0); rl##i = *reinterpret_cast<float2*>(&u); } }
        OFX_B_ISSUE(0) OFX_B_ISSUE(1) OFX_B_ISSUE(2) OFX_B_ISSUE(3)
#undef OFX_B_ISSUE
    };

    auto norm_a = [&](int i) __attribute__((always_inline)) -> float4 {
        float4 v = ra[i];
        if (NORM) {   // instance norm + ReLU of the producer layer, applied on the fly; padding stays 0
            v.x = fmaxf((v.x - rmu[i].x) * rrs[i].x, 0.f);
            v.y = fmaxf((v.y - rmu[i].y) * rrs[i].y, 0.f);
            v.z = fmaxf((v.z - rmu[i].z) * rrs[i].z, 0.f);
            v.w = fmaxf((v.w - rmu[i].w) * rrs[i].w, 0.f);
            if (!((okbits >> i) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return v;
    };
    // fp32 -> (hi, lo) bf16 pair, round-to-nearest-even both times (v_cvt_pk_bf16_f32): x = hi + lo + O(2^-17 |x|)
    auto split_store = [&](char* hi_row, char* lo_row, float4 v) __attribute__((always_inline)) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        typedef __bf16 b4 __attribute__((ext_vector_type(4)));
        const f4 x = {v.x, v.y, v.z, v.w};
        const b4 h = __builtin_convertvector(x, b4);
        const f4 r = x - __builtin_convertvector(h, f4);
        const b4 l = __builtin_convertvector(r, b4);
        *reinterpret_cast<b4*>(hi_row) = h;
        *reinterpret_cast<b4*>(lo_row) = l;
    };

    // fp32 -> (hi, mid, lo) bf16 triple: x = hi + mid + lo exactly unless lo underflows (3 x 8 mantissa bits)
    auto split_store3 = [&](char* hi_row, char* mid_row, char* lo_row, float4 v) __attribute__((always_inline)) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        typedef __bf16 b4 __attribute__((ext_vector_type(4)));
        const f4 x = {v.x, v.y, v.z, v.w};
        const b4 h = __builtin_convertvector(x, b4);
        const f4 r1 = x - __builtin_convertvector(h, f4);
        const b4 m = __builtin_convertvector(r1, b4);
        const f4 r2 = r1 - __builtin_convertvector(m, f4);
        const b4 l = __builtin_convertvector(r2, b4);
        *reinterpret_cast<b4*>(hi_row) = h;
        *reinterpret_cast<b4*>(mid_row) = m;
        *reinterpret_cast<b4*>(lo_row) = l;
    };

    auto commit = [&](float* stage) __attribute__((always_inline)) {
        if constexpr (X6) {
            char* a0 = reinterpret_cast<char*>(stage);
            char* b0 = a0 + 3 * BM * ROWB;
#pragma unroll
            for (int i = 0; i < A_PER; ++i) {
                const int o = (r0 + RPG * i) * ROWB + kq * 8;
                split_store3(a0 + o, a0 + BM * ROWB + o, a0 + 2 * BM * ROWB + o, norm_a(i));
            }
#define OFX_B_COMMIT(i) \
    if constexpr (B_PER > i) { \
        const int o = (r0 + RPG * i) * ROWB + kq * 8; \
        if constexpr (PREC == 4) { \
            *reinterpret_cast<float2*>(b0 + o) = make_float2(rb##i.x, rb##i.y); \
            *reinterpret_cast<float2*>(b0 + BN_ST * ROWB + o) = make_float2(rb##i.z, rb##i.w); \
            *reinterpret_cast<float2*>(b0 + 2 * BN_ST * ROWB + o) = rl##i; \
        } else { \
            split_store3(b0 + o, b0 + BN_ST * ROWB + o, b0 + 2 * BN_ST * ROWB + o, rb##i); \
        } \
    }
            OFX_B_COMMIT(0) OFX_B_COMMIT(1) OFX_B_COMMIT(2) OFX_B_COMMIT(3)
#undef OFX_B_COMMIT
        } else if constexpr (PREC == 0) {
            float* As = stage;
            float* Bs = As + BM * LDK;
#pragma unroll
            for (int i = 0; i < A_PER; ++i) *reinterpret_cast<float4*>(&As[(r0 + RPG * i) * LDK + kq * 4]) = norm_a(i);
#define OFX_B_COMMIT(i) \
    if constexpr (B_PER > i) *reinterpret_cast<float4*>(&Bs[(r0 + RPG * i) * LDK + kq * 4]) = rb##i;
            OFX_B_COMMIT(0) OFX_B_COMMIT(1) OFX_B_COMMIT(2) OFX_B_COMMIT(3)
#undef OFX_B_COMMIT
        } else {
            char* a_hi = reinterpret_cast<char*>(stage);
            char* a_lo = a_hi + BM * ROWB;
            char* b_hi = a_lo + BM * ROWB;
            char* b_lo = b_hi + BN_ST * ROWB;
#pragma unroll
            for (int i = 0; i < A_PER; ++i) {
                const int o = (r0 + RPG * i) * ROWB + kq * 8;
                split_store(a_hi + o, a_lo + o, norm_a(i));
            }
            // PREC = 2: the weight matrix arrives pre-split ([hi x4 | lo x4] per 16 bytes): two 8-byte copies, no VALU
#define OFX_B_COMMIT(i) \
    if constexpr (B_PER > i) { \
        const int o = (r0 + RPG * i) * ROWB + kq * 8; \
        if constexpr (PREC == 2) { \
            *reinterpret_cast<float2*>(b_hi + o) = make_float2(rb##i.x, rb##i.y); \
            *reinterpret_cast<float2*>(b_lo + o) = make_float2(rb##i.z, rb##i.w); \
        } else { \
            split_store(b_hi + o, b_lo + o, rb##i); \
        } \
    }
            OFX_B_COMMIT(0) OFX_B_COMMIT(1) OFX_B_COMMIT(2) OFX_B_COMMIT(3)
#undef OFX_B_COMMIT
        }
    };

    // ---- prologue: chunk 0 committed, chunk 1 in flight
    offsets(0);
    issue();
    offsets(min(1, nk - 1));
    commit(smem);
    __syncthreads();
    issue();
    // ---- steady state, one barrier per chunk:
    //   [offsets of chunk kt+2 interleaved with the MFMA block on buf[kt&1]] -> commit chunk kt+1 to
    //   buf[(kt+1)&1] -> barrier -> issue the loads of chunk kt+2 (they land during the next MFMA block)
    for (int kt = 0; kt < nk; ++kt) {
        offsets(min(kt + 2, nk - 1));
        if constexpr (PREC == 0) {
            const float* As = smem + (kt & 1) * STAGE;
            const float* Bs = As + BM * LDK;
#pragma unroll
            for (int ks = 0; ks < BK / 8; ++ks) {
                float4 fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    fa[i] = *reinterpret_cast<const float4*>(&As[(wm * WM + i * 32 + frag_row) * LDK + ks * 8 + frag_k]);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    fb[j] = *reinterpret_cast<const float4*>(&Bs[(wn * WN + j * 32 + frag_row) * LDK + ks * 8 + frag_k]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
                    }
            }
        } else if constexpr (X6) {
            // bf16x6: the six products of weight >= 2^-16, smallest first, fp32 accumulate
            typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
            const char* a0 = reinterpret_cast<const char*>(smem + (kt & 1) * STAGE);
            const char* b0 = a0 + 3 * BM * ROWB;
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                bf16x8 ah[TM], am[TM], al[TM], bh[TN], bm_[TN], bl[TN];
                const int ko = ks * 32 + (lane >> 5) * 16;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int o = (wm * WM + i * 32 + frag_row) * ROWB + ko;
                    ah[i] = *reinterpret_cast<const bf16x8*>(a0 + o);
                    am[i] = *reinterpret_cast<const bf16x8*>(a0 + BM * ROWB + o);
                    al[i] = *reinterpret_cast<const bf16x8*>(a0 + 2 * BM * ROWB + o);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int o = (wn * WN + j * 32 + frag_row) * ROWB + ko;
                    bh[j] = *reinterpret_cast<const bf16x8*>(b0 + o);
                    bm_[j] = *reinterpret_cast<const bf16x8*>(b0 + BN_ST * ROWB + o);
                    bl[j] = *reinterpret_cast<const bf16x8*>(b0 + 2 * BN_ST * ROWB + o);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bm_[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bm_[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    }
            }
        } else {
            // bf16x3: acc += a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on v_mfma_f32_32x32x16_bf16 (fp32 accumulate).
            // Lane l supplies row (l & 31) and the 8 consecutive k of group (l >> 5): one 16-byte LDS read per operand.
            typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
            const char* a_hi = reinterpret_cast<const char*>(smem + (kt & 1) * STAGE);
            const char* a_lo = a_hi + BM * ROWB;
            const char* b_hi = a_lo + BM * ROWB;
            const char* b_lo = b_hi + BN_ST * ROWB;
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
                const int ko = ks * 32 + (lane >> 5) * 16;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int o = (wm * WM + i * 32 + frag_row) * ROWB + ko;
                    ah[i] = *reinterpret_cast<const bf16x8*>(a_hi + o);
                    al[i] = *reinterpret_cast<const bf16x8*>(a_lo + o);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int o = (wn * WN + j * 32 + frag_row) * ROWB + ko;
                    bh[j] = *reinterpret_cast<const bf16x8*>(b_hi + o);
                    bl[j] = *reinterpret_cast<const bf16x8*>(b_lo + o);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    }
            }
        }
        // chunk kt+1 has had a whole MFMA block to land; on the last iteration this rewrites the idle
        // buffer with a duplicate that nobody reads
        commit(smem + ((kt + 1) & 1) * STAGE);
        __syncthreads();
        issue();
        __builtin_amdgcn_sched_barrier(0);   // keep the loads ahead of the next MFMA block
    }
    }   // !PATCH

    if constexpr (KS == 2) {
        // add the two pipelines' accumulators: the odd one parks its tile in LDS and retires
        __syncthreads();                                  // every wave is done reading the staging buffers
        if (grp == 1) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) smem_all[((i * TN + j) * 16 + e) * 256 + tid] = acc[i][j][e];
        }
        __syncthreads();
        if (grp == 1) return;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] += smem_all[((i * TN + j) * 16 + e) * 256 + tid];
    }

    if constexpr (SK) {
        // Serial split-K: every split parks its accumulator tile, the LAST workgroup to arrive (per-tile counter) sums
        // the S partial tiles in index order (so the result does not depend on who arrived last) and runs the normal
        // epilogue -- no separate reduction launch.  The partial tiles travel through agent-scope (sc1) stores and
        // loads, which are coherent across CUs and XCDs without flushing the L2 (an agent-scope release fence writes
        // the whole L2 back: measured 2x slower end to end).
        __shared__ int sk_last;
        const int tile_id = mt * p.ntiles + nt;
        // partial tiles travel as 16-byte agent-scope (sc1) stores and loads through buffer descriptors: tile layout
        // [(i, j) sub-tile][e / 4][thread][4 floats] -- one instruction moves 1 KB per wave where the first version's 4-byte atomics
        // moved 256 B (round 5: 8.57 -> 8.4x ms on one 512x768 pair; same values, same summation order)
        typedef int v4i_sk __attribute__((ext_vector_type(4)));
        constexpr int kSc1 = 16;                                   // cache-policy bit sc1 of gfx940+ buffer instructions
        const __amdgpu_buffer_rsrc_t rs_mine = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.sk_part + ((long)tile_id * p.ksplit + split) * (BM * BN)), (short)0, BM * BN * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    v4i_sk t;
                    t[0] = __float_as_int(acc[i][j][e4 * 4 + 0]); t[1] = __float_as_int(acc[i][j][e4 * 4 + 1]);
                    t[2] = __float_as_int(acc[i][j][e4 * 4 + 2]); t[3] = __float_as_int(acc[i][j][e4 * 4 + 3]);
                    __builtin_amdgcn_raw_buffer_store_b128(t, rs_mine, (((i * TN + j) * 4 + e4) * 256 + tid) * 16, 0, kSc1);
                }
        // Every wave must see its sc1 stores ACKNOWLEDGED (performed at agent scope) before the arrival counter moves.
        // A workgroup-scope release fence does not wait for vmcnt on this target (waves of a workgroup share the
        // L1), and an agent-scope one also writes the L2 back; the explicit wait is exactly what is needed.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);                             // vmcnt(0) expcnt(0) lgkmcnt(0)
        __syncthreads();
        if (threadIdx.x == 0) {
            const int old = __hip_atomic_fetch_add(p.sk_count + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sk_last = old == p.ksplit - 1;
            if (sk_last) __hip_atomic_store(p.sk_count + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        }
        __syncthreads();
        if (!sk_last) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                f32x16 sum;
#pragma unroll
                for (int e = 0; e < 16; ++e) sum[e] = 0.f;
                for (int s2 = 0; s2 < p.ksplit; ++s2) {
                    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(
                        (void*)(p.sk_part + ((long)tile_id * p.ksplit + s2) * (BM * BN)), (short)0, BM * BN * 4, 0x00020000);
                    const bool own = s2 == split;                  // wave-uniform: this workgroup's part never left its registers
                    v4i_sk t[4];
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4)
                        t[e4] = own ? v4i_sk{0, 0, 0, 0} : __builtin_amdgcn_raw_buffer_load_b128(rs_o, (((i * TN + j) * 4 + e4) * 256 + tid) * 16, 0, kSc1);
#pragma unroll
                    for (int e = 0; e < 16; ++e) sum[e] += own ? acc[i][j][e] : __int_as_float(t[e >> 2][e & 3]);
                }
                acc[i][j] = sum;
            }
    }

    // ---- epilogue.  C/D layout of the 32x32 MFMA: col (n) = lane&31, row (m) = (e&3) + 8*(e>>2) + 4*(lane>>5).
    // Each 32x32 sub-tile runs in two phases -- every global read it needs (addend, residual, z, h) is
    // issued first, then the arithmetic and the stores.  The epilogue reads and writes the same buffers (h
    // is updated in place), so a naive element loop keeps each load behind the previous element's store and
    // turns the 64 elements of a thread into 64 serial memory round trips.  All accesses go through buffer
    // descriptors: one VGPR byte offset per array (the thread's first row and column), the row step as a
    // scalar offset, and the ragged last M / N tile handled by the hardware range check (offset | kOOB).
    float* __restrict__ out = p.out ? p.out + (long)z * p.o_zs : nullptr;
    const int Mrows = p.M;
    auto rsrc_of = [&](const void* ptr, int ld) {
        const long bytes = ptr ? (long)Mrows * ld * 4 : 0;
        return __builtin_amdgcn_make_buffer_rsrc((void*)ptr, (short)0, (int)bytes, 0x00020000);
    };
    auto ldf = [](__amdgpu_buffer_rsrc_t r, int voff, int soff) {
        return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
    };
    auto stf = [](float v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0);
    };
    const bool has_add = p.addend != nullptr;
    const bool has_res = p.res != nullptr;
    // first row of this thread.  Patch mode: tile row r is pixel (r / 16, r % 16) of the patch, so the thread's first row is a
    // pixel index and the step from it to row_of(q) is srow(q) pixels (wave-uniform: it goes into the scalar offset as before)
    const int mb0 = PATCH ? (pt_b * p.Hin + pt_y0 + (wm * WM) / kPW) * p.Win + pt_x0 + 4 * (lane >> 5) : m0 + wm * WM + 4 * (lane >> 5);
    const int lim = PATCH ? 0x7fffffff : Mrows - mb0;     // relative rows r < lim exist
    // overhanging patches: rows / columns of the map left below / right of this thread's first pixel
    const int pt_ylim = p.Hin - pt_y0 - (wm * WM) / kPW;
    const int pt_xlim = p.Win - pt_x0 - 4 * (lane >> 5);
    constexpr int EB = EPI == OFX_EPI_GRU_Q ? 4 : 8;
    // Control flow is kept out of the element loops: the optional reads are decided once per batch of EB
    // elements, ReLU / the residual ReLU are a max against 0 or -FLT_MAX, the transcendental activations of the
    // plain epilogue are a separate instantiation (a per-element `switch (act)` compiled to ~20 branches per
    // output and 80 KB of code; the epilogue then took as long as ten K-chunks).
    // per-column scale / shift of all TN sub-tiles in one round trip (loaded inside the j loop they cost one
    // dependent global-load latency per sub-tile: ~10 k cycles of a 24 k-cycle epilogue)
    float scv[TN], shv[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nn = min(n0 + wn * WN + j * 32 + (lane & 31), p.Cout - 1);
        scv[j] = p.scale ? p.scale[nn] : 1.0f;
        shv[j] = p.shift ? p.shift[nn] : 0.0f;
    }
    const float act_lo = p.act == OFX_ACT_RELU ? 0.0f : -3.402823466e38f;
    const float res_lo = has_res ? 0.0f : -3.402823466e38f;
    constexpr bool TRANSC = EPI == kEpiPlainT;            // plain epilogue with sigmoid / tanh
    constexpr bool VOLPOOL = EPI == kEpiVolPool;          // blocked correlation volume: also writes pyramid level 1
    constexpr int EPK = (TRANSC || VOLPOOL) ? OFX_EPI_PLAIN : EPI;     // epilogue kind
    auto epilogue = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;  // tile entirely inside M: no per-row masks
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nbase = n0 + wn * WN + j * 32;      // wave-uniform
            const int n = nbase + (lane & 31);
            const bool nok = n < p.Cout;
            const int cmask = nok ? 0 : kOOB;
            const float sc = scv[j] * p.alpha;
            const float sh = shv[j];
            if constexpr (EPK == OFX_EPI_FLOW) {
                // Cout = 2: a handful of lanes; plain pointer code
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int m = mb0 + i * 32 + (e & 3) + 8 * (e >> 2);
                        if (!nok || m >= Mrows) continue;
                        float v = acc[i][j][e] * sc + sh;
                        if (has_add) v += p.addend[(long)m * p.ldadd + n];
                        const int rem = m % HWo;
                        const int oy = rem / p.Wout;
                        const int ox = rem - oy * p.Wout;
                        const float c1 = p.aux_coords[(long)m * 2 + n] + v;
                        p.aux_coords[(long)m * 2 + n] = c1;
                        const float fl = c1 - (float)(n == 0 ? ox : oy);
                        p.aux_h[(long)m * p.ldh + n] = fl;
                        p.aux_flow4[(long)m * 4 + n] = fl;
                    }
            } else {
                const int hd = p.Cout >> 1;
                // the z | r split of the fused GRU gate conv is 32-aligned -> uniform per wave and j
                const bool r_half = EPK == OFX_EPI_GRU_ZR && __builtin_amdgcn_readfirstlane(nbase) >= hd;
                const __amdgpu_buffer_rsrc_t rs_add = rsrc_of(p.addend, p.ldadd);
                const int vo_add = ((mb0 * p.ldadd + n) * 4) | cmask;
                // array 1 / array 2 read by this epilogue, array written
                const void* p1 = EPK == OFX_EPI_PLAIN ? (const void*)p.res : EPK == OFX_EPI_GRU_ZR ? (const void*)p.aux_h : (const void*)p.aux_z;
                const int ld1 = EPK == OFX_EPI_PLAIN ? p.ldres : EPK == OFX_EPI_GRU_ZR ? p.ldh : p.Cout;
                const int c1 = EPK == OFX_EPI_GRU_ZR ? n - hd : n;
                const __amdgpu_buffer_rsrc_t rs_1 = rsrc_of(p1, ld1);
                const int vo_1 = ((mb0 * ld1 + c1) * 4) | cmask;
                const __amdgpu_buffer_rsrc_t rs_h = rsrc_of(p.aux_h, p.ldh);          // GRU_Q: h read and written
                const int vo_h = ((mb0 * p.ldh + n) * 4) | cmask;
                const void* pw = EPK == OFX_EPI_PLAIN ? (void*)out : EPK == OFX_EPI_GRU_Q ? (void*)p.aux_h : r_half ? (void*)p.aux_rh : (void*)p.aux_z;
                const int ldw = EPK == OFX_EPI_PLAIN ? p.ldo : EPK == OFX_EPI_GRU_Q ? p.ldh : hd;
                const int cw = (EPK == OFX_EPI_GRU_ZR && r_half) ? n - hd : n;
                const __amdgpu_buffer_rsrc_t rs_w = rsrc_of(pw, ldw);
                const int vo_w = ((mb0 * ldw + cw) * 4) | cmask;
                const bool need1 = EPK == OFX_EPI_PLAIN ? has_res : EPK == OFX_EPI_GRU_ZR ? r_half : true;
                // VOLPOOL: the 32 columns of this sub-tile are one 4x8 block of the pixel's level-0 slice (column order is
                // blocked, n0 % 32 == 0); its 2x2 averages are a 2x4 patch of level 1.  The partners of column c are c ^ 1
                // (same row, next column) and c ^ 8 (next row): two DPP adds inside the 16-lane row, then the lanes with
                // bits 0 and 3 clear hold the sums.
                __amdgpu_buffer_rsrc_t rs_p = rs_w;
                int vo_p = 0;
                bool pool_lane = false;
                if constexpr (VOLPOOL) {
                    const int blk = __builtin_amdgcn_readfirstlane(nbase) >> 5;
                    const int by = blk / p.pool_wb0, bx = blk - by * p.pool_wb0;
                    const int c = lane & 31;
                    const int y1 = (by << 1) + ((c >> 4) & 1), x1 = (bx << 2) + ((c >> 1) & 3);
                    const int idx1 = ((((y1 >> 2) * p.pool_wb1) + (x1 >> 3)) << 5) + ((y1 & 3) << 3) + (x1 & 7);
                    pool_lane = nok && (c & 9) == 0;
                    rs_p = __builtin_amdgcn_make_buffer_rsrc((void*)(p.pool_out + (long)z * p.pool_zs), (short)0, (int)((long)Mrows * p.pool_slice1 * 4), 0x00020000);
                    vo_p = ((mb0 * p.pool_slice1 + idx1) * 4) | (pool_lane ? 0 : kOOB);
                }
                const bool do_stats = EPK == OFX_EPI_PLAIN && !TRANSC && !VOLPOOL && p.stats != nullptr;
                float st_s = 0.0f, st_q = 0.0f;                 // this lane's column over the wave's WM rows (its 16 * TM elements)
#pragma unroll
                for (int ib = 0; ib < TM * (16 / EB); ++ib) {
                    // EB elements per phase: enough loads in flight to cover the latency, few enough live
                    // registers to keep the kernel at its three / four workgroups per CU
                    const int i = ib / (16 / EB), e0 = (ib % (16 / EB)) * EB;
                    float ad[EB], x1[EB], x2[EB];
                    auto row_of = [&](int q) { const int e = e0 + q; return i * 32 + (e & 3) + 8 * (e >> 2); };
                    auto srow = [&](int q) {
                        const int e = e0 + q;
                        return !PATCH ? row_of(q)
                               : kPW == 16 ? (2 * i + (e >> 3)) * p.Win + (e & 3) + 8 * ((e >> 2) & 1)
                                           : (4 * i + (e >> 2)) * p.Win + (e & 3);
                    };
                    // rows of the tile that do not exist: past M (general), or pixels of an overhanging patch outside the map
                    auto mask_of = [&](int r) {
                        if constexpr (FULL) return 0;
                        if constexpr (!PATCH) return r < lim ? 0 : kOOB;
                        const int e = r & 31;                 // r = i * 32 + (e & 3) + 8 * (e >> 2), the lane's 4 * h sits in pt_xl
                        const int ii = r >> 5;
                        const int py = kPW == 16 ? 2 * ii + (e >> 4) : 4 * ii + (e >> 3);
                        const int px = kPW == 16 ? (e & 3) + (e & 8) : (e & 3);
                        return (py < pt_ylim && px < pt_xlim) ? 0 : kOOB;
                    };
                    if (has_add) {
#pragma unroll
                        for (int q = 0; q < EB; ++q) ad[q] = ldf(rs_add, vo_add | mask_of(row_of(q)), srow(q) * p.ldadd * 4);
                    } else {
#pragma unroll
                        for (int q = 0; q < EB; ++q) ad[q] = 0.0f;
                    }
                    if (need1) {
#pragma unroll
                        for (int q = 0; q < EB; ++q) x1[q] = ldf(rs_1, vo_1 | mask_of(row_of(q)), srow(q) * ld1 * 4);
                    } else {
#pragma unroll
                        for (int q = 0; q < EB; ++q) x1[q] = EPK == OFX_EPI_GRU_ZR ? 1.0f : 0.0f;   // z half: no r*h product
                    }
#pragma unroll
                    for (int q = 0; q < EB; ++q) x2[q] = EPK == OFX_EPI_GRU_Q ? ldf(rs_h, vo_h | mask_of(row_of(q)), srow(q) * p.ldh * 4) : 0.0f;
                    float v[EB];
#pragma unroll
                    for (int q = 0; q < EB; ++q) v[q] = acc[i][j][e0 + q] * sc + sh + ad[q];
                    if (EPK == OFX_EPI_PLAIN && TRANSC) {
                        if (p.act == OFX_ACT_SIGMOID) {
#pragma unroll
                            for (int q = 0; q < EB; ++q) v[q] = ofx_sigmoid(v[q]);
                        } else {
#pragma unroll
                            for (int q = 0; q < EB; ++q) v[q] = ofx_tanh(v[q]);
                        }
                    }
#pragma unroll
                    for (int q = 0; q < EB; ++q) {
                        if (EPK == OFX_EPI_PLAIN) {
                            if (!TRANSC) v[q] = fmaxf(v[q], act_lo);   // ReLU or identity
                            v[q] = fmaxf(v[q] + x1[q], res_lo);        // residual merge: relu(y + res), or y (x1 = 0)
                        } else if (EPK == OFX_EPI_GRU_ZR) {
                            v[q] = ofx_sigmoid(v[q]) * x1[q];          // z, or r * h
                        } else {
                            v[q] = (1.0f - x1[q]) * x2[q] + x1[q] * ofx_tanh(v[q]);
                        }
                        stf(v[q], rs_w, vo_w | mask_of(row_of(q)), srow(q) * ldw * 4);
                        if constexpr (EPK == OFX_EPI_PLAIN && !TRANSC && !VOLPOOL) {
                            if (do_stats) {                    // wave-uniform
                                const float vv = (FULL || mask_of(row_of(q)) == 0) ? v[q] : 0.0f;
                                st_s += vv;
                                st_q = fmaf(vv, vv, st_q);
                            }
                        }
                        if constexpr (VOLPOOL) {
                            const float s1 = v[q] + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[q]), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]: lane ^ 1
                            const float s2 = s1 + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s1), 0x128, 0xF, 0xF, false));    // row_ror:8: lane ^ 8
                            stf(s2 * 0.25f, rs_p, vo_p | mask_of(row_of(q)), srow(q) * p.pool_slice1 * 4);
                        }
                    }
                }
                if constexpr (EPK == OFX_EPI_PLAIN && !TRANSC && !VOLPOOL) {
                    if (do_stats) {
                        // the two lane halves hold the other rows of the same column
                        st_s += __shfl_xor(st_s, 32, 64);
                        st_q += __shfl_xor(st_q, 32, 64);
                        if (lane < 32 && nok) {
                            float* o = p.stats + (((long)mt * (BM / WM) + wm) * p.Cout + n) * 2;
                            o[0] = st_s;
                            o[1] = st_q;
                        }
                    }
                }
            }
        }
    };
    if (PATCH ? (pt_y0 + kPH <= p.Hin && pt_x0 + kPW <= p.Win) : (m0 + BM <= Mrows)) epilogue(std::true_type{});
    else epilogue(std::false_type{});
}

template <int BM, int BN, int WM, int WN, int BK, int PREC, int KS, bool SK, int UK>
int launch_tile_uk(const ConvK& k, int epi, bool norm, int nz, hipStream_t s) {
    dim3 grid((unsigned)(k.mtiles * k.ntiles * (k.ksplit > 1 ? k.ksplit : 1)), (unsigned)nz, 1);
    dim3 block(256 * KS, 1, 1);
    switch (epi) {
        case OFX_EPI_PLAIN:
            if (k.act >= OFX_ACT_SIGMOID) {
                if (norm) return OFX_EINVAL;   // fused-norm producer layers are followed by ReLU / identity only
                OFX_LAUNCH((igemm_kernel<BM, BN, WM, WN, kEpiPlainT, false, BK, PREC, KS, SK, UK>), grid, block, s, k);
            } else if (norm) {
                OFX_LAUNCH((igemm_kernel<BM, BN, WM, WN, OFX_EPI_PLAIN, true, BK, PREC, KS, SK, UK>), grid, block, s, k);
            } else {
                OFX_LAUNCH((igemm_kernel<BM, BN, WM, WN, OFX_EPI_PLAIN, false, BK, PREC, KS, SK, UK>), grid, block, s, k);
            }
            break;
        case OFX_EPI_GRU_ZR: OFX_LAUNCH((igemm_kernel<BM, BN, WM, WN, OFX_EPI_GRU_ZR, false, BK, PREC, KS, SK, UK>), grid, block, s, k); break;
        case OFX_EPI_GRU_Q: OFX_LAUNCH((igemm_kernel<BM, BN, WM, WN, OFX_EPI_GRU_Q, false, BK, PREC, KS, SK, UK>), grid, block, s, k); break;
        case OFX_EPI_FLOW:
            if constexpr (UK != 2) {
                OFX_LAUNCH((igemm_kernel<BM, BN, WM, WN, OFX_EPI_FLOW, false, BK, PREC, KS, SK, UK>), grid, block, s, k);
                break;
            }
            return OFX_EINVAL;
        case kEpiVolPool:
            if constexpr (BM == 128 && BN == 128 && BK == 16 && KS == 1 && !SK && UK != 2) {
                OFX_LAUNCH((igemm_kernel<128, 128, 64, 64, kEpiVolPool, false, 16, PREC, 1, false, UK>), grid, block, s, k);
                break;
            }
            return OFX_EINVAL;
        default: return OFX_EINVAL;
    }
    return ofx_launch_status();
}

template <int BM, int BN, int WM, int WN, int BK, int PREC = 0, int KS = 1, bool SK = false>
int launch_tile(const ConvK& k, int epi, bool norm, int nz, hipStream_t s) {
    if constexpr (KS == 1 && ((PREC == 0 && BM == 128 && BK == 16 && !SK && (BN == 64 || BN == 96 || BN == 128 || BN == 192)) || (PREC == 0 && BM == 64 && BN == 64 && BK == 32) ||
                              (PREC == 0 && BM == 256 && BN == 64 && BK == 16 && !SK) ||
                              ((PREC == 1 || PREC == 2 || PREC == 3 || PREC == 4) && BM == 128 && BK == 16 && !SK && (BN == 64 || BN == 128)))) {
        if (k.patch && epi != OFX_EPI_FLOW && epi != kEpiVolPool) return launch_tile_uk<BM, BN, WM, WN, BK, PREC, KS, SK, 2>(k, epi, norm, nz, s);
    }
    if constexpr (PREC == 0 && BN != 192) {   // the 128x192 tile measured 0.8 % slower with scalar chunk coordinates
        if (k.uk) return launch_tile_uk<BM, BN, WM, WN, BK, PREC, KS, SK, 1>(k, epi, norm, nz, s);
    }
    return launch_tile_uk<BM, BN, WM, WN, BK, PREC, KS, SK, 0>(k, epi, norm, nz, s);
}

}  // namespace

namespace {
struct VolPool {            // set by ofx_conv2d_volpool around one ofx_conv2d_alpha call (same thread)
    bool on = false;
    float* out = nullptr;
    long zs = 0;
    int wb0 = 0, wb1 = 0, slice1 = 0;
};
thread_local VolPool tl_pool;

// Instance-norm statistics from the accumulators (set by ofx_conv2d_stats around one ofx_conv2d call): every wave of a tile writes
// the per-channel sum and sum of squares of its WM rows; a finalize kernel adds them per image in a fixed order.  Saves the
// statistics pass over the tensor the convolution has just written.
struct StatsReq {
    bool on = false;
    float* part = nullptr;
    size_t cap_floats = 0;
    int rows_per_image = 0;     // out: 0 = not produced (the caller falls back to ofx_inorm_stats)
};
thread_local StatsReq tl_stats;
}  // namespace

extern "C" int ofx_conv2d(const ofx_conv_desc* d, void* stream) { return ofx_conv2d_alpha(d, 1.0f, stream); }

// ofx_conv2d that also leaves, when the launch qualifies, the per-wave (sum, sum of squares) of every output channel in `part`
// ([B][rows_per_image][Cout][2] floats) for ofx_inorm_finalize_part; *rows_per_image = 0 means "not produced".
int ofx_conv2d_stats(const ofx_conv_desc* d, float* part, size_t part_floats, int* rows_per_image, void* stream) {
    tl_stats.on = true;
    tl_stats.part = part;
    tl_stats.cap_floats = part_floats;
    tl_stats.rows_per_image = 0;
    const int st = ofx_conv2d_alpha(d, 1.0f, stream);
    *rows_per_image = st ? 0 : tl_stats.rows_per_image;
    tl_stats = StatsReq{};
    return st;
}

// Correlation volume in the blocked layout + pyramid level 1 from the accumulators (corr.hip decides when it applies:
// 128x128 tiles, h % 8 == 0 and w % 16 == 0 so that level 1 is tiled by whole blocks; any arithmetic).  pool_out: level 1,
// [nz][M][slice1]; wb0 / wb1: blocks per slice row of level 0 / 1.
int ofx_conv2d_volpool(const ofx_conv_desc* d, float alpha, float* pool_out, long pool_zs, int wb0, int wb1, int slice1, void* stream) {
    OFX_REQUIRE(d && pool_out && wb0 > 0 && wb1 > 0 && slice1 > 0, OFX_EINVAL);
    OFX_REQUIRE(d->epi == OFX_EPI_PLAIN && d->act == OFX_ACT_NONE && !d->res && !d->addend && !d->nmean &&
                    !d->scale && !d->shift && d->Cout % 128 == 0 && d->tile == 0,
                OFX_EINVAL);
    tl_pool.on = true; tl_pool.out = pool_out; tl_pool.zs = pool_zs; tl_pool.wb0 = wb0; tl_pool.wb1 = wb1; tl_pool.slice1 = slice1;
    const int st = ofx_conv2d_alpha(d, alpha, stream);
    tl_pool = VolPool{};
    return st;
}

extern "C" int ofx_conv2d_alpha(const ofx_conv_desc* d, float alpha, void* stream) {
    OFX_REQUIRE(d != nullptr, OFX_EINVAL);
    OFX_REQUIRE(d->in0 && d->w, OFX_EINVAL);
    OFX_REQUIRE(d->c0 > 0 && d->c0 % 4 == 0 && d->ld0 % 4 == 0, OFX_EALIGN);
    OFX_REQUIRE(ofx_aligned16(d->in0) && ofx_aligned16(d->w), OFX_EALIGN);
    if (d->in1) {
        OFX_REQUIRE(d->c1 > 0 && d->c1 % 4 == 0 && d->ld1 % 4 == 0 && ofx_aligned16(d->in1), OFX_EALIGN);
    } else {
        OFX_REQUIRE(d->c1 == 0, OFX_EINVAL);
    }
    OFX_REQUIRE(d->B > 0 && d->Hin > 0 && d->Win > 0 && d->Hout > 0 && d->Wout > 0 && d->Cout > 0, OFX_EINVAL);
    OFX_REQUIRE(d->KH > 0 && d->KW > 0 && d->stride > 0, OFX_EINVAL);
    if (d->nmean) {
        OFX_REQUIRE(d->nrstd && ofx_aligned16(d->nmean) && ofx_aligned16(d->nrstd), OFX_EALIGN);
        OFX_REQUIRE(d->in1 == nullptr && d->epi == OFX_EPI_PLAIN, OFX_EINVAL);   // fused norm: single segment, plain epilogue
    }
    if (d->addend) OFX_REQUIRE(d->ldadd >= d->Cout, OFX_EINVAL);
    const int nz = d->nz > 1 ? d->nz : 1;
    OFX_REQUIRE(d->precision >= OFX_PREC_FP32 && d->precision <= OFX_PREC_BF16X6_W, OFX_EINVAL);
    // OFX_PREC_BF16X6_W: the kernel finds the lo pieces of the pre-split matrix at w + Cout * Kpad * 4 -- true only for the WHOLE matrix
    // `ofx_split_conv_weight3` converted (Cout = its row count) of ONE problem; a batched GEMM advances w per problem
    if (d->precision == OFX_PREC_BF16X6_W) OFX_REQUIRE(nz == 1, OFX_EINVAL);
    {
        // The epilogue addresses out / res / addend / aux_* with 32-bit byte offsets (descriptor extent
        // M * ld * 4).  A pure GEMM (1x1, stride 1, no padding, plain epilogue -- the correlation volume of a
        // large frame is the case that gets here) is split along M; anything else must be sliced by the caller.
        int ld_epi = std::max(d->Cout, d->out ? d->ldo : 0);
        if (d->res) ld_epi = std::max(ld_epi, d->ldres);
        if (d->addend) ld_epi = std::max(ld_epi, d->ldadd);
        if (d->aux_h) ld_epi = std::max(ld_epi, d->ldh);
        const long Mtot = (long)d->B * d->Hout * d->Wout;
        const long lim = (1L << 31) - 64;
        if (Mtot * ld_epi * 4 >= lim) {
            const bool gemm = d->KH == 1 && d->KW == 1 && d->stride == 1 && d->padH == 0 && d->padW == 0 &&
                              d->epi == OFX_EPI_PLAIN && !d->nmean && d->Hout == d->Hin && d->Wout == d->Win;
            OFX_REQUIRE(gemm, OFX_EINVAL);
            const long rows = std::max<long>(128, ((lim / ((long)ld_epi * 4)) / 128 - 1) * 128);
            OFX_REQUIRE(rows * ld_epi * 4 < lim, OFX_EINVAL);
            for (long r0 = 0; r0 < Mtot; r0 += rows) {
                ofx_conv_desc part = *d;
                const long n = std::min(rows, Mtot - r0);
                part.B = 1; part.Hin = part.Hout = 1; part.Win = part.Wout = (int)n;
                part.in0 = d->in0 + r0 * d->ld0;
                if (d->in1) part.in1 = d->in1 + r0 * d->ld1;
                part.out = d->out + r0 * d->ldo;
                if (d->res) part.res = d->res + r0 * d->ldres;
                if (d->addend) part.addend = d->addend + r0 * d->ldadd;
                const VolPool saved = tl_pool;
                if (saved.on) tl_pool.out = saved.out + r0 * saved.slice1;
                const int st = ofx_conv2d_alpha(&part, alpha, stream);
                tl_pool = saved;
                if (st) return st;
            }
            return 0;
        }
    }

    ConvK k;
    k.in0 = d->in0; k.in1 = d->in1; k.w = d->w; k.scale = d->scale; k.shift = d->shift; k.addend = d->addend;
    k.out = d->out; k.res = d->res; k.nmean = d->nmean; k.nrstd = d->nrstd;
    k.aux_z = d->aux_z; k.aux_rh = d->aux_rh; k.aux_h = d->aux_h;
    k.aux_coords = d->aux_coords; k.aux_flow4 = d->aux_flow4;
    k.a_zs = nz > 1 ? d->a_zs : 0; k.w_zs = nz > 1 ? d->w_zs : 0; k.o_zs = nz > 1 ? d->o_zs : 0;
    k.ld0 = d->ld0; k.c0 = d->c0; k.ld1 = d->ld1; k.c1 = d->c1; k.cin = d->c0 + d->c1;
    k.ldo = d->ldo; k.ldres = d->ldres; k.ldh = d->ldh; k.ldadd = d->ldadd;
    k.Hin = d->Hin; k.Win = d->Win; k.Hout = d->Hout; k.Wout = d->Wout; k.Cout = d->Cout;
    k.KW = d->KW; k.stride = d->stride; k.padH = d->padH; k.padW = d->padW;
    k.K = d->KH * d->KW * k.cin;
    k.Kpad = ((k.K + kKAlign - 1) / kKAlign) * kKAlign;
    const long M = (long)d->B * d->Hout * d->Wout;
    OFX_REQUIRE(M < (1L << 31) && (long)d->B * d->Hin * d->Win < (1L << 31), OFX_EINVAL);
    k.M = (int)M;
    k.act = d->act;
    k.alpha = alpha;
    auto magic = [](int dv) -> unsigned { return dv <= 1 ? 0u : (unsigned)(((1ull << 32) + dv - 1) / dv); };
    k.magic_cin = magic(k.cin);
    k.magic_kw = magic(d->KW);
    k.kw1_mask = d->KW == 1 ? 0xFFFFFFFFu : 0u;
    // 32-bit byte offsets through buffer descriptors: every operand extent must stay below 2 GiB
    const long npix_in = (long)d->B * d->Hin * d->Win;
    const long ext0 = ((npix_in - 1) * d->ld0 + d->c0) * 4, ext1 = d->in1 ? ((npix_in - 1) * d->ld1 + d->c1) * 4 : 0;
    const long extw = (long)d->Cout * k.Kpad * 4;
    OFX_REQUIRE(ext0 < (1L << 31) - 64 && ext1 < (1L << 31) - 64 && extw < (1L << 31) - 64, OFX_EINVAL);
    OFX_REQUIRE(k.Kpad < 65536, OFX_EINVAL);                       // umulhi division is exact in this range
    if (d->in1) OFX_REQUIRE(d->c0 % kKAlign == 0, OFX_EALIGN);           // a K chunk never straddles the two segments
    k.bytes0 = (int)ext0; k.bytes1 = (int)ext1; k.bytesw = (int)extw;
    k.pool_out = tl_pool.on ? tl_pool.out : nullptr; k.pool_zs = tl_pool.zs; k.pool_wb0 = tl_pool.wb0; k.pool_wb1 = tl_pool.wb1;
    k.pool_slice1 = tl_pool.slice1;
    if (tl_pool.on) OFX_REQUIRE(M * (long)tl_pool.slice1 * 4 < (1L << 31) - 64, OFX_EINVAL);

    switch (d->epi) {
        case OFX_EPI_PLAIN:
            OFX_REQUIRE(d->out != nullptr && d->ldo >= d->Cout, OFX_EINVAL);
            if (d->res) OFX_REQUIRE(d->ldres >= d->Cout, OFX_EINVAL);
            break;
        case OFX_EPI_GRU_ZR:
            OFX_REQUIRE(d->aux_z && d->aux_rh && d->aux_h && d->Cout % 2 == 0 && d->ldh >= d->Cout / 2, OFX_EINVAL);
            break;
        case OFX_EPI_GRU_Q:
            OFX_REQUIRE(d->aux_z && d->aux_h && d->ldh >= d->Cout, OFX_EINVAL);
            break;
        case OFX_EPI_FLOW:
            OFX_REQUIRE(d->aux_coords && d->aux_h && d->aux_flow4 && d->Cout == 2, OFX_EINVAL);
            break;
        default: return OFX_EINVAL;
    }

    // ---- tile selection
    int bm, bn;
    if (d->tile) {
        bm = (d->tile % 1000000) / 1000;
        bn = d->tile % 1000;
    } else {
        auto waste = [&](int t) { return (double)(((d->Cout + t - 1) / t) * t) / d->Cout; };
        if (d->Cout <= 32) bn = 32;
        else if (waste(128) <= 1.13 && !(waste(192) <= 1.0 && waste(128) > 1.05 && d->precision == OFX_PREC_FP32 && !d->nmean && d->epi == OFX_EPI_PLAIN)) bn = 128;   // (576 channels: 3 x 192, not 4.5 x 128)
        else if (waste(192) <= 1.05 && d->precision == OFX_PREC_FP32 && !d->nmean && d->epi != OFX_EPI_FLOW) bn = 192;   // 192-channel layers: one 128x192 tile instead of 128x64 x 3
        else if (waste(96) <= 1.05 && d->precision == OFX_PREC_FP32 && d->epi == OFX_EPI_PLAIN) bn = 96;   // 96-channel encoder stage
        else if (waste(64) <= 1.13) bn = 64;
        else if (waste(32) < waste(64) - 0.1) bn = 32;
        else bn = 64;
        bm = 128;
        // Grids that do not fill the chip (B = 1 ... ~16 frames at 512x768).  Measured over every layer shape of the network at
        // B = 1, 2, 4, 8, 16, 32 (tools/small_batch_tune.py, profiles/r05_small_batch_tune.txt): a 128-row tile keeps its rate down
        // to THREE workgroups per CU (768 on this part) and loses to the next smaller tile below that -- so take the largest tile
        // whose grid still has 768 workgroups: 128x192 -> 128x96 -> 128x64 for the 192-channel layer, 128x128 -> 128x64 for the
        // 128- / 256-channel ones, and only then the 64x64 small-grid tile (with split-K / paired pipelines below).  The old rule
        // (64x64 below 1024 blocks of the 128-row tile) gave up 8-16 % per layer at B = 4 ... 16.
        const long mt128 = (M + 127) / 128;
        auto blocks_of = [&](int t) { return mt128 * ((d->Cout + t - 1) / t) * nz; };
        static const bool old_small = getenv("OFX_CONV_OLD_SMALL_TILES") != nullptr;      // A/B switch: the round-4 rule
        constexpr long kFill = 768;                  // three workgroups per CU
        const long blocks128 = blocks_of(bn);
        if (old_small) {
            if (bn == 96 && blocks128 < 1024) bn = 32;
            if (bn >= 64 && blocks128 < 1024) bm = 64;
            if (bm == 64 && bn >= 128) bn = 64;
        } else if (bn >= 64 && blocks128 < kFill) {
            if (bn == 192 && d->epi == OFX_EPI_PLAIN && blocks_of(96) >= kFill) bn = 96;
            else if (bn >= 128 && blocks_of(64) >= kFill) bn = 64;
            else if (bn == 96) bn = 32;              // no small-grid variant of the 96-wide tile
            else { bm = 64; bn = 64; }               // 64x64: the chip is filled by splitting K instead (below)
        }
    }
    k.mtiles = (int)((M + bm - 1) / bm);
    k.ntiles = (d->Cout + bn - 1) / bn;
    k.group_m = k.ntiles >= 8 ? 8 : 1;
    // split-K (fp32, 64x64 tiles, single z): a grid of a few hundred tiles leaves a 256-CU part with a half-empty
    // second round (384 tiles = 1.5 per CU: the makespan is 2 tiles); S splits make 384*S shorter work items that
    // balance.  Needs caller scratch: 64 KiB of counters + tiles * S * 64*64 floats.
    k.ksplit = 1; k.sk_part = nullptr; k.sk_count = nullptr;
    static const bool dbg_no_sk = getenv("OFX_NO_SPLITK") != nullptr;
    if (!dbg_no_sk && d->splitk_ws && d->precision == OFX_PREC_FP32 && bm == 64 && bn == 64 && nz == 1 && d->tile == 0) {
        const long tiles = (long)k.mtiles * k.ntiles;
        const int nk32 = (int)(k.Kpad / 32);
        // work per CU in tile units if the tiles are cut S ways: ceil(tiles * S / 256) / S -- take the S that
        // minimises it (ties: fewer splits), with at least six 32-wide chunks per split
        int S = 1;
        // (tiles up to 2048: with five workgroups per CU resident, `convc2` of four frames -- 1152 tiles = 4.5 per CU -- runs as
        // 2304 half-K workgroups: 24.15 -> 23.98 ms per four frames; the scratch bound below still applies.  OFX_SK_MAX_TILES overrides)
        static const char* skt_env = getenv("OFX_SK_MAX_TILES");
        const long sk_max_tiles = skt_env ? atol(skt_env) : 2048;
        if (tiles <= sk_max_tiles && nk32 >= 12) {
            const double base = (double)((tiles + 255) / 256);
            double best = base;
            for (int c = 2; c <= 4; ++c) {
                if (nk32 / c < 6) break;
                const double span = (double)((tiles * c + 255) / 256) / c;
                // a split must buy more balance than its seam costs: 4 % of the unsplit span per split (round 6: 960 tiles cut four
                // ways for a nominal 4 -> 3.75 ran 7 % SLOWER than unsplit -- `conv` / GRU q of five frames; 1152 tiles cut in two
                // for 5 -> 4.5 gained 3 %; the single pair's layers gain 25-37 %)
                if (span < best - 1e-9 && span < base * (1.0 - 0.04 * c) + 1e-9) { best = span; S = c; }
            }
        }
        const size_t need = 65536 + (size_t)tiles * S * 64 * 64 * sizeof(float);
        if (S > 1 && tiles * sizeof(int) <= 65536 && need <= d->splitk_ws_bytes && ofx_aligned16(d->splitk_ws)) {
            k.ksplit = S;
            k.sk_count = (int*)d->splitk_ws;
            k.sk_part = (float*)((char*)d->splitk_ws + 65536);
        }
    }
    hipStream_t s = (hipStream_t)stream;
    const bool norm = d->nmean != nullptr;
    const char* pname = nz > 1 ? "igemm_corr_volume"
                       : d->epi == OFX_EPI_GRU_ZR ? "igemm_conv_gru_zr"
                       : d->epi == OFX_EPI_GRU_Q  ? "igemm_conv_gru_q"
                       : d->epi == OFX_EPI_FLOW   ? "igemm_conv_flow"
                                                  : "igemm_conv";
    OfxProfScope prof(pname, s);
    prof.flops(2.0 * (double)M * d->Cout * k.K * nz);
    // BK = 16 keeps LDS at 40 KB and registers under 128 for the 128x128 tile -> 4 workgroups per CU (3 for 128x192); the
    // extra resident wave per SIMD hides the commit/barrier/issue phases better than a longer chunk does
    // (measured +4..10 % on every shape).  The 64x64 tile is only chosen for grids that under-fill the
    // machine (one workgroup per CU or fewer): there each chunk's load latency is exposed and the longer
    // chunk wins (+8..20 % at one 512x768 pair).  tile = BK*1e6 + BM*1e3 + BN overrides.
    const int tile_bk = (d->tile % 1000000000) / 1000000;
    const int bk = tile_bk ? tile_bk : ((bn == 32 || bm == 64) ? 32 : 16);
    // uniform-K fast path: every chunk of this launch's BK inside one tap and one segment (the 16-float flow rows of convf1 qualify
    // with BK = 16; a caller-forced tile keeps the conservative multiple-of-32 rule)
    static const bool no_uk = getenv("OFX_CONV_NO_UK") != nullptr;
    // (`bk` is the BK of the fp32 tile launched below; the split-bf16 remap further down changes tiles, but launch_tile takes the
    // scalar-coordinate schedule for PREC == 0 only, so `uk` is never read for a tile it was not derived from)
    const int ukm = d->tile ? 32 : bk;
    k.uk = (!no_uk && d->precision == OFX_PREC_FP32 && k.cin % ukm == 0 && (d->c1 == 0 || d->c0 % ukm == 0)) ? 1 : 0;
    // halo-patch kernel: stride-1 3x3 / 1x5 / 5x1, "same" padding, the map a whole number of 8x16 patches, whole 16-channel slabs
    static const bool no_patch = getenv("OFX_CONV_NO_PATCH") != nullptr;
    const bool shape_ok = (d->KH == 3 && d->KW == 3) || (d->KH == 1 && d->KW == 5) || (d->KH == 5 && d->KW == 1);
    k.KH = d->KH;
    const bool big = (bm == 128 || (bm == 256 && bn == 64)) && bk == 16, small = bm == 64 && bn == 64 && bk == 32;   // 8x16 (16x16) patches / 8x8 patches (small grids, split-K)
    const int pw = big ? 16 : 8, ph = bm == 256 ? 16 : 8;
    // the 128-row tiles also take maps that are not whole patches (the last patch of a row / column hangs over: its outside
    // pixels stage zeros and are masked in the epilogue); the small tile, whose grid feeds the split-K choice above, does not
    const bool whole = d->Hin % ph == 0 && d->Win % pw == 0;
    // ... as long as the overhang is cheap: the rows of a patch outside the map are multiplied like any others (a 68 x 120 map --
    // 544x960 frames -- is 9 x 8 patches = 72 x 128 pixels: 1.13x the work).  Beyond a cover of OFX_PATCH_MAX_WASTE (default 1.09) a
    // layer takes the scalar-coordinate / general kernels, which compute no row twice.  Measured, 16 frames, whole forward
    // (tools/odd_sizes.py, patch kernel -> general kernels): cover 1.07 (720x1280) 218 -> 227 ms, 1.13 (544x960) 137 -> 117,
    // 1.19 (600x800) 133 -> 117, 1.32 (776x520) 122 -> 107: the crossover sits near 1.09
    static const char* waste_env = getenv("OFX_PATCH_MAX_WASTE");
    static const double max_waste = waste_env ? atof(waste_env) : 1.09;
    const double patch_waste = (double)(((d->Hin + ph - 1) / ph) * ph) * (double)(((d->Win + pw - 1) / pw) * pw) / ((double)d->Hin * d->Win);
    const bool no_patch_here = no_patch || (!whole && d->tile == 0 && patch_waste > max_waste);   // (a forced tile keeps the patch kernel: the tests' way in)
    k.patch = (!no_patch_here && d->precision == OFX_PREC_FP32 && shape_ok && d->stride == 1 && d->padH == d->KH / 2 && d->padW == d->KW / 2 &&
               d->Hin == d->Hout && d->Win == d->Wout && (whole || big) && k.cin % bk == 0 &&
               (d->c1 == 0 || d->c0 % bk == 0) && (!d->nmean || d->c1 == 0) && nz == 1 && (big || small) &&
               (bn == 64 || bn == 96 || bn == 128 || bn == 192) && d->epi != OFX_EPI_FLOW)
                  ? 1 : 0;
    static const bool old_p256 = getenv("OFX_CONV_OLD_SMALL_TILES") != nullptr;
    if (k.patch && bm == 128 && bn == 64 && d->tile == 0 && d->Hin % 16 == 0 && d->Win % 16 == 0 && M / 256 >= (old_p256 ? 4096 : 768)) {
        // 64-channel layers: 16x16 patches (256x64 tile, 64x64 per wave) halve the weight staging per MFMA: 136 -> 139-143 TF on
        // large grids, and ahead of the 128x64 tile from three workgroups per CU on (round 5 sweep: 110 vs 133 us at 768
        // workgroups, 217 vs 232 at 1536; round 4 switched at 4096)
        bm = 256;
        k.mtiles = (int)(M / 256);
    } else
    if (k.patch && !whole) k.mtiles = d->B * ((d->Hin + ph - 1) / ph) * ((d->Win + 15) / 16);
    // statistics from the accumulators (ofx_conv2d_stats): raw outputs only, tiles that stay inside one image.  Called once the tile
    // of the launch is final (the split-bf16 modes remap it below).
    k.stats = nullptr;
    auto setup_stats = [&](int tbm, int tbn) {
        k.stats = nullptr;
        if (!tl_stats.on) return;
        tl_stats.rows_per_image = 0;
        const int waves_m = (tbm == 256 && tbn == 64) ? 4 : (tbm == 128 && (tbn == 128 || tbn == 64 || tbn == 192)) ? 2 : (tbm == 128 && (tbn == 96 || tbn == 32)) ? 4
                            : (tbm == 64 && tbn == 64) ? 2 : 0;
        const long hw = (long)d->Hout * d->Wout;
        const bool ok = waves_m && d->epi == OFX_EPI_PLAIN && d->act == OFX_ACT_NONE && !d->res && nz == 1 &&
                        k.mtiles % d->B == 0 && (k.patch || hw % tbm == 0);
        const long rows = ok ? (long)(k.mtiles / d->B) * waves_m : 0;
        if (ok && (size_t)d->B * rows * d->Cout * 2 <= tl_stats.cap_floats) {
            k.stats = tl_stats.part;
            tl_stats.rows_per_image = (int)rows;
        }
    };
    if (d->precision == OFX_PREC_FP32) setup_stats(bm, bn);
#ifdef OFX_CONV_LEAN   // experiment builds (tools/build_variant.sh): fp32 only -- a third of the instantiations, a third of the compile time
    if (d->precision != OFX_PREC_FP32) return OFX_EINVAL;
#else
    if (d->precision != OFX_PREC_FP32) {
        // split-bf16 matrix-core path (opt-in): three tiles; every other choice is mapped onto them (the ragged
        // N of a 96- or 2-channel layer is zero-filled by the descriptors)
        if (bm == 64) bn = 64;
        else if (bn != 64) bn = 128;
        k.ntiles = (d->Cout + bn - 1) / bn;
        k.group_m = k.ntiles >= 8 ? 8 : 1;
        // bf16x3 on the halo patch (128-row tiles, BK = 16): the fp32 -> (hi, lo) conversion of the A side then runs once per
        // 16-channel slab instead of once per tap
        const bool whole16 = d->Hin % 8 == 0 && d->Win % 16 == 0;
        const double waste16 = (double)(((d->Hin + 7) / 8) * 8) * (double)(((d->Win + 15) / 16) * 16) / ((double)d->Hin * d->Win);
        k.patch = (!no_patch && (whole16 || d->tile != 0 || waste16 <= max_waste) && shape_ok && d->stride == 1 && d->padH == d->KH / 2 && d->padW == d->KW / 2 && d->Hin == d->Hout &&
                   d->Win == d->Wout && k.cin % 16 == 0 && (d->c1 == 0 || d->c0 % 16 == 0) && (!d->nmean || d->c1 == 0) && nz == 1 &&
                   bm == 128 && tile_bk != 32 && d->epi != OFX_EPI_FLOW)
                      ? 1 : 0;
        k.mtiles = k.patch && !whole16 ? d->B * ((d->Hin + 7) / 8) * ((d->Win + 15) / 16) : (int)((M + bm - 1) / bm);
        setup_stats(bm, bn);
        if (tl_pool.on) {   // the blocked correlation volume with pyramid level 1 out of the accumulators, in the split arithmetic
            k.mtiles = (int)((M + 127) / 128);
            k.ntiles = (d->Cout + 127) / 128;
            k.group_m = k.ntiles >= 8 ? 8 : 1;
            k.ksplit = 1;
            if (d->precision == OFX_PREC_BF16X6 || d->precision == OFX_PREC_BF16X6_W) return launch_tile<128, 128, 64, 64, 16, 3>(k, kEpiVolPool, false, nz, s);   // (its B operand is data, never pre-split)
            if (d->precision == OFX_PREC_BF16X3_W) return launch_tile<128, 128, 64, 64, 16, 2>(k, kEpiVolPool, false, nz, s);
            return launch_tile<128, 128, 64, 64, 16, 1>(k, kEpiVolPool, false, nz, s);
        }
        if (d->precision == OFX_PREC_BF16X6_W) {       // bf16x6 with the weights pre-split (ofx_split_conv_weight3)
            if (bm == 128 && bn == 128) return launch_tile<128, 128, 64, 64, 16, 4>(k, d->epi, norm, nz, s);
            if (bm == 128 && bn == 64) return launch_tile<128, 64, 64, 32, 16, 4>(k, d->epi, norm, nz, s);
            if (bm == 64 && bn == 64) return launch_tile<64, 64, 32, 32, 16, 4>(k, d->epi, norm, nz, s);
            return OFX_EINVAL;
        }
        if (d->precision == OFX_PREC_BF16X6) {
            if (bm == 128 && bn == 128) return launch_tile<128, 128, 64, 64, 16, 3>(k, d->epi, norm, nz, s);
            if (bm == 128 && bn == 64) return launch_tile<128, 64, 64, 32, 16, 3>(k, d->epi, norm, nz, s);
            if (bm == 64 && bn == 64) return launch_tile<64, 64, 32, 32, 16, 3>(k, d->epi, norm, nz, s);
            return OFX_EINVAL;
        }
        const bool wsplit = d->precision == OFX_PREC_BF16X3_W;
        if (bm == 128 && bn == 128 && tile_bk == 32 && !wsplit) return launch_tile<128, 128, 64, 64, 32, 1>(k, d->epi, norm, nz, s);
        if (bm == 128 && bn == 128) return wsplit ? launch_tile<128, 128, 64, 64, 16, 2>(k, d->epi, norm, nz, s) : launch_tile<128, 128, 64, 64, 16, 1>(k, d->epi, norm, nz, s);
        if (bm == 128 && bn == 64) return wsplit ? launch_tile<128, 64, 64, 32, 16, 2>(k, d->epi, norm, nz, s) : launch_tile<128, 64, 64, 32, 16, 1>(k, d->epi, norm, nz, s);
        if (bm == 64 && bn == 64) return wsplit ? launch_tile<64, 64, 32, 32, 16, 2>(k, d->epi, norm, nz, s) : launch_tile<64, 64, 32, 32, 16, 1>(k, d->epi, norm, nz, s);
        return OFX_EINVAL;
    }
#endif
    if (tl_pool.on) {
        k.mtiles = (int)((M + 127) / 128);
        k.ntiles = (d->Cout + 127) / 128;
        k.group_m = k.ntiles >= 8 ? 8 : 1;
        static const char* gm_env = getenv("OFX_VOL_GROUP_M");      // raster probe: M-tiles per group of the wide-N volume GEMM
        if (gm_env && atoi(gm_env) > 0) k.group_m = atoi(gm_env);
        k.ksplit = 1;
        return launch_tile<128, 128, 64, 64, 16>(k, kEpiVolPool, false, nz, s);
    }
    if (bm == 256 && bn == 64) return launch_tile<256, 64, 64, 64, 16>(k, d->epi, norm, nz, s);
    if (bm == 128 && bn == 192) return launch_tile<128, 192, 64, 96, 16>(k, d->epi, norm, nz, s);
    if (bm == 128 && bn == 96) return launch_tile<128, 96, 32, 96, 16>(k, d->epi, norm, nz, s);
    if (bm == 128 && bn == 128 && bk == 16) return launch_tile<128, 128, 64, 64, 16>(k, d->epi, norm, nz, s);
    if (bm == 128 && bn == 64 && bk == 16) return launch_tile<128, 64, 64, 32, 16>(k, d->epi, norm, nz, s);
    if (bm == 128 && bn == 128) return launch_tile<128, 128, 64, 64, 32>(k, d->epi, norm, nz, s);
    if (bm == 128 && bn == 64) return launch_tile<128, 64, 64, 32, 32>(k, d->epi, norm, nz, s);
    if (bm == 128 && bn == 32) return launch_tile<128, 32, 32, 32, 32>(k, d->epi, norm, nz, s);
    if (bm == 64 && bn == 64 && bk == 16) return launch_tile<64, 64, 32, 32, 16>(k, d->epi, norm, nz, s);
    // a grid of at most ~2 workgroups per CU is latency-bound: pair the pipelines (tile + 2e9 forces it, an explicit tile without that forbids it)
    const long blocks = (long)k.mtiles * k.ntiles * nz;
    if (k.ksplit > 1) return launch_tile<64, 64, 32, 32, 32, 0, 1, true>(k, d->epi, norm, nz, s);
    // (round 5: up to 320 blocks, not 640 -- at 384 blocks, `convc1` on one 512x768 pair, the plain tile is ahead: 8.76 -> 8.62 ms per
    // pair; OFX_CONV_PAIR_MAX=<blocks> overrides, 0 = never)
    static const char* pair_env = getenv("OFX_CONV_PAIR_MAX");
    const long pair_max = pair_env ? atol(pair_env) : 320;
    const bool pair = d->tile >= 2000000000 || (d->tile < 1000000 && blocks <= pair_max && k.Kpad >= 8 * 32);
    if (bm == 64 && bn == 64 && pair) return launch_tile<64, 64, 32, 32, 32, 0, 2>(k, d->epi, norm, nz, s);
    if (bm == 64 && bn == 64) return launch_tile<64, 64, 32, 32, 32>(k, d->epi, norm, nz, s);
    return OFX_EINVAL;
}

namespace {
inline uint16_t bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40u);   // NaN stays NaN
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
inline float bf16_to_f32(uint16_t h) {
    const uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
}  // namespace

extern "C" int ofx_split_conv_weight(const float* packed, long n, float* out) {
    if (!packed || !out || n <= 0 || (n & 3)) return OFX_EINVAL;
    uint16_t* o = reinterpret_cast<uint16_t*>(out);
    for (long g = 0; g < n; g += 4)
        for (int i = 0; i < 4; ++i) {
            const float x = packed[g + i];
            const uint16_t hi = bf16_rne(x);
            const volatile float rem = x - bf16_to_f32(hi);       // exact in fp32; volatile: no contraction / reassociation
            o[2 * g + i] = hi;
            o[2 * g + 4 + i] = bf16_rne(rem);
        }
    return 0;
}

// packed fp32 weights [n] -> the bf16x6 operand format: n floats of [hi x4 | mid x4] groups (16 bytes per four consecutive k, the layout
// of ofx_split_conv_weight with mid in lo's place), then n / 2 floats of [lo x4] groups (8 bytes per four k).  hi = bf16(x),
// mid = bf16(x - hi), lo = bf16(x - hi - mid), each round-to-nearest-even: the pieces the kernel's on-the-fly split makes
extern "C" int ofx_split_conv_weight3(const float* packed, long n, float* out) {
    if (!packed || !out || n <= 0 || (n & 3)) return OFX_EINVAL;
    uint16_t* o = reinterpret_cast<uint16_t*>(out);
    uint16_t* o2 = o + 2 * n;
    for (long g = 0; g < n; g += 4)
        for (int i = 0; i < 4; ++i) {
            const float x = packed[g + i];
            const uint16_t hi = bf16_rne(x);
            const volatile float r1 = x - bf16_to_f32(hi);        // exact in fp32; volatile: no contraction / reassociation
            const uint16_t mid = bf16_rne(r1);
            const volatile float r2 = r1 - bf16_to_f32(mid);
            o[2 * g + i] = hi;
            o[2 * g + 4 + i] = mid;
            o2[g + i] = bf16_rne(r2);
        }
    return 0;
}

extern "C" long ofx_pack_conv_weight(const float* w, int Cout, int Cin, int KH, int KW, int cin_pad, float* out) {
    if (Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0 || cin_pad < Cin || cin_pad % 4) return OFX_EINVAL;
    const long K = (long)KH * KW * cin_pad;
    const long Kpad = ((K + kKAlign - 1) / kKAlign) * kKAlign;
    if (!out) return Kpad;
    if (!w) return OFX_EINVAL;
    for (long i = 0; i < (long)Cout * Kpad; ++i) out[i] = 0.f;
    for (int o = 0; o < Cout; ++o)
        for (int c = 0; c < Cin; ++c)
            for (int ky = 0; ky < KH; ++ky)
                for (int kx = 0; kx < KW; ++kx)
                    out[(long)o * Kpad + (long)(ky * KW + kx) * cin_pad + c] =
                        w[(((long)o * Cin + c) * KH + ky) * KW + kx];
    return Kpad;
}
