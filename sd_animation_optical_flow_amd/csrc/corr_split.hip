// The all-pairs correlation volume (RAFT/core/corr.py:13-27,52-60) on its own kernel -- the executor's default since round 6 in exact
// fp32 (bit-identical to the generic batched GEMM of conv.hip it replaces), and the opt-in split-bf16 arithmetics of the volume alone
// (`volume_precision`; every convolution of the network stays exact fp32 and so does the headline).
//
//   corr[b, i, j] = <fmap1[b, i, :], fmap2[b, j, :]> / sqrt(256)         (corr.py:52-60)
//   level 1       = avg_pool2d(level 0, 2, 2) over j                     (corr.py:24-26)
//
// Operands arrive in MFMA FRAGMENT ORDER from a small kernel of their own: [32-row block][k-step of 16][plane][lane][16 B] -- a fragment
// of v_mfma_f32_32x32x16_bf16 (8 bf16 per lane) or of four v_mfma_f32_32x32x2_f32 (4 floats per lane) is one contiguous KB, whichever
// side of the product it feeds.  Split forms: fp32 -> bf16 PLANES (hi, lo: "bf16x3" = products hh, hl, lh; hi, mid, lo: "bf16x6" = hh, hm,
// mh, hl, lh, mm -- 3 x 8 mantissa bits, the three dropped products sit below 2^-24 relative).  The 1/sqrt(D) = 2^-4 scale is folded
// into the fmap1 side (exact).
//
// The GEMM is "A-stationary": K is only 256, so a wave keeps ALL of K for its 64 (96) rows of fmap1 in registers (2 - 3 row blocks x
// 16 k-steps x 2 - 3 fragments x 4 VGPRs = 256 - 384 of the 512-entry register file, one wave per SIMD) and streams the 32-column blocks
// of fmap2 past them: per column block and k-step 2 - 3 fragment reads from LDS feed 6 - 24 MFMAs -- a quarter to a third of a fragment
// read per MFMA where the generic 2x2 wave tile needs half of one, and no A-side traffic at all.  The four waves of a workgroup own
// 256 (384) rows and share the column-block stream, which arrives in LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip), one
// 32 - 48 KB stage ahead, one workgroup barrier per column block.  The product is formed TRANSPOSED (C^T = B A^T: the streamed fragment is
// the MFMA's first operand), so a lane's accumulator registers hold four CONSECUTIVE columns j of one row i and the 2x2 average over
// (j_y, j_x) is three in-lane additions -- no DPP; the finished tile is parked in a per-wave LDS staging area and leaves as whole
// 128-byte lines.  Column blocks are walked in QUAD order (the four 4x8 blocks under one 4x8 block of level 1 back to back), so the
// level-1 pieces of a quad collect into whole lines too.  Measurements and the dead ends: DESIGN.md section 0, tools/experiments/README.md.
#include "ofx_internal.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));

constexpr int kKS = 16;          // k-steps of 16: D = 256

// column-block t of the quad-ordered stream -> its level-0 block (by, bx); level-1 block = t >> 2
__host__ __device__ __forceinline__ void quad_block(int t, int wb1, int& by, int& bx) {
    const int q1 = t >> 2, qy = q1 / wb1, qx = q1 - qy * wb1;
    by = 2 * qy + ((t >> 1) & 1);
    bx = 2 * qx + (t & 1);
}

// ------------------------------------------------------------------------------------------
// fp32 rows -> bf16 planes in fragment order
// ------------------------------------------------------------------------------------------
struct SplitArgs {
    const float* src;     // [n][N][256]
    char* dst;            // [n][N / 32][16][NP][64][16 B]
    int N, w, wb1;
    int quad;             // rows in quad-blocked column order (the streamed operand) instead of pixel order
    float alpha;
    int nblk;             // N / 32
};

template <int NP>
__global__ __launch_bounds__(256) void split_planes_kernel(const SplitArgs a) {
    __shared__ __attribute__((aligned(16))) float tile[32][260];
    const int tid = threadIdx.x;
    const int blk = blockIdx.x % a.nblk, img = blockIdx.x / a.nblk;
    int by = 0, bx = 0;
    if (a.quad) quad_block(blk, a.wb1, by, bx);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = i * 256 + tid, row = idx >> 6, c4 = idx & 63;
        const int pix = a.quad ? ((by << 2) + (row >> 3)) * a.w + (bx << 3) + (row & 7) : blk * 32 + row;
        f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(a.src + ((long)img * a.N + pix) * 256) + c4);
        v *= a.alpha;
        *reinterpret_cast<f4v*>(&tile[row][c4 * 4]) = v;
    }
    __syncthreads();
    char* dst = a.dst + ((long)img * a.nblk + blk) * (kKS * NP * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = i * 256 + tid, ks = j >> 6, ln = j & 63;
        const float* s = &tile[ln & 31][ks * 16 + (ln >> 5) * 8];
        const f4v x0 = *reinterpret_cast<const f4v*>(s), x1 = *reinterpret_cast<const f4v*>(s + 4);
        // round-to-nearest-even at every step (v_cvt_pk_bf16_f32): x = hi + (mid +) lo + O(2^-17 |x|) / exactly (three planes)
        f4v r0 = x0, r1 = x1;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const bf16x4 h0 = __builtin_convertvector(r0, bf16x4), h1 = __builtin_convertvector(r1, bf16x4);
            r0 -= __builtin_convertvector(h0, f4v);
            r1 -= __builtin_convertvector(h1, f4v);
            bf16x8 o;
            o[0] = h0[0]; o[1] = h0[1]; o[2] = h0[2]; o[3] = h0[3];
            o[4] = h1[0]; o[5] = h1[1]; o[6] = h1[2]; o[7] = h1[3];
            *reinterpret_cast<bf16x8*>(dst + ((ks * NP + p) * 64 + ln) * 16) = o;
        }
    }
}

// fp32 rows in the fragment order of v_mfma_f32_32x32x2_f32 (the exact-fp32 form of the same GEMM): per 32-row block 32 groups of
// [lane][4 floats] -- lane l holds, for row l & 31, k = 8g + 4 (l >> 5) + slot, slot = 0..3: one 16-byte fragment read feeds four MFMAs,
// MFMA `slot` multiplies the k pair (8g + slot, 8g + 4 + slot).  That is the k assignment of the generic fp32 kernel (conv.hip: a lane
// half reads the float4 at k offset 4 (lane >> 5) of every 8-wide group), hence the same summation order and bit-identical sums.
__global__ __launch_bounds__(256) void frag_order_f32_kernel(const SplitArgs a) {
    __shared__ __attribute__((aligned(16))) float tile[32][260];
    const int tid = threadIdx.x;
    const int blk = blockIdx.x % a.nblk, img = blockIdx.x / a.nblk;
    int by = 0, bx = 0;
    if (a.quad) quad_block(blk, a.wb1, by, bx);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = i * 256 + tid, row = idx >> 6, c4 = idx & 63;
        const int pix = a.quad ? ((by << 2) + (row >> 3)) * a.w + (bx << 3) + (row & 7) : blk * 32 + row;
        f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(a.src + ((long)img * a.N + pix) * 256) + c4);
        v *= a.alpha;
        *reinterpret_cast<f4v*>(&tile[row][c4 * 4]) = v;
    }
    __syncthreads();
    char* dst = a.dst + ((long)img * a.nblk + blk) * (32 * 1024);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int j = i * 256 + tid, g = j >> 6, ln = j & 63;
        *reinterpret_cast<f4v*>(dst + (g * 64 + ln) * 16) = *reinterpret_cast<const f4v*>(&tile[ln & 31][8 * g + 4 * (ln >> 5)]);
    }
}

// ------------------------------------------------------------------------------------------
// the GEMM
// ------------------------------------------------------------------------------------------
struct VolArgs {
    const char* ap;       // fmap1 planes (pixel order, scaled), fragment order
    const char* bp;       // fmap2 planes (quad-blocked order), fragment order
    const int* ia;        // optional device arrays: image of pair z on either side
    const int* ib;
    long a_zs, b_zs;      // bytes from pair z to pair z + 1 when there is no index array (0: one shared image)
    long img_bytes;       // bytes of one image's planes
    float* l0;
    float* l1;
    long l0_zs, l1_zs;    // floats per pair
    int N, T, G, nz, ntask;
    int wb0, wb1, slice1;
    int swz;              // XCD-grouped task order (ntask % 8 == 0)
    int stagger;          // quads between the stream starts of consecutive tasks (0: everyone starts at column block 0)
    int dbg;              // diagnostic (OFX_VOLSPLIT_DBG): 1 = drop every store, 2 = no fetches, 4 = drop level-1 stores
};

// four 1 KB LDS-DMA pieces: LDS [m0 + i * 1024 + lane * 16] <- global [sbase + i * 1024 + voff]  (the instruction offset advances both
// sides).  M0 is saved and restored: the compiler does not know this block touches it.
__device__ __forceinline__ void dma4(const char* sbase, unsigned voff, unsigned lds_addr) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:3072\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_addr)
        : "memory");
}
// one piece, OFS (0 / 1024 / 2048 / 3072) bytes into both sides
template <int OFS>
__device__ __forceinline__ void dma1(const char* sbase, unsigned voff, unsigned lds_addr) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:%4\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_addr), "n"(OFS)
        : "memory");
}

template <int NP, int RB, bool DB, bool F32>
__global__ __launch_bounds__(256, 1) void corr_vol_split_kernel(const VolArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int STAGE = kKS * NP * 1024;       // one column block of the streamed operand
    constexpr int WSH = STAGE / 4;               // the share of it one wave fetches
    constexpr int STG = 2 * STAGE;               // behind the two stages: output staging per wave (level 0 | level 1)
    constexpr int SLV = RB * 4096;               // bytes of one staging level: the wave's RB * 32 rows x 128 B
    static_assert(RB == 2 || (RB == 3 && !DB), "row blocks per wave: 2, or 3 with a single accumulator set");
    // F32: the exact-fp32 form on v_mfma_f32_32x32x2_f32 -- the same kernel, a "plane" is then a group of four k-pairs in fp32 (a
    // column block is 32 KB like two bf16 planes: NP = 2), a fragment read feeds 4 MFMAs per row block
    static_assert(!F32 || NP == 2, "fp32 form: two 16-byte fragments per k-step");
    typedef typename std::conditional<F32, f4v, bf16x8>::type frag_t;
    constexpr int SAUX = 0;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int task = blockIdx.x;
    if (a.swz) task = (task & 7) * (a.ntask >> 3) + (task >> 3);    // block b runs on XCD b % 8: an XCD walks consecutive tasks
    const int z = task / a.G, g = task - z * a.G;
    const int row0 = g * (RB * 128) + wave * (RB * 32);
    const char* abase = a.ap + (a.ia ? (long)a.ia[z] * a.img_bytes : (long)z * a.a_zs);
    const char* bbase = a.bp + (a.ib ? (long)a.ib[z] * a.img_bytes : (long)z * a.b_zs);
    typedef __attribute__((address_space(3))) char* lds_p;
    const unsigned lds0 = (unsigned)(size_t)(lds_p)lds;

    // ---- the stationary operand: RB * 32 rows x K = 256 x NP planes.  Row blocks past the map (a last, partial row group) hold
    // zeros and their stores are dropped; the wave still takes part in the fetches and barriers
    frag_t A[RB][kKS][NP];
    int rbkill[RB];
    {
        const char* src = abase + (long)(row0 >> 5) * STAGE + lane * 16;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const bool in = row0 + rb * 32 < a.N;        // wave-uniform
            rbkill[rb] = in ? 0 : (int)0x80000000;
#pragma unroll
            for (int ks = 0; ks < kKS; ++ks)
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    A[rb][ks][p] = in ? *reinterpret_cast<const frag_t*>(src + ((rb * kKS + ks) * NP + p) * 1024) : frag_t{};
        }
    }

    // ---- the way out.  The tile comes out of the MFMAs transposed: lane (m = lane & 31, hl = lane >> 5) holds, for row m of a row
    // block, columns 8q + 4hl .. + 3 in registers 4q .. 4q + 3 -- a store straight from there touches 32 rows x 32 bytes, four partial
    // writes per 128-byte line (measured: 6.7 ms for the bf16x6 kernel, 4.9 with the stores dropped).  So a finished tile is parked in
    // the wave's own LDS staging area, row-major with the 16-byte chunks of a row XOR-swizzled by (row & 7) (conflict-free both ways),
    // and read back with eight lanes per row: every store instruction then writes eight WHOLE lines.  Level 1 collects the 2x2
    // averages of the four column blocks of a quad in a second staging area and leaves as whole lines once per quad.  Staging doubles
    // as the accumulators' second buffer: the read-back and the stores run under the MFMAs of the next column block.
    constexpr int kOOB = (int)0x80000000;
    const __amdgpu_buffer_rsrc_t rs0 =
        __builtin_amdgcn_make_buffer_rsrc((void*)(a.l0 + (long)z * a.l0_zs), (short)0, (int)((long)a.N * a.N * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 =
        __builtin_amdgcn_make_buffer_rsrc((void*)(a.l1 + (long)z * a.l1_zs), (short)0, (int)((long)a.N * a.slice1 * 4), 0x00020000);
    const int m = lane & 31, hl = lane >> 5, rr = lane >> 3, cc = lane & 7;
    char* const stg = lds + STG + wave * (2 * SLV);
    // staging write: row m, chunk (2q + hl) ^ (m & 7) = the q = 0 address with bits 5, 6 flipped by q
    const int sw0 = m * 128 + ((hl ^ (m & 7)) << 4);
    // level-1 staging write: row m, chunk ((2 sy + py) * 2 + sx) ^ (m & 7), 8 bytes at 8 hl
    const int sw1 = SLV + m * 128 + ((m & 7) << 4) + 8 * hl;
    // staging read: row rr + 8k (k = 0..3), chunk cc: eight lanes cover a row
    const int sr = rr * 128 + ((cc ^ rr) << 4);
    const int rowr = row0 + rr;
    const int vg0 = !(a.dbg & 1) ? (rowr * a.N + cc * 4) * 4 : kOOB;
    const int vg1 = !(a.dbg & 5) ? (rowr * a.slice1 + cc * 4) * 4 : kOOB;
    const int step0 = 8 * a.N * 4, step1 = 8 * a.slice1 * 4;      // eight rows down, bytes

    // k-step at which piece e of a parked tile is read back (it is stored one k-step later): the way out is paced over the whole block
    static constexpr auto rd_step = [](int e) constexpr {
        if (RB == 3) return (e * 5) / 4;                        // 12 pieces: 0 1 2 3 5 6 7 8 10 11 12 13
        const int r = (DB ? 1 : 0) + 2 * e;                     // 8 pieces: every second k-step (two-set form: after the park)
        return r < 14 ? r : 14;
    };
    // park a finished tile (sub = its place in the quad): pieces e = 0 .. 4 RB - 1 level 0, then 2 RB pieces of 2x2 averages
    constexpr int NP0 = 4 * RB, NPK = 6 * RB;
    auto park = [&](const f32x16 (&acc)[RB], int e, int sub) __attribute__((always_inline)) {
        if (e < NP0) {
            const int rb = e >> 2, q = e & 3;
            f4v t = {acc[rb][4 * q + 0], acc[rb][4 * q + 1], acc[rb][4 * q + 2], acc[rb][4 * q + 3]};
            *reinterpret_cast<f4v*>(stg + rb * 4096 + (sw0 ^ (q << 5))) = t;
        } else {
            // registers 4q + c: block row q, block column 4hl + c.  2x2 average: rows (2py, 2py + 1), columns (2c', 2c' + 1)
            const int rb = (e - NP0) >> 1, py = (e - NP0) & 1, u = 8 * py;
            const int chunk = ((2 * (sub >> 1) + py) << 1) + (sub & 1);
            f2v t;
            t[0] = ((acc[rb][u + 0] + acc[rb][u + 1]) + (acc[rb][u + 4] + acc[rb][u + 5])) * 0.25f;
            t[1] = ((acc[rb][u + 2] + acc[rb][u + 3]) + (acc[rb][u + 6] + acc[rb][u + 7])) * 0.25f;
            *reinterpret_cast<f2v*>(stg + rb * 4096 + (sw1 ^ (chunk << 4))) = t;
        }
    };
    // read piece e (row block e >> 2, rows 8 (e & 3) ..) of a staging area back / store it as whole lines
    auto unpark = [&](int e, int lvl) __attribute__((always_inline)) -> v4i {
        return *reinterpret_cast<const v4i*>(stg + lvl * SLV + (e >> 2) * 4096 + (e & 3) * 1024 + sr);
    };
    auto send = [&](v4i t, int e, int lvl, int so, int kill) __attribute__((always_inline)) {
        if (lvl == 0) __builtin_amdgcn_raw_buffer_store_b128(t, rs0, vg0 | kill | rbkill[e >> 2], so + e * step0, SAUX);
        else __builtin_amdgcn_raw_buffer_store_b128(t, rs1, vg1 | kill | rbkill[e >> 2], so + e * step1, SAUX);
    };

    // ---- the stream
    constexpr int NPIECE = WSH / 1024;           // 1 KB LDS-DMA pieces per wave and column block
    auto fetch = [&](int t, int stage) __attribute__((always_inline)) {
        if (a.dbg & 2) return;
        const char* g = bbase + (long)t * STAGE + wave * WSH;
        const unsigned l = lds0 + stage * STAGE + wave * WSH;
#pragma unroll
        for (int i = 0; i < WSH / 4096; ++i) dma4(g + i * 4096, lane * 16, l + i * 4096);
    };
    // piece i of the next stage (g / l: this wave's share of it), issued at the head of k-step i: a burst of all pieces in front of the
    // first MFMA costs ~100 idle matrix-pipe cycles per piece
    auto fetch_piece = [&](int i, const char* g, unsigned l) __attribute__((always_inline)) {
        const char* gb = g + (i >> 2) * 4096;
        const unsigned lb = l + (i >> 2) * 4096;
        switch (i & 3) {
            case 0: dma1<0>(gb, lane * 16, lb); break;
            case 1: dma1<1024>(gb, lane * 16, lb); break;
            case 2: dma1<2048>(gb, lane * 16, lb); break;
            default: dma1<3072>(gb, lane * 16, lb); break;
        }
    };
    // One column block: 16 k-steps of NP fragment reads (issued a k-step ahead) and 4 * NP MFMAs into `acc`.  Spread over the first
    // k-steps: (DB) the previous tile `prev` (place PS in its quad) is parked; its level-0 lines are read back and stored at so0; when
    // it closed a quad (PS == 3) the quad's level-1 lines follow at so1.  kill = kOOB: there is no previous tile.
    auto compute = [&](f32x16 (&acc)[RB], const f32x16 (&prev)[RB], int stage, int CS, int PS, int so0, int so1, int kill, bool more,
                       const char* ng, unsigned nl) __attribute__((always_inline)) {
        const char* s = lds + stage * STAGE + lane * 16;
        frag_t B[2][NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) B[0][p] = *reinterpret_cast<const frag_t*>(s + p * 1024);
        v4i hold[2];
        // the way out is paced: HBM takes a column block's 40 KB per CU no faster than the MFMAs produce it, and a store that cannot
        // issue holds up the MFMAs behind it (one wave per SIMD issues in order) -- so one read-back / store per two k-steps, spread over
        // the whole block, not a burst behind the park
        auto rd_at = [](int e) { return rd_step(e); };
#pragma unroll
        for (int ks = 0; ks < kKS; ++ks) {
            const int c = ks & 1;
            if (ks < NPIECE && more) fetch_piece(ks, ng, nl);
            if (ks + 1 < kKS) {
#pragma unroll
                for (int p = 0; p < NP; ++p) B[c ^ 1][p] = *reinterpret_cast<const frag_t*>(s + ((ks + 1) * NP + p) * 1024);
            }
            // products smallest first; the streamed fragment is the FIRST operand: the tile comes out transposed (rows = columns j)
            auto prod = [&](int rb, int i) __attribute__((always_inline)) {
                const f32x16 zero = {};
                const bool first = ks == 0 && i == 0;     // the first product of a tile starts from zero
                if constexpr (F32) {
                    // fragment i >> 2 (eight k), k pair i & 3: the summation order of the generic fp32 kernel (bit-identical results)
                    acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(B[c][i >> 2][i & 3], A[rb][ks][i >> 2][i & 3], first ? zero : acc[rb], 0, 0, 0);
                } else {
                    constexpr int PB3[6] = {2, 1, 0, 1, 0, 0}, PA3[6] = {0, 1, 2, 0, 1, 0}, PB2[3] = {1, 0, 0}, PA2[3] = {0, 1, 0};
                    const int pb = NP == 3 ? PB3[i] : PB2[i], pa = NP == 3 ? PA3[i] : PA2[i];
                    acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(B[c][pb], A[rb][ks][pa], first ? zero : acc[rb], 0, 0, 0);
                }
            };
            constexpr int NPROD = F32 ? 8 : NP == 3 ? 6 : 3;
            // the two row blocks alternate: consecutive MFMAs never wait for each other's accumulator.  (Parking the single-accumulator
            // form's tile in two halves, each under the other row block's MFMAs of the first / last k-step, measured SLOWER -- 5.8 against
            // 5.65 ms: six back-to-back MFMAs on one accumulator cost more than the park they cover.)
#pragma unroll
            for (int i = 0; i < NPROD; ++i)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) prod(rb, i);
            if constexpr (DB) {
                if (ks < RB) {    // park row block ks of the previous tile
#pragma unroll
                    for (int q = 0; q < 4; ++q) park(prev, 4 * ks + q, PS);
                    park(prev, NP0 + 2 * ks, PS);
                    park(prev, NP0 + 1 + 2 * ks, PS);
                }
            }
#pragma unroll
            for (int e = 0; e < NP0; ++e) {
                if (ks == rd_at(e) + 1) {
                    send(hold[0], e, 0, so0, kill);
                    if (PS == 3) send(hold[1], e, 1, so1, kill);
                }
            }
#pragma unroll
            for (int e = 0; e < NP0; ++e) {
                if (ks == rd_at(e)) {
                    hold[0] = unpark(e, 0);
                    if (PS == 3) hold[1] = unpark(e, 1);
                }
            }
            // pin the k-step: left alone, the scheduler sinks every store of the previous tile behind the last MFMA of this one,
            // straight in front of the wait that guards the next stage
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // scalar offsets of column block t: its level-0 block (bytes into a row's slice) and its quad's level-1 line
    // (diagnostic, a.stagger > 0: workgroups start at different quads of the stream and wrap around)
    const int quads = a.T >> 2;
    const int tq0 = a.stagger ? (task * a.stagger) % quads : 0;
    int qy = tq0 / a.wb1, qx = tq0 - qy * a.wb1;
    auto offsets = [&](int t, int& so0, int& so1) __attribute__((always_inline)) {
        const int sy = (t >> 1) & 1, sx = t & 1;
        so0 = (((2 * qy + sy) * a.wb0 + 2 * qx + sx) << 7);
        so1 = (t >> 2) << 7;
        if ((t & 3) == 3 && ++qx == a.wb1) { qx = 0; ++qy; }
    };

    if (a.dbg & 32) {   // probe: de-phase the workgroups
        for (int i = 0; i < (task & 15); ++i) __builtin_amdgcn_s_sleep(3);
    }
    fetch(tq0 * 4, 0);
    f32x16 acc0[RB], acc1[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc0[rb][e] = acc1[rb][e] = 0.0f;
    int p0 = 0, p1 = 0, c0 = 0, c1 = 0;     // offsets of the tile on its way out / of the current tile
    // Vector-memory operations retire in order, and behind the last fetch piece of the next stage a column block issues exactly
    // kBehind level-0 stores (twice that when it also flushes a quad's level 1): waiting for "all but those" means the fetch has landed
    // while the stores stay in flight under the next block's MFMAs (vmcnt(0) here serialises the write stream with the arithmetic).
    // level-0 stores of a block behind its last fetch piece (the piece goes out at the head of k-step NPIECE - 1, store e at the end
    // of k-step min(R0 + 2e, 14) + 1)
    constexpr int kBehind = [] {
        int n = 0;
        for (int e = 0; e < NP0; ++e) n += (rd_step(e) + 1 >= NPIECE - 1) ? 1 : 0;
        return n;
    }();
    auto block = [&](auto sub_tag, int t, int tn, bool first, bool last) __attribute__((always_inline)) {
        constexpr int SUB = decltype(sub_tag)::value, PS = (SUB + 3) & 3;
        if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if constexpr (PS == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * kBehind) : "memory");     // the block before flushed a quad
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kBehind) : "memory");
        __builtin_amdgcn_s_barrier();                          // every wave's share has; and nobody reads the other stage any more
        offsets(t, c0, c1);
        const int kill = first ? kOOB : 0;
        const char* ng = bbase + (long)tn * STAGE + wave * WSH;
        const unsigned nl = lds0 + ((SUB + 1) & 1) * STAGE + wave * WSH;
        // (the last block fetches the stream's next block too: a harmless read into the free stage, instead of a branch per k-step)
        constexpr bool more = true;
        (void)last;
        if constexpr (DB) {
            if constexpr (SUB & 1) compute(acc1, acc0, 1, SUB, PS, p0, p1, kill, more, ng, nl);
            else compute(acc0, acc1, 0, SUB, PS, p0, p1, kill, more, ng, nl);
        } else {
            compute(acc0, acc0, SUB & 1, SUB, PS, p0, p1, kill, more, ng, nl);
#pragma unroll
            for (int e = 0; e < NPK; ++e) park(acc0, e, SUB);
        }
        p0 = c0; p1 = c1;
    };
    int t = tq0 * 4;
    for (int i = 0; i < quads; ++i) {       // T % 4 == 0: whole quads
        int tn = t + 4;
        if (tn == a.T) { tn = 0; }
        block(std::integral_constant<int, 0>{}, t, t + 1, i == 0, false);
        block(std::integral_constant<int, 1>{}, t + 1, t + 2, false, false);
        block(std::integral_constant<int, 2>{}, t + 2, t + 3, false, false);
        block(std::integral_constant<int, 3>{}, t + 3, tn, false, i == quads - 1);
        if (tn == 0) { qx = 0; qy = 0; }
        t = tn;
    }
    // the last tile (it closes a quad)
    if constexpr (DB) {
#pragma unroll
        for (int e = 0; e < NPK; ++e) park(acc1, e, 3);
    }
#pragma unroll
    for (int lvl = 0; lvl < 2; ++lvl)
#pragma unroll
        for (int e = 0; e < NP0; ++e) send(unpark(e, lvl), e, lvl, lvl ? p1 : p0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA piece may still be on its way when the workgroup's LDS is handed on
}

// Row blocks per wave (2: 256-row groups, 3: 384-row groups; bf16x6 has the registers for 2 only) and how well the tasks -- one
// workgroup per (pair, row group), one workgroup per CU at a time -- fill the 256 CUs: tasks / (rounds * 256)
int pick_rb(int nz, long N, int planes, double* fill) {
    auto eff = [&](int rb) {
        const long t = (long)nz * ((N + rb * 128 - 1) / (rb * 128));
        return (double)t / (double)(((t + 255) / 256) * 256);
    };
    static const char* v = getenv("OFX_VOLSPLIT_VARIANT");   // diagnostic: "db" / "nodb" (64 rows per wave), "r3" (96)
    int rb = 2;
    if (planes != 3 && N % 384 == 0 && (v ? v[0] == 'r' : eff(3) >= eff(2) - 0.05)) rb = 3;
    if (fill) *fill = eff(rb);
    return rb;
}

// planes: 1 = exact fp32, 2 = bf16x3, 3 = bf16x6
int launch_vol(VolArgs a, int planes, hipStream_t s) {
    // Row blocks per wave and accumulator sets.  bf16x6 keeps 384 registers of operand for 64 rows: one accumulator set, the tile is
    // parked behind its last MFMA.  bf16x3 and fp32 (256 registers per 64 rows) have the registers either for two sets on 64 rows or for
    // 96 rows with one set: 96 rows win (1.5x the MFMAs per fragment read, a third fewer column-stream bytes and barriers) wherever
    // 384-row groups tile the map.
    static const char* v = getenv("OFX_VOLSPLIT_VARIANT");
    const int rb = pick_rb(a.nz, a.N, planes, nullptr);
    const bool db = planes == 2 && rb == 2 && (v ? v[0] == 'd' : true);
    const int np = planes == 3 ? 3 : 2;
    a.G = (a.N + rb * 128 - 1) / (rb * 128);
    a.ntask = a.nz * a.G;
    a.swz = (a.ntask % 8 == 0) ? 1 : 0;
    const size_t ldsb = 2 * kKS * np * 1024 + 4 * rb * 8192;
    auto go = [&](auto kern) {
        // > 64 KB of dynamic LDS: allowed once per instantiation (a generic lambda's body is one per kernel type) and device
        static size_t allowed[64] = {};
        int dev = 0;
        (void)hipGetDevice(&dev);
        dev &= 63;
        if (allowed[dev] < ldsb) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
            allowed[dev] = ldsb;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)a.ntask), dim3(256), ldsb, s, a);
    };
    if (planes == 1) {
        if (rb == 3) go(corr_vol_split_kernel<2, 3, false, true>);
        else go(corr_vol_split_kernel<2, 2, false, true>);
    } else if (planes == 2) {
        if (rb == 3) go(corr_vol_split_kernel<2, 3, false, false>);
        else if (db) go(corr_vol_split_kernel<2, 2, true, false>);
        else go(corr_vol_split_kernel<2, 2, false, false>);
    } else {
        go(corr_vol_split_kernel<3, 2, false, false>);
    }
    return ofx_launch_status();
}

}  // namespace

// ---- internal entry points (the RAFT executor) ---------------------------------------------------------------------------

bool ofx_corr_volsplit_ok(int h, int w, int D) {
    const long N = (long)h * w;
    return D == 256 && h % 8 == 0 && w % 16 == 0 && N >= 64 && N * N * 4 < (1L << 31) - 64;
}

// A task is one workgroup on one CU for ~(rows / 128) x 0.75 ms: the kernel pays when its tasks fill the part's rounds.  A single
// 512x768 pair is 16-24 tasks on 256 CUs (2.2 ms where the generic GEMM's 2304 tiles take 0.2): the executor keeps the generic
// kernel for those (same bits in fp32).
bool ofx_corr_volsplit_pays(int nz, int h, int w, int planes) {
    double fill = 0.0;
    pick_rb(nz, (long)h * w, planes, &fill);
    // against the generic fp32 GEMM (10.5 ms per 64 pairs at 512x768, any batch) a full part of this kernel takes 8.7 / 5.3 / 3.3 ms
    // (fp32 / bf16x6 / bf16x3): it wins from a fill of 8.7 / 10.5 ... on
    return fill >= (planes == 1 ? 0.84 : planes == 3 ? 0.52 : 0.35);
}

size_t ofx_corr_planes_bytes(int h, int w, int planes) { return (size_t)h * w / 32 * kKS * (planes == 3 ? 3 : 2) * 1024; }   // fp32 (1): 32 KB per 32 rows, like two bf16 planes

int ofx_corr_split_planes(const float* src, void* dst, int n, int h, int w, int planes, int quad, float alpha, hipStream_t s) {
    OFX_REQUIRE(src && dst && n > 0 && planes >= 1 && planes <= 3 && ofx_corr_volsplit_ok(h, w, 256), OFX_EINVAL);
    OFX_REQUIRE(ofx_aligned16(src) && ofx_aligned16(dst), OFX_EALIGN);
    SplitArgs a{};
    a.src = src; a.dst = (char*)dst; a.N = h * w; a.w = w; a.wb1 = w / 16; a.quad = quad; a.alpha = alpha; a.nblk = a.N / 32;
    OfxProfScope prof("corr_split_planes", s);
    if (planes == 3) hipLaunchKernelGGL(split_planes_kernel<3>, dim3((unsigned)(n * a.nblk)), dim3(256), 0, s, a);
    else if (planes == 2) hipLaunchKernelGGL(split_planes_kernel<2>, dim3((unsigned)(n * a.nblk)), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(frag_order_f32_kernel, dim3((unsigned)(n * a.nblk)), dim3(256), 0, s, a);
    return ofx_launch_status();
}

// level 0 (blocked) and level 1 of nz pairs from pre-split planes.  ia / ib: device arrays of image indices, or null with byte strides
int ofx_corr_vol_split_launch(const void* ap, const void* bp, const int* ia, const int* ib, long a_zs, long b_zs, float* l0, float* l1,
                              int nz, int h, int w, int planes, hipStream_t s) {
    OFX_REQUIRE(ap && bp && l0 && l1 && nz > 0 && planes >= 1 && planes <= 3 && ofx_corr_volsplit_ok(h, w, 256), OFX_EINVAL);
    VolArgs a{};
    a.ap = (const char*)ap; a.bp = (const char*)bp; a.ia = ia; a.ib = ib; a.a_zs = a_zs; a.b_zs = b_zs;
    a.N = h * w; a.T = a.N / 32; a.nz = nz;
    a.img_bytes = (long)ofx_corr_planes_bytes(h, w, planes);
    a.wb0 = w / 8; a.wb1 = w / 16; a.slice1 = ofx_corr_slice_floats_l(h >> 1, w >> 1);
    a.l0 = l0; a.l1 = l1; a.l0_zs = (long)a.N * a.N; a.l1_zs = (long)a.N * a.slice1;
    static const char* stg = getenv("OFX_VOLSPLIT_STAGGER");
    // 0: every workgroup starts the stream at column block 0.  Staggered starts (OFX_VOLSPLIT_STAGGER=5: tried against a suspected
    // channel hot spot of rows that lie a multiple of 8 KB apart) changed no timing -- and cost the XCD's L2 its hit rate on the stream:
    // with them every workgroup fetched the whole of fmap2's planes from beyond the L2 (PMC: 6.9 GB of fetches per fp32 launch,
    // 15.2 GB in bf16x6 form, for 0.4 GB of operands)
    a.stagger = stg ? atoi(stg) : 0;
    static const char* dbg = getenv("OFX_VOLSPLIT_DBG");
    a.dbg = dbg ? atoi(dbg) : 0;
    OfxProfScope prof(planes == 3 ? "corr_vol_split6" : planes == 2 ? "corr_vol_split3" : "corr_vol_f32", s);
    prof.flops(2.0 * nz * (double)a.N * a.N * 256.0 * (planes == 3 ? 6 : planes == 2 ? 3 : 1));
    return launch_vol(a, planes, s);
}

extern "C" {

int ofx_corr_volume_split(const float* f1, const float* f2, float* const* pyr, int B, int h, int w, int D, int levels, int planes,
                          int shared_f2, void* stream) {
    OFX_REQUIRE(f1 && f2 && pyr && B > 0 && h > 0 && w > 0, OFX_EINVAL);
    OFX_REQUIRE(levels >= 2 && levels <= 4 && planes >= 1 && planes <= 3, OFX_EINVAL);
    OFX_REQUIRE(ofx_corr_volsplit_ok(h, w, D), OFX_EINVAL);
    for (int l = 0; l < levels; ++l) OFX_REQUIRE(pyr[l] != nullptr && ofx_aligned16(pyr[l]), OFX_EINVAL);
    OFX_REQUIRE(ofx_aligned16(f1) && ofx_aligned16(f2), OFX_EALIGN);
    hipStream_t s = (hipStream_t)stream;
    const size_t ib = ofx_corr_planes_bytes(h, w, planes);
    const int n2 = shared_f2 ? 1 : B;
    char* pl = nullptr;   // stream-ordered scratch: the planes of both operands
    OFX_HIP_CHECK(hipMallocAsync((void**)&pl, ib * (size_t)(B + n2), s));
    int st = ofx_corr_split_planes(f1, pl, B, h, w, planes, 0, 1.0f / sqrtf((float)D), s);
    if (!st) st = ofx_corr_split_planes(f2, pl + ib * B, n2, h, w, planes, 1, 1.0f, s);
    if (!st) st = ofx_corr_vol_split_launch(pl, pl + ib * B, nullptr, nullptr, (long)ib, shared_f2 ? 0 : (long)ib, pyr[0], pyr[1], B, h, w, planes, s);
    const hipError_t fe = hipFreeAsync(pl, s);
    if (st) return st;
    if (fe != hipSuccess) return (int)fe;
    return ofx_corr_pool_launch(pyr[0], pyr[1], levels > 2 ? pyr[2] : nullptr, levels > 3 ? pyr[3] : nullptr, B, h, w, levels, s, true);
}

}  // extern "C"
