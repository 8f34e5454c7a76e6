// The all-pairs correlation volume (RAFT/core/corr.py:52-60) as a dedicated, PERSISTENT batched GEMM on the fp32 matrix cores:
//
//     vol[z] (N x Nb) = fmap1[z] (N x K) * fmap2b[z]^T (K x Nb) * alpha        K = 256, alpha = 1 / sqrt(K)
//
// with the columns in the blocked order of the pyramid (the rows of fmap2b are permuted once, corr.hip:ofx_corr_block_rows) and,
// when level 1 is tiled by whole blocks, the 2x2 average of every 4x8 block written as pyramid level 1 from the accumulators.
//
// Why its own kernel (round 5).  Through `igemm_kernel` this GEMM ran at 0.75 of the fp32 MFMA peak where the long-K convolutions
// reach 0.92: K = 256 is sixteen 16-wide chunks, so every 128x128 tile pays a pipeline fill (two dependent global-load latencies
// before the first MFMA) and an 80 KB epilogue for only sixteen chunks of matrix work -- with the stores removed the old kernel
// still took 9.74 of its 10.4 ms against 8.5 ms of pure MFMA time.  Here a workgroup is resident for the whole launch and walks its
// tile list with ONE software pipeline that runs across tile boundaries: while the last chunks of tile t are multiplied the first
// chunk of tile t + 1 is already committed to LDS and its second is in flight, so the epilogue of tile t (stores only, no LDS)
// is followed by an MFMA block at once -- no fill, and the other workgroups of the CU multiply while this one stores.
//
// Same arithmetic as before, bit for bit: v_mfma_f32_32x32x2_f32, chunks in ascending k, the same k permutation inside a chunk,
// one multiply by alpha -- the parity tests of the volume do not move.
//
// Tiling: 128x128 tile, 4 waves of 64x64 (2x2 MFMA tiles of 32x32), BK = 16, LDS rows of 20 floats (conflict-free b128 fragment
// reads), two stages = 40 KB -> four workgroups per CU; operands through buffer descriptors (rows past N / Nb read as zero).
// Tile order: z-major, inside a z groups of 8 M-tiles walked M-fastest (the 128 workgroups an XCD runs side by side share 8 A
// tiles and 16 B tiles = 3 MB of its 4 MB L2); XCD x owns a contiguous eighth of the order, its resident workgroups interleave.
#include "ofx_internal.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace {

struct VolK {
    const float* A; long a_zs;          // [nz][N][K]   (a_zs = 0: one shared image-1 feature map)
    const float* Bm; long b_zs;         // [nz][Nb][K]  rows in blocked order (b_zs = 0: one shared key frame)
    float* out; long o_zs;              // [nz][N][Nb]
    float* pool; long pool_zs;          // [nz][N][slice1] or null
    int N, Nb, K, nz;
    int mtiles, ntiles, group_m;
    int wb0, wb1, slice1;
    float alpha;
    int total;                          // tiles of the launch
    int wgs;                            // resident workgroups (grid size), a multiple of 8
    int stagger, stagger_mode;          // start-phase probe: workgroup phase x `stagger` x 64 cycles of s_sleep before the first load
};

constexpr int BM = 128, BN = 128, BK = 16, LDK = BK + 4;
constexpr int STAGE = (BM + BN) * LDK;            // floats per stage
constexpr int kOOB = 0x7FFFFFF0;

template <bool POOL>
__global__ __launch_bounds__(256, 4) void vol_gemm_kernel(const VolK p) {
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];
    typedef int v4i __attribute__((ext_vector_type(4)));
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // staging: thread -> (row group, float4 slot); rows permuted so that the 16 lanes of one ds_write_b128 pass hit 16 distinct
    // 16-byte bank groups (rows 4 apart inside a group of 16: conv.hip has the derivation)
    const int kq = tid & 3, j0 = tid >> 2;
    const int r0 = (j0 & 3) * 4 + ((j0 >> 2) & 3) + (j0 >> 4) * 16;              // 0..63; the thread also stages row r0 + 64
    const int frow = lane & 31, fk = (lane >> 5) * 4;

    // this workgroup's tiles: XCD x (= block id mod 8, where the hardware puts the block) owns the contiguous range
    // [x0, x0 + cnt) of the global order; its workgroups take local indices slot, slot + per_xcd, ...
    const int xcd = (int)(blockIdx.x & 7), slot = (int)(blockIdx.x >> 3), per_xcd = p.wgs >> 3;
    const int q8 = p.total >> 3, r8 = p.total & 7;
    const int x0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int cnt = q8 + (xcd < r8 ? 1 : 0);
    const int mine = slot < cnt ? (cnt - slot + per_xcd - 1) / per_xcd : 0;      // tiles of this workgroup
    if (mine == 0) return;
    const int NK = p.K / BK;
    const int per_z = p.mtiles * p.ntiles;
    if (p.stagger > 0) {
        const int ph = p.stagger_mode == 0 ? (slot >> 5) & 3 : p.stagger_mode == 1 ? slot & 3 : (slot >> 3) & 3;
        for (int i = 0; i < ph * p.stagger; ++i) __builtin_amdgcn_s_sleep(64);
    }

    auto decode = [&](int i, int& z, int& mt, int& nt) {                           // i-th tile of this workgroup
        const int L = x0 + slot + i * per_xcd;
        z = L / per_z;
        const int rem = L - z * per_z;
        const int per = p.group_m * p.ntiles;
        const int g = rem / per, rr = rem - g * per;
        const int gm = min(p.group_m, p.mtiles - g * p.group_m);                  // the last group may be short
        nt = rr / gm;
        mt = g * p.group_m + (rr - nt * gm);
    };

    // ---- issue side: the tile whose chunks are being LOADED (up to two chunks ahead of the one being multiplied)
    int it = 0;                          // tile index (of this workgroup) on the issue side
    int ik = 0;                          // its next chunk
    __amdgpu_buffer_rsrc_t rsA, rsB;
    int voa0, voa1, vob0, vob1;
    auto issue_setup = [&]() {
        int z, mt, nt;
        decode(it, z, mt, nt);
        const int rows_a = min(BM, p.N - mt * BM), rows_b = min(BN, p.Nb - nt * BN);
        rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (long)z * p.a_zs + (long)mt * BM * p.K), (short)0, rows_a * p.K * 4, 0x00020000);
        rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Bm + (long)z * p.b_zs + (long)nt * BN * p.K), (short)0, rows_b * p.K * 4, 0x00020000);
        // rows past the operand fall off the descriptor's extent and read as zero
        voa0 = r0 * (p.K * 4) + kq * 16;
        voa1 = (r0 + 64) * (p.K * 4) + kq * 16;
        vob0 = voa0;
        vob1 = voa1;
    };
    float4 ra0, ra1, rb0, rb1;
    auto issue = [&]() __attribute__((always_inline)) {
        const int so = ik * (BK * 4);
        v4i t;
        t = __builtin_amdgcn_raw_buffer_load_b128(rsA, voa0, so, 0); ra0 = *reinterpret_cast<float4*>(&t);
        t = __builtin_amdgcn_raw_buffer_load_b128(rsA, voa1, so, 0); ra1 = *reinterpret_cast<float4*>(&t);
        t = __builtin_amdgcn_raw_buffer_load_b128(rsB, vob0, so, 0); rb0 = *reinterpret_cast<float4*>(&t);
        t = __builtin_amdgcn_raw_buffer_load_b128(rsB, vob1, so, 0); rb1 = *reinterpret_cast<float4*>(&t);
    };
    auto advance = [&]() __attribute__((always_inline)) {    // next chunk to issue; past the last one: keep re-issuing it (never multiplied)
        if (ik + 1 < NK) { ++ik; return; }
        if (it + 1 < mine) { ++it; ik = 0; issue_setup(); }
    };
    auto commit = [&](float* st) __attribute__((always_inline)) {
        float* As = st;
        float* Bs = st + BM * LDK;
        *reinterpret_cast<float4*>(&As[r0 * LDK + kq * 4]) = ra0;
        *reinterpret_cast<float4*>(&As[(r0 + 64) * LDK + kq * 4]) = ra1;
        *reinterpret_cast<float4*>(&Bs[r0 * LDK + kq * 4]) = rb0;
        *reinterpret_cast<float4*>(&Bs[(r0 + 64) * LDK + kq * 4]) = rb1;
    };

    f32x16 acc[2][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    };
    zero_acc();

    // ---- epilogue of the tile on the compute side.  C/D layout of the 32x32 MFMA: column = lane & 31, row = (e & 3) + 8 (e >> 2)
    // + 4 (lane >> 5).  Stores only (no LDS: the next tile's first chunk already sits there), through descriptors based at the
    // tile's first row, so offsets stay small whatever the size of the volume.
    auto epilogue = [&](int ct) __attribute__((always_inline)) {
        int z, mt, nt;
        decode(ct, z, mt, nt);
        const int m0 = mt * BM, n0 = nt * BN;
        const int rows = min(BM, p.N - m0);
        const __amdgpu_buffer_rsrc_t rs_o =
            __builtin_amdgcn_make_buffer_rsrc((void*)(p.out + (long)z * p.o_zs + (long)m0 * p.Nb), (short)0, rows * p.Nb * 4, 0x00020000);
        __amdgpu_buffer_rsrc_t rs_p = rs_o;
        if constexpr (POOL)
            rs_p = __builtin_amdgcn_make_buffer_rsrc((void*)(p.pool + (long)z * p.pool_zs + (long)m0 * p.slice1), (short)0, rows * p.slice1 * 4, 0x00020000);
        const int mr0 = wm * 64 + 4 * (lane >> 5);                                 // first tile row of this lane
        // (the row step rides in the scalar offset, which the descriptor's range check does not see: the launcher only takes
        // N % 128 == 0, so every M tile is whole -- a masked variant of this loop pushed the kernel into scratch)
        {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int nbase = n0 + wn * 64 + j * 32;                          // wave-uniform
                const int n = nbase + (lane & 31);
                const int cmask = n < p.Nb ? 0 : kOOB;
                const int vo = ((mr0 * p.Nb + n) * 4) | cmask;
                int vp = 0;
                if constexpr (POOL) {
                    // the 32 columns of this sub-tile are ONE 4x8 block of the pixel's level-0 slice; its 2x2 averages are a 2x4 patch
                    // of level 1.  Partners of column c: c ^ 1 (next column) and c ^ 8 (next row): two DPP adds inside the 16-lane row
                    const int blk = nbase >> 5;
                    const int by = blk / p.wb0, bx = blk - by * p.wb0;
                    const int c = lane & 31;
                    const int y1 = (by << 1) + ((c >> 4) & 1), x1 = (bx << 2) + ((c >> 1) & 3);
                    const int idx1 = ((((y1 >> 2) * p.wb1) + (x1 >> 3)) << 5) + ((y1 & 3) << 3) + (x1 & 7);
                    vp = ((mr0 * p.slice1 + idx1) * 4) | (((c & 9) == 0 && n < p.Nb) ? 0 : kOOB);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int r = i * 32 + (e & 3) + 8 * (e >> 2);             // row step from mr0: compile-time
                        const float v = acc[i][j][e] * p.alpha;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs_o, vo, r * p.Nb * 4, 0);
                        if constexpr (POOL) {
                            const float s1 = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]: lane ^ 1
                            const float s2 = s1 + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s1), 0x128, 0xF, 0xF, false));  // row_ror:8: lane ^ 8
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(s2 * 0.25f), rs_p, vp, r * p.slice1 * 4, 0);
                        }
                    }
            }
        }
    };

    // ---- prologue: chunk 0 committed, chunk 1 in flight
    issue_setup();
    issue();
    advance();
    commit(smem);
    __syncthreads();
    issue();
    advance();
    // ---- one pipeline over every chunk of every tile of this workgroup, one barrier per chunk:
    //   MFMA block on stage[g & 1] -> commit chunk g + 1 to the other stage -> barrier -> issue chunk g + 2
    int g = 0;
    for (int ct = 0; ct < mine; ++ct) {
        for (int kt = 0; kt < NK; ++kt, ++g) {
            const float* As = smem + (g & 1) * STAGE;
            const float* Bs = As + BM * LDK;
#pragma unroll
            for (int ks = 0; ks < BK / 8; ++ks) {
                float4 fa[2], fb[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const float4*>(&As[(wm * 64 + i * 32 + frow) * LDK + ks * 8 + fk]);
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const float4*>(&Bs[(wn * 64 + j * 32 + frow) * LDK + ks * 8 + fk]);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
                    }
            }
            commit(smem + ((g + 1) & 1) * STAGE);      // chunk g + 1 has had this MFMA block to land
            __syncthreads();
            issue();                                    // chunk g + 2 (possibly of the next tile)
            advance();
            __builtin_amdgcn_sched_barrier(0);
        }
        epilogue(ct);
        zero_acc();
    }
}

int resident_wgs() {
    static int cached = 0;
    if (cached) return cached;
    int dev = 0, cus = 0, occ = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 0;
    // the smaller of the two instantiations' occupancies (they share their register budget: __launch_bounds__(256, 4))
    int o1 = 0, o2 = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o1, (const void*)vol_gemm_kernel<true>, 256, 0) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o2, (const void*)vol_gemm_kernel<false>, 256, 0) != hipSuccess) return 0;
    occ = std::min(o1, o2);
    if (occ <= 0) return 0;
    static const char* env = getenv("OFX_VOL_WGS_PER_CU");                  // probe: fewer resident workgroups per CU
    if (env && atoi(env) > 0) occ = std::min(occ, atoi(env));
    cached = (cus * occ) & ~7;
    return cached;
}

}  // namespace

// internal entry (ofx_internal.h).  Returns OFX_EINVAL for shapes it does not take; the caller then runs the generic GEMM.
int ofx_vol_gemm_launch(const float* A, long a_zs, const float* Bm, long b_zs, float* out, long o_zs, float* pool, long pool_zs, int N, int Nb,
                        int K, int nz, int wb0, int wb1, int slice1, float alpha, hipStream_t s) {
    OFX_REQUIRE(A && Bm && out && N > 0 && Nb > 0 && nz > 0, OFX_EINVAL);
    OFX_REQUIRE(N % BM == 0, OFX_EINVAL);                                         // whole M tiles only (see the epilogue); else the generic GEMM
    OFX_REQUIRE(K >= 2 * BK && K % BK == 0 && K <= 4096, OFX_EINVAL);
    OFX_REQUIRE(ofx_aligned16(A) && ofx_aligned16(Bm) && (a_zs % 4) == 0 && (b_zs % 4) == 0, OFX_EINVAL);
    OFX_REQUIRE((long)BM * Nb * 4 < (1L << 31) - 64, OFX_EINVAL);                 // per-tile store offsets are 32-bit
    if (pool) {
        OFX_REQUIRE(Nb % 128 == 0 && wb0 > 0 && wb1 > 0 && slice1 > 0 && (long)BM * slice1 * 4 < (1L << 31) - 64, OFX_EINVAL);
    }
    VolK k{};
    k.A = A; k.a_zs = a_zs; k.Bm = Bm; k.b_zs = b_zs; k.out = out; k.o_zs = o_zs; k.pool = pool; k.pool_zs = pool_zs;
    k.N = N; k.Nb = Nb; k.K = K; k.nz = nz;
    k.mtiles = (N + BM - 1) / BM; k.ntiles = (Nb + BN - 1) / BN;
    k.group_m = k.ntiles >= 8 ? 8 : 1;
    k.wb0 = wb0; k.wb1 = wb1; k.slice1 = slice1;
    k.alpha = alpha;
    const long total = (long)k.mtiles * k.ntiles * nz;
    OFX_REQUIRE(total < (1L << 30), OFX_EINVAL);
    k.total = (int)total;
    static const char* st_env = getenv("OFX_VOL_STAGGER");                  // probe: "<units of 64x64 cycles>[,mode]"
    if (st_env) { k.stagger = atoi(st_env); const char* c = strchr(st_env, ','); k.stagger_mode = c ? atoi(c + 1) : 0; }
    const int res = resident_wgs();
    OFX_REQUIRE(res >= 8, OFX_EINVAL);
    // fewer tiles than resident workgroups: one tile each (the grid stays a multiple of 8 so that every XCD gets its range)
    k.wgs = (int)std::min<long>(res, ((total + 7) / 8) * 8);
    OfxProfScope prof("igemm_corr_volume", s);
    prof.flops(2.0 * (double)N * Nb * K * nz);
    if (pool) hipLaunchKernelGGL(vol_gemm_kernel<true>, dim3((unsigned)k.wgs), dim3(256), 0, s, k);
    else hipLaunchKernelGGL(vol_gemm_kernel<false>, dim3((unsigned)k.wgs), dim3(256), 0, s, k);
    return ofx_launch_status();
}
