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
// Arithmetic is exact fp32: v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate, bit-identical to an
// fmaf chain, 157 TFLOP/s peak = the roofline of this kernel).  The reference runs fp32
// (mixed_precision=False, ofgen_keyframe_inpaint.py:57) and the parity bar is EPE <= 1e-3 px after
// 20 recurrent iterations, so no reduced-precision operand format is used.
//
// Tiling (64-wide wavefronts): 256 threads = 4 waves per workgroup, a BMxBN output tile, BK = 32.
// Both operands are staged k-contiguous in LDS with a row stride of 36 floats, which makes the
// per-lane ds_read_b128 fragment reads conflict-free; each b128 read feeds four MFMAs (lane half h
// supplies k = 8*ks + 4*h + s for s = 0..3 -- the k order inside a chunk is permuted identically for
// A and B, which leaves the dot product unchanged).  The next k-chunk's global loads are issued
// before the MFMA block so HBM/L2 latency hides under ~1-4k cycles of matrix work, and the linear
// block id is remapped so that the N-tiles that share an A tile run on the same XCD (private L2).
//
// Epilogues fuse bias / folded BatchNorm, activation, residual add, the GRU gate algebra
// (z, r*h, h = (1-z)h + z*q) and the flow/coords update, so none of those run as separate passes.
#include "ofx_internal.h"

namespace {

struct ConvK {
    const float* in0;
    const float* in1;
    const float* w;
    const float* scale;
    const float* shift;
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
    int ld0, c0, ld1, c1, cin, ldo, ldres, ldh;
    int Hin, Win, Hout, Wout, Cout, KW, stride, padH, padW;
    int K, Kpad, M, act;
    int mtiles, ntiles;
    float alpha;
};

constexpr int kBK = 32;
constexpr int kLDK = 36;   // LDS row stride in floats: 144 B keeps b128 fragment reads conflict-free

template <int ACT>
__device__ __forceinline__ float apply_act(float v) {
    if (ACT == OFX_ACT_RELU) return fmaxf(v, 0.0f);
    if (ACT == OFX_ACT_SIGMOID) return ofx_sigmoid(v);
    if (ACT == OFX_ACT_TANH) return tanhf(v);
    return v;
}

__device__ __forceinline__ float apply_act_rt(float v, int act) {
    switch (act) {
        case OFX_ACT_RELU: return fmaxf(v, 0.0f);
        case OFX_ACT_SIGMOID: return ofx_sigmoid(v);
        case OFX_ACT_TANH: return tanhf(v);
        default: return v;
    }
}

template <int BM, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(256) void igemm_kernel(const ConvK p) {
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    constexpr int A_PER = BM / 32, B_PER = BN / 32;
    static_assert((BM / WM) * (BN / WN) == 4, "4 waves per workgroup");
    __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * kLDK];
    float* As = smem;
    float* Bs = smem + BM * kLDK;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N;
    const int wn = wave % WAVES_N;

    // XCD-aware bijective remap: hardware places block b on XCD b%8; give each XCD a contiguous
    // range of logical tiles so N-tiles sharing an A tile (and neighbouring M-tiles sharing halo
    // rows) hit the same private L2.
    const int nblk = p.mtiles * p.ntiles;
    const int bid = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    const int nt = L % p.ntiles;
    const int mt = L / p.ntiles;
    const int m0 = mt * BM;
    const int n0 = nt * BN;
    const int z = blockIdx.y;

    const float* __restrict__ in0 = p.in0 + (long)z * p.a_zs;
    const float* __restrict__ in1 = p.in1;
    const float* __restrict__ wgt = p.w + (long)z * p.w_zs;

    // ---- per-thread gather coordinates for the A (im2col) tile
    const int kq = tid & 7;    // float4 slot inside the 32-wide k chunk
    const int r0 = tid >> 3;   // 0..31: row inside each 32-row group
    const int HWo = p.Hout * p.Wout;
    int a_pix[A_PER], a_iy0[A_PER], a_ix0[A_PER], a_b[A_PER];
    bool a_ok[A_PER];
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
        int m = m0 + r0 + 32 * i;
        bool ok = m < p.M;
        int mm = ok ? m : 0;
        int b = mm / HWo;
        int rem = mm - b * HWo;
        int oy = rem / p.Wout;
        int ox = rem - oy * p.Wout;
        a_ok[i] = ok;
        a_b[i] = b;
        a_pix[i] = b * p.Hin * p.Win;
        a_iy0[i] = oy * p.stride - p.padH;
        a_ix0[i] = ox * p.stride - p.padW;
    }

    float4 ra[A_PER], rb[B_PER];
    auto load_tiles = [&](int k0) {
        const int k = k0 + kq * 4;
        const bool kok = k < p.K;
        const int tap = k / p.cin;
        const int c = k - tap * p.cin;
        const int ky = tap / p.KW;
        const int kx = tap - ky * p.KW;
        const bool seg0 = c < p.c0;
        const float* base = seg0 ? in0 : in1;
        const int ld = seg0 ? p.ld0 : p.ld1;
        const int cc = seg0 ? c : c - p.c0;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int iy = a_iy0[i] + ky;
            const int ix = a_ix0[i] + kx;
            const bool ok = a_ok[i] && kok && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) {
                v = *reinterpret_cast<const float4*>(base + (long)(a_pix[i] + iy * p.Win + ix) * ld + cc);
                if (p.nmean != nullptr && seg0) {
                    const float4 mu = *reinterpret_cast<const float4*>(p.nmean + (long)a_b[i] * p.c0 + cc);
                    const float4 rs = *reinterpret_cast<const float4*>(p.nrstd + (long)a_b[i] * p.c0 + cc);
                    v.x = fmaxf((v.x - mu.x) * rs.x, 0.f);
                    v.y = fmaxf((v.y - mu.y) * rs.y, 0.f);
                    v.z = fmaxf((v.z - mu.z) * rs.z, 0.f);
                    v.w = fmaxf((v.w - mu.w) * rs.w, 0.f);
                }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int n = n0 + r0 + 32 * i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n < p.Cout) v = *reinterpret_cast<const float4*>(wgt + (long)n * p.Kpad + k0 + kq * 4);
            rb[i] = v;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = p.Kpad / kBK;
    load_tiles(0);
    const int frag_row = lane & 31;
    const int frag_k = (lane >> 5) * 4;
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i)
            *reinterpret_cast<float4*>(&As[(r0 + 32 * i) * kLDK + kq * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < B_PER; ++i)
            *reinterpret_cast<float4*>(&Bs[(r0 + 32 * i) * kLDK + kq * 4]) = rb[i];
        __syncthreads();
        if (kt + 1 < nk) load_tiles((kt + 1) * kBK);   // in flight during the MFMA block
#pragma unroll
        for (int ks = 0; ks < kBK / 8; ++ks) {
            float4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[i] = *reinterpret_cast<const float4*>(&As[(wm * WM + i * 32 + frag_row) * kLDK + ks * 8 + frag_k]);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                fb[j] = *reinterpret_cast<const float4*>(&Bs[(wn * WN + j * 32 + frag_row) * kLDK + ks * 8 + frag_k]);
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
        __syncthreads();
    }

    // ---- epilogue.  C/D layout of the 32x32 MFMA: col (n) = lane&31, row (m) = (e&3) + 8*(e>>2) + 4*(lane>>5)
    float* __restrict__ out = p.out ? p.out + (long)z * p.o_zs : nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WN + j * 32 + (lane & 31);
        if (n >= p.Cout) continue;
        const float sc = p.scale ? p.scale[n] * p.alpha : p.alpha;
        const float sh = p.shift ? p.shift[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (m >= p.M) continue;
                float v = acc[i][j][e] * sc + sh;
                if (EPI == OFX_EPI_PLAIN) {
                    v = apply_act_rt(v, p.act);
                    if (p.res) v = fmaxf(v + p.res[(long)m * p.ldres + n], 0.f);
                    out[(long)m * p.ldo + n] = v;
                } else if (EPI == OFX_EPI_GRU_ZR) {
                    const int hd = p.Cout >> 1;
                    v = ofx_sigmoid(v);
                    if (n < hd) {
                        p.aux_z[(long)m * hd + n] = v;
                    } else {
                        const int c = n - hd;
                        p.aux_rh[(long)m * hd + c] = v * p.aux_h[(long)m * p.ldh + c];
                    }
                } else if (EPI == OFX_EPI_GRU_Q) {
                    const float qv = tanhf(v);
                    const float zz = p.aux_z[(long)m * p.Cout + n];
                    const long hi = (long)m * p.ldh + n;
                    const float h = p.aux_h[hi];
                    p.aux_h[hi] = (1.0f - zz) * h + zz * qv;
                } else if (EPI == OFX_EPI_FLOW) {
                    const int rem = m % HWo;
                    const int oy = rem / p.Wout;
                    const int ox = rem - oy * p.Wout;
                    const float c1 = p.aux_coords[(long)m * 2 + n] + v;
                    p.aux_coords[(long)m * 2 + n] = c1;
                    const float fl = c1 - (float)(n == 0 ? ox : oy);
                    p.aux_h[(long)m * p.ldh + n] = fl;
                    p.aux_flow4[(long)m * 4 + n] = fl;
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN>
int launch_tile(const ConvK& k, int epi, int nz, hipStream_t s) {
    dim3 grid((unsigned)(k.mtiles * k.ntiles), (unsigned)nz, 1);
    dim3 block(256, 1, 1);
    switch (epi) {
        case OFX_EPI_PLAIN: hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, OFX_EPI_PLAIN>), grid, block, 0, s, k); break;
        case OFX_EPI_GRU_ZR: hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, OFX_EPI_GRU_ZR>), grid, block, 0, s, k); break;
        case OFX_EPI_GRU_Q: hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, OFX_EPI_GRU_Q>), grid, block, 0, s, k); break;
        case OFX_EPI_FLOW: hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, OFX_EPI_FLOW>), grid, block, 0, s, k); break;
        default: return OFX_EINVAL;
    }
    return ofx_launch_status();
}

}  // namespace

extern "C" int ofx_conv2d(const ofx_conv_desc* d, void* stream) { return ofx_conv2d_alpha(d, 1.0f, stream); }

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
    if (d->nmean) OFX_REQUIRE(d->nrstd && ofx_aligned16(d->nmean) && ofx_aligned16(d->nrstd), OFX_EALIGN);
    const int nz = d->nz > 1 ? d->nz : 1;

    ConvK k;
    k.in0 = d->in0; k.in1 = d->in1; k.w = d->w; k.scale = d->scale; k.shift = d->shift;
    k.out = d->out; k.res = d->res; k.nmean = d->nmean; k.nrstd = d->nrstd;
    k.aux_z = d->aux_z; k.aux_rh = d->aux_rh; k.aux_h = d->aux_h;
    k.aux_coords = d->aux_coords; k.aux_flow4 = d->aux_flow4;
    k.a_zs = nz > 1 ? d->a_zs : 0; k.w_zs = nz > 1 ? d->w_zs : 0; k.o_zs = nz > 1 ? d->o_zs : 0;
    k.ld0 = d->ld0; k.c0 = d->c0; k.ld1 = d->ld1; k.c1 = d->c1; k.cin = d->c0 + d->c1;
    k.ldo = d->ldo; k.ldres = d->ldres; k.ldh = d->ldh;
    k.Hin = d->Hin; k.Win = d->Win; k.Hout = d->Hout; k.Wout = d->Wout; k.Cout = d->Cout;
    k.KW = d->KW; k.stride = d->stride; k.padH = d->padH; k.padW = d->padW;
    k.K = d->KH * d->KW * k.cin;
    k.Kpad = ((k.K + kBK - 1) / kBK) * kBK;
    const long M = (long)d->B * d->Hout * d->Wout;
    OFX_REQUIRE(M < (1L << 31) && (long)d->B * d->Hin * d->Win < (1L << 31), OFX_EINVAL);
    k.M = (int)M;
    k.act = d->act;
    k.alpha = alpha;

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
        bm = d->tile / 1000;
        bn = d->tile % 1000;
    } else {
        auto waste = [&](int t) { return (double)(((d->Cout + t - 1) / t) * t) / d->Cout; };
        if (d->Cout <= 32) bn = 32;
        else if (waste(128) <= 1.13) bn = 128;
        else if (waste(64) <= 1.13) bn = 64;
        else if (waste(32) < waste(64) - 0.1) bn = 32;
        else bn = 64;
        bm = 128;
        const long blocks128 = ((M + 127) / 128) * ((d->Cout + bn - 1) / bn) * nz;
        if (bn >= 64 && blocks128 < 1024) bm = 64;   // under ~4 waves of 256 CUs: smaller tiles fill the chip
        if (bm == 64 && bn == 128) bn = 64;
    }
    k.mtiles = (int)((M + bm - 1) / bm);
    k.ntiles = (d->Cout + bn - 1) / bn;
    hipStream_t s = (hipStream_t)stream;
    const char* pname = nz > 1 ? "igemm_corr_volume"
                       : d->epi == OFX_EPI_GRU_ZR ? "igemm_conv_gru_zr"
                       : d->epi == OFX_EPI_GRU_Q  ? "igemm_conv_gru_q"
                       : d->epi == OFX_EPI_FLOW   ? "igemm_conv_flow"
                                                  : "igemm_conv";
    OfxProfScope prof(pname, s);
    if (bm == 128 && bn == 128) return launch_tile<128, 128, 64, 64>(k, d->epi, nz, s);
    if (bm == 128 && bn == 64) return launch_tile<128, 64, 64, 32>(k, d->epi, nz, s);
    if (bm == 128 && bn == 32) return launch_tile<128, 32, 32, 32>(k, d->epi, nz, s);
    if (bm == 64 && bn == 64) return launch_tile<64, 64, 32, 32>(k, d->epi, nz, s);
    return OFX_EINVAL;
}

extern "C" long ofx_pack_conv_weight(const float* w, int Cout, int Cin, int KH, int KW, int cin_pad, float* out) {
    if (Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0 || cin_pad < Cin || cin_pad % 4) return OFX_EINVAL;
    const long K = (long)KH * KW * cin_pad;
    const long Kpad = ((K + kBK - 1) / kBK) * kBK;
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
