// Operators of the SD-inpaint hand-off beyond the Pillow steps: what `img2img_inpaint` runs
// between the composited frame and the first denoising step, and the attention primitive its UNet calls.
//
//   ofx_groupnorm        ldm/modules/diffusionmodules/model.py:40-41 (`Normalize` = GroupNorm(32, C, eps 1e-6, affine))
//                        followed, when asked, by `nonlinearity` = x * sigmoid(x) (:35-37) -- the pair that precedes every
//                        convolution of the VAE encoder (:129-141, :540-541)
//   ofx_softmax_rows     the row softmax of AttnBlock (:191-193) and of `memory_efficient_attention`
//   ofx_attention_f32    softmax(q k^T * scale + bias) v for [BH, N, D] tensors: the semantics of
//                        xformers.ops.memory_efficient_attention(q, k, v, attn_bias) at ldm/modules/attention.py:314,426
//                        and of AttnBlock.forward (model.py:179-203).  Both GEMMs run on the fp32 matrix cores through
//                        the batched mode of the implicit-GEMM kernel (conv.hip); the score matrix lives in a caller
//                        workspace (1 GiB for the VAE's 16384-token mid block at 1024x1024: 288 GB of HBM make the
//                        unfused form the simple and exact one).  The UNet's head sizes are routed to the fused kernel of
//                        attn_flash.hip instead.
#include "ofx_internal.h"

#include <algorithm>
#include <cmath>

namespace {

// slices of an image summed by separate workgroups: few images -> many slices, so that the statistics pass fills the chip
static inline int gn_slices(int B) { return B >= 4 ? 64 : 256; }

// per (image, slice) per-channel sums in f64 (same scheme as the instance-norm statistics of net_misc.hip)
__global__ __launch_bounds__(256) void gn_partial_kernel(const float* __restrict__ x, double* __restrict__ part, long HW, int C, int kSlices) {
    const int cg = C / 4;
    const int rows = 256 / cg > 0 ? 256 / cg : 1;
    const int b = blockIdx.y, sl = blockIdx.x;
    const long per = (HW + kSlices - 1) / kSlices;
    const long beg = sl * per, end = beg + per < HW ? beg + per : HW;
    __shared__ double red[256 * 8];
    // channel groups beyond 256 threads (C > 1024) are walked in passes
    for (int c0 = 0; c0 < cg; c0 += 256) {
        const int ncg = min(256, cg - c0);
        const int rws = 256 / ncg;
        const int tc = threadIdx.x % ncg, tr = threadIdx.x / ncg;
        double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
        if (tr < rws) {
            const float* base = x + ((long)b * HW) * C + (c0 + tc) * 4;
            for (long i = beg + tr; i < end; i += rws) {
                const float4 v = *reinterpret_cast<const float4*>(base + i * C);
                s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
                q[0] += (double)v.x * v.x; q[1] += (double)v.y * v.y; q[2] += (double)v.z * v.z; q[3] += (double)v.w * v.w;
            }
        }
        for (int k = 0; k < 4; ++k) {
            red[threadIdx.x * 8 + k] = s[k];
            red[threadIdx.x * 8 + 4 + k] = q[k];
        }
        __syncthreads();
        if ((int)threadIdx.x < ncg) {
            double ts[4] = {0, 0, 0, 0}, tq[4] = {0, 0, 0, 0};
            for (int r = 0; r < rws; ++r)
                for (int k = 0; k < 4; ++k) {
                    ts[k] += red[(r * ncg + threadIdx.x) * 8 + k];
                    tq[k] += red[(r * ncg + threadIdx.x) * 8 + 4 + k];
                }
            double* o = part + (((long)b * kSlices + sl) * C + (c0 + threadIdx.x) * 4) * 2;
            for (int k = 0; k < 4; ++k) {
                o[k * 2] = ts[k];
                o[k * 2 + 1] = tq[k];
            }
        }
        __syncthreads();
    }
    (void)rows;
}

// one workgroup per image: group statistics -> per-channel scale / shift (y = x * scale + shift).  The 256 threads are
// dealt to the groups (256 / groups threads each, every one summing its share of the slices in a fixed order, then one
// thread per group adding the shares in index order: deterministic).
__global__ __launch_bounds__(256) void gn_finalize_kernel(const double* __restrict__ part, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ scale,
                                                          float* __restrict__ shift, long HW, int C, int groups, float eps, int kSlices) {
    __shared__ double rs[256], rq[256];
    __shared__ double gmu[256], grs[256];
    const int b = blockIdx.x;
    const int cpg = C / groups;
    const int tpg = groups >= 256 ? 1 : 256 / groups;          // threads per group
    for (int g0 = 0; g0 < groups; g0 += 256 / tpg) {
        const int gl = threadIdx.x / tpg, sub = threadIdx.x - gl * tpg;
        const int g = g0 + gl;
        double s = 0, q = 0;
        if (g < groups)
            for (int sl = sub; sl < kSlices; sl += tpg) {
                const double* o = part + (((long)b * kSlices + sl) * C + g * cpg) * 2;
                for (int c = 0; c < cpg; ++c) {
                    s += o[2 * c];
                    q += o[2 * c + 1];
                }
            }
        rs[threadIdx.x] = s;
        rq[threadIdx.x] = q;
        __syncthreads();
        if (sub == 0 && g < groups) {
            s = 0; q = 0;
            for (int k = 0; k < tpg; ++k) {
                s += rs[threadIdx.x + k];
                q += rq[threadIdx.x + k];
            }
            const double n = (double)HW * cpg;
            const double mu = s / n;
            double var = q / n - mu * mu;
            if (var < 0) var = 0;
            gmu[gl] = mu;
            grs[gl] = 1.0 / sqrt(var + (double)eps);
        }
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += 256) {
            const int g2 = c / cpg - g0;
            if (g2 >= 0 && g2 < 256 / tpg) {
                const double r = grs[g2] * (double)(gamma ? gamma[c] : 1.f);
                scale[(long)b * C + c] = (float)r;
                shift[(long)b * C + c] = (float)((double)(beta ? beta[c] : 0.f) - gmu[g2] * r);
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, float* __restrict__ out, long HW, int C,
                                                       long total4, int silu) {
    const int cg = C / 4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cg) * 4;
        const long b = (i / cg) / HW;
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        const float4 sc = *reinterpret_cast<const float4*>(scale + b * C + c);
        const float4 sh = *reinterpret_cast<const float4*>(shift + b * C + c);
        float4 y = make_float4(fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w));
        if (silu) {
            y.x = y.x / (1.0f + expf(-y.x));
            y.y = y.y / (1.0f + expf(-y.y));
            y.z = y.z / (1.0f + expf(-y.z));
            y.w = y.w / (1.0f + expf(-y.w));
        }
        reinterpret_cast<float4*>(out)[i] = y;
    }
}

// in-place row softmax of x[r][0..n) * scale (+ bias[r % bias_rows][0..n)); columns n..ld-1 are zeroed (they are
// the K padding of the following GEMM).  One workgroup per row, three passes over the row (max, sum, normalise).
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ x, long ld, int n, float scale, const float* __restrict__ bias,
                                                           long ldb, long bias_rows) {
    __shared__ float red[256];
    float* row = x + (long)blockIdx.x * ld;
    const float* brow = bias ? bias + ((long)blockIdx.x % bias_rows) * ldb : nullptr;
    float mx = -3.402823466e38f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float v = row[i] * scale + (brow ? brow[i] : 0.f);
        row[i] = v;
        mx = fmaxf(mx, v);
    }
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    mx = red[0];
    __syncthreads();
    float sum = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float e = expf(row[i] - mx);
        row[i] = e;
        sum += e;
    }
    red[threadIdx.x] = sum;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    const float inv = 1.0f / red[0];
    for (int i = threadIdx.x; i < (int)ld; i += 256) row[i] = i < n ? row[i] * inv : 0.f;
}

// dst[z][r][0..ldd) = src[z][r][0..C) zero-padded (rows of a GEMM's B operand, K padded to the kernel's chunk)
__global__ __launch_bounds__(256) void pad_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, long rows, int C, int ldd) {
    const long total = rows * ldd;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / ldd;
        const int c = (int)(i - r * ldd);
        dst[i] = c < C ? src[r * C + c] : 0.f;
    }
}

// dst[z][c][0..ldd) = src[z][0..R)[c] zero-padded: V [R][C] -> V^T [C][ldd] through a 32x32 LDS tile
__global__ __launch_bounds__(256) void transpose_pad_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int C, int ldd) {
    __shared__ float tile[32][33];
    const long z = blockIdx.z;
    const float* s = src + z * (long)R * C;
    float* d = dst + z * (long)C * ldd;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8) {
        const int r = r0 + k, c = c0 + tx;
        tile[k][tx] = (r < R && c < C) ? s[(long)r * C + c] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, r = r0 + tx;
        if (c < C && r < ldd) d[(long)c * ldd + r] = tile[tx][k];
    }
}

inline long round_up(long v, long m) { return (v + m - 1) / m * m; }

}  // namespace

extern "C" {

size_t ofx_groupnorm_scratch_bytes(int B, int C) { return (size_t)B * gn_slices(B) * C * 2 * sizeof(double) + (size_t)B * C * 2 * sizeof(float); }

int ofx_groupnorm(const float* x, const float* gamma, const float* beta, float* out, void* scratch, size_t scratch_bytes, int B,
                  long HW, int C, int groups, float eps, int silu, void* stream) {
    OFX_REQUIRE(x && out && scratch && B > 0 && B <= 65535 && HW > 0 && C > 0 && groups > 0 && C % groups == 0, OFX_EINVAL);
    OFX_REQUIRE(C % 4 == 0 && ofx_aligned16(x) && ofx_aligned16(out) && ofx_aligned16(scratch), OFX_EALIGN);
    OFX_REQUIRE(scratch_bytes >= ofx_groupnorm_scratch_bytes(B, C), OFX_ENOMEM);
    hipStream_t s = (hipStream_t)stream;
    const int kSlices = gn_slices(B);
    double* part = reinterpret_cast<double*>(scratch);
    float* scale = reinterpret_cast<float*>(part + (size_t)B * kSlices * C * 2);
    float* shift = scale + (size_t)B * C;
    {
        OfxProfScope prof("groupnorm_stats", s);
        hipLaunchKernelGGL(gn_partial_kernel, dim3(kSlices, B), dim3(256), 0, s, x, part, HW, C, kSlices);
        hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(256), 0, s, part, gamma, beta, scale, shift, HW, C, groups, eps, kSlices);
    }
    int st = ofx_launch_status();
    if (st) return st;
    const long total4 = (long)B * HW * (C / 4);
    OfxProfScope prof("groupnorm_apply", s);
    hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)std::min<long>((total4 + 255) / 256, 65536)), dim3(256), 0, s, x, scale, shift, out, HW,
                       C, total4, silu);
    return ofx_launch_status();
}

int ofx_softmax_rows(float* x, long rows, long ld, int n, float scale, const float* bias, long ld_bias, long bias_rows, void* stream) {
    OFX_REQUIRE(x && rows > 0 && n > 0 && ld >= n && rows < (1L << 31), OFX_EINVAL);
    if (bias) OFX_REQUIRE(ld_bias >= n && bias_rows > 0, OFX_EINVAL);
    hipStream_t s = (hipStream_t)stream;
    OfxProfScope prof("softmax_rows", s);
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, s, x, ld, n, scale, bias, ld_bias, bias ? bias_rows : 1);
    return ofx_launch_status();
}

// floats of workspace for one call: padded K rows + scores + V^T for `bh` batch-heads at a time
size_t ofx_attention_workspace_bytes(int BH, int Nq, int Nk, int D) {
    if (BH <= 0 || Nq <= 0 || Nk <= 0 || D <= 0) return 0;
    if (ofx_attention_flash_ok(D)) return 0;            // fused kernel: the scores never leave the CU
    const long kp = round_up(D, 32), lds = round_up(Nk, 4), vp = round_up(Nk, 32);
    return (size_t)BH * ((size_t)Nk * kp + (size_t)Nq * lds + (size_t)D * vp) * sizeof(float) + 1024;
}

int ofx_attention_f32(const float* q, const float* k, const float* v, const float* bias, long bias_bstride, float* out, int BH, int Nq, int Nk,
                      int D, float scale, void* workspace, size_t workspace_bytes, void* stream) {
    OFX_REQUIRE(q && k && v && out && BH > 0 && Nq > 0 && Nk > 0 && D > 0, OFX_EINVAL);
    OFX_REQUIRE(D % 4 == 0 && ofx_aligned16(q) && ofx_aligned16(k) && ofx_aligned16(v) && ofx_aligned16(out), OFX_EALIGN);
    if (ofx_attention_flash_ok(D))
        return ofx_attention_flash_launch(q, k, v, bias, bias_bstride, out, BH, Nq, Nk, D, scale, (hipStream_t)stream);
    OFX_REQUIRE(workspace && ofx_aligned16(workspace), OFX_EALIGN);
    OFX_REQUIRE(workspace_bytes >= ofx_attention_workspace_bytes(BH, Nq, Nk, D), OFX_ENOMEM);
    hipStream_t s = (hipStream_t)stream;
    const long kp = round_up(D, 32), lds = round_up(Nk, 4), vp = round_up(Nk, 32);
    float* kpad = reinterpret_cast<float*>(workspace);
    float* sc = kpad + (size_t)BH * Nk * kp;
    float* vt = sc + (size_t)BH * Nq * lds;
    {
        OfxProfScope prof("attn_pack", s);
        const long tot = (long)BH * Nk * kp;
        hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)std::min<long>((tot + 255) / 256, 65536)), dim3(256), 0, s, k, kpad, (long)BH * Nk, D, (int)kp);
        hipLaunchKernelGGL(transpose_pad_kernel, dim3(ofx_cdiv(D, 32), ofx_cdiv(vp, 32), BH), dim3(256), 0, s, v, vt, Nk, D, (int)vp);
    }
    int st = ofx_launch_status();
    if (st) return st;
    // scores[z] = q[z] (Nq x D) * kpad[z]^T (D x Nk); the batched mode needs nz > 1, a single batch-head is a plain GEMM
    ofx_conv_desc d{};
    d.in0 = q; d.ld0 = D; d.c0 = D;
    d.w = kpad;
    d.out = sc; d.ldo = (int)lds;
    d.nz = BH; d.a_zs = (long)Nq * D; d.w_zs = (long)Nk * kp; d.o_zs = (long)Nq * lds;
    d.B = 1; d.Hin = 1; d.Win = Nq; d.Hout = 1; d.Wout = Nq; d.Cout = Nk;
    d.KH = 1; d.KW = 1; d.stride = 1;
    d.act = OFX_ACT_NONE; d.epi = OFX_EPI_PLAIN;
    st = ofx_conv2d(&d, stream);
    if (st) return st;
    st = ofx_softmax_rows(sc, (long)BH * Nq, lds, Nk, scale, bias, bias ? Nk : 0, bias ? (bias_bstride ? (long)BH * Nq : Nq) : 1, stream);
    if (st) return st;
    // out[z] = P[z] (Nq x Nk) * vt[z]^T (Nk x D)
    ofx_conv_desc e{};
    e.in0 = sc; e.ld0 = (int)lds; e.c0 = (int)lds;
    e.w = vt;
    e.out = out; e.ldo = D;
    e.nz = BH; e.a_zs = (long)Nq * lds; e.w_zs = (long)D * vp; e.o_zs = (long)Nq * D;
    e.B = 1; e.Hin = 1; e.Win = Nq; e.Hout = 1; e.Wout = Nq; e.Cout = D;
    e.KH = 1; e.KW = 1; e.stride = 1;
    e.act = OFX_ACT_NONE; e.epi = OFX_EPI_PLAIN;
    return ofx_conv2d(&e, stream);
}

}  // extern "C"
