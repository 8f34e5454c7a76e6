// Streaming kernels around the convolutions of the flow network: image pre-processing, instance
// normalisation (statistics + apply), and the convex 8x flow upsample.  All are HBM-bound, read and
// write every byte once, and use 16-byte per-lane accesses on NHWC rows.
//
//   preprocess      RAFT/core/raft.py:89-90   (2*(x/255)-1), plus BGR->RGB of RAFT_2.calc
//   instance norm   RAFT/core/extractor.py:27-31,123-124 (nn.InstanceNorm2d: no affine, eps=1e-5, biased var)
//   residual merge  RAFT/core/extractor.py:46-56
//   convex upsample RAFT/core/raft.py:72-83
#include "ofx_internal.h"
#include "upsample_inl.h"

namespace {

__global__ __launch_bounds__(256) void preprocess_kernel(const uint8_t* __restrict__ img, float* __restrict__ out,
                                                         long npix, int bgr) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
        const uint8_t* p = img + i * 3;
        const float c0 = (float)p[bgr ? 2 : 0], c1 = (float)p[1], c2 = (float)p[bgr ? 0 : 2];
        float4 o;
        o.x = __fsub_rn(__fmul_rn(2.0f, __fdiv_rn(c0, 255.0f)), 1.0f);
        o.y = __fsub_rn(__fmul_rn(2.0f, __fdiv_rn(c1, 255.0f)), 1.0f);
        o.z = __fsub_rn(__fmul_rn(2.0f, __fdiv_rn(c2, 255.0f)), 1.0f);
        o.w = 0.f;
        reinterpret_cast<float4*>(out)[i] = o;
    }
}

// ---- instance-norm statistics: per (b, slice) partial sums in f64, then a finalize pass
// Slices per image: 64 fill the machine for a full encoder chunk; a handful of images (the single-pair
// configuration) would leave most CUs idle in this reduction, so small batches are cut 4x finer.
static inline int inorm_slices(int B) { return B >= 8 ? 64 : 256; }

__global__ __launch_bounds__(256) void inorm_partial_kernel(const float* __restrict__ x, int ld, double* __restrict__ part,
                                                            long HW, int C, int kSlices) {
    // grid: (kSlices, B).  Threads: (C/4) channel groups x rows.
    const int cg = C / 4;
    const int rows = 256 / cg;
    const int tc = threadIdx.x % cg, tr = threadIdx.x / cg;
    const int b = blockIdx.y, sl = blockIdx.x;
    const long per = (HW + kSlices - 1) / kSlices;
    const long beg = sl * per, end = beg + per < HW ? beg + per : HW;
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    if (tr < rows) {
        const float* base = x + ((long)b * HW) * ld + tc * 4;
        for (long i = beg + tr; i < end; i += rows) {
            const float4 v = *reinterpret_cast<const float4*>(base + i * ld);
            s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
            q[0] += (double)v.x * v.x; q[1] += (double)v.y * v.y; q[2] += (double)v.z * v.z; q[3] += (double)v.w * v.w;
        }
    }
    __shared__ double red[256 * 8];
    for (int k = 0; k < 4; ++k) {
        red[threadIdx.x * 8 + k] = s[k];
        red[threadIdx.x * 8 + 4 + k] = q[k];
    }
    __syncthreads();
    if ((int)threadIdx.x < cg) {
        double ts[4] = {0, 0, 0, 0}, tq[4] = {0, 0, 0, 0};
        for (int r = 0; r < rows; ++r)
            for (int k = 0; k < 4; ++k) {
                ts[k] += red[(r * cg + threadIdx.x) * 8 + k];
                tq[k] += red[(r * cg + threadIdx.x) * 8 + 4 + k];
            }
        double* o = part + (((long)b * kSlices + sl) * C + threadIdx.x * 4) * 2;
        for (int k = 0; k < 4; ++k) {
            o[k * 2] = ts[k];
            o[k * 2 + 1] = tq[k];
        }
    }
}

// one workgroup per (image, 16 channels): thread t sums the slices t / 16, t / 16 + 16, ... of channel t % 16, then the
// 16 partial sums of a channel are added through LDS in a fixed order (deterministic)
// Optional affine (gamma, beta: the context encoder's BatchNorm evaluated with per-image statistics, see raft_engine.cpp): folded into
// the pair the consumers already apply as (x - mean) * rstd:   (x - mu) * rs * g + b  =  (x - (mu - b / (rs g))) * (rs g).
__device__ __forceinline__ void inorm_store(float* mean, float* rstd, int i, int c, double mu, double var, float eps,
                                            const float* gamma, const float* beta) {
    if (var < 0) var = 0;
    double rs = 1.0 / sqrt(var + (double)eps);
    if (gamma) {
        double a = rs * (double)gamma[c];
        if (fabs(a) < 1e-30) a = 1e-30;       // gamma == 0: the output is beta whatever x is
        mu -= (double)beta[c] / a;
        rs = a;
    }
    mean[i] = (float)mu;
    rstd[i] = (float)rs;
}

__global__ __launch_bounds__(256) void inorm_finalize_kernel(const double* __restrict__ part, float* __restrict__ mean,
                                                             float* __restrict__ rstd, long HW, int C, float eps, int kSlices,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta) {
    __shared__ double red[256 * 2];
    const int b = blockIdx.y;
    const int c = blockIdx.x * 16 + (threadIdx.x & 15), gsl = threadIdx.x >> 4;
    double s = 0, q = 0;
    if (c < C)
        for (int sl = gsl; sl < kSlices; sl += 16) {
            const double* o = part + (((long)b * kSlices + sl) * C + c) * 2;
            s += o[0];
            q += o[1];
        }
    red[threadIdx.x * 2] = s;
    red[threadIdx.x * 2 + 1] = q;
    __syncthreads();
    if (threadIdx.x >= 16 || c >= C) return;
    s = 0; q = 0;
    for (int g2 = 0; g2 < 16; ++g2) {
        s += red[(g2 * 16 + threadIdx.x) * 2];
        q += red[(g2 * 16 + threadIdx.x) * 2 + 1];
    }
    const double mu = s / (double)HW;
    inorm_store(mean, rstd, b * C + c, c, mu, q / (double)HW - mu * mu, eps, gamma, beta);
}

// The same finalisation for statistics that come out of the convolution epilogue (conv.hip, ofx_conv2d_stats): `part` holds, per
// image, `rows` rows of [C][2] floats (sum, sum of squares over the 32 or 64 tile rows of one wave).  One workgroup per (image, 4
// channels): thread t adds the rows t / 4, t / 4 + 64, ... of channel t % 4 in f64 (a single pair has thousands of rows and
// only a few images: the work has to spread over rows), then the 64 partial sums of a channel are added through LDS in a fixed
// order.
__global__ __launch_bounds__(256) void inorm_finalize_part_kernel(const float* __restrict__ part, float* __restrict__ mean,
                                                                  float* __restrict__ rstd, long HW, int C, float eps, int rows,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta) {
    __shared__ double red[256 * 2];
    const int b = blockIdx.y;
    const int c = blockIdx.x * 4 + (threadIdx.x & 3), g = threadIdx.x >> 2;
    double s = 0, q = 0;
    if (c < C) {
        // eight rows' loads in flight per trip (one at a time this loop was a chain of ~50 dependent round trips: 27 us per call on a
        // single pair, 30 calls per forward); the additions keep their order
        const float* base = part + ((long)b * rows * C + c) * 2;
        int r = g;
        for (; r + 7 * 64 < rows; r += 8 * 64) {
            float2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float2*>(base + (long)(r + u * 64) * C * 2);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                s += (double)v[u].x;
                q += (double)v[u].y;
            }
        }
        for (; r < rows; r += 64) {
            const float2 v = *reinterpret_cast<const float2*>(base + (long)r * C * 2);
            s += (double)v.x;
            q += (double)v.y;
        }
    }
    red[threadIdx.x * 2] = s;
    red[threadIdx.x * 2 + 1] = q;
    __syncthreads();
    if (threadIdx.x >= 4 || c >= C) return;
    s = 0; q = 0;
    for (int g2 = 0; g2 < 64; ++g2) {
        s += red[(g2 * 4 + threadIdx.x) * 2];
        q += red[(g2 * 4 + threadIdx.x) * 2 + 1];
    }
    const double mu = s / (double)HW;
    inorm_store(mean, rstd, b * C + c, c, mu, q / (double)HW - mu * mu, eps, gamma, beta);
}

// grid (slices, B), block = a multiple of C / 4 threads: a thread keeps one channel quad of one image for its whole loop, so its
// mean / rstd are loaded once and the element loop has no index arithmetic beyond an add (the first version divided a 64-bit
// index three times per float4)
__global__ __launch_bounds__(256) void inorm_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const float* __restrict__ res,
                                                          const float* __restrict__ rmean, const float* __restrict__ rrstd,
                                                          float* __restrict__ out, long HW, int C, int relu) {
    const int cg = C / 4;
    const int b = blockIdx.y;
    const int c = (int)(threadIdx.x % cg) * 4;
    const long per = HW * cg;                                   // float4 per image
    const float4 mu = *reinterpret_cast<const float4*>(mean + (long)b * C + c);
    const float4 rs = *reinterpret_cast<const float4*>(rstd + (long)b * C + c);
    float4 m2 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = make_float4(1.f, 1.f, 1.f, 1.f);
    if (rmean) {
        m2 = *reinterpret_cast<const float4*>(rmean + (long)b * C + c);
        s2 = *reinterpret_cast<const float4*>(rrstd + (long)b * C + c);
    }
    const float4* xi = reinterpret_cast<const float4*>(x) + (long)b * per;
    const float4* ri = res ? reinterpret_cast<const float4*>(res) + (long)b * per : nullptr;
    float4* oi = reinterpret_cast<float4*>(out) + (long)b * per;
    const long step = (long)gridDim.x * blockDim.x;             // a multiple of cg: the channel quad stays put
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += step) {
        typedef float f4v __attribute__((ext_vector_type(4)));
        const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(xi + i));   // the raw convolution output is read once here
        const float4 v = make_float4(t.x, t.y, t.z, t.w);
        float4 y;
        y.x = (v.x - mu.x) * rs.x; y.y = (v.y - mu.y) * rs.y; y.z = (v.z - mu.z) * rs.z; y.w = (v.w - mu.w) * rs.w;
        if ((relu & 1) || ri) {
            y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f);
        }
        if (ri) {
            float4 r = ri[i];
            if (rmean) {
                r.x = (r.x - m2.x) * s2.x; r.y = (r.y - m2.y) * s2.y; r.z = (r.z - m2.z) * s2.z; r.w = (r.w - m2.w) * s2.w;
                if (relu & 2) {   // the residual is itself relu(norm(raw)): the stem's output, never materialised (raft_engine.cpp)
                    r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f);
                }
            }
            y.x = fmaxf(r.x + y.x, 0.f); y.y = fmaxf(r.y + y.y, 0.f); y.z = fmaxf(r.z + y.z, 0.f); y.w = fmaxf(r.w + y.w, 0.f);
        }
        oi[i] = y;
    }
}

// ---- convex upsample (the per-lane arithmetic lives in upsample_inl.h, shared with the upsample + warp kernel of warp_fast.hip)
__global__ __launch_bounds__(256) void upsample_kernel(const float* __restrict__ coords1, const float* __restrict__ mask,
                                                       float* __restrict__ flow_up, int h, int w, long M) {
    const OfxUpLane o = ofx_upsample_lane(coords1, mask, h, w, M);
    if (o.valid) ofx_upsample_store(o, flow_up, h, w);
}

// coords1 = pixel grid, frows (the flow rows convf1 reads, flow_head.hip) = 0, hx[:, flow_off:flow_off+2] = 0   (RAFT.initialize_flow, raft.py:63-70)
__global__ __launch_bounds__(256) void init_state_kernel(float* __restrict__ coords1, float* __restrict__ frows,
                                                         float* __restrict__ hx, int ldh, int flow_off, int h, int w, long M) {
    for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (long)gridDim.x * blockDim.x) {
        const int rem = (int)(m % ((long)h * w));
        const int y = rem / w, x = rem - y * w;
        reinterpret_cast<float2*>(coords1)[m] = make_float2((float)x, (float)y);
#pragma unroll
        for (int q = 0; q < 4; ++q) reinterpret_cast<float4*>(frows)[m * 4 + q] = make_float4(0.f, 0.f, 0.f, 0.f);
        hx[m * ldh + flow_off] = 0.f;
        hx[m * ldh + flow_off + 1] = 0.f;
    }
}

__global__ __launch_bounds__(256) void coords_to_flow_kernel(const float* __restrict__ coords1, float* __restrict__ flow,
                                                             int h, int w, long M) {
    for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (long)gridDim.x * blockDim.x) {
        const int rem = (int)(m % ((long)h * w));
        const int y = rem / w, x = rem - y * w;
        const float2 c = reinterpret_cast<const float2*>(coords1)[m];
        reinterpret_cast<float2*>(flow)[m] = make_float2(c.x - (float)x, c.y - (float)y);
    }
}

}  // namespace

int ofx_init_state(float* coords1, float* frows, float* hx, int ldh, int flow_off, int B, int h, int w, hipStream_t s) {
    const long M = (long)B * h * w;
    OfxProfScope prof("init_state", s);
    hipLaunchKernelGGL(init_state_kernel, dim3((unsigned)std::min<long>((M + 255) / 256, 8192)), dim3(256), 0, s, coords1,
                       frows, hx, ldh, flow_off, h, w, M);
    return ofx_launch_status();
}

// hx[b, :, 0:half] = ctx[idx[b], :, 0:half], hx[b, :, off2:off2+half] = ctx[idx[b], :, half:2*half]
// (per-pair gather of the per-image context features in the indexed-pairs forward)
__global__ __launch_bounds__(256) void ctx_gather_kernel(const float* __restrict__ ctx, const int* __restrict__ idx,
                                                         float* __restrict__ hx, int ldh, int off2, int half, long N, long total4) {
    const int q = half / 4;   // float4 per half row
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % (2 * q));
        const long pp = i / (2 * q);          // b*N + p
        const long b = pp / N, p = pp - b * N;
        const float4 v = reinterpret_cast<const float4*>(ctx + ((long)idx[b] * N + p) * (2 * half))[c4];
        float* dst = hx + pp * ldh + (c4 < q ? c4 * 4 : off2 + (c4 - q) * 4);
        *reinterpret_cast<float4*>(dst) = v;
    }
}

int ofx_ctx_gather(const float* ctx, const int* idx_dev, float* hx, int ldh, int off2, int half, int B, long N, hipStream_t s) {
    const long total4 = (long)B * N * (2 * half / 4);
    OfxProfScope prof("ctx_gather", s);
    hipLaunchKernelGGL(ctx_gather_kernel, dim3((unsigned)std::min<long>((total4 + 255) / 256, 8192)), dim3(256), 0, s, ctx, idx_dev,
                       hx, ldh, off2, half, N, total4);
    return ofx_launch_status();
}

int ofx_coords_to_flow(const float* coords1, float* flow, int B, int h, int w, hipStream_t s) {
    const long M = (long)B * h * w;
    OfxProfScope prof("coords_to_flow", s);
    hipLaunchKernelGGL(coords_to_flow_kernel, dim3((unsigned)std::min<long>((M + 255) / 256, 8192)), dim3(256), 0, s,
                       coords1, flow, h, w, M);
    return ofx_launch_status();
}

extern "C" {

int ofx_preprocess_u8(const uint8_t* img, float* out, long npix, int bgr, void* stream) {
    OFX_REQUIRE(img && out && npix > 0, OFX_EINVAL);
    OFX_REQUIRE(ofx_aligned16(out), OFX_EALIGN);
    hipStream_t s = (hipStream_t)stream;
    OfxProfScope prof("preprocess_u8", s);
    hipLaunchKernelGGL(preprocess_kernel, dim3((unsigned)std::min<long>((npix + 255) / 256, 16384)), dim3(256), 0, s, img,
                       out, npix, bgr);
    return ofx_launch_status();
}

int ofx_inorm_stats(const float* x, int ld, float* mean, float* rstd, float* scratch, int B, long HW, int C, float eps,
                    void* stream) {
    return ofx_inorm_stats_affine(x, ld, mean, rstd, scratch, B, HW, C, eps, nullptr, nullptr, (hipStream_t)stream);
}

}  // extern "C"

int ofx_inorm_stats_affine(const float* x, int ld, float* mean, float* rstd, float* scratch, int B, long HW, int C, float eps,
                           const float* gamma, const float* beta, hipStream_t stream) {
    OFX_REQUIRE(!gamma == !beta, OFX_EINVAL);
    OFX_REQUIRE(x && mean && rstd && scratch && B > 0 && HW > 0 && C > 0, OFX_EINVAL);
    OFX_REQUIRE(C % 4 == 0 && C <= 256 && ld % 4 == 0 && ld >= C && ofx_aligned16(x), OFX_EALIGN);
    OFX_REQUIRE((((uintptr_t)scratch) & 7u) == 0, OFX_EALIGN);
    hipStream_t s = (hipStream_t)stream;
    double* part = reinterpret_cast<double*>(scratch);   // needs B*slices*C*2 doubles (see ofx.h)
    const int slices = inorm_slices(B);
    {
        OfxProfScope prof("inorm_stats", s);
        hipLaunchKernelGGL(inorm_partial_kernel, dim3(slices, B), dim3(256), 0, s, x, ld, part, HW, C, slices);
    }
    int st = ofx_launch_status();
    if (st) return st;
    OfxProfScope prof("inorm_finalize", s);
    hipLaunchKernelGGL(inorm_finalize_kernel, dim3(ofx_cdiv(C, 16), B), dim3(256), 0, s, part, mean, rstd, HW, C, eps, slices, gamma, beta);
    return ofx_launch_status();
}

extern "C" {

int ofx_inorm_apply(const float* x, const float* mean, const float* rstd, const float* res, const float* res_mean,
                    const float* res_rstd, float* out, int B, long HW, int C, int relu, void* stream) {
    OFX_REQUIRE(x && mean && rstd && out && B > 0 && HW > 0 && C > 0, OFX_EINVAL);
    OFX_REQUIRE(C % 4 == 0 && ofx_aligned16(x) && ofx_aligned16(out) && ofx_aligned16(mean) && ofx_aligned16(rstd), OFX_EALIGN);
    if (res_mean) OFX_REQUIRE(res && res_rstd, OFX_EINVAL);
    const int cg = C / 4;
    OFX_REQUIRE(cg <= 256, OFX_EINVAL);
    const int bs = (256 / cg) * cg;                              // block size: the largest multiple of C / 4 up to 256
    const long per = HW * cg;
    // enough workgroups to fill the chip whatever the batch; each thread then walks its image slice with a fixed channel quad
    const long want = std::max<long>(1, (long)4096 / B);
    const unsigned slices = (unsigned)std::min<long>((per + bs - 1) / bs, want);
    hipStream_t s = (hipStream_t)stream;
    OfxProfScope prof("inorm_apply", s);
    // the image index rides on gridDim.y (<= 65535): larger batches go in several launches
    for (long b0 = 0; b0 < B; b0 += 65535) {
        const unsigned nb = (unsigned)std::min<long>(65535, B - b0);
        const long eo = b0 * HW * C, so = b0 * C;
        hipLaunchKernelGGL(inorm_apply_kernel, dim3(slices, nb), dim3(bs), 0, s, x + eo, mean + so, rstd + so, res ? res + eo : nullptr,
                           res_mean ? res_mean + so : nullptr, res_rstd ? res_rstd + so : nullptr, out + eo, HW, C, relu);
    }
    return ofx_launch_status();
}

int ofx_upsample_flow(const float* coords1, const float* mask, float* flow_up, int B, int h, int w, void* stream) {
    OFX_REQUIRE(coords1 && mask && flow_up && B > 0 && h > 0 && w > 0, OFX_EINVAL);
    OFX_REQUIRE((((uintptr_t)coords1) & 7u) == 0 && ofx_aligned16(flow_up) && ofx_aligned16(mask), OFX_EALIGN);
    const long M = (long)B * h * w;
    hipStream_t s = (hipStream_t)stream;
    OfxProfScope prof("upsample_flow", s);
    const long groups = (M / w) * ((w + 3) / 4);   // 4 coarse pixels per wavefront, 4 wavefronts per workgroup
    hipLaunchKernelGGL(upsample_kernel, dim3((unsigned)((groups + 3) / 4)), dim3(256), 0, s, coords1, mask, flow_up, h, w, M);
    return ofx_launch_status();
}

}  // extern "C"

int ofx_inorm_finalize_part(const float* part, float* mean, float* rstd, int B, int rows, long HW, int C, float eps, hipStream_t s,
                            const float* gamma, const float* beta) {
    OFX_REQUIRE(part && mean && rstd && B > 0 && rows > 0 && HW > 0 && C > 0 && !gamma == !beta, OFX_EINVAL);
    OfxProfScope prof("inorm_finalize", s);
    hipLaunchKernelGGL(inorm_finalize_part_kernel, dim3(ofx_cdiv(C, 4), B), dim3(256), 0, s, part, mean, rstd, HW, C, eps, rows, gamma, beta);
    return ofx_launch_status();
}

