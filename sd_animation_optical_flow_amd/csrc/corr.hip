// Correlation kernels of the flow path: pyramid pooling, the multi-scale bilinear lookup
// (CorrBlock.__call__) and the on-the-fly local correlation that replaces `alt_cuda_corr.forward`.
//
//   volume + pyramid   RAFT/core/corr.py:13-27,52-60  -- the GEMM itself runs on the fp32 matrix cores
//                      through the batched mode of the implicit-GEMM kernel (conv.hip); this file adds
//                      the 3-level average-pool pyramid, computed from one read of level 0 with the
//                      intermediate levels held in LDS.
//   lookup             RAFT/core/corr.py:29-50 ; RAFT/core/utils/utils.py:57-71.  HBM-bound gather:
//                      per pixel, four (2r+2)^2 windows (40-byte row runs) -> 324 contiguous floats.
//                      One wavefront per pixel: the 400 window taps are fetched once into LDS (all 81
//                      samples of a level share one fractional offset), the 324 outputs are written as
//                      one coalesced 1296-byte run.
//   local correlation  RAFT/alt_cuda_corr/correlation_kernel.cu:18-119 (forward only; the backward
//                      kernel is training-only and never reached by the reference's no_grad callers).
#include "ofx_internal.h"

namespace {

// ------------------------------------------------------------------------------------------
// pyramid levels 1..3 from level 0
// ------------------------------------------------------------------------------------------
struct PoolArgs {
    const float* l0;
    float* l1;
    float* l2;
    float* l3;
    int h0, w0, h1, w1, h2, w2, h3, w3;
    int levels;   // total pyramid levels (2..4)
};

__global__ __launch_bounds__(256) void pyramid_pool_kernel(const PoolArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* s1 = lds;                       // level 1 map
    float* s2 = lds + a.h1 * a.w1;         // level 2 map
    const long p = blockIdx.x;             // source pixel (row of the volume)
    const float* src = a.l0 + p * (long)a.h0 * a.w0;
    float* d1 = a.l1 + p * (long)a.h1 * a.w1;
    const int n1 = a.h1 * a.w1;
    if ((a.w0 & 3) == 0) {
        // rows are 16-byte aligned: one thread pools two horizontally adjacent outputs from two float4 loads,
        // four such pairs per trip so that 8 x 16 B are in flight per lane (same additions, same order)
        const int wp = a.w1 >> 1;                  // output pairs per row
        const int npair = a.h1 * wp;
        for (int i0 = 0; i0 < npair; i0 += 4 * 256) {
            float4 t[4], u[4];
            int o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = min(i0 + q * 256 + (int)threadIdx.x, npair - 1);
                const int y = i / wp, x2 = i - y * wp;
                o[q] = y * a.w1 + 2 * x2;
                const float* r = src + (long)(2 * y) * a.w0 + 4 * x2;
                t[q] = *reinterpret_cast<const float4*>(r);
                u[q] = *reinterpret_cast<const float4*>(r + a.w0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (i0 + q * 256 + (int)threadIdx.x >= npair) continue;
                float2 v;
                v.x = (((t[q].x + t[q].y) + u[q].x) + u[q].y) * 0.25f;   // avg_pool2d: window sum, then /4
                v.y = (((t[q].z + t[q].w) + u[q].z) + u[q].w) * 0.25f;
                *reinterpret_cast<float2*>(d1 + o[q]) = v;
                *reinterpret_cast<float2*>(s1 + o[q]) = v;
            }
        }
    } else
    for (int i = threadIdx.x; i < n1; i += 256) {
        const int y = i / a.w1, x = i - y * a.w1;
        float2 t, u;
        if (a.w0 & 1) {   // odd row length: rows are not 8-byte aligned
            const float* q = src + (long)(2 * y) * a.w0 + 2 * x;
            t = make_float2(q[0], q[1]);
            u = make_float2(q[a.w0], q[a.w0 + 1]);
        } else {
            t = *reinterpret_cast<const float2*>(src + (long)(2 * y) * a.w0 + 2 * x);
            u = *reinterpret_cast<const float2*>(src + (long)(2 * y + 1) * a.w0 + 2 * x);
        }
        const float v = (((t.x + t.y) + u.x) + u.y) * 0.25f;   // avg_pool2d: window sum, then /4
        d1[i] = v;
        s1[i] = v;
    }
    if (a.levels < 3) return;
    __syncthreads();
    float* d2 = a.l2 + p * (long)a.h2 * a.w2;
    const int n2 = a.h2 * a.w2;
    for (int i = threadIdx.x; i < n2; i += 256) {
        const int y = i / a.w2, x = i - y * a.w2;
        const float* r0 = s1 + (2 * y) * a.w1 + 2 * x;
        const float* r1 = r0 + a.w1;
        const float v = (((r0[0] + r0[1]) + r1[0]) + r1[1]) * 0.25f;
        d2[i] = v;
        s2[i] = v;
    }
    if (a.levels < 4) return;
    __syncthreads();
    float* d3 = a.l3 + p * (long)a.h3 * a.w3;
    const int n3 = a.h3 * a.w3;
    for (int i = threadIdx.x; i < n3; i += 256) {
        const int y = i / a.w3, x = i - y * a.w3;
        const float* r0 = s2 + (2 * y) * a.w2 + 2 * x;
        const float* r1 = r0 + a.w2;
        d3[i] = (((r0[0] + r0[1]) + r1[0]) + r1[1]) * 0.25f;
    }
}

// ------------------------------------------------------------------------------------------
// lookup
// ------------------------------------------------------------------------------------------
constexpr int kMaxLevels = 4;
constexpr int kMaxWin = 10;   // 2r+2 with r <= 4

struct LookupArgs {
    const float* pyr[kMaxLevels];
    int hl[kMaxLevels], wl[kMaxLevels];
    const float* coords;   // [M][2] (x, y)
    float* out;
    int ldo;
    long M;
    int levels, r;
};

// LEVELS / RADIUS are compile-time so that the tap loop is fully unrolled: all ceil(L*(2r+2)^2/64) gather
// loads of a wavefront are in flight together (with run-time bounds the loop serialised 7 HBM round trips
// behind integer divisions: 18 us per pixel, 16 % of HBM peak).  LEVELS = 0 selects the generic fallback.
template <int LEVELS, int RADIUS>
__global__ __launch_bounds__(256) void corr_lookup_kernel(const LookupArgs a) {
    __shared__ float win[4][kMaxLevels * kMaxWin * kMaxWin];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long m = (long)blockIdx.x * 4 + wave;
    const bool live = m < a.M;
    const int levels = LEVELS ? LEVELS : a.levels;
    const int r = LEVELS ? RADIUS : a.r;
    const int rd = 2 * r + 1, wn = rd + 1, wn2 = wn * wn;
    float cx = 0.f, cy = 0.f;
    if (live) {
        const float2 c = reinterpret_cast<const float2*>(a.coords)[m];
        cx = c.x;
        cy = c.y;
    }
    float* s = win[wave];
    constexpr int kRounds = LEVELS ? (LEVELS * (2 * RADIUS + 2) * (2 * RADIUS + 2) + 63) / 64 : 0;
    if (LEVELS) {
        float vals[kRounds ? kRounds : 1];
#pragma unroll
        for (int q = 0; q < kRounds; ++q) {
            const int t = q * 64 + lane;
            const int l = t / wn2;                       // compile-time divisors
            const int rem = t - l * wn2;
            const int ty = rem / wn, tx = rem - ty * wn;
            const float inv = 1.0f / (float)(1 << l);    // exact: coords / 2**l
            const float xs = cx * inv, ys = cy * inv;
            float v = 0.f;
            if (live && t < LEVELS * wn2 && fabsf(xs) < 1.0e7f && fabsf(ys) < 1.0e7f) {
                const int xx = (int)floorf(xs) - r + tx;
                const int yy = (int)floorf(ys) - r + ty;
                const int hl = a.hl[l], wl = a.wl[l];
                if ((unsigned)xx < (unsigned)wl && (unsigned)yy < (unsigned)hl)
                    v = a.pyr[l][m * (long)hl * wl + (long)yy * wl + xx];
            }
            vals[q] = v;
        }
#pragma unroll
        for (int q = 0; q < kRounds; ++q) {
            const int t = q * 64 + lane;
            if (t < LEVELS * wn2) s[t] = vals[q];
        }
    } else if (live) {
        for (int t = lane; t < levels * wn2; t += 64) {
            const int l = t / wn2;
            const int rem = t - l * wn2;
            const int ty = rem / wn, tx = rem - ty * wn;
            const float inv = 1.0f / (float)(1 << l);
            const float xs = cx * inv, ys = cy * inv;
            float v = 0.f;
            if (fabsf(xs) < 1.0e7f && fabsf(ys) < 1.0e7f) {
                const int xx = (int)floorf(xs) - r + tx;
                const int yy = (int)floorf(ys) - r + ty;
                if ((unsigned)xx < (unsigned)a.wl[l] && (unsigned)yy < (unsigned)a.hl[l])
                    v = a.pyr[l][m * (long)a.hl[l] * a.wl[l] + (long)yy * a.wl[l] + xx];
            }
            s[t] = v;
        }
    }
    // each wavefront owns its LDS window (win[wave]): LDS operations of one wave complete in order, so a
    // wave-level fence is enough and the four pixels of a workgroup never wait for each other
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (!live) return;
    float* o = a.out + m * (long)a.ldo;
    const int rd2 = rd * rd;
    const int nout = levels * rd2;
    constexpr int kOutRounds = LEVELS ? (LEVELS * (2 * RADIUS + 1) * (2 * RADIUS + 1) + 63) / 64 : 1;
#pragma unroll
    for (int q = 0; q < (LEVELS ? kOutRounds : 1); ++q) {
        for (int k = q * 64 + lane; k < (LEVELS ? min(nout, (q + 1) * 64) : nout); k += 64) {
            const int l = k / rd2;
            const int rem = k - l * rd2;
            const int i = rem / rd, j = rem - i * rd;   // i: x offset (slow), j: y offset (fast)
            const float inv = 1.0f / (float)(1 << l);
            const float xs = cx * inv, ys = cy * inv;
            const float fx = xs - floorf(xs), fy = ys - floorf(ys);
            const float* b = s + l * wn2 + j * wn + i;
            const float v00 = b[0], v01 = b[1], v10 = b[wn], v11 = b[wn + 1];
            float acc = v00 * ((1.f - fx) * (1.f - fy));
            acc = acc + v01 * (fx * (1.f - fy));
            acc = acc + v10 * ((1.f - fx) * fy);
            acc = acc + v11 * (fx * fy);
            o[k] = acc;
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

int ofx_corr_pool_launch(const float* l0, float* l1, float* l2, float* l3, int B, int h, int w, int levels, hipStream_t s) {
    PoolArgs a{};
    a.l0 = l0; a.l1 = l1; a.l2 = l2; a.l3 = l3;
    a.h0 = h; a.w0 = w; a.h1 = h / 2; a.w1 = w / 2; a.h2 = h / 4; a.w2 = w / 4; a.h3 = h / 8; a.w3 = w / 8;
    a.levels = levels;
    OFX_REQUIRE(a.h1 > 0 && a.w1 > 0, OFX_EINVAL);
    const size_t lds = sizeof(float) * ((size_t)a.h1 * a.w1 + (size_t)a.h2 * a.w2);
    OFX_REQUIRE(lds <= 64 * 1024, OFX_EINVAL);
    OfxProfScope prof("corr_pyramid_pool", s);
    hipLaunchKernelGGL(pyramid_pool_kernel, dim3((unsigned)((long)B * h * w)), dim3(256), lds, s, a);
    return ofx_launch_status();
}

extern "C" {

int ofx_corr_volume(const float* f1, const float* f2, float* const* pyr, int B, int h, int w, int D, int levels,
                    void* stream) {
    OFX_REQUIRE(f1 && f2 && pyr && B > 0 && h > 0 && w > 0 && D > 0, OFX_EINVAL);
    OFX_REQUIRE(levels >= 1 && levels <= kMaxLevels, OFX_EINVAL);
    OFX_REQUIRE(D % 32 == 0, OFX_EALIGN);
    for (int l = 0; l < levels; ++l) OFX_REQUIRE(pyr[l] != nullptr, OFX_EINVAL);
    const int N = h * w;
    // level 0: batched GEMM  vol[b] = f1[b] (N x D) * f2[b]^T (D x N) / sqrt(D)
    ofx_conv_desc d{};
    d.in0 = f1; d.ld0 = D; d.c0 = D;
    d.w = f2;
    d.out = pyr[0]; d.ldo = N;
    d.nz = B; d.a_zs = (long)N * D; d.w_zs = (long)N * D; d.o_zs = (long)N * N;
    d.B = 1; d.Hin = h; d.Win = w; d.Hout = h; d.Wout = w; d.Cout = N;
    d.KH = 1; d.KW = 1; d.stride = 1; d.padH = 0; d.padW = 0;
    d.act = OFX_ACT_NONE; d.epi = OFX_EPI_PLAIN;
    hipStream_t s = (hipStream_t)stream;
    int st = ofx_conv2d_alpha(&d, 1.0f / sqrtf((float)D), stream);   // D = 256 -> exactly /16
    if (st) return st;
    if (levels == 1) return 0;
    return ofx_corr_pool_launch(pyr[0], pyr[1], levels > 2 ? pyr[2] : nullptr, levels > 3 ? pyr[3] : nullptr, B, h, w, levels, s);
}

int ofx_corr_lookup(const float* const* pyr, const float* coords, float* out, int ldo, int B, int h, int w,
                    int levels, int radius, void* stream) {
    OFX_REQUIRE(pyr && coords && out && B > 0 && h > 0 && w > 0, OFX_EINVAL);
    OFX_REQUIRE(levels >= 1 && levels <= kMaxLevels && radius >= 0 && 2 * radius + 2 <= kMaxWin, OFX_EINVAL);
    OFX_REQUIRE(ldo >= levels * (2 * radius + 1) * (2 * radius + 1), OFX_EINVAL);
    OFX_REQUIRE((((uintptr_t)coords) & 7u) == 0, OFX_EALIGN);
    LookupArgs a{};
    for (int l = 0; l < levels; ++l) {
        OFX_REQUIRE(pyr[l] != nullptr, OFX_EINVAL);
        a.pyr[l] = pyr[l];
        a.hl[l] = h >> l;
        a.wl[l] = w >> l;
        OFX_REQUIRE(a.hl[l] > 0 && a.wl[l] > 0, OFX_EINVAL);
    }
    a.coords = coords; a.out = out; a.ldo = ldo; a.M = (long)B * h * w; a.levels = levels; a.r = radius;
    hipStream_t s = (hipStream_t)stream;
    OfxProfScope prof("corr_lookup", s);
    if (levels == 4 && radius == 4)
        hipLaunchKernelGGL((corr_lookup_kernel<4, 4>), dim3((unsigned)((a.M + 3) / 4)), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((corr_lookup_kernel<0, 0>), dim3((unsigned)((a.M + 3) / 4)), dim3(256), 0, s, a);
    return ofx_launch_status();
}

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
