// Convex 8x upsample of the flow (RAFT.upsample_flow, RAFT/core/raft.py:72-83) for ONE lane: shared by the plain upsample kernel
// (net_misc.hip) and the upsample + warp kernel (warp_fast.hip) so that both produce the same bits.
//
// A wavefront takes FOUR horizontally adjacent coarse pixels; lane = (pixel p, quad q): sub-pixels (i, j..j+3) with i = q >> 1,
// j = (q & 1) * 4.  The 9 x 64 mask logits of a pixel are read as 16-byte loads and the lane ends with the flow of four
// consecutive fine pixels: (ax.k, ay.k) for fine column 8 x + j + k of fine row 8 y + i.
#pragma once
#include "ofx_internal.h"

// the 9 x 64 mask logits of a pixel are read exactly once: non-temporal loads keep them from displacing the key frame and the
// coordinates in L2 (tools/upsample_warp_bench.py: 263 -> 250 us per 64 frames; non-temporal STORES of the outputs measured slower)
typedef float ofx_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ofx_nt_load4(const float4* p) {
    const ofx_f4v v = __builtin_nontemporal_load(reinterpret_cast<const ofx_f4v*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}

struct OfxUpLane {
    bool valid;
    int x, y, i, j;          // coarse pixel, sub-row, first sub-column
    long b;                  // image
    float4 ax, ay;           // flow x / y of the four fine pixels
};

__device__ __forceinline__ OfxUpLane ofx_upsample_lane(const float* __restrict__ coords1, const float* __restrict__ mask, int h, int w, long M) {
    OfxUpLane o;
    o.valid = false;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wq = (w + 3) >> 2;                                   // groups of 4 pixels per coarse row
    const long grp = (long)blockIdx.x * 4 + wave;
    const long rows = M / w;                                       // B * h
    if (grp >= rows * wq) return o;
    const long row = grp / wq;                                     // b*h + y
    const int xg = (int)(grp - row * wq);
    const int p = lane >> 4, q = lane & 15;
    const int x = xg * 4 + p;
    if (x >= w) return o;
    const int y = (int)(row % h);
    const long b = row / h;
    const long m = row * w + x;
    const float* mk = mask + m * 576 + q * 4;
    float4 lg[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) lg[k] = ofx_nt_load4(reinterpret_cast<const float4*>(mk + k * 64));
    float4 mx = lg[0];
#pragma unroll
    for (int k = 1; k < 9; ++k) {
        mx.x = fmaxf(mx.x, lg[k].x); mx.y = fmaxf(mx.y, lg[k].y); mx.z = fmaxf(mx.z, lg[k].z); mx.w = fmaxf(mx.w, lg[k].w);
    }
    float4 den = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        // v_exp_f32 path (~2 ulp): 576 exponentials per coarse pixel made the libm version VALU-bound
        lg[k].x = __expf(lg[k].x - mx.x); lg[k].y = __expf(lg[k].y - mx.y);
        lg[k].z = __expf(lg[k].z - mx.z); lg[k].w = __expf(lg[k].w - mx.w);
        den.x += lg[k].x; den.y += lg[k].y; den.z += lg[k].z; den.w += lg[k].w;
    }
    const float4 inv = make_float4(__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y), __builtin_amdgcn_rcpf(den.z),
                                   __builtin_amdgcn_rcpf(den.w));
    float4 ax = make_float4(0.f, 0.f, 0.f, 0.f), ay = ax;
    const long hw = (long)h * w;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
        float fx = 0.f, fy = 0.f;
        if ((unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w) {
            const float2 c = reinterpret_cast<const float2*>(coords1)[b * hw + (long)yy * w + xx];
            fx = 8.0f * (c.x - (float)xx);    // 8 * (coords1 - coords0)
            fy = 8.0f * (c.y - (float)yy);
        }
        const float4 wgt = make_float4(lg[k].x * inv.x, lg[k].y * inv.y, lg[k].z * inv.z, lg[k].w * inv.w);
        ax.x += wgt.x * fx; ax.y += wgt.y * fx; ax.z += wgt.z * fx; ax.w += wgt.w * fx;
        ay.x += wgt.x * fy; ay.y += wgt.y * fy; ay.z += wgt.z * fy; ay.w += wgt.w * fy;
    }
    o.valid = true;
    o.x = x; o.y = y; o.b = b;
    o.i = q >> 1; o.j = (q & 1) * 4;
    o.ax = ax; o.ay = ay;
    return o;
}

// the lane's four flow vectors -> flow_up f32[B, 8h, 8w, 2] (two 16-byte stores; the two lanes of a row and the four pixels of the
// wave make 128-byte output runs)
__device__ __forceinline__ void ofx_upsample_store(const OfxUpLane& o, float* __restrict__ flow_up, int h, int w) {
    const long W8 = (long)w * 8;
    float4* d = reinterpret_cast<float4*>(flow_up + 2 * ((o.b * h * 8 + (long)o.y * 8 + o.i) * W8 + (long)o.x * 8 + o.j));
    d[0] = make_float4(o.ax.x, o.ay.x, o.ax.y, o.ay.y);
    d[1] = make_float4(o.ax.z, o.ay.z, o.ax.w, o.ay.w);
}
