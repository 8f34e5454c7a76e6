// Flow-head output convolution: 3x3, 256 -> 2 channels, fused with coords1 += delta (RAFT/core/update.py:6-14,
// RAFT/core/raft.py:131).  With two output channels the implicit-GEMM kernel runs its smallest tile at 1/16
// utilisation (2.7 % of the step); here each lane owns four input channels of all nine taps and keeps its
// 72 weights in registers, a wavefront walks over pixels, and the two dot products are reduced with
// cross-lane adds.  Input rows are re-read from L1/L2 by the 3x3 window; HBM sees every byte once.
#include "ofx_internal.h"

namespace {

struct FlowHeadArgs {
    const float* x;        // [M][ldx], first 256 channels used
    const float* w;        // packed [2][Kpad], k = (ky*3 + kx)*256 + c
    const float* bias;     // [2]
    float* coords1;        // [M][2]  (in/out)
    float* hx_flow;        // &hx[0][flow_off], row stride ldh: flow = coords1 - coords0
    float* flow4;          // [M][4]
    int ldx, ldh, Kpad, h, w_, M;
};

__global__ __launch_bounds__(256) void flow_head_kernel(const FlowHeadArgs a) {
    const int lane = threadIdx.x & 63;
    const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;      // global wave id
    const int nw = (gridDim.x * blockDim.x) >> 6;
    // lane l owns input channels 4l..4l+3 for all 9 taps and both outputs: 72 weights in registers
    float4 w0[9], w1[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        w0[t] = *reinterpret_cast<const float4*>(a.w + t * 256 + lane * 4);
        w1[t] = *reinterpret_cast<const float4*>(a.w + a.Kpad + t * 256 + lane * 4);
    }
    const float b0 = a.bias[0], b1 = a.bias[1];
    const int hw = a.h * a.w_;
    for (int m = gw; m < a.M; m += nw) {
        const int rem = m % hw;
        const int y = rem / a.w_, x = rem - y * a.w_;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            if ((unsigned)yy < (unsigned)a.h && (unsigned)xx < (unsigned)a.w_) {     // wave-uniform
                const float4 v = *reinterpret_cast<const float4*>(a.x + (long)(m + (t / 3 - 1) * a.w_ + (t % 3 - 1)) * a.ldx + lane * 4);
                s0 = fmaf(v.x, w0[t].x, s0); s0 = fmaf(v.y, w0[t].y, s0); s0 = fmaf(v.z, w0[t].z, s0); s0 = fmaf(v.w, w0[t].w, s0);
                s1 = fmaf(v.x, w1[t].x, s1); s1 = fmaf(v.y, w1[t].y, s1); s1 = fmaf(v.z, w1[t].z, s1); s1 = fmaf(v.w, w1[t].w, s1);
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            s0 += __shfl_xor(s0, off, 64);
            s1 += __shfl_xor(s1, off, 64);
        }
        if (lane < 2) {
            const float delta = (lane == 0 ? s0 + b0 : s1 + b1);
            const float c1 = a.coords1[(long)m * 2 + lane] + delta;
            a.coords1[(long)m * 2 + lane] = c1;
            const float fl = c1 - (float)(lane == 0 ? x : y);
            a.hx_flow[(long)m * a.ldh + lane] = fl;
            a.flow4[(long)m * 4 + lane] = fl;
        }
    }
}

}  // namespace

int ofx_flow_head_launch(const float* x, int ldx, const float* w, int Kpad, const float* bias, float* coords1, float* hx_flow,
                         int ldh, float* flow4, int B, int h, int w_, hipStream_t s) {
    FlowHeadArgs a;
    a.x = x; a.w = w; a.bias = bias; a.coords1 = coords1; a.hx_flow = hx_flow; a.flow4 = flow4;
    a.ldx = ldx; a.ldh = ldh; a.Kpad = Kpad; a.h = h; a.w_ = w_;
    const long M = (long)B * h * w_;
    OFX_REQUIRE(M < (1L << 31), OFX_EINVAL);
    a.M = (int)M;
    const int blocks = (int)std::min<long>((M + 3) / 4, 256L * 16);     // up to 16 workgroups per CU, grid-stride over pixels
    OfxProfScope prof("flow_head", s);
    hipLaunchKernelGGL(flow_head_kernel, dim3(blocks), dim3(256), 0, s, a);
    return ofx_launch_status();
}
