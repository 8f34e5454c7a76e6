// Flow-head output convolution: 3x3, 256 -> 2 channels, fused with coords1 += delta (RAFT/core/update.py:6-14,
// RAFT/core/raft.py:131).  With two output channels the implicit-GEMM kernel runs its smallest tile at 1/16
// utilisation (2.7 % of the step); here each lane owns four input channels of all nine taps and keeps its
// 72 weights in registers, a wavefront walks over strips of adjacent pixels, and the dot products are reduced
// with cross-lane adds.  HBM sees every input byte once.
#include "ofx_internal.h"

namespace {

struct FlowHeadArgs {
    const float* x;        // [M][ldx], first 256 channels used
    const float* w;        // packed [2][Kpad], k = (ky*3 + kx)*256 + c
    const float* bias;     // [2]
    float* coords1;        // [M][2]  (in/out)
    float* hx_flow;        // &hx[0][flow_off], row stride ldh: flow = coords1 - coords0
    float* frows;          // [M][16]: per pixel the flow of its 7-wide row neighbourhood, (x-3 .. x+3) x (fx, fy), 2 zero floats
    int ldx, ldh, Kpad, h, w_, M;
};

constexpr int kPW = 8;   // output pixels per strip row
constexpr int kPR = 4;   // output rows per strip

// One wavefront produces a strip of kPR x kPW adjacent pixels: the (kPR + 2) x (kPW + 2) input pixels it needs are each loaded
// once (1.9 pixel reads per output pixel instead of 9; 3.75 with one-row strips, 2.5 with two-row ones) and every loaded pixel feeds all the taps it
// belongs to.  Lane l owns input channels 4l..4l+3 of all nine taps and both outputs (72 weights in registers); the
// 2 * kPR * kPW = 64 partial sums are reduced across the 64 lanes with a transposing butterfly (63 cross-lane exchanges per
// strip instead of 6 per value); every lane ends up with one finished output.
__global__ __launch_bounds__(256, 2) void flow_head_kernel(const FlowHeadArgs a) {
    const int lane = threadIdx.x & 63;
    const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;      // global wave id
    const int nw = (gridDim.x * blockDim.x) >> 6;
    float4 w0[9], w1[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        w0[t] = *reinterpret_cast<const float4*>(a.w + t * 256 + lane * 4);
        w1[t] = *reinterpret_cast<const float4*>(a.w + a.Kpad + t * 256 + lane * 4);
    }
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, (short)0, a.M * a.ldx * 4, 0x00020000);
    const int spr = (a.w_ + kPW - 1) / kPW;          // strips per strip row
    const int rpi = (a.h + kPR - 1) / kPR;           // strip rows per image
    const int nimg = a.M / (a.h * a.w_);
    const int nstrips = nimg * rpi * spr;
    for (int sidx = gw; sidx < nstrips; sidx += nw) {
        const int srow = sidx / spr;                 // image * rpi + strip row
        const int x0 = (sidx - srow * spr) * kPW;
        const int img = srow / rpi;
        const int y0 = (srow - img * rpi) * kPR;     // first output row of the strip
        const int row0 = img * a.h + y0;             // b*h + y of that row
        float acc[2 * kPR * kPW];                    // [output row][pixel][output]
#pragma unroll
        for (int i = 0; i < 2 * kPR * kPW; ++i) acc[i] = 0.f;
        // rows are read through a buffer descriptor: one VGPR offset (lane * 16), the pixel as a scalar
        // offset, out-of-image taps as an out-of-range offset (the hardware returns 0)
#pragma unroll
        for (int r = 0; r < kPR + 2; ++r) {          // input row y0 - 1 + r
            const bool yin = (unsigned)(y0 + r - 1) < (unsigned)a.h;                       // wave-uniform
            const int pix0 = (row0 + r - 1) * a.w_ + x0 - 1;
            float4 v[kPW + 2];
#pragma unroll
            for (int c = 0; c < kPW + 2; ++c) {
                const bool in = yin && (unsigned)(x0 + c - 1) < (unsigned)a.w_;
                const auto raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, in ? lane * 16 : 0x7FFFFFF0, in ? (pix0 + c) * a.ldx * 4 : 0, 0);
                v[c] = make_float4(__uint_as_float(raw[0]), __uint_as_float(raw[1]), __uint_as_float(raw[2]), __uint_as_float(raw[3]));
            }
#pragma unroll
            for (int oy = 0; oy < kPR; ++oy) {
                const int ky = r - oy;               // output row y0 + oy reads input row y0 + oy + ky - 1
                if (ky < 0 || ky > 2) continue;
#pragma unroll
                for (int c = 0; c < kPW + 2; ++c)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int j = c - kx;        // output pixel x0 + j reads input column x0 + j + kx - 1 = x0 + c - 1
                        if (j < 0 || j >= kPW) continue;
                        const float4 q0 = w0[ky * 3 + kx], q1 = w1[ky * 3 + kx];
                        float s0 = acc[(oy * kPW + j) * 2], s1 = acc[(oy * kPW + j) * 2 + 1];
                        s0 = fmaf(v[c].x, q0.x, s0); s0 = fmaf(v[c].y, q0.y, s0); s0 = fmaf(v[c].z, q0.z, s0); s0 = fmaf(v[c].w, q0.w, s0);
                        s1 = fmaf(v[c].x, q1.x, s1); s1 = fmaf(v[c].y, q1.y, s1); s1 = fmaf(v[c].z, q1.z, s1); s1 = fmaf(v[c].w, q1.w, s1);
                        acc[(oy * kPW + j) * 2] = s0;
                        acc[(oy * kPW + j) * 2 + 1] = s1;
                    }
            }
        }
        // transposing butterfly: after the step with distance d the lane keeps half of its values, chosen by
        // its bit d; 64 -> 32 -> 16 -> 8 -> 4 -> 2 -> 1 values.  Lane l ends with the full sum of value id = l.
        static_assert(kPW == 8 && kPR == 4, "the butterfly below is written for 64 values");
        float r32[32], r16[16], r8[8], r4[4], r2[2], r1;
        {
            const bool hi = (lane & 32) != 0;
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const float send = hi ? acc[i] : acc[i + 32];
                const float keep = hi ? acc[i + 32] : acc[i];
                r32[i] = keep + __shfl_xor(send, 32, 64);
            }
        }
        {
            const bool hi = (lane & 16) != 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float send = hi ? r32[i] : r32[i + 16];
                const float keep = hi ? r32[i + 16] : r32[i];
                r16[i] = keep + __shfl_xor(send, 16, 64);
            }
        }
        {
            const bool hi = (lane & 8) != 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float send = hi ? r16[i] : r16[i + 8];
                const float keep = hi ? r16[i + 8] : r16[i];
                r8[i] = keep + __shfl_xor(send, 8, 64);
            }
        }
        {
            const bool hi = (lane & 4) != 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float send = hi ? r8[i] : r8[i + 4];
                const float keep = hi ? r8[i + 4] : r8[i];
                r4[i] = keep + __shfl_xor(send, 4, 64);
            }
        }
        {
            const bool hi = (lane & 2) != 0;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float send = hi ? r4[i] : r4[i + 2];
                const float keep = hi ? r4[i + 2] : r4[i];
                r2[i] = keep + __shfl_xor(send, 2, 64);
            }
        }
        {
            const bool hi = (lane & 1) != 0;
            const float send = hi ? r2[0] : r2[1];
            const float keep = hi ? r2[1] : r2[0];
            r1 = keep + __shfl_xor(send, 1, 64);
        }
        const int id = lane;
        const int oy = id >> 4, j = (id >> 1) & 7, o = id & 1;
        const int x = x0 + j, y = y0 + oy;
        if (x < a.w_ && y < a.h) {
            const long m = (long)(row0 + oy) * a.w_ + x;
            // delta = conv + bias first, THEN coords1 + delta (raft.py:128-131).  (coords1 + conv) + bias rounds twice at the magnitude of
            // the coordinate, and the second rounding is the same for every pixel of a binade (a constant added to multiples of one
            // ulp): a coherent -frac(bias / ulp) * ulp per iteration that grew linearly to 2.7e-4 px after 20 iterations and the 8x
            // upsample (profiles/r03_epe_curve.txt)
            const float c1 = a.coords1[m * 2 + o] + (r1 + a.bias[o]);
            a.coords1[m * 2 + o] = c1;
            const float fl = c1 - (float)(o == 0 ? x : y);
            a.hx_flow[m * a.ldh + o] = fl;
            // convf1's operand (update.py:93, 7x7 on the 2-channel flow): pixel x' = x + d keeps the flow of x in slot 3 - d of its
            // row, so that the 7x7 convolution runs as a 7x1 one over 16-float rows -- K = 7 * 16 = 112 with 98 real products where
            // the NHWC gather of 4-float pixels needed 7 * 7 * 4 = 196 with half of them zeros.  Slots of neighbours outside the
            // image are never written and keep the zeros of ofx_init_state (= the convolution's zero padding).
#pragma unroll
            for (int d = -3; d <= 3; ++d)
                if ((unsigned)(x + d) < (unsigned)a.w_) a.frows[(m + d) * 16 + (3 - d) * 2 + o] = fl;
        }
    }
}

}  // namespace

int ofx_flow_head_launch(const float* x, int ldx, const float* w, int Kpad, const float* bias, float* coords1, float* hx_flow,
                         int ldh, float* frows, int B, int h, int w_, hipStream_t s) {
    FlowHeadArgs a;
    a.x = x; a.w = w; a.bias = bias; a.coords1 = coords1; a.hx_flow = hx_flow; a.frows = frows;
    a.ldx = ldx; a.ldh = ldh; a.Kpad = Kpad; a.h = h; a.w_ = w_;
    const long M = (long)B * h * w_;
    OFX_REQUIRE(M * ldx * 4 < (1L << 31) - 64, OFX_EINVAL);      // 32-bit byte offsets into x
    a.M = (int)M;
    const long strips = (long)B * ((h + kPR - 1) / kPR) * ((w_ + kPW - 1) / kPW);
    OfxProfScope prof("flow_head", s);
    if (strips <= 1024) {
        // a single pair has 192 strips: as four-wave workgroups they would sit on 48 of the 256 CUs, each wave waiting on its own 60
        // dependent-latency loads; one wave per workgroup spreads them over the chip (12.6 -> 11.1 us per launch on one 512x768 pair)
        OFX_LAUNCH(flow_head_kernel, dim3((unsigned)strips), dim3(64), s, a);
        return ofx_launch_status();
    }
    const int blocks = (int)std::min<long>((strips + 3) / 4, 256L * 16);   // one strip per wavefront, grid-stride beyond 16 workgroups per CU
    OFX_LAUNCH(flow_head_kernel, dim3(blocks), dim3(256), s, a);
    return ofx_launch_status();
}
