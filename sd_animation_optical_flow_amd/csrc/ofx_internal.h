// Internal helpers shared by the HIP translation units of libofx.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/ofx.h"

#define OFX_HIP_CHECK(expr)                         \
    do {                                            \
        hipError_t _e = (expr);                     \
        if (_e != hipSuccess) return (int)_e;       \
    } while (0)

#define OFX_REQUIRE(cond, code) \
    do {                        \
        if (!(cond)) return (code); \
    } while (0)

static inline bool ofx_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
static inline int ofx_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- profiling hooks (prof.cpp) -------------------------------------------------------
// Usage in a launcher:   OfxProfScope _p("kernel_name", stream);  kernel<<<...>>>(...);
// The scope records an event pair on `stream` when profiling is enabled and is free otherwise.
// ofx_prof_enable(2): launches are accumulated per "family:tag"; the RAFT executor tags each layer.
void ofx_prof_set_tag(const char* tag);   // thread-local; the pointer must outlive the launches it labels
struct OfxProfScope {
    int slot;
    hipStream_t stream;
    OfxProfScope(const char* name, hipStream_t s);
    ~OfxProfScope();
    void flops(double f);   // executed FLOPs of the bracketed launch (reported by ofx_prof_collect)
};

// One-shot completion event for the NEXT convolution / flow-head launch of this thread: the launcher hands it to
// hipExtLaunchKernelGGL as the kernel's own stop event, so the dependency edge to another stream rides on the dispatch packet's
// completion signal instead of a marker packet behind it (hipEventRecord): the caller's stream does not stall on the marker.
// The launcher that consumes it clears it; a caller that finds it still set after the call records the event the plain way.
extern thread_local hipEvent_t ofx_tl_stop_event;
#ifdef __HIPCC__
#include <hip/hip_ext.h>
#define OFX_LAUNCH(kern, grid, block, s, ...)                                                      \
    do {                                                                                           \
        hipEvent_t ofx_ev_ = ofx_tl_stop_event;                                                    \
        if (ofx_ev_) {                                                                             \
            ofx_tl_stop_event = nullptr;                                                           \
            hipExtLaunchKernelGGL(kern, grid, block, 0, s, nullptr, ofx_ev_, 0, __VA_ARGS__);      \
        } else {                                                                                   \
            hipLaunchKernelGGL(kern, grid, block, 0, s, __VA_ARGS__);                              \
        }                                                                                          \
    } while (0)
#endif

static inline int ofx_launch_status() {
    hipError_t e = hipGetLastError();
    return (int)e;
}

// conv.hip: ofx_conv2d with an extra scalar multiplier on the accumulator (out = act(acc*alpha*scale + shift))
extern "C" int ofx_conv2d_alpha(const ofx_conv_desc* d, float alpha, void* stream);
// conv.hip: the blocked correlation volume GEMM that also writes pyramid level 1 from its accumulators
int ofx_conv2d_volpool(const ofx_conv_desc* d, float alpha, float* pool_out, long pool_zs, int wb0, int wb1, int slice1, void* stream);

// conv.hip / net_misc.hip: instance-norm statistics out of the convolution epilogue (rows_per_image = 0: not produced, use
// ofx_inorm_stats) and their per-image reduction
int ofx_conv2d_stats(const ofx_conv_desc* d, float* part, size_t part_floats, int* rows_per_image, void* stream);
// gamma / beta (both or neither): an affine folded into the (mean, rstd) pair, (x - mean') * rstd' = (x - mu) * rs * gamma + beta
int ofx_inorm_finalize_part(const float* part, float* mean, float* rstd, int B, int rows, long HW, int C, float eps, hipStream_t s,
                            const float* gamma = nullptr, const float* beta = nullptr);
int ofx_inorm_stats_affine(const float* x, int ld, float* mean, float* rstd, float* scratch, int B, long HW, int C, float eps,
                           const float* gamma, const float* beta, hipStream_t stream);

// attn_flash.hip: fused attention for the UNet's head sizes (no workspace)
bool ofx_attention_flash_ok(int D);
int ofx_attention_flash_launch(const float* q, const float* k, const float* v, const float* bias, long bias_bstride, float* out, int BH, int Nq,
                               int Nk, int D, float scale, hipStream_t s);

// corr.hip / net_misc.hip: internal launchers used by the RAFT engine
int ofx_local_corr_launch(const float* f1, const float* f2, const float* coords, float* out, long sb, long sn,
                          long sc, long sp, int B, int H1, int W1, int H2, int W2, int C, int N, int r, float scale,
                          float cscale, hipStream_t s);
// blocked pyramid layout (corr.hip): floats per pixel slice of an hl x wl level; fmap rows -> blocked order
int ofx_corr_slice_floats_l(int hl, int wl);
// ofx_corr_lookup that also writes `pad` (<= 47; 4 levels, radius 4 only) zeros behind the features of every output row
int ofx_corr_lookup_pad(const float* const* pyr, const float* coords, float* out, int ldo, int pad, int B, int h, int w, int levels,
                        int radius, void* stream);
int ofx_corr_block_rows(const float* src, float* dst, int n, int h, int w, int D, hipStream_t s);
int ofx_corr_pool_launch(const float* l0, float* l1, float* l2, float* l3, int B, int h, int w, int levels, hipStream_t s,
                         bool from_l1 = false);
bool ofx_corr_volpool_ok(int h, int w);
// corr_split.hip: the volume GEMM in split-bf16 form (opt-in `volume_precision`): fp32 rows -> bf16 planes in MFMA fragment order
// (planes = 2: bf16x3, 3: bf16x6; quad = 1: rows in the quad-blocked column order of the streamed operand), and the A-stationary GEMM
// that writes level 0 (blocked) and level 1 of nz pairs.  ia / ib: device arrays of image indices per pair, or null with byte strides
bool ofx_corr_volsplit_ok(int h, int w, int D);
bool ofx_corr_volsplit_pays(int nz, int h, int w, int planes);   // enough (pair, row group) tasks to fill the CUs' rounds
size_t ofx_corr_planes_bytes(int h, int w, int planes);
int ofx_corr_split_planes(const float* src, void* dst, int n, int h, int w, int planes, int quad, float alpha, hipStream_t s);
int ofx_corr_vol_split_launch(const void* ap, const void* bp, const int* ia, const int* ib, long a_zs, long b_zs, float* l0, float* l1,
                              int nz, int h, int w, int planes, hipStream_t s);
// mask_bits.hip: binary threshold/edge source -> elliptical dilation on bit planes
enum { OFX_MSRC_CONF_LT = 0, OFX_MSRC_CONF_NGT = 1, OFX_MSRC_EDGES = 3 };   // conf < t | !(conf > t) | Laplacian edges
int ofx_mask_bits_launch(int src, const float* conf, float* log_conf, const uint8_t* image, const uint8_t* or_mask,
                         uint8_t* out, int B, int H, int W, float thres, int edge_thres, int r, const signed char* hw,
                         const char* name, hipStream_t s);
// warp_fast.hip: bilinear warp of one shared uint8 RGB key frame (0 = launched, OFX_EINVAL = shape not taken)
int ofx_warp_bilinear_shared_launch(const uint8_t* frame, const float* flow, uint8_t* out, int B, int H, int W, float sign,
                                    hipStream_t s);
// warp_fast.hip: convex upsample + bilinear warp of one shared key frame in one pass (pad = zero-bordered RGBX copy of the frame)
size_t ofx_warp_pad_bytes(int H, int W);
bool ofx_upsample_warp_ok(int B, int H, int W);
int ofx_warp_pad_launch(const uint8_t* frame, void* pad, int H, int W, hipStream_t s);
int ofx_upsample_warp_launch(const float* coords1, const float* mask, float* flow_up, const void* pad, uint8_t* warped, int B, int h, int w,
                             float sign, hipStream_t s);
int ofx_init_state(float* coords1, float* frows, float* hx, int ldh, int flow_off, int B, int h, int w, hipStream_t s);
int ofx_coords_to_flow(const float* coords1, float* flow, int B, int h, int w, hipStream_t s);
int ofx_flow_head_launch(const float* x, int ldx, const float* w, int Kpad, const float* bias, float* coords1, float* hx_flow,
                         int ldh, float* frows, int B, int h, int w_, hipStream_t s);
int ofx_ctx_gather(const float* ctx, const int* idx_dev, float* hx, int ldh, int off2, int half, int B, long N, hipStream_t s);

// ---- device helpers --------------------------------------------------------------------
#ifdef __HIPCC__
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Gate non-linearities of the conv epilogues on the hardware transcendental units (v_exp_f32 / v_rcp_f32,
// <= 2 ulp each; absolute error ~2e-7, far inside the 1e-3 px flow tolerance).  The libm forms (expf, a
// correctly rounded divide, OCML tanhf) cost 40-60 VALU instructions per element and showed up as a
// 25 % epilogue tax on the GRU convolutions.
#ifdef OFX_EXACT_GATES   // diagnostic build (tools/epe_curve.py): libm gates, to separate their error from everything else's
__device__ __forceinline__ float ofx_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float ofx_tanh(float x) { return tanhf(x); }
#else
__device__ __forceinline__ float ofx_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float ofx_tanh(float x) {
    const float t = __expf(2.0f * fminf(fmaxf(x, -15.0f), 15.0f));
    return (t - 1.0f) * __builtin_amdgcn_rcpf(t + 1.0f);
}
#endif

// Keys cubic weights, A = -0.75 (same polynomial form as OpenCV's interpolateCubic)
__device__ __forceinline__ void ofx_cubic_coeffs(float t, float w[4]) {
    const float A = -0.75f;
    float t1 = t + 1.0f;
    w[0] = __fsub_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fsub_rn(__fmul_rn(A, t1), 5.0f * A), t1), 8.0f * A), t1), 4.0f * A);
    w[1] = __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(__fmul_rn(A + 2.0f, t), A + 3.0f), t), t), 1.0f);
    float u = __fsub_rn(1.0f, t);
    w[2] = __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(__fmul_rn(A + 2.0f, u), A + 3.0f), u), u), 1.0f);
    w[3] = __fsub_rn(__fsub_rn(__fsub_rn(1.0f, w[0]), w[1]), w[2]);
}
#endif
