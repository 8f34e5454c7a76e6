/*
 * ofx.h -- C ABI of the MI355X-native optical-flow hot path (libofx.so).
 *
 * Drop-in boundary for the flow / warp / mask path of zyddnys/sd_animation_optical_flow.
 * Every entry point takes plain device pointers and sizes plus a `hipStream_t` passed as void*;
 * there are no torch types in any signature.  All functions return 0 on success, a positive
 * hipError_t on a HIP failure, or a negative OFX_E* code on a precondition failure (the reference's
 * TORCH_CHECKs, RAFT/alt_cuda_corr/correlation.cpp:19-21, become return codes).  Work is enqueued
 * on the given stream and is asynchronous, like the reference's launches.
 *
 * Reference interface each group replaces (file:line relative to the reference repo):
 *   ofx_local_corr_fwd          RAFT/alt_cuda_corr/correlation.cpp:23-33,51-54  (alt_cuda_corr.forward)
 *                               RAFT/alt_cuda_corr/correlation_kernel.cu:18-119,260-286
 *   ofx_corr_volume / _lookup   RAFT/core/corr.py:13-60   (CorrBlock)
 *   ofx_conv2d, ofx_inorm_*     RAFT/core/extractor.py:118-192, RAFT/core/update.py:6-136
 *   ofx_upsample_flow           RAFT/core/raft.py:72-83
 *   ofx_raft_*                  RAFT/core/raft.py:86-144 ; ofgen_keyframe_inpaint.py:47-71 (RAFT_2)
 *   ofx_warp_*                  pdcnet_of.py:34-42 ; ofgen_keyframe_inpaint.py:92-98 (cv2.remap)
 *   ofx_generate_mask, ofx_dilate_u8, ofx_expand_mask, ofx_travel_distance, ofx_flow_magnitude, ofx_merge_images,
 *   ofx_mix_frames, ofx_conf_sum
 *   ofx_groupnorm, ofx_softmax_rows, ofx_attention_f32
 *                               ldm/modules/diffusionmodules/model.py:35-41,152-203 ; ldm/modules/attention.py:314,426
 *                               ofgen_keyframe_inpaint.py:113-133,237-248,306-322,676-688,968-973,995-1027
 *
 * Layout conventions: images and flow are HWC ("channels-last"); network activations are
 * NHWC fp32; a flow field is f32[H,W,2] = (dx, dy) exactly as `algo.calc` returns it
 * (pdcnet_of.py:72).
 */
#ifndef OFX_H
#define OFX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OFX_VERSION 100

/* negative error codes (positive values are hipError_t) */
#define OFX_EINVAL   (-1)   /* bad argument (null pointer, non-positive size, unsupported mode) */
#define OFX_EALIGN   (-2)   /* pointer / leading dimension not 16-byte aligned where required   */
#define OFX_ENOMEM   (-3)   /* workspace too small                                             */
#define OFX_EKEY     (-4)   /* weight tensor missing or of the wrong shape                      */
#define OFX_ENODEV   (-5)   /* no gfx950 device                                                 */

int ofx_version(void);
const char* ofx_error_string(int code);

/* ---------------------------------------------------------------- profiling (HIP events) */
/* When enabled, every kernel launch issued through this library is bracketed by a pair of
 * hipEvents recorded on the launch stream.  ofx_prof_collect synchronises, accumulates the elapsed
 * times per kernel name and writes a JSON object {"name": {"calls": n, "ms": total, "flops": executed}, ...}
 * ("flops" is non-zero for the GEMM/conv launches).
 * on = 1: one entry per kernel family; on = 2: the RAFT executor's convolutions are split per layer
 * ("family:layer"). */
int ofx_prof_enable(int on);
int ofx_prof_collect(char* json_out, size_t cap);

/* ---------------------------------------------------------------- warp */
#define OFX_WARP_BILINEAR  0   /* grid_sample(bilinear, zeros, align_corners=True) semantics  */
#define OFX_WARP_BICUBIC   1   /* float Keys cubic A=-0.75, zero border                        */
#define OFX_WARP_CV2_CUBIC 2   /* OpenCV remap INTER_CUBIC: 1/32-px coords, 15-bit weights     */

/* out[b,y,x,:] = frame[b', y + sign*flow[b,y,x,1], x + sign*flow[b,y,x,0], :].
 * frame_bstride = elements between consecutive frames (0 = one shared key frame for all B flows).
 * sign = +1: pdcnet_of.warp_frame; sign = -1: the RAFT-convention warp_frame. C in [1,4]. */
int ofx_warp_u8(const uint8_t* frame, long frame_bstride, const float* flow, uint8_t* out,
                int B, int H, int W, int C, int mode, float sign, void* stream);
int ofx_warp_f32(const float* frame, long frame_bstride, const float* flow, float* out,
                 int B, int H, int W, int C, int mode, float sign, void* stream);
/* cv2.resize(INTER_CUBIC) of an f32 HWC image (used by warp_frame_latent, pdcnet_of.py:24,30) */
int ofx_resize_cubic_f32(const float* src, float* dst, int B, int Hs, int Ws, int Hd, int Wd, int C,
                         void* stream);

/* ---------------------------------------------------------------- masks */
/* mask = dilate(255*(conf < thres), ellipse ksize); if log_conf != NULL, log_conf[conf<thres] = 0
 * in place (generate_mask, ofgen_keyframe_inpaint.py:317-322).  cmp_gt = 0: low = conf < thres;
 * cmp_gt = 1: low = !(conf > thres) (the keyframe path's convention, :995). ksize odd, <= 31;
 * ksize = 1 means no dilation. */
int ofx_generate_mask(const float* conf, float* log_conf, uint8_t* mask, int B, int H, int W,
                      float thres, int ksize, int cmp_gt, void* stream);
int ofx_dilate_u8(const uint8_t* in, uint8_t* out, int B, int H, int W, int ksize, void* stream);
/* out = mask | dilate(255*(gray(|laplacian(image)| mod 256) > edge_thres), ellipse ksize) */
int ofx_expand_mask(const uint8_t* mask, const uint8_t* image_bgr, uint8_t* out, uint8_t* scratch,
                    int B, int H, int W, int edge_thres, int ksize, void* stream);
/* v = |flow| with v[conf < conf_floor] = 0 (of_calc, ofgen_keyframe_inpaint.py:118-126) */
int ofx_travel_distance(const float* flow, const float* conf, float* out, int B, int H, int W,
                        float conf_floor, void* stream);
/* The RAFT-variant of_calc (reference ofgen.py:45-49, caller :137): v = sqrt(fx*fx + fy*fy) of a bare flow, n pixels; f32
 * multiply, multiply, add (no contraction) and a correctly rounded square root: numpy's bits. */
int ofx_flow_magnitude(const float* flow, float* out, long n, void* stream);
/* confidence_to_mask (:237-248): travel' = warp(travel, flow) + dist; travel'[conf<0.9] = 0;
 * raw = 255*(conf<0.9 | travel' > thres); travel'[travel'>thres] = 0.  `raw` is un-dilated; the
 * caller dilates with ofx_dilate_u8(ksize=15). travel_in and travel_out must not alias. */
int ofx_travel_mask(const float* conf, const float* flow, const float* dist, const float* travel_in,
                    float* travel_out, uint8_t* raw, int B, int H, int W, float thres, int warp_mode,
                    void* stream);
/* merge_images 'naive' (:676-681): out = base*(1-m) + second*m with m = (mask==255) */
int ofx_merge_images(const uint8_t* base, const uint8_t* second, const uint8_t* mask, uint8_t* out,
                     int B, int H, int W, int C, void* stream);
/* mix_propagated_ai_frame (:306-315) */
int ofx_mix_frames(const uint8_t* raw, const uint8_t* warped, const uint8_t* mask, uint8_t* out,
                   int B, int H, int W, int C, float ppw, void* stream);
/* sums[n] = sum over HW of conf_like[n, :, :, chan] for a f32[N,H,W,nchan] tensor (f64 accumulate);
 * used for `einops.reduce(flow_mat[...,2], 's t h w -> s', 'sum')` (:666, :1000) */
int ofx_conf_sum(const float* x, double* sums, int N, long HW, int nchan, int chan, void* stream);
/* EXTENSION (no reference counterpart: RAFT emits no confidence and PDCNet+ is not in the reference
 * tree): forward-backward consistency confidence.  flow_fw f32[B,H,W,2] is defined on the target
 * grid, flow_bw on the source grid; e = fw(p) + bw(p + fw(p)); log_conf = -|e|^2/(2 sigma^2);
 * conf = exp(log_conf), shaped like PDCNetPlus.calc's (confidence, log_confidence), pdcnet_of.py:73-74 */
int ofx_fb_confidence(const float* flow_fw, const float* flow_bw, float* conf, float* log_conf, int B,
                      int H, int W, float sigma, void* stream);
/* fused hot-path tail: warped = warp(frame, flow); mask = dilate(255*(conf<thres)) in one call. */
int ofx_warp_and_mask(const uint8_t* frame, long frame_bstride, const float* flow, const float* conf,
                      uint8_t* warped, uint8_t* mask, int B, int H, int W, int C, int warp_mode,
                      float sign, float thres, int ksize, int cmp_gt, void* stream);

/* ---------------------------------------------------------------- SD-inpaint hand-off */
/* What img2img_inpaint derives from (frame, reference, mask) before its first VAE call
 * (ofgen_keyframe_inpaint.py:255-290 -> guided_ldm_inpainting.py:290-316,139-154), bit-exact to Pillow:
 * PIL.ImageFilter.GaussianBlur(radius) on an 8-bit single-channel image [B,H,W] (BoxBlur.c: three extended-box
 * passes per axis).  scratch: B*H*W bytes, distinct from in / out; in == out is allowed. */
int ofx_gaussian_blur_u8(const uint8_t* in, uint8_t* out, uint8_t* scratch, int B, int H, int W, float radius,
                         void* stream);
/* PIL Image.resize((Wout, Hout)) with the default BICUBIC resample on an 8-bit channel (Resample.c: antialiased
 * support, 22-bit fixed-point taps, horizontal pass to 8 bits then vertical).  scratch: B*Hin*Wout bytes. */
int ofx_resize_bicubic_u8(const uint8_t* in, uint8_t* out, uint8_t* scratch, int B, int Hin, int Win, int Hout,
                          int Wout, void* stream);
/* image_bgr / reference_bgr u8[B,H,W,3] (cv2 order), image_mask u8[B,H,W] = the blurred mask, mask_latent
 * u8[B,h,w] = the blurred mask resized to the latent grid.  Writes
 *   image            f32[B,3,H,W]  RGB planar: Image.composite(reference, image, image_mask) / 127.5 - 1
 *   cond_mask        f32[B,H,W]    round(image_mask / 255)
 *   cond_image       f32[B,3,H,W]  image * (1 - cond_mask)
 *   latmask          f32[B,4,h,w]  around(mask_latent / 255), tiled over the 4 latent channels
 *   cond_mask_latent f32[B,h,w]    nearest-neighbour resize of cond_mask (F.interpolate default) */
int ofx_sd_handoff(const uint8_t* image_bgr, const uint8_t* reference_bgr, const uint8_t* image_mask,
                   const uint8_t* mask_latent, float* image, float* cond_image, float* cond_mask, float* latmask,
                   float* cond_mask_latent, int B, int H, int W, int h, int w, void* stream);

/* GroupNorm(groups, C, eps, affine) of an NHWC fp32 tensor [B,HW,C] (ldm/modules/diffusionmodules/model.py:40-41,
 * `Normalize` = 32 groups, eps 1e-6), optionally followed by x * sigmoid(x) (`nonlinearity`, :35-37): the pair in front
 * of every convolution of the VAE encoder.  gamma / beta: [C] or NULL.  Statistics in f64.  out may alias x.
 * scratch: ofx_groupnorm_scratch_bytes(B, C) bytes, 16-byte aligned.  B <= 65535 (OFX_EINVAL beyond: slice the batch). */
size_t ofx_groupnorm_scratch_bytes(int B, int C);
int ofx_groupnorm(const float* x, const float* gamma, const float* beta, float* out, void* scratch, size_t scratch_bytes,
                  int B, long HW, int C, int groups, float eps, int silu, void* stream);
/* in place: x[r][0..n) = softmax(x[r][0..n) * scale + bias[r % bias_rows][0..n)); columns n..ld-1 are set to 0 */
int ofx_softmax_rows(float* x, long rows, long ld, int n, float scale, const float* bias, long ld_bias, long bias_rows,
                     void* stream);
/* out[z] = softmax(q[z] k[z]^T * scale + bias) v[z] for z < BH; q [BH,Nq,D], k/v [BH,Nk,D], out [BH,Nq,D], fp32, D % 4 == 0.
 * The semantics of xformers.ops.memory_efficient_attention(q, k, v, attn_bias) (ldm/modules/attention.py:314,426) and of
 * AttnBlock.forward (model.py:179-203, BH = batch, D = channels).  bias: NULL, [Nq,Nk] shared by every z
 * (bias_bstride = 0) or [BH,Nq,Nk] (bias_bstride = Nq*Nk); -inf entries mask keys (a row with every key masked is NaN,
 * like softmax).  D in {40, 64, 80, 128, 160} (the UNet's head sizes) runs one fused kernel: online softmax, the scores
 * never leave the CU, ofx_attention_workspace_bytes() = 0 and `workspace` may be NULL.  Other D (the VAE's single
 * 512-wide head) run unfused: both GEMMs on the fp32 matrix cores, the score matrix in the workspace
 * (ofx_attention_workspace_bytes, 16-byte aligned; slice BH to bound it). */
size_t ofx_attention_workspace_bytes(int BH, int Nq, int Nk, int D);
int ofx_attention_f32(const float* q, const float* k, const float* v, const float* bias, long bias_bstride, float* out,
                      int BH, int Nq, int Nk, int D, float scale, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------- key-frame detector */
/* edges[b] = cv2.dilate(cv2.Canny(V, low, high), ones(ksize, ksize)) for BGR frames u8[B,H,W,3], with V = max(B,G,R)
 * (the HSV value channel) and low/high = int((1 -/+ 1/3) * np.median(V)) clipped to [0,255]
 * (_detect_edges, ofgen_keyframe_inpaint.py:161-192).  ksize odd (estimated_kernel_size, :153-158).
 * scratch: ofx_detect_edges_scratch_bytes(B,H,W) bytes, 256-byte aligned.  The hysteresis stage iterates to
 * convergence and SYNCHRONISES the stream once per batch of 8 sweeps (the only blocking call of the library).  Parity: OpenCV's algorithm restated; unpinned. */
size_t ofx_detect_edges_scratch_bytes(int B, int H, int W);
int ofx_detect_edges(const uint8_t* frames_bgr, uint8_t* edges, void* scratch, size_t scratch_bytes, int B, int H,
                     int W, int ksize, void* stream);
/* sums[b] = sum_i |a[b*a_bstride + i] - b_[b*b_bstride + i]|, i < n (mean_pixel_distance's numerator, :143-150);
 * a stride of 0 compares every image with one shared image.  sums: device u64[B]. */
int ofx_abs_diff_sum_u8(const uint8_t* a, long a_bstride, const uint8_t* b, long b_bstride,
                        unsigned long long* sums, int B, long n, void* stream);

/* ---------------------------------------------------------------- implicit-GEMM conv */
#define OFX_ACT_NONE    0
#define OFX_ACT_RELU    1
#define OFX_ACT_SIGMOID 2
#define OFX_ACT_TANH    3

#define OFX_EPI_PLAIN   0   /* y = act(acc*scale + shift); if res: y = relu(y + res)            */
#define OFX_EPI_GRU_ZR  1   /* Cout=2*hd: n<hd -> z=sigmoid -> aux_z ; n>=hd -> r -> aux_rh=r*h   */
#define OFX_EPI_GRU_Q   2   /* q=tanh; h = (1-z)*h + z*q written in place to aux_h               */
#define OFX_EPI_FLOW    3   /* Cout=2: coords1 += delta; flow=coords1-grid -> aux_h slot + flow4.  The generic form of the flow
                               head's second convolution; the RAFT executor itself runs flow_head.hip, which leaves convf1's
                               operand as 16-float flow rows instead of this [M][4] array */

/* Arithmetic of the matrix-core contractions.  FP32 is the reference's (and the default): v_mfma_f32_32x32x2_f32,
 * bit-identical to an fmaf chain.  BF16X3 is an opt-in fast mode: each fp32 operand is split on the fly into
 * hi = bf16(x), lo = bf16(x - hi) and the product is formed as hi*hi + hi*lo + lo*hi on the bf16 matrix cores
 * with fp32 accumulation (operands carry ~16 mantissa bits; measured end-to-end flow EPE ~1e-4 px). */
#define OFX_PREC_FP32   0
#define OFX_PREC_BF16X3 1
#define OFX_PREC_BF16X3_W 2   /* bf16x3 with `w` already in the split format of ofx_split_conv_weight */
#define OFX_PREC_BF16X6_W 4   /* bf16x6 with `w` already in the split format of ofx_split_conv_weight3 */
#define OFX_PREC_BF16X6 3     /* opt-in: fp32 operands split into THREE bf16 pieces (exact), the six products of weight >= 2^-16
                                 on the bf16 matrix cores, fp32 accumulate: fp32-level accuracy (dropped terms < 2^-23), not the
                                 bit pattern of an fmaf chain */

typedef struct ofx_conv_desc {
    /* input: NHWC fp32, up to two channel segments (torch.cat along C without materialising) */
    const float* in0; int ld0; int c0;
    const float* in1; int ld1; int c1;          /* in1 may be NULL (c1 = 0) */
    /* weights packed [Cout][Kpad], k = (ky*KW + kx)*(c0+c1) + c, Kpad = K rounded up to 32 */
    const float* w;
    const float* scale;                          /* [Cout] or NULL (=1) */
    const float* shift;                          /* [Cout] or NULL (=0) */
    float* out; int ldo;                         /* out[m*ldo + n]; may be NULL for GRU/FLOW epilogues */
    const float* res; int ldres;                 /* residual for EPI_PLAIN, or NULL */
    const float* nmean; const float* nrstd;      /* instance-norm(+ReLU) applied to in0 on load, [B][c0], or NULL */
    const float* addend; int ldadd;              /* optional pre-activation term: v += addend[m*ldadd + n]       */
    float* aux_z; float* aux_rh; float* aux_h; int ldh;   /* GRU buffers; hidden dim = Cout(Q) */
    float* aux_coords; float* aux_flow4;         /* EPI_FLOW only: coords1 [M][2], flow4 [M][4] = (fx, fy, 0, 0) per pixel */
    long a_zs, w_zs, o_zs; int nz;               /* batched-GEMM mode (nz>1): per-z strides in elements */
    int B, Hin, Win, Hout, Wout, Cout, KH, KW, stride, padH, padW;
    int act, epi;
    int tile;                                    /* 0 = auto; else [2000000000 +] BK*1000000 + BM*1000 + BN, e.g. 16128128;
                                                    tiles 128x{32,64,128,192}, 64x64; BK 16 or 32; the 2e9 marker selects the
                                                    paired-pipeline variant of the 64x64 / BK 32 tile (small grids) */
    int precision;                               /* OFX_PREC_FP32 (default, exact fp32 MFMA) or OFX_PREC_BF16X3 */
    void* splitk_ws;                             /* optional device scratch (256-byte aligned) for split-K on small grids: */
    size_t splitk_ws_bytes;                      /* partial tiles + per-tile arrival counters.  The first 64 KiB hold the
                                                    counters and must be ZERO before the first use (the kernel leaves them
                                                    zero); one scratch per stream that may run a convolution concurrently.
                                                    NULL / 0: never split. */
} ofx_conv_desc;

int ofx_conv2d(const ofx_conv_desc* d, void* stream);
/* host-side helper: OIHW fp32 -> packed [Cout][Kpad] with Cin padded to cin_pad (>= Cin, %4==0).
 * Returns Kpad (or negative error).  `out` may be NULL to query the size. */
long ofx_pack_conv_weight(const float* w_oihw, int Cout, int Cin, int KH, int KW, int cin_pad,
                          float* out);
/* Host-side: packed fp32 weights [n_floats] (n_floats % 4 == 0) -> the pre-split bf16x3 operand format: every
 * group of four consecutive k becomes 16 bytes [hi0 hi1 hi2 hi3 | lo0 lo1 lo2 lo3] with hi = bf16(x),
 * lo = bf16(x - hi), both round-to-nearest-even -- bit-identical to what the kernel's on-the-fly split makes.
 * Same size as the input; use with precision = OFX_PREC_BF16X3_W. Returns 0 or OFX_EINVAL. */
int ofx_split_conv_weight(const float* packed, long n_floats, float* out);
/* The same for the three-piece arithmetic (OFX_PREC_BF16X6_W): `out` holds 1.5 * n_floats floats -- first the [hi x4 | mid x4] groups
 * (16 bytes per four consecutive k), then the [lo x4] groups (8 bytes per four k); hi + mid + lo = x exactly unless lo underflows.
 * The whole matrix [Cout][Kpad] must be converted in one call (the lo groups are addressed from its end): a convolution that uses it
 * passes d->Cout = that row count and d->nz <= 1 (a row slice or a batched GEMM would read the lo groups from the wrong place;
 * ofx_conv2d returns OFX_EINVAL for nz > 1, and for a precision outside OFX_PREC_FP32 .. OFX_PREC_BF16X6_W). */
int ofx_split_conv_weight3(const float* packed, long n_floats, float* out);

/* instance norm statistics over HW per (b,c): mean and 1/sqrt(var+eps) (biased var), NHWC input; C <= 256.
 * scratch: max(B*64, min(B,7)*256) * C * 2 doubles (f64 partial sums per image slice), 8-byte aligned. */
int ofx_inorm_stats(const float* x, int ld, float* mean, float* rstd, float* scratch,
                    int B, long HW, int C, float eps, void* stream);
/* out = relu?( (x-mean)*rstd ) ; with res: out = relu( r + relu((x-mean)*rstd) ) where
 * r = res (res_mean==NULL) or (res-res_mean)*res_rstd -- or relu of that when bit 1 of `relu` is set (relu = 3: the residual is
 * itself the raw output of a normalised + ReLU layer, RAFT/core/extractor.py:160-165 feeding :44-56).  C % 4 == 0 and C <= 1024 (a thread keeps one channel quad:
 * OFX_EINVAL beyond); any B (batches over 65535 images are split into several launches). */
int ofx_inorm_apply(const float* x, const float* mean, const float* rstd, const float* res,
                    const float* res_mean, const float* res_rstd, float* out, int B, long HW, int C,
                    int relu, void* stream);
/* u8 HWC3 image -> f32 NHWC4 (4th channel 0), value 2*(x/255)-1 (raft.py:89-90); bgr=1 swaps to RGB */
int ofx_preprocess_u8(const uint8_t* img, float* out, long npix, int bgr, void* stream);

/* ---------------------------------------------------------------- correlation */
/* vol0[b,i,j] = <f1[b,i,:], f2[b,j,:]> / sqrt(D) and the avg-pooled pyramid (CorrBlock.__init__).
 * f1,f2: [B, h*w, D] (NHWC).  pyr[l] holds, for each of the B*h*w source pixels, that pixel's h_l x w_l slice
 * (h_l = h >> l, floor) in the BLOCKED layout the lookup reads: 4-row x 8-column blocks of 32 floats (128 bytes = one
 * HBM line), blocks row-major, padding elements zero:
 *     slice[((y / 4) * ceil(w_l / 8) + x / 8) * 32 + (y % 4) * 8 + (x % 8)] = corr(pixel, (y, x))
 * so pyr[l] is [B*h*w][ofx_corr_slice_floats(h_l, w_l)] and must be 16-byte aligned.  levels in [1,4]. */
int ofx_corr_slice_floats(int h_l, int w_l);      /* ceil(h_l/4) * ceil(w_l/8) * 32 */
int ofx_corr_volume(const float* f1, const float* f2, float* const* pyr, int B, int h, int w, int D,
                    int levels, void* stream);
/* The same pyramid on the "A-stationary" volume kernel (csrc/corr_split.hip: all of K = 256 for a wave's rows in registers, the other
 * operand streamed through LDS, whole-line stores):
 *   planes = 1 -> exact fp32 on v_mfma_f32_32x32x2_f32: BIT-IDENTICAL to ofx_corr_volume on every level (what ofx_raft_forward runs
 *                 wherever the batch fills the part);
 *   planes = 2 -> "bf16x3" (opt-in; RAFT/core/corr.py:52-60 computes in fp32): each fp32 operand = hi + lo bf16, products hh + hl + lh,
 *                 ~16 mantissa bits;
 *   planes = 3 -> "bf16x6" (opt-in): hi + mid + lo, the six products >= 2^-16: fp32-level accuracy;
 * both split forms on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  shared_f2 != 0: f2 is ONE feature map [h*w, D] shared by the
 * B pairs (a key frame).  Needs D == 256, h % 8 == 0, w % 16 == 0, levels >= 2 and (h*w)^2 * 4 < 2 GiB per pair; anything else returns
 * OFX_EINVAL (use ofx_corr_volume). */
int ofx_corr_volume_split(const float* f1, const float* f2, float* const* pyr, int B, int h, int w, int D,
                          int levels, int planes, int shared_f2, void* stream);
/* CorrBlock.__call__ on the blocked pyramid: out[m, l*(2r+1)^2 + i*(2r+1) + j] for coords [B*h*w][2];
 * out row stride ldo */
int ofx_corr_lookup(const float* const* pyr, const float* coords, float* out, int ldo, int B, int h,
                    int w, int levels, int radius, void* stream);
/* alt_cuda_corr.forward: fmap1 [B,H1,W1,C], fmap2 [B,H2,W2,C], coords [B,N,H1,W1,2] ->
 * corr [B,N,(2r+1)^2,H1,W1] (overwritten, not accumulated into).  C % 4 == 0. */
int ofx_local_corr_fwd(const float* fmap1, const float* fmap2, const float* coords, float* corr,
                       int B, int H1, int W1, int H2, int W2, int C, int N, int r, void* stream);
/* alt_cuda_corr.backward (correlation_kernel.cu:122-256,288-324): gradients of ofx_local_corr_fwd with respect to
 * the two feature maps for corr_grad f32[B,N,(2r+1)^2,H1,W1]; fmap1_grad f32[B,H1,W1,C] and fmap2_grad
 * f32[B,H2,W2,C] are overwritten (the reference returns fresh zero-initialised tensors; its third output,
 * coords_grad, is all zeros and is left to the caller).  fmap2_grad is accumulated with float atomics like the
 * reference's atomicAdd, so its low-order bits depend on scheduling.  Training-only in the reference. */
int ofx_local_corr_bwd(const float* fmap1, const float* fmap2, const float* coords, const float* corr_grad,
                       float* fmap1_grad, float* fmap2_grad, int B, int H1, int W1, int H2, int W2, int C, int N,
                       int r, void* stream);
/* 2x2 average pool of an NHWC tensor (AlternateCorrBlock pyramid, corr.py:68-72) */
int ofx_avgpool2_nhwc(const float* in, float* out, int B, int H, int W, int C, void* stream);

/* convex 8x upsample: coords1 [B*h*w][2], mask [B*h*w][576] -> flow_up f32[B, 8h, 8w, 2] */
int ofx_upsample_flow(const float* coords1, const float* mask, float* flow_up, int B, int h, int w,
                      void* stream);
/* ofx_upsample_flow + the bilinear backward warp of one shared frame u8 [8h][8w][3] in ONE kernel (the flow of a lane's four fine
 * pixels is sampled while it is still in registers): warped u8 [B][8h][8w][3]; flow_up f32 [B][8h][8w][2] or NULL (not written).
 * sign = +1 (PDCNet convention, pdcnet_of.py:34-42) or -1 (RAFT convention, ofgen_keyframe_inpaint.py:92-98).
 * Bit-identical to ofx_upsample_flow followed by ofx_warp_u8(OFX_WARP_BILINEAR) with frame_batch_stride = 0. */
int ofx_upsample_flow_warp(const float* coords1, const float* mask, float* flow_up, const uint8_t* frame,
                           uint8_t* warped, int B, int h, int w, float sign, void* stream);

/* ---------------------------------------------------------------- RAFT engine */
typedef struct ofx_tensor {            /* one entry of a checkpoint state_dict (host memory, fp32) */
    const char* name;                  /* reference key, e.g. "fnet.layer1.0.conv1.weight"        */
    const float* data;
    int ndim; long shape[4];
} ofx_tensor;

typedef struct ofx_raft ofx_raft;      /* opaque */

/* Builds the engine on the current device: packs and uploads all weights (folds cnet BatchNorm into
 * per-channel scale/shift).  Keys may carry the "module." DataParallel prefix. */
int ofx_raft_create(const ofx_tensor* tensors, int n, ofx_raft** out);
int ofx_raft_destroy(ofx_raft* r);
/* bytes of device workspace needed for a batch of B pairs of HxW images (H,W multiples of 8) */
size_t ofx_raft_workspace_bytes(const ofx_raft* r, int B, int H, int W);

#define OFX_RAFT_BGR          1   /* input images are BGR (calc) instead of RGB (calc_batch)       */
#define OFX_RAFT_SHARED_IMG2  2   /* image2 is ONE image shared by the whole batch (key frame)     */
#define OFX_RAFT_SHARED_IMG1  4   /* image1 is ONE image shared by the whole batch                 */
#define OFX_RAFT_ALT_CORR     8   /* on-the-fly local correlation instead of the volume (alt_cuda_corr) */
#define OFX_RAFT_BF16X3      16   /* opt-in: split-bf16 matrix-core arithmetic for every convolution / the volume */
#define OFX_RAFT_BF16X6      64   /* opt-in: three-piece split-bf16 arithmetic (OFX_PREC_BF16X6) for every convolution / the volume */
#define OFX_RAFT_BN_BATCH   128   /* context-encoder BatchNorm on the statistics of the image itself (the reference's RAFT_2 as written:
                                     a model never put in .eval(), one image per call) instead of the folded running statistics */
#define OFX_RAFT_SEPARATE_STATS 256 /* diagnostic: instance-norm statistics by their own f64 pass over the stored tensor instead of
                                     out of the convolution epilogues (fp32 partial sums per wave) */
#define OFX_RAFT_VOL_BF16X3  512  /* opt-in: ONLY the correlation-volume GEMM in split-bf16 (hi + lo) arithmetic, operands pre-split into
                                     bf16 planes (ofx_corr_volume_split); every convolution stays exact fp32 */
#define OFX_RAFT_VOL_BF16X6 1024  /* the same with three planes per operand (fp32-level accuracy) */
#define OFX_RAFT_SERIAL       32  /* keep every launch on the caller's stream (default: small batches run their
                                     independent chains on internal side streams, joined before returning) */

/* RAFT.forward(test_mode=True): image1/image2 u8 [B,H,W,3] on device -> flow_up f32[B,H,W,2]
 * (flow on image1's grid pointing into image2) and, if non-NULL, flow_low f32[B,H/8,W/8,2].
 * One call takes at most (2^31 - 4096) / ((H/8)*(W/8)*3072) pairs (113 at 512x768): the kernels address their
 * operands with 32-bit byte offsets; OFX_EINVAL beyond that -- pairs are independent, slice the batch. */
int ofx_raft_forward(ofx_raft* r, const uint8_t* image1, const uint8_t* image2, int B, int H, int W,
                     int iters, int flags, float* flow_up, float* flow_low, void* workspace,
                     size_t workspace_bytes, void* stream);
/* The same forward with the backward warp of ONE shared uint8 RGB frame (the rendered AI key frame) done inside the convex
 * upsample -- the tail of the hot path (pdcnet_of.py:34-42 in its bilinear mode; warp_sign = +1: out(y,x) = frame(y + fy, x + fx),
 * -1: the RAFT-side convention of ofgen_keyframe_inpaint.py:92-98): warped u8 [B,H,W,3] = the bilinear warp of warp_frame u8 [H,W,3]
 * along the final flow, bit-identical to ofx_raft_forward followed by ofx_warp_u8(bilinear).  flow_up may be NULL (then the full-
 * resolution flow is never written: 201 MB less traffic per 64 frames at 512x768). */
int ofx_raft_forward_warp(ofx_raft* r, const uint8_t* image1, const uint8_t* image2, int B, int H, int W,
                          int iters, int flags, float* flow_up, float* flow_low, const uint8_t* warp_frame,
                          float warp_sign, uint8_t* warped, void* workspace, size_t workspace_bytes, void* stream);
/* Indexed pairs ("next" row f1, KeyframeConv / calculate_pairwise, ofgen_keyframe_inpaint.py:627-668):
 * n_images unique uint8 frames [n,H,W,3] on the device and B pairs (idx1[b], idx2[b]) given as HOST int
 * arrays; flow b is defined on image idx1[b] and points into image idx2[b].  Every image is encoded once
 * (feature + context network), so N*(N-1) ordered pairs cost N encoder passes instead of 3*N*(N-1). */
size_t ofx_raft_workspace_bytes_pairs(const ofx_raft* r, int n_images, int B, int H, int W);
int ofx_raft_forward_pairs(ofx_raft* r, const uint8_t* images, int n_images, const int* idx1,
                           const int* idx2, int B, int H, int W, int iters, int flags, float* flow_up,
                           float* flow_low, void* workspace, size_t workspace_bytes, void* stream);
/* ofx_raft_forward_pairs with the tail of ofx_raft_forward_warp for its FIRST n_warp pairs: warped u8 [n_warp,H,W,3] = the bilinear
 * backward warp of the shared frame warp_frame u8 [H,W,3] along the final flow of pair b < n_warp, produced inside the convex upsample.
 * This is the call behind pdcnet_of's `calc_batch_device(..., warp_frame=)`: pairs [0, n) = frame -> key frame (warped: pdcnet_of.py:34-42
 * in its bilinear mode), pairs [n, 2n) = key frame -> frame (only feed the forward-backward confidence, no warp). */
int ofx_raft_forward_pairs_warp(ofx_raft* r, const uint8_t* images, int n_images, const int* idx1,
                                const int* idx2, int B, int H, int W, int iters, int flags, float* flow_up,
                                float* flow_low, const uint8_t* warp_frame, float warp_sign, int n_warp,
                                uint8_t* warped, void* workspace, size_t workspace_bytes, void* stream);
/* debug / stage-parity access to the buffers of the last forward: returns a device pointer and
 * element count for a named intermediate ("fmap1","fmap2","hx","corr","pyr0".."pyr3","mask",...) */
int ofx_raft_buffer(const ofx_raft* r, const char* name, void** ptr, size_t* nfloats);

#ifdef __cplusplus
}
#endif
#endif /* OFX_H */
