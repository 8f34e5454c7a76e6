// SD-inpaint hand-off (SURVEY section 8 "next" row f3): what img2img_inpaint computes from the flow path's
// outputs before its first VAE call (ofgen_keyframe_inpaint.py:255-290 -> guided_ldm_inpainting.py:290-316,
// 139-154), kept on the device:
//
//   image_mask = GaussianBlur(mask_blur)(mask)                       Pillow BoxBlur.c: 3 extended-box passes per axis
//   image      = composite(reference, image, image_mask) / 127.5 - 1 Pillow Paste.c BLEND8 (div255), RGB, planar f32
//   latmask    = around(resize_bicubic(image_mask, h/8 x w/8) / 255) Pillow Resample.c, 22-bit fixed-point taps
//   conditioning mask / image, nearest-neighbour latent conditioning mask
//
// All of it is integer arithmetic on 8-bit data (plus Pillow's double-precision tap weights, recomputed per
// output sample with contraction off so that they are the host library's values bit for bit).  The byte counts
// are tiny (0.4 MB mask per frame); the point is that (warped, mask) never leave HBM between the flow path and
// the diffusion model.
#include "ofx_internal.h"

#include <cmath>

namespace {

// ------------------------------------------------------------------------------------------
// Pillow GaussianBlur on an 8-bit single-channel image
// ------------------------------------------------------------------------------------------
// BoxBlur.c:_gaussian_blur_radius (all locals are C floats there)
float gaussian_box_radius(float radius, int passes) {
    float sigma2, L, l, a;
    sigma2 = radius * radius / passes;
    L = (float)std::sqrt(12.0 * sigma2 + 1.0);
    l = (float)std::floor((L - 1.0) / 2.0);
    a = (2 * l + 1) * (l * (l + 1) - 3 * sigma2);
    a /= 6 * (sigma2 - (l + 1) * (l + 1));
    return l + a;
}

// one workgroup per image row: the row lives in LDS, `passes` extended-box passes ping-pong between two copies
// (ImagingLineBoxBlur8: replicated edges, out = (acc*ww + (left+right)*fw + 2^23) >> 24 in uint32)
__global__ __launch_bounds__(256) void box_blur_rows_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int W,
                                                            int r, unsigned ww, unsigned fw, int passes) {
    extern __shared__ uint8_t srow[];
    uint8_t* a = srow;
    uint8_t* b = srow + W;
    const long base = (long)blockIdx.x * W;
    for (int x = threadIdx.x; x < W; x += 256) a[x] = in[base + x];
    __syncthreads();
    for (int p = 0; p < passes; ++p) {
        for (int x = threadIdx.x; x < W; x += 256) {
            unsigned acc = 0;
            for (int d = -r; d <= r; ++d) acc += a[min(max(x + d, 0), W - 1)];
            const unsigned far_ = (unsigned)a[min(max(x - r - 1, 0), W - 1)] + (unsigned)a[min(max(x + r + 1, 0), W - 1)];
            const unsigned bulk = acc * ww + far_ * fw;
            b[x] = (uint8_t)((bulk + (1u << 23)) >> 24);
        }
        __syncthreads();
        uint8_t* t = a; a = b; b = t;
    }
    for (int x = threadIdx.x; x < W; x += 256) out[base + x] = a[x];
}

// out[b][x][y] = in[b][y][x]
__global__ __launch_bounds__(256) void transpose_u8_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int H, int W) {
    __shared__ uint8_t tile[32][33];
    const long b = blockIdx.z;
    const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int y = y0 + j, x = x0 + tx;
        if (y < H && x < W) tile[j][tx] = in[(b * H + y) * (long)W + x];
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int x = x0 + j, y = y0 + tx;
        if (y < H && x < W) out[(b * W + x) * (long)H + y] = tile[tx][j];
    }
}

// ------------------------------------------------------------------------------------------
// Pillow Image.resize(BICUBIC) on an 8-bit channel (Resample.c)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double bicubic_filter(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

constexpr int kPrecBits = 32 - 8 - 2;

// One output sample: precompute_coeffs + normalize_coeffs_8bpc for position `xx`, then the fixed-point dot product.
// `stride` is the distance between consecutive input samples along the resampled axis.
__device__ __forceinline__ uint8_t resample_one(const uint8_t* __restrict__ line, long stride, int in_size, int out_size, int xx) {
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * filterscale;
    const double ss = 1.0 / filterscale;
    const double center = (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) ww += bicubic_filter((x + xmin - center + 0.5) * ss);
    int acc = 1 << (kPrecBits - 1);
    for (int x = 0; x < xmax; ++x) {
        double k = bicubic_filter((x + xmin - center + 0.5) * ss);
        if (ww != 0.0) k /= ww;
        const int kk = k < 0 ? (int)(-0.5 + k * (double)(1 << kPrecBits)) : (int)(0.5 + k * (double)(1 << kPrecBits));
        acc += (int)line[(long)(x + xmin) * stride] * kk;
    }
    const int v = acc >> kPrecBits;
    return (uint8_t)min(max(v, 0), 255);
}

// horizontal pass: in [B*H][Win] -> out [B*H][Wout]
__global__ __launch_bounds__(256) void resample_rows_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, long rows,
                                                            int Win, int Wout) {
    const long total = rows * Wout;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long row = i / Wout;
        const int xx = (int)(i - row * Wout);
        out[i] = resample_one(in + row * Win, 1, Win, Wout, xx);
    }
}

// vertical pass: in [B][Hin][W] -> out [B][Hout][W]
__global__ __launch_bounds__(256) void resample_cols_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int B, int Hin,
                                                            int Hout, int W) {
    const long total = (long)B * Hout * W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int x = (int)(i % W);
        const long t = i / W;
        const int yy = (int)(t % Hout);
        const long b = t / Hout;
        out[i] = resample_one(in + b * (long)Hin * W + x, W, Hin, Hout, yy);
    }
}

// ------------------------------------------------------------------------------------------
// composite + normalisation + conditioning tensors
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned div255(unsigned a) {   // Paste.c DIV255
    const unsigned tmp = a + 128u;
    return ((tmp >> 8) + tmp) >> 8;
}

__global__ __launch_bounds__(256) void handoff_pixels_kernel(const uint8_t* __restrict__ image_bgr, const uint8_t* __restrict__ ref_bgr,
                                                             const uint8_t* __restrict__ mask, float* __restrict__ image,
                                                             float* __restrict__ cond_image, float* __restrict__ cond_mask, long HW,
                                                             long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long b = i / HW, p = i - b * HW;
        const unsigned m = mask[i];
        const float cm = m >= 128u ? 1.0f : 0.0f;           // round(m / 255): no ties on 8-bit values
        cond_mask[i] = cm;
        const float keep = 1.0f - cm;
#pragma unroll
        for (int c = 0; c < 3; ++c) {                        // output channel c = R,G,B <- input channel 2-c of BGR
            const unsigned a = image_bgr[i * 3 + (2 - c)], r = ref_bgr[i * 3 + (2 - c)];
            const unsigned v = div255(a * (255u - m) + r * m);
            const float f = (float)v / 127.5f - 1.0f;
            image[(b * 3 + c) * HW + p] = f;
            cond_image[(b * 3 + c) * HW + p] = f * keep;
        }
    }
}

__global__ __launch_bounds__(256) void handoff_latent_kernel(const uint8_t* __restrict__ mask, const uint8_t* __restrict__ mask_lat,
                                                             float* __restrict__ latmask, float* __restrict__ cond_mask_lat, int H, int W,
                                                             int h, int w, long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int x = (int)(i % w);
        const long t = i / w;
        const int y = (int)(t % h);
        const long b = t / h;
        const float lm = mask_lat[i] >= 128 ? 1.0f : 0.0f;   // np.around(v / 255)
#pragma unroll
        for (int c = 0; c < 4; ++c) latmask[((b * 4 + c) * h + y) * (long)w + x] = lm;
        // F.interpolate(mode='nearest'): src = floor(dst * (in / out)) evaluated in float like ATen
        const int sy = min((int)floorf((float)y * ((float)H / (float)h)), H - 1);
        const int sx = min((int)floorf((float)x * ((float)W / (float)w)), W - 1);
        cond_mask_lat[i] = mask[(b * H + sy) * (long)W + sx] >= 128 ? 1.0f : 0.0f;
    }
}

int grid1d(long total) { return (int)std::min<long>((total + 255) / 256, 256L * 32); }

}  // namespace

extern "C" {

int ofx_gaussian_blur_u8(const uint8_t* in, uint8_t* out, uint8_t* scratch, int B, int H, int W, float radius, void* stream) {
    OFX_REQUIRE(in && out && B > 0 && H > 0 && W > 0 && radius >= 0.f, OFX_EINVAL);
    hipStream_t s = (hipStream_t)stream;
    const long n = (long)B * H * W;
    if (radius == 0.f) {
        if (in != out) OFX_HIP_CHECK(hipMemcpyAsync(out, in, (size_t)n, hipMemcpyDeviceToDevice, s));
        return 0;
    }
    OFX_REQUIRE(scratch && scratch != in && scratch != out, OFX_EINVAL);
    OFX_REQUIRE(2L * std::max(H, W) <= 60 * 1024, OFX_EINVAL);                       // a row (and a column) fits LDS twice
    const int passes = 3;
    const float fr = gaussian_box_radius(radius, passes);
    const int r = (int)fr;
    const unsigned ww = (unsigned)((float)(1u << 24) / (fr * 2 + 1));               // ImagingHorizontalBoxBlur
    const unsigned fw = ((1u << 24) - (unsigned)(r * 2 + 1) * ww) / 2;
    OfxProfScope prof("gaussian_blur_u8", s);
    hipLaunchKernelGGL(box_blur_rows_kernel, dim3((unsigned)((long)B * H)), dim3(256), (size_t)2 * W, s, in, out, W, r, ww, fw, passes);
    hipLaunchKernelGGL(transpose_u8_kernel, dim3(ofx_cdiv(W, 32), ofx_cdiv(H, 32), B), dim3(256), 0, s, out, scratch, H, W);
    hipLaunchKernelGGL(box_blur_rows_kernel, dim3((unsigned)((long)B * W)), dim3(256), (size_t)2 * H, s, scratch, scratch, H, r, ww, fw, passes);
    hipLaunchKernelGGL(transpose_u8_kernel, dim3(ofx_cdiv(H, 32), ofx_cdiv(W, 32), B), dim3(256), 0, s, scratch, out, W, H);
    return ofx_launch_status();
}

int ofx_resize_bicubic_u8(const uint8_t* in, uint8_t* out, uint8_t* scratch, int B, int Hin, int Win, int Hout, int Wout, void* stream) {
    OFX_REQUIRE(in && out && scratch && B > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0, OFX_EINVAL);
    hipStream_t s = (hipStream_t)stream;
    OfxProfScope prof("resize_bicubic_u8", s);
    // Resample.c: horizontal pass to 8 bits, then the vertical pass
    hipLaunchKernelGGL(resample_rows_kernel, dim3(grid1d((long)B * Hin * Wout)), dim3(256), 0, s, in, scratch, (long)B * Hin, Win, Wout);
    hipLaunchKernelGGL(resample_cols_kernel, dim3(grid1d((long)B * Hout * Wout)), dim3(256), 0, s, scratch, out, B, Hin, Hout, Wout);
    return ofx_launch_status();
}

int ofx_sd_handoff(const uint8_t* image_bgr, const uint8_t* reference_bgr, const uint8_t* image_mask, const uint8_t* mask_latent,
                   float* image, float* cond_image, float* cond_mask, float* latmask, float* cond_mask_latent, int B, int H, int W,
                   int h, int w, void* stream) {
    OFX_REQUIRE(image_bgr && reference_bgr && image_mask && mask_latent && image && cond_image && cond_mask && latmask &&
                    cond_mask_latent,
                OFX_EINVAL);
    OFX_REQUIRE(B > 0 && H > 0 && W > 0 && h > 0 && w > 0, OFX_EINVAL);
    hipStream_t s = (hipStream_t)stream;
    OfxProfScope prof("sd_handoff", s);
    const long total = (long)B * H * W;
    hipLaunchKernelGGL(handoff_pixels_kernel, dim3(grid1d(total)), dim3(256), 0, s, image_bgr, reference_bgr, image_mask, image, cond_image,
                       cond_mask, (long)H * W, total);
    const long tl = (long)B * h * w;
    hipLaunchKernelGGL(handoff_latent_kernel, dim3(grid1d(tl)), dim3(256), 0, s, image_mask, mask_latent, latmask, cond_mask_latent, H, W, h,
                       w, tl);
    return ofx_launch_status();
}

}  // extern "C"
