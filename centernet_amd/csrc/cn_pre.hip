// cn_pre.hip -- BaseDetector.pre_process on the device and on the host.
//
// Replaces the host sequence of src/lib/detectors/base_detector.py:37-65:
//   resized = cv2.resize(image, (new_w, new_h))                         [scale != 1 only]
//   inp     = cv2.warpAffine(resized, trans_input, (inp_w, inp_h), flags=cv2.INTER_LINEAR)
//   inp     = ((inp / 255. - mean) / std).astype(np.float32)            (float64 arithmetic)
//   images  = inp.transpose(2, 0, 1)[None]; optional flip concat (:59-60)
// so that a uint8 frame (0.79 MB at 512x512) crosses PCIe instead of the fp32 tensor (3.1 MB)
// and the warp does not run on a host core.
//
// This is byte work and the bar is bit-exactness with what the reference computes, i.e. with
// OpenCV's uint8 INTER_LINEAR path -- a FIXED-POINT algorithm, not float bilinear (published in
// modules/imgproc/src/imgwarp.cpp and resize.cpp; restated with its constants in
// oracle/pre_oracle.py, which is the definition these kernels are tested against bit for bit):
//   warpAffine: source position of a destination pixel in 1/32 pixel,
//       X = (rn((m1*y + m2) * 1024) + 16 + rn((m0*x) * 1024)) >> 5      (same for Y with m3..m5;
//       m = the inverted matrix, double arithmetic, rn = round-half-even to int)
//     tap (X >> 5, Y >> 5), fractions fx = X & 31, fy = Y & 31, int16 weights
//       [(32-fy)(32-fx), (32-fy)fx, fy(32-fx), fy*fx] * 32   (sum 2^15; fraction (0,0): [32767,0,0,1]),
//     taps outside the image = 0, dst = clamp((sum + 2^14) >> 15).
//   resize (INTER_LINEAR): same size = copy; exactly half size = (a+b+c+d+2) >> 2; otherwise
//     separable with 11-bit coefficients rn((1-f)*2048), rn(f*2048), f from
//     float((d + 0.5)*scale - 0.5), exact horizontal pass and the uint8 vertical pass
//     (((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2.
// Normalisation: ((v/255.) - mean)/std in float64, rounded once to float32 (numpy's arithmetic
// for a uint8 array and float32 mean / std arrays).  No FMA contraction anywhere.
#include "cn_common.h"
#include <cstdlib>

// hipcc defaults to -ffp-contract=fast-honor-pragmas, and HIP's __dmul_rn/__dadd_rn are inline
// header functions compiled under that default (their results still fuse after inlining), so
// the arithmetic below is written with plain operators under an explicit contract(off).
#pragma clang fp contract(off)

namespace {

constexpr int AB_BITS = 10, INTER_BITS = 5, INTER_TAB = 1 << INTER_BITS, COEF_BITS = 15;

struct WarpArgs {
    const uint8_t *img;  // (H, W, 3) uint8, row pitch in bytes
    int H, W, pitch;
    double m[6];         // dst -> src (already inverted the way cv::warpAffine inverts it)
    int oh, ow;
    double mean[3], stdv[3];
    float *out;          // (1|2, 3, oh, ow) per image
    int flip;
    size_t img_stride, out_stride;   // bytes / floats from one image of a batch to the next (blockIdx.z)
};

// four int16-range weights of a 1/32-pixel fraction pair (initInterTab2D, INTER_LINEAR, fixed point)
__host__ __device__ inline void frac_weights(int fx, int fy, int w[4])
{
    if ((fx | fy) == 0) {      // 1.0 * 32768 saturates to 32767; the table's fix-up puts the 1 on tap 3
        w[0] = 32767; w[1] = 0; w[2] = 0; w[3] = 1;
        return;
    }
    w[0] = (INTER_TAB - fy) * (INTER_TAB - fx) * 32;
    w[1] = (INTER_TAB - fy) * fx * 32;
    w[2] = fy * (INTER_TAB - fx) * 32;
    w[3] = fy * fx * 32;
}

__host__ __device__ inline int clamp_i(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// one destination pixel of cv::warpAffine (uint8, C channels interleaved): v[c] = 0..255
template <int C>
__host__ __device__ inline void warp_pixel(const uint8_t *img, int H, int W, size_t pitch, const double *m,
                                           int x, int y, int rx0, int ry0, int v[C])
{
    // rx0 / ry0 = rn((m1*y + m2)*1024) + 16, rn((m4*y + m5)*1024) + 16: per row, from the caller
#ifdef __HIP_DEVICE_COMPILE__
    const int ad = __double2int_rn((m[0] * (double)x) * 1024.0);
    const int bd = __double2int_rn((m[3] * (double)x) * 1024.0);
#else
    const int ad = (int)__builtin_nearbyint((m[0] * (double)x) * 1024.0);
    const int bd = (int)__builtin_nearbyint((m[3] * (double)x) * 1024.0);
#endif
    const int X = (rx0 + ad) >> (AB_BITS - INTER_BITS), Y = (ry0 + bd) >> (AB_BITS - INTER_BITS);
    const int sx = clamp_i(X >> INTER_BITS, -32768, 32767), sy = clamp_i(Y >> INTER_BITS, -32768, 32767);
    int w[4];
    frac_weights(X & (INTER_TAB - 1), Y & (INTER_TAB - 1), w);
    const bool x0 = sx >= 0 && sx < W, x1 = sx + 1 >= 0 && sx + 1 < W;
    const bool y0 = sy >= 0 && sy < H, y1 = sy + 1 >= 0 && sy + 1 < H;
    const uint8_t *r0 = img + (size_t)clamp_i(sy, 0, H - 1) * pitch;
    const uint8_t *r1 = img + (size_t)clamp_i(sy + 1, 0, H - 1) * pitch;
    const int c0 = clamp_i(sx, 0, W - 1) * C, c1 = clamp_i(sx + 1, 0, W - 1) * C;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int t00 = (x0 && y0) ? r0[c0 + c] : 0, t01 = (x1 && y0) ? r0[c1 + c] : 0;
        const int t10 = (x0 && y1) ? r1[c0 + c] : 0, t11 = (x1 && y1) ? r1[c1 + c] : 0;
        const int s = t00 * w[0] + t01 * w[1] + t10 * w[2] + t11 * w[3];
        v[c] = clamp_i((s + (1 << (COEF_BITS - 1))) >> COEF_BITS, 0, 255);
    }
}

__host__ __device__ inline int row_base(const double *m, int a, int b, int y)
{
#ifdef __HIP_DEVICE_COMPILE__
    return __double2int_rn((m[a] * (double)y + m[b]) * 1024.0) + (1 << AB_BITS) / INTER_TAB / 2;
#else
    return (int)__builtin_nearbyint((m[a] * (double)y + m[b]) * 1024.0) + (1 << AB_BITS) / INTER_TAB / 2;
#endif
}

__global__ void warp_normalize_kernel(const WarpArgs a)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= a.ow) return;
    int v[3];
    const uint8_t *img = a.img + (size_t)blockIdx.z * a.img_stride;
    float *out = a.out + (size_t)blockIdx.z * a.out_stride;
    warp_pixel<3>(img, a.H, a.W, (size_t)a.pitch, a.m, x, y, row_base(a.m, 1, 2, y), row_base(a.m, 4, 5, y), v);
    const size_t plane = (size_t)a.oh * a.ow;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double n = ((double)v[c] / 255.0 - a.mean[c]) / a.stdv[c];
        const float f = (float)n;
        out[c * plane + (size_t)y * a.ow + x] = f;
        if (a.flip) out[(3 + c) * plane + (size_t)y * a.ow + (a.ow - 1 - x)] = f;
    }
}

// ---- cv::resize INTER_LINEAR, uint8 ----------------------------------------------------------
struct ResizeArgs {
    const uint8_t *img;
    int H, W, pitch, oh, ow;
    double scale_x, scale_y;   // 1. / (out / in), as resize.cpp forms them
    int mode;                  // 0 = linear, 1 = exactly half size (2x2 mean)
    uint8_t *out;              // (oh, ow, 3) dense
};

// left tap and the two 11-bit coefficients of destination index d (resize.cpp xofs/ialpha, yofs/ibeta)
__host__ __device__ inline void axis_coef(int d, double scale, int n_in, bool clamp_fraction, int *s, int *c0, int *c1)
{
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int i = (int)__builtin_floorf(f);
    f = f - (float)i;
    if (clamp_fraction) {
        if (i < 0) { f = 0.f; i = 0; }
        if (i >= n_in - 1) { f = 0.f; i = n_in - 1; }
    }
    const float g = 1.f - f;
#ifdef __HIP_DEVICE_COMPILE__
    *c0 = __float2int_rn(g * 2048.f);
    *c1 = __float2int_rn(f * 2048.f);
#else
    *c0 = (int)__builtin_nearbyintf(g * 2048.f);
    *c1 = (int)__builtin_nearbyintf(f * 2048.f);
#endif
    *s = i;
}

template <int C>
__host__ __device__ inline void resize_pixel(const uint8_t *img, int H, int W, size_t pitch, double scx,
                                             double scy, int mode, int x, int y, int v[C])
{
    if (mode == 1) {
        const uint8_t *r0 = img + (size_t)(2 * y) * pitch + (size_t)(2 * x) * C, *r1 = r0 + pitch;
#pragma unroll
        for (int c = 0; c < C; ++c) v[c] = (r0[c] + r0[C + c] + r1[c] + r1[C + c] + 2) >> 2;
        return;
    }
    int sx, a0, a1, sy, b0, b1;
    axis_coef(x, scx, W, true, &sx, &a0, &a1);
    axis_coef(y, scy, H, false, &sy, &b0, &b1);
    const uint8_t *r0 = img + (size_t)clamp_i(sy, 0, H - 1) * pitch;
    const uint8_t *r1 = img + (size_t)clamp_i(sy + 1, 0, H - 1) * pitch;
    const int c0 = sx * C, c1 = clamp_i(sx + 1, 0, W - 1) * C;   // a1 = 0 wherever sx + 1 is outside
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int S0 = r0[c0 + c] * a0 + r0[c1 + c] * a1;
        const int S1 = r1[c0 + c] * a0 + r1[c1 + c] * a1;
        v[c] = clamp_i((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2, 0, 255);
    }
}

__global__ void resize_u8_kernel(const ResizeArgs a)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= a.ow) return;
    int v[3];
    resize_pixel<3>(a.img, a.H, a.W, (size_t)a.pitch, a.scale_x, a.scale_y, a.mode, x, y, v);
#pragma unroll
    for (int c = 0; c < 3; ++c) a.out[((size_t)y * a.ow + x) * 3 + c] = (uint8_t)v[c];
}

}  // namespace

extern "C" int cn_warp_normalize_u8_f32(const uint8_t *image_hwc, int H, int W, int pitch_bytes,
                                        const double *dst_to_src_2x3, int out_h, int out_w,
                                        const float *mean3, const float *std3, int flip_concat,
                                        float *out_nchw, void *stream)
{
    return cn_warp_normalize_u8_f32_batch(image_hwc, 1, 0, H, W, pitch_bytes, dst_to_src_2x3, out_h, out_w,
                                          mean3, std3, flip_concat, out_nchw, stream);
}

extern "C" int cn_warp_normalize_u8_f32_batch(const uint8_t *images_hwc, int N, size_t image_stride_bytes,
                                              int H, int W, int pitch_bytes, const double *dst_to_src_2x3,
                                              int out_h, int out_w, const float *mean3, const float *std3,
                                              int flip_concat, float *out_nchw, void *stream)
{
    const uint8_t *image_hwc = images_hwc;
    if (!image_hwc || !dst_to_src_2x3 || !mean3 || !std3 || !out_nchw) return CN_ERR_NULL;
    if (H <= 0 || W <= 0 || out_h <= 0 || out_w <= 0 || pitch_bytes < 3 * W || out_h > 65535 ||
        H > 32767 || W > 32767 || N <= 0 || N > 65535)
        return CN_ERR_SHAPE;
    if (N > 1 && image_stride_bytes < (size_t)pitch_bytes * (size_t)H) return CN_ERR_SHAPE;
    WarpArgs a = {};
    a.img = image_hwc; a.H = H; a.W = W; a.pitch = pitch_bytes; a.oh = out_h; a.ow = out_w;
    for (int i = 0; i < 6; ++i) a.m[i] = dst_to_src_2x3[i];
    for (int c = 0; c < 3; ++c) {
        if (std3[c] == 0.f) return CN_ERR_SHAPE;
        a.mean[c] = (double)mean3[c];
        a.stdv[c] = (double)std3[c];
    }
    a.out = out_nchw; a.flip = flip_concat ? 1 : 0;
    a.img_stride = image_stride_bytes;
    a.out_stride = (size_t)(a.flip ? 6 : 3) * out_h * out_w;
    dim3 grid(cn_cdiv(out_w, 128), out_h, N);
    hipLaunchKernelGGL(warp_normalize_kernel, grid, dim3(128), 0, (hipStream_t)stream, a);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

extern "C" int cn_resize_bilinear_u8(const uint8_t *image_hwc, int H, int W, int pitch_bytes,
                                     int out_h, int out_w, uint8_t *out_hwc, void *stream)
{
    if (!image_hwc || !out_hwc) return CN_ERR_NULL;
    if (H <= 0 || W <= 0 || out_h <= 0 || out_w <= 0 || pitch_bytes < 3 * W || out_h > 65535) return CN_ERR_SHAPE;
    if (H == out_h && W == out_w) {      // cv::resize: same size = copy
        return hipMemcpy2DAsync(out_hwc, (size_t)3 * W, image_hwc, (size_t)pitch_bytes, (size_t)3 * W, H,
                                hipMemcpyDeviceToDevice, (hipStream_t)stream) == hipSuccess ? CN_OK : CN_ERR_LAUNCH;
    }
    ResizeArgs a = {};
    a.img = image_hwc; a.H = H; a.W = W; a.pitch = pitch_bytes; a.oh = out_h; a.ow = out_w;
    a.scale_x = 1.0 / ((double)out_w / (double)W);
    a.scale_y = 1.0 / ((double)out_h / (double)H);
    a.mode = (H == 2 * out_h && W == 2 * out_w) ? 1 : 0;
    a.out = out_hwc;
    dim3 grid(cn_cdiv(out_w, 128), out_h);
    hipLaunchKernelGGL(resize_u8_kernel, grid, dim3(128), 0, (hipStream_t)stream, a);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

// ---- the same two operations on the HOST (callers that keep BaseDetector.pre_process on host
// cores, e.g. DataLoader workers: base_detector.py:37-65): the integer algorithms above, pixel by
// pixel in C; channels <= 4.
template <int C>
static void warp_host(const uint8_t *img, int h_in, int w_in, const double *m, int h_out, int w_out, uint8_t *out)
{
    // the per-column terms rn(m0*x*1024), rn(m3*x*1024) once per call (cv::warpAffine's adelta /
    // bdelta tables), so that the pixel loop is integer-only; interior pixels (all four taps on
    // the image) skip the border tests.  Same integers as warp_pixel<C>.
    int *adelta = (int *)malloc((size_t)2 * w_out * sizeof(int));
    if (!adelta) {   // out of memory: the plain per-pixel form
        for (int y = 0; y < h_out; ++y) {
            const int rx0 = row_base(m, 1, 2, y), ry0 = row_base(m, 4, 5, y);
            for (int x = 0; x < w_out; ++x) {
                int v[C];
                warp_pixel<C>(img, h_in, w_in, (size_t)w_in * C, m, x, y, rx0, ry0, v);
                for (int c = 0; c < C; ++c) out[((size_t)y * w_out + x) * C + c] = (uint8_t)v[c];
            }
        }
        return;
    }
    int *bdelta = adelta + w_out;
    for (int x = 0; x < w_out; ++x) {
        adelta[x] = (int)__builtin_nearbyint((m[0] * (double)x) * 1024.0);
        bdelta[x] = (int)__builtin_nearbyint((m[3] * (double)x) * 1024.0);
    }
    const size_t pitch = (size_t)w_in * C;
    for (int y = 0; y < h_out; ++y) {
        const int rx0 = row_base(m, 1, 2, y), ry0 = row_base(m, 4, 5, y);
        uint8_t *orow = out + (size_t)y * w_out * C;
        for (int x = 0; x < w_out; ++x) {
            const int X = (rx0 + adelta[x]) >> (AB_BITS - INTER_BITS), Y = (ry0 + bdelta[x]) >> (AB_BITS - INTER_BITS);
            const int sx = clamp_i(X >> INTER_BITS, -32768, 32767), sy = clamp_i(Y >> INTER_BITS, -32768, 32767);
            int w[4];
            frac_weights(X & (INTER_TAB - 1), Y & (INTER_TAB - 1), w);
            if (sx >= 0 && sx + 1 < w_in && sy >= 0 && sy + 1 < h_in) {
                const uint8_t *p0 = img + (size_t)sy * pitch + (size_t)sx * C, *p1 = p0 + pitch;
                for (int c = 0; c < C; ++c) {
                    const int sum = p0[c] * w[0] + p0[C + c] * w[1] + p1[c] * w[2] + p1[C + c] * w[3];
                    orow[x * C + c] = (uint8_t)clamp_i((sum + (1 << (COEF_BITS - 1))) >> COEF_BITS, 0, 255);
                }
                continue;
            }
            const bool x0 = sx >= 0 && sx < w_in, x1 = sx + 1 >= 0 && sx + 1 < w_in;
            const bool y0 = sy >= 0 && sy < h_in, y1 = sy + 1 >= 0 && sy + 1 < h_in;
            const uint8_t *r0 = img + (size_t)clamp_i(sy, 0, h_in - 1) * pitch;
            const uint8_t *r1 = img + (size_t)clamp_i(sy + 1, 0, h_in - 1) * pitch;
            const int c0 = clamp_i(sx, 0, w_in - 1) * C, c1 = clamp_i(sx + 1, 0, w_in - 1) * C;
            for (int c = 0; c < C; ++c) {
                const int t00 = (x0 && y0) ? r0[c0 + c] : 0, t01 = (x1 && y0) ? r0[c1 + c] : 0;
                const int t10 = (x0 && y1) ? r1[c0 + c] : 0, t11 = (x1 && y1) ? r1[c1 + c] : 0;
                const int sum = t00 * w[0] + t01 * w[1] + t10 * w[2] + t11 * w[3];
                orow[x * C + c] = (uint8_t)clamp_i((sum + (1 << (COEF_BITS - 1))) >> COEF_BITS, 0, 255);
            }
        }
    }
    free(adelta);
}

extern "C" int cn_warp_affine_u8_host(const uint8_t *img, int h_in, int w_in, int channels,
                                      const double *dst_to_src_2x3, int h_out, int w_out, uint8_t *out)
{
    if (!img || !dst_to_src_2x3 || !out) return CN_ERR_NULL;
    if (h_in <= 0 || w_in <= 0 || h_out <= 0 || w_out <= 0 || channels <= 0 || channels > 4 ||
        h_in > 32767 || w_in > 32767)
        return CN_ERR_SHAPE;
    switch (channels) {
    case 1: warp_host<1>(img, h_in, w_in, dst_to_src_2x3, h_out, w_out, out); break;
    case 2: warp_host<2>(img, h_in, w_in, dst_to_src_2x3, h_out, w_out, out); break;
    case 3: warp_host<3>(img, h_in, w_in, dst_to_src_2x3, h_out, w_out, out); break;
    default: warp_host<4>(img, h_in, w_in, dst_to_src_2x3, h_out, w_out, out); break;
    }
    return CN_OK;
}

template <int C>
static void resize_host(const uint8_t *img, int h_in, int w_in, int h_out, int w_out, uint8_t *out)
{
    const double scx = 1.0 / ((double)w_out / (double)w_in), scy = 1.0 / ((double)h_out / (double)h_in);
    const int mode = (h_in == 2 * h_out && w_in == 2 * w_out) ? 1 : 0;
    for (int y = 0; y < h_out; ++y)
        for (int x = 0; x < w_out; ++x) {
            int v[C];
            resize_pixel<C>(img, h_in, w_in, (size_t)w_in * C, scx, scy, mode, x, y, v);
            for (int c = 0; c < C; ++c) out[((size_t)y * w_out + x) * C + c] = (uint8_t)v[c];
        }
}

extern "C" int cn_resize_linear_u8_host(const uint8_t *img, int h_in, int w_in, int channels, int h_out,
                                        int w_out, uint8_t *out)
{
    if (!img || !out) return CN_ERR_NULL;
    if (h_in <= 0 || w_in <= 0 || h_out <= 0 || w_out <= 0 || channels <= 0 || channels > 4) return CN_ERR_SHAPE;
    if (h_in == h_out && w_in == w_out) {
        __builtin_memcpy(out, img, (size_t)h_in * w_in * channels);
        return CN_OK;
    }
    switch (channels) {
    case 1: resize_host<1>(img, h_in, w_in, h_out, w_out, out); break;
    case 2: resize_host<2>(img, h_in, w_in, h_out, w_out, out); break;
    case 3: resize_host<3>(img, h_in, w_in, h_out, w_out, out); break;
    default: resize_host<4>(img, h_in, w_in, h_out, w_out, out); break;
    }
    return CN_OK;
}


// ---------------------------------------------------------------------------
// ctdet_post_process + the per-class split on the device (utils/post_process.py:83-100,
// utils/image.py:19-24,63-66, detectors/ctdet.py:47-56), so that the host tail of a batch is one small
// copy and 80 slices per image: for every image the K raw detections [x1, y1, x2, y2, score, class]
// in output-grid units become rows [x1, y1, x2, y2, score] in source-frame pixels, grouped by class
// (ascending; inside a class in their original, score-descending order: a stable sort), plus the
// class boundaries.  Arithmetic as the reference's: float32 point -> float64 (t0*x + t1*y) + t2 ->
// float32, then / scale in float32; rows whose class lies outside [0, num_classes) are dropped
// (they match no `classes == j`).
// ---------------------------------------------------------------------------
namespace {
constexpr int PP_KMAX = 128;
__global__ __launch_bounds__(PP_KMAX) void ctdet_post_kernel(const float *__restrict__ dets, int K, int num_classes,
                                                             const double *__restrict__ to_source, int per_image,
                                                             float scale, float *__restrict__ rows,
                                                             int32_t *__restrict__ bounds)
{
    __shared__ int cls_s[PP_KMAX];
    const int b = blockIdx.x, k = threadIdx.x;
    const double *t = to_source + (per_image ? (size_t)b * 6 : 0);
    float r[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    int cls = num_classes;             // sorts behind every class
    if (k < K) {
        const float *d = dets + ((size_t)b * K + k) * 6;
        const int c = (int)(long long)d[5];        // astype(np.int64): truncation
        if (c >= 0 && c < num_classes) cls = c;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const double x = (double)d[2 * p], y = (double)d[2 * p + 1];
            const double sx = (x * t[0] + y * t[1]) + t[2];
            const double sy = (x * t[3] + y * t[4]) + t[5];
            r[2 * p] = (float)sx / scale;
            r[2 * p + 1] = (float)sy / scale;
        }
        r[4] = d[4];
    }
    cls_s[k] = cls;
    __syncthreads();
    if (k < K) {
        int rank = 0;
        for (int j = 0; j < K; ++j) rank += (cls_s[j] < cls || (cls_s[j] == cls && j < k)) ? 1 : 0;
        float *o = rows + ((size_t)b * K + rank) * 5;
#pragma unroll
        for (int e = 0; e < 5; ++e) o[e] = r[e];
    }
    for (int c = k; c <= num_classes; c += PP_KMAX) {
        int n = 0;
        for (int j = 0; j < K; ++j) n += cls_s[j] < c ? 1 : 0;
        bounds[(size_t)b * (num_classes + 1) + c] = n;
    }
}
}  // namespace

extern "C" int cn_ctdet_post_process_f32(const float *dets, int B, int K, int num_classes,
                                         const double *to_source_2x3, int per_image, float scale,
                                         float *rows, int32_t *bounds, void *stream)
{
    if (!dets || !to_source_2x3 || !rows || !bounds) return CN_ERR_NULL;
    if (B <= 0 || K <= 0 || num_classes <= 0 || !(scale > 0.f)) return CN_ERR_SHAPE;
    if (K > PP_KMAX) return CN_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(ctdet_post_kernel, dim3(B), dim3(PP_KMAX), 0, (hipStream_t)stream, dets, K, num_classes,
                       to_source_2x3, per_image ? 1 : 0, scale, rows, bounds);
    CN_CHECK_LAUNCH();
    return CN_OK;
}
