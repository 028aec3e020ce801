// cn_pre.hip -- BaseDetector.pre_process on the device.
//
// Replaces the host sequence of src/lib/detectors/base_detector.py:37-65:
//   resized = cv2.resize(image, (new_w, new_h))                         [scale != 1 only]
//   inp     = cv2.warpAffine(resized, trans_input, (inp_w, inp_h), flags=cv2.INTER_LINEAR)
//   inp     = ((inp / 255. - mean) / std).astype(np.float32)            (float64 arithmetic)
//   images  = inp.transpose(2, 0, 1)[None]; optional flip concat (:59-60)
// so that a uint8 frame (0.79 MB at 512x512) crosses PCIe instead of the fp32 tensor (3.1 MB)
// and the bilinear warp does not run on a host core.
//
// Arithmetic contract (identical, operation for operation, in centernet_amd/image.py and in
// oracle/pre_oracle.py -- results are bit-identical): source coordinate
//   sx = (m0*x + m1*y) + m2,  sy = (m3*x + m4*y) + m5          (float64, dst -> src matrix)
// floor/fraction split, four taps (zero outside the image, or clamped when replicate != 0),
//   v = ((t00*(1-fx))*(1-fy) + (t01*fx)*(1-fy) + (t10*(1-fx))*fy + (t11*fx)*fy), left to right,
// rounded half-to-even and clamped to uint8 -- then ((v/255.) - mean)/std in float64, rounded once
// to float32.  No FMA contraction (#pragma clang fp contract(off)).  OpenCV's fixed-point bilinear
// (1/32-pixel coordinates, 15-bit weights) can differ from this by one uint8 level on
// non-identity warps; OpenCV is not available here, so that delta is unpinned.
#include "cn_common.h"

// hipcc defaults to -ffp-contract=fast-honor-pragmas, and HIP's __dmul_rn/__dadd_rn are inline
// header functions compiled under that default (their results still fuse after inlining), so
// the arithmetic below is written with plain operators under an explicit contract(off).
#pragma clang fp contract(off)

namespace {

struct WarpArgs {
    const uint8_t *img;  // (H, W, 3) uint8, row pitch in bytes
    int H, W, pitch;
    double m[6];         // dst -> src
    int oh, ow;
    int replicate;
    // normalise form
    double mean[3], stdv[3];
    float *out;          // (1|2, 3, oh, ow)
    int flip;
    // resize form
    uint8_t *out_u8;     // (oh, ow, 3), dense
};

__device__ __forceinline__ void bilinear3(const WarpArgs &a, int x, int y, double v[3])
{
    const double sx = (a.m[0] * (double)x + a.m[1] * (double)y) + a.m[2];
    const double sy = (a.m[3] * (double)x + a.m[4] * (double)y) + a.m[5];
    const double fx0 = floor(sx), fy0 = floor(sy);
    // coordinates far outside the image contribute nothing (and must not overflow int)
    const bool far = !(fx0 > -4.0 && fx0 < (double)a.W + 4.0 && fy0 > -4.0 && fy0 < (double)a.H + 4.0);
    const int x0 = far ? -4 : (int)fx0, y0 = far ? -4 : (int)fy0;
    const double fx = far ? 0.0 : sx - fx0, fy = far ? 0.0 : sy - fy0;
    const double gx = 1.0 - fx, gy = 1.0 - fy;
    int xs[2] = {x0, x0 + 1}, ys[2] = {y0, y0 + 1};
    bool okx[2], oky[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        okx[i] = xs[i] >= 0 && xs[i] < a.W;
        oky[i] = ys[i] >= 0 && ys[i] < a.H;
        xs[i] = min(max(xs[i], 0), a.W - 1);
        ys[i] = min(max(ys[i], 0), a.H - 1);
        if (a.replicate) okx[i] = oky[i] = !far;
    }
    const uint8_t *r0 = a.img + (size_t)ys[0] * a.pitch, *r1 = a.img + (size_t)ys[1] * a.pitch;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double t00 = (okx[0] && oky[0]) ? (double)r0[xs[0] * 3 + c] : 0.0;
        const double t01 = (okx[1] && oky[0]) ? (double)r0[xs[1] * 3 + c] : 0.0;
        const double t10 = (okx[0] && oky[1]) ? (double)r1[xs[0] * 3 + c] : 0.0;
        const double t11 = (okx[1] && oky[1]) ? (double)r1[xs[1] * 3 + c] : 0.0;
        double s = (t00 * gx) * gy;
        s = s + (t01 * fx) * gy;
        s = s + (t10 * gx) * fy;
        s = s + (t11 * fx) * fy;
        s = rint(s);  // half to even, as numpy.rint
        v[c] = fmin(fmax(s, 0.0), 255.0);
    }
}

__global__ void warp_normalize_kernel(const WarpArgs a)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= a.ow) return;
    double v[3];
    bilinear3(a, x, y, v);
    const size_t plane = (size_t)a.oh * a.ow;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double n = (v[c] / 255.0 - a.mean[c]) / a.stdv[c];
        const float f = (float)n;
        a.out[c * plane + (size_t)y * a.ow + x] = f;
        if (a.flip) a.out[(3 + c) * plane + (size_t)y * a.ow + (a.ow - 1 - x)] = f;
    }
}

__global__ void warp_u8_kernel(const WarpArgs a)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= a.ow) return;
    double v[3];
    bilinear3(a, x, y, v);
#pragma unroll
    for (int c = 0; c < 3; ++c) a.out_u8[((size_t)y * a.ow + x) * 3 + c] = (uint8_t)v[c];
}

int fill_common(WarpArgs &a, const uint8_t *img, int H, int W, int pitch, const double *m, int oh,
                int ow, int replicate)
{
    if (!img || !m) return CN_ERR_NULL;
    if (H <= 0 || W <= 0 || oh <= 0 || ow <= 0 || pitch < 3 * W || oh > 65535) return CN_ERR_SHAPE;
    a.img = img; a.H = H; a.W = W; a.pitch = pitch; a.oh = oh; a.ow = ow; a.replicate = replicate;
    for (int i = 0; i < 6; ++i) a.m[i] = m[i];
    return CN_OK;
}

}  // namespace

extern "C" int cn_warp_normalize_u8_f32(const uint8_t *image_hwc, int H, int W, int pitch_bytes,
                                        const double *dst_to_src_2x3, int out_h, int out_w,
                                        const float *mean3, const float *std3, int flip_concat,
                                        float *out_nchw, void *stream)
{
    WarpArgs a = {};
    const int rc = fill_common(a, image_hwc, H, W, pitch_bytes, dst_to_src_2x3, out_h, out_w, 0);
    if (rc != CN_OK) return rc;
    if (!mean3 || !std3 || !out_nchw) return CN_ERR_NULL;
    for (int c = 0; c < 3; ++c) {
        if (std3[c] == 0.f) return CN_ERR_SHAPE;
        a.mean[c] = (double)mean3[c];
        a.stdv[c] = (double)std3[c];
    }
    a.out = out_nchw; a.flip = flip_concat ? 1 : 0;
    dim3 grid(cn_cdiv(out_w, 128), out_h);
    hipLaunchKernelGGL(warp_normalize_kernel, grid, dim3(128), 0, (hipStream_t)stream, a);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

extern "C" int cn_resize_bilinear_u8(const uint8_t *image_hwc, int H, int W, int pitch_bytes,
                                     int out_h, int out_w, uint8_t *out_hwc, void *stream)
{
    // cv2.resize(INTER_LINEAR): src = (dst + 0.5) * (in/out) - 0.5, border replicated
    const double sx = (double)W / (double)out_w, sy = (double)H / (double)out_h;
    const double m[6] = {sx, 0.0, 0.5 * sx - 0.5, 0.0, sy, 0.5 * sy - 0.5};
    WarpArgs a = {};
    const int rc = fill_common(a, image_hwc, H, W, pitch_bytes, m, out_h, out_w, 1);
    if (rc != CN_OK) return rc;
    if (!out_hwc) return CN_ERR_NULL;
    a.out_u8 = out_hwc;
    dim3 grid(cn_cdiv(out_w, 128), out_h);
    hipLaunchKernelGGL(warp_u8_kernel, grid, dim3(128), 0, (hipStream_t)stream, a);
    CN_CHECK_LAUNCH();
    return CN_OK;
}
