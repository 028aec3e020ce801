// cn_misc.hip -- memory-bound helpers around the hot path: layout conversion at
// the API edge, max pooling, and the NCHW (reference-layout) entry point of DCNv2.
#include "cn_common.h"

namespace {

// (B,C,HW) -> (B,HW,pitch): 32x32 tiles through LDS, both sides coalesced.
__global__ void nchw_to_nhwc_kernel(const float *__restrict__ x, float *__restrict__ y, int C,
                                    int HW, int pitch)
{
    __shared__ float t[32][33];
    const int b = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const float *xb = x + (size_t)b * C * HW;
    float *yb = y + (size_t)b * HW * pitch;
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, p = p0 + tx;
        t[i][tx] = (c < C && p < HW) ? xb[(size_t)c * HW + p] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int p = p0 + i, c = c0 + tx;
        if (p < HW && c < pitch) yb[(size_t)p * pitch + c] = t[tx][i];  // c >= C -> zeros
    }
}

__global__ void nhwc_to_nchw_kernel(const float *__restrict__ x, float *__restrict__ y, int C,
                                    int HW, int pitch)
{
    __shared__ float t[32][33];
    const int b = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float *xb = x + (size_t)b * HW * pitch;
    float *yb = y + (size_t)b * C * HW;
    for (int i = ty; i < 32; i += 8) {
        const int p = p0 + i, c = c0 + tx;
        t[i][tx] = (p < HW && c < C) ? xb[(size_t)p * pitch + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, p = p0 + tx;
        if (c < C && p < HW) yb[(size_t)c * HW + p] = t[tx][i];
    }
}

// max pooling, NHWC, -inf padding (torch.nn.MaxPool2d semantics), 4 channels/thread
// IN_S: the input is an f32s tensor (fp16 high / low pairs, cn_common.h; C % 32 == 0); the output
// is plain fp32 either way
template <bool IN_S>
__global__ void maxpool_nhwc_kernel(const float *__restrict__ x, float *__restrict__ y, int B, int H,
                                    int W, int C, int Ho, int Wo, int k, int s, int pad, float out_mul)
{
    const int c4n = C >> 2;
    const size_t total = (size_t)B * Ho * Wo * c4n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        size_t r = i / c4n;
        const int ox = (int)(r % Wo);
        r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        const float ninf = -__builtin_huge_valf();
        cn_f32x4 m = {ninf, ninf, ninf, ninf};
        for (int dy = 0; dy < k; ++dy) {
            const int iy = oy * s - pad + dy;
            if (iy < 0 || iy >= H) continue;
            for (int dx = 0; dx < k; ++dx) {
                const int ix = ox * s - pad + dx;
                if (ix < 0 || ix >= W) continue;
                const size_t pix = ((size_t)b * H + iy) * W + ix;
                const cn_f32x4 v = IN_S ? cn_load4_f32s(x, pix, C, c4 * 4)
                                        : *reinterpret_cast<const cn_f32x4 *>(x + pix * C + c4 * 4);
                m.x = fmaxf(m.x, v.x);
                m.y = fmaxf(m.y, v.y);
                m.z = fmaxf(m.z, v.z);
                m.w = fmaxf(m.w, v.w);
            }
        }
        if (IN_S) m = m * out_mul;   // stored -> real units (a power of two; max commutes with it)
        *reinterpret_cast<cn_f32x4 *>(y + (((size_t)b * Ho + oy) * Wo + ox) * C + c4 * 4) = m;
    }
}

// f32s in (pixel pitch in_pitch), f32s out (pixel pitch out_pitch: the output may be a channel slice of
// a wider tensor, e.g. a concatenation buffer): y = split(max(window) * mul).  max commutes with the
// join of a (high, low) pair and with a positive factor, so this is the plain max-pool of the values.
__global__ void maxpool_nhwc_f32s_kernel(const void *__restrict__ x, void *__restrict__ y, int B, int H,
                                         int W, int C, int in_pitch, int out_pitch, int Ho, int Wo, int k,
                                         int s, int pad, float mul, uint32_t *range)
{
    const int c4n = C >> 2;
    const size_t total = (size_t)B * Ho * Wo * c4n;
    float rng = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        size_t r = i / c4n;
        const int ox = (int)(r % Wo);
        r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        const float ninf = -__builtin_huge_valf();
        cn_f32x4 m = {ninf, ninf, ninf, ninf};
        for (int dy = 0; dy < k; ++dy) {
            const int iy = oy * s - pad + dy;
            if (iy < 0 || iy >= H) continue;
            for (int dx = 0; dx < k; ++dx) {
                const int ix = ox * s - pad + dx;
                if (ix < 0 || ix >= W) continue;
                const cn_f32x4 v = cn_load4_f32s(x, ((size_t)b * H + iy) * W + ix, in_pitch, c4 * 4);
                m.x = fmaxf(m.x, v.x);
                m.y = fmaxf(m.y, v.y);
                m.z = fmaxf(m.z, v.z);
                m.w = fmaxf(m.w, v.w);
            }
        }
        m = m * mul;
        cn_rng_upd4(rng, m);
        cn_store4_f32s(y, ((size_t)b * Ho + oy) * Wo + ox, out_pitch, c4 * 4, m);
    }
    cn_rng_commit(range, 0, rng);
}

// offset (B,18,HW) + mask (B,9,HW) NCHW -> om (B,HW,32) NHWC
__global__ void pack_offset_mask_kernel(const float *__restrict__ offset,
                                        const float *__restrict__ mask, float *__restrict__ om,
                                        int HW, size_t total)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i & 31);
        const size_t bp = i >> 5;
        const size_t b = bp / HW, p = bp - b * HW;
        float v = 0.f;
        if (ch < 18)
            v = offset[(b * 18 + ch) * HW + p];
        else if (ch < 27)
            v = mask[(b * 9 + (ch - 18)) * HW + p];
        om[i] = v;
    }
}

// depthwise ConvTranspose2d(C, C, kernel 2f, stride f, padding f/2, groups=C), NHWC,
// optionally fused with the element-wise add that follows it in IDAUp.forward
// (pose_dla_dcn.py:370-373, 381-386).  Each output pixel receives 2x2 taps.
__global__ void dw_deconv_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                 const float *__restrict__ add, float *__restrict__ y, int B, int H,
                                 int W, int C, int f)
{
    const int c4n = C >> 2;
    const int OH = H * f, OW = W * f, K2 = 2 * f, pad = f / 2;
    const size_t total = (size_t)B * OH * OW * c4n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        size_t r = i / c4n;
        const int ox = (int)(r % OW);
        r /= OW;
        const int oy = (int)(r % OH);
        const int b = (int)(r / OH);
        const int ky0 = (oy + pad) % f, kx0 = (ox + pad) % f;
        const int iy0 = (oy + pad - ky0) / f, ix0 = (ox + pad - kx0) / f;
        cn_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ty = 0; ty < 2; ++ty) {
            const int iy = iy0 - ty, ky = ky0 + ty * f;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int tx = 0; tx < 2; ++tx) {
                const int ix = ix0 - tx, kx = kx0 + tx * f;
                if (ix < 0 || ix >= W) continue;
                const cn_f32x4 v = *reinterpret_cast<const cn_f32x4 *>(
                    x + (((size_t)b * H + iy) * W + ix) * C + c4 * 4);
                const cn_f32x4 wv =
                    *reinterpret_cast<const cn_f32x4 *>(w + (size_t)(ky * K2 + kx) * C + c4 * 4);
                acc += v * wv;
            }
        }
        const size_t o = (((size_t)b * OH + oy) * OW + ox) * C + c4 * 4;
        if (add) acc += *reinterpret_cast<const cn_f32x4 *>(add + o);
        *reinterpret_cast<cn_f32x4 *>(y + o) = acc;
    }
}

// copy C channels of npix pixels between two NHWC tensors with different pitches
// (torch.cat(x, 1) of Root.forward, pose_dla_dcn.py:159, without a permute)
__global__ void copy_channels_kernel(const float *__restrict__ src, int src_pitch,
                                     float *__restrict__ dst, int dst_pitch, size_t npix, int C)
{
    const int c4n = C >> 2;
    const size_t total = npix * c4n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        const size_t p = i / c4n;
        *reinterpret_cast<cn_f32x4 *>(dst + p * dst_pitch + c4 * 4) =
            *reinterpret_cast<const cn_f32x4 *>(src + p * src_pitch + c4 * 4);
    }
}

// nearest-neighbour x2 up-sampling (nn.Upsample(scale_factor=2), large_hourglass.py:102-103)
// fused with the add of the skip branch (merge layer, :104-109)
__global__ void upsample2x_add_kernel(const float *__restrict__ x, const float *__restrict__ add,
                                      float *__restrict__ y, int B, int H, int W, int C)
{
    const int c4n = C >> 2;
    const int OH = 2 * H, OW = 2 * W;
    const size_t total = (size_t)B * OH * OW * c4n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        size_t r = i / c4n;
        const int ox = (int)(r % OW);
        r /= OW;
        const int oy = (int)(r % OH);
        const int b = (int)(r / OH);
        cn_f32x4 v = *reinterpret_cast<const cn_f32x4 *>(
            x + (((size_t)b * H + (oy >> 1)) * W + (ox >> 1)) * C + c4 * 4);
        const size_t o = (((size_t)b * OH + oy) * OW + ox) * C + c4 * 4;
        if (add) v += *reinterpret_cast<const cn_f32x4 *>(add + o);
        *reinterpret_cast<cn_f32x4 *>(y + o) = v;
    }
}

inline int blocks_for(size_t total, int per_block, int cap)
{
    size_t b = (total + per_block - 1) / per_block;
    return (int)(b > (size_t)cap ? cap : (b ? b : 1));
}

}  // namespace

extern "C" int cn_version(void) { return 300; }  // 0.3.0: f32s range control (cn_f32s_ctl)

extern "C" const char *cn_arch(void) { return "gfx950"; }

extern "C" const char *cn_status_string(int s)
{
    switch (s) {
    case CN_OK: return "ok";
    case CN_ERR_SHAPE: return "shape mismatch";
    case CN_ERR_UNSUPPORTED: return "unsupported configuration";
    case CN_ERR_WORKSPACE: return "workspace too small";
    case CN_ERR_LAUNCH: return "kernel launch failed";
    case CN_ERR_NULL: return "null pointer";
    case CN_ERR_ALIGN: return "pointer not 16-byte aligned";
    default: return "unknown status";
    }
}

extern "C" int cn_nchw_to_nhwc_f32(const float *x, float *y, int B, int C, int H, int W,
                                   int out_pitch, void *stream)
{
    if (!x || !y) return CN_ERR_NULL;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || out_pitch < C) return CN_ERR_SHAPE;
    const int HW = H * W;
    dim3 grid(cn_cdiv(HW, 32), cn_cdiv(out_pitch, 32), B);
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, y, C, HW,
                       out_pitch);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

extern "C" int cn_nhwc_to_nchw_f32(const float *x, float *y, int B, int C, int H, int W,
                                   int in_pitch, void *stream)
{
    if (!x || !y) return CN_ERR_NULL;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || in_pitch < C) return CN_ERR_SHAPE;
    const int HW = H * W;
    dim3 grid(cn_cdiv(HW, 32), cn_cdiv(C, 32), B);
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, y, C, HW,
                       in_pitch);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

extern "C" int cn_maxpool_nhwc_f32(const float *x, float *y, int B, int H, int W, int C, int k,
                                   int s, int pad, void *stream)
{
    if (!x || !y) return CN_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || k <= 0 || s <= 0 || pad < 0) return CN_ERR_SHAPE;
    if (C & 3) return CN_ERR_UNSUPPORTED;
    if (!cn_aligned16(x) || !cn_aligned16(y)) return CN_ERR_ALIGN;
    const int Ho = (H + 2 * pad - k) / s + 1, Wo = (W + 2 * pad - k) / s + 1;
    if (Ho <= 0 || Wo <= 0) return CN_ERR_SHAPE;
    const size_t total = (size_t)B * Ho * Wo * (C >> 2);
    hipLaunchKernelGGL(maxpool_nhwc_kernel<false>, dim3(blocks_for(total, 256, 65535)), dim3(256), 0,
                       (hipStream_t)stream, x, y, B, H, W, C, Ho, Wo, k, s, pad, 1.f);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

extern "C" int cn_maxpool_nhwc(const void *x, float *y, int B, int H, int W, int C, int k, int s,
                               int pad, int in_dtype, void *stream)
{
    return cn_maxpool_nhwc_scaled(x, y, B, H, W, C, k, s, pad, in_dtype, 1.f, stream);
}

extern "C" int cn_maxpool_nhwc_scaled(const void *x, float *y, int B, int H, int W, int C, int k,
                                      int s, int pad, int in_dtype, float out_mul, void *stream)
{
    if (out_mul == 0.f) out_mul = 1.f;
    if (in_dtype == CN_DTYPE_F32)
        return cn_maxpool_nhwc_f32((const float *)x, y, B, H, W, C, k, s, pad, stream);
    if (in_dtype != CN_DTYPE_F32S) return CN_ERR_UNSUPPORTED;
    if (!x || !y) return CN_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || k <= 0 || s <= 0 || pad < 0) return CN_ERR_SHAPE;
    if (C & 31) return CN_ERR_UNSUPPORTED;   // f32s tensors are whole 32-channel groups
    if (!cn_aligned16(x) || !cn_aligned16(y)) return CN_ERR_ALIGN;
    const int Ho = (H + 2 * pad - k) / s + 1, Wo = (W + 2 * pad - k) / s + 1;
    if (Ho <= 0 || Wo <= 0) return CN_ERR_SHAPE;
    const size_t total = (size_t)B * Ho * Wo * (C >> 2);
    hipLaunchKernelGGL(maxpool_nhwc_kernel<true>, dim3(blocks_for(total, 256, 65535)), dim3(256), 0,
                       (hipStream_t)stream, (const float *)x, y, B, H, W, C, Ho, Wo, k, s, pad, out_mul);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

extern "C" int cn_maxpool_nhwc_f32s(const void *x, void *y, int B, int H, int W, int C, int in_pitch,
                                    int out_pitch, int k, int s, int pad, float mul, uint32_t *range,
                                    void *stream)
{
    if (mul == 0.f) mul = 1.f;
    if (!x || !y) return CN_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || k <= 0 || s <= 0 || pad < 0 || in_pitch < C || out_pitch < C || !(mul > 0.f))
        return CN_ERR_SHAPE;
    if ((C & 31) || (in_pitch & 31) || (out_pitch & 31)) return CN_ERR_UNSUPPORTED;   // whole 32-channel groups
    if ((((uintptr_t)x) & 127u) || (((uintptr_t)y) & 127u)) return CN_ERR_ALIGN;
    const int Ho = (H + 2 * pad - k) / s + 1, Wo = (W + 2 * pad - k) / s + 1;
    if (Ho <= 0 || Wo <= 0) return CN_ERR_SHAPE;
    const size_t total = (size_t)B * Ho * Wo * (C >> 2);
    hipLaunchKernelGGL(maxpool_nhwc_f32s_kernel, dim3(blocks_for(total, 256, 65535)), dim3(256), 0,
                       (hipStream_t)stream, x, y, B, H, W, C, in_pitch, out_pitch, Ho, Wo, k, s, pad, mul, range);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

extern "C" int cn_maxpool3x3s2_nhwc_f32(const float *x, float *y, int B, int H, int W, int C,
                                        void *stream)
{
    return cn_maxpool_nhwc_f32(x, y, B, H, W, C, 3, 2, 1, stream);
}

// ---- plain fp32 <-> f32s (fp16 high/low pairs, 32-channel groups; cn_common.h) -------------
namespace {
__global__ void f32_to_f32s_kernel(const float *__restrict__ x, void *__restrict__ y, size_t npix,
                                   int C, int in_pitch, int out_pitch, float mul, uint32_t *range)
{
    const int c4n = (C + 3) >> 2;
    const size_t total = npix * c4n;
    float rng = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / c4n;
        const int n = (int)(i - p * c4n) * 4;
        cn_f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (n + 4 <= C && (in_pitch & 3) == 0) {
            v = *reinterpret_cast<const cn_f32x4 *>(x + p * in_pitch + n);
        } else {
            for (int e = 0; e < 4 && n + e < C; ++e) v[e] = x[p * in_pitch + n + e];
        }
        v = v * mul;
#pragma unroll
        for (int e = 0; e < 4; ++e) cn_rng_upd1_in(rng, v[e]);    // (user data may enter here: NaN-sticky)
        cn_store4_f32s(y, p, out_pitch, n, v);   // channels past C inside the group: zeros
    }
    cn_rng_commit(range, 1, rng);
}
__global__ void f32s_to_f32_kernel(const void *__restrict__ x, float *__restrict__ y, size_t npix,
                                   int C, int in_pitch, int out_pitch, float mul)
{
    const int c4n = (C + 3) >> 2;
    const size_t total = npix * c4n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / c4n;
        const int n = (int)(i - p * c4n) * 4;
        const cn_f32x4 v = cn_load4_f32s(x, p, in_pitch, n) * mul;
        if (n + 4 <= C && (out_pitch & 3) == 0) {
            *reinterpret_cast<cn_f32x4 *>(y + p * out_pitch + n) = v;
        } else {
            for (int e = 0; e < 4 && n + e < C; ++e) y[p * out_pitch + n + e] = v[e];
        }
    }
}
// max |x| of a plain fp32 NHWC tensor (calibration of the f32s exponents, engine.py)
__global__ void absmax_f32_kernel(const float *__restrict__ x, size_t npix, int C, int pitch,
                                  uint32_t *word)
{
    const size_t total = npix * (size_t)C;
    float rng = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / C;
        const float v = x[p * pitch + (i - p * C)];
        // a NaN / inf anywhere must not pass as "small": report it as +inf
        rng = (v - v == 0.f) ? __builtin_fmaxf(rng, __builtin_fabsf(v)) : __builtin_huge_valf();
    }
    cn_rng_commit1(word, rng);
}
}  // namespace

namespace {
// one 64-thread block per (launch, side): max over the slots -> hi / lo, slots re-zeroed
__global__ void range_fold_kernel(uint32_t *cur, uint32_t *hi, uint32_t *lo, uint32_t *summary)
{
    uint32_t *w = cur + ((size_t)blockIdx.x * CN_RANGE_SLOTS + threadIdx.x) * CN_RANGE_STRIDE;
    uint32_t v = *w;
    *w = 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o, 64));
    if (threadIdx.x == 0) {
        hi[blockIdx.x] = max(hi[blockIdx.x], v);
        if (v) lo[blockIdx.x] = min(lo[blockIdx.x], v);   // an all-zero forward is not a 'low' one (as in the digest)
        if (summary) {   // sticky two-word digest: largest value seen, smallest non-zero per-forward maximum
            atomicMax(&summary[0], v);
            if (v) atomicMin(&summary[1], v);
        }
    }
}
}  // namespace

extern "C" int cn_range_fold(uint32_t *cur, uint32_t *hi, uint32_t *lo, int n_launches, void *stream)
{
    return cn_range_fold_digest(cur, hi, lo, nullptr, n_launches, stream);
}

extern "C" int cn_range_fold_digest(uint32_t *cur, uint32_t *hi, uint32_t *lo, uint32_t *summary,
                                    int n_launches, void *stream)
{
    if (!cur || !hi || !lo) return CN_ERR_NULL;
    if (n_launches <= 0) return CN_OK;
    static_assert(CN_RANGE_SLOTS == 64, "one wave per (launch, side)");
    hipLaunchKernelGGL(range_fold_kernel, dim3(2 * n_launches), dim3(CN_RANGE_SLOTS), 0,
                       (hipStream_t)stream, cur, hi, lo, summary);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

extern "C" int cn_absmax_f32(const float *x, size_t npix, int C, int pitch, uint32_t *word,
                             void *stream)
{
    if (!x || !word) return CN_ERR_NULL;
    if (C <= 0 || pitch < C) return CN_ERR_SHAPE;
    const size_t total = npix * (size_t)C;
    if (!total) return CN_OK;
    hipLaunchKernelGGL(absmax_f32_kernel, dim3(blocks_for(total, 256, 512)), dim3(256), 0,
                       (hipStream_t)stream, x, npix, C, pitch, word);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

extern "C" int cn_f32_to_f32s(const float *x, void *y, size_t npix, int C, int in_pitch,
                              int out_pitch, void *stream)
{
    return cn_f32_to_f32s_scaled(x, y, npix, C, in_pitch, out_pitch, 1.f, nullptr, stream);
}

extern "C" int cn_f32_to_f32s_scaled(const float *x, void *y, size_t npix, int C, int in_pitch,
                                     int out_pitch, float mul, uint32_t *range, void *stream)
{
    if (mul == 0.f) mul = 1.f;
    if (!x || !y) return CN_ERR_NULL;
    if (C <= 0 || in_pitch < C || out_pitch < C) return CN_ERR_SHAPE;
    if ((out_pitch & 31) || (((uintptr_t)y) & 127u)) return CN_ERR_ALIGN;
    if ((in_pitch & 3) == 0 && !cn_aligned16(x)) return CN_ERR_ALIGN;
    const size_t total = npix * ((C + 3) >> 2);
    if (!total) return CN_OK;
    hipLaunchKernelGGL(f32_to_f32s_kernel, dim3(blocks_for(total, 256, 65535)), dim3(256), 0,
                       (hipStream_t)stream, x, y, npix, C, in_pitch, out_pitch, mul, range);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

extern "C" int cn_f32s_to_f32(const void *x, float *y, size_t npix, int C, int in_pitch,
                              int out_pitch, void *stream)
{
    return cn_f32s_to_f32_scaled(x, y, npix, C, in_pitch, out_pitch, 1.f, stream);
}

extern "C" int cn_f32s_to_f32_scaled(const void *x, float *y, size_t npix, int C, int in_pitch,
                                     int out_pitch, float mul, void *stream)
{
    if (mul == 0.f) mul = 1.f;
    if (!x || !y) return CN_ERR_NULL;
    if (C <= 0 || in_pitch < C || out_pitch < C) return CN_ERR_SHAPE;
    if ((in_pitch & 31) || (((uintptr_t)x) & 127u)) return CN_ERR_ALIGN;
    if ((out_pitch & 3) == 0 && !cn_aligned16(y)) return CN_ERR_ALIGN;
    const size_t total = npix * ((C + 3) >> 2);
    if (!total) return CN_OK;
    hipLaunchKernelGGL(f32s_to_f32_kernel, dim3(blocks_for(total, 256, 65535)), dim3(256), 0,
                       (hipStream_t)stream, x, y, npix, C, in_pitch, out_pitch, mul);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

// ---- reference-layout (NCHW) DCNv2 entry point --------------------------------
namespace {
struct DcnWs {
    size_t x_off, om_off, w_off, y_off, split_off, split_bytes, total;
};
DcnWs dcn_ws_plan(int B, int Cin, int H, int W, int Cout, int kh, int kw)
{
    DcnWs p;
    const size_t hw = (size_t)B * H * W;
    size_t o = 0;
    p.x_off = o;
    o += cn_align_up(hw * Cin * 4, 256);
    p.om_off = o;
    o += cn_align_up(hw * 32 * 4, 256);
    p.w_off = o;
    o += cn_align_up(cn_packed_conv_weight_floats(Cout, Cin, kh, kw) * 4, 256);
    p.y_off = o;
    o += cn_align_up(hw * Cout * 4, 256);
    p.split_off = o;
    p.split_bytes = cn_dcn_v2_forward_nhwc_workspace_bytes(B, Cin, H, W, Cout);
    o += cn_align_up(p.split_bytes, 256);
    p.total = o;
    return p;
}
}  // namespace

// cn_dcn_general.hip
int cn_dcn_general_launch(const float *input, const float *weight, const float *bias,
                          const float *offset, const float *mask, float *output, int B, int Cin,
                          int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw,
                          int dh, int dw, int dg, int mask_sigmoid, hipStream_t st);

// The tuned CenterNet path (NHWC implicit GEMM on MFMA) takes 3x3 kernels at 4-aligned channel
// counts; whether a call also has stride 1 / pad 1 / dilation 1 / one group is only known at
// the forward call, so the query sizes for it whenever the kernel is 3x3 (an upper bound).
static bool dcn_tuned_shape(int Cin, int kh, int kw)
{
    return kh == 3 && kw == 3 && Cin != 3 && (Cin & 3) == 0;
}

extern "C" size_t cn_dcn_v2_forward_workspace_bytes(int B, int Cin, int H, int W, int Cout,
                                                    int kernel_h, int kernel_w, int layout)
{
    if (layout != CN_LAYOUT_NCHW) return 0;
    if (B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0 || kernel_h <= 0 || kernel_w <= 0)
        return 0;
    if (!dcn_tuned_shape(Cin, kernel_h, kernel_w)) return 256;   // general kernel: no scratch
    return dcn_ws_plan(B, Cin, H, W, Cout, kernel_h, kernel_w).total;
}

extern "C" int cn_dcn_v2_forward_f32(const float *input, const float *weight, const float *bias,
                                     const float *offset, const float *mask, float *output, int B,
                                     int Cin, int H, int W, int Cout, int kernel_h, int kernel_w,
                                     int stride_h, int stride_w, int pad_h, int pad_w,
                                     int dilation_h, int dilation_w, int deformable_group,
                                     int apply_mask_sigmoid, void *workspace,
                                     size_t workspace_bytes, void *stream)
{
    if (!input || !weight || !bias || !offset || !mask || !output) return CN_ERR_NULL;
    if (B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return CN_ERR_SHAPE;
    if (kernel_h <= 0 || kernel_w <= 0 || stride_h <= 0 || stride_w <= 0 || pad_h < 0 ||
        pad_w < 0 || dilation_h <= 0 || dilation_w <= 0 || deformable_group <= 0 ||
        Cin % deformable_group != 0)
        return CN_ERR_SHAPE;
    const bool tuned = dcn_tuned_shape(Cin, kernel_h, kernel_w) && stride_h == 1 &&
                       stride_w == 1 && pad_h == 1 && pad_w == 1 && dilation_h == 1 &&
                       dilation_w == 1 && deformable_group == 1;
    if (!tuned)   // the rest of the reference operator's domain (cn_dcn_general.hip)
        return cn_dcn_general_launch(input, weight, bias, offset, mask, output, B, Cin, H, W, Cout,
                                     kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w,
                                     dilation_h, dilation_w, deformable_group,
                                     apply_mask_sigmoid, (hipStream_t)stream);
    if (!workspace) return CN_ERR_NULL;
    if (!cn_aligned16(workspace)) return CN_ERR_ALIGN;
    const DcnWs p = dcn_ws_plan(B, Cin, H, W, Cout, 3, 3);
    if (workspace_bytes < p.total) return CN_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)workspace;
    float *x_nhwc = (float *)(ws + p.x_off);
    float *om = (float *)(ws + p.om_off);
    float *wp = (float *)(ws + p.w_off);
    float *y_nhwc = (float *)(ws + p.y_off);
    int rc = cn_nchw_to_nhwc_f32(input, x_nhwc, B, Cin, H, W, Cin, stream);
    if (rc != CN_OK) return rc;
    const size_t tot = (size_t)B * H * W * 32;
    hipLaunchKernelGGL(pack_offset_mask_kernel, dim3(blocks_for(tot, 256, 65535)), dim3(256), 0, st,
                       offset, mask, om, H * W, tot);
    CN_CHECK_LAUNCH();
    rc = cn_pack_conv_weight_f32(weight, wp, Cout, Cin, 3, 3, stream);
    if (rc != CN_OK) return rc;
    rc = cn_dcn_v2_forward_nhwc_f32(x_nhwc, wp, bias, om, 32, nullptr, nullptr, y_nhwc, B, Cin, H, W,
                                    Cout, apply_mask_sigmoid, 0, ws + p.split_off, p.split_bytes,
                                    stream);
    if (rc != CN_OK) return rc;
    return cn_nhwc_to_nchw_f32(y_nhwc, output, B, Cout, H, W, Cout, stream);
}

extern "C" int cn_dw_conv_transpose_f32(const float *x, const float *w_taps, const float *add,
                                        float *y, int B, int H, int W, int C, int f, void *stream)
{
    if (!x || !w_taps || !y) return CN_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || f < 2 || (f & 1)) return CN_ERR_SHAPE;
    if (C & 3) return CN_ERR_UNSUPPORTED;
    if (!cn_aligned16(x) || !cn_aligned16(w_taps) || !cn_aligned16(y) || (add && !cn_aligned16(add)))
        return CN_ERR_ALIGN;
    const size_t total = (size_t)B * H * f * W * f * (C >> 2);
    hipLaunchKernelGGL(dw_deconv_kernel, dim3(blocks_for(total, 256, 65535)), dim3(256), 0,
                       (hipStream_t)stream, x, w_taps, add, y, B, H, W, C, f);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

extern "C" int cn_copy_channels_f32(const float *src, int src_pitch, float *dst, int dst_pitch,
                                    size_t npix, int C, void *stream)
{
    if (!src || !dst) return CN_ERR_NULL;
    if (C <= 0 || src_pitch < C || dst_pitch < C) return CN_ERR_SHAPE;
    if ((C & 3) || (src_pitch & 3) || (dst_pitch & 3)) return CN_ERR_UNSUPPORTED;
    if (!cn_aligned16(src) || !cn_aligned16(dst)) return CN_ERR_ALIGN;
    const size_t total = npix * (C >> 2);
    hipLaunchKernelGGL(copy_channels_kernel, dim3(blocks_for(total, 256, 65535)), dim3(256), 0,
                       (hipStream_t)stream, src, src_pitch, dst, dst_pitch, npix, C);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

extern "C" int cn_upsample2x_add_f32(const float *x, const float *add, float *y, int B, int H,
                                     int W, int C, void *stream)
{
    if (!x || !y) return CN_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return CN_ERR_SHAPE;
    if (C & 3) return CN_ERR_UNSUPPORTED;
    if (!cn_aligned16(x) || !cn_aligned16(y) || (add && !cn_aligned16(add))) return CN_ERR_ALIGN;
    const size_t total = (size_t)B * 4 * H * W * (C >> 2);
    hipLaunchKernelGGL(upsample2x_add_kernel, dim3(blocks_for(total, 256, 65535)), dim3(256), 0,
                       (hipStream_t)stream, x, add, y, B, H, W, C);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

namespace {
// fp16 variant: 8 channels (16 bytes) per thread
__global__ void upsample2x_add_f16_kernel(const _Float16 *__restrict__ x,
                                          const _Float16 *__restrict__ add,
                                          _Float16 *__restrict__ y, int B, int H, int W, int C)
{
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    const int c8n = C >> 3;
    const int OH = 2 * H, OW = 2 * W;
    const size_t total = (size_t)B * OH * OW * c8n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % c8n);
        size_t r = i / c8n;
        const int ox = (int)(r % OW);
        r /= OW;
        const int oy = (int)(r % OH);
        const int b = (int)(r / OH);
        h8 v = *reinterpret_cast<const h8 *>(x + (((size_t)b * H + (oy >> 1)) * W + (ox >> 1)) * C + c8 * 8);
        const size_t o = (((size_t)b * OH + oy) * OW + ox) * C + c8 * 8;
        if (add) v += *reinterpret_cast<const h8 *>(add + o);
        *reinterpret_cast<h8 *>(y + o) = v;
    }
}
}  // namespace

extern "C" int cn_upsample2x_add_f16(const void *x, const void *add, void *y, int B, int H, int W,
                                     int C, void *stream)
{
    if (!x || !y) return CN_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return CN_ERR_SHAPE;
    if (C & 7) return CN_ERR_UNSUPPORTED;
    if (!cn_aligned16(x) || !cn_aligned16(y) || (add && !cn_aligned16(add))) return CN_ERR_ALIGN;
    const size_t total = (size_t)B * 4 * H * W * (C >> 3);
    hipLaunchKernelGGL(upsample2x_add_f16_kernel, dim3(blocks_for(total, 256, 65535)), dim3(256), 0,
                       (hipStream_t)stream, (const _Float16 *)x, (const _Float16 *)add,
                       (_Float16 *)y, B, H, W, C);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

// ((image / 255. - mean) / std).astype(float32), HWC -> CHW on the HOST (base_detector.py:56-58): the
// float64 arithmetic numpy performs for `uint8_array / 255.`, then one rounding to float32.
extern "C" int cn_normalize_u8_chw_f32_host(const uint8_t *img, int h, int w, const float *mean,
                                            const float *std, float *out)
{
#pragma clang fp contract(off)
    if (!img || !mean || !std || !out) return CN_ERR_NULL;
    if (h <= 0 || w <= 0) return CN_ERR_SHAPE;
    const size_t hw = (size_t)h * w;
    for (int c = 0; c < 3; ++c) {
        const double m = (double)mean[c], sd = (double)std[c];
        double lut[256];   // 256 possible inputs per channel: the division is done once each
        for (int v = 0; v < 256; ++v) lut[v] = ((double)v / 255.0 - m) / sd;
        float *o = out + (size_t)c * hw;
        const uint8_t *p = img + c;
        for (size_t i = 0; i < hw; ++i) o[i] = (float)lut[p[3 * i]];
    }
    return CN_OK;
}

// ---- soft-NMS on a small host array (merge_outputs, detectors/ctdet.py:63-64) -------------
// Host-side, in place, same greedy/swap/discard order as src/lib/external/nms.pyx:77-170
// (soft_nms) and :172-275 (soft_nms_39): `stride` floats per row, box in [0..3], score in [4].
// The max box swaps with row i as a whole; a discarded box takes columns 0..4 of row N-1 and
// SWAPS columns 5.. with it (nms.pyx:260-268), so the whole in-place array -- rows past N
// included, which MultiPoseDetector.merge_outputs returns -- equals the reference's.  Returns
// the number of boxes kept (the rows [0, N) at exit).  Pinned bit-for-bit against the
// reference's own cython build (oracle/_ref, tests/test_oracle_ref.py).
extern "C" int cn_soft_nms_f32(float *boxes, int n, int stride, float sigma, float Nt,
                               float threshold, int method)
{
#pragma clang fp contract(off)
    if (!boxes || n < 0 || stride < 5) return CN_ERR_SHAPE;
    int N = n;
    float *tmp = (float *)alloca(sizeof(float) * (size_t)stride);
    for (int i = 0; i < n; ++i) {  // the reference iterates over the ORIGINAL count
        if (i >= N) break;         // rows past N are only self-swapped there: no effect
        float *bi = boxes + (size_t)i * stride;
        float maxscore = bi[4];
        int maxpos = i;
        for (int pos = i + 1; pos < N; ++pos)
            if (maxscore < boxes[(size_t)pos * stride + 4]) {
                maxscore = boxes[(size_t)pos * stride + 4];
                maxpos = pos;
            }
        float *bm = boxes + (size_t)maxpos * stride;
        for (int c = 0; c < stride; ++c) tmp[c] = bi[c];
        for (int c = 0; c < stride; ++c) bi[c] = bm[c];
        for (int c = 0; c < stride; ++c) bm[c] = tmp[c];
        const float tx1 = bi[0], ty1 = bi[1], tx2 = bi[2], ty2 = bi[3];
        int pos = i + 1;
        while (pos < N) {
            float *bp = boxes + (size_t)pos * stride;
            const float x1 = bp[0], y1 = bp[1], x2 = bp[2], y2 = bp[3];
            // Cython writes the integer literals of nms.pyx:133-143 as the double constant 1.0, so
            // `x2 - x1 + 1` is a float difference widened to double; each assignment to a `cdef
            // float` rounds once.  Reproduced here term by term.
            const float area = (float)(((double)(x2 - x1) + 1.0) * ((double)(y2 - y1) + 1.0));
            const float iw = (float)((double)(fminf(tx2, x2) - fmaxf(tx1, x1)) + 1.0);
            if (iw > 0) {
                const float ih = (float)((double)(fminf(ty2, y2) - fmaxf(ty1, y1)) + 1.0);
                if (ih > 0) {
                    const float ua = (float)((((double)(tx2 - tx1) + 1.0) * ((double)(ty2 - ty1) + 1.0) +
                                              (double)area) - (double)(iw * ih));
                    const float ov = (iw * ih) / ua;
                    float weight;
                    if (method == 1)
                        weight = ov > Nt ? (float)(1.0 - (double)ov) : 1.0f;
                    else if (method == 2)
                        weight = (float)exp((double)(-(ov * ov) / sigma));  // np.exp on a C float
                    else
                        weight = ov > Nt ? 0 : 1;
                    bp[4] = weight * bp[4];
                    if (bp[4] < threshold) {
                        float *bl = boxes + (size_t)(N - 1) * stride;
                        for (int c = 0; c < 5; ++c) bp[c] = bl[c];
                        for (int c = 5; c < stride; ++c) {
                            const float t = bp[c];
                            bp[c] = bl[c];
                            bl[c] = t;
                        }
                        N = N - 1;
                        pos = pos - 1;
                    }
                }
            }
            pos = pos + 1;
        }
    }
    return N;
}

// ---- flip-test averaging (detectors/ctdet.py:34-37, multi_pose.py:44-55, models/utils.py:28-50): image 1
// of the pair is the mirrored frame; out[c, y, x] = (f(x0[c, y, x]) + sign[c] * f(x1[src[c], y, W-1-x])) / 2
// with f = the logistic when asked (then written back in place: the reference's sigmoid_()), src = the
// left / right joint permutation, sign = -1 for the x components of joint offsets
namespace {
__global__ void flip_average_kernel(float *__restrict__ x, float *__restrict__ out, int C, int H, int W,
                                    const int32_t *__restrict__ src, const float *__restrict__ sign,
                                    int apply_sigmoid)
{
    const size_t total = (size_t)C * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int xx = (int)(i % W);
        size_t r = i / W;
        const int y = (int)(r % H);
        const int c = (int)(r / H);
        const int cs = src ? src[c] : c;
        const size_t j = total + ((size_t)cs * H + y) * W + (W - 1 - xx);
        float a = x[i], b = x[j];
        if (apply_sigmoid) {
            a = sigmoidf_ref(a);      // the decode kernels' definition (cn_common.h)
            b = sigmoidf_ref(b);
            x[i] = a;
            x[j] = b;
        }
        if (sign) b = b * sign[c];
        out[i] = (a + b) / 2.0f;
    }
}
}  // namespace

extern "C" int cn_flip_average_f32(float *x_pair, float *out, int C, int H, int W, const int32_t *chan_src,
                                   const float *chan_sign, int apply_sigmoid, void *stream)
{
    if (!x_pair || !out) return CN_ERR_NULL;
    if (C <= 0 || H <= 0 || W <= 0) return CN_ERR_SHAPE;
    const size_t total = (size_t)C * H * W;
    hipLaunchKernelGGL(flip_average_kernel, dim3(blocks_for(total, 256, 65535)), dim3(256), 0,
                       (hipStream_t)stream, x_pair, out, C, H, W, chan_src, chan_sign, apply_sigmoid ? 1 : 0);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

// ---- box calibration (bench.py "box_calibration"): what THIS box's matrix pipe and HBM deliver
// right now, so that a headline can be read against the box it ran on.  Not on the product path.
namespace {

// register-only v_mfma_f32_32x32x16_f16 loop: 4 independent accumulators per wave, no memory
__global__ __launch_bounds__(256) void calib_mfma_f16_kernel(float *sink, int iters)
{
    cn_f16x8v a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = (_Float16)(0.001f * (float)((threadIdx.x + i) & 7));
        b[i] = (_Float16)(0.002f * (float)((threadIdx.x + 3 * i) & 7));
    }
    cn_f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    if (s == 12345.678f) sink[0] = s;     // never true: keeps the accumulators live
}

__global__ __launch_bounds__(256) void calib_copy_kernel(const cn_f32x4 *__restrict__ src,
                                                         cn_f32x4 *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        dst[i] = src[i];
}

// ONE lane walks a chain of dependent loads (chain[i] = index of the next element), then a chain of
// dependent device-scope atomic additions on one word: out[0], out[1] = ticks of the constant 100 MHz
// clock (s_memrealtime) each took, out[2] = a value that depends on every step.  What bandwidth and the
// matrix loop do not show: the latency of this box's memory system, which the kernels with round trips
// on their critical path (arrival counters, barrier-paced DMA) are sensitive to.
__global__ __launch_bounds__(64) void calib_latency_kernel(const uint32_t *__restrict__ chain, uint32_t start,
                                                           int steps, uint32_t *atom, unsigned long long *out)
{
    if (threadIdx.x != 0) return;
    uint32_t i = start;
    const unsigned long long t0 = wall_clock64();
    for (int s = 0; s < steps; ++s) {
        asm volatile("" : "+v"(i));            // a vector load (L1 -> L2 -> Infinity Cache -> HBM), not the scalar cache
        i = chain[i];
    }
    asm volatile("" : "+v"(i));
    const unsigned long long t1 = wall_clock64();
    uint32_t v = i & 1u;
    for (int s = 0; s < steps; ++s)
        v = __hip_atomic_fetch_add(atom, (v >> 31) + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" : "+v"(v));
    const unsigned long long t2 = wall_clock64();
    out[0] = t1 - t0;
    out[1] = t2 - t1;
    out[2] = (unsigned long long)i + v;
}

// The shader clock right now: one lane counts core cycles (s_memtime) over `ticks` ticks of the constant
// 100 MHz clock (s_memrealtime).  out[0] = core cycles, out[1] = 100 MHz ticks.  Launched right behind
// a loaded stretch it reads the clock the power management left the part at -- which the register-only
// matrix loop above does not pull down the way the real kernel mix does.
__global__ __launch_bounds__(64) void calib_clock_kernel(unsigned long long *out, int ticks)
{
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    const unsigned long long c0 = clock64();
    unsigned long long t1 = t0;
    while ((long long)(t1 - t0) < (long long)ticks) t1 = wall_clock64();
    const unsigned long long c1 = clock64();
    out[0] = c1 - c0;
    out[1] = t1 - t0;
}

}  // namespace

extern "C" int cn_calib_clock(unsigned long long *out, int ticks, void *stream)
{
    if (!out) return CN_ERR_NULL;
    if (ticks <= 0 || ticks > 100000000) return CN_ERR_SHAPE;
    hipLaunchKernelGGL(calib_clock_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out, ticks);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

extern "C" int cn_calib_latency(const uint32_t *chain, uint32_t start, int steps, uint32_t *atom,
                                unsigned long long *out, void *stream)
{
    if (!chain || !atom || !out) return CN_ERR_NULL;
    if (steps <= 0) return CN_ERR_SHAPE;
    hipLaunchKernelGGL(calib_latency_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, chain, start, steps, atom, out);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

extern "C" double cn_calib_mfma_f16(float *sink, int iters, void *stream)
{
    if (!sink || iters <= 0) return 0.0;
    const int blocks = 256 * 4;            // 4 workgroups of 4 waves per CU: 4 waves per SIMD
    hipLaunchKernelGGL(calib_mfma_f16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, sink, iters);
    if (hipGetLastError() != hipSuccess) return 0.0;
    // FLOPs of the launch: waves x iters x 16 MFMAs x 2 * 32 * 32 * 16
    return (double)blocks * 4.0 * (double)iters * 16.0 * 32768.0;
}

extern "C" int cn_calib_copy(const void *src, void *dst, size_t bytes, void *stream)
{
    if (!src || !dst) return CN_ERR_NULL;
    if (!cn_aligned16(src) || !cn_aligned16(dst) || (bytes & 15)) return CN_ERR_ALIGN;
    if (!bytes) return CN_OK;
    hipLaunchKernelGGL(calib_copy_kernel, dim3(256 * 16), dim3(256), 0, (hipStream_t)stream,
                       (const cn_f32x4 *)src, (cn_f32x4 *)dst, bytes / 16);
    CN_CHECK_LAUNCH();
    return CN_OK;
}
