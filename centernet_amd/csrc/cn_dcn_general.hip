// cn_dcn_general.hip -- modulated deformable convolution (DCNv2) forward over the reference
// operator's WHOLE domain: any kernel size, stride, padding, dilation, deformable-group count
// and channel count, in the reference's own NCHW layout.
//
// Replaces: dcn_v2_cuda_forward (DCNv2/src/dcn_v2_cuda.c:10-102) for every configuration that
// the tuned CenterNet path (cn_conv.hip, igemm_kernel<A_DCN>: 3x3 / stride 1 / pad 1 /
// dilation 1 / one deformable group / Cin % 4 == 0) does not take -- e.g. the reference's own
// test shapes, DCNv2/test.py:16-19 (inC = 2) and :169-179 (deformable_groups = 2).
//
// Same fusion as the tuned path, scalar fp32 arithmetic instead of MFMA: a workgroup owns
// 64 output pixels x 64 output channels of one image; per 32-deep slice of K = Cin*kh*kw it
//   * samples the 32 x 64 column values ONCE into LDS (lanes run along pixels, so the
//     offset / mask reads of the NCHW maps are coalesced; sampling rule and deformable-group
//     indexing exactly dcn_v2_im2col_cuda.cu:118-180, bilinear :18-47),
//   * stages the 64 x 32 weight slice, and
//   * accumulates a 4 x 4 register tile per thread.
// No column buffer, no per-sample host loop, bias in the epilogue.
#include "cn_common.h"

namespace {

constexpr int G_NT = 256;
constexpr int G_BM = 64;   // pixels per workgroup
constexpr int G_BN = 64;   // output channels per workgroup
constexpr int G_KC = 32;   // K slice

struct DcnGenArgs {
    const float *x, *w, *bias, *offset, *mask;
    float *y;
    int B, Cin, H, W, Cout, Ho, Wo;
    int kh, kw, sh, sw, ph, pw, dh, dw, dg, cpg, mask_sigmoid;
};

__device__ __forceinline__ float gen_bilinear(const float *__restrict__ plane, int H, int W,
                                              float h, float w)
{
    // dcn_v2_im2col_cuda.cu:18-47
    const int h_low = (int)floorf(h), w_low = (int)floorf(w);
    const int h_high = h_low + 1, w_high = w_low + 1;
    const float lh = h - (float)h_low, lw = w - (float)w_low;
    const float hh = 1.0f - lh, hw = 1.0f - lw;
    float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
    if (h_low >= 0 && w_low >= 0) v1 = plane[h_low * W + w_low];
    if (h_low >= 0 && w_high <= W - 1) v2 = plane[h_low * W + w_high];
    if (h_high <= H - 1 && w_low >= 0) v3 = plane[h_high * W + w_low];
    if (h_high <= H - 1 && w_high <= W - 1) v4 = plane[h_high * W + w_high];
    const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
    return w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
}

__global__ __launch_bounds__(G_NT) void dcn_general_kernel(const DcnGenArgs a)
{
    __shared__ float cols[G_KC][G_BM];          // [k][pixel]
    __shared__ float wts[G_KC][G_BN + 1];       // [k][cout]

    const int t = threadIdx.x;
    const int b = blockIdx.z;
    const int HoWo = a.Ho * a.Wo;
    const int p0 = blockIdx.x * G_BM;
    const int co0 = blockIdx.y * G_BN;
    const int khkw = a.kh * a.kw;
    const int K = a.Cin * khkw;

    // sampling role: one pixel, every 4th k of the slice
    const int sp = t & (G_BM - 1);
    const int sr = t >> 6;
    const int pix = p0 + sp;
    const bool pix_ok = pix < HoWo;
    const int h_col = pix_ok ? pix / a.Wo : 0;
    const int w_col = pix_ok ? pix - h_col * a.Wo : 0;
    const int h_in = h_col * a.sh - a.ph;
    const int w_in = w_col * a.sw - a.pw;
    const float *xb = a.x + (size_t)b * a.Cin * a.H * a.W;
    const float *offb = a.offset + (size_t)b * a.dg * 2 * khkw * HoWo;
    const float *mskb = a.mask + (size_t)b * a.dg * khkw * HoWo;

    // compute role: 4 pixels x 4 output channels
    const int tx = t & 15, ty = t >> 4;
    float acc[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] = 0.f;

    for (int k0 = 0; k0 < K; k0 += G_KC) {
        // ---- columns: (c, tap) x pixel, sampled exactly like the reference kernel ----
#pragma unroll
        for (int q = 0; q < G_KC / 4; ++q) {
            const int r = sr + 4 * q;
            const int kk = k0 + r;
            float val = 0.f;
            if (pix_ok && kk < K) {
                const int c = kk / khkw;
                const int tap = kk - c * khkw;
                const int i = tap / a.kw, j = tap - i * a.kw;
                const int g = c / a.cpg;   // deformable group (dcn_v2_im2col_cuda.cu:144)
                const float *og = offb + (size_t)g * 2 * khkw * HoWo;
                const float oh = og[(size_t)(2 * tap) * HoWo + pix];
                const float ow = og[(size_t)(2 * tap + 1) * HoWo + pix];
                float m = mskb[((size_t)g * khkw + tap) * HoWo + pix];
                if (a.mask_sigmoid) m = 1.0f / (1.0f + expf(-m));
                const float h_im = (float)(h_in + i * a.dh) + oh;
                const float w_im = (float)(w_in + j * a.dw) + ow;
                if (h_im > -1 && w_im > -1 && h_im < a.H && w_im < a.W)   // :165
                    val = gen_bilinear(xb + (size_t)c * a.H * a.W, a.H, a.W, h_im, w_im);
                val *= m;
            }
            cols[r][sp] = val;
        }
        // ---- weights: w[(co0+n)][k0 + r], K-contiguous in the reference's OIHW tensor ----
#pragma unroll
        for (int q = 0; q < (G_KC * G_BN) / G_NT; ++q) {
            const int e = t + q * G_NT;
            const int r = e & (G_KC - 1), n = e >> 5;
            const int co = co0 + n, kk = k0 + r;
            wts[r][n] = (co < a.Cout && kk < K) ? a.w[(size_t)co * K + kk] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int r = 0; r < G_KC; ++r) {
            const cn_f32x4 av = *reinterpret_cast<const cn_f32x4 *>(&cols[r][tx * 4]);
            float bv[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) bv[v] = wts[r][ty * 4 + v];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[u][v] = fmaf(av[u], bv[v], acc[u][v]);
        }
        __syncthreads();
    }

    float *yb = a.y + (size_t)b * a.Cout * HoWo;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int co = co0 + ty * 4 + v;
        if (co >= a.Cout) continue;
        const float bs = a.bias[co];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = p0 + tx * 4 + u;
            if (p < HoWo) yb[(size_t)co * HoWo + p] = acc[u][v] + bs;
        }
    }
}

}  // namespace

// Internal (declared in cn_misc.hip): launched by cn_dcn_v2_forward_f32 for every
// configuration outside the tuned path's domain.
int cn_dcn_general_launch(const float *input, const float *weight, const float *bias,
                          const float *offset, const float *mask, float *output, int B, int Cin,
                          int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw,
                          int dh, int dw, int dg, int mask_sigmoid, hipStream_t st)
{
    DcnGenArgs a;
    a.x = input; a.w = weight; a.bias = bias; a.offset = offset; a.mask = mask; a.y = output;
    a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout;
    a.Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;   // dcn_v2_cuda.c:40-41
    a.Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
    if (a.Ho <= 0 || a.Wo <= 0) return CN_ERR_SHAPE;
    a.kh = kh; a.kw = kw; a.sh = sh; a.sw = sw; a.ph = ph; a.pw = pw; a.dh = dh; a.dw = dw;
    a.dg = dg; a.cpg = Cin / dg; a.mask_sigmoid = mask_sigmoid;
    if (B > 65535 || cn_cdiv(Cout, G_BN) > 65535) return CN_ERR_UNSUPPORTED;
    dim3 grid(cn_cdiv(a.Ho * a.Wo, G_BM), cn_cdiv(Cout, G_BN), B);
    hipLaunchKernelGGL(dcn_general_kernel, grid, dim3(G_NT), 0, st, a);
    CN_CHECK_LAUNCH();
    return CN_OK;
}
