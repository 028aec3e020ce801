/*
 * oracle/dcn_v2_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of the reference's modulated deformable convolution
 * (DCNv2) forward pass.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this file; the product path (centernet_amd/) never
 * does.
 *
 * What it follows (paths relative to /root/reference):
 *   - bilinear sampler ......... src/lib/models/networks/DCNv2/src/cuda/dcn_v2_im2col_cuda.cu:18-47
 *   - tap / offset / mask index, sampling window test (h_im > -1 && ... < H)
 *                                src/lib/models/networks/DCNv2/src/cuda/dcn_v2_im2col_cuda.cu:118-180
 *   - output shape rule ......... src/lib/models/networks/DCNv2/src/dcn_v2_cuda.c:40-41
 *   - per-sample order: y = bias (sgemm with ones, k=1), then y += W . columns
 *                                src/lib/models/networks/DCNv2/src/dcn_v2_cuda.c:61-97
 *
 * Parity status: the reference DCNv2 cannot be built in this image (THC,
 * torch.utils.ffi, nvcc are gone) and its CPU entry point only prints
 * (DCNv2/src/dcn_v2.c:5-16), so this restatement is pinned by the reference's
 * one known-answer test (DCNv2/test.py:32-65, zero-offset identity) plus the
 * analytic cases in tests/test_oracle_dcn.py (offset 0 / mask 1 == conv2d,
 * integer shifts, half-pixel bilinear, window-edge rule).
 *
 * The contraction is accumulated in double and rounded once to float ("ideal
 * fp32 result"); cuBLAS's summation order is unspecified, so any fp32 order is
 * an equally valid reading of the reference.  acc_mode=1 switches to a plain
 * sequential fp32 fmaf-free accumulation for comparison.
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

/* dcn_v2_im2col_cuda.cu:18-47 */
static float oracle_bilinear(const float *plane, int data_width, int height,
                             int width, float h, float w)
{
    int h_low = (int)floorf(h);
    int w_low = (int)floorf(w);
    int h_high = h_low + 1;
    int w_high = w_low + 1;

    float lh = h - (float)h_low;
    float lw = w - (float)w_low;
    float hh = 1.0f - lh, hw = 1.0f - lw;

    float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
    if (h_low >= 0 && w_low >= 0)
        v1 = plane[h_low * data_width + w_low];
    if (h_low >= 0 && w_high <= width - 1)
        v2 = plane[h_low * data_width + w_high];
    if (h_high <= height - 1 && w_low >= 0)
        v3 = plane[h_high * data_width + w_low];
    if (h_high <= height - 1 && w_high <= width - 1)
        v4 = plane[h_high * data_width + w_high];

    float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
    /* Same association as the reference expression
     * (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4), no fused multiply-add. */
    volatile float t1 = w1 * v1;
    volatile float t2 = w2 * v2;
    volatile float t3 = w3 * v3;
    volatile float t4 = w4 * v4;
    float val = ((t1 + t2) + t3) + t4;
    return val;
}

/*
 * Modulated deformable im2col for ONE sample.
 * columns layout: [(c*kh*kw + tap), h_out, w_out]   (dcn_v2_im2col_cuda.cu:144-176)
 */
static void oracle_im2col(const float *im, const float *offset,
                          const float *mask, int channels, int height,
                          int width, int height_col, int width_col, int kh,
                          int kw, int pad_h, int pad_w, int stride_h,
                          int stride_w, int dil_h, int dil_w, int dg,
                          float *columns)
{
    const int cpg = channels / dg; /* channel_per_deformable_group */
    const size_t hw_col = (size_t)height_col * width_col;
    for (int c = 0; c < channels; ++c) {
        const int g = c / cpg;
        const float *plane = im + (size_t)c * height * width;
        const float *off_g = offset + (size_t)g * 2 * kh * kw * hw_col;
        const float *msk_g = mask + (size_t)g * kh * kw * hw_col;
        for (int h_col = 0; h_col < height_col; ++h_col) {
            for (int w_col = 0; w_col < width_col; ++w_col) {
                const int h_in = h_col * stride_h - pad_h;
                const int w_in = w_col * stride_w - pad_w;
                for (int i = 0; i < kh; ++i) {
                    for (int j = 0; j < kw; ++j) {
                        const int tap = i * kw + j;
                        const size_t pos = (size_t)h_col * width_col + w_col;
                        const float offset_h = off_g[(size_t)(2 * tap) * hw_col + pos];
                        const float offset_w = off_g[(size_t)(2 * tap + 1) * hw_col + pos];
                        const float m = msk_g[(size_t)tap * hw_col + pos];
                        float val = 0.f;
                        const float h_im = (float)(h_in + i * dil_h) + offset_h;
                        const float w_im = (float)(w_in + j * dil_w) + offset_w;
                        if (h_im > -1 && w_im > -1 && h_im < height && w_im < width)
                            val = oracle_bilinear(plane, width, height, width, h_im, w_im);
                        columns[((size_t)c * kh * kw + tap) * hw_col + pos] = val * m;
                    }
                }
            }
        }
    }
}

/*
 * Full forward, NCHW everywhere (the reference's layout):
 *   input  (B, Cin, H, W)        offset (B, dg*2*kh*kw, Ho, Wo)
 *   mask   (B, dg*kh*kw, Ho, Wo) weight (Cout, Cin, kh, kw)   bias (Cout)
 *   output (B, Cout, Ho, Wo)
 * Returns 0 on success, -1 on a shape error (the reference raises THError,
 * dcn_v2_cuda.c:33-38), -2 on allocation failure.
 */
int oracle_dcn_v2_forward(const float *input, const float *weight,
                          const float *bias, const float *offset,
                          const float *mask, float *output, int B, int Cin,
                          int H, int W, int Cout, int kh, int kw, int stride_h,
                          int stride_w, int pad_h, int pad_w, int dil_h,
                          int dil_w, int dg, int acc_mode)
{
    if (B < 0 || Cin <= 0 || Cout <= 0 || dg <= 0 || Cin % dg != 0 || kh <= 0 ||
        kw <= 0 || stride_h <= 0 || stride_w <= 0)
        return -1;
    const int Ho = (H + 2 * pad_h - (dil_h * (kh - 1) + 1)) / stride_h + 1;
    const int Wo = (W + 2 * pad_w - (dil_w * (kw - 1) + 1)) / stride_w + 1;
    if (Ho <= 0 || Wo <= 0)
        return -1;
    const size_t hw = (size_t)Ho * Wo;
    const size_t Kdim = (size_t)Cin * kh * kw;
    float *columns = (float *)malloc(sizeof(float) * Kdim * hw);
    if (!columns)
        return -2;

    for (int b = 0; b < B; ++b) {
        const float *in_n = input + (size_t)b * Cin * H * W;
        const float *off_n = offset + (size_t)b * dg * 2 * kh * kw * hw;
        const float *msk_n = mask + (size_t)b * dg * kh * kw * hw;
        float *out_n = output + (size_t)b * Cout * hw;

        oracle_im2col(in_n, off_n, msk_n, Cin, H, W, Ho, Wo, kh, kw, pad_h,
                      pad_w, stride_h, stride_w, dil_h, dil_w, dg, columns);

#pragma omp parallel for schedule(static)
        for (int o = 0; o < Cout; ++o) {
            const float *wrow = weight + (size_t)o * Kdim;
            float *orow = out_n + (size_t)o * hw;
            if (acc_mode == 0) {
                double *acc = (double *)malloc(sizeof(double) * hw);
                for (size_t p = 0; p < hw; ++p)
                    acc[p] = (double)bias[o]; /* bias first (dcn_v2_cuda.c:75-78) */
                for (size_t k = 0; k < Kdim; ++k) {
                    const double wv = (double)wrow[k];
                    const float *crow = columns + k * hw;
                    for (size_t p = 0; p < hw; ++p)
                        acc[p] += wv * (double)crow[p];
                }
                for (size_t p = 0; p < hw; ++p)
                    orow[p] = (float)acc[p];
                free(acc);
            } else {
                for (size_t p = 0; p < hw; ++p)
                    orow[p] = bias[o];
                for (size_t k = 0; k < Kdim; ++k) {
                    const float wv = wrow[k];
                    const float *crow = columns + k * hw;
                    for (size_t p = 0; p < hw; ++p) {
                        volatile float prod = wv * crow[p];
                        orow[p] = orow[p] + prod;
                    }
                }
            }
        }
    }
    free(columns);
    return 0;
}

/* Exposed for the im2col-only checks (window-edge rule, mask product). */
int oracle_dcn_v2_im2col(const float *input, const float *offset,
                         const float *mask, float *columns, int Cin, int H,
                         int W, int kh, int kw, int stride_h, int stride_w,
                         int pad_h, int pad_w, int dil_h, int dil_w, int dg)
{
    if (Cin <= 0 || dg <= 0 || Cin % dg != 0)
        return -1;
    const int Ho = (H + 2 * pad_h - (dil_h * (kh - 1) + 1)) / stride_h + 1;
    const int Wo = (W + 2 * pad_w - (dil_w * (kw - 1) + 1)) / stride_w + 1;
    if (Ho <= 0 || Wo <= 0)
        return -1;
    oracle_im2col(input, offset, mask, Cin, H, W, Ho, Wo, kh, kw, pad_h, pad_w,
                  stride_h, stride_w, dil_h, dil_w, dg, columns);
    return 0;
}
