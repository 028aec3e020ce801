/*
 * oracle/dcn_v2_ref_driver.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Host driver around the REFERENCE's own `modulated_deformable_im2col_cuda`
 * (DCNv2/src/cuda/dcn_v2_im2col_cuda.cu:314-337, compiled into the same library from
 * /root/reference by oracle/Makefile).  It plays the role of `dcn_v2_cuda_forward`
 * (DCNv2/src/dcn_v2_cuda.c:10-102), whose THC tensor plumbing cannot be built any more:
 *
 *   for each sample b (dcn_v2_cuda.c:61):
 *       output_n = bias (x) ones                       (:71-78, sgemm with k = 1)
 *       columns  = reference im2col(input_n, ...)      (:80-86, batch_size = 1)
 *       output_n += weight . columns                   (:92-96)
 *
 * The two sgemms are cuBLAS calls in the reference (summation order unspecified); here the
 * contraction is accumulated in double and rounded once, the same reading as
 * oracle/dcn_v2_oracle.c, so that any difference between the two libraries is a difference
 * in the SAMPLING arithmetic -- which is the reference's own code on this side.
 */
#include <cstddef>
#include <cstdlib>
#include <vector>

typedef void *cudaStream_t;
extern "C" void modulated_deformable_im2col_cuda(
    cudaStream_t stream, const float *data_im, const float *data_offset, const float *data_mask,
    const int batch_size, const int channels, const int height_im, const int width_im,
    const int height_col, const int width_col, const int kernel_h, const int kenerl_w,
    const int pad_h, const int pad_w, const int stride_h, const int stride_w,
    const int dilation_h, const int dilation_w, const int deformable_group, float *data_col);

extern "C" int ref_dcn_v2_out_hw(int H, int W, int kh, int kw, int sh, int sw, int ph, int pw,
                                 int dh, int dw, int *Ho, int *Wo)
{
    /* dcn_v2_cuda.c:40-41 */
    *Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;
    *Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
    return (*Ho > 0 && *Wo > 0) ? 0 : -1;
}

/* columns for one sample, layout [(c*kh*kw + tap)][Ho][Wo] (batch_size = 1, as the
 * reference's forward calls it). */
extern "C" int ref_dcn_v2_im2col(const float *input, const float *offset, const float *mask,
                                 float *columns, int Cin, int H, int W, int kh, int kw, int sh,
                                 int sw, int ph, int pw, int dh, int dw, int dg)
{
    int Ho, Wo;
    if (Cin <= 0 || dg <= 0 || Cin % dg != 0 ||
        ref_dcn_v2_out_hw(H, W, kh, kw, sh, sw, ph, pw, dh, dw, &Ho, &Wo))
        return -1;
    modulated_deformable_im2col_cuda(nullptr, input, offset, mask, 1, Cin, H, W, Ho, Wo, kh, kw,
                                     ph, pw, sh, sw, dh, dw, dg, columns);
    return 0;
}

extern "C" int ref_dcn_v2_forward(const float *input, const float *weight, const float *bias,
                                  const float *offset, const float *mask, float *output, int B,
                                  int Cin, int H, int W, int Cout, int kh, int kw, int sh, int sw,
                                  int ph, int pw, int dh, int dw, int dg)
{
    int Ho, Wo;
    if (Cin <= 0 || Cout <= 0 || dg <= 0 || Cin % dg != 0 ||
        ref_dcn_v2_out_hw(H, W, kh, kw, sh, sw, ph, pw, dh, dw, &Ho, &Wo))
        return -1;
    const size_t hw = (size_t)Ho * Wo, K = (size_t)Cin * kh * kw;
    std::vector<float> columns(K * hw);
    for (int b = 0; b < B; ++b) {
        const float *in_n = input + (size_t)b * Cin * H * W;
        const float *off_n = offset + (size_t)b * dg * 2 * kh * kw * hw;
        const float *msk_n = mask + (size_t)b * dg * kh * kw * hw;
        float *out_n = output + (size_t)b * Cout * hw;
        modulated_deformable_im2col_cuda(nullptr, in_n, off_n, msk_n, 1, Cin, H, W, Ho, Wo, kh,
                                         kw, ph, pw, sh, sw, dh, dw, dg, columns.data());
#pragma omp parallel for schedule(static)
        for (int o = 0; o < Cout; ++o) {
            std::vector<double> acc(hw, (double)bias[o]);
            const float *wrow = weight + (size_t)o * K;
            for (size_t k = 0; k < K; ++k) {
                const double wv = wrow[k];
                const float *crow = columns.data() + k * hw;
                for (size_t p = 0; p < hw; ++p) acc[p] += wv * (double)crow[p];
            }
            for (size_t p = 0; p < hw; ++p) out_n[(size_t)o * hw + p] = (float)acc[p];
        }
    }
    return 0;
}
