// cn_stem.hip -- first convolution of the backbones (3-channel NCHW image -> NHWC features)
// as an im2col-free LDS tiling on the fp32 matrix cores.
//
// Replaces: conv1 + bn1 + relu of resnet_dcn.py:138-141 / msra_resnet.py (7x7/2, 3->64),
// base_layer of pose_dla_dcn.py:229-233 (7x7/1, 3->16), pre[0] of large_hourglass.py:193
// (7x7/2, 3->128).
//
// The generic implicit-GEMM kernel builds its A tile with one global load per element for
// this layer (Cin = 3 gives no channel vector to load), which made the stem gather-bound
// (50 TFLOP/s).  Here one workgroup owns 128 consecutive output pixels of one image:
//   * the input window those pixels can touch (<= 9 x 261 x 3 floats) is copied ONCE into
//     LDS with row-contiguous global reads, zero-filled outside the image, so the padding
//     rule costs nothing afterwards;
//   * the whole packed weight matrix [Cout tile][K = taps*3, padded to 8] sits in LDS too;
//   * the K loop has no barriers and no global loads: MFMA A fragments are read straight
//     from the window (address = row base of the pixel + table offset of k), B fragments with
//     ds_read_b128.
#include "cn_common.h"

namespace {

constexpr int NT = 256;
constexpr int BM = 128;       // pixels per workgroup
constexpr int WIN_MAX = 8192; // floats of input window (32 KB)

struct StemArgs {
    const float *x;  // (B,3,H,W)
    const float *w;  // packed [cout_pad][KP], k = tap*3 + c
    const float *scale, *shift;
    float *y;        // (B,Ho,Wo,out_pitch)
    int H, W, Ho, Wo, Cout, KH, KW, stride, pad, relu, out_pitch, KP, tiles_per_image, cout_pad;
};

template <int BN>
__global__ __launch_bounds__(NT) void stem_conv_f32_kernel(const StemArgs a)
{
    constexpr int WN = BN / 32;            // waves along N (1 or 2)
    constexpr int WM = 4 / WN;             // waves along M
    constexpr int TM = BM / WM;            // 32 or 64 pixels per wave
    constexpr int MB = TM / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int KP = a.KP, LDW = KP + 4;
    float *win = reinterpret_cast<float *>(smem);              // [3][WY][WXP]
    float *Ws = win + WIN_MAX;                                  // [BN][LDW]
    int *kb = reinterpret_cast<int *>(Ws + BN * LDW);           // [KP]
    int *rb = kb + KP;                                          // [BM]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int b = blockIdx.z;
    const int n0 = blockIdx.y * BN;
    const int HoWo = a.Ho * a.Wo;
    const int p0 = blockIdx.x * BM;                      // first pixel of the tile in the image
    const int p1 = min(p0 + BM, HoWo) - 1;               // last valid pixel
    const int oy0 = p0 / a.Wo, ox0 = p0 - oy0 * a.Wo;
    const int oy1 = p1 / a.Wo, ox1 = p1 - oy1 * a.Wo;
    const int s = a.stride;
    const int iy_min = oy0 * s - a.pad;
    const int WY = (oy1 - oy0) * s + a.KH;
    const int ix_min = (oy0 == oy1) ? ox0 * s - a.pad : -a.pad;
    const int WX = (oy0 == oy1) ? (ox1 - ox0) * s + a.KW : (a.Wo - 1) * s + a.KW;
    const int WXP = WX | 1;
    const float *xb = a.x + (size_t)b * 3 * a.H * a.W;

    // ---- input window -> LDS (rows are contiguous in global memory).  Loads are issued in
    // batches of 8 rows before their first use so that their latencies overlap.
    {
        const int nrows = 3 * WY;
        for (int r0 = 0; r0 < nrows; r0 += 8) {
            float v[8][2];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = r0 + u;
                const int c = r / WY, wy = r - c * WY;
                const int iy = iy_min + wy;
                const bool rok = r < nrows && iy >= 0 && iy < a.H;
                const float *row = rok ? xb + ((size_t)c * a.H + iy) * a.W : xb;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int ix = ix_min + tid + j * NT;
                    const bool ok = rok && (tid + j * NT) < WX && ix >= 0 && ix < a.W;
                    const float t = row[ok ? ix : 0];
                    v[u][j] = ok ? t : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = r0 + u;
                if (r < nrows) {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        if (tid + j * NT < WX) win[r * WXP + tid + j * NT] = v[u][j];
                }
            }
        }
    }
    // ---- weights of this N tile -> LDS (batched the same way)
    {
        const int k4n = KP >> 2;
        const int total = BN * k4n;
        for (int i0 = tid; i0 < total; i0 += 4 * NT) {
            cn_f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = min(i0 + u * NT, total - 1);
                const int k4 = i % k4n, n = i / k4n;
                const int nn = min(n0 + n, a.cout_pad - 1);  // rows past the padded Cout: never stored
                v[u] = *reinterpret_cast<const cn_f32x4 *>(a.w + (size_t)nn * KP + k4 * 4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * NT;
                if (i < total) {
                    const int k4 = i % k4n, n = i / k4n;
                    *reinterpret_cast<cn_f32x4 *>(Ws + n * LDW + k4 * 4) = v[u];
                }
            }
        }
    }
    // ---- offset tables
    for (int k = tid; k < KP; k += NT) {
        const int tap = k / 3, c = k - tap * 3;
        const int ky = tap / a.KW, kx = tap - ky * a.KW;
        kb[k] = (tap < a.KH * a.KW) ? (c * WY + ky) * WXP + kx : 0;  // padded k: weight is 0
    }
    for (int m = tid; m < BM; m += NT) {
        const int p = min(p0 + m, p1);
        const int oy = p / a.Wo, ox = p - oy * a.Wo;
        rb[m] = ((oy - oy0) * s) * WXP + (ox * s - a.pad - ix_min);
    }
    __syncthreads();

    cn_f32x16 acc[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    int rbase[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i) rbase[i] = rb[wm * TM + i * 32 + l31];
    const float *wrow = Ws + (wn * 32 + l31) * LDW + 4 * lh;
    const int *kbl = kb + 4 * lh;

#pragma unroll 4
    for (int kc = 0; kc < KP; kc += 8) {
        const cn_f32x4 bf = *reinterpret_cast<const cn_f32x4 *>(wrow + kc);
        const int k0 = kbl[kc], k1 = kbl[kc + 1], k2 = kbl[kc + 2], k3 = kbl[kc + 3];
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            const float *wp = win + rbase[i];
            const float a0 = wp[k0], a1 = wp[k1], a2 = wp[k2], a3 = wp[k3];
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bf[0], acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bf[1], acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, bf[2], acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, bf[3], acc[i], 0, 0, 0);
        }
    }

    // ---- epilogue: BN (scale/shift) + ReLU, NHWC; lanes run along Cout (128-byte rows)
    const int n = n0 + wn * 32 + l31;
    if (n < a.Cout) {
        const float sc = a.scale ? a.scale[n] : 1.f;
        const float sf = a.shift ? a.shift[n] : 0.f;
        float *yb = a.y + (size_t)b * HoWo * a.out_pitch;
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int p = p0 + m;
                if (p < HoWo) {
                    float v = acc[i][r] * sc + sf;
                    if (a.relu) v = fmaxf(v, 0.f);
                    yb[(size_t)p * a.out_pitch + n] = v;
                }
            }
    }
}

}  // namespace

// Returns CN_ERR_UNSUPPORTED when the shape does not fit this kernel (the caller then uses
// the generic implicit-GEMM stem).
int cn_stem_conv_f32(const float *x, const float *w_packed, const float *scale, const float *shift,
                     float *y, int B, int H, int W, int Ho, int Wo, int Cout, int KH, int KW,
                     int stride, int pad, int relu, int out_pitch, int KP, hipStream_t st)
{
    if (KP & 7) return CN_ERR_UNSUPPORTED;
    // worst-case window of a 128-pixel tile
    int wy, wx;
    if (Wo >= BM) {
        wy = stride + KH;  // a tile touches at most two output rows
        wx = (Wo - 1) * stride + KW;
        if (Wo % BM == 0) {  // tiles never straddle rows
            wy = KH;
            wx = (BM - 1) * stride + KW;
        }
    } else {
        const int rows = BM / Wo + 2;
        wy = (rows - 1) * stride + KH;
        wx = (Wo - 1) * stride + KW;
    }
    if ((long)3 * wy * (wx | 1) > WIN_MAX || wx > 2 * NT) return CN_ERR_UNSUPPORTED;
    StemArgs a;
    a.x = x; a.w = w_packed; a.scale = scale; a.shift = shift; a.y = y;
    a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout; a.KH = KH; a.KW = KW;
    a.stride = stride; a.pad = pad; a.relu = relu; a.out_pitch = out_pitch; a.KP = KP;
    a.tiles_per_image = cn_cdiv(Ho * Wo, BM);
    a.cout_pad = (Cout + 31) / 32 * 32;
    const int bn = Cout > 32 ? 64 : 32;
    const size_t lds = (size_t)WIN_MAX * 4 + (size_t)bn * (KP + 4) * 4 + (size_t)KP * 4 + BM * 4;
    dim3 grid(a.tiles_per_image, cn_cdiv(Cout, bn), B);
    if (bn == 64) {
        CN_SET_MAX_LDS_ONCE(stem_conv_f32_kernel<64>, 160 * 1024);
        hipLaunchKernelGGL(stem_conv_f32_kernel<64>, grid, dim3(NT), lds, st, a);
    } else {
        CN_SET_MAX_LDS_ONCE(stem_conv_f32_kernel<32>, 160 * 1024);
        hipLaunchKernelGGL(stem_conv_f32_kernel<32>, grid, dim3(NT), lds, st, a);
    }
    CN_CHECK_LAUNCH();
    return CN_OK;
}
