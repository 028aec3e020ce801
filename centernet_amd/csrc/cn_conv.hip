// cn_conv.hip -- im2col-free implicit-GEMM convolution and fused deformable
// convolution (DCNv2) on the gfx950 fp32 matrix cores.
//
// One kernel template, three A-operand producers:
//   A_DENSE : ordinary convolution, NHWC activations.  Replaces the
//             Conv2d / BatchNorm2d(eval) / ReLU / residual call sites of the
//             reference backbones (resnet_dcn.py:38-67,155-177; msra_resnet.py;
//             pose_dla_dcn.py:147-221; large_hourglass.py:17-74) and, through the
//             output-scatter arguments, ConvTranspose2d(4,2,1) (resnet_dcn.py:228-235).
//   A_STEM  : first 7x7/2 convolution reading the user's NCHW 3-channel image
//             directly (resnet_dcn.py:138-139), K packed as (tap, rgb0).
//   A_DCN   : modulated deformable convolution.  Replaces dcn_v2_cuda_forward
//             (DCNv2/src/dcn_v2_cuda.c:10-102) + modulated_deformable_im2col_gpu_kernel
//             (DCNv2/src/cuda/dcn_v2_im2col_cuda.cu:118-180) + dmcn_im2col_bilinear
//             (:18-47): offsets/mask are read once per (pixel, tap), the four
//             bilinear corners are fetched as 16-byte channel vectors (NHWC, so a
//             corner of 32 channels is one 128-byte line), combined, multiplied
//             by the mask and written straight into the LDS A tile that feeds
//             the MFMAs.  No column buffer, no per-sample host loop, bias and
//             the following BatchNorm+ReLU are the epilogue.
//
// GEMM view: D[M = B*Ho*Wo pixels][N = Cout] = A[M][K = taps*Cin] * W[K][N].
// Matrix instruction: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 64 cycles).
// LDS tiles are [rows][32 + 4] floats: the +4 pad makes the 16-lane groups of a
// ds_read_b128 hit 16 distinct 16-byte slots (row*144 B mod 256), i.e. no bank
// conflicts, and one b128 read feeds four consecutive MFMAs because K is
// consumed in the order {k, k+4}: lane half h holds k = 4h..4h+3 of each 8-group
// for both operands (any fixed K permutation is a valid dot-product order).
#include "cn_common.h"
#include <type_traits>

// f32s implicit GEMM: register sets of the tile prefetch.  2 = tiles requested two chunks ahead:
// measured SLOWER on every layer that uses this kernel (3x3/s2 0.109 -> 0.179 ms, deformable
// 0.135 -> 0.202 ms, resdcn_18 B=32 7929 -> 7399 img/s): the second set costs 50-90 registers, i.e.
// one resident workgroup per CU, and these layers live on occupancy.  Kept for A/B builds only.
#ifndef CN_IGEMM_F32S_SETS
#define CN_IGEMM_F32S_SETS 1
#endif

namespace {

constexpr int NT = 256;
constexpr int BK = 32;
constexpr int LDT = BK + 4;  // LDS row pitch in floats

// A_DCN_PAD: the deformable form for Cin % 32 != 0 (zero-fills the padded channels of the last
// chunk); CenterNet's own layers (Cin = 64..512) take A_DCN, whose gather has no select at all
enum { A_DENSE = 0, A_STEM = 1, A_DCN = 2, A_DCN_PAD = 3 };
constexpr int STEM_KMAX = 512;  // largest padded K of a 3-channel stem (11x11 -> 363 -> 384)

struct IgemmArgs {
    const void *x;       // activations, element type T (fp32 or fp16); fp32 NCHW for the stem
    const void *w;       // packed weights, element type T
    const float *bias;   // added before scale (DCN bias), may be null
    const float *scale;  // per-Cout, may be null (=1)
    const float *shift;  // per-Cout, may be null (=0)
    const void *residual;  // element type T
    void *y;               // element type T (NHWC) or fp32 (NCHW head outputs)
    const float *om;  // DCN: NHWC offsets(18) + mask(9) per pixel
    int om_pitch, mask_sigmoid;
    int B, H, W, Cin;
    int Ho, Wo, Cout;
    int KH, KW, stride, pad_h, pad_w, dil;
    int in_pitch, out_pitch;
    int OH, OW, oy_mul, oy_add, ox_mul, ox_add;
    int relu;
    int M;         // B*Ho*Wo
    int cin_pad;   // K extent per tap in the packed weight (multiple of 32)
    int cout_pad;  // rows per tap in the packed weight (multiple of 32)
    int nchunk;    // cin_pad / 32
    int KT;        // taps * nchunk
    int zparity;   // ConvTranspose 4x4/s2: blockIdx.z = output parity class (py*2+px)
    int w_zstride; // floats between the packed weights of two parity classes
    int vec_out;   // NHWC output base/pitch allow 16-byte stores
    int xcd_swizzle; // remap blockIdx.x so that each XCD owns a contiguous range of M tiles
    int setprio;     // raise the wave priority around the MFMA clusters (cn_set_tuning key 8)
    int dbgskip;     // ablation: skip A (bit 0) / B (bit 1) staging after the first chunk (WRONG results)
    int ksplit;    // split-K: blockIdx.z owns chunks [z*KT/ksplit, (z+1)*KT/ksplit)
    float *partial; // split-K: raw fp32 partial sums [ksplit][M][cout_pad]
    int in_plain, out_plain, res_plain;  // f32s kernels: x / y / residual are plain fp32 tensors
    int tile2d;                          // deformable kernel: tile rows are an 8-wide pixel BLOCK of one image
    float x_mul, res_mul;                // f32s range control (cn_f32s_ctl)
    uint32_t *range;                     // f32s: [0] max |stored output|, [1] max |split input|
};

__device__ __forceinline__ float sigmoidf_dev(float x) { return 1.0f / (1.0f + expf(-x)); }

typedef _Float16 cn_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 cn_f16x4 __attribute__((ext_vector_type(4)));

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> { static constexpr int EPV = 4; };
template <> struct ElemTraits<_Float16> { static constexpr int EPV = 8; };
template <> struct ElemTraits<cn_f32s> { static constexpr int EPV = 4; };   // 128-byte groups, as fp32

__device__ __forceinline__ cn_f32x4 load4_as_f32(const float *p) { return *reinterpret_cast<const cn_f32x4 *>(p); }
__device__ __forceinline__ cn_f32x4 load4_as_f32(const _Float16 *p)
{
    const cn_f16x4 h = *reinterpret_cast<const cn_f16x4 *>(p);
    cn_f32x4 r = {(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    return r;
}
__device__ __forceinline__ void store4_from_f32(float *p, cn_f32x4 v) { *reinterpret_cast<cn_f32x4 *>(p) = v; }
__device__ __forceinline__ void store4_from_f32(_Float16 *p, cn_f32x4 v)
{
    cn_f16x4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
    *reinterpret_cast<cn_f16x4 *>(p) = h;
}

// T = float : v_mfma_f32_32x32x2_f32, chunk = 32 channels
// T = fp16  : v_mfma_f32_32x32x16_f16 (fp32 accumulate), chunk = 64 channels.  Both use the
//             same 128-byte LDS rows and the same ds_read_b128 addresses: lane half h of a
//             32-row block reads bytes [32*kk + 16*h, +16) of its row, which is k = 4h..4h+3
//             of an 8-group in fp32 (four MFMAs) and k = 16*kk + 8h..8h+7 in fp16 (one MFMA).
template <typename T, int BM, int BN, int WM, int WN, int AMODE, bool OUT_NCHW, int NBUF>
__global__ __launch_bounds__(NT) void igemm_kernel(const IgemmArgs a)
{
    constexpr int EPV = ElemTraits<T>::EPV;  // elements per 16-byte vector
    constexpr int BKE = 8 * EPV;             // channels per chunk (one 128-byte LDS row)
    constexpr bool F16 = (EPV == 8);
    // T = cn_f32s: fp32 values as fp16 (high, low) pairs in the same 128-byte rows; three
    // v_mfma_f32_32x32x16_f16 per 16-deep K step (cn_common.h).  The deformable form reads a
    // PLAIN fp32 input (one 16-byte gather per corner), blends in fp32 and splits the blended
    // value when it writes the A tile.
    constexpr bool SPLIT = std::is_same<T, cn_f32s>::value;
    constexpr bool DCN = (AMODE == A_DCN || AMODE == A_DCN_PAD);
    static_assert(!(F16 && DCN), "the deformable kernel is fp32 / f32s only");
    static_assert(!(SPLIT && AMODE == A_STEM), "the stem reads the fp32 image: fp32 kernel");
    static_assert(NBUF == 1 || NBUF == 2, "LDS tile buffers");
    static_assert(WM * WN == NT / CN_WAVE, "4 waves");
    constexpr int TM = BM / WM, TN = BN / WN;  // wave tile
    constexpr int MB = TM / 32, NB = TN / 32;  // 32x32 MFMA blocks per wave
    static_assert(TM % 32 == 0 && TN % 32 == 0, "wave tile must be 32-aligned");
    constexpr int PA = BM / 32;  // A rows per thread per chunk
    constexpr int PB = BN / 32;  // B rows per thread per chunk

    extern __shared__ __attribute__((aligned(16))) char smem[];
    // layout: [tiles | epilogue staging] (union), then rowoff, then the DCN records
    constexpr int TILE_FLOATS = NBUF * (BM + BN) * LDT;
    constexpr int CS_FLOATS = OUT_NCHW ? 0 : (BM / WM) * (BN + 4);
    constexpr int UNION_FLOATS = TILE_FLOATS > CS_FLOATS ? TILE_FLOATS : CS_FLOATS;
    float *As = reinterpret_cast<float *>(smem);  // [NBUF][BM][LDT]
    float *Bs = As + NBUF * BM * LDT;             // [NBUF][BN][LDT]
    int *rowoff = reinterpret_cast<int *>(As + UNION_FLOATS);  // [BM]
    // DCN sampling records (4 corner byte offsets, 4 bilinear weights, mask) of every
    // (tap, tile pixel), built ONCE in the prologue by all threads: 9 * BM * 36 bytes
    // stem: k -> (plane offset, ky, kx) table, k = tap*3 + rgb  [STEM_KMAX] x 2 ints
    int *ktab = rowoff + BM;
    int *sidx = rowoff + BM;                                  // [9][BM][4] corner byte offsets
    float *swt = reinterpret_cast<float *>(sidx + 9 * BM * 4);  // [9][BM][4] corner weights
    float *smk = swt + 9 * BM * 4;                            // [9][BM]    modulation mask

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware tile order (cn_set_tuning key 7): workgroup b runs on XCD b % 8 and every XCD
    // has its own 4 MB L2, so each XCD is given a CONTIGUOUS range of pixel tiles.  Pays for
    // the deformable gather (+5-10 %); the dense kernels are MFMA-bound and the Infinity Cache
    // absorbs their re-reads (no gain measured), so there it stays off.
    int bx = blockIdx.x;
    if (a.xcd_swizzle) {
        const int q8 = gridDim.x >> 3;
        if (bx < (q8 << 3)) bx = (bx & 7) * q8 + (bx >> 3);
    }
    const int m0 = bx * BM;
    const int n0 = blockIdx.y * BN;
    const int lrow = tid >> 3;  // 0..31: row inside a 32-row pass
    const int q = tid & 7;      // float4 slot inside the 32-float chunk
    const int HoWo = a.Ho * a.Wo;
    // ConvTranspose2d(4,2,1): output parity (py,px) is a 2x2 convolution with its own
    // weights, padding (1-py, 1-px) and output offset (py,px)
    int pad_h = a.pad_h, pad_w = a.pad_w, oy_add = a.oy_add, ox_add = a.ox_add;
    const T *wbase = reinterpret_cast<const T *>(a.w);
    const T *xT = reinterpret_cast<const T *>(a.x);
    if (a.zparity) {
        const int py = blockIdx.z >> 1, px = blockIdx.z & 1;
        pad_h = 1 - py;
        pad_w = 1 - px;
        oy_add = py;
        ox_add = px;
        wbase += (size_t)blockIdx.z * a.w_zstride;
    }

    // ---- per-thread A rows: conv-grid coordinates
    int a_iy0[PA], a_ix0[PA], a_pix[PA];  // a_pix: b*H*W, or -1 if row >= M
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        const int m = m0 + p * 32 + lrow;
        if (m < a.M) {
            const int b = m / HoWo;
            const int r = m - b * HoWo;
            const int oy = r / a.Wo;
            const int ox = r - oy * a.Wo;
            a_iy0[p] = oy * a.stride - pad_h;
            a_ix0[p] = ox * a.stride - pad_w;
            a_pix[p] = b * a.H * a.W;
        } else {
            a_iy0[p] = 0;
            a_ix0[p] = 0;
            a_pix[p] = -1;
        }
    }
    // Deformable kernel, maps whose sides are multiples of the block: a tile is an 8 x (BM/8)
    // pixel BLOCK of one image instead of BM consecutive pixels of a row.  The nine taps of a
    // block sample a (8 + 2 + reach)^2 neighbourhood -- 41 KB per 32-channel chunk at reach 4 --
    // where a 64 x 1 row segment touches 66 x (3 + 2 reach): the gather's lines stay in the
    // 32 KB L1 / in L2 far better (the gather moves 16 B per sample and is L1 / L2-bandwidth
    // bound on the narrow layers: 39 B/clk/CU measured on 128->64@64^2).
    auto tile_pixel = [&](int r) -> int {   // linear pixel index (b*Ho + oy)*Wo + ox of tile row r, or -1
        if (DCN && a.tile2d) {
            constexpr int TH = BM / 8;
            const int tx_n = a.Wo >> 3, tpi = tx_n * (a.Ho / TH);
            const int b = bx / tpi, t = bx - b * tpi;
            const int ty = t / tx_n, tx = t - ty * tx_n;
            return (b * a.Ho + ty * TH + (r >> 3)) * a.Wo + tx * 8 + (r & 7);
        }
        const int m = m0 + r;
        return m < a.M ? m : -1;
    };
    // ---- output pixel index of every tile row (epilogue + residual)
    for (int r = tid; r < BM; r += NT) {
        const int m = tile_pixel(r);
        int off = -1;
        if (m >= 0) {
            const int b = m / HoWo;
            const int rr = m - b * HoWo;
            const int oy = rr / a.Wo;
            const int ox = rr - oy * a.Wo;
            off = (b * a.OH + oy * a.oy_mul + oy_add) * a.OW + ox * a.ox_mul + ox_add;
        }
        rowoff[r] = off;
    }
    // dcn_v2_im2col_cuda.cu:151-176 and :18-47, evaluated once per (pixel, tap)
    auto dcn_records = [&](int tap_lo, int tap_hi) {
        for (int i = tid; i < (tap_hi - tap_lo) * BM; i += NT) {
            const int tap = tap_lo + i / BM, r = i % BM;
            const int pb = tap;
            const int m = tile_pixel(r);
            int i0 = 0, i1 = 0, i2 = 0, i3 = 0;
            float w1 = 0.f, w2 = 0.f, w3 = 0.f, w4 = 0.f, mk = 0.f;
            if (m >= 0) {
                const int b = m / HoWo;
                const int rr = m - b * HoWo;
                const int oy = rr / a.Wo;
                const int ox = rr - oy * a.Wo;
                const float *om = a.om + (size_t)m * a.om_pitch;
                const float off_h = om[2 * tap];
                const float off_w = om[2 * tap + 1];
                mk = om[18 + tap];
                if (a.mask_sigmoid) mk = sigmoidf_dev(mk);  // dcn_v2.py:67
                // f32s: the plain input is brought to stored units (x * 2^-e, exact) through the
                // modulation factor it is multiplied with anyway
                if constexpr (SPLIT) mk *= a.x_mul;
                const int ki = tap / 3, kj = tap - ki * 3;
                const float h_im = (float)(oy - 1 + ki) + off_h;
                const float w_im = (float)(ox - 1 + kj) + off_w;
                const int H = a.H, W = a.W;
                if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
                    const float hf = floorf(h_im), wf = floorf(w_im);
                    const int h_low = (int)hf, w_low = (int)wf;
                    const int h_high = h_low + 1, w_high = w_low + 1;
                    const float lh = h_im - hf, lw = w_im - wf;
                    const float hh = 1.f - lh, hw = 1.f - lw;
                    const bool hl_ok = h_low >= 0, wl_ok = w_low >= 0;
                    const bool hh_ok = h_high <= H - 1, wh_ok = w_high <= W - 1;
                    w1 = (hl_ok && wl_ok) ? hh * hw : 0.f;
                    w2 = (hl_ok && wh_ok) ? hh * lw : 0.f;
                    w3 = (hh_ok && wl_ok) ? lh * hw : 0.f;
                    w4 = (hh_ok && wh_ok) ? lh * lw : 0.f;
                    const int yl = max(h_low, 0), yh = min(h_high, H - 1);
                    const int xl = max(w_low, 0), xh = min(w_high, W - 1);
                    // byte offsets of the four corner pixels (fit 32 bits: checked by the host),
                    // so the gather below is `global_load base(SGPR) + offset(VGPR)`
                    const int base = b * H * W;
                    const int pb4 = a.in_pitch * (int)sizeof(float);
                    i0 = (base + yl * W + xl) * pb4;
                    i1 = (base + yl * W + xh) * pb4;
                    i2 = (base + yh * W + xl) * pb4;
                    i3 = (base + yh * W + xh) * pb4;
                }
            }
            int *si = sidx + (pb * BM + r) * 4;
            float *sw = swt + (pb * BM + r) * 4;
            si[0] = i0; si[1] = i1; si[2] = i2; si[3] = i3;
            sw[0] = w1; sw[1] = w2; sw[2] = w3; sw[3] = w4;
            smk[pb * BM + r] = mk;
        }
    };
    // split-K: blockIdx.z owns chunks [kt0, kt1); for the deformable kernel the split is on tap
    // boundaries (ksplit divides 9), so a workgroup owns taps [tap0, tap1)
    int kt0 = 0, kt1 = a.KT;
    if (a.ksplit > 1) {  // small-M / deep-K layers: several workgroups share one output tile
        kt0 = (int)((long)blockIdx.z * a.KT / a.ksplit);
        kt1 = (int)((long)(blockIdx.z + 1) * a.KT / a.ksplit);
    }
    const int tap0 = kt0 / a.nchunk, tap1 = kt1 / a.nchunk;
    if (DCN) dcn_records(tap0, tap1);
    if (AMODE == A_STEM) {
        const int taps = a.KH * a.KW;
        for (int k = tid; k < a.cin_pad; k += NT) {
            const int tap = k / 3, c = k - tap * 3;
            const int ky = tap / a.KW, kx = tap - ky * a.KW;
            const bool ok = tap < taps;
            ktab[2 * k] = ok ? (c * a.H * a.W + ky * a.dil * a.W + kx * a.dil) : 0;
            ktab[2 * k + 1] = ok ? ((ky * a.dil) << 16 | (kx * a.dil)) : 0x7fff7fff;  // fails the bounds test
        }
    }
    __syncthreads();

    cn_f32x16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    constexpr int NCORN = DCN ? 4 : 1;
    // NSET = 2 (A/B builds, see CN_IGEMM_F32S_SETS): the tiles of chunk k + 2 are requested while
    // chunk k is multiplied
    constexpr int NSET = SPLIT ? CN_IGEMM_F32S_SETS : 1;
    cn_f32x4 ra_[NSET][PA][NCORN];
    cn_f32x4 rb_[NSET][PB];
    float rng_in = 0.f, rng_out = 0.f;   // largest |value| split on the input / output side
    const cn_f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    auto load_tiles = [&](int kt, auto SET) {
        auto &ra = ra_[decltype(SET)::value];
        auto &rb = rb_[decltype(SET)::value];
        const int tap = kt / a.nchunk;
        const int c0 = (kt - tap * a.nchunk) * BKE;
        // ---- B: packed weight [tap][cout_pad][cin_pad]
        if (!(a.dbgskip & 2))
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            // rows past cout_pad are clamped: their columns are never stored
            const int n = min(n0 + p * 32 + lrow, a.cout_pad - 1);
            rb[p] = *reinterpret_cast<const cn_f32x4 *>(
                wbase + ((size_t)(tap * a.cout_pad + n) * a.cin_pad + c0 + EPV * q));
        }
        // ---- A
        if (a.dbgskip & 1) return;
        if constexpr (AMODE == A_DENSE) {
            const int ky = tap / a.KW, kx = tap - ky * a.KW;
            const int c = c0 + EPV * q;
#pragma unroll
            for (int p = 0; p < PA; ++p) {
                const int iy = a_iy0[p] + ky * a.dil;
                const int ix = a_ix0[p] + kx * a.dil;
                const bool ok = a_pix[p] >= 0 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W &&
                                ((SPLIT && !a.in_plain) ? (c0 < a.Cin) : (c < a.Cin));
                // always load from a valid address, then select: no exec-mask branches
                const size_t off = ok ? ((size_t)(a_pix[p] + iy * a.W + ix) * a.in_pitch + c) : 0;
                const cn_f32x4 v = *reinterpret_cast<const cn_f32x4 *>(xT + off);
                ra[p][0] = ok ? v : zero4;
            }
        } else if constexpr (AMODE == A_STEM) {
            // K is packed densely as k = tap*3 + rgb (147 -> 160 for 7x7, not 196 -> 224);
            // the image is fp32 NCHW; (offset, ky, kx) of every k come from the LDS table
            const float *xin = reinterpret_cast<const float *>(a.x);
            const int kbase = (kt * 8 + q) * EPV;
#pragma unroll
            for (int p = 0; p < PA; ++p) {
                T vals[EPV];
                const size_t img = (size_t)(a_pix[p] >= 0 ? a_pix[p] : 0) * 3 +
                                   (size_t)a_iy0[p] * a.W + a_ix0[p];  // may point before the
                                                                       // image: only used when ok
#pragma unroll
                for (int e = 0; e < EPV; ++e) {
                    const int off = ktab[2 * (kbase + e)];
                    const int kk = ktab[2 * (kbase + e) + 1];
                    const int iy = a_iy0[p] + (kk >> 16), ix = a_ix0[p] + (kk & 0xffff);
                    const bool ok = a_pix[p] >= 0 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
                    const float v = xin[ok ? (img + off) : 0];
                    vals[e] = ok ? (T)v : (T)0.f;
                }
                ra[p][0] = *reinterpret_cast<const cn_f32x4 *>(vals);
            }
        } else {  // A_DCN: four bilinear corners, each a 16-byte channel vector
            const int c = c0 + EPV * q;
            const char *xb = reinterpret_cast<const char *>(a.x);
            const bool cok = c < a.Cin;  // false only in the padded tail of a Cin % 32 != 0 layer
            const unsigned cbyte = (unsigned)(cok ? c : 0) * (unsigned)sizeof(float);
#pragma unroll
            for (int p = 0; p < PA; ++p) {
                const int r = p * 32 + lrow;
                const cn_i32x4 si = *reinterpret_cast<const cn_i32x4 *>(sidx + (tap * BM + r) * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    ra[p][j] = *reinterpret_cast<const cn_f32x4 *>(xb + ((unsigned)si[j] + cbyte));
            }
            if (AMODE == A_DCN_PAD) {
#pragma unroll
                for (int p = 0; p < PA; ++p)
#pragma unroll
                    for (int j = 0; j < 4; ++j) ra[p][j] = cok ? ra[p][j] : zero4;
            }
        }
    };

    auto store_tiles = [&](int buf, int kt, auto SET) {
        auto &ra = ra_[decltype(SET)::value];
        auto &rb = rb_[decltype(SET)::value];
        float *Ad = As + buf * BM * LDT;
        float *Bd = Bs + buf * BN * LDT;
        if (!(a.dbgskip & 2))
#pragma unroll
        for (int p = 0; p < PB; ++p)
            *reinterpret_cast<cn_f32x4 *>(Bd + (p * 32 + lrow) * LDT + 4 * q) = rb[p];
        if (a.dbgskip & 1) return;
        if (DCN) {
            const int tap = kt / a.nchunk;
#pragma unroll
            for (int p = 0; p < PA; ++p) {
                const int r = p * 32 + lrow;
                const float *wt = swt + (tap * BM + r) * 4;
                const float mk = smk[tap * BM + r];
                const float w1 = wt[0], w2 = wt[1], w3 = wt[2], w4 = wt[3];
                cn_f32x4 v;
                // (w1*v1 + w2*v2 + w3*v3 + w4*v4) * mask   (dcn_v2_im2col_cuda.cu:43-45,174)
                v = ra[p][0] * w1 + ra[p][1] * w2 + ra[p][2] * w3 + ra[p][3] * w4;
                v = v * mk;
                if constexpr (SPLIT) {
                    cn_f16x4v hi, lo;
                    cn_rng_upd4(rng_in, v);
                    cn_split4(v, hi, lo);
                    char *row = reinterpret_cast<char *>(Ad + r * LDT);
                    *reinterpret_cast<cn_f16x4v *>(row + 8 * q) = hi;
                    *reinterpret_cast<cn_f16x4v *>(row + 64 + 8 * q) = lo;
                } else {
                    *reinterpret_cast<cn_f32x4 *>(Ad + r * LDT + 4 * q) = v;
                }
            }
        } else {
            if constexpr (SPLIT) {
                if (a.in_plain) {  // plain fp32 input: channels 4q..4q+3 split while staging
#pragma unroll
                    for (int p = 0; p < PA; ++p) {
                        cn_f16x4v hi, lo;
                        const cn_f32x4 xs = ra[p][0] * a.x_mul;   // real -> stored units
                        cn_rng_upd4(rng_in, xs);
                        cn_split4(xs, hi, lo);
                        char *row = reinterpret_cast<char *>(Ad + (p * 32 + lrow) * LDT);
                        *reinterpret_cast<cn_f16x4v *>(row + 8 * q) = hi;
                        *reinterpret_cast<cn_f16x4v *>(row + 64 + 8 * q) = lo;
                    }
                    return;
                }
            }
#pragma unroll
            for (int p = 0; p < PA; ++p)
                *reinterpret_cast<cn_f32x4 *>(Ad + (p * 32 + lrow) * LDT + 4 * q) = ra[p][0];
        }
    };

    const int l31 = lane & 31, lh = lane >> 5;
    auto compute = [&](int buf) {
        const float *Ab = As + buf * BM * LDT + (wm * TM + l31) * LDT + 4 * lh;
        const float *Bb = Bs + buf * BN * LDT + (wn * TN + l31) * LDT + 4 * lh;
        if (a.setprio) __builtin_amdgcn_s_setprio(1);
        if constexpr (SPLIT) {
            // row quarters: high parts k 0-15, 16-31, low parts k 0-15, 16-31
            cn_f16x8 af[4][MB], bf[4][NB];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                for (int i = 0; i < MB; ++i)
                    af[kk][i] = *reinterpret_cast<const cn_f16x8 *>(Ab + i * 32 * LDT + kk * 8);
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    bf[kk][j] = *reinterpret_cast<const cn_f16x8 *>(Bb + j * 32 * LDT + kk * 8);
            }
            // All twelve fragments are in registers BEFORE the first MFMA issues, and the compiler
            // may not sink a read back into the MFMA stream: an MFMA that waits in the matrix pipe
            // for its accumulator (dependent chain, pipe shared with co-resident waves) reads its
            // A/B registers late, and a ds_read issued right behind it into the same registers
            // (hipcc re-used them to save VGPRs) overwrote the operand first -- rare wrong rows
            // that came and went with occupancy (measured: tools/dbg_determinism.py).
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int term = 0; term < 3; ++term)   // lo*hi, hi*lo, hi*hi
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int i = 0; i < MB; ++i)
#pragma unroll
                        for (int j = 0; j < NB; ++j) {
                            const int ka = (term == 0) ? 2 + s2 : s2;
                            const int kb = (term == 1) ? 2 + s2 : s2;
                            if (OUT_NCHW)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                                    bf[kb][j], af[ka][i], acc[i][j], 0, 0, 0);
                            else
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                                    af[ka][i], bf[kb][j], acc[i][j], 0, 0, 0);
                        }
            if (a.setprio) __builtin_amdgcn_s_setprio(0);
            return;
        }
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            cn_f32x4 af[MB], bf[NB];
#pragma unroll
            for (int i = 0; i < MB; ++i)
                af[i] = *reinterpret_cast<const cn_f32x4 *>(Ab + i * 32 * LDT + kk * 8);
#pragma unroll
            for (int j = 0; j < NB; ++j)
                bf[j] = *reinterpret_cast<const cn_f32x4 *>(Bb + j * 32 * LDT + kk * 8);
            if constexpr (F16) {
#pragma unroll
                for (int i = 0; i < MB; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        const cn_f16x8 a8 = __builtin_bit_cast(cn_f16x8, af[i]);
                        const cn_f16x8 b8 = __builtin_bit_cast(cn_f16x8, bf[j]);
                        if (OUT_NCHW)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b8, a8, acc[i][j], 0, 0, 0);
                        else
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, acc[i][j], 0, 0, 0);
                    }
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < MB; ++i)
#pragma unroll
                        for (int j = 0; j < NB; ++j) {
                            if (OUT_NCHW)  // D rows = cout, cols = pixels
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                    bf[j][s], af[i][s], acc[i][j], 0, 0, 0);
                            else  // D rows = pixels, cols = cout
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                    af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
                        }
            }
        }
        if (a.setprio) __builtin_amdgcn_s_setprio(0);
    };

    // ---- main loop: register prefetch of chunk k+1 during the MFMAs of chunk k.
    // NBUF=2: LDS double buffer, one barrier per chunk.  NBUF=1: half the LDS (more
    // workgroups per CU hide the second barrier), two barriers per chunk.
    if constexpr (NSET == 2) {
        constexpr std::integral_constant<int, 0> S0{};
        constexpr std::integral_constant<int, 1> S1{};
        load_tiles(kt0, S0);
        if (kt0 + 1 < kt1) load_tiles(kt0 + 1, S1);
        store_tiles(0, kt0, S0);
        __syncthreads();
        // chunk kt lives in register set / LDS buffer (kt - kt0) & 1: static inside a pair
        auto step = [&](int kt, auto SET) {
            constexpr int sv = decltype(SET)::value;
            constexpr std::integral_constant<int, sv ^ 1> OTHER{};
            if (kt + 2 < kt1) load_tiles(kt + 2, SET);   // this set's chunk went to LDS a step ago
            const bool more = (kt + 1) < kt1;
            if (NBUF == 2) {
                compute(sv);
                if (more) store_tiles(sv ^ 1, kt + 1, OTHER);
                __syncthreads();
            } else {
                compute(0);
                __syncthreads();
                if (more) store_tiles(0, kt + 1, OTHER);
                __syncthreads();
            }
        };
#pragma unroll 1
        for (int kt = kt0; kt < kt1; kt += 2) {
            step(kt, S0);
            if (kt + 1 < kt1) step(kt + 1, S1);
        }
    } else {
    constexpr std::integral_constant<int, 0> SZ{};
    load_tiles(kt0, SZ);
    store_tiles(0, kt0, SZ);
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
        const bool more = (kt + 1) < kt1 && !(a.dbgskip == 3);
        if (more) load_tiles(kt + 1, SZ);
        if (NBUF == 2) {
            const int buf = (kt - kt0) & 1;
            compute(buf);
            if (more) store_tiles(buf ^ 1, kt + 1, SZ);
            __syncthreads();
        } else {
            compute(0);
            __syncthreads();
            if (more) store_tiles(0, kt + 1, SZ);
            __syncthreads();
        }
    }
    }

    // ---- epilogue: y = relu?((acc + bias) * scale + shift + residual)
    // C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    if (!OUT_NCHW) {
        // Stage the tile through LDS (As/Bs are free after the loop's last barrier) so that
        // residual loads and output stores are 16-byte, row-contiguous accesses.  One pass
        // per wave-row (TM rows) keeps the staging buffer at TM x (BN+4) floats.
        constexpr int LDC = BN + 4;
        float *Cs = reinterpret_cast<float *>(smem);
        constexpr int C4 = BN / 4;    // float4 columns per tile row
        constexpr int RPI = NT / C4;  // tile rows covered per pass of the block
        constexpr int ITERS = (TM + RPI - 1) / RPI;
        const int c4 = tid % C4, r0 = tid / C4;
        const int n = n0 + c4 * 4;
        float bs[4], sc[4], sf[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool ok = (n + e) < a.Cout;
            bs[e] = (a.bias && ok) ? a.bias[n + e] : 0.f;
            sc[e] = (a.scale && ok) ? a.scale[n + e] : 1.f;
            sf[e] = (a.shift && ok) ? a.shift[n + e] : 0.f;
        }
        const bool vec = a.vec_out && (n + 4 <= a.Cout);
        const bool raw = a.ksplit > 1;  // partial sums: no epilogue, fp32, row index = m
#pragma unroll 1
        for (int pass = 0; pass < WM; ++pass) {
            if (pass) __syncthreads();  // previous pass fully read
            if (wm == pass) {
#pragma unroll
                for (int i = 0; i < MB; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                            Cs[row * LDC + wn * TN + j * 32 + l31] = acc[i][j][r];
                        }
            }
            __syncthreads();
            const int rbase = pass * TM;  // tile row of staging row 0
            if (raw) {
                float *pz = a.partial + (size_t)blockIdx.z * a.M * a.cout_pad;
#pragma unroll
                for (int it = 0; it < ITERS; ++it) {
                    const int lr = it * RPI + r0;
                    // partial rows are indexed by the OUTPUT pixel (split-K layers write y in
                    // conv-grid order: rowoff == the linear pixel index, whatever the tile shape)
                    const int m = lr < TM ? rowoff[rbase + lr] : -1;
                    if (m >= 0 && n < a.cout_pad)
                        *reinterpret_cast<cn_f32x4 *>(pz + (size_t)m * a.cout_pad + n) =
                            *reinterpret_cast<const cn_f32x4 *>(Cs + lr * LDC + c4 * 4);
                }
            } else if (vec) {
                cn_f32x4 res[ITERS];
                int offs[ITERS];
#pragma unroll
                for (int it = 0; it < ITERS; ++it) {
                    const int lr = it * RPI + r0;
                    offs[it] = (lr < TM) ? rowoff[rbase + lr] : -1;
                    if (a.residual) {
                        const size_t po = (size_t)(offs[it] >= 0 ? offs[it] : 0);
                        const size_t o = po * a.out_pitch + n;
                        if constexpr (SPLIT)
                            res[it] = a.res_plain
                                          ? load4_as_f32(reinterpret_cast<const float *>(a.residual) + o)
                                          : cn_load4_f32s(a.residual, po, a.out_pitch, n);
                        else
                            res[it] = load4_as_f32(reinterpret_cast<const T *>(a.residual) + o);
                    }
                }
#pragma unroll
                for (int it = 0; it < ITERS; ++it) {
                    if (offs[it] < 0) continue;
                    cn_f32x4 v =
                        *reinterpret_cast<const cn_f32x4 *>(Cs + (it * RPI + r0) * LDC + c4 * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = (v[e] + bs[e]) * sc[e] + sf[e];
                        if (a.residual) {
                            if constexpr (SPLIT) t = fmaf(res[it][e], a.res_mul, t);
                            else t += res[it][e];
                        }
                        v[e] = a.relu ? fmaxf(t, 0.f) : t;
                    }
                    if constexpr (SPLIT) {
                        if (a.out_plain)
                            store4_from_f32(reinterpret_cast<float *>(a.y) +
                                            (size_t)offs[it] * a.out_pitch + n, v);
                        else {
                            cn_rng_upd4(rng_out, v);
                            cn_store4_f32s(a.y, (size_t)offs[it], a.out_pitch, n, v);
                        }
                    } else {
                        store4_from_f32(reinterpret_cast<T *>(a.y) + (size_t)offs[it] * a.out_pitch + n, v);
                    }
                }
            } else if (n < a.Cout) {
                for (int it = 0; it < ITERS; ++it) {
                    const int lr = it * RPI + r0;
                    if (lr >= TM) continue;
                    const int off = rowoff[rbase + lr];
                    if (off < 0) continue;
                    for (int e = 0; e < 4 && (n + e) < a.Cout; ++e) {
                        const size_t o = (size_t)off * a.out_pitch + n + e;
                        float t = (Cs[lr * LDC + c4 * 4 + e] + bs[e]) * sc[e] + sf[e];
                        if constexpr (SPLIT) {
                            if (a.residual)
                                t = fmaf(a.res_plain ? reinterpret_cast<const float *>(a.residual)[o]
                                                     : cn_load1_f32s(a.residual, (size_t)off, a.out_pitch, n + e),
                                         a.res_mul, t);
                            t = a.relu ? fmaxf(t, 0.f) : t;
                            if (a.out_plain)
                                reinterpret_cast<float *>(a.y)[o] = t;
                            else {
                                cn_rng_upd1(rng_out, t);
                                cn_store1_f32s(a.y, (size_t)off, a.out_pitch, n + e, t);
                            }
                        } else {
                            if (a.residual) t += (float)reinterpret_cast<const T *>(a.residual)[o];
                            reinterpret_cast<T *>(a.y)[o] = (T)(a.relu ? fmaxf(t, 0.f) : t);
                        }
                    }
                }
            }
        }
    } else {
        const int OHW = a.OH * a.OW;
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            const int off = rowoff[wm * TM + i * 32 + l31];  // this lane's pixel
            const int b = off >= 0 ? off / OHW : 0;
            const int rem = off - b * OHW;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = n0 + wn * TN + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (off >= 0 && n < a.Cout) {
                        const float bs = a.bias ? a.bias[n] : 0.f;
                        const float sc = a.scale ? a.scale[n] : 1.f;
                        const float sf = a.shift ? a.shift[n] : 0.f;
                        float v = (acc[i][j][r] + bs) * sc + sf;
                        if (a.relu) v = fmaxf(v, 0.f);
                        reinterpret_cast<float *>(a.y)[((size_t)b * a.Cout + n) * OHW + rem] = v;
                    }
                }
            }
        }
    }
    if constexpr (SPLIT) {
        if (a.range) {   // uniform; the whole workgroup reaches this point
            if (!OUT_NCHW && !a.out_plain && a.ksplit <= 1) cn_rng_commit(a.range, 0, rng_out);
            if (DCN || a.in_plain) cn_rng_commit(a.range, 1, rng_in);
        }
    }
}

template <int BM, int BN, int WM, int AMODE, bool OUT_NCHW, int NBUF>
constexpr size_t igemm_lds_bytes()
{
    constexpr size_t tiles = (size_t)NBUF * (BM + BN) * LDT;
    constexpr size_t cs = OUT_NCHW ? 0 : (size_t)(BM / WM) * (BN + 4);
    return (tiles > cs ? tiles : cs) * 4 + BM * 4 +
           ((AMODE == A_DCN || AMODE == A_DCN_PAD) ? (size_t)9 * BM * (4 * 4 + 4 * 4 + 4) : 0) +
           (AMODE == A_STEM ? (size_t)STEM_KMAX * 8 : 0);
}

int g_tune_setprio = 1; // cn_set_tuning key 8: s_setprio(1) around the MFMA clusters (+0.9 % measured)
int g_tune_dbgskip = 0; // cn_set_tuning key 9 (ablation only): bit0 skip A staging, bit1 skip B staging
int g_tune_swz = 0;   // cn_set_tuning key 7: XCD-aware tile order, 0 = deformable kernel only (default), 1 = all, 2 = none
int g_tune_nbuf = 0;  // 0 = per-shape default, 1 / 2 = force (cn_set_tuning key 1)
int g_tune_narrow = 0; // cn_set_tuning key 2: 0 = default, 1 = never prefer 64-wide tiles
int g_tune_bm = 0;       // cn_set_tuning key 4: 0 = default, 64 / 128 = force the dense pixel tile
int g_tune_nosplit = 0;  // cn_set_tuning key 5: 1 = never split K
int g_tune_stem_persist = 1; // cn_set_tuning key 12: persistent, prefetching stem kernel (cn_stem.hip)
int g_tune_dcn_split = 0;   // cn_set_tuning key 13: 0 = auto, 1 = never, 3 / 9 = force tap split of the deformable kernel
int g_tune_bm256 = 0;       // cn_set_tuning key 14: 1 = 256-pixel tiles for 64-wide layers in the halo kernel (no gain, measured)
int g_tune_waves8 = 1;      // cn_set_tuning key 15: 8-wave workgroups for the 128-wide halo tiles
int g_tune_occ4 = 0;        // cn_set_tuning key 19: 4-workgroups-per-CU form of the 64-wide halo tiles: 0 = by rounds rule, 1 = always, 2 = never
int g_tune_dcn_form = 0;   // cn_set_tuning key 23: f32s deformable kernel, 0 = by shape and grid (team form per key 36, else the register-sampling window form, else the gather form), 1 = global-gather form always, 2 = register-sampling form (cn_dcn2.hip) for every shape it takes, 4 / 5 = team form (cn_dcn3.hip) in T / N mode for every shape it takes, 6 / 7 = wide form (cn_dcn4.hip; 7: four blocks per workgroup) for every shape it takes
int g_tune_stem16s = 1;     // cn_set_tuning key 27: f32s form of the stride-1 16-channel stem (DLA base_layer); 0 = fp32 kernel
int g_tune_dcn_tile2d = 1; // cn_set_tuning key 22: deformable kernel, 1 = 8-wide pixel blocks as tiles (default), 0 = row segments
int g_tune_nohalo = 0;   // cn_set_tuning key 10: 1 = generic implicit GEMM for 3x3/s1 instead of cn_conv3x3.hip
int g_tune_nostem = 0;   // cn_set_tuning key 6: 1 = generic implicit-GEMM stem instead of cn_stem.hip

template <typename T, int BM, int BN, int WM, int WN, int AMODE, bool OUT_NCHW, int NBUF>
int launch_igemm_n(const IgemmArgs &a, hipStream_t st)
{
    constexpr size_t lds = igemm_lds_bytes<BM, BN, WM, AMODE, OUT_NCHW, NBUF>();
    CN_SET_MAX_LDS_ONCE((igemm_kernel<T, BM, BN, WM, WN, AMODE, OUT_NCHW, NBUF>), lds);
    dim3 grid(cn_cdiv(a.M, BM), cn_cdiv(a.Cout, BN), a.zparity ? 4 : (a.ksplit > 1 ? a.ksplit : 1));
    IgemmArgs b = a;
    // XCD-aware tile order: on for the deformable kernel (its gather re-reads every input line
    // ~36x; with round-robin tile -> XCD placement only 31 % of those hit the 4 MB L2, measured
    // TCC_HIT/TCC_MISS; contiguous tile ranges per XCD: +5-10 %, tools/bench_dcn.py SWZ=1);
    // cn_set_tuning key 7: 0 = default, 1 = also the dense kernels, 2 = nowhere
    b.xcd_swizzle = (grid.x >= 16 && g_tune_swz != 2 && (AMODE == A_DCN || AMODE == A_DCN_PAD || g_tune_swz == 1)) ? 1 : 0;
    b.setprio = g_tune_setprio;
    b.dbgskip = g_tune_dbgskip;
    hipLaunchKernelGGL((igemm_kernel<T, BM, BN, WM, WN, AMODE, OUT_NCHW, NBUF>), grid, dim3(NT),
                       lds, st, b);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

// fp16 activations/weights (fp32 accumulate): dense + stem forms only
template <int BM, int BN, int WM, int WN, int AMODE, bool OUT_NCHW>
int launch_igemm_h(const IgemmArgs &a, hipStream_t st)
{
    static_assert(AMODE != A_DCN && AMODE != A_DCN_PAD, "fp16 DCN is not built");
    if (!OUT_NCHW && AMODE == A_DENSE && a.stride == 1)
        return launch_igemm_n<_Float16, BM, BN, WM, WN, AMODE, OUT_NCHW, 1>(a, st);
    return launch_igemm_n<_Float16, BM, BN, WM, WN, AMODE, OUT_NCHW, 2>(a, st);
}

// f32s (fp16 high/low pairs, fp32 accumulate): dense and deformable forms
template <int BM, int BN, int WM, int WN, int AMODE, bool OUT_NCHW>
int launch_igemm_s(const IgemmArgs &a, hipStream_t st)
{
    static_assert(AMODE != A_STEM, "the stem stays on the fp32 kernel");
    if (!OUT_NCHW && a.stride == 1 && g_tune_nbuf != 2)
        return launch_igemm_n<cn_f32s, BM, BN, WM, WN, AMODE, OUT_NCHW, 1>(a, st);
    return launch_igemm_n<cn_f32s, BM, BN, WM, WN, AMODE, OUT_NCHW, 2>(a, st);
}

template <int BM, int BN, int WM, int WN, int AMODE, bool OUT_NCHW>
int launch_igemm(const IgemmArgs &a, hipStream_t st)
{
    // measured on MI355X (tools/bench_kernels.py, profiles/): single-buffered LDS (more
    // workgroups per CU) wins for stride-1 layers, double-buffered for strided gathers
    int nbuf = g_tune_nbuf ? g_tune_nbuf : (a.stride == 1 ? 1 : 2);
    if (AMODE == A_STEM || OUT_NCHW) nbuf = 2;  // only the NHWC dense / DCN kernels carry both forms
    if (nbuf == 1) {
        if constexpr (AMODE != A_STEM && !OUT_NCHW)
            return launch_igemm_n<float, BM, BN, WM, WN, AMODE, OUT_NCHW, 1>(a, st);
    }
    return launch_igemm_n<float, BM, BN, WM, WN, AMODE, OUT_NCHW, 2>(a, st);
}

// ---- weight packing: (Cout,Cin,KH,KW) -> [tap][cout_pad][cin_pad], zero padded
template <typename T>
__global__ void pack_weight_kernel(const float *__restrict__ w, T *__restrict__ wp, int Cout,
                                   int Cin, int taps, int cout_pad, int cin_pad)
{
    const size_t total = (size_t)taps * cout_pad * cin_pad;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cin_pad);
        const int n = (int)((i / cin_pad) % cout_pad);
        const int t = (int)(i / ((size_t)cin_pad * cout_pad));
        float v = 0.f;
        if (c < Cin && n < Cout) v = w[((size_t)n * Cin + c) * taps + t];
        wp[i] = (T)v;
    }
}
// f32s form of the same layout: every 32-channel group of a row is 32 fp16 high parts followed
// by 32 fp16 low parts (cn_common.h); `wp` is addressed in fp16 units (2 per packed float)
__global__ void pack_weight_f32s_kernel(const float *__restrict__ w, _Float16 *__restrict__ wp,
                                        int Cout, int Cin, int taps, int cout_pad, int cin_pad)
{
    const size_t total = (size_t)taps * cout_pad * cin_pad;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cin_pad);
        const int n = (int)((i / cin_pad) % cout_pad);
        const int t = (int)(i / ((size_t)cin_pad * cout_pad));
        float v = 0.f;
        if (c < Cin && n < Cout) v = w[((size_t)n * Cin + c) * taps + t];
        const _Float16 hi = (_Float16)v;
        const size_t g = (i - (size_t)(c & 31)) * 2;   // first fp16 of the 32-channel group
        wp[g + (c & 31)] = hi;
        wp[g + 32 + (c & 31)] = (_Float16)(v - (float)hi);
    }
}
// Fragment-ordered f32s copy for the kernels that stream weights straight into MFMA operand registers
// (cn_dcn2.hip / cn_dcn3.hip; the register-streamed form of the LDS-halo kernel, cn_conv3x3.hip NBUFB = 0):
// [tap][chunk][cout block of 32][quarter 4][lane 64][8 fp16] -- a wave's load of one quarter is 1 KiB of
// CONTIGUOUS memory, eight whole cache lines (round 5; the round-2 order [lane][quarter] made every load
// instruction touch 32 lines for a quarter of their bytes, and the deformable kernels are bound by this
// stream through the vector L1).
// Lane (l31 = lane & 31, h = lane >> 5) of a 32 x 32 x 16 MFMA holds, for quarter kk, the
// channels 16 * (kk & 1) + 8 * h .. + 7 of output channel 32 * block + l31 -- high parts for
// kk < 2, low parts for kk >= 2 -- i.e. exactly what it would ds_read_b128 from the row form.
__global__ void pack_weight_f32s_frag_kernel(const float *__restrict__ w, _Float16 *__restrict__ wf,
                                             int Cout, int Cin, int taps, int ncb, int nchunk)
{
    const size_t total = (size_t)taps * nchunk * ncb * 64 * 32;   // fp16 elements
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(i & 7);
        const int lane = (int)((i >> 3) & 63);
        const int kk = (int)((i >> 9) & 3);
        size_t r = i >> 11;
        const int nb = (int)(r % ncb); r /= ncb;
        const int chunk = (int)(r % nchunk);
        const int t = (int)(r / nchunk);
        const int n = nb * 32 + (lane & 31);
        const int c = chunk * 32 + 16 * (kk & 1) + 8 * (lane >> 5) + e;
        float v = 0.f;
        if (c < Cin && n < Cout) v = w[((size_t)n * Cin + c) * taps + t];
        const _Float16 hi = (_Float16)v;
        wf[i] = (kk < 2) ? hi : (_Float16)(v - (float)hi);
    }
}
// stem: (Cout,3,KH,KW) -> [cout_pad][kpad], k = tap*3 + rgb
template <typename T>
__global__ void pack_stem_weight_kernel(const float *__restrict__ w, T *__restrict__ wp,
                                        int Cout, int taps, int cout_pad, int kpad)
{
    const size_t total = (size_t)cout_pad * kpad;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % kpad);
        const int n = (int)(i / kpad);
        const int t = k / 3, c = k - t * 3;
        float v = 0.f;
        if (n < Cout && t < taps) v = w[((size_t)n * 3 + c) * taps + t];
        wp[i] = (T)v;
    }
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
}  // namespace
int cn_stem_conv_f32(const float *x, const float *w_packed, const float *scale, const float *shift,
                     float *y, int B, int H, int W, int Ho, int Wo, int Cout, int KH, int KW,
                     int stride, int pad, int relu, int out_pitch, int KP, int persistent,
                     const cn_f32s_ctl *ctl, hipStream_t st);
int cn_stem_pool_rows(int B, int Ho, int Wo, int Cout, int KH, int KW, int stride, int KP);
int cn_stem_pool_f32s(const float *x, const float *w_packed, const float *scale, const float *shift,
                      float *y, int B, int H, int W, int Ho, int Wo, int Cout, int KH, int KW,
                      int stride, int pad, int relu, int out_pitch, int KP, int y_f32s, const cn_f32s_ctl *ctl,
                      hipStream_t st);
int cn_dcn_window_f32s(const float *x, const void *w_packed, const float *bias, const float *om,
                       int om_pitch, const float *scale, const float *shift, void *y, int out_pitch,
                       int out_plain, int B, int Cin, int H, int W, int Cout, int mask_sigmoid, int relu,
                       float x_mul, uint32_t *range, int min_wgs, int dbg, float *partial,
                       size_t partial_bytes, int *ksplit_out, hipStream_t st);
bool cn_offconv_takes(int B, int H, int W, int Cin, int Cout, int in_pitch, int out_pitch, int ksplit);
int cn_offconv_f32s(const float *x, const void *w_packed, const float *scale, const float *shift, float *y,
                    int B, int H, int W, int Cin, int Cout, int out_pitch, int relu, const cn_f32s_ctl *ctl,
                    int ksplit, float *partial, hipStream_t st);
extern int cn_tune_offconv, cn_tune_offconv_teams1;   // cn_offconv.hip
int cn_dcn_team_f32s(const float *x, const void *w_packed, const float *bias, const float *om,
                     int om_pitch, const float *scale, const float *shift, void *y, int out_pitch,
                     int out_plain, int B, int Cin, int H, int W, int Cout, int mask_sigmoid, int relu,
                     float x_mul, uint32_t *range, int nmode, int dbg, float *partial,
                     size_t partial_bytes, int *ksplit_out, hipStream_t st);
extern int cn_tune_dcn_team, cn_tune_dcn_team_wgs, cn_tune_dcn_team_stagger;   // cn_dcn3.hip
int cn_dcn_wide_f32s(const float *x, const void *w_packed, const float *bias, const float *om,
                     int om_pitch, const float *scale, const float *shift, void *y, int out_pitch,
                     int out_plain, int B, int Cin, int H, int W, int Cout, int mask_sigmoid, int relu,
                     float x_mul, uint32_t *range, int nb, int dbg, float *partial, size_t partial_bytes,
                     int *ksplit_out, hipStream_t st);
extern int cn_tune_dcn_wide, cn_tune_dcn_wide_wgs, cn_tune_dcn_wide_prefetch;   // cn_dcn4.hip
bool cn_proj1x1_takes(int B, int H, int W, int Cin, int Cout, int stride, int in_pitch, int out_pitch);
int cn_proj1x1_f32s(const void *x, const void *w_packed, const float *scale, const float *shift, void *y, int B, int H,
                    int W, int Cin, int Cout, int stride, int in_pitch, int out_pitch, int relu, int out_plain,
                    const cn_f32s_ctl *ctl, hipStream_t st);
extern int cn_tune_proj;   // cn_proj.hip
extern int cn_tune_stem_stagger, cn_tune_stem_dbg;                         // cn_stem.hip (probe instantiation of the stem + max-pool kernel)
bool cn_conv3x3s2p_takes(int B, int Hi, int Wi, int Cin, int Cout, int in_pitch, int out_pitch);
int cn_conv3x3s2_persist(const void *x, const void *w_packed, const float *scale, const float *shift, void *y,
                         int B, int Hi, int Wi, int Cin, int Cout, int in_pitch, int out_pitch, int relu,
                         int out_plain, const cn_f32s_ctl *ctl, hipStream_t st);
int cn_conv3x3s1(const void *x, const void *w_packed, const float *scale, const float *shift,
                 const void *residual, void *y, int B, int H, int W, int Cin, int Cout,
                 int in_pitch, int out_pitch, int res_pitch, int relu, int vec_out, int setprio, int bn_class,
                 int f16, const cn_f32s_ctl *ctl, hipStream_t st);
int cn_conv3x3_c16(const float *x, const float *w_packed, const float *scale, const float *shift,
                   float *y, int B, int H, int W, int Ho, int Wo, int Cin, int Cout, int stride,
                   int in_pitch, int out_pitch, int relu, hipStream_t st);
int cn_conv3x3_c16s(const float *x, const void *w_packed, const float *scale, const float *shift,
                    float *y, int B, int H, int W, int Ho, int Wo, int Cin, int Cout, int stride,
                    int in_pitch, int out_pitch, int relu, const cn_f32s_ctl *ctl, hipStream_t st);
extern int cn_tune_stagger_pct;  // cn_conv3x3.hip
extern int cn_tune_f32s_lds_weights;  // cn_conv3x3.hip
extern int cn_tune_f32s_policy;       // cn_conv3x3.hip
extern int cn_tune_heads_remap;       // cn_conv3x3.hip
extern int cn_tune_heads_reg;         // cn_conv3x3.hip
int cn_deconv4x4s2_halo(const void *x, const void *w_packed, const float *scale, const float *shift,
                        void *y, int B, int H, int W, int Cin, int Cout, int in_pitch, int out_pitch,
                        int relu, int vec_out, int setprio, int dtype_flags, const cn_f32s_ctl *ctl,
                        hipStream_t st);
namespace {

// split-K second stage: sum the partial tiles, then the usual epilogue
// y = relu?((sum + bias) * scale + shift + residual), NHWC (element type T)
template <typename T>
__global__ void splitk_reduce_kernel(const IgemmArgs a)
{
    const int n4 = a.cout_pad >> 2;
    const size_t total = (size_t)a.M * n4;
    const int HoWo = a.Ho * a.Wo;
    const size_t zstride = (size_t)a.M * a.cout_pad;
    float rng_out = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % n4);
        const int m = (int)(i / n4);
        const int n = c4 * 4;
        if (n >= a.Cout) continue;
        cn_f32x4 v = *reinterpret_cast<const cn_f32x4 *>(a.partial + (size_t)m * a.cout_pad + n);
        for (int z = 1; z < a.ksplit; ++z)
            v += *reinterpret_cast<const cn_f32x4 *>(a.partial + z * zstride + (size_t)m * a.cout_pad + n);
        const int b = m / HoWo;
        const int rr = m - b * HoWo;
        const int oy = rr / a.Wo;
        const int ox = rr - oy * a.Wo;
        const size_t off = (size_t)((b * a.OH + oy * a.oy_mul + a.oy_add) * a.OW + ox * a.ox_mul + a.ox_add);
        if constexpr (std::is_same<T, cn_f32s>::value) {
            for (int e = 0; e < 4 && (n + e) < a.Cout; ++e) {
                const float bs = a.bias ? a.bias[n + e] : 0.f;
                const float sc = a.scale ? a.scale[n + e] : 1.f;
                const float sf = a.shift ? a.shift[n + e] : 0.f;
                float t = (v[e] + bs) * sc + sf;
                const size_t o = off * a.out_pitch + n + e;
                if (a.residual)
                    t = fmaf(a.res_plain ? reinterpret_cast<const float *>(a.residual)[o]
                                         : cn_load1_f32s(a.residual, off, a.out_pitch, n + e),
                             a.res_mul, t);
                t = a.relu ? fmaxf(t, 0.f) : t;
                if (a.out_plain)
                    reinterpret_cast<float *>(a.y)[o] = t;
                else {
                    cn_rng_upd1(rng_out, t);
                    cn_store1_f32s(a.y, off, a.out_pitch, n + e, t);
                }
            }
        } else {
            const T *res = reinterpret_cast<const T *>(a.residual);
            T *y = reinterpret_cast<T *>(a.y);
            for (int e = 0; e < 4 && (n + e) < a.Cout; ++e) {
                const float bs = a.bias ? a.bias[n + e] : 0.f;
                const float sc = a.scale ? a.scale[n + e] : 1.f;
                const float sf = a.shift ? a.shift[n + e] : 0.f;
                float t = (v[e] + bs) * sc + sf;
                const size_t o = off * a.out_pitch + n + e;
                if (res) t += (float)res[o];
                y[o] = (T)(a.relu ? fmaxf(t, 0.f) : t);
            }
        }
    }
    if constexpr (std::is_same<T, cn_f32s>::value) {
        if (a.range && !a.out_plain) cn_rng_commit(a.range, 0, rng_out);
    }
}

// How many K-splits a dense NHWC layer gets: enough workgroups to put ~2 on every CU,
// at least 8 chunks of K per split, only when the plain grid is badly under-filled.
int g_tune_split_min_chunks = 8;  // cn_set_tuning key 16: K chunks per split-K slice, at least
int g_tune_split_max = 16;        // cn_set_tuning key 17: split-K slices, at most
inline int plan_ksplit(int M, int Cout, int KT, int bm, int bn)
{
    const long wgs = (long)cn_cdiv(M, bm) * cn_cdiv(Cout, bn);
    if (wgs >= 256 || KT < 16) return 1;
    int s = (int)((512 + wgs - 1) / wgs);
    if (s > KT / g_tune_split_min_chunks) s = KT / g_tune_split_min_chunks;
    if (s > g_tune_split_max) s = g_tune_split_max;
    return s < 2 ? 1 : s;
}
inline bool is_stem(int Cin, int in_layout) { return in_layout == CN_LAYOUT_NCHW && Cin == 3; }
inline void set_ctl(IgemmArgs &a, const cn_f32s_ctl *ctl)
{
    a.x_mul = (ctl && ctl->x_mul != 0.f) ? ctl->x_mul : 1.f;
    a.res_mul = (ctl && ctl->res_mul != 0.f) ? ctl->res_mul : 1.f;
    a.range = ctl ? ctl->range : nullptr;
}
// f32s tensors are addressed in 128-byte groups of 32 channels
inline bool aligned128(const void *p) { return (((uintptr_t)p) & 127u) == 0; }

}  // namespace

static size_t packed_elems(int Cout, int Cin, int KH, int KW, int bke)
{
    if (Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0) return 0;
    if (Cin == 3)  // stem form: [cout_pad][round_up(taps*3, chunk)]
        return (size_t)round_up(Cout, 32) * round_up(KH * KW * 3, bke);
    return (size_t)KH * KW * round_up(Cout, 32) * round_up(Cin, bke);
}

extern "C" size_t cn_packed_conv_weight_floats(int Cout, int Cin, int KH, int KW)
{
    return packed_elems(Cout, Cin, KH, KW, 32);
}

extern "C" size_t cn_packed_conv_weight_elems(int Cout, int Cin, int KH, int KW, int dtype)
{
    // f32s: counted in 4-byte units like fp32 (the stem keeps plain fp32 weights); 3x3 and 1x1 kernels
    // carry a second, fragment-ordered copy of the same size behind the row-ordered one
    const size_t n = packed_elems(Cout, Cin, KH, KW, dtype == CN_DTYPE_F16 ? 64 : 32);
    return (dtype == CN_DTYPE_F32S && ((KH == 3 && KW == 3) || (KH == 1 && KW == 1)) && Cin != 3) ? 2 * n : n;
}

template <typename T>
static int pack_conv_weight_t(const float *w_oihw, void *w_packed, int Cout, int Cin, int KH,
                              int KW, int bke, hipStream_t st)
{
    const int taps = KH * KW;
    const int cout_pad = round_up(Cout, 32);
    if (Cin == 3) {
        const int kpad = round_up(taps * 3, bke);
        const size_t total = (size_t)cout_pad * kpad;
        hipLaunchKernelGGL(pack_stem_weight_kernel<T>, dim3((unsigned)cn_cdiv((int)total, 256)),
                           dim3(256), 0, st, w_oihw, (T *)w_packed, Cout, taps, cout_pad, kpad);
    } else {
        const int cin_pad = round_up(Cin, bke);
        const size_t total = (size_t)taps * cout_pad * cin_pad;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(pack_weight_kernel<T>, dim3(blocks), dim3(256), 0, st, w_oihw,
                           (T *)w_packed, Cout, Cin, taps, cout_pad, cin_pad);
    }
    CN_CHECK_LAUNCH();
    return CN_OK;
}

extern "C" int cn_pack_conv_weight(const float *w_oihw, void *w_packed, int Cout, int Cin, int KH,
                                   int KW, int dtype, void *stream)
{
    if (!w_oihw || !w_packed) return CN_ERR_NULL;
    if (Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0) return CN_ERR_SHAPE;
    if (dtype == CN_DTYPE_F16)
        return pack_conv_weight_t<_Float16>(w_oihw, w_packed, Cout, Cin, KH, KW, 64, (hipStream_t)stream);
    if (dtype == CN_DTYPE_F32S && Cin != 3) {
        const int taps = KH * KW, cout_pad = round_up(Cout, 32), cin_pad = round_up(Cin, 32);
        const size_t total = (size_t)taps * cout_pad * cin_pad;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(pack_weight_f32s_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                           w_oihw, (_Float16 *)w_packed, Cout, Cin, taps, cout_pad, cin_pad);
        CN_CHECK_LAUNCH();
        if ((KH == 3 && KW == 3) || (KH == 1 && KW == 1)) {
            hipLaunchKernelGGL(pack_weight_f32s_frag_kernel, dim3(blocks), dim3(256), 0,
                               (hipStream_t)stream, w_oihw, (_Float16 *)w_packed + 2 * total, Cout, Cin,
                               taps, cout_pad / 32, cin_pad / 32);
            CN_CHECK_LAUNCH();
        }
        return CN_OK;
    }
    if (dtype != CN_DTYPE_F32 && dtype != CN_DTYPE_F32S) return CN_ERR_UNSUPPORTED;
    return pack_conv_weight_t<float>(w_oihw, w_packed, Cout, Cin, KH, KW, 32, (hipStream_t)stream);
}

extern "C" int cn_pack_conv_weight_f32(const float *w_oihw, float *w_packed, int Cout, int Cin,
                                       int KH, int KW, void *stream)
{
    return cn_pack_conv_weight(w_oihw, w_packed, Cout, Cin, KH, KW, CN_DTYPE_F32, stream);
}

static int conv_fill_args(const cn_conv_desc *d, IgemmArgs *a)
{
    if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->KH <= 0 ||
        d->KW <= 0 || d->stride <= 0 || d->dil <= 0 || d->Ho <= 0 || d->Wo <= 0)
        return CN_ERR_SHAPE;
    // shape rule (same as torch / dcn_v2_cuda.c:40-41)
    const int ho = (d->H + 2 * d->pad_h - (d->dil * (d->KH - 1) + 1)) / d->stride + 1;
    const int wo = (d->W + 2 * d->pad_w - (d->dil * (d->KW - 1) + 1)) / d->stride + 1;
    if (d->oy_mul == 1 && d->ox_mul == 1 && (ho != d->Ho || wo != d->Wo)) return CN_ERR_SHAPE;
    if ((long)d->B * d->H * d->W * (long)(d->in_layout == CN_LAYOUT_NHWC ? d->in_pitch : d->Cin) >=
        (1L << 31))
        return CN_ERR_UNSUPPORTED;
    if ((long)d->B * d->OH * d->OW * (long)(d->out_layout == CN_LAYOUT_NHWC ? d->out_pitch : d->Cout) >=
        (1L << 31))
        return CN_ERR_UNSUPPORTED;
    a->B = d->B; a->H = d->H; a->W = d->W; a->Cin = d->Cin;
    a->Ho = d->Ho; a->Wo = d->Wo; a->Cout = d->Cout;
    a->KH = d->KH; a->KW = d->KW; a->stride = d->stride;
    a->pad_h = d->pad_h; a->pad_w = d->pad_w; a->dil = d->dil;
    a->in_pitch = d->in_pitch; a->out_pitch = d->out_pitch;
    a->OH = d->OH; a->OW = d->OW;
    a->oy_mul = d->oy_mul; a->oy_add = d->oy_add; a->ox_mul = d->ox_mul; a->ox_add = d->ox_add;
    a->relu = d->relu;
    a->M = d->B * d->Ho * d->Wo;
    a->cout_pad = round_up(d->Cout, 32);
    const int f16 = (d->dtype == CN_DTYPE_F16);
    const int f32s = (d->dtype == CN_DTYPE_F32S);
    if (d->dtype != CN_DTYPE_F32 && !f16 && !f32s) return CN_ERR_UNSUPPORTED;
    // f32s tensors are addressed in 128-byte groups of 32 channels
    if (f32s && d->in_layout == CN_LAYOUT_NHWC && !(d->flags & CN_CONV_X_PLAIN) && (d->in_pitch & 31))
        return CN_ERR_UNSUPPORTED;
    if (f32s && d->out_layout == CN_LAYOUT_NHWC && !(d->flags & CN_CONV_Y_PLAIN) && (d->out_pitch & 31))
        return CN_ERR_UNSUPPORTED;
    const int bke = f16 ? 64 : 32, epv = f16 ? 8 : 4;
    if (is_stem(d->Cin, d->in_layout)) {
        a->cin_pad = round_up(d->KH * d->KW * 3, bke);
        if (a->cin_pad > STEM_KMAX) return CN_ERR_UNSUPPORTED;
        a->nchunk = a->cin_pad / bke;
        a->KT = a->nchunk;
    } else {
        if (d->in_layout != CN_LAYOUT_NHWC) return CN_ERR_UNSUPPORTED;
        if ((d->Cin % epv) || (d->in_pitch % epv) || d->in_pitch < d->Cin) return CN_ERR_UNSUPPORTED;
        a->cin_pad = round_up(d->Cin, bke);
        a->nchunk = a->cin_pad / bke;
        a->KT = d->KH * d->KW * a->nchunk;
    }
    return CN_OK;
}

// tile class of a dense layer: N tile 128 / 64 / 32 wide, pixel tile 128 or 64
static void dense_tile_class(const cn_conv_desc *d, const IgemmArgs &a, int *cls, bool *bm64)
{
    // 128-wide N tiles unless their padding wastes a whole 64-wide tile (e.g. Cout = 192)
    const int waste128 = cn_cdiv(d->Cout, 128) * 128 - d->Cout;
    const int waste64 = cn_cdiv(d->Cout, 64) * 64 - d->Cout;
    const bool narrow = !g_tune_narrow && (waste128 - waste64 >= 64);
    *cls = (d->Cout > 64 && !narrow) ? 2 : (d->Cout > 32 ? 1 : 0);
    // fewer than four workgroups per CU with 128-pixel tiles: halve the pixel tile
    const long wgs128 = (long)cn_cdiv(a.M, 128) * cn_cdiv(d->Cout, 128) * (a.zparity ? 4 : 1);
    // ... unless the 128-pixel grid is exactly one round of two workgroups per CU (512): then the
    // 64-pixel grid (1024 at three per CU = 1.33 rounds) loses (128->256/s2@32^2: 0.210 -> 0.183 ms)
    *bm64 = (*cls == 2) && (g_tune_bm ? (g_tune_bm == 64) : (wgs128 < 1024 && wgs128 != 512));
}

static int dense_ksplit(const cn_conv_desc *d, const IgemmArgs &a)
{
    if (g_tune_nosplit || d->out_layout != CN_LAYOUT_NHWC || is_stem(d->Cin, d->in_layout) ||
        a.zparity)
        return 1;
    int cls;
    bool bm64;
    dense_tile_class(d, a, &cls, &bm64);
    const int bm = bm64 ? 64 : 128, bn = cls == 2 ? 128 : (cls == 1 ? 64 : 32);
    return plan_ksplit(a.M, d->Cout, a.KT, bm, bn);
}

static bool is_3x3s1(const cn_conv_desc *d)
{
    return d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad_h == 1 && d->pad_w == 1 && d->dil == 1 &&
           d->oy_mul == 1 && d->ox_mul == 1 && d->oy_add == 0 && d->ox_add == 0 && d->OH == d->Ho &&
           d->OW == d->Wo;
}

// (mirror of cn_conv2d's route to cn_conv3x3s1; the caller passes the workspace its query asked for,
// so a layer that wants split-K gets it and stays on the implicit-GEMM kernel)
extern "C" int cn_conv2d_res_pitch_supported(const cn_conv_desc *d)
{
    IgemmArgs a = {};
    if (!d || conv_fill_args(d, &a) != CN_OK) return 0;
    if (g_tune_nohalo || !is_3x3s1(d) || is_stem(d->Cin, d->in_layout) || d->out_layout != CN_LAYOUT_NHWC) return 0;
    if (d->dtype == CN_DTYPE_F32 && d->Cin == 16 && d->Cout <= 32) return 0;     // cn_conv16.hip
    if (d->dtype == CN_DTYPE_F32S && (d->flags & CN_CONV_X_PLAIN) && (d->flags & CN_CONV_Y_PLAIN) && d->Cin == 16 &&
        d->Cout <= 32)
        return 0;
    a.in_plain = (d->flags & CN_CONV_X_PLAIN) ? 1 : 0;
    a.out_plain = (d->flags & CN_CONV_Y_PLAIN) ? 1 : 0;
    a.res_plain = (d->flags & CN_CONV_R_PLAIN) ? 1 : 0;
    return dense_ksplit(d, a) > 1 ? 0 : 1;
}

extern "C" size_t cn_conv2d_workspace_bytes(const cn_conv_desc *d)
{
    IgemmArgs a = {};
    if (!d || conv_fill_args(d, &a) != CN_OK) return 0;
    const int s = dense_ksplit(d, a);
    return s > 1 ? (size_t)s * a.M * a.cout_pad * sizeof(float) : 0;
}

extern "C" int cn_conv2d(const cn_conv_desc *d, const void *x, const void *w_packed,
                         const float *scale, const float *shift, const void *residual, void *y,
                         void *workspace, size_t workspace_bytes, void *stream)
{
    if (!d || !x || !w_packed || !y) return CN_ERR_NULL;
    if (!cn_aligned16(x) || !cn_aligned16(w_packed)) return CN_ERR_ALIGN;
    IgemmArgs a = {};
    int rc = conv_fill_args(d, &a);
    if (rc != CN_OK) return rc;
    a.x = x; a.w = w_packed; a.bias = nullptr; a.scale = scale; a.shift = shift;
    a.residual = residual; a.y = y; a.om = nullptr;
    set_ctl(a, &d->ctl);
    if (d->dtype == CN_DTYPE_F32S) {
        if (d->in_layout == CN_LAYOUT_NHWC && !(d->flags & CN_CONV_X_PLAIN) && !aligned128(x)) return CN_ERR_ALIGN;
        if (d->out_layout == CN_LAYOUT_NHWC && !(d->flags & CN_CONV_Y_PLAIN) && !aligned128(y)) return CN_ERR_ALIGN;
        if (residual && !(d->flags & CN_CONV_R_PLAIN) && !aligned128(residual)) return CN_ERR_ALIGN;
    }
    const bool f16 = (d->dtype == CN_DTYPE_F16);
    const size_t valign = f16 ? 8 : 16;  // 4 output elements per store
    a.vec_out = (d->out_layout == CN_LAYOUT_NHWC && (d->out_pitch & 3) == 0 &&
                 (((uintptr_t)y) % valign) == 0 &&
                 (!residual || (((uintptr_t)residual) % valign) == 0)) ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    const bool stem = is_stem(d->Cin, d->in_layout);
    const bool f32s = (d->dtype == CN_DTYPE_F32S);
    int cls;
    bool bm64;
    dense_tile_class(d, a, &cls, &bm64);
    // split-K for under-filled grids (needs the caller's workspace; skipped without it)
    a.ksplit = 1;
    a.in_plain = (d->flags & CN_CONV_X_PLAIN) ? 1 : 0;
    a.out_plain = (d->flags & CN_CONV_Y_PLAIN) ? 1 : 0;
    a.res_plain = (d->flags & CN_CONV_R_PLAIN) ? 1 : 0;
    const int want = dense_ksplit(d, a);
    if (want > 1 && workspace && cn_aligned16(workspace) &&
        workspace_bytes >= (size_t)want * a.M * a.cout_pad * sizeof(float)) {
        a.ksplit = want;
        a.partial = (float *)workspace;
    }
    // the stem reads the fp32 image and keeps fp32 packed weights; CN_CONV_STEM_F32S asks the
    // persistent stem kernel for f32s arithmetic (split inside the kernel), output plain fp32
    if (f32s && stem) return CN_ERR_UNSUPPORTED;
    if (d->out_layout == CN_LAYOUT_NCHW) {
        if (residual || stem) return CN_ERR_UNSUPPORTED;
        if (f32s) {
            if (cls == 2) return launch_igemm_s<128, 128, 2, 2, A_DENSE, true>(a, st);
            if (cls == 1) return launch_igemm_s<128, 64, 2, 2, A_DENSE, true>(a, st);
            return launch_igemm_s<128, 32, 4, 1, A_DENSE, true>(a, st);
        }
        if (f16) {
            if (cls == 2) return launch_igemm_h<128, 128, 2, 2, A_DENSE, true>(a, st);
            if (cls == 1) return launch_igemm_h<128, 64, 2, 2, A_DENSE, true>(a, st);
            return launch_igemm_h<128, 32, 4, 1, A_DENSE, true>(a, st);
        }
        if (cls == 2) return launch_igemm<128, 128, 2, 2, A_DENSE, true>(a, st);
        if (cls == 1) return launch_igemm<128, 64, 2, 2, A_DENSE, true>(a, st);
        return launch_igemm<128, 32, 4, 1, A_DENSE, true>(a, st);
    }
    if (stem) {
        if (f16) {
            if (d->Cout > 64) return launch_igemm_h<128, 128, 2, 2, A_STEM, false>(a, st);
            if (d->Cout > 32) return launch_igemm_h<128, 64, 2, 2, A_STEM, false>(a, st);
            return launch_igemm_h<128, 32, 4, 1, A_STEM, false>(a, st);
        }
        // LDS-window kernel (cn_stem.hip) when the tile's input window fits; else generic
        if (d->flags & CN_CONV_STEM_MAXPOOL) {
            // y is the max-pooled map (B, Ho/2, Wo/2): only the fused f32s kernel produces it
            if (!(d->flags & CN_CONV_STEM_F32S) || residual || d->pad_h != d->pad_w || d->dil != 1)
                return CN_ERR_UNSUPPORTED;
            return cn_stem_pool_f32s((const float *)x, (const float *)w_packed, scale, shift,
                                     (float *)y, d->B, d->H, d->W, d->Ho, d->Wo, d->Cout, d->KH,
                                     d->KW, d->stride, d->pad_h, d->relu, d->out_pitch, a.cin_pad,
                                     (d->flags & CN_CONV_STEM_Y_F32S) ? 1 : 0, &d->ctl, st);
        }
        if (!g_tune_nostem && d->pad_h == d->pad_w && d->dil == 1 && d->oy_mul == 1 &&
            d->ox_mul == 1 && d->OH == d->Ho && d->OW == d->Wo) {
            rc = cn_stem_conv_f32((const float *)x, (const float *)w_packed, scale, shift,
                                  (float *)y, d->B, d->H, d->W, d->Ho, d->Wo, d->Cout, d->KH,
                                  d->KW, d->stride, d->pad_h, d->relu, d->out_pitch, a.cin_pad,
                                  g_tune_stem_persist | ((d->flags & CN_CONV_STEM_F32S) ? 2 : 0), &d->ctl, st);
            if (rc != CN_ERR_UNSUPPORTED) return rc;
        }
        if (d->Cout > 64) return launch_igemm<128, 128, 2, 2, A_STEM, false>(a, st);
        if (d->Cout > 32) return launch_igemm<128, 64, 2, 2, A_STEM, false>(a, st);
        return launch_igemm<128, 32, 4, 1, A_STEM, false>(a, st);
    }
    // 16-channel input, <= 32 output channels (DLA level0 / level1): cn_conv16.hip
    if (!g_tune_nohalo && !f16 && !f32s && !residual && a.ksplit == 1 && d->Cin == 16 && d->Cout <= 32 &&
        d->KH == 3 && d->KW == 3 && d->pad_h == 1 && d->pad_w == 1 && d->dil == 1 &&
        d->oy_mul == 1 && d->ox_mul == 1 && d->oy_add == 0 && d->ox_add == 0 && d->OH == d->Ho &&
        d->OW == d->Wo && d->in_layout == CN_LAYOUT_NHWC) {
        rc = cn_conv3x3_c16((const float *)x, (const float *)w_packed, scale, shift, (float *)y, d->B,
                            d->H, d->W, d->Ho, d->Wo, d->Cin, d->Cout, d->stride, d->in_pitch,
                            d->out_pitch, d->relu, st);
        if (rc != CN_ERR_UNSUPPORTED) return rc;
    }
    // the same layers in f32s arithmetic: plain input split while staged, plain output (cn_conv16.hip)
    if (!g_tune_nohalo && f32s && (d->flags & CN_CONV_X_PLAIN) && (d->flags & CN_CONV_Y_PLAIN) && !residual &&
        a.ksplit == 1 && d->Cin == 16 && d->Cout <= 32 && d->KH == 3 && d->KW == 3 && d->pad_h == 1 &&
        d->pad_w == 1 && d->dil == 1 && d->oy_mul == 1 && d->ox_mul == 1 && d->oy_add == 0 &&
        d->ox_add == 0 && d->OH == d->Ho && d->OW == d->Wo && d->in_layout == CN_LAYOUT_NHWC) {
        rc = cn_conv3x3_c16s((const float *)x, w_packed, scale, shift, (float *)y, d->B, d->H, d->W,
                             d->Ho, d->Wo, d->Cin, d->Cout, d->stride, d->in_pitch, d->out_pitch,
                             d->relu, &d->ctl, st);
        if (rc != CN_ERR_UNSUPPORTED) return rc;
    }
    // <= 32 output channels on a plain fp32 tensor, plain output (the offset / mask convolution of the deformable
    // modules): cn_offconv.hip, with or without the K split
    if (f32s && (d->flags & CN_CONV_X_PLAIN) && (d->flags & CN_CONV_Y_PLAIN) && !residual && is_3x3s1(d) &&
        d->in_layout == CN_LAYOUT_NHWC && d->out_layout == CN_LAYOUT_NHWC && a.vec_out &&
        cn_offconv_takes(d->B, d->H, d->W, d->Cin, d->Cout, d->in_pitch, d->out_pitch, a.ksplit)) {
        rc = cn_offconv_f32s((const float *)x, w_packed, scale, shift, (float *)y, d->B, d->H, d->W, d->Cin, d->Cout,
                             d->out_pitch, d->relu, &d->ctl, a.ksplit, a.partial, st);
        if (rc != CN_OK || a.ksplit == 1) return rc;
        const size_t tot = (size_t)a.M * (a.cout_pad >> 2);
        hipLaunchKernelGGL(splitk_reduce_kernel<cn_f32s>, dim3((unsigned)((tot + 255) / 256 < 8192 ? (tot + 255) / 256 : 8192)),
                           dim3(256), 0, st, a);
        CN_CHECK_LAUNCH();
        return CN_OK;
    }
    // 3x3 / stride 1 / pad 1: the LDS-halo kernel (cn_conv3x3.hip) unless split-K applies
    const int res_pitch = d->res_pitch > 0 ? d->res_pitch : d->out_pitch;
    const bool to_halo = !g_tune_nohalo && a.ksplit == 1 && is_3x3s1(d);
    // a residual at its own pixel pitch (channel slices of wider tensors): only the 3x3 / s1 kernels
    if (residual && res_pitch != d->out_pitch && !to_halo) return CN_ERR_UNSUPPORTED;
    if (to_halo)
        return cn_conv3x3s1(x, w_packed, scale, shift, residual, y, d->B, d->H, d->W, d->Cin,
                            d->Cout, d->in_pitch, d->out_pitch, res_pitch, d->relu, a.vec_out,
                            g_tune_setprio | ((g_tune_bm256 & 1) << 1) | ((g_tune_bm256 >> 1) << 3) | (g_tune_waves8 << 2) | (g_tune_occ4 << 7) |
                                ((g_tune_dbgskip & 7) << 4) | ((g_tune_dbgskip >> 3) << 9), cls, d->dtype | (d->flags << 8),
                            &d->ctl, st);
    // 3x3 / stride 2 / pad 1, f32s tensors on both sides: the persistent kernel's parity-plane form
    if (!g_tune_nohalo && f32s && !(d->flags & (CN_CONV_X_PLAIN | CN_CONV_R_PLAIN)) && !residual && scale &&
        a.vec_out && d->KH == 3 && d->KW == 3 && d->stride == 2 && d->pad_h == 1 && d->pad_w == 1 && d->dil == 1 &&
        d->oy_mul == 1 && d->ox_mul == 1 && d->oy_add == 0 && d->ox_add == 0 && d->OH == d->Ho && d->OW == d->Wo &&
        d->in_layout == CN_LAYOUT_NHWC && d->out_layout == CN_LAYOUT_NHWC &&
        cn_conv3x3s2p_takes(d->B, d->H, d->W, d->Cin, d->Cout, d->in_pitch, d->out_pitch))
        return cn_conv3x3s2_persist(x, w_packed, scale, shift, y, d->B, d->H, d->W, d->Cin, d->Cout, d->in_pitch,
                                    d->out_pitch, d->relu, (d->flags & CN_CONV_Y_PLAIN) ? 1 : 0, &d->ctl, st);
    // 1x1 (stride 1 or 2), f32s tensors, no residual -- the `downsample` projections: the direct-fragment kernel
    // (cn_proj.hip), bound by the layer's HBM bytes instead of the implicit GEMM's fixed costs
    if (f32s && !(d->flags & (CN_CONV_X_PLAIN | CN_CONV_R_PLAIN)) && !residual && scale && a.vec_out && d->KH == 1 &&
        d->KW == 1 && d->pad_h == 0 && d->pad_w == 0 && d->dil == 1 && d->oy_mul == 1 && d->ox_mul == 1 &&
        d->oy_add == 0 && d->ox_add == 0 && d->OH == d->Ho && d->OW == d->Wo && d->in_layout == CN_LAYOUT_NHWC &&
        d->out_layout == CN_LAYOUT_NHWC && d->Ho == (d->H - 1) / d->stride + 1 && d->Wo == (d->W - 1) / d->stride + 1 &&
        cn_proj1x1_takes(d->B, d->H, d->W, d->Cin, d->Cout, d->stride, d->in_pitch, d->out_pitch))
        return cn_proj1x1_f32s(x, w_packed, scale, shift, y, d->B, d->H, d->W, d->Cin, d->Cout, d->stride, d->in_pitch,
                               d->out_pitch, d->relu, (d->flags & CN_CONV_Y_PLAIN) ? 1 : 0, &d->ctl, st);
    if (f32s) {
        if (cls == 2)
            rc = bm64 ? launch_igemm_s<64, 128, 2, 2, A_DENSE, false>(a, st)
                      : launch_igemm_s<128, 128, 2, 2, A_DENSE, false>(a, st);
        else if (cls == 1)
            rc = launch_igemm_s<128, 64, 2, 2, A_DENSE, false>(a, st);
        else
            rc = launch_igemm_s<128, 32, 4, 1, A_DENSE, false>(a, st);
    } else if (f16) {
        if (cls == 2)
            rc = bm64 ? launch_igemm_h<64, 128, 2, 2, A_DENSE, false>(a, st)
                      : launch_igemm_h<128, 128, 2, 2, A_DENSE, false>(a, st);
        else if (cls == 1)
            rc = launch_igemm_h<128, 64, 2, 2, A_DENSE, false>(a, st);
        else
            rc = launch_igemm_h<128, 32, 4, 1, A_DENSE, false>(a, st);
    } else {
        if (cls == 2)
            rc = bm64 ? launch_igemm<64, 128, 2, 2, A_DENSE, false>(a, st)
                      : launch_igemm<128, 128, 2, 2, A_DENSE, false>(a, st);
        else if (cls == 1)
            rc = launch_igemm<128, 64, 2, 2, A_DENSE, false>(a, st);
        else
            rc = launch_igemm<128, 32, 4, 1, A_DENSE, false>(a, st);
    }
    if (rc != CN_OK || a.ksplit == 1) return rc;
    const size_t total = (size_t)a.M * (a.cout_pad >> 2);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    if (f32s)
        hipLaunchKernelGGL(splitk_reduce_kernel<cn_f32s>, dim3(blocks), dim3(256), 0, st, a);
    else if (f16)
        hipLaunchKernelGGL(splitk_reduce_kernel<_Float16>, dim3(blocks), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL(splitk_reduce_kernel<float>, dim3(blocks), dim3(256), 0, st, a);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

extern "C" int cn_conv2d_f32(const cn_conv_desc *d, const float *x, const float *w_packed,
                             const float *scale, const float *shift, const float *residual,
                             float *y, void *stream)
{
    if (d && d->dtype != CN_DTYPE_F32) return CN_ERR_UNSUPPORTED;
    return cn_conv2d(d, x, w_packed, scale, shift, residual, y, nullptr, 0, stream);
}

// Tap-split factor of the deformable kernel: the gather makes every K chunk latency-bound, so a
// layer needs ~4 workgroups per CU to keep the matrix cores busy; small maps get them by
// splitting the 9 taps over 3 or 9 workgroups per tile (fp32 partial sums + reduce kernel).
static int dcn_ksplit(int B, int H, int W, int Cout)
{
    if (g_tune_nosplit || g_tune_dcn_split == 1) return 1;
    if (g_tune_dcn_split == 3 || g_tune_dcn_split == 9) return g_tune_dcn_split;
    const long wgs = (long)cn_cdiv(B * H * W, 64) * cn_cdiv(Cout, Cout > 64 ? 128 : 64);
    if (wgs >= 1024) return 1;
    return wgs * 3 >= 600 ? 3 : 9;  // measured (tools/bench_dcn.py): 3 wins from ~2 workgroups/CU
}

extern "C" size_t cn_dcn_v2_forward_nhwc_workspace_bytes(int B, int Cin, int H, int W, int Cout)
{
    if (B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return 0;
    // slabs of the gather form's tap split (3 / 9) or of the window form's K-chunk split (<= 8):
    // small grids only (either form splits when its plain grid has < 1024 / < 192 workgroups)
    const int s = dcn_ksplit(B, H, W, Cout);
    return s > 1 ? (size_t)(s > 8 ? s : 8) * B * H * W * round_up(Cout, 32) * sizeof(float) : 0;
}

extern "C" int cn_dcn_v2_forward_nhwc_f32(const float *input_nhwc, const float *weight_packed,
                                          const float *bias, const float *offset_mask_nhwc,
                                          int om_pitch, const float *scale, const float *shift,
                                          float *output_nhwc, int B, int Cin, int H, int W,
                                          int Cout, int mask_sigmoid, int relu, void *workspace,
                                          size_t workspace_bytes, void *stream)
{
    return cn_dcn_v2_forward_nhwc(input_nhwc, weight_packed, bias, offset_mask_nhwc, om_pitch, scale,
                                  shift, output_nhwc, Cout, B, Cin, H, W, Cout, mask_sigmoid, relu,
                                  CN_DTYPE_F32, 0, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int cn_dcn_v2_forward_nhwc(const float *input_nhwc, const void *weight_packed,
                                      const float *bias, const float *offset_mask_nhwc,
                                      int om_pitch, const float *scale, const float *shift,
                                      void *output_nhwc, int out_pitch, int B, int Cin, int H,
                                      int W, int Cout, int mask_sigmoid, int relu, int dtype,
                                      int flags, const cn_f32s_ctl *ctl, void *workspace,
                                      size_t workspace_bytes, void *stream)
{
    if (!input_nhwc || !weight_packed || !offset_mask_nhwc || !output_nhwc) return CN_ERR_NULL;
    if (B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0 || out_pitch < Cout) return CN_ERR_SHAPE;
    if (om_pitch < 27) return CN_ERR_SHAPE;
    if ((Cin & 3) != 0) return CN_ERR_UNSUPPORTED;
    const bool f32s = (dtype == CN_DTYPE_F32S);
    if (dtype != CN_DTYPE_F32 && !f32s) return CN_ERR_UNSUPPORTED;
    if (f32s && !(flags & CN_CONV_Y_PLAIN) && (out_pitch & 31)) return CN_ERR_UNSUPPORTED;
    if (!cn_aligned16(input_nhwc) || !cn_aligned16(weight_packed)) return CN_ERR_ALIGN;
    if (f32s && !(flags & CN_CONV_Y_PLAIN) && !aligned128(output_nhwc)) return CN_ERR_ALIGN;
    if ((long)B * H * W * (long)(Cin > out_pitch ? Cin : out_pitch) >= (1L << 30)) return CN_ERR_UNSUPPORTED;  // 32-bit byte offsets
    // f32s: the LDS-window forms (cn_dcn3.hip team form, cn_dcn2.hip register-sampling form) for every
    // shape they take and every grid that fills the chip; the gather form below (tap split) serves the rest
    if (f32s && g_tune_dcn_form != 1) {
        const bool forced = g_tune_dcn_form >= 2;
        const bool ws_ok = workspace && cn_aligned16(workspace) && !g_tune_nosplit;
        int ks = 1;
        int rc = CN_ERR_UNSUPPORTED;
        // team form: forced by key 23 = 4 (T mode) / 5 (N mode), or chosen by key 36 for grids that fill the chip
        const long tiles128 = (long)B * (H / 8) * (W / 16);
        const bool team_auto = g_tune_dcn_form == 0 && tiles128 >= 64 &&
                               ((cn_tune_dcn_team == 1 && Cout <= 64) || cn_tune_dcn_team >= 2);
        // wide form (cn_dcn4.hip): every sample once per tile for ALL output channels; Cout % 128 == 0
        const bool wide_auto = g_tune_dcn_form == 0 && cn_tune_dcn_wide && tiles128 >= 64 && (Cout & 127) == 0 && !g_tune_dbgskip;
        if (g_tune_dcn_form == 6 || g_tune_dcn_form == 7 || wide_auto)
            rc = cn_dcn_wide_f32s(input_nhwc, weight_packed, bias, offset_mask_nhwc, om_pitch, scale, shift,
                                  output_nhwc, out_pitch, (flags & CN_CONV_Y_PLAIN) ? 1 : 0, B, Cin, H, W, Cout,
                                  mask_sigmoid, relu, (ctl && ctl->x_mul != 0.f) ? ctl->x_mul : 1.f,
                                  ctl ? ctl->range : nullptr, g_tune_dcn_form == 7 ? 4 : 0, g_tune_dbgskip,
                                  ws_ok ? (float *)workspace : nullptr, ws_ok ? workspace_bytes : 0, &ks,
                                  (hipStream_t)stream);
        if (rc == CN_ERR_UNSUPPORTED && (g_tune_dcn_form == 4 || g_tune_dcn_form == 5 || team_auto))
            rc = cn_dcn_team_f32s(input_nhwc, weight_packed, bias, offset_mask_nhwc, om_pitch, scale, shift,
                                  output_nhwc, out_pitch, (flags & CN_CONV_Y_PLAIN) ? 1 : 0, B, Cin, H, W, Cout,
                                  mask_sigmoid, relu, (ctl && ctl->x_mul != 0.f) ? ctl->x_mul : 1.f,
                                  ctl ? ctl->range : nullptr,
                                  g_tune_dcn_form == 5 ? 2 : ((team_auto && cn_tune_dcn_team == 3) ? 1 : 0), g_tune_dbgskip,
                                  ws_ok ? (float *)workspace : nullptr, ws_ok ? workspace_bytes : 0, &ks,
                                  (hipStream_t)stream);
        if (rc == CN_ERR_UNSUPPORTED && g_tune_dcn_form < 4)
            rc = cn_dcn_window_f32s(input_nhwc, weight_packed, bias, offset_mask_nhwc, om_pitch,
                                    scale, shift, output_nhwc, out_pitch,
                                    (flags & CN_CONV_Y_PLAIN) ? 1 : 0, B, Cin, H, W, Cout,
                                    mask_sigmoid, relu, (ctl && ctl->x_mul != 0.f) ? ctl->x_mul : 1.f,
                                    ctl ? ctl->range : nullptr, forced ? 1 : 192, g_tune_dbgskip,
                                    ws_ok ? (float *)workspace : nullptr, ws_ok ? workspace_bytes : 0,
                                    &ks, (hipStream_t)stream);
        if (rc == CN_OK && ks > 1) {
            // second stage of the K split: fixed-order sum of the slabs + the usual epilogue
            IgemmArgs r = {};
            r.bias = bias; r.scale = scale; r.shift = shift; r.residual = nullptr; r.y = output_nhwc;
            set_ctl(r, ctl);
            r.B = B; r.H = H; r.W = W; r.Ho = H; r.Wo = W; r.Cout = Cout; r.out_pitch = out_pitch;
            r.out_plain = (flags & CN_CONV_Y_PLAIN) ? 1 : 0;
            r.OH = H; r.OW = W; r.oy_mul = 1; r.oy_add = 0; r.ox_mul = 1; r.ox_add = 0;
            r.relu = relu; r.M = B * H * W; r.cout_pad = round_up(Cout, 32);
            r.ksplit = ks; r.partial = (float *)workspace;
            const size_t tot = (size_t)r.M * (r.cout_pad >> 2);
            const int nb = (int)((tot + 255) / 256 < 4096 ? (tot + 255) / 256 : 4096);
            hipLaunchKernelGGL(splitk_reduce_kernel<cn_f32s>, dim3(nb), dim3(256), 0, (hipStream_t)stream, r);
            CN_CHECK_LAUNCH();
            return CN_OK;
        }
        if (rc != CN_ERR_UNSUPPORTED) return rc;
    }
    IgemmArgs a = {};
    a.x = input_nhwc; a.w = weight_packed; a.bias = bias; a.scale = scale; a.shift = shift;
    a.residual = nullptr; a.y = output_nhwc; a.om = offset_mask_nhwc; a.om_pitch = om_pitch;
    a.mask_sigmoid = mask_sigmoid;
    set_ctl(a, ctl);
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Ho = H; a.Wo = W; a.Cout = Cout;
    a.KH = 3; a.KW = 3; a.stride = 1; a.pad_h = 1; a.pad_w = 1; a.dil = 1;
    a.in_pitch = Cin; a.out_pitch = out_pitch;
    a.in_plain = 1;                                   // the gather reads plain fp32 in every mode
    a.out_plain = (flags & CN_CONV_Y_PLAIN) ? 1 : 0;
    a.OH = H; a.OW = W; a.oy_mul = 1; a.oy_add = 0; a.ox_mul = 1; a.ox_add = 0;
    a.relu = relu;
    a.M = B * H * W;
    a.cin_pad = round_up(Cin, 32);
    a.cout_pad = round_up(Cout, 32);
    a.nchunk = a.cin_pad / 32;
    a.KT = 9 * a.nchunk;
    a.vec_out = ((Cout & 3) == 0 && (out_pitch & 3) == 0 && cn_aligned16(output_nhwc)) ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    // tiles as pixel blocks (8 x 8, or 8 x 16 for the 128-pixel tiles of Cout <= 32) when the map
    // divides into them
    a.tile2d = (g_tune_dcn_tile2d && (W & 7) == 0 && (H % (Cout > 32 ? 8 : 16)) == 0) ? 1 : 0;
    // 64-pixel tiles: 128-pixel tiles were measured slower at every CenterNet shape
    // (tools/bench_dcn.py) and are no longer built
    // tap split (needs the caller's workspace; without one the layer runs unsplit)
    a.ksplit = 1;
    if (Cout > 32) {
        const int want = dcn_ksplit(B, H, W, Cout);
        const size_t need = (size_t)want * a.M * a.cout_pad * sizeof(float);
        if (want > 1 && workspace && workspace_bytes >= need && cn_aligned16(workspace)) {
            a.ksplit = want;
            a.partial = (float *)workspace;
        }
    }
    int rc;
    const bool padk = (Cin & 31) != 0;
    if (f32s) {
        if (Cout > 64)
            rc = padk ? launch_igemm_s<64, 128, 2, 2, A_DCN_PAD, false>(a, st)
                      : launch_igemm_s<64, 128, 2, 2, A_DCN, false>(a, st);
        else if (Cout > 32)
            rc = padk ? launch_igemm_s<64, 64, 2, 2, A_DCN_PAD, false>(a, st)
                      : launch_igemm_s<64, 64, 2, 2, A_DCN, false>(a, st);
        else
            rc = padk ? launch_igemm_s<128, 32, 4, 1, A_DCN_PAD, false>(a, st)
                      : launch_igemm_s<128, 32, 4, 1, A_DCN, false>(a, st);
        if (rc != CN_OK || a.ksplit == 1) return rc;
        const size_t tot = (size_t)a.M * (a.cout_pad >> 2);
        const int nb = (int)((tot + 255) / 256 < 4096 ? (tot + 255) / 256 : 4096);
        hipLaunchKernelGGL(splitk_reduce_kernel<cn_f32s>, dim3(nb), dim3(256), 0, st, a);
        CN_CHECK_LAUNCH();
        return CN_OK;
    }
    if (Cout > 64)
        rc = padk ? launch_igemm<64, 128, 2, 2, A_DCN_PAD, false>(a, st)
                  : launch_igemm<64, 128, 2, 2, A_DCN, false>(a, st);
    else if (Cout > 32)
        rc = padk ? launch_igemm<64, 64, 2, 2, A_DCN_PAD, false>(a, st)
                  : launch_igemm<64, 64, 2, 2, A_DCN, false>(a, st);
    else
        rc = padk ? launch_igemm<128, 32, 4, 1, A_DCN_PAD, false>(a, st)
                  : launch_igemm<128, 32, 4, 1, A_DCN, false>(a, st);
    if (rc != CN_OK || a.ksplit == 1) return rc;
    const size_t total = (size_t)a.M * (a.cout_pad >> 2);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(splitk_reduce_kernel<float>, dim3(blocks), dim3(256), 0, st, a);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

// ---- ConvTranspose2d(kernel 4, stride 2, padding 1, no output padding) -----------------
// out[2q+py, 2r+px] = sum_{ty,tx,ci} in[q - (1-py) + ty, r - (1-px) + tx, ci] * w[ci, co, ky(py,ty), kx(px,tx)]
// with ky(0,.) = {3,1}, ky(1,.) = {2,0}: four 2x2 convolutions, one launch (blockIdx.z).
namespace {
__global__ void pack_deconv_weight_kernel(const float *__restrict__ w, float *__restrict__ wp,
                                          int Cin, int Cout, int cout_pad, int cin_pad, int f32s)
{
    const size_t total = (size_t)16 * cout_pad * cin_pad;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cin_pad);
        const int n = (int)((i / cin_pad) % cout_pad);
        const int t = (int)((i / ((size_t)cin_pad * cout_pad)) % 4);
        const int z = (int)(i / ((size_t)cin_pad * cout_pad * 4));
        const int py = z >> 1, px = z & 1, ty = t >> 1, tx = t & 1;
        const int ky = py ? (ty ? 0 : 2) : (ty ? 1 : 3);
        const int kx = px ? (tx ? 0 : 2) : (tx ? 1 : 3);
        float v = 0.f;
        if (c < Cin && n < Cout) v = w[(((size_t)c * Cout + n) * 4 + ky) * 4 + kx];
        if (f32s) {   // fp16 (high, low) pairs per 32-channel group, see pack_weight_f32s_kernel
            _Float16 *wh = reinterpret_cast<_Float16 *>(wp);
            const _Float16 hi = (_Float16)v;
            const size_t g = (i - (size_t)(c & 31)) * 2;
            wh[g + (c & 31)] = hi;
            wh[g + 32 + (c & 31)] = (_Float16)(v - (float)hi);
        } else {
            wp[i] = v;
        }
    }
}
}  // namespace

extern "C" size_t cn_packed_deconv4x4s2_weight_floats(int Cin, int Cout)
{
    if (Cin <= 0 || Cout <= 0) return 0;
    return (size_t)16 * round_up(Cout, 32) * round_up(Cin, 32);
}

extern "C" int cn_pack_deconv4x4s2_weight_f32(const float *w_iohw, float *w_packed, int Cin,
                                              int Cout, void *stream)
{
    return cn_pack_deconv4x4s2_weight(w_iohw, w_packed, Cin, Cout, CN_DTYPE_F32, stream);
}

extern "C" int cn_pack_deconv4x4s2_weight(const float *w_iohw, void *w_packed, int Cin, int Cout,
                                          int dtype, void *stream)
{
    if (!w_iohw || !w_packed) return CN_ERR_NULL;
    if (Cin <= 0 || Cout <= 0) return CN_ERR_SHAPE;
    if (dtype != CN_DTYPE_F32 && dtype != CN_DTYPE_F32S) return CN_ERR_UNSUPPORTED;
    const size_t total = cn_packed_deconv4x4s2_weight_floats(Cin, Cout);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_deconv_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       w_iohw, (float *)w_packed, Cin, Cout, round_up(Cout, 32), round_up(Cin, 32),
                       dtype == CN_DTYPE_F32S ? 1 : 0);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

extern "C" int cn_conv_transpose4x4s2_f32(const float *x_nhwc, const float *w_packed,
                                          const float *scale, const float *shift, float *y_nhwc,
                                          int B, int H, int W, int Cin, int Cout, int in_pitch,
                                          int out_pitch, int relu, void *stream)
{
    return cn_conv_transpose4x4s2(x_nhwc, w_packed, scale, shift, y_nhwc, B, H, W, Cin, Cout,
                                  in_pitch, out_pitch, relu, CN_DTYPE_F32, 0, nullptr, stream);
}

extern "C" int cn_conv_transpose4x4s2(const void *x_nhwc, const void *w_packed, const float *scale,
                                      const float *shift, void *y_nhwc, int B, int H, int W,
                                      int Cin, int Cout, int in_pitch, int out_pitch, int relu,
                                      int dtype, int flags, const cn_f32s_ctl *ctl, void *stream)
{
    if (!x_nhwc || !w_packed || !y_nhwc) return CN_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return CN_ERR_SHAPE;
    if ((Cin & 3) || (in_pitch & 3) || in_pitch < Cin || out_pitch < Cout) return CN_ERR_UNSUPPORTED;
    const bool f32s = (dtype == CN_DTYPE_F32S);
    if (dtype != CN_DTYPE_F32 && !f32s) return CN_ERR_UNSUPPORTED;
    if (f32s && !(flags & CN_CONV_X_PLAIN) && (in_pitch & 31)) return CN_ERR_UNSUPPORTED;
    if (f32s && !(flags & CN_CONV_Y_PLAIN) && (out_pitch & 31)) return CN_ERR_UNSUPPORTED;
    if (!cn_aligned16(x_nhwc) || !cn_aligned16(w_packed)) return CN_ERR_ALIGN;
    if (f32s && ((!(flags & CN_CONV_X_PLAIN) && !aligned128(x_nhwc)) ||
                 (!(flags & CN_CONV_Y_PLAIN) && !aligned128(y_nhwc))))
        return CN_ERR_ALIGN;
    if ((long)B * H * W * 4 * (long)out_pitch >= (1L << 31) || (long)B * H * W * (long)in_pitch >= (1L << 31))
        return CN_ERR_UNSUPPORTED;
    IgemmArgs a = {};
    set_ctl(a, ctl);
    a.x = x_nhwc; a.w = w_packed; a.scale = scale; a.shift = shift; a.y = y_nhwc;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Ho = H; a.Wo = W; a.Cout = Cout;
    a.KH = 2; a.KW = 2; a.stride = 1; a.pad_h = 1; a.pad_w = 1; a.dil = 1;
    a.in_pitch = in_pitch; a.out_pitch = out_pitch;
    a.OH = 2 * H; a.OW = 2 * W; a.oy_mul = 2; a.ox_mul = 2; a.oy_add = 0; a.ox_add = 0;
    a.relu = relu;
    a.M = B * H * W;
    a.cin_pad = round_up(Cin, 32);
    a.cout_pad = round_up(Cout, 32);
    a.nchunk = a.cin_pad / 32;
    a.KT = 4 * a.nchunk;
    a.zparity = 1;
    a.w_zstride = 4 * a.cout_pad * a.cin_pad;
    a.vec_out = ((out_pitch & 3) == 0 && cn_aligned16(y_nhwc)) ? 1 : 0;
    a.in_plain = (flags & CN_CONV_X_PLAIN) ? 1 : 0;
    a.out_plain = (flags & CN_CONV_Y_PLAIN) ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    // LDS-halo form (cn_conv3x3.hip) unless disabled (cn_set_tuning key 10) or the tile would be
    // mostly padding (Cout <= 32)
    if (!g_tune_nohalo && Cout > 32 && a.vec_out && (in_pitch & 3) == 0)
        return cn_deconv4x4s2_halo(x_nhwc, w_packed, scale, shift, y_nhwc, B, H, W, Cin, Cout,
                                   in_pitch, out_pitch, relu, a.vec_out,
                                   g_tune_setprio | (g_tune_occ4 << 7), dtype | (flags << 8), ctl, st);
    if (f32s) {
        if (Cout > 64) return launch_igemm_s<128, 128, 2, 2, A_DENSE, false>(a, st);
        if (Cout > 32) return launch_igemm_s<128, 64, 2, 2, A_DENSE, false>(a, st);
        return launch_igemm_s<128, 32, 4, 1, A_DENSE, false>(a, st);
    }
    if (Cout > 64) return launch_igemm<128, 128, 2, 2, A_DENSE, false>(a, st);
    if (Cout > 32) return launch_igemm<128, 64, 2, 2, A_DENSE, false>(a, st);
    return launch_igemm<128, 32, 4, 1, A_DENSE, false>(a, st);
}

extern "C" int cn_stem_maxpool_supported(const cn_conv_desc *d)
{
    if (!d || !is_stem(d->Cin, d->in_layout) || d->out_layout != CN_LAYOUT_NHWC) return 0;
    if (d->dtype != CN_DTYPE_F32 || d->pad_h != d->pad_w || d->dil != 1) return 0;
    if (round_up(d->KH * d->KW * 3, 32) > STEM_KMAX) return 0;
    return cn_stem_pool_rows(d->B, d->Ho, d->Wo, d->Cout, d->KH, d->KW, d->stride,
                             round_up(d->KH * d->KW * 3, 32)) > 0;
}

// Will cn_conv2d honour CN_CONV_STEM_F32S for this stem descriptor (mirror of its dispatch)?
extern "C" int cn_stem_f32s_supported(const cn_conv_desc *d)
{
    if (!d || !is_stem(d->Cin, d->in_layout) || d->out_layout != CN_LAYOUT_NHWC) return 0;
    if (d->dtype != CN_DTYPE_F32 || d->pad_h != d->pad_w || d->dil != 1) return 0;
    const int kp = round_up(d->KH * d->KW * 3, 32);
    if (kp > STEM_KMAX) return 0;
    if (d->flags & CN_CONV_STEM_MAXPOOL) return cn_stem_maxpool_supported(d);
    if (g_tune_nostem || !g_tune_stem_persist) return 0;
    if (d->oy_mul != 1 || d->ox_mul != 1 || d->OH != d->Ho || d->OW != d->Wo) return 0;
    // cn_stem_conv_f32: persistent 7x7 window kernel, rows of whole 128-pixel tiles, stride 2,
    // more than 16 output channels
    if (d->KH != 7 || d->KW != 7 || d->Wo % 128 != 0) return 0;
    if (d->stride == 2 && d->Cout > 16) return 1;
    // stem16s_kernel: stride 1, pad 3, <= 16 output channels (DLA base_layer)
    return (d->stride == 1 && d->Cout <= 16 && d->pad_h == 3 && (d->W & 3) == 0 && g_tune_stem16s) ? 1 : 0;
}

extern int cn_tune_c3p, cn_tune_c3p_stagger, cn_tune_c3p_knobs, cn_tune_c3p_heads, cn_tune_c3p_deconv, cn_tune_c3p_s2;
extern "C" int cn_set_tuning(int key, int value)
{
    if (key == 28 && value >= 0 && value <= 7) {
        cn_tune_c3p = value;
        return CN_OK;
    }
    if (key == 29 && value >= 0 && value <= 255) {
        cn_tune_c3p_stagger = value;
        return CN_OK;
    }
    if (key == 30 && value >= 0 && value <= 255) {
        cn_tune_c3p_knobs = value;
        return CN_OK;
    }
    if (key == 31 && (value == 0 || value == 1)) {
        cn_tune_c3p_heads = value;
        return CN_OK;
    }
    if (key == 32 && (value == 0 || value == 1)) {
        cn_tune_c3p_deconv = value;
        return CN_OK;
    }
    if (key == 33 && (value == 0 || value == 1)) {
        cn_tune_c3p_s2 = value;
        return CN_OK;
    }

    if (key == 36 && value >= 0 && value <= 3) {
        cn_tune_dcn_team = value;
        return CN_OK;
    }
    if (key == 37 && value >= 1 && value <= 4096) {
        cn_tune_dcn_team_wgs = value;
        return CN_OK;
    }
    if (key == 39 && (value == 0 || value == 1)) {
        cn_tune_offconv = value;
        return CN_OK;
    }
    if (key == 40 && value >= 0 && value <= 1000000) {
        cn_tune_offconv_teams1 = value;
        return CN_OK;
    }
    if (key == 38 && value >= 0 && value <= 1024) {
        cn_tune_dcn_team_stagger = value;
        return CN_OK;
    }

    if (key == 20 && (value == 0 || value == 1)) {
        cn_tune_f32s_lds_weights = value;
        return CN_OK;
    }
    if (key == 21 && value >= 0 && value <= 7) {
        cn_tune_f32s_policy = value;
        return CN_OK;
    }
    if (key == 1 && value >= 0 && value <= 2) {
        g_tune_nbuf = value;
        return CN_OK;
    }
    if (key == 2 && (value == 0 || value == 1)) {
        g_tune_narrow = value;
        return CN_OK;
    }
    if (key == 3 && (value == 0 || value == 64)) return CN_OK;  // 128-pixel DCN tiles: retired
    if (key == 4 && (value == 0 || value == 64 || value == 128)) {
        g_tune_bm = value;
        return CN_OK;
    }
    if (key == 5 && (value == 0 || value == 1)) {
        g_tune_nosplit = value;
        return CN_OK;
    }
    if (key == 6 && (value == 0 || value == 1)) {
        g_tune_nostem = value;
        return CN_OK;
    }
    if (key == 8 && (value == 0 || value == 1)) {
        g_tune_setprio = value;
        return CN_OK;
    }
    if (key == 9 && value >= 0 && value <= 2047) {
        g_tune_dbgskip = value;
        return CN_OK;
    }
    if (key == 10 && (value == 0 || value == 1)) {
        g_tune_nohalo = value;
        return CN_OK;
    }
    if (key == 11 && value == 0) return CN_OK;   // fp32 LDS-window DCN kernel: retired in round 5 (20-30 % slower than the gather form)
    if (key == 7 && (value == 0 || value == 1 || value == 2)) {
        g_tune_swz = value;
        return CN_OK;
    }
    if (key == 13 && (value == 0 || value == 1 || value == 3 || value == 9)) {
        g_tune_dcn_split = value;
        return CN_OK;
    }
    if (key == 14 && value >= 0 && value <= 3) {
        g_tune_bm256 = value;
        return CN_OK;
    }
    if (key == 16 && value >= 1 && value <= 64) {
        g_tune_split_min_chunks = value;
        return CN_OK;
    }
    if (key == 17 && value >= 1 && value <= 64) {
        g_tune_split_max = value;
        return CN_OK;
    }
    if (key == 18 && value >= 0 && value <= 255) {
        cn_tune_stagger_pct = value;
        return CN_OK;
    }
    if (key == 19 && value >= 0 && value <= 2) {
        g_tune_occ4 = value;
        return CN_OK;
    }
    if (key == 15 && (value == 0 || value == 1)) {
        g_tune_waves8 = value;
        return CN_OK;
    }
    if (key == 12 && (value == 0 || value == 1)) {
        g_tune_stem_persist = value;
        return CN_OK;
    }
    if (key == 22 && (value == 0 || value == 1)) {
        g_tune_dcn_tile2d = value;
        return CN_OK;
    }
    if (key == 27 && (value == 0 || value == 1)) {
        g_tune_stem16s = value;
        return CN_OK;
    }
    if (key == 26 && value >= 0 && value <= 3) {
        cn_tune_heads_reg = value;
        return CN_OK;
    }
    if (key == 24 && value >= 0 && value <= 3) {
        cn_tune_heads_remap = value;
        return CN_OK;
    }
    if (key == 46 && (value == 0 || value == 1)) {
        cn_tune_proj = value;
        return CN_OK;
    }
    if (key == 45 && (value == 0 || value == 1)) {
        cn_tune_dcn_wide_prefetch = value;
        return CN_OK;
    }
    if (key == 44 && value >= 0 && value <= 1024) {
        cn_tune_stem_stagger = value;
        return CN_OK;
    }
    if (key == 43 && value >= 0 && value <= 31) {
        cn_tune_stem_dbg = value;
        return CN_OK;
    }
    if (key == 41 && (value == 0 || value == 1)) {
        cn_tune_dcn_wide = value;
        return CN_OK;
    }
    if (key == 42 && value >= 1 && value <= 4096) {
        cn_tune_dcn_wide_wgs = value;
        return CN_OK;
    }
    if (key == 23 && (value == 0 || value == 1 || value == 2 || (value >= 4 && value <= 7))) {
        g_tune_dcn_form = value;
        return CN_OK;
    }
    return CN_ERR_UNSUPPORTED;
}
