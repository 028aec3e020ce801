// cn_conv16.hip -- 3x3 convolution (pad 1, stride 1 or 2) of a 16-channel NHWC map into 16 or
// 32 channels: DLA's level0 / level1 (pose_dla_dcn.py:237-240, `_make_conv_level`, 16 -> 16 at
// full resolution and 16 -> 32 at half), the only layers of the zoo whose channel counts are
// below one MFMA tile of the generic kernels (2-4x of their work was padding).
//
// K = 9 taps x 16 channels = 144 exactly: v_mfma_f32_16x16x4_f32 (M = 16 pixels, N = 16 output
// channels, K = 4), the four lane quarters taking channels 4q..4q+3 of one tap, so ONE
// ds_read_b128 per (tap, 16-pixel block) feeds four MFMAs and 16 consecutive pixels x 64 bytes
// form one contiguous, conflict-free 1 KiB LDS read.  The lane's 36 (x2) weights stay in
// registers.  Persistent workgroups (two per CU) walk 128-pixel row tiles; the 3-row input
// window of the next tile is fetched into registers during the MFMAs of the current one.
// The op is balanced between the matrix pipe and HBM (AI = 36 FLOP/B for 16 -> 16).
#include "cn_common.h"

namespace {

constexpr int NT = 256;
constexpr int BM = 128;
constexpr int CI = 16;

struct C16Args {
    const float *x, *w;  // x: NHWC pitch in_pitch; w: packed [tap][cout_pad][32]
    const float *scale, *shift;
    float *y;
    int H, W, Ho, Wo, Cout, cout_pad, in_pitch, out_pitch, relu;
};

template <int NBLK, int S>
__global__ __launch_bounds__(NT) void conv16_kernel(const C16Args a, int total_tiles)
{
    constexpr int WX = (BM - 1) * S + 3;           // window columns
    constexpr int NQ = 3 * WX * (CI / 4);          // float4 elements of the window
    constexpr int PQ = (NQ + NT - 1) / NT;         // per thread
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cn_f32x4 *win = reinterpret_cast<cn_f32x4 *>(smem);  // [3][WX][4] float4

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int tpr = a.Wo / BM;

    // weights of this lane: n = nb*16 + l15, input channels 4*lq .. 4*lq+3 of every tap
    float wreg[NBLK][9][4];
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int n = min(nb * 16 + l15, a.cout_pad - 1);
            const cn_f32x4 v = *reinterpret_cast<const cn_f32x4 *>(
                a.w + ((size_t)(t * a.cout_pad + n) * 32 + 4 * lq));
#pragma unroll
            for (int s = 0; s < 4; ++s) wreg[nb][t][s] = (nb * 16 + l15 < a.Cout) ? v[s] : 0.f;
        }
    float sc[NBLK], sf[NBLK];
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb) {
        const int n = nb * 16 + l15;
        sc[nb] = (a.scale && n < a.Cout) ? a.scale[n] : 1.f;
        sf[nb] = (a.shift && n < a.Cout) ? a.shift[n] : 0.f;
        asm volatile("" : "+v"(sc[nb]), "+v"(sf[nb]));  // settle before the tile loop
    }

    cn_f32x4 v[PQ];
    unsigned vmask = 0;
    auto prefetch = [&](int tile) {
        const int xt = tile % tpr;
        const int rowid = tile / tpr;  // b*Ho + oy
        const int b = rowid / a.Ho, oy = rowid - b * a.Ho;
        const int iy0 = oy * S - 1, ix0 = xt * BM * S - 1;
        const char *xb = reinterpret_cast<const char *>(a.x + (size_t)b * a.H * a.W * a.in_pitch);
        unsigned mk = 0;
#pragma unroll
        for (int u = 0; u < PQ; ++u) {
            const int i = tid + u * NT;
            const int row = i / (WX * 4), rem = i - row * (WX * 4);
            const int col = rem >> 2, c4 = rem & 3;
            const int iy = iy0 + row, ix = ix0 + col;
            const bool ok = i < NQ && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            const unsigned off = ok ? (unsigned)(((iy * a.W + ix) * a.in_pitch + c4 * 4) * 4) : 0u;
            v[u] = *reinterpret_cast<const cn_f32x4 *>(xb + off);
            mk |= ok ? (1u << u) : 0u;
        }
        vmask = mk;
    };
    auto store_window = [&]() {
        const cn_f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < PQ; ++u) {
            const int i = tid + u * NT;
            if (i < NQ) win[i] = ((vmask >> u) & 1u) ? v[u] : z;
        }
    };

    const float *abase[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
        abase[mb] = reinterpret_cast<const float *>(win) + (wave * 32 + mb * 16 + l15) * S * CI + 4 * lq;

    int tile = blockIdx.x;
    if (tile < total_tiles) prefetch(tile);
    for (; tile < total_tiles; tile += gridDim.x) {
        store_window();
        __syncthreads();
        const int next = tile + gridDim.x;
        if (next < total_tiles) prefetch(next);

        cn_f32x4 acc[2][NBLK];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb) acc[mb][nb] = cn_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int off = ((t / 3) * WX + (t % 3)) * CI;  // compile-time immediate
            cn_f32x4 af[2];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) af[mb] = *reinterpret_cast<const cn_f32x4 *>(abase[mb] + off);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NBLK; ++nb)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mb][s], wreg[nb][t][s],
                                                                           acc[mb][nb], 0, 0, 0);
        }
        // D: col = lane & 15 (cout), rows 4*(lane >> 4) + r (pixels)
        const int xt = tile % tpr;
        const int rowid = tile / tpr;
        float *yb = a.y + ((size_t)rowid * a.Wo + (size_t)xt * BM) * a.out_pitch;
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb) {
            const int n = nb * 16 + l15;
            if (n < a.Cout) {
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m = wave * 32 + mb * 16 + 4 * lq + r;
                        float t = acc[mb][nb][r] * sc[nb] + sf[nb];
                        if (a.relu) t = fmaxf(t, 0.f);
                        yb[(size_t)m * a.out_pitch + n] = t;
                    }
            }
        }
        __syncthreads();
    }
}

template <int NBLK, int S>
int launch_c16(const C16Args &a, int B, hipStream_t st)
{
    constexpr int WX = (BM - 1) * S + 3;
    constexpr size_t lds = (size_t)3 * WX * CI * sizeof(float);
    const long total = (long)B * a.Ho * (a.Wo / BM);
    const int wgs = (int)(total < 512 ? total : 512);  // two resident workgroups per CU
    CN_SET_MAX_LDS_ONCE((conv16_kernel<NBLK, S>), lds);
    hipLaunchKernelGGL((conv16_kernel<NBLK, S>), dim3(wgs), dim3(NT), lds, st, a, (int)total);
    CN_CHECK_LAUNCH();
    return CN_OK;
}


// ---- f32s form (round 3).  The fp32 kernel above is balanced between HBM and the fp32 matrix pipe
// (16 -> 16 at 512^2, B = 32: 77 GFLOP against 157 TFLOP/s = 0.49 ms, measured 0.48); computing the
// products as (high, low) fp16 pairs on v_mfma_f32_16x16x32_f16 -- three instructions of 16 cycles
// per 32-deep K step instead of eight of 32 cycles -- leaves the layer to its 1.07 GB of traffic.
// K = 9 taps x 16 channels is walked as five steps of two taps (the tenth half-step multiplies
// zero weights): lane group q = lane >> 4 of the 16x16x32 instruction holds k = 8q .. 8q+7, i.e.
// tap 2u + (q >> 1), channels 8 (q & 1) .. + 7.  The window sits in LDS as two fp16 planes
// (32 bytes per pixel each: 16 consecutive pixels x {half 0, half 1} = 16 distinct bank groups per
// ds_read_b128 group at stride 1), split ONCE while it is staged (x * x_mul, range word fed there);
// the lane's weights are pre-split registers (40 per 16 output channels).  Output: plain fp32.
typedef _Float16 c16_f16x8 __attribute__((ext_vector_type(8)));

struct C16sArgs {
    const float *x;
    const void *w;       // f32s-packed [tap][cout_pad][128-byte group of 32 channels: 32 high | 32 low halves]
    const float *scale, *shift;
    float *y;
    int H, W, Ho, Wo, Cout, cout_pad, in_pitch, out_pitch, relu;
    float x_mul;
    uint32_t *range;
};

template <int NBLK, int S>
__global__ __launch_bounds__(NT, 2) void conv16s_kernel(const C16sArgs a, int total_tiles)
{
    constexpr int WX = (BM - 1) * S + 3;           // window columns
    constexpr int NQ = 3 * WX * (CI / 4);          // float4 elements of the window
    constexpr int PQ = (NQ + NT - 1) / NT;         // per thread
    constexpr int PLANE = 3 * WX * 32;             // bytes of one fp16 plane
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *win_hi = smem, *win_lo = smem + PLANE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int tpr = a.Wo / BM;
    const float x_mul = a.x_mul;

    // weights of this lane: n = nb*16 + l15; step u: tap 2u + (lq >> 1), channels 8 (lq & 1) ..
    c16_f16x8 wh[NBLK][5], wl[NBLK][5];
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int t = 2 * u + (lq >> 1);
            const int n = min(nb * 16 + l15, a.cout_pad - 1);
            const char *g = reinterpret_cast<const char *>(a.w) + ((size_t)(min(t, 8) * a.cout_pad + n)) * 128 + 16 * (lq & 1);
            const bool ok = t < 9 && nb * 16 + l15 < a.Cout;
            const c16_f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            wh[nb][u] = ok ? *reinterpret_cast<const c16_f16x8 *>(g) : z;
            wl[nb][u] = ok ? *reinterpret_cast<const c16_f16x8 *>(g + 64) : z;
        }
    float sc[NBLK], sf[NBLK];
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb) {
        const int n = nb * 16 + l15;
        sc[nb] = (a.scale && n < a.Cout) ? a.scale[n] : 1.f;
        sf[nb] = (a.shift && n < a.Cout) ? a.shift[n] : 0.f;
        asm volatile("" : "+v"(sc[nb]), "+v"(sf[nb]));  // settle before the tile loop
    }

    cn_f32x4 v[PQ];
    unsigned vmask = 0;
    float rng_in = 0.f;
    auto prefetch = [&](int tile) {
        const int xt = tile % tpr;
        const int rowid = tile / tpr;  // b*Ho + oy
        const int b = rowid / a.Ho, oy = rowid - b * a.Ho;
        const int iy0 = oy * S - 1, ix0 = xt * BM * S - 1;
        const char *xb = reinterpret_cast<const char *>(a.x + (size_t)b * a.H * a.W * a.in_pitch);
        unsigned mk = 0;
#pragma unroll
        for (int u = 0; u < PQ; ++u) {
            const int i = tid + u * NT;
            const int row = i / (WX * 4), rem = i - row * (WX * 4);
            const int col = rem >> 2, c4 = rem & 3;
            const int iy = iy0 + row, ix = ix0 + col;
            const bool ok = i < NQ && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            const unsigned off = ok ? (unsigned)(((iy * a.W + ix) * a.in_pitch + c4 * 4) * 4) : 0u;
            v[u] = *reinterpret_cast<const cn_f32x4 *>(xb + off);
            mk |= ok ? (1u << u) : 0u;
        }
        vmask = mk;
    };
    auto store_window = [&]() {
        const cn_f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < PQ; ++u) {
            const int i = tid + u * NT;
            if (i < NQ) {
                const cn_f32x4 xs = ((vmask >> u) & 1u) ? v[u] * x_mul : z;   // real -> stored units
                cn_rng_upd4(rng_in, xs);
                cn_f16x4v hi, lo;
                cn_split4(xs, hi, lo);
                // element i = (pixel i >> 2, channel quad i & 3): 8 bytes in each plane
                *reinterpret_cast<cn_f16x4v *>(win_hi + (size_t)i * 8) = hi;
                *reinterpret_cast<cn_f16x4v *>(win_lo + (size_t)i * 8) = lo;
            }
        }
    };

    // A fragment of (16-pixel block mb, step u): pixel + tap offset, 16 bytes of half (lq & 1)
    unsigned aoff[2][5];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int t = min(2 * u + (lq >> 1), 8);   // (the tenth half-step re-reads tap 8: its weights are zero)
            aoff[mb][u] = (unsigned)(((wave * 32 + mb * 16 + l15) * S + (t / 3) * WX + (t % 3)) * 32 + 16 * (lq & 1));
        }

    int tile = blockIdx.x;
    if (tile < total_tiles) prefetch(tile);
    for (; tile < total_tiles; tile += gridDim.x) {
        store_window();
        __syncthreads();
        const int next = tile + gridDim.x;
        if (next < total_tiles) prefetch(next);

        cn_f32x4 acc[2][NBLK];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb) acc[mb][nb] = cn_f32x4{0.f, 0.f, 0.f, 0.f};
        c16_f16x8 ah[2][5], al[2][5];
#pragma unroll
        for (int u = 0; u < 5; ++u)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                ah[mb][u] = *reinterpret_cast<const c16_f16x8 *>(win_hi + aoff[mb][u]);
                al[mb][u] = *reinterpret_cast<const c16_f16x8 *>(win_lo + aoff[mb][u]);
            }
        // every fragment is in registers before the first MFMA (operand hazard note, cn_conv.hip)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 5; ++u)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < NBLK; ++nb) {
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mb][u], wh[nb][u], acc[mb][nb], 0, 0, 0);
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mb][u], wl[nb][u], acc[mb][nb], 0, 0, 0);
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mb][u], wh[nb][u], acc[mb][nb], 0, 0, 0);
                }
        // D: col = lane & 15 (cout), rows 4*(lane >> 4) + r (pixels)
        const int xt = tile % tpr;
        const int rowid = tile / tpr;
        float *yb = a.y + ((size_t)rowid * a.Wo + (size_t)xt * BM) * a.out_pitch;
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb) {
            const int n = nb * 16 + l15;
            if (n < a.Cout) {
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m = wave * 32 + mb * 16 + 4 * lq + r;
                        float t = acc[mb][nb][r] * sc[nb] + sf[nb];
                        if (a.relu) t = fmaxf(t, 0.f);
                        yb[(size_t)m * a.out_pitch + n] = t;
                    }
            }
        }
        __syncthreads();
    }
    if (a.range) cn_rng_commit(a.range, 1, rng_in);
}

template <int NBLK, int S>
int launch_c16s(const C16sArgs &a, int B, hipStream_t st)
{
    constexpr int WX = (BM - 1) * S + 3;
    constexpr size_t lds = (size_t)2 * 3 * WX * 32;
    const long total = (long)B * a.Ho * (a.Wo / BM);
    const int wgs = (int)(total < 512 ? total : 512);  // two resident workgroups per CU
    CN_SET_MAX_LDS_ONCE((conv16s_kernel<NBLK, S>), lds);
    hipLaunchKernelGGL((conv16s_kernel<NBLK, S>), dim3(wgs), dim3(NT), lds, st, a, (int)total);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

}  // namespace

// Returns CN_ERR_UNSUPPORTED when the layer is not of this form (the caller then takes the
// LDS-halo / generic kernels).
int cn_conv3x3_c16(const float *x, const float *w_packed, const float *scale, const float *shift,
                   float *y, int B, int H, int W, int Ho, int Wo, int Cin, int Cout, int stride,
                   int in_pitch, int out_pitch, int relu, hipStream_t st)
{
    if (Cin != CI || Cout > 32 || (stride != 1 && stride != 2) || Wo % BM != 0 || (in_pitch & 3))
        return CN_ERR_UNSUPPORTED;
    if (Ho != (H + 2 - 3) / stride + 1 || Wo != (W + 2 - 3) / stride + 1) return CN_ERR_UNSUPPORTED;
    if ((long)H * W * in_pitch >= (1L << 29)) return CN_ERR_UNSUPPORTED;  // 32-bit byte offsets per image
    C16Args a;
    a.x = x; a.w = w_packed; a.scale = scale; a.shift = shift; a.y = y;
    a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout; a.cout_pad = (Cout + 31) / 32 * 32;
    a.in_pitch = in_pitch; a.out_pitch = out_pitch; a.relu = relu;
    if (Cout <= 16)
        return stride == 1 ? launch_c16<1, 1>(a, B, st) : launch_c16<1, 2>(a, B, st);
    return stride == 1 ? launch_c16<2, 1>(a, B, st) : launch_c16<2, 2>(a, B, st);
}

// f32s form: plain fp32 input (split while it is staged: ctl->x_mul, range word side 1), f32s-packed
// weight (cn_pack_conv_weight, CN_DTYPE_F32S: 128-byte groups, channels 16..31 zero), plain fp32
// output.  Stride 1 only (see below).
int cn_conv3x3_c16s(const float *x, const void *w_packed, const float *scale, const float *shift,
                    float *y, int B, int H, int W, int Ho, int Wo, int Cin, int Cout, int stride,
                    int in_pitch, int out_pitch, int relu, const cn_f32s_ctl *ctl, hipStream_t st)
{
    if (Cin != CI || Cout > 32 || (stride != 1 && stride != 2) || Wo % BM != 0 || (in_pitch & 3))
        return CN_ERR_UNSUPPORTED;
    if (Ho != (H + 2 - 3) / stride + 1 || Wo != (W + 2 - 3) / stride + 1) return CN_ERR_UNSUPPORTED;
    if ((long)H * W * in_pitch >= (1L << 29)) return CN_ERR_UNSUPPORTED;  // 32-bit byte offsets per image
    if (!cn_aligned16(x) || !cn_aligned16(w_packed)) return CN_ERR_UNSUPPORTED;
    C16sArgs a;
    a.x = x; a.w = w_packed; a.scale = scale; a.shift = shift; a.y = y;
    a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout; a.cout_pad = (Cout + 31) / 32 * 32;
    a.in_pitch = in_pitch; a.out_pitch = out_pitch; a.relu = relu;
    a.x_mul = (ctl && ctl->x_mul != 0.f) ? ctl->x_mul : 1.f;
    a.range = ctl ? ctl->range : nullptr;
    // stride 2 (level1) was measured slower in this form than on the fp32 kernel (0.476 vs 0.245 ms:
    // a 257-column window = 52 staging registers next to 80 fragment + 80 weight registers, 2-way
    // bank conflicts at the 64-byte pixel stride): not built
    if (stride != 1) return CN_ERR_UNSUPPORTED;
    return Cout <= 16 ? launch_c16s<1, 1>(a, B, st) : launch_c16s<2, 1>(a, B, st);
}
