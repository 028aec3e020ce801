// cn_offconv.hip -- 3x3 / stride 1 / pad 1 convolution with <= 32 output channels on a PLAIN fp32 NHWC
// tensor, f32s arithmetic, plain fp32 output: the `conv_offset_mask` of every deformable module
// (DCNv2/dcn_v2.py:52-62: Conv2d(Cin, 27, 3, 1, 1) -> 18 offsets + 9 mask logits), round 5.
//
// Why its own kernel.  These layers ran on the LDS-halo kernel's 32-wide instantiation
// (conv3x3s1_kernel<cn_f32s, 32, 32, 4, 1>): 42-51 us per launch at B = 32, matrix pipe 12-18 % busy -- 0.14 ms
// of a resdcn_18 step and ~15 % of dla_34 (sixteen of them).  With 27 outputs the layer is a READ of its input
// (67 MB for 128 channels at 64 x 64, B = 32) with 12 us of matrix work attached; the input is the plain fp32
// tensor the deformable sampler needs, so the persistent f32s kernel (cn_conv3x3p.hip) cannot take it.  Here
//   * a workgroup is eight waves on one 8 x 16 pixel tile: four pixel blocks x two teams that split the
//     (tap, chunk) steps and add their accumulators at the end (as in cn_dcn3.hip) -- 32 KB of LDS, small
//     register count: four to five workgroups per CU cover each other's window swaps;
//   * the 10 x 18 halo of a 32-channel chunk comes by LDS-DMA (unpadded 128-byte pixels, XOR-swizzled
//     through the source address); the plain floats are scaled by the input exponent and split into fp16
//     (high, low) pairs in registers, eight channels per lane and tap -- the layer has the VALU time for it;
//   * the weights (one 32-row block) are the MFMA's A operand, straight from the fragment-ordered copy;
//   * the input-side range word is fed by reading the thread's own DMA pieces back (a DMA cannot track).
// y = acc * scale + shift (conv bias inside shift), optional ReLU; K split for maps with few tiles (raw partial
// slabs, cn_conv.hip's splitk_reduce_kernel applies the epilogue).
#include "cn_common.h"

int cn_tune_offconv = 1;        // cn_set_tuning key 39: 0 = off (the LDS-halo kernel takes these layers)
int cn_tune_offconv_teams1 = 768;   // cn_set_tuning key 40: workgroups from which the four-wave form is used

// one 128-byte line of zeros: the DMA source of halo pixels outside the image
__device__ __attribute__((aligned(128))) unsigned char cn_oc_zero_line[128];

namespace {

constexpr int O_TX = 16, O_TY = 8;
constexpr int O_HW = O_TX + 2;                 // 18 halo columns (even: 18 x 128 = 9 x 256 bytes per row)
constexpr int O_HR = (O_TY + 2) * O_HW;        // 180 halo pixels
constexpr int O_HP = (O_HR + 7) / 8;           // 23 DMA pieces of 8 pixels (1 KiB)
constexpr int O_HBYTES = O_HP * 1024;          // 23552
constexpr int O_STG = 4 * 4096;                 // epilogue (two teams): team 1's 16 accumulator registers x 64 lanes x 4 pixel blocks
constexpr int O_LDS = O_STG > O_HBYTES ? O_STG : O_HBYTES;

struct OcArgs {
    const float *x;            // (B, H, W, Cin) plain fp32, pitch Cin
    const void *w;             // f32s-packed [tap][32][cin_pad] row form + the fragment-ordered copy behind it
    const float *scale, *shift;
    float *y;                  // (B, H, W, out_pitch) plain fp32
    int B, H, W, Cin, Cout, out_pitch, relu, nchunk, tiles_x, tiles_y;
    float x_mul;
    uint32_t *range;
    int ksplit;
    float *partial;            // [ksplit][B*H*W][32] raw sums
};

typedef _Float16 oc_f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) char oc_lds_char;
typedef __attribute__((address_space(1))) char oc_glb_char;
typedef __attribute__((address_space(3))) cn_f32x4 oc_lds_f32x4;
typedef __attribute__((address_space(3))) void oc_lds_void;
typedef __attribute__((address_space(1))) const void oc_glb_void;

// TEAMS = 2: eight waves, the two teams split the (tap, chunk) steps (maps with few tiles: more waves per tile);
// TEAMS = 1: four waves, every wave walks all nine taps (many tiles: more, independent workgroups per CU)
template <int TEAMS>
__global__ __launch_bounds__(256 * TEAMS, 4) void offconv_kernel(const OcArgs a)
{
    constexpr int NW = 4 * TEAMS;                  // waves
    constexpr int NPW = (O_HP + NW - 1) / NW;      // DMA pieces per wave
    extern __shared__ __attribute__((aligned(128))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int pb = wave & 3, team = wave >> 2;
    const int H = a.H, W = a.W;
    int bx = blockIdx.x;
    {   // XCD-aware tile order: contiguous tile ranges per XCD
        const int q8 = gridDim.x >> 3;
        if (bx < (q8 << 3)) bx = (bx & 7) * q8 + (bx >> 3);
    }
    const int tiles = a.tiles_x * a.tiles_y;
    const int b = bx / tiles;
    const int tr = bx - b * tiles;
    const int ty0 = (tr / a.tiles_x) * O_TY, tx0 = (tr % a.tiles_x) * O_TX;
    const unsigned pix_bytes = (unsigned)a.Cin * 4u;
    const oc_glb_char *xg = (const oc_glb_char *)a.x;
    const oc_lds_char *lds = (const oc_lds_char *)smem;
    const int cpw = a.nchunk / a.ksplit;
    const int c_lo = (int)blockIdx.z * cpw, c_hi = c_lo + cpw;
    float rng_in = 0.f;

    // ---- halo of one chunk by LDS-DMA: 23 pieces of 8 halo pixels; wave w issues pieces w, w + 8, w + 16.
    // lane = (pixel in piece, physical 16-byte slot s), which receives logical quad s ^ ((hx >> 1) & 7)
    unsigned doff[NPW];
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
        const int p = wave + NW * j;
        const int r = 8 * p + (lane >> 3), ps = lane & 7;
        const int hy = r / O_HW, hx = r - hy * O_HW;
        const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
        const bool ok = p < O_HP && r < O_HR && iy >= 0 && iy < H && ix >= 0 && ix < W;
        doff[j] = ok ? ((unsigned)(b * H + iy) * (unsigned)W + (unsigned)ix) * pix_bytes + 16u * (unsigned)(ps ^ ((hx >> 1) & 7))
                     : 0xffffffffu;
    }
    const oc_glb_char *zline = (const oc_glb_char *)cn_oc_zero_line + 16 * (lane & 7);
    auto dma = [&](int chunk) {
#pragma unroll
        for (int j = 0; j < NPW; ++j) {
            const int p = wave + NW * j;
            if (p < O_HP) {
                const oc_glb_char *src = (doff[j] != 0xffffffffu) ? xg + (doff[j] + (unsigned)chunk * 128u) : zline;
                __builtin_amdgcn_global_load_lds((oc_glb_void *)src, (oc_lds_void *)(smem + p * 1024), 16, 0, 0);
            }
        }
    };
    auto track = [&]() {
#pragma unroll
        for (int j = 0; j < NPW; ++j) {
            const int p = wave + NW * j;
            if (p < O_HP) {
                const cn_f32x4 v = *reinterpret_cast<const oc_lds_f32x4 *>(lds + p * 1024 + lane * 16);
                cn_rng_upd4(rng_in, v);
            }
        }
    };

    // this lane's pixel: tile row 2 pb + (l31 >> 4), column l31 & 15; tap (ky, kx) reads halo pixel
    // (row + ky, column + kx).  Swizzle key of the three columns, the lane's quad pair 2 h folded in.
    const int prow = 2 * pb + (l31 >> 4), pcol = l31 & 15;
    const unsigned arow = (unsigned)(prow * O_HW + pcol) * 128u;
    unsigned aswz[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) aswz[kx] = (unsigned)(((2 * h) ^ (((pcol + kx) >> 1) & 7)) << 4);
    const char *wfrag = reinterpret_cast<const char *>(a.w) + (size_t)9 * 32 * a.nchunk * 32 * 4;   // behind [9][32][cin_pad] floats
    const unsigned laneoff = (unsigned)lane * 16u;
    cn_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float xm = a.x_mul;

    // weight fragments of one tap (both K halves, high and low parts: four 16-byte loads), requested one tap
    // ahead of their use: a fragment takes ~1000 cycles from L2, a tap's own work ~300
    auto load_w = [&](oc_f16x8 (&dst)[4], int t, int chunk) {
        const char *sw = wfrag + (size_t)(t * a.nchunk + chunk) * 4096 + laneoff;
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[q] = *reinterpret_cast<const oc_f16x8 *>(sw + q * 1024);
    };
    for (int chunk = c_lo; chunk < c_hi; ++chunk) {
        constexpr int TS = TEAMS;                   // tap stride of a wave
        const int t0 = TEAMS == 2 ? ((team ^ (chunk - c_lo)) & 1) : 0;
        oc_f16x8 wq[2][4];
        // (MFMA operand hazard, DESIGN.md 3.0: the halo reads of the next K half are issued right behind this K
        // half's MFMAs and must not land in their source registers -- the operands are kept alive until then)
        oc_f16x8 keep_a = {}, keep_b = {}, keep_c = {}, keep_d = {};
        if (chunk != c_lo) __syncthreads();            // every wave is done with the previous halo
        dma(chunk);
        load_w(wq[0], t0, chunk);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (a.range) track();
        __syncthreads();
#pragma unroll
        for (int tt = 0; tt < (TEAMS == 2 ? 5 : 9); ++tt) {
            const int t = t0 + TS * tt;
            if (t < 9) {
                if (t + TS < 9) load_w(wq[(tt + 1) & 1], t + TS, chunk);
                const int ky = t / 3, kx = t - 3 * ky;
                const unsigned a0 = arow + (unsigned)((ky * O_HW + kx) * 128) + (kx == 0 ? aswz[0] : kx == 1 ? aswz[1] : aswz[2]);
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const oc_f16x8 wh = wq[tt & 1][kk], wl = wq[tt & 1][2 + kk];
                    const unsigned ad = a0 ^ ((unsigned)kk << 6);
                    cn_f32x4 va = *reinterpret_cast<const oc_lds_f32x4 *>(lds + ad);
                    cn_f32x4 vb = *reinterpret_cast<const oc_lds_f32x4 *>(lds + (ad ^ 16u));
                    asm volatile("" :: "v"(keep_a), "v"(keep_b), "v"(keep_c), "v"(keep_d));
                    va = va * xm;                          // plain input -> stored units (a power of two)
                    vb = vb * xm;
                    cn_f16x4v ha, la, hb, lb;
                    cn_split4<false>(va, ha, la);          // (beyond the fp16 range: seen by the range word, re-run by the host)
                    cn_split4<false>(vb, hb, lb);
                    const oc_f16x8 xhi = __builtin_shufflevector(ha, hb, 0, 1, 2, 3, 4, 5, 6, 7);
                    const oc_f16x8 xlo = __builtin_shufflevector(la, lb, 0, 1, 2, 3, 4, 5, 6, 7);
                    asm volatile("" :: "v"(wh), "v"(wl));  // (this tap's fragments have landed: waited for here)
                    __builtin_amdgcn_sched_barrier(0);     // every operand in registers before the first MFMA
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xhi, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xlo, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xhi, acc, 0, 0, 0);
                    keep_a = wh; keep_b = wl; keep_c = xhi; keep_d = xlo;
                }
            }
        }
    }

    // ---- epilogue: team 1 hands its sums over (lane-major through LDS), team 0 adds, applies
    // y = acc * scale + shift and stores.  acc[r]: channel (r & 3) + 8 (r >> 2) + 4 h of pixel l31.
    if (TEAMS == 2) __syncthreads();
    if (TEAMS == 2 && team == 1) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const cn_f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
            *reinterpret_cast<cn_f32x4 *>(smem + pb * 4096 + g * 1024 + lane * 16) = v;
        }
    }
    if (TEAMS == 2) __syncthreads();
    if (team == 0) {
        const int m = pb * 32 + l31;
        const size_t pix = (size_t)((b * H + ty0 + (m >> 4)) * W + tx0 + (m & 15));
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            cn_f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
            if (TEAMS == 2) v = v + *reinterpret_cast<const cn_f32x4 *>(smem + pb * 4096 + g * 1024 + lane * 16);
            const int n = 8 * g + 4 * h;
            if (a.partial) {
                *reinterpret_cast<cn_f32x4 *>(a.partial + ((size_t)blockIdx.z * ((size_t)a.B * H * W) + pix) * 32 + n) = v;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool ok = (n + e) < a.Cout;
                    const float sc = (ok && a.scale) ? a.scale[n + e] : (ok ? 1.f : 0.f);
                    const float sh = (ok && a.shift) ? a.shift[n + e] : 0.f;
                    const float tt = v[e] * sc + sh;
                    v[e] = ok ? (a.relu ? fmaxf(tt, 0.f) : tt) : 0.f;
                }
                if (n < a.out_pitch) *reinterpret_cast<cn_f32x4 *>(a.y + pix * a.out_pitch + n) = v;
            }
        }
    }
    if (a.range) cn_rng_commit(a.range, 1, rng_in * a.x_mul);
}

}  // namespace

// Does this kernel take the layer?  (the caller has checked: 3x3 / stride 1 / pad 1, NHWC both sides, f32s
// arithmetic with plain input and plain output, no residual)
bool cn_offconv_takes(int B, int H, int W, int Cin, int Cout, int in_pitch, int out_pitch, int ksplit)
{
    if (!cn_tune_offconv) return false;
    if (Cout > 32 || Cout < 1 || (Cin & 31) || in_pitch != Cin) return false;
    if ((H & 7) || (W & 15)) return false;
    if (out_pitch != 32) return false;     // the epilogue stores the whole 32-channel block (zeros behind Cout): the output must own it
    if ((size_t)B * H * W * Cin * 4 >= ((size_t)1 << 32)) return false;
    if (ksplit < 1 || (Cin / 32) % ksplit) return false;
    return true;
}

int cn_offconv_f32s(const float *x, const void *w_packed, const float *scale, const float *shift, float *y,
                    int B, int H, int W, int Cin, int Cout, int out_pitch, int relu, const cn_f32s_ctl *ctl,
                    int ksplit, float *partial, hipStream_t st)
{
    if (!cn_aligned16(x) || !cn_aligned16(y) || !cn_aligned16(w_packed)) return CN_ERR_ALIGN;
    OcArgs a = {};
    a.x = x; a.w = w_packed; a.scale = scale; a.shift = shift; a.y = y;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.out_pitch = out_pitch; a.relu = relu;
    a.nchunk = Cin / 32;
    a.tiles_x = W / O_TX; a.tiles_y = H / O_TY;
    a.x_mul = (ctl && ctl->x_mul != 0.f) ? ctl->x_mul : 1.f;
    a.range = ctl ? ctl->range : nullptr;
    a.ksplit = ksplit;
    a.partial = ksplit > 1 ? partial : nullptr;
    dim3 grid((unsigned)(B * a.tiles_x * a.tiles_y), 1, (unsigned)ksplit);
    // four-wave workgroups from three tiles per CU up (four of them fit a CU), else eight waves in two teams
    if ((long)grid.x * ksplit >= cn_tune_offconv_teams1) {
        CN_SET_MAX_LDS_ONCE(offconv_kernel<1>, O_HBYTES);
        hipLaunchKernelGGL(offconv_kernel<1>, grid, dim3(256), O_HBYTES, st, a);
    } else {
        CN_SET_MAX_LDS_ONCE(offconv_kernel<2>, O_LDS);
        hipLaunchKernelGGL(offconv_kernel<2>, grid, dim3(512), O_LDS, st, a);
    }
    CN_CHECK_LAUNCH();
    return CN_OK;
}
