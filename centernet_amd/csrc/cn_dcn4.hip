// cn_dcn4.hip -- fused modulated deformable convolution (DCNv2) forward, f32s arithmetic: the WIDE
// form (round 6): one workgroup owns a pixel tile and ALL of a layer's output channels (128 or 256),
// so every bilinear sample of the tile is formed exactly once.
//
// Replaces: DCN.forward -> DCNv2Function.forward -> dcn_v2_cuda_forward
//   (DCNv2/dcn_v2.py:64-70, dcn_v2_func.py:22-38, src/dcn_v2_cuda.c:10-102): per sample a bias
//   SGEMM, modulated_deformable_im2col_gpu_kernel (src/cuda/dcn_v2_im2col_cuda.cu:118-180, bilinear
//   sampler :18-47) writing a Cin*9*HW column buffer, and the main SGEMM.
//
// Why another form.  The team form (cn_dcn3.hip) gives a workgroup 64 output channels: a layer with
// Cout = 256 rebuilds every sample four times (Cout = 128: twice), and the sampler -- not the matrix
// pipe -- is what a deformable step costs on this part (~220 VALU cycles per 16-channel
// K half of a 32-pixel block against 192 matrix cycles for 64 output channels).  Here
//   * a workgroup is eight waves on ONE 8 x 16 pixel tile: four pixel blocks x two K-HALF teams.
//     Wave (pb, kk) samples channels 16 kk .. 16 kk + 15 of every (tap, 32-channel chunk) step for
//     its 32 pixels -- the sample IS the 32x32x16 MFMA's B operand, as in the team form -- and
//     multiplies it into NB = 4 or 8 blocks of 32 output channels (12 / 24 MFMAs per sample set
//     instead of 6): every (pixel, tap, channel) sample is computed once per tile;
//   * the weights of a step (NB x 4 KiB of ready-made fragments) can no longer come through L1 --
//     four pixel-block waves x two teams would pull 64-128 KiB per step -- so they go through a
//     two-slot LDS ring filled by LDS-DMA one step ahead (contiguous 1 KiB pieces of the
//     fragment-ordered copy: no VALU, no staging registers) and are read conflict-free;
//   * one barrier per step hands the ring slot over; the two teams add their accumulators in the
//     epilogue;
//   * window, records, far path, range tracking: as in the team form.
// LDS: window 48 KiB + records 27 KiB + ring 2 x NB x 4 KiB = 109 / 142 KiB (NB = 4 adds a second window:
// 157 KiB): one workgroup per CU, two waves per SIMD, up to 256 registers (NB = 8: 128 accumulator
// registers).  With two waves per SIMD nothing but the partner wave covers a round trip, so the step loop is a
// written-out schedule: the two teams of a pixel block share a SIMD and run in anti-phase, every request
// (record, window reads, far corners) is issued a whole MFMA phase before its blend, team 0 alone owns the
// weight DMA (see the loop; DESIGN.md 3.2 "Round 6" has the measurements that led there).
// Semantics held: sampling domain h_im > -1 && w_im > -1 && h_im < H && w_im < W
// (dcn_v2_im2col_cuda.cu:165), corner rule (:30-41), weights hh*hw, hh*lw, lh*hw, lh*lw (:26-28,43),
// value * mask (:174; the mask multiplies the four corner weights), bias then accumulate
// (dcn_v2_cuda.c:61-97).
#include "cn_common.h"

int cn_tune_dcn_wide = 1;        // cn_set_tuning key 41: 0 = off, 1 = layers with Cout % 128 == 0 that the form takes
int cn_tune_dcn_wide_prefetch = 1;   // cn_set_tuning key 45: L2 prefetch of the weight slabs three steps ahead
int cn_tune_dcn_wide_wgs = 256;  // cn_set_tuning key 42: K split until a launch has this many workgroups

__device__ __attribute__((aligned(128))) unsigned char cn_d4_zero_line[128];
// probe build, key 9 bit 512: cycle stamps of waves 0 and 4 of workgroup 0 over its first 64 steps
__device__ unsigned long long cn_d4_trace[2 * 64 * 8];

namespace {

constexpr int W_NT = 512;                      // 8 waves: 4 pixel blocks x 2 K-half teams
constexpr int W_TX = 16, W_TY = 8, W_PM = W_TX * W_TY;
constexpr int W_RCH = 3;                       // offsets up to +-3 px sample inside the window
constexpr int W_WX = W_TX + 2 + 2 * W_RCH;     // 24
constexpr int W_WY = W_TY + 2 + 2 * W_RCH;     // 16
constexpr int W_WPIX = W_WX * W_WY;            // 384
constexpr int W_PIXB = 128;                    // bytes per window pixel: 32 plain floats, unpadded
constexpr int W_ROWB = W_WX * W_PIXB;          // 3072 = 12 x 256: a row starts on bank group 0
constexpr int W_WBYTES = W_WPIX * W_PIXB;      // 49152
constexpr int W_NP = W_WPIX * 8 / W_NT;        // 6 DMA pieces (16 B) per thread and chunk
constexpr int W_RECW = W_WBYTES;               // float4 [9][128]: corner weights (mask, exponent, validity folded in)
constexpr int W_RECP = W_RECW + 9 * W_PM * 16; // uint2 [9][128]: swizzled LDS offsets of corners 1 and 2 | far flag + corner
constexpr int W_EPI = W_RECP + 9 * W_PM * 8;   // 76800: float [3][256]: bias, scale, shift of the workgroup's output channels
constexpr int W_RING = W_EPI + 3 * 256 * 4;    // 79872: two slots of NB x 4096 bytes
constexpr int W_LDC = 68;                      // floats per staged pixel row (64 + 4)
constexpr int W_STG = 32 * W_LDC * 4;          // 8704 bytes per wave (8 waves: 69632 <= W_EPI)
static_assert(8 * W_STG <= W_EPI, "epilogue strips alias the window and the records only");
static_assert(W_ROWB % 256 == 0, "window rows keep the bank-group phase");

struct D4Args {
    const float *x;            // (B, H, W, Cin) plain fp32
    const void *w;             // f32s-packed [tap][cout_pad][cin_pad] row form + the fragment-ordered copy behind it
    const float *bias, *scale, *shift, *om;
    void *y;
    int B, H, W, Cin, Cout, om_pitch, relu;
    int cin_pad, cout_pad, nchunk, tiles_x, tiles_y, out_pitch, out_plain;
    float x_mul;
    uint32_t *range;
    int ksplit;                // K-chunk ranges per tile (blockIdx.z); > 1: raw partial sums
    float *partial;            // [ksplit][B*H*W][cout_pad] fp32 (splitk_reduce_kernel applies the epilogue)
    int prefetch;              // 1 = L2 prefetch of the weights three steps ahead (cn_set_tuning key 45, default 1)
    int dbg;                   // probe build (cn_set_tuning key 9): 1 = no weight DMA, 2 = no MFMAs, 4 = no sampling, 8 = no per-step barrier, 16 = no far path, 32 = no steps, 64 = no window swaps, 128 = no epilogue
};

typedef _Float16 d4_f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) char d4_lds_char;
typedef __attribute__((address_space(1))) char d4_glb_char;
typedef __attribute__((address_space(3))) cn_f32x4 d4_lds_f32x4;
typedef __attribute__((address_space(1))) cn_f32x4 d4_glb_f32x4;
typedef __attribute__((address_space(3))) d4_f16x8 d4_lds_f16x8;
typedef __attribute__((address_space(3))) void d4_lds_void;
typedef __attribute__((address_space(1))) const void d4_glb_void;
typedef float d4_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned d4_u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void d4_barrier()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// swizzled LDS byte offsets of window pixels (wy, wx) and (wy, wx + 1), quad 0 of lane half 0:
// physical 16-byte slot of logical quad q of a pixel = q ^ ((wx >> 1) & 7)
__device__ __forceinline__ unsigned d4_enc(int wy, int wx)
{
    const unsigned p = (unsigned)(wy * W_WX + wx);
    const unsigned q1 = p * W_PIXB + ((((unsigned)wx >> 1) & 7u) << 4);
    const unsigned q2 = (p + 1u) * W_PIXB + (((((unsigned)wx + 1u) >> 1) & 7u) << 4);
    return q1 | (q2 << 16);
}

// NB:   blocks of 32 output channels per workgroup (4 or 8)
// MSIG: the mask is sigmoid(conv output) (dcn_v2.py:67), hence in [0, 1]: a sample is a convex blend of
//       window values times <= 1 and needs neither clamp nor range tracking of its own; false = a
//       caller-supplied mask of any size (clamp + track per sample)
template <int NB, bool MSIG, bool DBG>
__global__ __launch_bounds__(W_NT, 2) void dcn_wide_kernel(const D4Args a)
{
    extern __shared__ __attribute__((aligned(128))) char smem[];
    constexpr int SLOT = NB * 4096;                // bytes of one step's weight fragments
    // NB = 4 leaves room for a SECOND window behind the ring: the next chunk's window is copied one piece per
    // thread and step while the current one is sampled, and a chunk boundary costs one barrier instead of a
    // barrier + 48 KiB DMA round trip + barrier
    constexpr bool WDB = (NB == 4);
    constexpr int W_WIN1 = W_RING + 2 * SLOT;
    constexpr int NPW = NB / 2;                    // 1 KiB DMA pieces per wave and step
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int pb = wave & 3, kk = wave >> 2;
    const int H = a.H, W = a.W;
    int bx = blockIdx.x;
    {   // XCD-aware tile order: contiguous tile ranges per XCD (block b runs on XCD b % 8)
        const int q8 = gridDim.x >> 3;
        if (bx < (q8 << 3)) bx = (bx & 7) * q8 + (bx >> 3);
    }
    const int tiles = a.tiles_x * a.tiles_y;
    const int b = bx / tiles;
    const int tr = bx - b * tiles;
    const int ty0 = (tr / a.tiles_x) * W_TY, tx0 = (tr % a.tiles_x) * W_TX;
    const int wy0 = ty0 - 1 - W_RCH, wx0 = tx0 - 1 - W_RCH;
    const int n0 = (int)blockIdx.y * (32 * NB);
    const int dbg = DBG ? a.dbg : 0;
    const unsigned pix_bytes = (unsigned)a.Cin * 4u;
    const unsigned img_base = (unsigned)(b * H) * (unsigned)W;
    const d4_glb_char *xg = (const d4_glb_char *)a.x;
    const d4_lds_char *lds = (const d4_lds_char *)smem;
    float rng_in = 0.f, rng_out = 0.f;
    // K split (small maps with deep K): this workgroup's share of the 32-channel chunks
    const int cpw = a.nchunk / a.ksplit;
    const int c_lo = (int)blockIdx.z * cpw, c_hi = c_lo + cpw;

    // ---- window of one chunk by LDS-DMA: 3072 16-byte pieces, six per thread; piece i = pixel i >> 3,
    // physical slot i & 7, which receives the pixel's logical quad (i & 7) ^ ((wx >> 1) & 7)
    unsigned doff[W_NP];            // byte offset of the piece's source in x (chunk 0), ~0 = zero line
#pragma unroll
    for (int p = 0; p < W_NP; ++p) {
        const int i = p * W_NT + tid;
        const int wp = i >> 3, pq = i & 7;
        const int wy = wp / W_WX, wx = wp - wy * W_WX;
        const int lq = pq ^ ((wx >> 1) & 7);
        const int iy = wy0 + wy, ix = wx0 + wx;
        const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
        doff[p] = ok ? (img_base + (unsigned)(iy * W + ix)) * pix_bytes + 16u * (unsigned)lq : 0xffffffffu;
    }
    const d4_glb_char *zline = (const d4_glb_char *)cn_d4_zero_line + 16 * (lane & 7);
    unsigned wbase = 0;             // byte offset of the window being sampled (0 or W_WIN1)
    auto dma1 = [&](int chunk, int p, unsigned base) {
        const unsigned cb = (unsigned)chunk * 128u;
        const d4_glb_char *src = (doff[p] != 0xffffffffu) ? xg + (doff[p] + cb) : zline;
        __builtin_amdgcn_global_load_lds((d4_glb_void *)src, (d4_lds_void *)(smem + base + (p * W_NT + wave * 64) * 16), 16, 0, 0);
    };
    auto dma = [&](int chunk) {
#pragma unroll
        for (int p = 0; p < W_NP; ++p) dma1(chunk, p, wbase);
    };
    auto dma_piece = [&](int chunk, int t, unsigned base) {   // piece t (uniform): scalar branches, doff stays in registers
        static_assert(W_NP == 6, "six pieces per thread");
        switch (t) {
        case 0: dma1(chunk, 0, base); break;
        case 1: dma1(chunk, 1, base); break;
        case 2: dma1(chunk, 2, base); break;
        case 3: dma1(chunk, 3, base); break;
        case 4: dma1(chunk, 4, base); break;
        case 5: dma1(chunk, 5, base); break;
        default: break;
        }
    };
    auto track = [&]() {
#pragma unroll
        for (int p = 0; p < W_NP; ++p) {
            const cn_f32x4 v = *reinterpret_cast<const d4_lds_f32x4 *>(lds + wbase + (p * W_NT + tid) * 16);
            cn_rng_upd4(rng_in, v);
        }
    };
    // ---- weight fragments of one (tap, chunk) step: NB x 4 KiB, contiguous in the fragment copy
    // ([tap][chunk][block][quarter][lane] x 16 bytes), copied as 1 KiB pieces into ring slot `slot`
    const d4_glb_char *wfrag = (const d4_glb_char *)a.w + (size_t)9 * a.cout_pad * a.cin_pad * 4;
    const int ncb = a.cout_pad >> 5;
    auto dma_w = [&](int t, int chunk, int slot) {          // all eight waves: NB / 2 pieces each (prologue)
        const d4_glb_char *src = wfrag + ((size_t)(t * a.nchunk + chunk) * ncb + (n0 >> 5)) * 4096 + (unsigned)lane * 16u;
#pragma unroll
        for (int j = 0; j < NPW; ++j) {
            const int piece = j * 8 + wave;
            __builtin_amdgcn_global_load_lds((d4_glb_void *)(src + piece * 1024),
                                             (d4_lds_void *)(smem + W_RING + slot * SLOT + piece * 1024), 16, 0, 0);
        }
    };
    auto dma_w1 = [&](int t, int chunk, int slot) {         // one team's four waves: NB pieces each (step loop)
        const d4_glb_char *src = wfrag + ((size_t)(t * a.nchunk + chunk) * ncb + (n0 >> 5)) * 4096 + (unsigned)lane * 16u;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int piece = j * 4 + pb;
            __builtin_amdgcn_global_load_lds((d4_glb_void *)(src + piece * 1024),
                                             (d4_lds_void *)(smem + W_RING + slot * SLOT + piece * 1024), 16, 0, 0);
        }
    };

    // ---- prologue: offsets / masks of the tile, the window of the first chunk and the first step's
    // weights behind them, then the records (dcn_v2_im2col_cuda.cu:151-176; cn_dcn3.hip for the format)
    {
        constexpr int NR = (9 * W_PM + W_NT - 1) / W_NT;   // 3 (the last trip a quarter full)
        float off_h[NR], off_w[NR], mkv[NR];
#pragma unroll
        for (int p = 0; p < NR; ++p) {
            const int i = min(p * W_NT + tid, 9 * W_PM - 1);
            const int tap = i >> 7, m = i & (W_PM - 1);
            const int oy = ty0 + (m >> 4), ox = tx0 + (m & 15);
            const float *om = a.om + (size_t)((b * H + oy) * W + ox) * a.om_pitch;
            off_h[p] = om[2 * tap];
            off_w[p] = om[2 * tap + 1];
            mkv[p] = om[18 + tap];
        }
        float epv[2] = {0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int i = e * W_NT + tid;                   // 768 words: [3][256]
            if (i < 768) {
                const int which = i >> 8, cn = n0 + (i & 255);
                const float *src = which == 0 ? a.bias : (which == 1 ? a.scale : a.shift);
                epv[e] = which == 1 ? 1.f : 0.f;
                if (src && (i & 255) < 32 * NB && cn < a.Cout) epv[e] = src[cn];
            }
        }
        dma(c_lo);
        dma_w(0, c_lo, 0);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int i = e * W_NT + tid;
            if (i < 768) reinterpret_cast<float *>(smem + W_EPI)[i] = epv[e];
        }
#pragma unroll
        for (int p = 0; p < NR; ++p) {
            const int i = p * W_NT + tid;
            const int tap = i >> 7, m = i & (W_PM - 1);
            const int oy = ty0 + (m >> 4), ox = tx0 + (m & 15);
            float mk = mkv[p];
            if (MSIG) mk = sigmoidf_ref(mk);            // dcn_v2.py:67
            mk *= a.x_mul;                              // plain input -> stored units (a power of two)
            const int ki = tap / 3, kj = tap - ki * 3;
            const float h_im = (float)(oy - 1 + ki) + off_h[p];
            const float w_im = (float)(ox - 1 + kj) + off_w[p];
            cn_f32x4 wv = {0.f, 0.f, 0.f, 0.f};
            unsigned p0 = d4_enc((m >> 4) + 1 + W_RCH, (m & 15) + 1 + W_RCH), p1 = 0u;
            if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {   // :165
                const float hf = floorf(h_im), wf = floorf(w_im);
                const int yl = (int)hf, xl = (int)wf;
                const float lh = h_im - hf, lw = w_im - wf;
                const float hh = 1.f - lh, hw = 1.f - lw;
                const bool yl_ok = yl >= 0, xl_ok = xl >= 0;
                const bool yh_ok = yl + 1 <= H - 1, xh_ok = xl + 1 <= W - 1;
                wv[0] = (yl_ok && xl_ok) ? hh * hw * mk : 0.f;      // :30-45
                wv[1] = (yl_ok && xh_ok) ? hh * lw * mk : 0.f;
                wv[2] = (yh_ok && xl_ok) ? lh * hw * mk : 0.f;
                wv[3] = (yh_ok && xh_ok) ? lh * lw * mk : 0.f;
                const int wyl = yl - wy0, wxl = xl - wx0;
                const bool inwin = (unsigned)wyl <= (unsigned)(W_WY - 2) && (unsigned)wxl <= (unsigned)(W_WX - 2);
                if (inwin) p0 = d4_enc(wyl, wxl);
                else p1 = 0x80000000u | ((unsigned)(yl + 1) << 15) | (unsigned)(xl + 1);
            }
            if (i < 9 * W_PM) {
                *reinterpret_cast<cn_f32x4 *>(smem + W_RECW + i * 16) = wv;
                *reinterpret_cast<d4_u32x2 *>(smem + W_RECP + i * 8) = d4_u32x2{p0, p1};
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MSIG && a.range) track();
    __syncthreads();                               // records, first window, first weights visible

    const int m = pb * 32 + l31;                   // this lane's pixel of the tile
    const unsigned hx = (unsigned)h << 5;
    const unsigned kx = (unsigned)kk << 6;
    const unsigned laneoff = (unsigned)lane * 16u;
    cn_f32x16 acc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // ---- the step loop.  A step of a wave is sample (record -> eight window reads or eight L2 loads -> blend ->
    // split: ~45 VALU instructions behind two round trips) and multiply (NB x 3 MFMAs); at two waves per SIMD
    // nothing but the partner wave covers a round trip, and the per-step barrier keeps the partners in step.
    // So the two K-half teams -- waves w and w + 4 share a SIMD -- run in ANTI-PHASE and every request is
    // issued a whole MFMA phase before its blend:
    //   team 0:  barrier | weight DMA(t + 1) | MFMA(t) | blend(t + 1) | vmcnt(0) | request(t + 2)
    //   team 1:  barrier | blend(t) | request(t + 1) | MFMA(t)
    // Team 0 copies the whole weight slab and waits for it behind its blend, a whole MFMA phase later (a
    // vmcnt(0) in front of the barrier would wait for the far corners just requested); team 1 never has a DMA
    // in flight.  Records are read one
    // request ahead (no dependent LDS round trip inside a request).
    cn_f32x4 c1a, c1b, c2a, c2b, c3a, c3b, c4a, c4b, wv, wv_n;
    d4_u32x2 pp_n;
    d4_f16x8 shi = {}, slo = {};
    d4_f16x8 fh[2][2] = {}, fl[2][2] = {};
    unsigned far_base = 0;
    auto record = [&](int t) {
        wv_n = *reinterpret_cast<const d4_lds_f32x4 *>(lds + W_RECW + (t * W_PM + m) * 16);
        pp_n = *reinterpret_cast<const __attribute__((address_space(3))) d4_u32x2 *>(lds + W_RECP + (t * W_PM + m) * 8);
    };
    // request(t): corners of step t (its record was read by record(t)); reads the record of step t + 1
    auto request = [&](int t) {
        wv = wv_n;
        const d4_u32x2 pp = pp_n;
        if (t < 8) record(t + 1);
        const unsigned B1 = ((pp[0] & 0xffffu) ^ hx ^ kx) + wbase, B2 = ((pp[0] >> 16) ^ hx ^ kx) + wbase;
        if ((int)pp[1] < 0 && !(dbg & 16)) {
            // beyond the window's reach: the four corners from global memory (clamped addresses --
            // off-map corners carry zero weight); 24-bit integer multiplies (checked by the launcher)
            const int yl = (int)((pp[1] >> 15) & 0x7fffu) - 1, xl = (int)(pp[1] & 0x7fffu) - 1;
            const int y0 = max(yl, 0), y1 = min(yl + 1, H - 1);
            const int x0 = max(xl, 0), x1 = min(xl + 1, W - 1);
            const unsigned r0 = __umul24((unsigned)y0, (unsigned)W), r1 = __umul24((unsigned)y1, (unsigned)W);
            const unsigned o1 = __umul24(r0 + (unsigned)x0, pix_bytes) + far_base;
            const unsigned o2 = __umul24(r0 + (unsigned)x1, pix_bytes) + far_base;
            const unsigned o3 = __umul24(r1 + (unsigned)x0, pix_bytes) + far_base;
            const unsigned o4 = __umul24(r1 + (unsigned)x1, pix_bytes) + far_base;
            c1a = *reinterpret_cast<const d4_glb_f32x4 *>(xg + o1);
            c1b = *reinterpret_cast<const d4_glb_f32x4 *>(xg + o1 + 16);
            c2a = *reinterpret_cast<const d4_glb_f32x4 *>(xg + o2);
            c2b = *reinterpret_cast<const d4_glb_f32x4 *>(xg + o2 + 16);
            c3a = *reinterpret_cast<const d4_glb_f32x4 *>(xg + o3);
            c3b = *reinterpret_cast<const d4_glb_f32x4 *>(xg + o3 + 16);
            c4a = *reinterpret_cast<const d4_glb_f32x4 *>(xg + o4);
            c4b = *reinterpret_cast<const d4_glb_f32x4 *>(xg + o4 + 16);
        } else {
            // channels 16 kk + 8 h .. + 7 of the four corners: eight window reads
            c1a = *reinterpret_cast<const d4_lds_f32x4 *>(lds + B1);
            c1b = *reinterpret_cast<const d4_lds_f32x4 *>(lds + (B1 ^ 16u));
            c2a = *reinterpret_cast<const d4_lds_f32x4 *>(lds + B2);
            c2b = *reinterpret_cast<const d4_lds_f32x4 *>(lds + (B2 ^ 16u));
            c3a = *reinterpret_cast<const d4_lds_f32x4 *>(lds + B1 + W_ROWB);
            c3b = *reinterpret_cast<const d4_lds_f32x4 *>(lds + (B1 ^ 16u) + W_ROWB);
            c4a = *reinterpret_cast<const d4_lds_f32x4 *>(lds + B2 + W_ROWB);
            c4b = *reinterpret_cast<const d4_lds_f32x4 *>(lds + (B2 ^ 16u) + W_ROWB);
        }
        // the previous MFMA block's operands stay allocated until these reads have been issued (operand
        // hazard note, DESIGN.md 3.0 / cn_dcn3.hip)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" :: "v"(fh[0][0]), "v"(fh[0][1]), "v"(fl[0][0]), "v"(fl[0][1]), "v"(fh[1][0]), "v"(fh[1][1]), "v"(fl[1][0]), "v"(fl[1][1]));
        __builtin_amdgcn_sched_barrier(0);
    };
    // w1*v1 + w2*v2 + w3*v3 + w4*v4 (dcn_v2_im2col_cuda.cu:43-45; mask and exponent inside the weights), split
    auto blend = [&]() {
        const d4_f32x2 w1 = {wv[0], wv[0]}, w2 = {wv[1], wv[1]}, w3 = {wv[2], wv[2]}, w4 = {wv[3], wv[3]};
        cn_f32x4 va, vb;
        {
// plain v_fma_f32 (this file is built with -fno-slp-vectorize): v_pk_fma_f32 issues through the matrix pipe's
            // slot and does not overlap the partner wave's MFMAs (5 % on every shape, profiles/r06_dcn_wide.txt)
            #pragma unroll
            for (int e = 0; e < 4; ++e) {
                va[e] = __builtin_fmaf(c4a[e], w4[0], __builtin_fmaf(c3a[e], w3[0], __builtin_fmaf(c2a[e], w2[0], c1a[e] * w1[0])));
                vb[e] = __builtin_fmaf(c4b[e], w4[0], __builtin_fmaf(c3b[e], w3[0], __builtin_fmaf(c2b[e], w2[0], c1b[e] * w1[0])));
            }
        }
        cn_f16x4v ha, la, hb, lb;
        if (MSIG) {
            cn_split4<false>(va, ha, la);        // |v| <= max |x'|, which the range word reports
            cn_split4<false>(vb, hb, lb);
        } else {
            cn_rng_upd4(rng_in, va);             // caller-supplied mask: any size
            cn_rng_upd4(rng_in, vb);
            cn_split4<true>(va, ha, la);
            cn_split4<true>(vb, hb, lb);
        }
        shi = __builtin_shufflevector(ha, hb, 0, 1, 2, 3, 4, 5, 6, 7);
        slo = __builtin_shufflevector(la, lb, 0, 1, 2, 3, 4, 5, 6, 7);
    };

    uint32_t pf = 0;                               // L2 prefetch word of team 1 (see the step loop)
    int step = 0;                                  // linear (chunk, tap) step: ring slot = step & 1
    for (int chunk = c_lo; chunk < c_hi; ++chunk) {
        if (chunk != c_lo && !(dbg & 64)) {
            if (WDB) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own pieces of this chunk's window (copied under the previous chunk)
                __syncthreads();                   // everybody's; every wave is done with the previous window
                wbase ^= (unsigned)W_WIN1;
                if (MSIG && a.range) track();
            } else {
                __syncthreads();                   // every wave is done with the previous window
                dma(chunk);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the window and this step's weights (issued a step ago)
                if (MSIG && a.range) track();
                __syncthreads();
            }
        }
        // byte offset, in x, of channel 16 kk + 8 h of this chunk in pixel 0 of the image (global path)
        far_base = img_base * pix_bytes + (unsigned)chunk * 128u + hx + kx;
        // pipeline fill of the chunk: team 0 holds step 0's operand and step 1's corners, team 1 step 0's corners
        if (!(dbg & 4)) {
            record(0);
            request(0);
            if (kk == 0) {
                blend();
                request(1);
            }
        }
#pragma unroll 1
        for (int t = 0; t < ((dbg & 32) ? 0 : 9); ++t, ++step) {
            unsigned long long ts[8] = {};
            const bool tr_on = DBG && (dbg & 512) && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && pb == 0 && step < 64;
            if (tr_on) ts[0] = __builtin_readcyclecounter();
            if (t != 0) {
                // this step's weights: team 1 has waited for its pieces; the barrier also frees the other slot
                if (tr_on) ts[1] = __builtin_readcyclecounter();
                if (!(dbg & 8)) d4_barrier();
            }
            if (tr_on) ts[2] = __builtin_readcyclecounter();
            if (kk == 1) {
                if (!(dbg & 4)) blend();             // step t's operand (requested a step ago)
                if (tr_on) ts[3] = __builtin_readcyclecounter();
                if (t < 8 && !(dbg & 4)) request(t + 1);
                if (tr_on) ts[5] = __builtin_readcyclecounter();
                // L2 prefetch of the weights three steps ahead: inside a network step 4 GB pass between two uses of
                // a layer's weights, the slab's first touch is an HBM miss and the DMA runs only ONE step ahead of
                // its use (cold: +13 % on the four-block form).  One dword per 128-byte line, by the team without
                // DMA duty; the value is "used" a step later so that the compiler keeps the load.
                if (a.prefetch) {
                    asm volatile("" :: "v"(pf));
                    int tp = t + 3, cp = chunk;
                    if (tp >= 9) { tp -= 9; cp = chunk + 1; }
                    if (cp < c_hi && pb * 64 < NB * 32)
                        pf = *reinterpret_cast<const __attribute__((address_space(1))) uint32_t *>(
                            wfrag + ((size_t)(tp * a.nchunk + cp) * ncb + (n0 >> 5)) * 4096 + (unsigned)(pb * 64 + lane) * 128u);
                }
                if (WDB && chunk + 1 < c_hi && !(dbg & 64)) dma_piece(chunk + 1, t, wbase ^ (unsigned)W_WIN1);
            } else {
                if (WDB && chunk + 1 < c_hi && !(dbg & 64)) dma_piece(chunk + 1, t, wbase ^ (unsigned)W_WIN1);
                // next step's weights into the other slot (its last readers are behind this step's barrier)
                int tn = t + 1, cnx = chunk;
                if (tn == 9) { tn = 0; cnx = chunk + 1; }
                if (cnx < c_hi && !(dbg & 1)) dma_w1(tn, cnx, (step + 1) & 1);
                if (tr_on) ts[4] = __builtin_readcyclecounter();
            }
            const d4_f16x8 bhi = shi, blo = slo;   // this step's B operand
            // NB blocks of 32 output channels against this sample set; fragments of a pair of blocks are read
            // into one of two register sets whose previous readers are a full block of MFMAs back
            const d4_lds_char *ring = lds + W_RING + (step & 1) * SLOT + laneoff;
            auto frag = [&](int pair, int set) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const d4_lds_char *g = ring + (2 * pair + e) * 4096;
                    fh[set][e] = *reinterpret_cast<const d4_lds_f16x8 *>(g + kk * 1024);
                    fl[set][e] = *reinterpret_cast<const d4_lds_f16x8 *>(g + (2 + kk) * 1024);
                }
            };
            frag(0, 0);
#pragma unroll
            for (int pr = 0; pr < NB / 2; ++pr) {
                const int set = pr & 1;
                if (pr + 1 < NB / 2) frag(pr + 1, set ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                if (dbg & 2) {
                    acc[2 * pr][0] += (float)fl[set][0][0] + (float)fl[set][1][0] + (float)fh[set][0][0] + (float)fh[set][1][0] + (float)bhi[0] + (float)blo[0];
                    continue;
                }
                acc[2 * pr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[set][0], bhi, acc[2 * pr], 0, 0, 0);
                acc[2 * pr + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[set][1], bhi, acc[2 * pr + 1], 0, 0, 0);
                acc[2 * pr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[set][0], blo, acc[2 * pr], 0, 0, 0);
                acc[2 * pr + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[set][1], blo, acc[2 * pr + 1], 0, 0, 0);
                acc[2 * pr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[set][0], bhi, acc[2 * pr], 0, 0, 0);
                acc[2 * pr + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[set][1], bhi, acc[2 * pr + 1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (tr_on) ts[6] = __builtin_readcyclecounter();
            if (kk == 0) {
                if (t < 8 && !(dbg & 4)) blend();                    // step t + 1's operand
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // own weight pieces (issued in front of the MFMAs) landed
                if (t < 7 && !(dbg & 4)) request(t + 2);
            }
            if (tr_on) {
                ts[7] = __builtin_readcyclecounter();
                if (lane == 0)
#pragma unroll
                    for (int e = 0; e < 8; ++e) cn_d4_trace[(kk * 64 + step) * 8 + e] = ts[e];
            }
        }
    }

    // ---- epilogue, one pair of 32-channel blocks at a time: every wave stages its 32 pixels x 64 channels
    // (window and records are dead), the two K-half teams' sums are added on the way out,
    // y = relu?((acc + bias) * scale + shift), whole lines stored.  acc[j][r]: channel 32 j + (r & 3) +
    // 8 (r >> 2) + 4 h of pixel l31.
    const int cq = lane & 15, rr = lane >> 4;     // 16 lanes per pixel row, four rows per pass
#pragma unroll
    for (int pr = 0; pr < ((dbg & 128) ? 0 : NB / 2); ++pr) {
        __syncthreads();
        {
            float *Cs = reinterpret_cast<float *>(smem + wave * W_STG);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const cn_f32x16 &A = acc[2 * pr + j];
                    const cn_f32x4 v = {A[4 * g], A[4 * g + 1], A[4 * g + 2], A[4 * g + 3]};
                    *reinterpret_cast<cn_f32x4 *>(Cs + l31 * W_LDC + 32 * j + 8 * g + 4 * h) = v;
                }
        }
        __syncthreads();
        const int n = n0 + 64 * pr + cq * 4;
        const float *epl = reinterpret_cast<const float *>(smem + W_EPI) + 64 * pr + cq * 4;
        const cn_f32x4 bs = *reinterpret_cast<const cn_f32x4 *>(epl);
        const cn_f32x4 sc = *reinterpret_cast<const cn_f32x4 *>(epl + 256);
        const cn_f32x4 sf2 = *reinterpret_cast<const cn_f32x4 *>(epl + 512);
        const float *C0 = reinterpret_cast<const float *>(smem + pb * W_STG);
        const float *C1 = reinterpret_cast<const float *>(smem + (pb + 4) * W_STG);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = kk * 16 + it * 4 + rr;
            const int mm = pb * 32 + row;
            const size_t off = (size_t)((b * H + ty0 + (mm >> 4)) * W + tx0 + (mm & 15));
            cn_f32x4 v = *reinterpret_cast<const cn_f32x4 *>(C0 + row * W_LDC + cq * 4);
            v = v + *reinterpret_cast<const cn_f32x4 *>(C1 + row * W_LDC + cq * 4);
            if (a.partial) {   // K split: raw sums, one slab per split; the reduce kernel does the rest
                if (n < a.cout_pad)
                    *reinterpret_cast<cn_f32x4 *>(a.partial + ((size_t)blockIdx.z * ((size_t)a.B * H * W) + off) * a.cout_pad + n) = v;
            } else if (n + 4 <= a.Cout) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float tt = (v[e] + bs[e]) * sc[e] + sf2[e];
                    v[e] = a.relu ? fmaxf(tt, 0.f) : tt;
                }
                if (a.out_plain)
                    *reinterpret_cast<cn_f32x4 *>(reinterpret_cast<float *>(a.y) + off * a.out_pitch + n) = v;
                else {
                    cn_rng_upd4(rng_out, v);
                    cn_store4_f32s(a.y, off, a.out_pitch, n, v);
                }
            }
        }
    }
    if (a.range) {
        if (!a.out_plain && !a.partial) cn_rng_commit(a.range, 0, rng_out);
        cn_rng_commit(a.range, 1, MSIG ? rng_in * a.x_mul : rng_in);
    }
}

template <int NB>
int launch_dcn_wide(const D4Args &a, int mask_sigmoid, hipStream_t st)
{
    constexpr int LDS = W_RING + 2 * NB * 4096 + (NB == 4 ? W_WBYTES : 0);
    static_assert(LDS <= 163840, "one workgroup per CU");
    dim3 grid((unsigned)(a.B * a.tiles_x * a.tiles_y), (unsigned)(a.Cout / (32 * NB)), (unsigned)a.ksplit);
    if (a.dbg && mask_sigmoid) {
        CN_SET_MAX_LDS_ONCE((dcn_wide_kernel<NB, true, true>), LDS);
        hipLaunchKernelGGL((dcn_wide_kernel<NB, true, true>), grid, dim3(W_NT), LDS, st, a);
    } else if (mask_sigmoid) {
        CN_SET_MAX_LDS_ONCE((dcn_wide_kernel<NB, true, false>), LDS);
        hipLaunchKernelGGL((dcn_wide_kernel<NB, true, false>), grid, dim3(W_NT), LDS, st, a);
    } else {
        CN_SET_MAX_LDS_ONCE((dcn_wide_kernel<NB, false, false>), LDS);
        hipLaunchKernelGGL((dcn_wide_kernel<NB, false, false>), grid, dim3(W_NT), LDS, st, a);
    }
    CN_CHECK_LAUNCH();
    return CN_OK;
}

}  // namespace

// probe build: the cycle stamps of the last launch with key 9 bit 512 (2 waves x 64 steps x 8 stamps)
extern "C" int cn_dcn_wide_trace(unsigned long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(cn_d4_trace), sizeof(unsigned long long) * 2 * 64 * 8) == hipSuccess ? CN_OK : CN_ERR_LAUNCH;
}

// Shapes this kernel takes (the caller falls back to the team form otherwise): maps of whole 8 x 16 pixel
// tiles, whole 32-channel chunks, Cout a multiple of 128.  nb: 0 = by shape (8 blocks where Cout % 256 == 0
// and the grid still fills the chip, else 4), 4 / 8 = forced.
int cn_dcn_wide_f32s(const float *x, const void *w_packed, const float *bias, const float *om,
                     int om_pitch, const float *scale, const float *shift, void *y, int out_pitch,
                     int out_plain, int B, int Cin, int H, int W, int Cout, int mask_sigmoid, int relu,
                     float x_mul, uint32_t *range, int nb, int dbg, float *partial, size_t partial_bytes,
                     int *ksplit_out, hipStream_t st)
{
    if (ksplit_out) *ksplit_out = 1;
    if ((H & 7) || (W & 15) || (Cin & 31) || (Cout & 127)) return CN_ERR_UNSUPPORTED;
    if (H > 16383 || W > 16383 || (out_pitch & 3) || !cn_aligned16(y) || !cn_aligned16(x)) return CN_ERR_UNSUPPORTED;
    if ((size_t)B * H * W * Cin * 4 >= ((size_t)1 << 32)) return CN_ERR_UNSUPPORTED;   // 32-bit byte offsets
    // global path: 15-bit corner coordinates, 24-bit integer multiplies (pixel index, bytes per pixel)
    if ((size_t)B * H * W >= ((size_t)1 << 24) || (size_t)Cin * 4 >= ((size_t)1 << 24)) return CN_ERR_UNSUPPORTED;
    const long tiles = (long)B * (H / W_TY) * (W / W_TX);
    if (nb != 4 && nb != 8) nb = (Cout % 256 == 0) ? 8 : 4;
    if (nb == 8 && (Cout & 255)) nb = 4;
    const long wgs = tiles * (Cout / (32 * nb));
    // Too few tiles for the chip but a deep K (512 -> 256 @ 16^2): split the 32-channel chunks over
    // 2 / 4 / 8 workgroups per tile -- raw fp32 partial sums in the caller's workspace, summed in a
    // fixed order by splitk_reduce_kernel (deterministic)
    int ksplit = 1;
    {
        const int nchunk = Cin / 32;
        const int cout_pad = (Cout + 31) / 32 * 32;
        for (int s2 = 2; s2 <= 8 && partial && wgs * ksplit < cn_tune_dcn_wide_wgs; s2 *= 2)
            if (nchunk % s2 == 0 && nchunk / s2 >= 2 &&
                (size_t)s2 * B * H * W * cout_pad * sizeof(float) <= partial_bytes)
                ksplit = s2;
    }
    D4Args a = {};
    a.x = x; a.w = w_packed; a.bias = bias; a.scale = scale; a.shift = shift; a.om = om; a.y = y;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.om_pitch = om_pitch;
    a.relu = relu; a.out_pitch = out_pitch; a.out_plain = out_plain;
    a.cin_pad = Cin;
    a.cout_pad = (Cout + 31) / 32 * 32;
    a.nchunk = Cin / 32;
    a.tiles_x = W / W_TX;
    a.tiles_y = H / W_TY;
    a.x_mul = x_mul; a.range = range; a.dbg = dbg;
    // L2 prefetch: pays where the weight stream is long and cold (four-block form on deep K: +5 % cold); costs 1-4 % elsewhere
    a.prefetch = (cn_tune_dcn_wide_prefetch && nb == 4 && Cin >= 256) ? 1 : 0;
    a.ksplit = ksplit;
    a.partial = ksplit > 1 ? partial : nullptr;
    if (ksplit_out) *ksplit_out = ksplit;
    return nb == 8 ? launch_dcn_wide<8>(a, mask_sigmoid, st) : launch_dcn_wide<4>(a, mask_sigmoid, st);
}
