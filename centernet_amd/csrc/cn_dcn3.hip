// cn_dcn3.hip -- fused modulated deformable convolution (DCNv2) forward, f32s arithmetic: the
// TEAM form of the register-sampling window kernel (round 5).
//
// Replaces: DCN.forward -> DCNv2Function.forward -> dcn_v2_cuda_forward
//   (DCNv2/dcn_v2.py:64-70, dcn_v2_func.py:22-38, src/dcn_v2_cuda.c:10-102): per sample a bias
//   SGEMM, modulated_deformable_im2col_gpu_kernel (src/cuda/dcn_v2_im2col_cuda.cu:118-180, bilinear
//   sampler :18-47) writing a Cin*9*HW column buffer, and the main SGEMM.
//
// Why another form.  dcn_reg_kernel (cn_dcn2.hip) is latency-bound: every wave is a dependent chain
// record -> decode -> 16 window reads -> blend -> split -> MFMA, it needs 215-256 registers (two waves
// per SIMD), and its counters show LDS, VALU and the matrix pipe each 20-35 % busy -- they take turns.
// Per (pixel, tap, 32 channels) the units need 2 (LDS, 512 B at 256 B/clk) / 1.6 (VALU) / 3 (matrix, 64
// output channels) cycles of a CU; the kernel took 14.  What buys overlap on this part is waves, so
// this form is built for FOUR waves per SIMD:
//   * a workgroup is eight waves on ONE 8 x 16 pixel tile and one window: four pixel blocks x two
//     TEAMS.  The teams split the (tap, chunk) steps of the tile between them (step parity) and add
//     their accumulators in the epilogue (T mode, 64 output channels per workgroup), or each takes
//     64 of the workgroup's 128 output channels (N mode);
//   * <= 128 registers: a lane-step is walked as two K halves of 16 channels (8 per lane: eight
//     ds_read_b128, 32 blend FMAs, one split, six MFMAs), the sample's four corner weights come
//     READY from the record (validity rules, mask, sigmoid and the f32s input exponent folded in
//     at the prologue), so the per-step decode is a handful of address XORs;
//   * the window is filled by LDS-DMA (global_load_lds_dwordx4: no staging registers, no VALU):
//     unpadded 128-byte pixels, XOR-swizzled through the per-lane SOURCE address so that the
//     16-lane groups of a ds_read_b128 hit 16 distinct bank groups for undisplaced samples.  A DMA
//     copies bytes -- it cannot scale, clamp or track -- so the input exponent rides in the corner
//     weights and the range word is fed by reading the thread's own six pieces back from LDS
//     (+8 % LDS reads); values beyond the fp16 range are then SEEN (the forward is re-run by the
//     host) rather than clamped;
//   * samples beyond the window's reach (|offset| > 3 px) take their corners from global memory
//     inside a rare branch, as before, so unbounded offsets (dcn_v2.py:65-67) stay exact.
// Semantics held: sampling domain h_im > -1 && w_im > -1 && h_im < H && w_im < W
// (dcn_v2_im2col_cuda.cu:165), corner rule (:30-41), weights hh*hw, hh*lw, lh*hw, lh*lw (:26-28,43),
// value * mask (:174; here the mask multiplies the four corner weights -- a reassociation of
// 1-2 ulp in fp32, far below the 2^-22 of the f32s split), bias then accumulate (dcn_v2_cuda.c:61-97).
#include "cn_common.h"

int cn_tune_dcn_team = 3;       // cn_set_tuning key 36: 0 = off, 1 = layers with <= 64 output channels, 2 = every layer it takes (T mode),
                                // 3 = every layer, N mode where Cout is a multiple of 128 and that still fills the chip
int cn_tune_dcn_team_wgs = 512; // cn_set_tuning key 37: K split until a launch has this many workgroups
int cn_tune_dcn_team_stagger = 32; // cn_set_tuning key 38: start delay of the second resident workgroup of every CU, in units of 256 cycles (sweep 0 .. 128 at B = 32: 32-64 is 5-9 % faster on the multi-round shapes, nothing on the others; profiles/r05_dcn_team_stagger.txt)

// one 128-byte line of zeros: the DMA source of window pixels outside the image
__device__ __attribute__((aligned(128))) unsigned char cn_d3_zero_line[128];
// probe build, key 9 bit 256: cycle stamps of wave 0 of every 64th workgroup (tile life: start, records, first window, taps, epilogue, end)
__device__ unsigned long long cn_d3_trace[64 * 8];

namespace {

constexpr int T_NT = 512;                      // 8 waves: 4 pixel blocks x 2 teams
constexpr int T_TX = 16, T_TY = 8, T_PM = T_TX * T_TY;
constexpr int T_RCH = 3;                       // offsets up to +-3 px sample inside the window
constexpr int T_WX = T_TX + 2 + 2 * T_RCH;     // 24
constexpr int T_WY = T_TY + 2 + 2 * T_RCH;     // 16
constexpr int T_WPIX = T_WX * T_WY;            // 384
constexpr int T_PIXB = 128;                    // bytes per window pixel: 32 plain floats, unpadded
constexpr int T_ROWB = T_WX * T_PIXB;          // 3072 = 12 x 256: a row starts on bank group 0
constexpr int T_WBYTES = T_WPIX * T_PIXB;      // 49152
constexpr int T_NP = T_WPIX * 8 / T_NT;        // 6 DMA pieces (16 B) per thread and chunk
constexpr int T_RECW = T_WBYTES;               // float4 [9][128]: corner weights (mask, exponent, validity folded in)
constexpr int T_RECP = T_RECW + 9 * T_PM * 16; // uint2 [9][128]: swizzled LDS offsets of corners 1 and 2 | beyond-the-window flag + corner
constexpr int T_LDS_MAIN = T_RECP + 9 * T_PM * 8;          // 76800
constexpr int T_LDC = 68;                      // floats per staged pixel row (64 + 4)
constexpr int T_STG = 32 * T_LDC * 4;          // 8704 bytes per wave
constexpr int T_EPI = T_LDS_MAIN;              // float [3][128]: bias, scale, shift of the workgroup's output channels
constexpr int T_LDS = T_EPI + 3 * 128 * 4;     // 78336 (the epilogue strips, 8 x 8704 = 69632, alias the window and the records)
static_assert(2 * T_LDS <= 163840, "two workgroups per CU");
static_assert(T_ROWB % 256 == 0, "window rows keep the bank-group phase");

struct D3Args {
    const float *x;            // (B, H, W, Cin) plain fp32
    const void *w;             // f32s-packed [tap][cout_pad][cin_pad] row form + the fragment-ordered copy behind it
    const float *bias, *scale, *shift, *om;
    void *y;
    int B, H, W, Cin, Cout, om_pitch, relu;
    int cin_pad, cout_pad, nchunk, tiles_x, tiles_y, out_pitch, out_plain;
    float x_mul;
    uint32_t *range;
    int stagger;               // start delay of workgroups 256 .. 511 (the second occupant of every CU), units of 256 cycles
    int dbg;                   // probe build (cn_set_tuning key 9): 1 = every sample takes the global path, 8 = no MFMAs, 128 = no taps,
                               // 16 = no offset / mask loads, 32 = no output stores, 64 = no window DMA, 256 = cycle stamps of a tile's life (cn_dcn_team_trace)
    int ksplit;                // K-chunk ranges per tile (blockIdx.z); > 1: raw partial sums
    float *partial;            // [ksplit][B*H*W][cout_pad] fp32 (splitk_reduce_kernel applies the epilogue)
};

typedef _Float16 d3_f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) char d3_lds_char;
typedef __attribute__((address_space(1))) char d3_glb_char;
typedef __attribute__((address_space(3))) cn_f32x4 d3_lds_f32x4;
typedef __attribute__((address_space(1))) cn_f32x4 d3_glb_f32x4;
typedef __attribute__((address_space(3))) void d3_lds_void;
typedef __attribute__((address_space(1))) const void d3_glb_void;

__device__ __forceinline__ float d3_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ void d3_barrier()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// swizzled LDS byte offsets of window pixels (wy, wx) and (wy, wx + 1), quad 0 of lane half 0:
// physical 16-byte slot of logical quad q of a pixel = q ^ ((wx >> 1) & 7)
__device__ __forceinline__ unsigned d3_enc(int wy, int wx)
{
    const unsigned p = (unsigned)(wy * T_WX + wx);
    const unsigned q1 = p * T_PIXB + ((((unsigned)wx >> 1) & 7u) << 4);
    const unsigned q2 = (p + 1u) * T_PIXB + (((((unsigned)wx + 1u) >> 1) & 7u) << 4);
    return q1 | (q2 << 16);
}

// NMODE: teams take the two 64-channel halves of a 128-channel block (all nine taps each);
//        false: teams take alternate (tap, chunk) steps of ONE 64-channel block and add up
// MSIG:  the mask is sigmoid(conv output) (dcn_v2.py:67), hence in [0, 1]: a sample is a convex blend
//        of window values times <= 1 and needs neither clamp nor range tracking of its own;
//        false = caller-supplied mask of any size (clamp + track per sample)
// DBG:   probe build (D3Args.dbg)
//
// Counters of the first build of this kernel (profiles/r05_dcn_team_counters.txt): 171 VALU instructions per
// (tap, chunk) wave-step at 4.4 cycles each against twelve MFMAs -- the vector ALU, not the matrix pipe, LDS or
// the weight stream, is what a step costs (v_fma_mixlo_f16 7.5 cycles, v_pk_fma_f32 4.6, v_cvt_pk_f16_f32 4,
// plain fp32 / integer 2.5-2.8: tools/ubench/valu_rates.hip).  The loop below is therefore written for
// instruction count: corner weights and window addresses come ready from the records, the rare global path
// keeps ALL its arithmetic inside its branch, the weight fragments are scalar-base + lane-offset loads.
template <bool NMODE, bool MSIG, bool DBG>
__global__ __launch_bounds__(T_NT, 4) void dcn_team_kernel(const D3Args a)
{
    extern __shared__ __attribute__((aligned(128))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int pb = wave & 3, team = wave >> 2;
    const int H = a.H, W = a.W;
    int bx = blockIdx.x;
    if (a.stagger) {
        // two workgroups share a CU and do the same work: started together they reach their window swaps,
        // prologues and epilogues together and nothing covers them.  One-off phase shift of the second occupant.
        const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (lin >= 256u && lin < 512u) {
            const unsigned long long t0 = __builtin_readcyclecounter();
            while (__builtin_readcyclecounter() - t0 < (unsigned long long)a.stagger * 256u) __builtin_amdgcn_s_sleep(32);
        }
    }
    {   // XCD-aware tile order: contiguous tile ranges per XCD (block b runs on XCD b % 8)
        const int q8 = gridDim.x >> 3;
        if (bx < (q8 << 3)) bx = (bx & 7) * q8 + (bx >> 3);
    }
    const int tiles = a.tiles_x * a.tiles_y;
    const int b = bx / tiles;
    const int tr = bx - b * tiles;
    const int ty0 = (tr / a.tiles_x) * T_TY, tx0 = (tr % a.tiles_x) * T_TX;
    const int wy0 = ty0 - 1 - T_RCH, wx0 = tx0 - 1 - T_RCH;
    const int dbg = DBG ? a.dbg : 0;
    unsigned long long ts[8] = {};
    const bool tr_on = DBG && (a.dbg & 256) && (blockIdx.x & 63) == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0 && (blockIdx.x >> 6) < 64;
    if (tr_on) ts[0] = __builtin_readcyclecounter();
    const int n0 = NMODE ? (int)blockIdx.y * 128 + 64 * team : (int)blockIdx.y * 64;
    const unsigned pix_bytes = (unsigned)a.Cin * 4u;
    const unsigned img_base = (unsigned)(b * H) * (unsigned)W;
    const d3_glb_char *xg = (const d3_glb_char *)a.x;
    const d3_lds_char *lds = (const d3_lds_char *)smem;
    float rng_in = 0.f, rng_out = 0.f;
    // K split (small maps with deep K): this workgroup's share of the 32-channel chunks
    const int cpw = a.nchunk / a.ksplit;
    const int c_lo = (int)blockIdx.z * cpw, c_hi = c_lo + cpw;

    // ---- window of one chunk by LDS-DMA: 3072 16-byte pieces, six per thread; piece i = pixel i >> 3,
    // physical slot i & 7, which receives the pixel's logical quad (i & 7) ^ ((wx >> 1) & 7)
    unsigned doff[T_NP];            // byte offset of the piece's source in x (chunk 0), ~0 = zero line
#pragma unroll
    for (int p = 0; p < T_NP; ++p) {
        const int i = p * T_NT + tid;
        const int wp = i >> 3, pq = i & 7;
        const int wy = wp / T_WX, wx = wp - wy * T_WX;
        const int lq = pq ^ ((wx >> 1) & 7);
        const int iy = wy0 + wy, ix = wx0 + wx;
        const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
        doff[p] = ok ? (img_base + (unsigned)(iy * W + ix)) * pix_bytes + 16u * (unsigned)lq : 0xffffffffu;
    }
    const d3_glb_char *zline = (const d3_glb_char *)cn_d3_zero_line + 16 * (lane & 7);
    auto dma = [&](int chunk) {
        const unsigned cb = (unsigned)chunk * 128u;
        if (DBG && (a.dbg & 64)) return;
#pragma unroll
        for (int p = 0; p < T_NP; ++p) {
            const d3_glb_char *src = (doff[p] != 0xffffffffu) ? xg + (doff[p] + cb) : zline;
            __builtin_amdgcn_global_load_lds((d3_glb_void *)src, (d3_lds_void *)(smem + (p * T_NT + wave * 64) * 16), 16, 0, 0);
        }
    };
    // the range word of the input side: max |x| over the thread's own pieces, read back from LDS
    // (own DMA writes are complete once vmcnt has drained)
    auto track = [&]() {
#pragma unroll
        for (int p = 0; p < T_NP; ++p) {
            const cn_f32x4 v = *reinterpret_cast<const d3_lds_f32x4 *>(lds + (p * T_NT + tid) * 16);
            cn_rng_upd4(rng_in, v);
        }
    };

    // ---- prologue: offsets / masks of the tile (1152 records over 512 threads, every load requested
    // before the first record is formed), the window of the first chunk behind them, then the records
    // (dcn_v2_im2col_cuda.cu:151-176).  A record = the four corner weights (validity rule :30-41, mask
    // (:174), sigmoid (dcn_v2.py:67) and the f32s input exponent folded in) + two words: the swizzled LDS
    // offsets of corners 1 and 2 (always valid window addresses) and, for a sample beyond the window's
    // reach, bit 31 + the top-left corner (yl + 1, xl + 1) in 15 bits each.
    {
        constexpr int NR = (9 * T_PM + T_NT - 1) / T_NT;   // 3 (the last trip a quarter full)
        float off_h[NR], off_w[NR], mkv[NR];
#pragma unroll
        for (int p = 0; p < NR; ++p) {
            const int i = min(p * T_NT + tid, 9 * T_PM - 1);
            const int tap = i >> 7, m = i & (T_PM - 1);
            const int oy = ty0 + (m >> 4), ox = tx0 + (m & 15);
            const float *om = a.om + (size_t)((b * H + oy) * W + ox) * a.om_pitch;
            if (dbg & 16) { off_h[p] = 0.3f; off_w[p] = -0.2f; mkv[p] = 0.1f; continue; }
            off_h[p] = om[2 * tap];
            off_w[p] = om[2 * tap + 1];
            mkv[p] = om[18 + tap];
        }
        // bias / scale / shift of this workgroup's output channels: requested here, parked in LDS for the
        // epilogue (loaded there, twelve dependent L2 round trips sat between the last MFMA and the first store)
        float epv = 0.f;
        if (tid < 384) {
            const int which = tid >> 7, cn = (NMODE ? (int)blockIdx.y * 128 : (int)blockIdx.y * 64) + (tid & 127);
            const float *src = which == 0 ? a.bias : (which == 1 ? a.scale : a.shift);
            epv = which == 1 ? 1.f : 0.f;
            if (src && cn < a.Cout) epv = src[cn];
        }
        dma(c_lo);
        if (tid < 384) reinterpret_cast<float *>(smem + T_EPI)[tid] = epv;
#pragma unroll
        for (int p = 0; p < NR; ++p) {
            const int i = p * T_NT + tid;
            const int tap = i >> 7, m = i & (T_PM - 1);
            const int oy = ty0 + (m >> 4), ox = tx0 + (m & 15);
            float mk = mkv[p];
            if (MSIG) mk = sigmoidf_ref(mk);            // dcn_v2.py:67 (v_exp + v_rcp: within 2e-7 of the IEEE form)
            mk *= a.x_mul;                              // plain input -> stored units (a power of two)
            const int ki = tap / 3, kj = tap - ki * 3;
            const float h_im = (float)(oy - 1 + ki) + off_h[p];
            const float w_im = (float)(ox - 1 + kj) + off_w[p];
            cn_f32x4 wv = {0.f, 0.f, 0.f, 0.f};
            // outside the sampling domain (:165): the sample is 0; the pixel's own position as
            // "corner" keeps the reads inside the window
            unsigned p0 = d3_enc((m >> 4) + 1 + T_RCH, (m & 15) + 1 + T_RCH), p1 = 0u;
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
                const bool inwin = (unsigned)wyl <= (unsigned)(T_WY - 2) && (unsigned)wxl <= (unsigned)(T_WX - 2);
                if (inwin && !(dbg & 1)) p0 = d3_enc(wyl, wxl);
                else p1 = 0x80000000u | ((unsigned)(yl + 1) << 15) | (unsigned)(xl + 1);
            }
            if (i < 9 * T_PM) {
                *reinterpret_cast<cn_f32x4 *>(smem + T_RECW + i * 16) = wv;
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                *reinterpret_cast<u32x2 *>(smem + T_RECP + i * 8) = u32x2{p0, p1};
            }
        }
    }
    if (tr_on) ts[1] = __builtin_readcyclecounter();          // records written (offset / mask loads waited for on the way)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tr_on) ts[2] = __builtin_readcyclecounter();          // first window landed
    if (MSIG && a.range) track();
    __syncthreads();                               // records and the window of the first chunk visible
    if (tr_on) ts[3] = __builtin_readcyclecounter();

    const int m = pb * 32 + l31;                   // this lane's pixel of the tile
    const unsigned hx = (unsigned)h << 5;
    const unsigned laneoff = (unsigned)lane * 16u;
    const char *wfrag = reinterpret_cast<const char *>(a.w) + (size_t)9 * a.cout_pad * a.cin_pad * 4;
    const int ncb = a.cout_pad >> 5;
    const int nb0 = min(n0 >> 5, ncb - 1), nb1 = min((n0 >> 5) + 1, ncb - 1);
    cn_f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    typedef float d3_f32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned d3_u32x2 __attribute__((ext_vector_type(2)));

    constexpr int TSTEP = NMODE ? 1 : 2;
    // MFMA operand hazard: an LDS read issued right behind an MFMA block must not land in
    // the A / B source registers of its (dependent, possibly still queued) MFMAs -- a rare wrong tile otherwise
    // (one launch in a few hundred at B = 32).  The record and window reads of the NEXT K half are issued
    // right behind this K half's block, so its operands are kept alive (an empty asm that "uses" them) until
    // those reads have been issued: the register allocator cannot hand them out as the reads' destinations.
    // The weight fragments (global loads: hundreds of cycles away) are requested behind that point and may
    // take the same registers.  No instruction is added.  (Requested one K half ahead through a second
    // register set the fragments arrived no earlier in wall time -- 0.104 vs 0.105 ms -- and the set left no
    // room for this guard.)
    d3_f16x8 kp0 = {}, kp1 = {}, kp2 = {}, kp3 = {}, kp4 = {}, kp5 = {};

    for (int chunk = c_lo; chunk < c_hi; ++chunk) {
        if (chunk != c_lo) {
            __syncthreads();                       // every wave is done with the previous window
            dma(chunk);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (MSIG && a.range) track();
            __syncthreads();
        }
        // byte offset, in x, of channel 8 h of this chunk in pixel 0 of the image (global path)
        const unsigned far_base = img_base * pix_bytes + (unsigned)chunk * 128u + hx;
        const int t0 = NMODE ? 0 : ((team ^ (chunk - c_lo)) & 1);
#pragma unroll 1
        for (int t = t0; t < ((dbg & 128) ? 0 : 9); t += TSTEP) {
            const cn_f32x4 wv = *reinterpret_cast<const d3_lds_f32x4 *>(lds + T_RECW + (t * T_PM + m) * 16);
            const d3_u32x2 pp = *reinterpret_cast<const __attribute__((address_space(3))) d3_u32x2 *>(lds + T_RECP + (t * T_PM + m) * 8);
            // fragment copy of this (tap, chunk): uniform base, lane offset; quarter kk at + 1 KiB kk
            const char *sw = wfrag + (size_t)(t * a.nchunk + chunk) * ncb * 4096;
            const char *g0 = sw + (size_t)nb0 * 4096, *g1 = sw + (size_t)nb1 * 4096;
            const d3_f32x2 w1 = {wv[0], wv[0]}, w2 = {wv[1], wv[1]}, w3 = {wv[2], wv[2]}, w4 = {wv[3], wv[3]};
            const unsigned A1 = (pp[0] & 0xffffu) ^ hx, A2 = (pp[0] >> 16) ^ hx;
            const bool far = (int)pp[1] < 0;
            unsigned o1 = 0, o2 = 0, o3 = 0, o4 = 0;
            if (far) {
                // beyond the window's reach: the four corners from global memory (clamped addresses --
                // off-map corners carry zero weight); 24-bit integer multiplies (pixel index and pixel
                // pitch are below 2^24, checked by the launcher)
                const int yl = (int)((pp[1] >> 15) & 0x7fffu) - 1, xl = (int)(pp[1] & 0x7fffu) - 1;
                const int y0 = max(yl, 0), y1 = min(yl + 1, H - 1);
                const int x0 = max(xl, 0), x1 = min(xl + 1, W - 1);
                const unsigned r0 = __umul24((unsigned)y0, (unsigned)W), r1 = __umul24((unsigned)y1, (unsigned)W);
                o1 = __umul24(r0 + (unsigned)x0, pix_bytes) + far_base;
                o2 = __umul24(r0 + (unsigned)x1, pix_bytes) + far_base;
                o3 = __umul24(r1 + (unsigned)x0, pix_bytes) + far_base;
                o4 = __umul24(r1 + (unsigned)x1, pix_bytes) + far_base;
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                // channels 16 kk + 8 h .. + 7 of the four corners: eight window reads
                const unsigned B1 = A1 ^ ((unsigned)kk << 6), B2 = A2 ^ ((unsigned)kk << 6);
                cn_f32x4 c1a = *reinterpret_cast<const d3_lds_f32x4 *>(lds + B1);
                cn_f32x4 c1b = *reinterpret_cast<const d3_lds_f32x4 *>(lds + (B1 ^ 16u));
                cn_f32x4 c2a = *reinterpret_cast<const d3_lds_f32x4 *>(lds + B2);
                cn_f32x4 c2b = *reinterpret_cast<const d3_lds_f32x4 *>(lds + (B2 ^ 16u));
                cn_f32x4 c3a = *reinterpret_cast<const d3_lds_f32x4 *>(lds + B1 + T_ROWB);
                cn_f32x4 c3b = *reinterpret_cast<const d3_lds_f32x4 *>(lds + (B1 ^ 16u) + T_ROWB);
                cn_f32x4 c4a = *reinterpret_cast<const d3_lds_f32x4 *>(lds + B2 + T_ROWB);
                cn_f32x4 c4b = *reinterpret_cast<const d3_lds_f32x4 *>(lds + (B2 ^ 16u) + T_ROWB);
                // the previous MFMA block's operands: alive until the reads above (and, through the loop,
                // the record reads of the step) have been issued
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("" :: "v"(kp0), "v"(kp1), "v"(kp2), "v"(kp3), "v"(kp4), "v"(kp5));
                __builtin_amdgcn_sched_barrier(0);
                // weights of this (tap, chunk, K half): the MFMA's A operand, straight from the fragment copy
                const d3_f16x8 wh0 = *reinterpret_cast<const d3_f16x8 *>(g0 + kk * 1024 + laneoff);
                const d3_f16x8 wl0 = *reinterpret_cast<const d3_f16x8 *>(g0 + (2 + kk) * 1024 + laneoff);
                const d3_f16x8 wh1 = *reinterpret_cast<const d3_f16x8 *>(g1 + kk * 1024 + laneoff);
                const d3_f16x8 wl1 = *reinterpret_cast<const d3_f16x8 *>(g1 + (2 + kk) * 1024 + laneoff);
                if (far) {
                    const d3_glb_char *g = xg + 64u * kk;
                    c1a = *reinterpret_cast<const d3_glb_f32x4 *>(g + o1);
                    c1b = *reinterpret_cast<const d3_glb_f32x4 *>(g + o1 + 16);
                    c2a = *reinterpret_cast<const d3_glb_f32x4 *>(g + o2);
                    c2b = *reinterpret_cast<const d3_glb_f32x4 *>(g + o2 + 16);
                    c3a = *reinterpret_cast<const d3_glb_f32x4 *>(g + o3);
                    c3b = *reinterpret_cast<const d3_glb_f32x4 *>(g + o3 + 16);
                    c4a = *reinterpret_cast<const d3_glb_f32x4 *>(g + o4);
                    c4b = *reinterpret_cast<const d3_glb_f32x4 *>(g + o4 + 16);
                    // drain here, inside the rare branch (vmcnt counts in order: waited for at the
                    // join, these loads would sit in front of every later weight fragment)
                    __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0)
                }
                // w1*v1 + w2*v2 + w3*v3 + w4*v4 (dcn_v2_im2col_cuda.cu:43-45; mask and exponent inside the
                // weights), two channels per instruction (v_pk_fma_f32)
                cn_f32x4 va, vb;
                {
    #ifdef CN_DCN_SCALAR_FMA
                #pragma unroll
                for (int e = 0; e < 4; ++e) {
                    va[e] = __builtin_fmaf(c4a[e], w4[0], __builtin_fmaf(c3a[e], w3[0], __builtin_fmaf(c2a[e], w2[0], c1a[e] * w1[0])));
                    vb[e] = __builtin_fmaf(c4b[e], w4[0], __builtin_fmaf(c3b[e], w3[0], __builtin_fmaf(c2b[e], w2[0], c1b[e] * w1[0])));
                }
                #else
                auto lo2 = [](cn_f32x4 v) { return d3_f32x2{v[0], v[1]}; };
                    auto hi2 = [](cn_f32x4 v) { return d3_f32x2{v[2], v[3]}; };
                    const d3_f32x2 a0 = lo2(c1a) * w1 + lo2(c2a) * w2 + lo2(c3a) * w3 + lo2(c4a) * w4;
                    const d3_f32x2 a1 = hi2(c1a) * w1 + hi2(c2a) * w2 + hi2(c3a) * w3 + hi2(c4a) * w4;
                    const d3_f32x2 b0 = lo2(c1b) * w1 + lo2(c2b) * w2 + lo2(c3b) * w3 + lo2(c4b) * w4;
                    const d3_f32x2 b1 = hi2(c1b) * w1 + hi2(c2b) * w2 + hi2(c3b) * w3 + hi2(c4b) * w4;
                    va = cn_f32x4{a0[0], a0[1], a1[0], a1[1]};
                    vb = cn_f32x4{b0[0], b0[1], b1[0], b1[1]};
                #endif
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
                const d3_f16x8 shi = __builtin_shufflevector(ha, hb, 0, 1, 2, 3, 4, 5, 6, 7);
                const d3_f16x8 slo = __builtin_shufflevector(la, lb, 0, 1, 2, 3, 4, 5, 6, 7);
                if (dbg & 8) {
                    acc[0][0] += (float)shi[0] + (float)slo[0] + (float)wh0[0] + (float)wl0[0] + (float)wh1[0] + (float)wl1[0];
                    continue;
                }
                // every operand is in registers before the first MFMA issues (operand hazard note, cn_conv.hip)
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl0, shi, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl1, shi, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh0, slo, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh1, slo, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh0, shi, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh1, shi, acc[1], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
                kp0 = wh0; kp1 = wl0; kp2 = wh1; kp3 = wl1; kp4 = shi; kp5 = slo;
            }
        }
    }

    if (tr_on) ts[4] = __builtin_readcyclecounter();          // taps and window swaps done
    // ---- epilogue: every wave stages its 32 pixels x 64 channels (the window and the records are
    // dead), the teams' sums are added on the way out (T mode), y = relu?((acc + bias) * scale + shift)
    // and whole lines are stored.  acc[j][r]: channel 32 j + (r & 3) + 8 (r >> 2) + 4 h of pixel l31.
    __syncthreads();
    {
        float *Cs = reinterpret_cast<float *>(smem + wave * T_STG);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const cn_f32x4 v = {acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
                *reinterpret_cast<cn_f32x4 *>(Cs + l31 * T_LDC + 32 * j + 8 * g + 4 * h) = v;
            }
    }
    __syncthreads();
    const int cq = lane & 15, rr = lane >> 4;     // 16 lanes per pixel row, four rows per pass
    const int n = n0 + cq * 4;
    const float *epl = reinterpret_cast<const float *>(smem + T_EPI) + (NMODE ? 64 * team : 0) + cq * 4;
    const cn_f32x4 bs = *reinterpret_cast<const cn_f32x4 *>(epl);
    const cn_f32x4 sc = *reinterpret_cast<const cn_f32x4 *>(epl + 128);
    const cn_f32x4 sf2 = *reinterpret_cast<const cn_f32x4 *>(epl + 256);
    constexpr int PASSES = NMODE ? 8 : 4;
    const float *C0 = reinterpret_cast<const float *>(smem + (NMODE ? wave : pb) * T_STG);
    const float *C1 = reinterpret_cast<const float *>(smem + (pb + 4) * T_STG);
#pragma unroll
    for (int it = 0; it < PASSES; ++it) {
        const int row = (NMODE ? 0 : team * 16) + it * 4 + rr;
        const int mm = pb * 32 + row;
        const size_t off = (size_t)((b * H + ty0 + (mm >> 4)) * W + tx0 + (mm & 15));
        cn_f32x4 v = *reinterpret_cast<const cn_f32x4 *>(C0 + row * T_LDC + cq * 4);
        if (!NMODE) v = v + *reinterpret_cast<const cn_f32x4 *>(C1 + row * T_LDC + cq * 4);
        if (a.partial) {   // K split: raw sums, one slab per split; the reduce kernel does the rest
            if (n < a.cout_pad)
                *reinterpret_cast<cn_f32x4 *>(a.partial + ((size_t)blockIdx.z * ((size_t)a.B * H * W) + off) * a.cout_pad + n) = v;
        } else if (n + 4 <= a.Cout && !(dbg & 32)) {
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
    if (tr_on) {
        ts[5] = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ts[6] = __builtin_readcyclecounter();      // stores drained
        ts[7] = __builtin_amdgcn_s_memrealtime();  // 100 MHz wall clock: when this workgroup ended
        for (int e = 0; e < 8; ++e) cn_d3_trace[(blockIdx.x >> 6) * 8 + e] = ts[e];
    }
    if (a.range) {
        if (!a.out_plain && !a.partial) cn_rng_commit(a.range, 0, rng_out);
        // window values were tracked in the tensor's own units: x' = x * x_mul (a power of two)
        cn_rng_commit(a.range, 1, MSIG ? rng_in * a.x_mul : rng_in);
    }
}

template <bool NMODE>
int launch_dcn_team(const D3Args &a, int mask_sigmoid, hipStream_t st)
{
    dim3 grid((unsigned)(a.B * a.tiles_x * a.tiles_y), cn_cdiv(a.Cout, NMODE ? 128 : 64), (unsigned)a.ksplit);
    if (a.dbg && mask_sigmoid) {
        CN_SET_MAX_LDS_ONCE((dcn_team_kernel<NMODE, true, true>), T_LDS);
        hipLaunchKernelGGL((dcn_team_kernel<NMODE, true, true>), grid, dim3(T_NT), T_LDS, st, a);
    } else if (mask_sigmoid) {
        CN_SET_MAX_LDS_ONCE((dcn_team_kernel<NMODE, true, false>), T_LDS);
        hipLaunchKernelGGL((dcn_team_kernel<NMODE, true, false>), grid, dim3(T_NT), T_LDS, st, a);
    } else {
        CN_SET_MAX_LDS_ONCE((dcn_team_kernel<NMODE, false, false>), T_LDS);
        hipLaunchKernelGGL((dcn_team_kernel<NMODE, false, false>), grid, dim3(T_NT), T_LDS, st, a);
    }
    CN_CHECK_LAUNCH();
    return CN_OK;
}

}  // namespace

extern "C" int cn_dcn_team_trace(unsigned long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(cn_d3_trace), sizeof(unsigned long long) * 64 * 8) == hipSuccess ? CN_OK : CN_ERR_LAUNCH;
}

// Shapes this kernel takes (the caller falls back to the other forms otherwise): maps of whole
// 8 x 16 pixel tiles, whole 32-channel chunks, Cout a multiple of 4 and >= 33.  nmode: 1 (2 = even on small grids) = the teams
// split a 128-channel block (needs Cout % 128 == 0), 0 = they split the steps of a 64-channel block.
int cn_dcn_team_f32s(const float *x, const void *w_packed, const float *bias, const float *om,
                     int om_pitch, const float *scale, const float *shift, void *y, int out_pitch,
                     int out_plain, int B, int Cin, int H, int W, int Cout, int mask_sigmoid, int relu,
                     float x_mul, uint32_t *range, int nmode, int dbg, float *partial,
                     size_t partial_bytes, int *ksplit_out, hipStream_t st)
{
    if (ksplit_out) *ksplit_out = 1;
    if ((H & 7) || (W & 15) || (Cin & 31) || (Cout & 3) || Cout <= 32) return CN_ERR_UNSUPPORTED;
    if (H > 32767 || W > 32767 || (out_pitch & 3) || !cn_aligned16(y) || !cn_aligned16(x)) return CN_ERR_UNSUPPORTED;
    if ((size_t)B * H * W * Cin * 4 >= ((size_t)1 << 32)) return CN_ERR_UNSUPPORTED;   // 32-bit byte offsets
    // global path: 15-bit corner coordinates, 24-bit integer multiplies (pixel index, bytes per pixel)
    if (H > 16383 || W > 16383 || (size_t)B * H * W >= ((size_t)1 << 24) || (size_t)Cin * 4 >= ((size_t)1 << 24)) return CN_ERR_UNSUPPORTED;
    if (nmode && (Cout & 127)) nmode = 0;
    // N mode halves the workgroup count: only where two workgroups per CU remain (measured: 256 -> 128 @ 32^2 at
    // B = 32 has 256 tiles: 0.093 ms in T mode, 0.109 in N mode; 128 -> 128 @ 64^2 and 256 -> 256 @ 32^2: 0.179 /
    // 0.172 against 0.195 / 0.209)
    if (nmode == 1 && (long)B * (H / T_TY) * (W / T_TX) * (Cout / 128) < cn_tune_dcn_team_wgs) nmode = 0;
    const long wgs = (long)B * (H / T_TY) * (W / T_TX) * cn_cdiv(Cout, nmode ? 128 : 64);
    // Too few tiles for the chip but a deep K (512 -> 256 @ 16^2): split the 32-channel chunks over
    // 2 / 4 / 8 workgroups per tile -- raw fp32 partial sums in the caller's workspace, summed in a
    // fixed order by splitk_reduce_kernel (deterministic)
    int ksplit = 1;
    {
        const int nchunk = Cin / 32;
        const int cout_pad = (Cout + 31) / 32 * 32;
        for (int s2 = 2; s2 <= 8 && partial && wgs * ksplit < cn_tune_dcn_team_wgs; s2 *= 2)
            if (nchunk % s2 == 0 && nchunk / s2 >= 2 &&
                (size_t)s2 * B * H * W * cout_pad * sizeof(float) <= partial_bytes)
                ksplit = s2;
    }
    D3Args a = {};
    a.x = x; a.w = w_packed; a.bias = bias; a.scale = scale; a.shift = shift; a.om = om; a.y = y;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.om_pitch = om_pitch;
    a.relu = relu; a.out_pitch = out_pitch; a.out_plain = out_plain;
    a.cin_pad = Cin;
    a.cout_pad = (Cout + 31) / 32 * 32;
    a.nchunk = Cin / 32;
    a.tiles_x = W / T_TX;
    a.tiles_y = H / T_TY;
    a.x_mul = x_mul; a.range = range; a.dbg = dbg;
    a.stagger = cn_tune_dcn_team_stagger;
    a.ksplit = ksplit;
    a.partial = ksplit > 1 ? partial : nullptr;
    if (ksplit_out) *ksplit_out = ksplit;
    return nmode ? launch_dcn_team<true>(a, mask_sigmoid, st) : launch_dcn_team<false>(a, mask_sigmoid, st);
}
