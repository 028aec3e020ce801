// cn_dcn2.hip -- fused modulated deformable convolution (DCNv2) forward, f32s arithmetic, with
// the input staged as an LDS WINDOW: dcn_reg_kernel -- every lane samples its own MFMA operand from
// the window; no A tile, no per-step barrier.  (Round 5: the team form in cn_dcn3.hip is built on
// this one for four waves per SIMD; this kernel stays for 128-wide output tiles.  The first window
// design -- sampling waves writing an A tile that multiplying waves consume, dcn_win_kernel -- was
// slower than the global-gather form on every shape, 70-97 against 100-155 TFLOP/s at B = 32, and
// was removed in round 5; DESIGN.md section 3.2 keeps the A/B.)
//
// Replaces: DCN.forward -> DCNv2Function.forward -> dcn_v2_cuda_forward
//   (DCNv2/dcn_v2.py:64-70, dcn_v2_func.py:22-38, src/dcn_v2_cuda.c:10-102): per sample a bias
//   SGEMM, modulated_deformable_im2col_gpu_kernel (src/cuda/dcn_v2_im2col_cuda.cu:118-180, bilinear
//   sampler :18-47) writing a Cin*9*HW column buffer, and the main SGEMM.
//
// Why a window at all (measured, round 3): the global-gather form (cn_conv.hip igemm_kernel<A_DCN>)
// moves 16 bytes through L1 / L2 per (pixel, tap, channel quad) -- 4 corners x fp32 -- i.e. 295 KB
// per 64-pixel tile and 32-channel chunk; on 64->64@128^2 that is 17 TB/s of L2 -> L1 traffic at
// 108 TFLOP/s with the matrix pipe 15 % busy: every (tap, chunk) step is a dependent chain
// record -> four gathers -> blend -> LDS -> barrier -> MFMA of ~2.4 us.
// Epilogue as everywhere: y = relu?((acc + bias) * scale + shift), plain fp32 or f32s, range words.
#include "cn_common.h"

// K is split until a launch has this many workgroups (round-4 sweep of 256 / 512 / 1024 and of 64-wide N tiles
// for Cout > 64, cn_set_tuning keys 34 / 35: within +-5 % on every layer shape, profiles/r04_dcn_split_sweep.txt;
// the keys were removed in round 5)
constexpr int cn_tune_dcn_wgs = 256;

namespace {

constexpr int TS = 8;          // tiles are TS pixel rows high

struct Dcn2Args {
    const float *x;            // (B, H, W, Cin) plain fp32
    const void *w;             // f32s-packed [tap][cout_pad][cin_pad] (row form)
    const float *bias, *scale, *shift, *om;
    void *y;
    int B, H, W, Cin, Cout, om_pitch, mask_sigmoid, relu;
    int cin_pad, cout_pad, nchunk, tiles_x, tiles_y, out_pitch, out_plain;
    float x_mul;
    uint32_t *range;
    int dbg;                   // debug switches (cn_set_tuning key 9)
    int ksplit;                // K-chunk ranges per tile (blockIdx.z), > 1: raw partial sums
    float *partial;            // [ksplit][B*H*W][cout_pad] fp32 (splitk_reduce_kernel applies the epilogue)
};

typedef _Float16 d2_f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) char d2_lds_char;
typedef __attribute__((address_space(1))) char d2_glb_char;
typedef __attribute__((address_space(3))) cn_f32x4 d2_lds_f32x4;
typedef __attribute__((address_space(1))) cn_f32x4 d2_glb_f32x4;

__device__ __forceinline__ float d2_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// ======================================================================================
// Register-sampling form (round 3, second design).  The wave-specialised first design lost to
// the global-gather form (70-97 against 100-155 TFLOP/s at B = 32): its A tile and weight tile
// crossed LDS once more than the window reads themselves, and each (tap, chunk) step of a 64-pixel
// tile was far shorter than the one-step-ahead global loads and the barrier that closed it.  Here
//   * a workgroup of four waves owns 8 x 16 output pixels; the (8 + 8) x (16 + 8) input window
//     of one 32-channel chunk sits in LDS (144-byte pixels, 3584-byte rows: bank-conflict free
//     for undisplaced samples, see R_WLINE);
//   * every lane SAMPLES ITS OWN MFMA OPERAND: lane (l31, h) of a wave is output pixel
//     32 * wave + l31 and channels 8h..8h+7, 16+8h..16+8h+7 of the chunk -- four 16-byte quads,
//     four bilinear corners each (16 ds_read_b128), blended in fp32, modulated, split into
//     (high, low) fp16 and used directly as the 32x32x16 MFMA's B operand.  No A tile in LDS;
//   * the weights are the MFMA's A operand, read from the FRAGMENT-ordered packed copy
//     (cn_conv.hip pack_weight_f32s_frag_kernel: four 1 KiB quarters per (tap, chunk, 32 rows)),
//     global -> registers, requested at the top of the step and landed under the sampling;
//   * no barrier inside a chunk: waves run free over the nine taps; two barriers per chunk
//     swap the window.  Two workgroups per CU (74 KB of LDS each) overlap each other's swaps.
// acc[j][r]: output channel 32j + (r & 3) + 8 (r >> 2) + 4h of pixel l31 -- four consecutive
// channels per register quad, staged through (wave-private) LDS for whole-line stores.
constexpr int R_NT = 256;
constexpr int R_TX = 16, R_TY = 8, R_PM = R_TX * R_TY;
constexpr int R_RCH = 3;                         // offsets up to +-3 px sample inside the window
constexpr int R_WX = R_TX + 2 + 2 * R_RCH;       // 24
constexpr int R_WY = R_TY + 2 + 2 * R_RCH;       // 16
constexpr int R_WPIX = R_WX * R_WY;              // 384
constexpr int R_WROW = 36;                       // floats per window pixel (128 B + 16 B pad)
// Floats per window ROW: 24 pixels of 144 bytes + 128 bytes of pad = 3584 = 14 x 256 bytes.  A
// ds_read_b128 is served in groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} of each
// half wave (MI355X_MICROARCH.md, LDS) -- i.e. 8 pixels of tile row 0 and 8 of tile row 1 here;
// 144-byte pixels put 16 consecutive x on 16 distinct 16-byte bank groups, and a row stride that
// is a multiple of 256 bytes keeps the second row's eight on the groups the first row's eight
// leave free (with 24 x 144 bytes per row every group of the undisplaced pattern was 2-way).
constexpr int R_WLINE = R_WX * R_WROW + 32;      // 896
constexpr size_t R_WBYTES = (size_t)R_WY * R_WLINE * 4;                 // 57344
constexpr size_t R_LDS = R_WBYTES + (size_t)9 * R_PM * 16;              // + 18432 = 75776

// DBG: probe build (cn_set_tuning key 9 != 0): bit 1 = every sample takes the global path,
// 8 = no MFMAs, 128 = no taps at all (prologue + window swaps + epilogue only)
// MSIG: the mask is sigmoid(conv output) (dcn_v2.py:67), hence in [0, 1]; false = caller-supplied mask
template <int BN, bool DBG, bool MSIG>
__global__ __launch_bounds__(R_NT, 2) void dcn_reg_kernel(const Dcn2Args a)
{
    constexpr int NB = BN / 32;
    constexpr int LDC = BN + 4;
    static_assert((size_t)4 * 32 * LDC * 4 <= R_LDS, "epilogue staging");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *Win = reinterpret_cast<float *>(smem);
    cn_i32x4 *Rec = reinterpret_cast<cn_i32x4 *>(smem + R_WBYTES);         // [9][R_PM]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int H = a.H, W = a.W;
    int bx = blockIdx.x;
    {   // XCD-aware tile order: contiguous tile ranges per XCD (block b runs on XCD b % 8)
        const int q8 = gridDim.x >> 3;
        if (bx < (q8 << 3)) bx = (bx & 7) * q8 + (bx >> 3);
    }
    const int tiles = a.tiles_x * a.tiles_y;
    const int b = bx / tiles;
    const int tr = bx - b * tiles;
    const int ty0 = (tr / a.tiles_x) * R_TY, tx0 = (tr % a.tiles_x) * R_TX;
    const int wy0 = ty0 - 1 - R_RCH, wx0 = tx0 - 1 - R_RCH;
    const int dbg = DBG ? a.dbg : 0;
    const int n0 = blockIdx.y * BN;
    const float a_x_mul = a.x_mul;
    constexpr bool msig = MSIG;
    const cn_f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const unsigned pix_bytes = (unsigned)a.Cin * 4u;
    const unsigned img_base = (unsigned)(b * H) * (unsigned)W;
    const d2_glb_char *xg = (const d2_glb_char *)a.x;
    const d2_lds_char *win_lds = (const d2_lds_char *)smem;
    float rng_in = 0.f, rng_out = 0.f;
    // K split (small maps with deep K): this workgroup's share of the 32-channel chunks
    const int cpw = a.nchunk / a.ksplit;
    const int c_lo = (int)blockIdx.z * cpw, c_hi = c_lo + cpw;

    // ---- window of one chunk: 3072 16-byte pieces, 12 per thread; 8 lanes = one pixel's 128
    // bytes.  The window holds x' = x * x_mul (the f32s input exponent, a power of two) clamped to
    // the fp16 range, and the range word is fed HERE, 48 values per thread and chunk: a sample is
    // a convex blend of window values times a mask in [0, 1] (dcn_v2.py:67), so neither the
    // clamp nor the running maximum is needed per sample (64 values per lane and tap).
    constexpr int NP = R_WPIX * 8 / R_NT;
    cn_f32x4 rw[NP];
    unsigned okbits = 0u;
    auto fill_load = [&](int chunk) {
        // all twelve requests first; the zero fill of off-map pixels happens at the LDS write (a
        // select right behind each load would wait for the loads one by one)
        okbits = 0u;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int i = p * R_NT + tid;
            const int wp = i >> 3, q = i & 7;
            const int wy = wp / R_WX, wx = wp - wy * R_WX;
            const int iy = wy0 + wy, ix = wx0 + wx;
            const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
            okbits |= ok ? (1u << p) : 0u;
            const unsigned off = ok ? (img_base + (unsigned)(iy * W + ix)) * pix_bytes + (unsigned)chunk * 128u + 16u * q : 0u;
            rw[p] = *reinterpret_cast<const d2_glb_f32x4 *>(xg + off);
        }
    };
    auto fill_store = [&]() {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int i = p * R_NT + tid;
            cn_f32x4 v = ((okbits >> p) & 1u) ? rw[p] * a_x_mul : zero4;
            cn_rng_upd4(rng_in, v);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __builtin_fminf(__builtin_fmaxf(v[e], -65504.0f), 65504.0f);
            const int wp = i >> 3, wy = wp / R_WX;
            *reinterpret_cast<cn_f32x4 *>(Win + wy * R_WLINE + (wp - wy * R_WX) * R_WROW + (i & 7) * 4) = v;
        }
    };

    // ---- prologue: offsets / masks of the tile (1152 records over 256 threads: every load is
    // requested before the first record is formed -- one memory latency for the lot), the window
    // of chunk 0 requested behind them, then the sampling records of all nine taps
    // (dcn_v2_im2col_cuda.cu:151-176)
    {
        constexpr int NR = (9 * R_PM + R_NT - 1) / R_NT;   // 5 (the last trip half full)
        float off_h[NR], off_w[NR], mkv[NR];
#pragma unroll
        for (int p = 0; p < NR; ++p) {
            const int i = min(p * R_NT + tid, 9 * R_PM - 1);
            const int tap = i / R_PM, m = i - tap * R_PM;
            const int oy = ty0 + (m >> 4), ox = tx0 + (m & 15);
            const float *om = a.om + (size_t)((b * H + oy) * W + ox) * a.om_pitch;
            off_h[p] = om[2 * tap];
            off_w[p] = om[2 * tap + 1];
            mkv[p] = om[18 + tap];
        }
        fill_load(c_lo);
#pragma unroll
        for (int p = 0; p < NR; ++p) {
            const int i = p * R_NT + tid;
            const int tap = i / R_PM, m = i - tap * R_PM;
            const int oy = ty0 + (m >> 4), ox = tx0 + (m & 15);
            float mk = mkv[p];
            if (msig) mk = d2_sigmoid(mk);              // dcn_v2.py:67
            const int ki = tap / 3, kj = tap - ki * 3;
            const float h_im = (float)(oy - 1 + ki) + off_h[p];
            const float w_im = (float)(ox - 1 + kj) + off_w[p];
            cn_i32x4 rec = {0, 0, 0, 0};
            if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {   // :165
                const float hf = floorf(h_im), wf = floorf(w_im);
                const int yl = (int)hf, xl = (int)wf;
                rec[0] = __float_as_int(h_im - hf);
                rec[1] = __float_as_int(w_im - wf);
                rec[2] = __float_as_int(mk);
                rec[3] = (yl & 0xffff) | (int)((uint32_t)(xl & 0xffff) << 16);
            } else {
                rec[3] = (oy & 0xffff) | (int)((uint32_t)(ox & 0xffff) << 16);   // mask = 0, corner in the window
            }
            if (i < 9 * R_PM) Rec[i] = rec;
        }
        fill_store();
    }
    __syncthreads();                               // records and the window of chunk 0 visible

    const int m = wave * 32 + l31;                 // this lane's pixel of the tile
    const char *wfrag = reinterpret_cast<const char *>(a.w) + (size_t)9 * a.cout_pad * a.cin_pad * 4;
    const int ncb = a.cout_pad >> 5;
    const int nb0 = n0 >> 5;
    cn_f32x16 acc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // One sample = one sampling record decoded: corner weights (zero where the corner is off the
    // map, dcn_v2_im2col_cuda.cu:30-45), mask, top-left corner, and whether the four corners lie
    // inside the window.
    struct Samp { float w1, w2, w3, w4, mk; int yl, xl; bool inwin; };
    auto decode = [&](const cn_i32x4 r) -> Samp {
        // by value through locals: __builtin_bit_cast / casts on a vector ELEMENT expression
        // have read element 0 whatever the index (hipcc 7.2)
        const int r0 = r[0], r1 = r[1], r2 = r[2];
        const uint32_t pk = (uint32_t)r[3];
        const float lh = __int_as_float(r0), lw = __int_as_float(r1);
        Samp s;
        s.mk = __int_as_float(r2);
        s.yl = (int)(short)(pk & 0xffffu);
        s.xl = (int)(short)(pk >> 16);
        const float hh = 1.f - lh, hw = 1.f - lw;
        const bool yl_ok = s.yl >= 0, xl_ok = s.xl >= 0;
        const bool yh_ok = s.yl + 1 <= H - 1, xh_ok = s.xl + 1 <= W - 1;
        s.w1 = (yl_ok && xl_ok) ? hh * hw : 0.f;
        s.w2 = (yl_ok && xh_ok) ? hh * lw : 0.f;
        s.w3 = (yh_ok && xl_ok) ? lh * hw : 0.f;
        s.w4 = (yh_ok && xh_ok) ? lh * lw : 0.f;
        s.inwin = (unsigned)(s.yl - wy0) <= (unsigned)(R_WY - 2) &&
                  (unsigned)(s.xl - wx0) <= (unsigned)(R_WX - 2) && !(dbg & 1);
        return s;
    };
    // The sixteen window reads of a sample (a lane whose sample lies beyond the window's reach
    // reads its own pixel's position; the values are replaced from global memory when the sample
    // is blended).  Explicit address spaces: left generic, the compiler turns the two sources
    // into flat loads of a selected pointer.
    cn_f32x4 c1[4], c2[4], c3[4], c4[4];
    auto request_corners = [&](const Samp &s) {
        const int wyl = s.inwin ? s.yl - wy0 : (m >> 4) + 1 + R_RCH;
        const int wxl = s.inwin ? s.xl - wx0 : (m & 15) + 1 + R_RCH;
        const d2_lds_f32x4 *c0 = reinterpret_cast<const d2_lds_f32x4 *>(
            win_lds + (unsigned)(wyl * R_WLINE + wxl * R_WROW + 8 * h) * 4u);
#pragma unroll
        for (int qi = 0; qi < 4; ++qi) {
            const int fo = (qi >> 1) * 4 + (qi & 1);          // in 16-byte units
            c1[qi] = c0[fo];
            c2[qi] = c0[R_WROW / 4 + fo];
            c3[qi] = c0[R_WLINE / 4 + fo];
            c4[qi] = c0[R_WLINE / 4 + R_WROW / 4 + fo];
        }
    };

    // ---- main loop, software-pipelined by hand: the window reads of step k+1 are requested
    // BEFORE the MFMAs of step k issue and land under them; the record of step k+2 is read one
    // step ahead of that.  (Records do not depend on the chunk: tap 0 follows tap 8.)
    // (64-wide tiles; a 128-wide tile has no registers for a second set of corners next to its
    // 64 accumulators and 64 weight registers: it requests them at the top of their own step.)
    constexpr bool PIPE = (BN == 64);
    Samp cur = decode(Rec[m]);
    if (PIPE) request_corners(cur);
    cn_i32x4 rnext = Rec[R_PM + m];
    for (int chunk = c_lo; chunk < c_hi; ++chunk) {
        if (chunk != c_lo) {
            __syncthreads();                       // every wave is done with the previous window
            fill_load(chunk);
            fill_store();
            __syncthreads();
            if (PIPE) request_corners(cur);        // tap 0 of the new chunk
        }
        const unsigned cb = (unsigned)chunk * 128u;
#pragma unroll 1
        for (int t = 0; t < ((dbg & 128) ? 0 : 9); ++t) {
            // weights of this (tap, chunk): the MFMA's A operand, straight from the fragment copy
            d2_f16x8 wf[NB][4];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int nb = min(nb0 + j, ncb - 1);
                const char *g = wfrag + (size_t)((t * a.nchunk + chunk) * ncb + nb) * 4096 + lane * 16;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) wf[j][kk] = *reinterpret_cast<const d2_f16x8 *>(g + kk * 1024);
            }
            if (!PIPE) {
                cur = decode(Rec[t * R_PM + m]);
                request_corners(cur);
            }
            // (w1*v1 + w2*v2 + w3*v3 + w4*v4) * mask  (dcn_v2_im2col_cuda.cu:43-45,174); the
            // power-of-two x_mul inside v1..v4 commutes with every rounding on the way
            cn_f32x4 vq[4];
#pragma unroll
            for (int qi = 0; qi < 4; ++qi)
                vq[qi] = c1[qi] * cur.w1 + c2[qi] * cur.w2 + c3[qi] * cur.w3 + c4[qi] * cur.w4;
            if (!cur.inwin) {
                // beyond the window's reach: the four corners from global memory (clamped
                // addresses; off-map corners carry zero weight), scaled and clamped like the
                // window, blended inside the branch (only the four blends cross the join)
                const int y0 = max(cur.yl, 0), y1 = min(cur.yl + 1, H - 1);
                const int x0 = max(cur.xl, 0), x1 = min(cur.xl + 1, W - 1);
                const d2_glb_char *g = xg + cb + 32u * h;
                const unsigned o1 = (img_base + (unsigned)(y0 * W + x0)) * pix_bytes;
                const unsigned o2 = (img_base + (unsigned)(y0 * W + x1)) * pix_bytes;
                const unsigned o3 = (img_base + (unsigned)(y1 * W + x0)) * pix_bytes;
                const unsigned o4 = (img_base + (unsigned)(y1 * W + x1)) * pix_bytes;
                cn_f32x4 g1[4], g2[4], g3[4], g4[4];
#pragma unroll
                for (int qi = 0; qi < 4; ++qi) {
                    const int fo = ((qi >> 1) * 16 + (qi & 1) * 4) * 4;
                    g1[qi] = *reinterpret_cast<const d2_glb_f32x4 *>(g + o1 + fo);
                    g2[qi] = *reinterpret_cast<const d2_glb_f32x4 *>(g + o2 + fo);
                    g3[qi] = *reinterpret_cast<const d2_glb_f32x4 *>(g + o3 + fo);
                    g4[qi] = *reinterpret_cast<const d2_glb_f32x4 *>(g + o4 + fo);
                }
                // drain here, inside the rare branch: waited for at the join, these loads would
                // make every step wait for its weight fragments as well (vmcnt counts in order)
                __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0)
                auto prep = [&](cn_f32x4 v) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = __builtin_fminf(__builtin_fmaxf(v[e] * a_x_mul, -65504.0f), 65504.0f);
                    return v;
                };
#pragma unroll
                for (int qi = 0; qi < 4; ++qi)
                    vq[qi] = prep(g1[qi]) * cur.w1 + prep(g2[qi]) * cur.w2 + prep(g3[qi]) * cur.w3 + prep(g4[qi]) * cur.w4;
            }
            cn_f16x4v shi[4], slo[4];
#pragma unroll
            for (int qi = 0; qi < 4; ++qi) {
                const cn_f32x4 v = vq[qi] * cur.mk;
                if (msig) {
                    cn_split4<false>(v, shi[qi], slo[qi]);      // |v| <= max |x'| <= 65504
                } else {
                    cn_rng_upd4(rng_in, v);                     // caller-supplied mask: any size
                    cn_split4<true>(v, shi[qi], slo[qi]);
                }
            }
            d2_f16x8 sf[4];
            sf[0] = __builtin_shufflevector(shi[0], shi[1], 0, 1, 2, 3, 4, 5, 6, 7);   // high, channels 8h..
            sf[1] = __builtin_shufflevector(shi[2], shi[3], 0, 1, 2, 3, 4, 5, 6, 7);   // high, 16 + 8h..
            sf[2] = __builtin_shufflevector(slo[0], slo[1], 0, 1, 2, 3, 4, 5, 6, 7);   // low parts
            sf[3] = __builtin_shufflevector(slo[2], slo[3], 0, 1, 2, 3, 4, 5, 6, 7);
            // next step: decode its record, request its corners, read the record after it
            if (PIPE) {
                cur = decode(rnext);
                request_corners(cur);    // (after tap 8 these are discarded: the swap requests its own)
                rnext = Rec[(t >= 7 ? t - 7 : t + 2) * R_PM + m];
                __builtin_amdgcn_sched_barrier(0);
            }
            if (dbg & 8) {
                acc[0][0] += (float)sf[0][0] + (float)sf[1][0] + (float)sf[2][0] + (float)sf[3][0] +
                             (float)wf[0][0][0] + (float)wf[NB - 1][3][0];
                continue;
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int term = 0; term < 3; ++term)     // w_lo*s_hi, w_hi*s_lo, w_hi*s_hi
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        const int kw = (term == 0) ? 2 + s2 : s2;
                        const int ks = (term == 1) ? 2 + s2 : s2;
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[j][kw], sf[ks], acc[j], 0, 0, 0);
                    }
            __builtin_amdgcn_s_setprio(0);
        }
    }

    // ---- epilogue: y = relu?((acc + bias) * scale + shift).  Each wave stages its 32 pixels x BN
    // channels in its own LDS region (the window and records are dead) and stores whole lines.
    __syncthreads();
    float *Cs = reinterpret_cast<float *>(smem) + wave * 32 * LDC;
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const cn_f32x4 v = {acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
            *reinterpret_cast<cn_f32x4 *>(Cs + l31 * LDC + 32 * j + 8 * g + 4 * h) = v;
        }
    constexpr int C4 = BN / 4;               // lanes per pixel row
    constexpr int RPP = 64 / C4;             // pixel rows per pass of the wave
    const int cq = lane % C4, rr = lane / C4;
    const int n = n0 + cq * 4;
    float bs[4], sc[4], sf2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const bool ok = (n + e) < a.Cout;
        bs[e] = (a.bias && ok) ? a.bias[n + e] : 0.f;
        sc[e] = (a.scale && ok) ? a.scale[n + e] : 1.f;
        sf2[e] = (a.shift && ok) ? a.shift[n + e] : 0.f;
    }
#pragma unroll
    for (int it = 0; it < 32 / RPP; ++it) {
        const int row = it * RPP + rr;
        const int mm = wave * 32 + row;
        const size_t off = (size_t)((b * H + ty0 + (mm >> 4)) * W + tx0 + (mm & 15));
        if (a.partial) {   // K split: raw sums, one slab per split; the reduce kernel does the rest
            if (n < a.cout_pad)
                *reinterpret_cast<cn_f32x4 *>(a.partial + ((size_t)blockIdx.z * ((size_t)a.B * H * W) + off) * a.cout_pad + n) =
                    *reinterpret_cast<const cn_f32x4 *>(Cs + row * LDC + cq * 4);
        } else if (n + 4 <= a.Cout) {
            cn_f32x4 v = *reinterpret_cast<const cn_f32x4 *>(Cs + row * LDC + cq * 4);
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
    if (a.range) {
        if (!a.out_plain && !a.partial) cn_rng_commit(a.range, 0, rng_out);
        cn_rng_commit(a.range, 1, rng_in);
    }
}

template <int BN>
int launch_dcn_reg(const Dcn2Args &a, hipStream_t st)
{
    dim3 grid((unsigned)(a.B * a.tiles_x * a.tiles_y), cn_cdiv(a.Cout, BN), (unsigned)a.ksplit);
    if (a.dbg && a.mask_sigmoid) {
        CN_SET_MAX_LDS_ONCE((dcn_reg_kernel<BN, true, true>), R_LDS);
        hipLaunchKernelGGL((dcn_reg_kernel<BN, true, true>), grid, dim3(R_NT), R_LDS, st, a);
    } else if (a.mask_sigmoid) {
        CN_SET_MAX_LDS_ONCE((dcn_reg_kernel<BN, false, true>), R_LDS);
        hipLaunchKernelGGL((dcn_reg_kernel<BN, false, true>), grid, dim3(R_NT), R_LDS, st, a);
    } else {
        CN_SET_MAX_LDS_ONCE((dcn_reg_kernel<BN, false, false>), R_LDS);
        hipLaunchKernelGGL((dcn_reg_kernel<BN, false, false>), grid, dim3(R_NT), R_LDS, st, a);
    }
    CN_CHECK_LAUNCH();
    return CN_OK;
}

}  // namespace

// Shapes this kernel takes (the caller falls back to the global-gather form otherwise): maps of
// whole 8 x 16 pixel blocks, whole 32-channel chunks, Cout a multiple of 4 and >= 33, and at least
// `min_wgs` workgroups (the tap-split gather form serves the small grids).
int cn_dcn_window_f32s(const float *x, const void *w_packed, const float *bias, const float *om,
                       int om_pitch, const float *scale, const float *shift, void *y, int out_pitch,
                       int out_plain, int B, int Cin, int H, int W, int Cout, int mask_sigmoid, int relu,
                       float x_mul, uint32_t *range, int min_wgs, int dbg, float *partial,
                       size_t partial_bytes, int *ksplit_out, hipStream_t st)
{
    if (ksplit_out) *ksplit_out = 1;
    if ((H & 7) || (W & 15) || (Cin & 31) || (Cout & 3) || Cout <= 32) return CN_ERR_UNSUPPORTED;
    if (H > 32767 || W > 32767 || (out_pitch & 3) || !cn_aligned16(y) || !cn_aligned16(x)) return CN_ERR_UNSUPPORTED;
    if ((om_pitch & 1) || (((uintptr_t)om) & 7u)) return CN_ERR_UNSUPPORTED;   // (offset pairs are 8-byte loads)
    if ((size_t)B * H * W * Cin * 4 >= ((size_t)1 << 32)) return CN_ERR_UNSUPPORTED;   // 32-bit byte offsets
    const int bn = Cout > 64 ? 128 : 64;
    const long wgs = (long)B * (H / TS) * (W / R_TX) * cn_cdiv(Cout, bn);
    // Too few tiles for the chip but a deep K (512 -> 256 @ 16^2): split the 32-channel chunks over
    // 2 / 4 / 8 workgroups per tile -- raw fp32 partial sums in the caller's workspace, summed in a
    // fixed order by splitk_reduce_kernel (deterministic) -- when that yields >= 256 workgroups
    int ksplit = 1;
    if (wgs < cn_tune_dcn_wgs || wgs < min_wgs) {
        // the smallest split that reaches the workgroup target, else the deepest one that fits
        const int nchunk = (Cin + 31) / 32;
        const int cout_pad = (Cout + 31) / 32 * 32;
        for (int s2 = 2; s2 <= 8 && partial && wgs * ksplit < cn_tune_dcn_wgs; s2 *= 2)
            if (nchunk % s2 == 0 && nchunk / s2 >= 2 &&
                (size_t)s2 * B * H * W * cout_pad * sizeof(float) <= partial_bytes)
                ksplit = s2;
        if (wgs * ksplit < 256 && wgs < min_wgs) return CN_ERR_UNSUPPORTED;   // (the tap-split gather form serves these)
    }
    Dcn2Args a = {};
    a.x = x; a.w = w_packed; a.bias = bias; a.scale = scale; a.shift = shift; a.om = om; a.y = y;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.om_pitch = om_pitch;
    a.mask_sigmoid = mask_sigmoid; a.relu = relu; a.out_pitch = out_pitch; a.out_plain = out_plain;
    a.cin_pad = (Cin + 31) / 32 * 32;
    a.cout_pad = (Cout + 31) / 32 * 32;
    a.nchunk = a.cin_pad / 32;
    a.tiles_x = W / R_TX;
    a.tiles_y = H / TS;
    a.x_mul = x_mul; a.range = range; a.dbg = dbg;
    a.ksplit = ksplit;
    a.partial = ksplit > 1 ? partial : nullptr;
    if (ksplit_out) *ksplit_out = ksplit;
    if (bn == 64) return launch_dcn_reg<64>(a, st);
    return launch_dcn_reg<128>(a, st);
}
