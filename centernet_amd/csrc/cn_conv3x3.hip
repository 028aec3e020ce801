// cn_conv3x3.hip -- 3x3 / stride-1 / pad-1 convolution, the bulk of every CenterNet backbone
// (resnet_dcn.py:38-67 BasicBlock convs, the 3x3 of every head :155-177, conv_offset_mask of
// DCN dcn_v2.py:52-57, DLA tree blocks pose_dla_dcn.py:31-62, hourglass residuals
// large_hourglass.py:48-74), as an im2col-free LDS tiling on the fp32 matrix cores.
// Template variants of the same kernel: HEADS (the whole head stack conv3x3+ReLU+conv1x1 of
// all heads in one launch), DECONV (ConvTranspose2d 4x4/2 as four parity 2x2 convolutions over
// the same halo, resnet_dcn.py:228-235), KSKIP (Cin % 32 != 0), fp16 (fp32 accumulate).
//
// Why a second kernel next to the generic implicit GEMM (cn_conv.hip): there the A tile of
// every (tap, 32-channel chunk) is re-fetched from global memory and re-written to LDS, and
// an ablation (tools/bench_kernels.py, cn_set_tuning key 9) showed that this staging costs
// 25-35 % of the kernel (MFMA loop alone: 137-147 TFLOP/s; with staging: 88-110).  Here a
// workgroup owns a TH x TW block of output pixels of one image and stages, per 32-channel
// chunk, the (TH+2) x (TW+2) input HALO once; the nine taps read their A fragments from that
// one LDS image at a constant address offset ((ky*(TW+2)+kx) rows).  A traffic and A staging
// instructions drop ~6x; only the weight tile is re-staged per tap (double-buffered, one
// barrier per tap).
//
// Same numerics as the generic kernel: v_mfma_f32_32x32x2_f32, fp32 accumulate, K consumed
// chunk-major then tap, epilogue y = relu?(acc*scale + shift + residual).
#include "cn_common.h"
#include <type_traits>

// cn_set_tuning key 18: phase shift of co-resident workgroups, percent of one tile's MFMA time
// (0 = off); see the kernel prologue
int cn_tune_stagger_pct = 100;
int cn_tune_heads_reg = 1;     // cn_set_tuning key 26 (A/B): fused f32s heads with the hidden layer in registers
int cn_tune_heads_remap = 1;   // cn_set_tuning key 24 (A/B): bit 0 = fused heads, bit 1 = multi-block Cout, on a 1-D row-interleaved grid
int cn_tune_f32s_policy = 0;   // cn_set_tuning key 21 (A/B): bit 0 = 128-wide tiles as eight waves three taps ahead, bit 1 = 64-wide tiles two taps ahead
// f32s: taps a weight tile is requested ahead of its use (1 = the fp32 schedule; build-time so that
// the register allocation of each form is its own: -DCN_F32S_PREFETCH_TAPS=1 for A/B builds)
#ifndef CN_F32S_PREFETCH_TAPS
#define CN_F32S_PREFETCH_TAPS 3
#endif
#ifndef CN_F16_PREFETCH_TAPS
#define CN_F16_PREFETCH_TAPS 2
#endif

bool cn_conv3x3p_takes(int B, int H, int W, int Cin, int Cout, int in_pitch, int out_pitch, int res_pitch,
                       bool in_plain, bool has_res);
bool cn_heads3x3p_takes(int B, int H, int W, int in_pitch, int head_conv, int n_heads, const cn_head_out *heads,
                        bool in_plain);
int cn_heads3x3p(const void *x, int B, int H, int W, int Cin, int in_pitch, const void *w1_packed,
                 const float *scale1, const float *bias1, int n_heads, const cn_head_out *heads,
                 const cn_f32s_ctl *ctl, hipStream_t st);
bool cn_deconv4x4s2p_takes(int B, int H, int W, int Cin, int Cout, int in_pitch, int out_pitch, bool in_plain);
int cn_deconv4x4s2_persist(const void *x, const void *w_packed, const float *scale, const float *shift, void *y,
                           int B, int H, int W, int Cin, int Cout, int in_pitch, int out_pitch, int relu,
                           int out_plain, const cn_f32s_ctl *ctl, hipStream_t st);
int cn_conv3x3s1_persist(const void *x, const void *w_packed, const float *scale, const float *shift,
                         const void *residual, void *y, int B, int H, int W, int Cin, int Cout,
                         int in_pitch, int out_pitch, int res_pitch, int relu, int out_plain, int res_plain,
                         const cn_f32s_ctl *ctl, hipStream_t st);

namespace {

constexpr int LDT = 36;  // floats per LDS row (32 + 4 pad = 144 bytes, conflict-free b128 reads)

typedef _Float16 c3_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 c3_f16x4 __attribute__((ext_vector_type(4)));
template <typename T> struct C3Elem;
template <> struct C3Elem<float> { static constexpr int EPV = 4; };
template <> struct C3Elem<_Float16> { static constexpr int EPV = 8; };
template <> struct C3Elem<cn_f32s> { static constexpr int EPV = 4; };   // 128-byte groups, as fp32
__device__ __forceinline__ cn_f32x4 c3_load4(const float *p) { return *reinterpret_cast<const cn_f32x4 *>(p); }
__device__ __forceinline__ cn_f32x4 c3_load4(const _Float16 *p)
{
    const c3_f16x4 h = *reinterpret_cast<const c3_f16x4 *>(p);
    cn_f32x4 r = {(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    return r;
}
__device__ __forceinline__ void c3_store4(float *p, cn_f32x4 v) { *reinterpret_cast<cn_f32x4 *>(p) = v; }
__device__ __forceinline__ void c3_store4(_Float16 *p, cn_f32x4 v)
{
    c3_f16x4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
    *reinterpret_cast<c3_f16x4 *>(p) = h;
}

struct C3Args {
    const void *x, *w;           // element type T (fp32 or fp16)
    const float *scale, *shift;
    const void *residual;        // element type T
    void *y;                     // element type T
    int B, H, W, Cin, Cout, in_pitch, out_pitch, relu;
    int res_pitch;               // pixel pitch of the residual (= out_pitch unless stated)
    int cin_pad, cout_pad, nchunk, tiles_x, tiles_y, vec_out, setprio;
    int bm256, waves8, occ4;     // tile-shape knobs (cn_set_tuning keys 14, 15, 19)
    int nkk_last;                // KSKIP: 8-channel K groups of the last chunk that hold data
    int dbg;                     // ablation switches (cn_set_tuning key 9)
    int stagger, stagger_slots;  // phase shift of co-resident workgroups (cycles per slot, slots)
    int in_plain, out_plain, res_plain;  // f32s kernels: x / y / residual are plain fp32 tensors
    int ncb;                     // f32s: 32-channel output blocks in the packed weight (cout_pad / 32)
    int heads_remap;             // number of grid rows (heads / output-channel blocks) when the grid is 1-D and row-interleaved (0: blockIdx.y = row)
    size_t wfrag_off;            // f32s: byte offset of the fragment-ordered weight copy behind the row-ordered one
    float x_mul, res_mul;        // f32s range control (cn_f32s_ctl): plain-x and residual multipliers
    uint32_t *range;             // f32s: [0] max |stored output| / hidden tile, [1] max |split plain input|
};

// Fused detection heads (HEADS = true): blockIdx.y selects the head; its 64 hidden channels
// never leave the workgroup.
constexpr int MAX_HEADS = 8;
constexpr int HEAD_CONV = 64;          // hidden width this kernel is built for (= BN)
constexpr int LDS2 = HEAD_CONV + 4;    // floats per LDS row of the second GEMM (272 bytes)
constexpr int W2_ROWS = 96;            // 1x1 output channels staged per pass
struct C3Heads {
    const float *w[MAX_HEADS];     // (cout, 64) row-major
    const float *bias[MAX_HEADS];  // (cout) or null
    float *y[MAX_HEADS];           // (B, cout, H, W)
    int cout[MAX_HEADS];
    int slices;                    // hidden width / 64: 64-channel slices of the hidden layer (1..4)
    const float *oscale[MAX_HEADS];  // (cout) or null: y = acc * oscale + bias (cn_head_out.oscale)
};

// LDS floats of the kernel: the main loop's tiles, unioned with the epilogue staging
// rows of the output tile staged through LDS per epilogue pass: the whole tile when it fits
// in ~68 KB (keeps two 128 x 128 workgroups per CU), else one wave row
template <int BN, int WM, int BM>
constexpr int c3_epi_rows()
{
    return ((size_t)BM * (BN + 4) <= 17408) ? BM : BM / WM;
}

// fused heads with the hidden layer in registers: [2][BN] hidden scale / shift of the slice,
// [W2_ROWS][2] bias / output scale of the 1x1 rows (behind rowoff)
template <int BN, int WM, bool HEADS>
constexpr size_t c3_table_floats() { return (HEADS && WM == 4) ? (size_t)(2 * BN + 2 * W2_ROWS) : 0; }

template <int TW, int BN, int WM, bool HEADS, int BM, int NBUFB = 2>
constexpr size_t c3_union_floats()
{
    constexpr int TH = BM / TW;
    constexpr size_t tiles = (size_t)((TH + 2) * (TW + 2) * LDT + NBUFB * BN * LDT);
    // fused heads: S[128][LDS2] shares the main loop's tile space; the 1x1 weights sit behind it
    constexpr size_t cs = HEADS ? (size_t)128 * LDS2 : (size_t)c3_epi_rows<BN, WM, BM>() * (BN + 4);
    // register-resident hidden layer (fused heads as 4 x 1 waves, f32s): no S, no 1x1 weights in LDS
    if (HEADS && WM == 4) return tiles;
    return (tiles > cs ? tiles : cs) + (HEADS ? (size_t)W2_ROWS * LDS2 : 0);
}

// T = float: v_mfma_f32_32x32x2_f32, 32 channels per chunk; T = fp16: v_mfma_f32_32x32x16_f16
// (fp32 accumulate), 64 channels per chunk -- same 128-byte LDS rows and read addresses.
// T = cn_f32s: fp32 values as (high, low) fp16 pairs, 32 channels per 128-byte row; every
// 16-deep K step is three v_mfma_f32_32x32x16_f16 (hi*hi + hi*lo + lo*hi), see cn_common.h.
// NBUFB: LDS buffers of the per-tap weight tile (2: one barrier per tap; 1: two barriers per tap
// but 9 KB less LDS -- the 4-workgroups-per-CU variant below; 0 (f32s only): NO weight tile in
// LDS -- every wave streams its B fragments straight from the fragment-ordered copy of the packed
// weight (L1/L2-resident, 4 KB contiguous per 32 output channels x 32 input channels) into a
// double-buffered register set one tap ahead, so the only barriers left are the two around a
// halo restage per 32-channel chunk.  With three fp16 MFMAs per product a tap is only ~400
// cycles of matrix work: the per-tap weight staging + barrier of the LDS form cost more than that.
template <typename T, int TW, int BN, int WM, int WN, bool HEADS = false, int BM = 128,
          bool KSKIP = false, bool DECONV = false, int NBUFB = 2, int PDQ = CN_F32S_PREFETCH_TAPS>
__device__ __forceinline__ void conv3x3s1_body(const C3Args &a, const C3Heads &hd)
{
    // DECONV: ConvTranspose2d(4, stride 2, pad 1) -- blockIdx.z = output parity (py, px); each
    // parity is a 2x2 convolution over the same input halo (taps (ty+py, tx+px) of the 3x3
    // neighbourhood) with its own weights, writing out[2y+py][2x+px] (resnet_dcn.py:228-235)
    constexpr int NTAPS = DECONV ? 4 : 9;
    constexpr int TAPW = DECONV ? 2 : 3;
    constexpr bool SPLIT = std::is_same<T, cn_f32s>::value;
    constexpr bool WREG = (NBUFB == 0);
    static_assert(!WREG || SPLIT, "register-streamed weights are built for f32s");
    static_assert(!DECONV || (!HEADS && !KSKIP && sizeof(T) == 4), "deconv variant: fp32 / f32s");
    static_assert(!SPLIT || !KSKIP, "f32s multiplies the zero-padded channels");
    const int par_y = DECONV ? (int)(blockIdx.z >> 1) : 0, par_x = DECONV ? (int)(blockIdx.z & 1) : 0;
    constexpr int NT = WM * WN * 64;  // 4 waves (256 threads) or 8 waves (512 threads)
    constexpr int RPP = NT / 8;       // LDS rows staged per pass of the block
    static_assert(!HEADS || NT == 256, "fused heads are built for 4 waves");
    static_assert(!HEADS || BM == 128, "fused heads are built for 128-pixel tiles");
    // HREG: the head's hidden layer never leaves the registers (4 x 1 waves, f32s; BN = 64 or 128)
    constexpr bool HREG = HEADS && SPLIT && WM == 4 && WN == 1;
    static_assert(!HEADS || (WM == 4) == HREG, "4 x 1 fused heads are the f32s register form");
    static_assert(!HEADS || sizeof(T) == 4, "fused heads: fp32 / f32s");
    static_assert(!HEADS || HREG || BN == HEAD_CONV, "LDS-staged fused heads: 64 hidden channels per slice");
    static_assert(LDS2 * 4 >= 2 * 128 + 16, "a hidden row holds two 128-byte f32s groups + the bias column");
    constexpr int EPV = C3Elem<T>::EPV;
    constexpr int BKE = 8 * EPV;
    constexpr bool F16 = (EPV == 8);
    const T *xT = reinterpret_cast<const T *>(a.x);
    const T *wT = reinterpret_cast<const T *>(a.w) +
                  (DECONV ? (size_t)blockIdx.z * NTAPS * a.cout_pad * a.cin_pad : (size_t)0);
    constexpr int TH = BM / TW;
    constexpr int HW_ = TW + 2;              // halo width
    constexpr int HR = (TH + 2) * HW_;       // halo rows (pixels)
    constexpr int NPA = (HR + RPP - 1) / RPP;  // halo load passes per thread
    constexpr int PB = BN / RPP;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int MB = TM / 32, NB = TN / 32;
    static_assert((NT == 256 || NT == 512) && BN % RPP == 0 && TM % 32 == 0 && TN % 32 == 0,
                  "wave tiling");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int A_FLOATS = HR * LDT;
    constexpr int UNION = (int)c3_union_floats<TW, BN, WM, HEADS, BM, NBUFB>();
    float *As = reinterpret_cast<float *>(smem);  // [HR][LDT]
    float *Bs = As + A_FLOATS;                    // [2][BN][LDT]
    int *rowoff = reinterpret_cast<int *>(As + UNION);  // [BM]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = tid >> 3, q = tid & 7;
    const float a_x_mul = a.x_mul, a_res_mul = a.res_mul;   // scalars, not a stack copy of the struct tail
    uint32_t *const a_range = a.range;
    const int tiles = a.tiles_x * a.tiles_y;
    // 1-D row-interleaved grid (fused heads; key 24 bit 1: output-channel blocks too): the heads
    // of one pixel tile are workgroups 8 ids apart inside a run of 8 * heads consecutive ids -- dispatched together and onto the SAME XCD (id % 8), so the
    // tile's input halo is fetched into that XCD's L2 once instead of once per head at three
    // different times (the 2-D grid ran all tiles of head 0, then head 1, ...: 4.7x the feature
    // map in HBM / MALL reads, r02 counters).
    int vbx = blockIdx.x, vby = blockIdx.y;
    if (a.heads_remap) {
        const int run = 8 * a.heads_remap;
        const int g = vbx / run, r = vbx - g * run;
        vby = r >> 3;
        vbx = g * 8 + (r & 7);
    }
    const int b = vbx / tiles;
    const int tr = vbx - b * tiles;
    const int ty0 = (tr / a.tiles_x) * TH, tx0 = (tr % a.tiles_x) * TW;
    int n0 = vby * BN;   // fused heads: first hidden channel of the current slice (set per slice)
    const cn_f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    if (a.stagger) {
        // Phase shift of the workgroups that share a CU.  Workgroups of one launch all take the
        // same time, so the 2-3 residents of a CU start, reach their epilogue and get replaced
        // TOGETHER, round after round, and the matrix pipe idles during every such
        // prologue/epilogue window (ablation: 14 % of a 64->64 layer).  Delaying the first
        // occupant of slot s (dispatch is breadth-first: linear id / 256) by s * stagger cycles
        // once makes the residents take turns for the rest of the launch: 64->64@128^2 in a
        // back-to-back micro-benchmark 101.6 -> 118.4 TFLOP/s (tools/ablate_halo.py STAG=...),
        // inside the network 113 -> 115 (the layers are less lock-stepped there).  Only used
        // when the launch runs enough rounds to amortise the initial delay.
        const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const unsigned slot = lin / 256;
        if (slot >= 1 && slot < (unsigned)a.stagger_slots) {
            const unsigned long long t0 = __builtin_readcyclecounter();
            const unsigned long long wait = (unsigned long long)slot * (unsigned)a.stagger;
            while (__builtin_readcyclecounter() - t0 < wait) __builtin_amdgcn_s_sleep(32);
        }
    }
    // ---- per-thread halo rows: pixel offset in the input (or -1: outside image / halo)
    // as BYTE offsets of the lane's 16-byte slot in chunk 0 (tensors < 4 GiB, checked by the
    // caller): loads are SGPR base + 32-bit lane offset, no 64-bit address arithmetic in the loop
    constexpr unsigned NOPIX = 0xffffffffu;
    unsigned hoff[NPA];
#pragma unroll
    for (int p = 0; p < NPA; ++p) {
        const int hr = p * RPP + lrow;
        const int hy = hr / HW_, hx = hr - hy * HW_;
        const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
        hoff[p] = (hr < HR && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
                      ? (unsigned)(((b * a.H + iy) * a.W + ix) * a.in_pitch + EPV * q) * (unsigned)sizeof(T)
                      : NOPIX;
    }
    for (int m = tid; m < BM; m += NT) {
        const int ty = m / TW, tx = m - ty * TW;
        const int oy = ty0 + ty, ox = tx0 + tx;
        if (DECONV)
            rowoff[m] = (oy < a.H && ox < a.W)
                            ? (b * 2 * a.H + 2 * oy + par_y) * 2 * a.W + 2 * ox + par_x : -1;
        else
            rowoff[m] = (oy < a.H && ox < a.W) ? (b * a.H + oy) * a.W + ox : -1;
    }

    cn_f32x16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    cn_f32x4 ra[NPA], rb[PB];
    float rng_in = 0.f, rng_out = 0.f;   // largest |value| split on the input / output side
    auto load_A = [&](int chunk) {
        const int c = chunk * BKE + EPV * q;
        // f32s input: a 128-byte group holds 32 channels as high / low halves, so a 16-byte slot
        // is not "4 channels": the whole group is taken (channels past Cin are stored as zeros)
        const bool cok = (SPLIT && !a.in_plain) ? (chunk * BKE < a.Cin) : (c < a.Cin);
        const unsigned coff = (unsigned)(chunk * BKE) * (unsigned)sizeof(T);
        const char *xb = reinterpret_cast<const char *>(xT);
#pragma unroll
        for (int p = 0; p < NPA; ++p) {
            const bool ok = hoff[p] != NOPIX && cok;
            const cn_f32x4 v = *reinterpret_cast<const cn_f32x4 *>(xb + (ok ? hoff[p] + coff : 0u));
            ra[p] = ok ? v : zero4;
        }
    };
    auto store_A = [&]() {
        if constexpr (SPLIT) {
            if (a.in_plain) {  // plain fp32 input: channels 4q..4q+3 split while staging
#pragma unroll
                for (int p = 0; p < NPA; ++p) {
                    const int hr = p * RPP + lrow;
                    cn_f16x4v hi, lo;
                    const cn_f32x4 xs = ra[p] * a_x_mul;   // real -> stored units (a power of two)
                    cn_rng_upd4(rng_in, xs);
                    cn_split4(xs, hi, lo);
                    if (hr < HR) {
                        char *row = reinterpret_cast<char *>(As + hr * LDT);
                        *reinterpret_cast<cn_f16x4v *>(row + 8 * q) = hi;
                        *reinterpret_cast<cn_f16x4v *>(row + 64 + 8 * q) = lo;
                    }
                }
                return;
            }
        }
#pragma unroll
        for (int p = 0; p < NPA; ++p) {
            const int hr = p * RPP + lrow;
            if (hr < HR) *reinterpret_cast<cn_f32x4 *>(As + hr * LDT + 4 * q) = ra[p];
        }
    };
    // weight rows of this thread: byte offsets inside one tap's [cout_pad][cin_pad] matrix
    unsigned boff[PB];
    auto set_weight_rows = [&]() {
#pragma unroll
        for (int p = 0; p < PB; ++p)
            boff[p] = (unsigned)(min(n0 + p * RPP + lrow, a.cout_pad - 1) * a.cin_pad + EPV * q) * (unsigned)sizeof(T);
    };
    set_weight_rows();
    auto wtile = [&](int chunk, int tap) {   // uniform base of (tap, chunk)
        return reinterpret_cast<const char *>(wT) +
               ((size_t)tap * a.cout_pad * a.cin_pad + (size_t)chunk * BKE) * sizeof(T);
    };
    auto load_B = [&](int chunk, int tap) {
        const char *wb = wtile(chunk, tap);
#pragma unroll
        for (int p = 0; p < PB; ++p) rb[p] = *reinterpret_cast<const cn_f32x4 *>(wb + boff[p]);
    };
    auto store_B = [&](int buf) {
        float *Bd = Bs + buf * BN * LDT;
#pragma unroll
        for (int p = 0; p < PB; ++p)
            *reinterpret_cast<cn_f32x4 *>(Bd + (p * RPP + lrow) * LDT + 4 * q) = rb[p];
    };

    // A-fragment base of this lane's pixel in every M block (halo row of tap (0,0))
    int abase[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        const int m = wm * TM + i * 32 + l31;
        const int ty = m / TW, tx = m - ty * TW;
        abase[i] = (ty * HW_ + tx) * LDT + 4 * lh;
    }
    // KSKIP (Cin % 32 != 0, e.g. DLA level0's 16 channels): the last chunk's all-zero 8-channel
    // K groups are not multiplied at all
#ifdef CN_ABLATE
    c3_f16x8 abl_af[4][MB], abl_bf[4][NB];
    bool abl_loaded = false;
#endif
    auto compute = [&](int tap, int buf, int nkk) {
        const int ky = tap / TAPW + par_y, kx = tap % TAPW + par_x;
        const int toff = (ky * HW_ + kx) * LDT;
        const float *Bb = Bs + buf * BN * LDT + (wn * TN + l31) * LDT + 4 * lh;
        if (a.setprio) __builtin_amdgcn_s_setprio(1);
        if constexpr (SPLIT) {
            // the row's four 32-byte quarters: high parts k 0-15, 16-31, low parts k 0-15, 16-31
#ifdef CN_ABLATE   // variant builds only (tools/build_variant.sh): fragments outlive the call
            c3_f16x8 (&af)[4][MB] = abl_af; c3_f16x8 (&bf)[4][NB] = abl_bf;
            if (!(a.dbg & 16) || !abl_loaded)
#else
            c3_f16x8 af[4][MB], bf[4][NB];
#endif
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                for (int i = 0; i < MB; ++i)
                    af[kk][i] = *reinterpret_cast<const c3_f16x8 *>(As + abase[i] + toff + kk * 8);
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    bf[kk][j] = *reinterpret_cast<const c3_f16x8 *>(Bb + j * 32 * LDT + kk * 8);
            }
#ifdef CN_ABLATE
            abl_loaded = true;
#endif
            // every fragment read is issued before the first MFMA and stays there (cn_conv.hip:
            // a ds_read sunk behind an MFMA into that MFMA's operand registers can overwrite them
            // before a queued MFMA has read them)
            __builtin_amdgcn_sched_barrier(0);
            // smallest terms first (lo*hi, hi*lo, then hi*hi); independent accumulators interleaved
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int i = 0; i < MB; ++i)
#pragma unroll
                        for (int j = 0; j < NB; ++j) {
                            const int ka = (term == 0) ? 2 + s : s;
                            const int kb = (term == 1) ? 2 + s : s;
                            if constexpr (HREG)   // D[hidden][pixel]: the hidden layer lands lane = pixel
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[kb][j], af[ka][i],
                                                                                   acc[i][j], 0, 0, 0);
                            else
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ka][i], bf[kb][j],
                                                                                   acc[i][j], 0, 0, 0);
                        }
            if (a.setprio) __builtin_amdgcn_s_setprio(0);
            return;
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (KSKIP && kk >= nkk) break;
            cn_f32x4 af[MB], bf[NB];
#pragma unroll
            for (int i = 0; i < MB; ++i)
                af[i] = *reinterpret_cast<const cn_f32x4 *>(As + abase[i] + toff + kk * 8);
#pragma unroll
            for (int j = 0; j < NB; ++j)
                bf[j] = *reinterpret_cast<const cn_f32x4 *>(Bb + j * 32 * LDT + kk * 8);
            if constexpr (F16) {
#pragma unroll
                for (int i = 0; i < MB; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                            __builtin_bit_cast(c3_f16x8, af[i]), __builtin_bit_cast(c3_f16x8, bf[j]),
                            acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < MB; ++i)
#pragma unroll
                        for (int j = 0; j < NB; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
            }
        }
        if (a.setprio) __builtin_amdgcn_s_setprio(0);
    };

    // fused heads: rows [g0, g0 + W2_ROWS) of the head's 1x1 weight, hidden channels of slice sl,
    // + their bias in pad column 64 -- staged at the START of a slice so that the second GEMM
    // begins without a global round trip (the W2 region lies behind the main loop's tiles)
    auto stage_W2 = [&](int g0, int sl) {
        float *W2 = As + UNION - W2_ROWS * LDS2;
        const int head = vby;
        const int cout2 = hd.cout[head];
        const float *w2 = hd.w[head];
        const float *b2 = hd.bias[head];
        const int hidden = hd.slices * HEAD_CONV;
        const int rows = min(W2_ROWS, (cout2 - g0 + 31) / 32 * 32);
        for (int idx = tid; idx < rows * (HEAD_CONV / 4); idx += NT) {
            const int row = idx / (HEAD_CONV / 4), k4 = idx - row * (HEAD_CONV / 4);
            const int src = min(g0 + row, cout2 - 1);  // padded rows: computed, never stored
            const cn_f32x4 wv = *reinterpret_cast<const cn_f32x4 *>(
                w2 + (size_t)src * hidden + sl * HEAD_CONV + k4 * 4);
            if constexpr (SPLIT) {  // rows of two 128-byte groups [32 high | 32 low] (k = 4*k4 ..)
                cn_f16x4v hi, lo;
                cn_split4(wv, hi, lo);
                char *g = reinterpret_cast<char *>(W2 + row * LDS2) + (k4 >> 3) * 128 + (k4 & 7) * 8;
                *reinterpret_cast<cn_f16x4v *>(g) = hi;
                *reinterpret_cast<cn_f16x4v *>(g + 64) = lo;
            } else {
                *reinterpret_cast<cn_f32x4 *>(W2 + row * LDS2 + k4 * 4) = wv;
            }
        }
        const float *os2 = hd.oscale[head];
        for (int row = tid; row < rows; row += NT) {
            W2[row * LDS2 + HEAD_CONV] = (b2 && g0 + row < cout2) ? b2[g0 + row] : 0.f;
            W2[row * LDS2 + HEAD_CONV + 1] = (os2 && g0 + row < cout2) ? os2[g0 + row] : 1.f;
        }
    };

    auto main_loop = [&]() {
    if constexpr (WREG) {
        // ---- register-streamed weights, software-pipelined one tap ahead.  A lane's four
        // 16-byte quarter fragments of (tap, chunk, 32-channel output block) sit 1 KiB apart in the
        // fragment-ordered weight copy (a wave's load of one quarter is 1 KiB of contiguous memory).  Both operand sets (A from the LDS halo, B
        // from global) are double-buffered in registers and the set a load targets was last read
        // by MFMAs issued a full term earlier -- never by the MFMA just issued (see the operand
        // hazard note in cn_conv.hip).
        const char *wf = reinterpret_cast<const char *>(a.w) + (size_t)a.wfrag_off +
                         (DECONV ? (size_t)blockIdx.z * NTAPS * a.nchunk * a.ncb * 4096 : (size_t)0);
        int nbk[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) nbk[j] = min((n0 + wn * TN) / 32 + j, a.ncb - 1);
        c3_f16x8 afr[2][4][MB], bfr[2][4][NB];
        auto load_Bf = [&](c3_f16x8 (*dst)[NB], int chunk, int tap) {
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const char *src = wf + (((size_t)tap * a.nchunk + chunk) * a.ncb + nbk[j]) * 4096 + lane * 16;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    dst[kk][j] = *reinterpret_cast<const c3_f16x8 *>(src + kk * 1024);
            }
        };
        auto load_Af = [&](c3_f16x8 (*dst)[MB], int tap) {
            const int ky = tap / TAPW + par_y, kx = tap % TAPW + par_x;
            const int toff = (ky * HW_ + kx) * LDT;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < MB; ++i)
                    dst[kk][i] = *reinterpret_cast<const c3_f16x8 *>(As + abase[i] + toff + kk * 8);
        };
        auto mfma_term = [&](int term, const c3_f16x8 (*af)[MB], const c3_f16x8 (*bf)[NB]) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int i = 0; i < MB; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        const int ka = (term == 0) ? 2 + s2 : s2;   // lo*hi, hi*lo, hi*hi
                        const int kb = (term == 1) ? 2 + s2 : s2;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ka][i], bf[kb][j],
                                                                           acc[i][j], 0, 0, 0);
                    }
        };
        load_A(0);
        load_Bf(bfr[0], 0, 0);
        store_A();
        __syncthreads();
        load_Af(afr[0], 0);
        // one chunk = NTAPS MFMA phases without a barrier; P = register set of its first tap
        auto do_chunk = [&](auto P, int c) {
            constexpr int p0 = decltype(P)::value;
            const bool next_chunk = (c + 1) < a.nchunk;
            if (next_chunk) load_A(c + 1);   // halo of the next chunk: in flight behind the MFMAs
            if (a.setprio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int t = 0; t < NTAPS; ++t) {
                const int cur = (p0 + t) & 1;
                __builtin_amdgcn_sched_barrier(0);
                mfma_term(0, afr[cur], bfr[cur]);
                __builtin_amdgcn_sched_barrier(0);
                if (t + 1 < NTAPS) {
                    load_Bf(bfr[cur ^ 1], c, t + 1);
                    load_Af(afr[cur ^ 1], t + 1);
                } else if (next_chunk) {
                    load_Bf(bfr[cur ^ 1], c + 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                mfma_term(1, afr[cur], bfr[cur]);
                mfma_term(2, afr[cur], bfr[cur]);
            }
            if (a.setprio) __builtin_amdgcn_s_setprio(0);
            if (next_chunk) {
                __syncthreads();  // every wave is done with the old halo
                store_A();
                __syncthreads();
                load_Af(afr[(p0 + NTAPS) & 1], 0);
            }
        };
        for (int c = 0; c < a.nchunk; c += 2) {
            do_chunk(std::integral_constant<int, 0>{}, c);
            if (c + 1 < a.nchunk) do_chunk(std::integral_constant<int, NTAPS & 1>{}, c + 1);
        }
        __syncthreads();  // all waves out of the main loop before the epilogue reuses the LDS
    } else {
        // ---- f32s: weight tiles prefetched PD taps ahead.  With three fp16 MFMAs per product a
        // (tap, chunk) step is ~400 matrix cycles (0.17 us); a weight tile requested one step
        // ahead (the fp32 schedule below) arrives after ~1-1.5 us under load, so every step ended
        // up waiting for L2.  Here the request for step i + PD is issued when step i starts, into
        // a rotating register queue (slots static: NTAPS % PD == 0, the tap loop is unrolled), and
        // the next chunk's halo is requested AHEAD taps before it is needed.
        // (fp16 tensors: CN_F16_PREFETCH_TAPS, a chunk of 64 channels is 4 MFMAs per block)
        constexpr bool DEEP_T = SPLIT || (F16 && !HEADS && !DECONV && !KSKIP && CN_F16_PREFETCH_TAPS > 1);
        constexpr int PDR = SPLIT ? PDQ : CN_F16_PREFETCH_TAPS;
        constexpr int PD = (DEEP_T && NBUFB == 2 && PDR > 1) ? (DECONV ? 2 : PDR) : 1;
        if constexpr (PD > 1) {
            {
                // queue slot of step (chunk, tap) = (chunk * NTAPS + tap) % PD: static inside one
                // unrolled pass over the taps once the chunk's phase is a template constant
                constexpr int CP = (NTAPS % PD == 0) ? 1 : PD;   // chunks per slot period
                static_assert(CP == 1 || PD == 2, "slot phases");
                constexpr int AHEAD = NTAPS > 4 ? NTAPS - 4 : 0;
                cn_f32x4 rq[PD][PB];
                auto issue_B = [&](cn_f32x4 *dst, int chunk, int tap) {
                    const char *wb = wtile(chunk, tap);
#pragma unroll
                    for (int p = 0; p < PB; ++p) dst[p] = *reinterpret_cast<const cn_f32x4 *>(wb + boff[p]);
                };
                auto put_B = [&](const cn_f32x4 *src, int buf) {
                    float *Bd = Bs + buf * BN * LDT;
#pragma unroll
                    for (int p = 0; p < PB; ++p)
                        *reinterpret_cast<cn_f32x4 *>(Bd + (p * RPP + lrow) * LDT + 4 * q) = src[p];
                };
                load_A(0);
#pragma unroll
                for (int u = 0; u < PD; ++u) issue_B(rq[u], 0, u);
                store_A();
                put_B(rq[0], 0);
                __syncthreads();
                auto taps_of_chunk = [&](auto P, int c) {
                    constexpr int p0 = decltype(P)::value;   // (c * NTAPS) % PD
                    const bool nextc = (c + 1) < a.nchunk;
#pragma unroll
                    for (int t = 0; t < NTAPS; ++t) {
                        const int it = c * NTAPS + t;
                        const bool wrap = (t + PD) >= NTAPS;
                        // step it + PD goes into the slot step `it` left when it was put into LDS
                        if (!wrap || nextc) issue_B(rq[(p0 + t) % PD], wrap ? c + 1 : c, (t + PD) % NTAPS);
                        if (t == AHEAD && nextc) load_A(c + 1);
                        compute(t, it & 1, 4);
                        if (t == NTAPS - 1 && nextc) {
                            __syncthreads();  // every wave is done with the old halo
                            store_A();
                        }
                        if (t + 1 < NTAPS || nextc) put_B(rq[(p0 + t + 1) % PD], (it + 1) & 1);
                        __syncthreads();
                    }
                };
                if constexpr (CP == 1) {
                    for (int c = 0; c < a.nchunk; ++c) taps_of_chunk(std::integral_constant<int, 0>{}, c);
                } else {
                    // PD = 2 over an odd number of taps: walk the steps in pairs instead (slot
                    // and LDS buffer of a step = its parity, static; tap and chunk by counters)
                    const int total = a.nchunk * NTAPS;
                    int c = 0, t = 0;
                    auto step = [&](auto SL, int it) {
                        constexpr int sl = decltype(SL)::value;
                        const bool nextc = (c + 1) < a.nchunk;
                        if (it + 2 < total) {
                            const bool wrap = (t + 2) >= NTAPS;
                            issue_B(rq[sl], wrap ? c + 1 : c, wrap ? t + 2 - NTAPS : t + 2);
                        }
                        if (t == AHEAD && nextc) load_A(c + 1);
                        compute(t, sl, 4);
                        if (t == NTAPS - 1 && nextc) {
                            __syncthreads();  // every wave is done with the old halo
                            store_A();
                        }
                        if (it + 1 < total) put_B(rq[sl ^ 1], sl ^ 1);
                        __syncthreads();
                        if (++t == NTAPS) { t = 0; ++c; }
                    };
#pragma unroll 1
                    for (int it = 0; it < total; it += 2) {
                        step(std::integral_constant<int, 0>{}, it);
                        if (it + 1 < total) step(std::integral_constant<int, 1>{}, it + 1);
                    }
                }
            }
        } else {
    // ---- main loop: chunk-major, taps inner; B double-buffered, A halo single-buffered
        load_A(0);
        load_B(0, 0);
        store_A();
        store_B(0);
        __syncthreads();
        const int total = a.nchunk * NTAPS;
        int it = 0;
        for (int c = 0; c < a.nchunk; ++c) {
    #pragma unroll 1
            for (int t = 0; t < NTAPS; ++t, ++it) {
                const bool more = (it + 1) < total;
                const bool newA = (t == NTAPS - 1) && (c + 1 < a.nchunk);
                if (more && !(a.dbg & 2)) load_B(t == NTAPS - 1 ? c + 1 : c, t == NTAPS - 1 ? 0 : t + 1);
                if (newA && !(a.dbg & 4)) load_A(c + 1);
                if constexpr (NBUFB == 1) {
                    compute(t, 0, (KSKIP && c == a.nchunk - 1) ? a.nkk_last : 4);
                    __syncthreads();  // every wave is done with the weight tile (and the old halo)
                    if (newA) store_A();
                    if (more) store_B(0);
                    __syncthreads();
                    continue;
                }
                compute(t, it & 1, (KSKIP && c == a.nchunk - 1) ? a.nkk_last : 4);
                if (newA) {
                    __syncthreads();  // every wave is done with the old halo
                    store_A();
                }
#ifdef CN_ABLATE
                if (a.dbg & 8) continue;   // no weight-tile store, no per-tap barrier
#endif
                if (more) store_B((it + 1) & 1);
                __syncthreads();
            }
        }
        }
    }

    };   // main_loop

    if constexpr (HEADS) {
        if constexpr (HREG) {
            // ---- register-resident hidden layer (f32s, waves 4 x 1: wave w owns the tile's pixel row
            // w = 32 pixels and ALL hidden channels of the slice).  The 3x3 GEMM is computed as
            // D[hidden][pixel] (operands swapped in compute()), so lane (l31, h) holds, per 32-wide
            // hidden block j and register group g, the four CONSECUTIVE hidden channels
            // 32j + 8g + 4h + {0..3} of pixel l31: relu(acc * scale + bias1) is split there and fed
            // to the 1x1 GEMM as its B operand straight from registers -- no S tile in LDS, no
            // barrier between the two GEMMs, and slices of 128 hidden channels (the 128-wide main
            // loop of the trunk) instead of 64.  K order of a 16-deep step s of block j: lane half h
            // contributes hidden 32j + 16s + 4h + {0..3} and 32j + 16s + 8 + 4h + {0..3} -- the 1x1
            // weights are read from the row-major (cout, hidden) matrix in exactly that order (two
            // 16-byte loads per fragment, L1 / L2 resident: a head's matrix is <= 96 KB) and split
            // on the way.
            const int head = vby;
            const int cout2 = hd.cout[head];
            const int slices = hd.slices;
            const int hidden = slices * BN;
            float *y2 = hd.y[head];
            const float *w2 = hd.w[head];
            float *tab = reinterpret_cast<float *>(rowoff + BM);   // [2][BN] scale1 / bias1 of the slice
            float *tab2 = tab + 2 * BN;                              // [W2_ROWS][2] bias2 / oscale
            const int HWp = a.H * a.W;
            const int mpix = wave * 32 + l31;            // this lane's pixel (column of both D's)
            constexpr int NJB = W2_ROWS / 32;
            const int nblk = (cout2 + 31) / 32;          // <= NJB (checked by the caller)
            for (int row = tid; row < W2_ROWS; row += NT) {
                const float *b2 = hd.bias[head], *os2 = hd.oscale[head];
                tab2[2 * row] = (b2 && row < cout2) ? b2[row] : 0.f;
                tab2[2 * row + 1] = (os2 && row < cout2) ? os2[row] : 1.f;
            }
            cn_f32x16 acc2[NJB];
#pragma unroll
            for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[jb][r] = 0.f;
            // this lane's rows of the 1x1 matrix (padded rows: computed, never stored)
            const float *w2row[NJB];
#pragma unroll
            for (int jb = 0; jb < NJB; ++jb) w2row[jb] = w2 + (size_t)min(jb * 32 + l31, cout2 - 1) * hidden + 4 * lh;
            for (int sl = 0; sl < slices; ++sl) {
                n0 = (head * slices + sl) * BN;
                set_weight_rows();
#pragma unroll
                for (int i = 0; i < MB; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
                if (sl) __syncthreads();              // every wave has read the previous slice's table
                for (int n = tid; n < BN; n += NT) {
                    tab[n] = a.scale ? a.scale[n0 + n] : 1.f;
                    tab[BN + n] = a.shift ? a.shift[n0 + n] : 0.f;
                }
                main_loop();                           // (its first barrier publishes the tables)
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    // 1x1 weights of hidden block j for one output block: 2 steps x 2 runs of 4
                    // (requested one output block ahead; the first under the hidden block's VALU work)
                    cn_f32x4 wq[2][2][2];
                    auto load_w2 = [&](cn_f32x4 (&dst)[2][2], int jb) {
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                            for (int pq = 0; pq < 2; ++pq)
                                dst[s2][pq] = *reinterpret_cast<const cn_f32x4 *>(
                                    w2row[jb] + sl * BN + 32 * j + 16 * s2 + 8 * pq);
                    };
                    load_w2(wq[0], 0);
                    // hidden values of block j -> (high, low) B fragments
                    c3_f16x8 shi[2], slo[2];
                    {
                        cn_f16x4v hq[4], lq[4];
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int n = 32 * j + 8 * g + 4 * lh;
                            const cn_f32x4 s1 = *reinterpret_cast<const cn_f32x4 *>(tab + n);
                            const cn_f32x4 b1 = *reinterpret_cast<const cn_f32x4 *>(tab + BN + n);
                            cn_f32x4 t = {acc[0][j][4 * g], acc[0][j][4 * g + 1], acc[0][j][4 * g + 2], acc[0][j][4 * g + 3]};
                            t = t * s1 + b1;
                            if (a.relu) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) t[e] = fmaxf(t[e], 0.f);
                            }
                            cn_rng_upd4(rng_out, t);
                            cn_split4(t, hq[g], lq[g]);
                        }
                        shi[0] = __builtin_shufflevector(hq[0], hq[1], 0, 1, 2, 3, 4, 5, 6, 7);
                        shi[1] = __builtin_shufflevector(hq[2], hq[3], 0, 1, 2, 3, 4, 5, 6, 7);
                        slo[0] = __builtin_shufflevector(lq[0], lq[1], 0, 1, 2, 3, 4, 5, 6, 7);
                        slo[1] = __builtin_shufflevector(lq[2], lq[3], 0, 1, 2, 3, 4, 5, 6, 7);
                    }
#pragma unroll
                    for (int jb = 0; jb < NJB; ++jb)
                        if (jb < nblk) {   // uniform
                            if (jb + 1 < nblk) load_w2(wq[(jb + 1) & 1], jb + 1);
#pragma unroll
                            for (int s2 = 0; s2 < 2; ++s2) {
                                cn_f16x4v wh0, wl0, wh1, wl1;
                                cn_split4(wq[jb & 1][s2][0], wh0, wl0);
                                cn_split4(wq[jb & 1][s2][1], wh1, wl1);
                                const c3_f16x8 whi = __builtin_shufflevector(wh0, wh1, 0, 1, 2, 3, 4, 5, 6, 7);
                                const c3_f16x8 wlo = __builtin_shufflevector(wl0, wl1, 0, 1, 2, 3, 4, 5, 6, 7);
                                acc2[jb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo, shi[s2], acc2[jb], 0, 0, 0);
                                acc2[jb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, slo[s2], acc2[jb], 0, 0, 0);
                                acc2[jb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, shi[s2], acc2[jb], 0, 0, 0);
                            }
                        }
                    __builtin_amdgcn_sched_barrier(0);   // hidden blocks one after the other (registers)
                }
            }
            {
                const int off = rowoff[mpix];                // (b*H + oy)*W + ox, or -1
                const int pix = off - b * HWp;
                if (off >= 0) {
#pragma unroll
                    for (int jb = 0; jb < NJB; ++jb)
                        if (jb < nblk) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int co = jb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                                if (co < cout2)
                                    y2[((size_t)b * cout2 + co) * HWp + pix] = acc2[jb][r] * tab2[2 * co + 1] + tab2[2 * co];
                            }
                        }
                }
            }
            if (a_range) {
                cn_rng_commit(a_range, 0, rng_out);
                if (a.in_plain) cn_rng_commit(a_range, 1, rng_in);
            }
            return;
        } else {
        // ---- fused heads (resnet_dcn.py:155-177, pose_dla_dcn.py:456-468, large_hourglass.py:
        // make_kp_layer): per 64-channel slice of the head's hidden layer, hidden =
        // relu(conv3x3 + bias1) stays in LDS as S[128 pixels][64] and the 1x1 convolution
        // accumulates out[cout][pixel] += W2[cout][slice] . S^T (D rows = cout, cols = pixel, so
        // that a wave's stores run along x of the NCHW map the decode consumes).  head_conv = 64
        // is one slice; 256 (dla_34, hourglass) four: the 256-channel hidden tensor never exists.
        const int head = vby;
        const int cout2 = hd.cout[head];
        // the multi-slice form keeps the 1x1 accumulators (48 registers) live across the main loop:
        // it is the PDQ = 1 instantiation (one tap of weight prefetch, 232 registers, two workgroups
        // per CU); head_conv = 64 keeps the deep-prefetch build with the short-lived accumulators
        const int slices = (PDQ == 1) ? hd.slices : 1;
        float *y2 = hd.y[head];
        float *S = reinterpret_cast<float *>(smem);      // [BM][LDS2] (main-loop tiles are dead)
        float *W2 = As + UNION - W2_ROWS * LDS2;         // [W2_ROWS][LDS2], column 64 = bias
        const int HWp = a.H * a.W;
        const int mpix = wave * 32 + l31;            // this lane's pixel (column of D)
        const float *Sa = S + mpix * LDS2 + 4 * lh;
        const float *Wb = W2 + l31 * LDS2 + 4 * lh;
        cn_f32x16 acc2[W2_ROWS / 32];
#pragma unroll
        for (int jb = 0; jb < W2_ROWS / 32; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[jb][r] = 0.f;
        for (int sl = 0; sl < slices; ++sl) {
            n0 = (head * slices + sl) * BN;
            set_weight_rows();
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            stage_W2(0, sl);
            main_loop();
            {
                const int n = wn * TN + l31;  // NB == 1 for the 64-wide tile
                const float s1 = a.scale ? a.scale[n0 + n] : 1.f;
                const float b1 = a.shift ? a.shift[n0 + n] : 0.f;
#pragma unroll
                for (int i = 0; i < MB; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                            float t = acc[i][j][r] * s1 + b1;
                            t = a.relu ? fmaxf(t, 0.f) : t;
                            if constexpr (SPLIT) {   // hidden channel nn of the row, as (high, low)
                                const int nn = n + j * 32;
                                cn_rng_upd1(rng_out, t);
                                const float c = fminf(fmaxf(t, -65504.0f), 65504.0f);
                                const _Float16 hi = (_Float16)c;
                                char *g = reinterpret_cast<char *>(S + row * LDS2) + (nn >> 5) * 128 + (nn & 31) * 2;
                                *reinterpret_cast<_Float16 *>(g) = hi;
                                *reinterpret_cast<_Float16 *>(g + 64) = (_Float16)(c - (float)hi);
                            } else {
                                S[row * LDS2 + n + j * 32] = t;
                            }
                        }
            }
            // heads wider than W2_ROWS (only with one slice): one group of 1x1 rows after the other
            for (int g0 = 0; g0 < cout2; g0 += W2_ROWS) {
                const int rows = min(W2_ROWS, (cout2 - g0 + 31) / 32 * 32);
                if (g0) {
                    __syncthreads();  // previous group's W2 fully read
                    stage_W2(g0, sl);
#pragma unroll
                    for (int jb = 0; jb < W2_ROWS / 32; ++jb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc2[jb][r] = 0.f;
                }
                __syncthreads();  // S (and a restaged W2) visible
                const int nblk = rows / 32;
                if constexpr (SPLIT) {
#pragma unroll
                    for (int g = 0; g < HEAD_CONV / 32; ++g) {
                        c3_f16x8 sf[4];
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk)
                            sf[kk] = *reinterpret_cast<const c3_f16x8 *>(Sa + g * 32 + kk * 8);
#pragma unroll
                        for (int jb = 0; jb < W2_ROWS / 32; ++jb) {
                            if (jb < nblk) {  // uniform
                                c3_f16x8 wf[4];
#pragma unroll
                                for (int kk = 0; kk < 4; ++kk)
                                    wf[kk] = *reinterpret_cast<const c3_f16x8 *>(Wb + jb * 32 * LDS2 + g * 32 + kk * 8);
                                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                for (int term = 0; term < 3; ++term)
#pragma unroll
                                    for (int s2 = 0; s2 < 2; ++s2) {
                                        const int kw = (term == 0) ? 2 + s2 : s2;   // weight part
                                        const int ks = (term == 1) ? 2 + s2 : s2;   // activation part
                                        acc2[jb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[kw], sf[ks],
                                                                                          acc2[jb], 0, 0, 0);
                                    }
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    }
                } else
#pragma unroll
                for (int kk = 0; kk < HEAD_CONV / 8; ++kk) {
                    const cn_f32x4 af = *reinterpret_cast<const cn_f32x4 *>(Sa + kk * 8);
#pragma unroll
                    for (int jb = 0; jb < W2_ROWS / 32; ++jb) {
                        if (jb < nblk) {  // uniform
                            const cn_f32x4 bf =
                                *reinterpret_cast<const cn_f32x4 *>(Wb + jb * 32 * LDS2 + kk * 8);
#pragma unroll
                            for (int s = 0; s < 4; ++s)
                                acc2[jb] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[s], af[s], acc2[jb],
                                                                                0, 0, 0);
                        }
                    }
                }
                if (sl == slices - 1) {
                    const int off = rowoff[mpix];                // (b*H + oy)*W + ox, or -1
                    const int pix = off - b * HWp;
                    if (off >= 0) {
#pragma unroll
                        for (int jb = 0; jb < W2_ROWS / 32; ++jb) {
                            if (jb < nblk) {
#pragma unroll
                                for (int r = 0; r < 16; ++r) {
                                    const int rr = jb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                                    const int co = g0 + rr;
                                    if (co < cout2)
                                        y2[((size_t)b * cout2 + co) * HWp + pix] =
                                            acc2[jb][r] * W2[rr * LDS2 + HEAD_CONV + 1] + W2[rr * LDS2 + HEAD_CONV];
                                }
                            }
                        }
                    }
                }
            }
            __syncthreads();   // S and W2 fully read before the next slice restages them
        }
        if constexpr (SPLIT) {
            if (a_range) {
                cn_rng_commit(a_range, 0, rng_out);
                if (a.in_plain) cn_rng_commit(a_range, 1, rng_in);
            }
        }
        return;
        }   // !HREG
    } else {
        main_loop();
    }

    if (a.dbg & 1) {  // ablation (cn_set_tuning key 9): no epilogue, one store keeps acc live
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) t += acc[i][j][0] + acc[i][j][7] + acc[i][j][15];
        if (t == 12345.678f) reinterpret_cast<float *>(a.y)[0] = t;
        return;
    }
    // ---- epilogue: the accumulator tile goes through LDS so that residual loads and output
    // stores are 16-byte accesses along Cout.  The whole tile is staged at once when it fits
    // the kernel's LDS (every shape but the 256-pixel tiles): one barrier, and the residual
    // loads are issued BEFORE the staging so their latency hides behind it (ablation: the
    // per-wave-row form cost 8-14 % of the 64- and 512-channel layers).
    constexpr int LDC = BN + 4;
    constexpr int EP = c3_epi_rows<BN, WM, BM>();  // rows staged per pass: BM or one wave row
    constexpr int NPASS = BM / EP;
    float *Cs = reinterpret_cast<float *>(smem);
    constexpr int C4 = BN / 4;
    constexpr int RPI = NT / C4;
    constexpr int ITERS = (EP + RPI - 1) / RPI;
    const int c4 = tid % C4, r0 = tid / C4;
    const int n = n0 + c4 * 4;
    float sc[4], sf[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const bool ok = (n + e) < a.Cout;
        sc[e] = (a.scale && ok) ? a.scale[n + e] : 1.f;
        sf[e] = (a.shift && ok) ? a.shift[n + e] : 0.f;
    }
    const bool vec = a.vec_out && (n + 4 <= a.Cout);
#pragma unroll 1
    for (int pass = 0; pass < NPASS; ++pass) {
        const int rbase = pass * EP;
        cn_f32x4 res[ITERS];
        int offs[ITERS];
        if (vec) {
#pragma unroll
            for (int k = 0; k < ITERS; ++k) {
                const int lr = k * RPI + r0;
                offs[k] = (lr < EP) ? rowoff[rbase + lr] : -1;
                if (a.residual) {
                    if constexpr (SPLIT)
                        res[k] = a.res_plain
                                     ? c3_load4(reinterpret_cast<const float *>(a.residual) +
                                                (size_t)(offs[k] >= 0 ? offs[k] : 0) * a.res_pitch + n)
                                     : cn_load4_f32s(a.residual, (size_t)(offs[k] >= 0 ? offs[k] : 0),
                                                     a.res_pitch, n);
                    else
                        res[k] = c3_load4(reinterpret_cast<const T *>(a.residual) +
                                          (size_t)(offs[k] >= 0 ? offs[k] : 0) * a.out_pitch + n);
                }
            }
        }
        if (pass) __syncthreads();  // previous pass fully read
        if (wm * TM >= rbase && wm * TM < rbase + EP) {
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = wm * TM - rbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        Cs[row * LDC + wn * TN + j * 32 + l31] = acc[i][j][r];
                    }
        }
        __syncthreads();
        if (vec) {
#pragma unroll
            for (int k = 0; k < ITERS; ++k) {
                if (offs[k] < 0) continue;
                cn_f32x4 v = *reinterpret_cast<const cn_f32x4 *>(Cs + (k * RPI + r0) * LDC + c4 * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = v[e] * sc[e] + sf[e];
                    if (a.residual) {
                        if constexpr (SPLIT) t = fmaf(res[k][e], a_res_mul, t);
                        else t += res[k][e];
                    }
                    v[e] = a.relu ? fmaxf(t, 0.f) : t;
                }
                if constexpr (SPLIT) {
                    if (a.out_plain)
                        c3_store4(reinterpret_cast<float *>(a.y) + (size_t)offs[k] * a.out_pitch + n, v);
                    else {
                        cn_rng_upd4(rng_out, v);
                        cn_store4_f32s(a.y, (size_t)offs[k], a.out_pitch, n, v);
                    }
                } else {
                    c3_store4(reinterpret_cast<T *>(a.y) + (size_t)offs[k] * a.out_pitch + n, v);
                }
            }
        } else if (n < a.Cout) {
            for (int k = 0; k < ITERS; ++k) {
                const int lr = k * RPI + r0;
                if (lr >= EP) continue;
                const int off = rowoff[rbase + lr];
                if (off < 0) continue;
                for (int e = 0; e < 4 && (n + e) < a.Cout; ++e) {
                    const size_t o = (size_t)off * a.out_pitch + n + e;
                    float t = Cs[lr * LDC + c4 * 4 + e] * sc[e] + sf[e];
                    if constexpr (SPLIT) {
                        if (a.residual)
                            t = fmaf(a.res_plain ? reinterpret_cast<const float *>(a.residual)[(size_t)off * a.res_pitch + n + e]
                                                 : cn_load1_f32s(a.residual, (size_t)off, a.res_pitch, n + e),
                                     a_res_mul, t);
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
    if constexpr (SPLIT) {
        if (a_range) {   // uniform; every lane of the workgroup reaches this point
            if (!a.out_plain) cn_rng_commit(a_range, 0, rng_out);
            if (a.in_plain) cn_rng_commit(a_range, 1, rng_in);
        }
    }
}

template <typename T, int TW, int BN, int WM, int WN, bool HEADS = false, int BM = 128,
          bool KSKIP = false, bool DECONV = false, int NBUFB = 2, int PDQ = CN_F32S_PREFETCH_TAPS>
__global__ __launch_bounds__(WM * WN * 64, (std::is_same<T, cn_f32s>::value && BM == 256 && WM * WN == 8) ? 4
                                           : (std::is_same<T, cn_f32s>::value && BN == 128 && WM * WN == 4 && NBUFB == 2 && PDQ == 2) ? 2
                                           : (std::is_same<T, cn_f32s>::value && BN == 64 && BM == 128 && WM * WN == 4 && !HEADS && !DECONV && NBUFB == 2 && PDQ == 2) ? 3
                                           : (HEADS && PDQ == 1) ? 2 : 1)
void conv3x3s1_kernel(const C3Args a, const C3Heads hd)
{
    conv3x3s1_body<T, TW, BN, WM, WN, HEADS, BM, KSKIP, DECONV, NBUFB, PDQ>(a, hd);
}

// 64-wide tiles at FOUR workgroups per CU (<= 128 registers, single-buffered weight tile ->
// 39 KB of LDS): the 64-channel layers at 128^2 have 4096 tiles at B = 32, i.e. 5.33 dispatch
// rounds at three workgroups per CU (the last one mostly empty) but exactly 4 rounds at four.
template <typename T, int TW, bool DECONV = false>
__global__ __launch_bounds__(256, 4) void conv3x3s1_occ4_kernel(const C3Args a, const C3Heads hd)
{
    conv3x3s1_body<T, TW, 64, 2, 2, false, 128, false, DECONV, 1>(a, hd);
}

// does a launch of `wgs` 64-wide tiles need fewer work-normalised dispatch rounds at four
// workgroups per CU (1024 per round) than at three (768)?
inline bool c3_occ4_pays(long wgs)
{
    return wgs > 768 && cn_cdiv((int)wgs, 1024) * 4 < cn_cdiv((int)wgs, 768) * 3;
}

template <typename T, int TW, bool DECONV = false>
int launch_c3_occ4(const C3Args &a, hipStream_t st)
{
    constexpr int TH = 128 / TW;
    constexpr size_t lds = c3_union_floats<TW, 64, 2, false, 128, 1>() * 4 + 128 * 4;
    CN_SET_MAX_LDS_ONCE((conv3x3s1_occ4_kernel<T, TW, DECONV>), lds);
    C3Args b = a;
    b.tiles_x = cn_cdiv(a.W, TW);
    b.tiles_y = cn_cdiv(a.H, TH);
    b.stagger = 0;
    dim3 grid((unsigned)(a.B * b.tiles_x * b.tiles_y), cn_cdiv(a.Cout, 64), DECONV ? 4 : 1);
    const C3Heads none = {};
    hipLaunchKernelGGL((conv3x3s1_occ4_kernel<T, TW, DECONV>), grid, dim3(256), lds, st, b, none);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

template <typename T, int TW, int BN, int WM, int WN, bool HEADS = false, int BM = 128,
          bool KSKIP = false, bool DECONV = false, int NBUFB = 2, int PDQ = CN_F32S_PREFETCH_TAPS>
int launch_c3(const C3Args &a, hipStream_t st, const C3Heads *hd = nullptr)
{
    constexpr int TH = BM / TW;
    constexpr size_t lds = c3_union_floats<TW, BN, WM, HEADS, BM, NBUFB>() * 4 + BM * 4 +
                           c3_table_floats<BN, WM, HEADS>() * 4;
    CN_SET_MAX_LDS_ONCE((conv3x3s1_kernel<T, TW, BN, WM, WN, HEADS, BM, KSKIP, DECONV, NBUFB, PDQ>), lds);
    C3Args b = a;
    b.tiles_x = cn_cdiv(a.W, TW);
    b.tiles_y = cn_cdiv(a.H, TH);
    // fused heads: one workgroup row per HEAD (its hidden slices run inside the workgroup)
    const int ny = (HEADS && hd) ? cn_cdiv(a.Cout, BN * (hd->slices > 0 ? hd->slices : 1)) : cn_cdiv(a.Cout, BN);
    dim3 grid((unsigned)(a.B * b.tiles_x * b.tiles_y), ny, DECONV ? 4 : 1);
    b.heads_remap = 0;
    if (!DECONV && ny > 1 && grid.x % 8 == 0 && (HEADS ? (cn_tune_heads_remap & 1) : (cn_tune_heads_remap & 2))) {
        b.heads_remap = ny;
        grid = dim3(grid.x * ny, 1, 1);
    }
    {
        // resident workgroups per CU of this variant (registers / LDS), MFMA cycles one tile
        // needs per SIMD, and the number of dispatch rounds of this launch
        constexpr int slots = (BN >= 128 || HEADS || BM > 128) ? 2 : (BN == 64 ? 3 : 4);
        // MFMA cycles per SIMD of one (tap, chunk): fp32 16 x 64-cycle, f32s 6 x 32-cycle, fp16
        // 4 x 32-cycle (64 channels) instructions per 32 x 32 block, four blocks' worth per SIMD
        constexpr long cyc_iter = std::is_same<T, float>::value ? (long)BM * BN / 4
                                  : std::is_same<T, cn_f32s>::value ? (long)BM * BN * 3 / 64
                                                                    : (long)BM * BN / 32;
        const long total = (long)grid.x * grid.y * grid.z;
        const long rounds = total / (256L * slots);
        const long tile_cycles = (long)b.nchunk * (DECONV ? 4 : 9) * cyc_iter;
        if (cn_tune_stagger_pct > 0 && rounds >= 4) {
            b.stagger_slots = slots;
            b.stagger = (int)(tile_cycles * cn_tune_stagger_pct / 100);
        } else {
            b.stagger = 0;
        }
    }
    const C3Heads none = {};
    hipLaunchKernelGGL((conv3x3s1_kernel<T, TW, BN, WM, WN, HEADS, BM, KSKIP, DECONV, NBUFB, PDQ>), grid,
                       dim3(WM * WN * 64), lds,
                       st, b,
                       hd ? *hd : none);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

}  // namespace

// f32s, 128-wide tiles: register-streamed weights (NBUFB = 0) when cn_set_tuning key 20 = 0.
// Default 1 (per-tap LDS weight tile): the streamed form wins a back-to-back micro-benchmark by
// 5-20 % (tools/bench_f32s.py) but not inside the network (resdcn_18 B=32: 4.425 vs 4.408 ms per
// step, two runs each on one box), where its 1-wave-per-SIMD occupancy hides less of the
// neighbouring launches' tails.
int cn_tune_f32s_lds_weights = 1;

// Measured (tools/bench_f32s.py, B = 32): 128-wide tiles as four waves of 64 x 64 with
// register-streamed weights beat the LDS-weight form by 5-20 % (128->128@64^2 0.148 -> 0.141 ms,
// 256->256@32^2 0.144 -> 0.126, 512->512@16^2 0.142 -> 0.115); eight waves of 32 x 64 lose
// (0.183 / 0.173 / 0.126), and for 64- / 32-wide tiles the LDS form is as fast or faster
// (64->64@128^2 0.175 vs 0.173, 128->27@64^2 0.068 vs 0.057), so those keep it.
static int c3_dispatch_f32s_wreg(C3Args &a, hipStream_t st)
{
    const bool wide = a.W >= 32;
    return wide ? launch_c3<cn_f32s, 32, 128, 2, 2, false, 128, false, false, 0>(a, st)
                : launch_c3<cn_f32s, 16, 128, 2, 2, false, 128, false, false, 0>(a, st);
}

template <typename T>
static int c3_dispatch(C3Args &a, int bn_class, hipStream_t st)
{
    // 4 x 32 tiles keep an MFMA block on one halo row; key 21 bit 2 (A/B): 8 x 16 tiles everywhere
    const bool wide = a.W >= 32 && !(std::is_same<T, cn_f32s>::value && (cn_tune_f32s_policy & 4));
    // fp32 layers whose last 32-channel chunk is less than 3/4 full skip its empty K groups
    const bool kskip = std::is_same<T, float>::value && a.nkk_last < 4;
    if (bn_class == 2) {
        // fewer than two 128-wide workgroups per CU: nothing overlaps a workgroup's halo / weight
        // staging and epilogue (ablation: 113 -> 133 TFLOP/s without them at 512->512@16^2), so
        // take 64-wide tiles and get twice the workgroups
        const long wgs = (long)a.B * cn_cdiv(a.H, wide ? 4 : 8) * cn_cdiv(a.W, wide ? 32 : 16) *
                         cn_cdiv(a.Cout, 128);
        // (threshold measured: 512 -> resdcn_18 B=8 2.77 -> 2.57 ms, B=32 7.96 -> 7.92; 768 / 1024 lose)
        if (wgs < 512 && (a.Cout % 64) == 0) bn_class = 1;
    }
    if constexpr (std::is_same<T, cn_f32s>::value) {
        if (bn_class == 2 && !cn_tune_f32s_lds_weights) return c3_dispatch_f32s_wreg(a, st);
    }
    if (bn_class == 2) {
        // 8 waves per 128 x 128 tile (wave tile 32 x 64): 4 waves/SIMD at 2 workgroups per CU
        // hide the barrier / LDS latency better than 4-wave workgroups: +1 % on resdcn_18 and
        // dla_34 (tools/bench_knob.py 15).  The same change on 64-wide tiles measured no gain.
        if constexpr (std::is_same<T, cn_f32s>::value) {
            // f32s: four waves of 64 x 64 (2/3 of the fragment reads per MFMA) with the weight
            // tiles two taps ahead -- 220 registers, two workgroups per CU.  Measured against
            // eight waves of 32 x 64 three taps ahead (cn_set_tuning key 21 bit 0): 128->128@64^2
            // 0.122 -> 0.115 ms, 256->256@32^2 0.117 -> 0.106, resdcn_18 B=32 +1.1 %
            if (!(cn_tune_f32s_policy & 1))
                return wide ? launch_c3<T, 32, 128, 2, 2, false, 128, false, false, 2, 2>(a, st)
                            : launch_c3<T, 16, 128, 2, 2, false, 128, false, false, 2, 2>(a, st);
            return wide ? launch_c3<T, 32, 128, 4, 2>(a, st) : launch_c3<T, 16, 128, 4, 2>(a, st);
        }
        if (a.waves8 & 1)
            return wide ? launch_c3<T, 32, 128, 4, 2>(a, st) : launch_c3<T, 16, 128, 4, 2>(a, st);
        return wide ? launch_c3<T, 32, 128, 2, 2>(a, st) : launch_c3<T, 16, 128, 2, 2>(a, st);
    }
    if (bn_class == 1) {
        // 64-wide N tiles: 256-pixel (8 x 32) tiles double the MFMA work per weight tile and
        // per barrier.  Measured on MI355X (tools/bench_knob.py 14): no gain (resdcn_18 8.21 vs
        // 8.23 ms, dla_34 24.36 vs 24.39 ms per 32 images), so opt-in only (cn_set_tuning 14).
        const long wgs256 = (long)a.B * cn_cdiv(a.H, 8) * cn_cdiv(a.W, 32) * cn_cdiv(a.Cout, 64);
        if (wide && a.H >= 8 && a.bm256 == 1 && wgs256 >= 1024)
            return launch_c3<T, 32, 64, 4, 1, false, 256>(a, st);
        if constexpr (std::is_same<T, cn_f32s>::value) {
            // 256-pixel tiles as EIGHT waves of 64 x 32 (two workgroups = 16 waves per CU)
            if (wide && a.H >= 8 && a.bm256 == 2 && wgs256 >= 1024)
                return launch_c3<T, 32, 64, 4, 2, false, 256, false, false, 2, 1>(a, st);
            if (wide && a.H >= 8 && a.bm256 == 3 && wgs256 >= 1024)
                return launch_c3<T, 32, 64, 4, 2, false, 256, false, false, 2, 3>(a, st);
        }
        if (kskip)
            return wide ? launch_c3<float, 32, 64, 2, 2, false, 128, true>(a, st)
                        : launch_c3<float, 16, 64, 2, 2, false, 128, true>(a, st);
        if (std::is_same<T, float>::value && a.occ4 != 2) {
            // four workgroups per CU when that needs fewer (work-normalised) dispatch rounds than
            // three: 4096 tiles = 5.33 rounds of 768 but exactly 4 of 1024 (resdcn_18 -1.0 %,
            // tools/bench_knob.py 19); cn_set_tuning key 19: 0 = this rule, 1 = always, 2 = never
            const long wgs = (long)a.B * cn_cdiv(a.H, wide ? 4 : 8) * cn_cdiv(a.W, wide ? 32 : 16) *
                             cn_cdiv(a.Cout, 64);
            if (a.occ4 == 1 || c3_occ4_pays(wgs))
                return wide ? launch_c3_occ4<float, 32>(a, st) : launch_c3_occ4<float, 16>(a, st);
        }
        if constexpr (std::is_same<T, cn_f32s>::value) {
            // deep weight prefetch costs the 64-wide tile one of its three workgroups per CU
            // (174 vs 132 registers): it pays on long K loops (512->512@16^2: 0.135 -> 0.117 ms)
            // and loses on the two-chunk 64->64 layers (0.149 -> 0.158), measured at B = 32
            if (cn_tune_f32s_policy & 2)   // experiment: two taps ahead for every 64-wide layer
                return wide ? launch_c3<T, 32, 64, 2, 2, false, 128, false, false, 2, 2>(a, st)
                            : launch_c3<T, 16, 64, 2, 2, false, 128, false, false, 2, 2>(a, st);
            if (a.nchunk < 4)
                return wide ? launch_c3<T, 32, 64, 2, 2, false, 128, false, false, 2, 1>(a, st)
                            : launch_c3<T, 16, 64, 2, 2, false, 128, false, false, 2, 1>(a, st);
        }
        return wide ? launch_c3<T, 32, 64, 2, 2>(a, st) : launch_c3<T, 16, 64, 2, 2>(a, st);
    }
    if (kskip)
        return wide ? launch_c3<float, 32, 32, 4, 1, false, 128, true>(a, st)
                    : launch_c3<float, 16, 32, 4, 1, false, 128, true>(a, st);
    // (32-wide N tiles -- the 27-channel offset convolutions -- on 256-pixel tiles: +0.3 %, removed in round 5;
    // profiles/r04_offset_conv_tile_ab.txt)
    return wide ? launch_c3<T, 32, 32, 4, 1>(a, st) : launch_c3<T, 16, 32, 4, 1>(a, st);
}

static void c3_set_ctl(C3Args &a, const cn_f32s_ctl *ctl)
{
    a.x_mul = (ctl && ctl->x_mul != 0.f) ? ctl->x_mul : 1.f;
    a.res_mul = (ctl && ctl->res_mul != 0.f) ? ctl->res_mul : 1.f;
    a.range = ctl ? ctl->range : nullptr;
}

// bn_class: 2 = 128-wide N tiles, 1 = 64, 0 = 32 (chosen by the caller, same rule as cn_conv.hip)
// f16: 0 = fp32 tensors, 1 = fp16 tensors (fp32 accumulate, scale/shift fp32)
int cn_conv3x3s1(const void *x, const void *w_packed, const float *scale, const float *shift,
                 const void *residual, void *y, int B, int H, int W, int Cin, int Cout,
                 int in_pitch, int out_pitch, int res_pitch, int relu, int vec_out, int setprio, int bn_class,
                 int f16, const cn_f32s_ctl *ctl, hipStream_t st)
{
    if (res_pitch <= 0) res_pitch = out_pitch;
    C3Args a = {};
    c3_set_ctl(a, ctl);
    a.bm256 = ((setprio >> 1) & 1) | ((setprio >> 2) & 2);  // bits 1 and 3 of the knob word: cn_set_tuning key 14
    a.waves8 = (setprio >> 2) & 1; // bit 2: cn_set_tuning key 15
    a.dbg = ((setprio >> 4) & 7) | (((setprio >> 9) & 3) << 3);  // bits 4-6, 9-10: ablation switches (cn_set_tuning key 9)
    a.occ4 = (setprio >> 7) & 3;   // bits 7-8: cn_set_tuning key 19
    setprio &= 1;
    a.x = x; a.w = w_packed; a.scale = scale; a.shift = shift; a.residual = residual; a.y = y;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.in_pitch = in_pitch;
    a.out_pitch = out_pitch; a.relu = relu; a.vec_out = vec_out; a.setprio = setprio;
    a.res_pitch = res_pitch;
    // f16: the dtype code CN_DTYPE_F32 / F16 / F32S in the low byte, cn_conv_desc.flags above it
    a.in_plain = (f16 >> 8) & CN_CONV_X_PLAIN ? 1 : 0;
    a.out_plain = (f16 >> 8) & CN_CONV_Y_PLAIN ? 1 : 0;
    a.res_plain = (f16 >> 8) & CN_CONV_R_PLAIN ? 1 : 0;
    f16 &= 255;
    const int bke = f16 == CN_DTYPE_F16 ? 64 : 32;
    a.cin_pad = (Cin + bke - 1) / bke * bke;
    a.cout_pad = (Cout + 31) / 32 * 32;
    a.nchunk = a.cin_pad / bke;
    a.nkk_last = f16 != CN_DTYPE_F32 ? 4 : ((Cin - (a.nchunk - 1) * 32) + 7) / 8;
    a.ncb = a.cout_pad / 32;
    a.wfrag_off = (size_t)9 * a.cout_pad * a.cin_pad * 4;   // behind the row-ordered copy
    // f32s tensors on both sides: the persistent loader / consumer kernel (cn_conv3x3p.hip)
    if (f16 == CN_DTYPE_F32S && !a.dbg && vec_out && scale &&
        cn_conv3x3p_takes(B, H, W, Cin, Cout, in_pitch, out_pitch, res_pitch, a.in_plain != 0, residual != nullptr))
        return cn_conv3x3s1_persist(x, w_packed, scale, shift, residual, y, B, H, W, Cin, Cout, in_pitch,
                                    out_pitch, res_pitch, relu, a.out_plain, a.res_plain, ctl, st);
    if (f16 == CN_DTYPE_F32S) return c3_dispatch<cn_f32s>(a, bn_class, st);
    return f16 == CN_DTYPE_F16 ? c3_dispatch<_Float16>(a, bn_class, st) : c3_dispatch<float>(a, bn_class, st);
}

// Fused CenterNet heads: for every head h, y_h = conv1x1(relu(conv3x3(x) + bias1_h)) + bias2_h,
// the 3x3 convolutions of all heads packed as one (n_heads*64, Cin, 3, 3) weight.
// Replaces the per-head nn.Sequential of resnet_dcn.py:155-177 / msra_resnet.py (head_conv = 64).
extern "C" int cn_heads3x3_1x1_f32(const float *x, int B, int H, int W, int Cin, int in_pitch,
                                   const float *w1_packed, const float *bias1, int head_conv,
                                   int n_heads, const cn_head_out *heads, void *stream)
{
    return cn_heads3x3_1x1(x, B, H, W, Cin, in_pitch, w1_packed, nullptr, bias1, head_conv, n_heads,
                           heads, CN_DTYPE_F32, 0, nullptr, stream);
}

// dtype-generic form: CN_DTYPE_F32S takes f32s activations (or plain ones with CN_CONV_X_PLAIN)
// and an f32s-packed 3x3 weight with its per-row prescale factors in `scale1`; the hidden tile
// and the 1x1 weights are split inside the kernel, outputs stay fp32 NCHW.
extern "C" int cn_heads3x3_1x1(const void *x, int B, int H, int W, int Cin, int in_pitch,
                               const void *w1_packed, const float *scale1, const float *bias1,
                               int head_conv, int n_heads, const cn_head_out *heads, int dtype,
                               int flags, const cn_f32s_ctl *ctl, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (!x || !w1_packed || !heads) return CN_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || n_heads <= 0) return CN_ERR_SHAPE;
    // hidden width: 64-channel slices handled one after the other inside the workgroup
    if (head_conv <= 0 || head_conv % HEAD_CONV || head_conv > 4 * HEAD_CONV || n_heads > MAX_HEADS)
        return CN_ERR_UNSUPPORTED;
    const bool f32s = (dtype == CN_DTYPE_F32S);
    if (dtype != CN_DTYPE_F32 && !f32s) return CN_ERR_UNSUPPORTED;
    if ((in_pitch & 3) || !cn_aligned16(x) || !cn_aligned16(w1_packed)) return CN_ERR_ALIGN;
    if (f32s && !(flags & CN_CONV_X_PLAIN) && (in_pitch & 31)) return CN_ERR_UNSUPPORTED;
    if (f32s && !(flags & CN_CONV_X_PLAIN) && (((uintptr_t)x) & 127u)) return CN_ERR_ALIGN;  // 128-byte groups
    C3Heads hd = {};
    for (int h = 0; h < n_heads; ++h) {
        if (!heads[h].w || !heads[h].y) return CN_ERR_NULL;
        if (heads[h].cout <= 0) return CN_ERR_SHAPE;
        if (!cn_aligned16(heads[h].w)) return CN_ERR_ALIGN;
        hd.w[h] = heads[h].w; hd.bias[h] = heads[h].bias; hd.y[h] = heads[h].y;
        hd.cout[h] = heads[h].cout;
        hd.oscale[h] = heads[h].oscale;
        // several slices accumulate into ONE register tile of 1x1 outputs
        if (head_conv > HEAD_CONV && heads[h].cout > W2_ROWS) return CN_ERR_UNSUPPORTED;
    }
    hd.slices = head_conv / HEAD_CONV;
    // f32s feature map, one 64-wide hidden layer per head: the persistent loader / consumer kernel
    // (cn_conv3x3p.hip, hidden layer in registers)
    if (f32s && cn_heads3x3p_takes(B, H, W, in_pitch, head_conv, n_heads, heads, (flags & CN_CONV_X_PLAIN) != 0))
        return cn_heads3x3p(x, B, H, W, Cin, in_pitch, w1_packed, scale1, bias1, n_heads, heads, ctl, st);
    C3Args a = {};
    c3_set_ctl(a, ctl);
    a.x = x; a.w = w1_packed; a.scale = scale1; a.shift = bias1; a.residual = nullptr; a.y = nullptr;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = n_heads * head_conv; a.in_pitch = in_pitch;
    a.out_pitch = 0; a.relu = 1; a.vec_out = 0; a.setprio = 1;
    a.in_plain = (flags & CN_CONV_X_PLAIN) ? 1 : 0;
    a.cin_pad = (Cin + 31) / 32 * 32;
    a.cout_pad = a.Cout;
    a.nchunk = a.cin_pad / 32;
    // f32s, maps of whole 32-pixel rows, every head <= 96 outputs: the hidden layer stays in registers
    // (4 x 1 waves; 128-wide slices when the hidden width allows); cn_set_tuning key 26 = 0: LDS form
    // (one 64-channel slice, i.e. res / resdcn heads: the LDS form below with its deeper weight prefetch
    // measured faster, 0.453 vs 0.492 ms; from two slices on this form wins: dla_34 1.84 -> 1.72 ms)
    if (f32s && W >= 32 && cn_tune_heads_reg && (head_conv > HEAD_CONV || cn_tune_heads_reg == 3)) {
        bool small = true;
        for (int h = 0; h < n_heads; ++h) small = small && heads[h].cout <= W2_ROWS;
        if (small) {
            if (head_conv % 128 == 0 && cn_tune_heads_reg == 2) {   // 128-wide slices: register-bound (107 spills), A/B only
                hd.slices = head_conv / 128;
                return launch_c3<cn_f32s, 32, 128, 4, 1, true, 128, false, false, 2, 1>(a, st, &hd);
            }
            return launch_c3<cn_f32s, 32, 64, 4, 1, true, 128, false, false, 2, 1>(a, st, &hd);
        }
    }
    if (hd.slices > 1) {
        if (f32s)
            return (W >= 32) ? launch_c3<cn_f32s, 32, 64, 2, 2, true, 128, false, false, 2, 1>(a, st, &hd)
                             : launch_c3<cn_f32s, 16, 64, 2, 2, true, 128, false, false, 2, 1>(a, st, &hd);
        return (W >= 32) ? launch_c3<float, 32, 64, 2, 2, true, 128, false, false, 2, 1>(a, st, &hd)
                         : launch_c3<float, 16, 64, 2, 2, true, 128, false, false, 2, 1>(a, st, &hd);
    }
    if (f32s)
        return (W >= 32) ? launch_c3<cn_f32s, 32, 64, 2, 2, true>(a, st, &hd)
                         : launch_c3<cn_f32s, 16, 64, 2, 2, true>(a, st, &hd);
    return (W >= 32) ? launch_c3<float, 32, 64, 2, 2, true>(a, st, &hd)
                     : launch_c3<float, 16, 64, 2, 2, true>(a, st, &hd);
}

// ConvTranspose2d(kernel 4, stride 2, padding 1) through the LDS-halo kernel: the four output
// parities are 2x2 convolutions over the 3x3 neighbourhood the halo already holds.
// w_packed: cn_pack_deconv4x4s2_weight_f32 layout [parity 4][tap 4][cout_pad][cin_pad].
int cn_deconv4x4s2_halo(const void *x, const void *w_packed, const float *scale, const float *shift,
                        void *y, int B, int H, int W, int Cin, int Cout, int in_pitch, int out_pitch,
                        int relu, int vec_out, int setprio, int dtype_flags, const cn_f32s_ctl *ctl,
                        hipStream_t st)
{
    C3Args a = {};
    c3_set_ctl(a, ctl);
    a.x = x; a.w = w_packed; a.scale = scale; a.shift = shift; a.residual = nullptr; a.y = y;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.in_pitch = in_pitch;
    a.out_pitch = out_pitch; a.relu = relu; a.vec_out = vec_out; a.setprio = setprio & 1;
    a.cin_pad = (Cin + 31) / 32 * 32;
    a.cout_pad = (Cout + 31) / 32 * 32;
    a.nchunk = a.cin_pad / 32;
    a.nkk_last = 4;
    a.res_pitch = out_pitch;
    a.in_plain = (dtype_flags >> 8) & CN_CONV_X_PLAIN ? 1 : 0;
    a.out_plain = (dtype_flags >> 8) & CN_CONV_Y_PLAIN ? 1 : 0;
    const bool wide = W >= 32;
    if ((dtype_flags & 255) == CN_DTYPE_F32S && vec_out && scale &&
        cn_deconv4x4s2p_takes(B, H, W, Cin, Cout, in_pitch, out_pitch, a.in_plain != 0))
        return cn_deconv4x4s2_persist(x, w_packed, scale, shift, y, B, H, W, Cin, Cout, in_pitch, out_pitch, relu,
                                      a.out_plain, ctl, st);
    if ((dtype_flags & 255) == CN_DTYPE_F32S) {
        if (Cout > 64)
            return wide ? launch_c3<cn_f32s, 32, 128, 4, 2, false, 128, false, true>(a, st)
                        : launch_c3<cn_f32s, 16, 128, 4, 2, false, 128, false, true>(a, st);
        // (64-wide: the one-tap-ahead schedule keeps three workgroups per CU, 0.092 vs 0.105 ms)
        return wide ? launch_c3<cn_f32s, 32, 64, 2, 2, false, 128, false, true, 2, 1>(a, st)
                    : launch_c3<cn_f32s, 16, 64, 2, 2, false, 128, false, true, 2, 1>(a, st);
    }
    if (Cout > 64)
        return wide ? launch_c3<float, 32, 128, 4, 2, false, 128, false, true>(a, st)
                    : launch_c3<float, 16, 128, 4, 2, false, 128, false, true>(a, st);
    const long wgs = 4L * B * cn_cdiv(H, wide ? 4 : 8) * cn_cdiv(W, wide ? 32 : 16);
    const int occ4 = (setprio >> 7) & 3;  // cn_set_tuning key 19
    if (occ4 != 2 && (occ4 == 1 || c3_occ4_pays(wgs)))
        return wide ? launch_c3_occ4<float, 32, true>(a, st) : launch_c3_occ4<float, 16, true>(a, st);
    return wide ? launch_c3<float, 32, 64, 2, 2, false, 128, false, true>(a, st)
                : launch_c3<float, 16, 64, 2, 2, false, 128, false, true>(a, st);
}
