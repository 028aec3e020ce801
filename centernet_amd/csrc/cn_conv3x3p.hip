// cn_conv3x3p.hip -- persistent, loader / consumer specialised 3x3 convolution family in f32s arithmetic
// (fp32 values as fp16 (high, low) pairs, cn_common.h), for layers whose input is an f32s tensor:
//   * 3x3 / stride 1 / pad 1 (+BN, ReLU, residual): the trunk convolutions of every backbone
//     (resnet_dcn.py:38-67 BasicBlock, pose_dla_dcn.py:31-62, large_hourglass.py:48-74);
//   * 3x3 / stride 2 / pad 1 (the first convolution of a down-sampling block) as four stride-1
//     convolutions over the input's parity planes (template S2);
//   * ConvTranspose2d(4, 2, 1) (resnet_dcn.py:228-235) as four parity 2x2 convolutions (NTAP = 4);
//   * the detection heads conv3x3 -> ReLU -> conv1x1 with a 64-wide hidden layer, all heads in one
//     launch, hidden layer in registers (resnet_dcn.py:155-177; template HEADS).
//
// Why a third 3x3 kernel.  Probes of the one-tile-per-workgroup halo kernel (cn_conv3x3.hip,
// DESIGN.md section 3.1b) show a tile's life as prologue (address set-up, first halo + weight
// round trip) -> taps -> epilogue, one after the other, with 52 staging registers and one
// ds_write pass per tile in the way of deeper prefetch.  Here
//   * a workgroup is PERSISTENT: it walks a list of (pixel tile, 64-channel output block) items,
//     and the load stream never drains at an item boundary -- the next item's halo and weight
//     tiles are in flight while the current item's epilogue runs;
//   * every global -> LDS byte moves by LDS-DMA (global_load_lds_dwordx4): no staging registers,
//     no ds_write, no VALU.  The LDS images are unpadded 128-byte rows, XOR-swizzled through the
//     per-lane SOURCE address (the destination of an LDS-DMA is lane-linear) so that the
//     ds_read_b128 fragment reads stay conflict-free;
//   * the work is split by wave role: waves 4 and 5 of the 384-thread workgroup are LOADERS (weight
//     tiles / halos: they issue all DMA pieces and own every vmcnt wait), waves 0-3 are CONSUMERS
//     (ds_read + MFMA + epilogue).
//     Consumers never have a DMA in flight, so their compiler-managed waits for residual loads /
//     output stores do not drain the pipeline, and the loader's issue stalls do not block MFMAs;
//   * two such workgroups share a CU (79.9 KB of LDS, <= 168 registers): one's epilogue and
//     barrier waits are covered by the other's matrix work.
//
// Geometry: tile = 8 rows x 16 columns of output pixels (halo 10 x 18 = 180 rows of 128 bytes per
// 32-channel chunk, double-buffered), BN = 64 output channels, four consumer waves as 2 (pixels)
// x 2 (channels): a wave owns 64 pixels x 32 channels = two 32x32 accumulators.  The MFMA
// operands are swapped (D[channel][pixel]) so that a lane ends up with four CONSECUTIVE output
// channels of its pixel: the epilogue splits them in place and transposes through a wave-private
// LDS strip into full 128-byte output rows (coalesced 16-byte stores, no cross-wave barrier).
// Weight tiles (64 rows x 128 bytes per (tap, chunk)) stream through a ring of four 8 KB slots,
// three steps ahead of their use; one s_barrier per (tap, chunk) step.
#include "cn_common.h"
#include <type_traits>

int cn_tune_c3p = 1;        // cn_set_tuning key 28: 0 = off, 1 = on for the shapes it takes
int cn_tune_c3p_stagger = 0;  // cn_set_tuning key 29: start delay of the second resident workgroup, in units of 256 cycles
                              // (64 was worth 1-2 % with the unpipelined schedule; with the pipelined one 0 is: r05_c3p_pipe.txt)
                              // (measured 0 ... 96: 48-64 is best on every trunk shape, +6 ... +13 % over none)
int cn_tune_c3p_knobs = 2;    // cn_set_tuning key 30 (A/B): see P3Args.knobs
int cn_tune_c3p_heads = 1;    // cn_set_tuning key 31: the fused heads (hidden width 64) on this kernel; 0 = halo kernel
int cn_tune_c3p_deconv = 1;   // cn_set_tuning key 32: ConvTranspose2d(4, 2, 1) in parity form on this kernel; 0 = halo kernel
int cn_tune_c3p_s2 = 1;       // cn_set_tuning key 33: 3x3 / stride 2 / pad 1 in parity-plane form on this kernel; 0 = implicit GEMM

// one 128-byte line of zeros: the DMA source of halo pixels outside the image
__device__ __attribute__((aligned(128))) unsigned char cn_p3_zero_line[128];

namespace {

typedef __attribute__((address_space(3))) void p3_lds_void;
typedef __attribute__((address_space(1))) const void p3_gl_void;
typedef _Float16 p3_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 p3_f16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t p3_u32x2 __attribute__((ext_vector_type(2)));

constexpr int P_TH = 8, P_TW = 16;            // output pixels per tile
constexpr int P_HW = P_TW + 2;                // halo columns (even: row parity = column parity)
constexpr int P_HR = (P_TH + 2) * P_HW;       // 180 halo rows
constexpr int P_HP = (P_HR + 7) / 8;          // 23 DMA pieces of 8 rows (1 KiB)
constexpr int P_HBYTES = P_HP * 1024;         // 23552 bytes per halo buffer (whole pieces)
constexpr int P_WSLOT = 64 * 128;             // one weight tile: 64 rows of 128 bytes
constexpr int P_NSLOT = 4;
constexpr int P_WOFF = 2 * P_HBYTES;          // weight ring behind the two halo buffers
constexpr int P_SSOFF = P_WOFF + P_NSLOT * P_WSLOT;       // per consumer wave: scale[32], shift[32] of its channels
constexpr int P_LDS = P_SSOFF + 6 * 256;      // 81408 bytes (heads: scale[64] shift[64] oscale[96] bias2[96])
constexpr int P_STG_ROW = 144;                // epilogue strip: 32 rows x 144 bytes per wave
constexpr int P_STG = 32 * P_STG_ROW;
static_assert(4 * P_STG <= P_HBYTES, "the epilogue strips alias one halo buffer");
static_assert(P_LDS <= 81920, "two workgroups per CU");

struct P3Args {
    const char *x;            // f32s NHWC input
    const char *w;            // f32s row-ordered weights [9][cout_pad][cin_pad], 4 bytes per element
    const float *scale, *shift;
    const char *residual;     // f32s or plain fp32 NHWC, or null
    char *y;                  // f32s or plain fp32 NHWC
    int H, W;                 // the map the tiles cover (= the output, in units of its parity classes for the transposed form)
    int Hi, Wi;               // the input map (= H, W except for stride 2)
    int in_pitchB, out_pitchB, res_pitchB;   // bytes per pixel
    int res_bytes;            // size of the residual tensor (buffer loads)
    int cin_padB;             // bytes per weight row (cin_pad * 4)
    int cout_pad;             // weight rows per tap
    int ngroups;              // 32-channel groups of the output that exist (cout_pad / 32)
    int nchunk, nblk;         // 32-channel K chunks; 64-channel output blocks
    int tiles_x, tiles_y;
    int items;                // B * tiles_y * tiles_x * npar * nblk, output block fastest, then parity
    int npar;                 // 1; 4 = ConvTranspose2d(4, stride 2, pad 1) as four parity 2x2 convolutions over the
                              // same halo: parity (py, px) = taps (t / 2 + py, t % 2 + px) of the 3x3 neighbourhood
                              // with weights [parity][tap][cout_pad][cin_pad], output pixel (2 y + py, 2 x + px)
    int relu, out_plain, res_plain;
    float res_mul;
    uint32_t *range;
    int stagger;
    int knobs;                // A/B switches (cn_set_tuning key 30): 1 = no s_setprio 1 around the consumers' MFMA block
                              // (unpipelined forms), 2 = the pipelined fragment schedule (template PIPE)
                              // (with the loaders at priority 3 the raised priority measured 3-5 % faster)
    // instrumented instantiation only (DBG = true; cn_conv3x3p_probe): ablation switches and cycle counters
    int dbg;                  // 1: no MFMAs, 2: no fragment reads (and no MFMAs), 4: no weight DMA, 8: no halo DMA,
                              // 16: no epilogue, 32: no output stores, 64: half of the fragment reads
    unsigned long long *prof; // [workgroup][wave 5][8] cycle counters, or null
};

__device__ __forceinline__ void p3_barrier()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Fused detection heads (HEADS = true, cn_heads3x3_1x1 with a 64-wide hidden layer; resnet_dcn.py:
// 155-177): output block nb of an item = head nb; its 64 hidden channels never leave the registers
// of the consumer waves and go straight into the head's 1x1 convolution.
constexpr int P_MAXH = 8;
// pipelined fragment schedule (template PIPE): MFMAs of a six-MFMA block in front of the next block's reads
#ifndef P3_SPLIT
#define P3_SPLIT 4
#endif

static_assert(P3_SPLIT >= 4, "the front part of a block must reference all six fragments (see the fence in front of the barrier)");
struct P3Heads {
    const char *wf[P_MAXH];       // 1x1 matrix as MFMA-ready (high, low) fragments (cn_pack_head_w2_f32s)
    const float *bias[P_MAXH];    // (cout) or null
    const float *oscale[P_MAXH];  // (cout) or null: y = acc * oscale + bias
    float *y[P_MAXH];             // (B, cout, H, W)
    int cout[P_MAXH];             // <= 96
};

__device__ __forceinline__ void p3_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// item -> (image, tile row, tile column, output block)
struct P3Item { int b, ty0, tx0, nb, par; };
__device__ __forceinline__ P3Item p3_decode(const P3Args &a, int item)
{
    P3Item it;
    it.nb = item % a.nblk;
    int t = item / a.nblk;
    it.par = t % a.npar;
    t /= a.npar;
    const int tx = t % a.tiles_x;
    t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    it.b = t / a.tiles_y;
    it.ty0 = ty * P_TH;
    it.tx0 = tx * P_TW;
    return it;
}

// RES: 0 = no residual, 1 = f32s residual, 2 = plain fp32 residual
// Stride 2 (S2): the 3x3 / stride-2 / pad-1 convolution as four stride-1 convolutions over the input's
// PARITY PLANES (p, q) = pixels (2 y + p, 2 x + q): plane (1, 1) carries the four corner taps, (0, 1) and
// (1, 0) two edge taps each, (0, 0) the centre tap -- every tap at offset (0 | 1, 0 | 1) of the plane's
// 10 x 18 halo around the output tile.  A chunk is four stages (one halo each) of 4, 2, 2 and 1 steps.
__device__ constexpr int p3_s2_oy(int i) { return i == 0 || i == 1 || i == 6 ? 0 : 1; }   // halo row offset of step i
__device__ constexpr int p3_s2_ox(int i) { return i == 0 || i == 2 || i == 4 ? 0 : 1; }   // halo column offset
__device__ constexpr int p3_s2_buf(int i) { return (i >= 4 && i <= 5) || i == 8 ? 1 : 0; } // halo buffer = stage parity
__device__ constexpr int p3_s2_tap(int i)     // weight matrix (ky * 3 + kx) of step i
{
    return i == 0 ? 0 : i == 1 ? 2 : i == 2 ? 6 : i == 3 ? 8 : i == 4 ? 3 : i == 5 ? 5 : i == 6 ? 1 : i == 7 ? 7 : 4;
}

// Halo loader waves: the stride-2 form stages four halos per chunk (one per parity plane) and the transposed form
// one per four steps -- 10 and 6 DMA pieces per step against 2.6 of the nine-tap form, more than one wave issues
// in a step's time (matrix pipe 30 % / 40 % busy with one loader: profiles/r05_sq_counters_v3_cfg1.txt) -- so
// those two forms run TWO halo loader waves (448 threads), each taking every other piece.
#ifndef P3_HALO2
#define P3_HALO2 1
#endif

constexpr int p3_halo_loaders(int ntap, bool s2) { return (P3_HALO2 && (s2 || ntap == 4)) ? 2 : 1; }
constexpr int p3_threads(int ntap, bool s2) { return 320 + 64 * p3_halo_loaders(ntap, s2); }

template <int RES, bool OUT_PLAIN, bool DBG = false, bool HEADS = false, int NTAP = 9, bool S2 = false, bool PIPE = false>
__global__ __launch_bounds__(p3_threads(NTAP, S2), 4) void conv3x3p_kernel(const P3Args a, const P3Heads hd)
{
    static_assert(!PIPE || !DBG, "pipelined fragment schedule: not in the instrumented instantiation");
    static_assert(!S2 || (NTAP == 9 && RES == 0 && !DBG && !HEADS), "stride 2: nine taps, no residual");
    // residual rows requested this many steps before an item's last step ends (0 .. 3 measured with
    // interleaved medians at B = 32: no difference on any trunk shape, so the shortest live range)
    constexpr int RES_AT = 0;
    static_assert(!HEADS || (RES == 0 && !DBG), "fused heads: no residual, no probes");
    static_assert(NTAP == 9 || (NTAP == 4 && RES == 0 && !DBG && !HEADS), "4 taps: the transposed convolution's parity form");
    constexpr bool DECONV = (NTAP == 4);
    extern __shared__ __attribute__((aligned(128))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // ---- this workgroup's items: XCD x = id % 8 owns a contiguous range, its workgroups take
    // every (gridDim / 8)-th item of it (neighbouring tiles and the output blocks of one tile
    // are in flight on one L2 at the same time)
    const int nx = gridDim.x >> 3;                       // workgroups per XCD
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int per = (a.items + 7) >> 3;
    const int lo = xcd * per, hi = min(lo + per, a.items);
    const int first = lo + jx;
    const int nit = first < hi ? (hi - first + nx - 1) / nx : 0;
    if (nit == 0) return;
    const int S = nit * a.nchunk;                         // stages of this workgroup

    unsigned long long pf_rt0 = 0;     // 100 MHz wall clock at entry (instrumented instantiation)
    if (DBG) pf_rt0 = __builtin_amdgcn_s_memrealtime();
    if (a.stagger && blockIdx.x >= 256) {
        const unsigned long long t0 = __builtin_readcyclecounter();
        while (__builtin_readcyclecounter() - t0 < (unsigned long long)a.stagger * 256u) __builtin_amdgcn_s_sleep(32);
    }

    if (wave >= 4) {
        // =============================== LOADERS ===============================
        // wave 4 streams the weight tiles, wave 5 the halos: each owns one in-order DMA queue (its
        // vmcnt), both join every barrier.  (One loader wave issuing all 12 pieces of a step was
        // measured busy for ~900 cycles per step: more than the step's matrix work.)
        const int prow = lane >> 3, ps = lane & 7;
        unsigned long long pf_wait = 0, pf_bar = 0, pf_t0 = 0;
        auto now = [&]() { return DBG ? (unsigned long long)__builtin_readcyclecounter() : 0ull; };
        if (DBG) pf_t0 = now();
        __builtin_amdgcn_s_setprio(3);
#define P3_IC(v) std::integral_constant<int, (v)>{}
        if (wave == 4) {
            // ---- weight tiles: piece p = rows 8p .. 8p+7 of the 64-row tile of (tap, chunk);
            // slot s of row n receives source column s ^ ((n >> 1) & 7)
            unsigned wofs[8];        // per item: byte offset of the lane's source inside one tap's matrix
            auto set_wofs = [&](int nb) {
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const int n = 8 * p + prow;
                    const int row = min(nb * 64 + n, a.cout_pad - 1);
                    wofs[p] = (unsigned)(row * a.cin_padB + ((ps ^ ((n >> 1) & 7)) << 4));
                }
            };
            set_wofs(first % a.nblk);
            const size_t tapB = (size_t)a.cout_pad * a.cin_padB;
            int wk = 0, wc = 0, wt = 0, wg = 0;
            // (transposed convolution: the item's parity selects its group of four tap matrices)
            size_t wpar = DECONV ? (size_t)((first / a.nblk) % a.npar) * NTAP * tapB : 0;
            auto issue_W = [&]() {   // weight tile of the cursor's step into ring slot wg & 3, then advance
                if (wk >= nit) return;
                const char *base = a.w + wpar + (size_t)(S2 ? p3_s2_tap(wt) : wt) * tapB + (size_t)wc * 128;
                char *dst = smem + P_WOFF + (wg & (P_NSLOT - 1)) * P_WSLOT;
                if (!DBG || !(a.dbg & 4)) {
#pragma unroll
                    for (int p = 0; p < 8; ++p)
                        __builtin_amdgcn_global_load_lds((p3_gl_void *)(base + wofs[p]), (p3_lds_void *)(dst + p * 1024), 16, 0, 0);
                }
                ++wg;
                if (++wt == NTAP) {
                    wt = 0;
                    if (++wc == a.nchunk) {
                        wc = 0;
                        ++wk;
                        if (wk < nit && a.nblk > 1) set_wofs((first + wk * nx) % a.nblk);
                        if (DECONV && wk < nit) wpar = (size_t)(((first + wk * nx) / a.nblk) % a.npar) * NTAP * tapB;
                    }
                }
            };
            // tiles of steps 0, 1, 2 up front; then tile g + 3 behind the barrier of step g (the
            // barrier says every consumer is done with tile g - 1, whose slot it takes), and before
            // that barrier tile g has landed: at most the two younger tiles (16 pieces) outstanding
            issue_W();
            issue_W();
            issue_W();
            for (int s = 0;; ++s) {
                const bool last = (s >= S - 1);   // the stream ends: drain instead of counting
#pragma unroll
                for (int t = 0; t < NTAP; ++t) {
                    const unsigned long long c0 = now();
                    if (last) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                    const unsigned long long c1 = now();
                    p3_barrier();
                    if (DBG) { pf_wait += c1 - c0; pf_bar += now() - c1; }
                    if (s == S) break;
                    issue_W();
                }
                if (s == S) break;
            }
        } else {
            // ---- halos: piece k covers halo rows 8k .. 8k+7; lane = (row in piece, 16-byte slot s);
            // slot s of row r receives source column s ^ key(r), key = (halo column >> 1) & 7
            // (two loader waves: wave 5 + h takes pieces 2 j + h -- one copy of this code per h, so that every
            // piece index stays a compile-time constant)
            auto halo_loader = [&](auto HW) {
            constexpr int NHL = p3_halo_loaders(NTAP, S2), NPW = (P_HP + NHL - 1) / NHL;
            constexpr int hw = decltype(HW)::value;
            int poff[NPW];           // byte offset of the lane's source relative to pixel (ty0, tx0) of the tile
            int pyx[NPW];            // hy | hx << 8, or -1: row beyond the halo
#pragma unroll
            for (int j = 0; j < NPW; ++j) {
                const int k = j * NHL + hw;
                const int r = 8 * k + prow;
                const int hy = r / P_HW, hx = r - hy * P_HW;
                const int col = ps ^ ((hx >> 1) & 7);
                poff[j] = (S2 ? 2 : 1) * ((hy - 1) * a.Wi + (hx - 1)) * a.in_pitchB + col * 16;
                pyx[j] = (r < P_HR) ? (hy | (hx << 8)) : -1;
            }
            const char *zero = reinterpret_cast<const char *>(cn_p3_zero_line) + ps * 16;
            int hk = 0, hc = 0, hj = 0;   // cursor: the stage whose halo goes out next (hj: parity plane, S2)
            int hp = S2 ? 1 : 0, hq = S2 ? 1 : 0;   // plane order: (1, 1), (0, 1), (1, 0), (0, 0)
            P3Item hit = p3_decode(a, first);
            const char *hbase = nullptr;   // address of pixel (ty0, tx0), chunk hc, of the cursor's item
            auto set_hbase = [&]() {
                if constexpr (S2) {
                    hp = (hj == 0 || hj == 2) ? 1 : 0;
                    hq = (hj == 0 || hj == 1) ? 1 : 0;
                    hbase = a.x + ((size_t)(hit.b * a.Hi + 2 * hit.ty0 + hp) * a.Wi + 2 * hit.tx0 + hq) * a.in_pitchB +
                            (size_t)hc * 128;
                } else {
                    hbase = a.x + ((size_t)(hit.b * a.Hi + hit.ty0) * a.Wi + hit.tx0) * a.in_pitchB + (size_t)hc * 128;
                }
            };
            set_hbase();
            auto issue_H = [&](auto K0, auto K1, int buf) {   // pieces [k0, k1) of the cursor's stage
                constexpr int k0 = decltype(K0)::value, k1 = decltype(K1)::value;
                char *dst = smem + buf * P_HBYTES;
                if (DBG && (a.dbg & 8)) return;
#pragma unroll
                for (int j = k0 / NHL; j < (k1 + NHL - 1) / NHL; ++j) {
                    const int k = j * NHL + hw;
                    if (NHL == 2 && (k < k0 || k >= k1)) continue;     // (wave-uniform)
                    const int hy = pyx[j] & 255, hx = pyx[j] >> 8;
                    const int iy = S2 ? 2 * (hit.ty0 - 1 + hy) + hp : hit.ty0 - 1 + hy;
                    const int ix = S2 ? 2 * (hit.tx0 - 1 + hx) + hq : hit.tx0 - 1 + hx;
                    const bool ok = pyx[j] >= 0 && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi;
                    const char *src = ok ? hbase + poff[j] : zero;
                    __builtin_amdgcn_global_load_lds((p3_gl_void *)src, (p3_lds_void *)(dst + k * 1024), 16, 0, 0);
                }
            };
            auto advance_H = [&]() {   // cursor to the next stage
                if (S2 && ++hj < 4) {
                    set_hbase();
                    return;
                }
                hj = 0;
                if (++hc == a.nchunk) {
                    hc = 0;
                    ++hk;
                    if (hk < nit) hit = p3_decode(a, first + hk * nx);
                }
                if (hk < nit) set_hbase();
            };
            // halo of stage 0 up front; the halo of stage s + 1 goes out during steps 1 .. 6 of stage
            // s (the buffer it takes was read last in stage s - 1 and holds the epilogue strips of
            // an item that ended there until the barrier of step 1) and is complete -- every older
            // piece of this wave's queue -- before the first barrier of stage s + 1
            issue_H(P3_IC(0), P3_IC(P_HP), 0);
            advance_H();
            if constexpr (S2) {
                // stages of 4, 2, 2, 1 steps.  The next stage's halo goes out behind the barriers of this
                // stage: from its second step on in a chunk's first stage (an item's first stage: the
                // buffer holds the previous item's epilogue strips until then), from its first step
                // otherwise (every consumer has left the stage before it: its buffer is free)
                const int ST = 4 * S;
                bool done = false;
                auto run_stage = [&](auto N, int s) {
                    constexpr int n = decltype(N)::value;
                    const bool last = (s >= ST - 1);
                    const int nbuf = (s + 1) & 1;
#pragma unroll
                    for (int t = 0; t < n; ++t) {
                        if (t == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        p3_barrier();
                        if (s == ST) { done = true; return; }
                        if (!last) {
                            if constexpr (n == 4) {
                                // (two loader waves: twelve pieces each in two slices / at once -- the halo has a
                                // step more to land than with three / two slices: stride-2 layers another -6 %)
                                if (t == 1) issue_H(P3_IC(0), P3_IC(12), nbuf);
                                if (t == 2) issue_H(P3_IC(12), P3_IC(P_HP), nbuf);
                            } else if constexpr (n == 2) {
                                if (t == 0) issue_H(P3_IC(0), P3_IC(P_HP), nbuf);
                            } else {
                                issue_H(P3_IC(0), P3_IC(P_HP), nbuf);
                            }
                        }
                    }
                };
                for (int s = 0;; ++s) {
                    const int sj = s & 3;
                    if (sj == 0) run_stage(P3_IC(4), s);
                    else if (sj == 3) run_stage(P3_IC(1), s);
                    else run_stage(P3_IC(2), s);
                    if (done) break;
                    if (s < ST - 1) advance_H();
                }
            } else
            for (int s = 0;; ++s) {
                const bool last = (s >= S - 1);
                const int nbuf = (s + 1) & 1;
#pragma unroll
                for (int t = 0; t < NTAP; ++t) {
                    const unsigned long long c0 = now();
                    if (t == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    const unsigned long long c1 = now();
                    p3_barrier();
                    if (DBG) { pf_wait += c1 - c0; pf_bar += now() - c1; }
                    if (s == S) break;
                    if (!last) {
                        if constexpr (DECONV) {     // four steps per stage: the halo goes out behind steps 1 and 2
                            if (t == 1) issue_H(P3_IC(0), P3_IC(12), nbuf);
                            if (t == 2) issue_H(P3_IC(12), P3_IC(P_HP), nbuf);
                        } else {
                            if (t == 1) issue_H(P3_IC(0), P3_IC(4), nbuf);
                            if (t == 2) issue_H(P3_IC(4), P3_IC(8), nbuf);
                            if (t == 3) issue_H(P3_IC(8), P3_IC(12), nbuf);
                            if (t == 4) issue_H(P3_IC(12), P3_IC(16), nbuf);
                            if (t == 5) issue_H(P3_IC(16), P3_IC(20), nbuf);
                            if (t == 6) issue_H(P3_IC(20), P3_IC(P_HP), nbuf);
                        }
                    }
                }
                if (s == S) break;
                if (!last) advance_H();
            }
            };
            if (p3_halo_loaders(NTAP, S2) == 1 || wave == 5) halo_loader(P3_IC(0));
            else halo_loader(P3_IC(1));
        }
#undef P3_IC
        if (DBG && a.prof && lane == 0) {
            unsigned long long *o = a.prof + ((size_t)blockIdx.x * 6 + wave) * 8;
            o[0] = now() - pf_t0; o[1] = pf_wait; o[2] = pf_bar; o[3] = (unsigned long long)S;
            o[4] = pf_rt0; o[5] = __builtin_amdgcn_s_memrealtime();
        }
        return;
    }


    if constexpr (HEADS) {
        // =============================== CONSUMERS, fused heads ===============================
        // Waves 4 x 1: wave w owns pixels 32 w .. 32 w + 31 of the tile (tile rows 2 w, 2 w + 1) and
        // ALL 64 hidden channels of the item's head: acc[j] = D[hidden 32 j ..][pixel], so a lane
        // holds four CONSECUTIVE hidden channels of its pixel per register group.  The epilogue turns
        // them into relu(acc * scale + bias1), splits them and uses them directly as the B operand of
        // the head's 1x1 GEMM out[cout][pixel] (K order of a 16-deep step s of block j: lane half h
        // contributes hidden 32 j + 16 s + 4 h + {0..3} and 32 j + 16 s + 8 + 4 h + {0..3}; the 1x1
        // matrix is read from global memory -- L1 / L2 resident, <= 24 KB per head -- in that order
        // and split on the way).  No hidden tile in LDS, no barrier between the two GEMMs; lanes run
        // along x of the NCHW map the decode consumes.
        int arow, aswz, b0;
        auto lane_consts = [&]() {
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const int l31 = ln & 31, lh = ln >> 5, px = ln & 15;
            arow = ((2 * wave + (l31 >> 4)) * P_HW + px) * 128;
            aswz = 0;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) aswz |= (((lh ^ (((px + kx) >> 1) & 7)) & 7) << 4) << (8 * kx);
            b0 = l31 * 128 + (((lh ^ ((l31 >> 1) & 7)) & 7) << 4);
        };
        lane_consts();
        cn_f32x16 acc[2];
        auto zero_acc = [&]() {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        };
        zero_acc();
        float rng_out = 0.f;
        auto lds128 = [&](int off) { return *reinterpret_cast<const p3_f16x8 *>(smem + off); };
        auto step = [&](auto T, int hb, int wb) {
            constexpr int t = decltype(T)::value;
            constexpr int ky = t / 3, kx = t % 3;
            constexpr int tapoff = (ky * P_HW + kx) * 128;
            constexpr int blk1 = 32 * 128;         // weight rows 32 .. 63: hidden block 1
            p3_f16x8 xf[4], wf[4][2];
            int ax = arow + ((aswz >> (8 * kx)) & 0xff), bx = b0;
            asm volatile("" : "+v"(ax), "+v"(bx));
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
                for (int j = 0; j < 2; ++j) wf[kh][j] = lds128(wb + j * blk1 + (bx ^ (kh << 5)));
                xf[2 + kh] = lds128(hb + tapoff + (ax ^ ((kh << 5) | 64)));
                xf[kh] = lds128(hb + tapoff + (ax ^ (kh << 5)));
#pragma unroll
                for (int j = 0; j < 2; ++j) wf[2 + kh][j] = lds128(wb + j * blk1 + (bx ^ ((kh << 5) | 64)));
            }
            __builtin_amdgcn_sched_barrier(0);     // (operand hazard: every read before the first MFMA)
            if (!(a.knobs & 1)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[kh][j], xf[2 + kh], acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[2 + kh][j], xf[kh], acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[kh][j], xf[kh], acc[j], 0, 0, 0);
            }
            if (!(a.knobs & 1)) __builtin_amdgcn_s_setprio(0);
        };
        const float relu_floor = a.relu ? 0.f : -__builtin_inff();
        const int HWp = a.H * a.W;
        // 1x1 weights of one 32-row output block jb as ready-made A fragments (cn_pack_head_w2_f32s:
        // [jb][hidden block j][step s2][high | low][lane] x 16 bytes: one coalesced 1 KiB line each)
        auto load_w2 = [&](p3_f16x8 (&dst)[2][2], const char *wbase, int jb, int j, int ln) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int part = 0; part < 2; ++part)
                    dst[s2][part] = *reinterpret_cast<const p3_f16x8 *>(
                        wbase + ((((jb * 2 + j) * 2 + s2) * 2 + part) * 64 + ln) * 16);
        };
        // 1x1 fragments of the current output block, hidden block 0 / 1 (of output block 0: the first half
        // is requested two steps before the item's main loop ends, the second between the two hidden
        // blocks of the epilogue -- registers)
        p3_f16x8 wfa[2][2], wfb[2][2];
        auto epilogue = [&](const P3Item &it) {
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const int l31 = ln & 31, lh = ln >> 5;
            const int head = it.nb;
            const int cout2 = hd.cout[head];
            const char *w2 = hd.wf[head];
            float *y2 = hd.y[head];
            const int nblk2 = (cout2 + 31) >> 5;
            // hidden layer -> (high, low) B fragments [hidden block j][step s2]
            p3_f16x8 shi[2][2], slo[2][2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (j == 1) {
                    __builtin_amdgcn_sched_barrier(0);
                    load_w2(wfb, w2, 0, 1, ln);
                    __builtin_amdgcn_sched_barrier(0);
                }
                cn_f16x4v hq[4], lq[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = 32 * j + 8 * g + 4 * lh;
                    const cn_f32x4 sc = *reinterpret_cast<const cn_f32x4 *>(smem + P_SSOFF + n * 4);
                    const cn_f32x4 sh = *reinterpret_cast<const cn_f32x4 *>(smem + P_SSOFF + 256 + n * 4);
                    cn_f32x4 t;
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] = fmaxf(acc[j][4 * g + e] * sc[e] + sh[e], relu_floor);
                    cn_rng_upd4(rng_out, t);
                    cn_split4(t, hq[g], lq[g]);
                }
                shi[j][0] = __builtin_shufflevector(hq[0], hq[1], 0, 1, 2, 3, 4, 5, 6, 7);
                shi[j][1] = __builtin_shufflevector(hq[2], hq[3], 0, 1, 2, 3, 4, 5, 6, 7);
                slo[j][0] = __builtin_shufflevector(lq[0], lq[1], 0, 1, 2, 3, 4, 5, 6, 7);
                slo[j][1] = __builtin_shufflevector(lq[2], lq[3], 0, 1, 2, 3, 4, 5, 6, 7);
            }
            const int p = 32 * wave + l31;
            const int oy = it.ty0 + (p >> 4), ox = it.tx0 + (p & 15);
            const bool ok = oy < a.H && ox < a.W;
            float *ypix = y2 + (size_t)it.b * cout2 * HWp + oy * a.W + ox;
#pragma unroll 1
            for (int jb = 0; jb < nblk2; ++jb) {
                cn_f32x16 acc2;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
                __builtin_amdgcn_sched_barrier(0);     // (operand hazard: fragments landed before the first MFMA)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfa[s2][1], shi[0][s2], acc2, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfa[s2][0], slo[0][s2], acc2, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfa[s2][0], shi[0][s2], acc2, 0, 0, 0);
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfb[s2][1], shi[1][s2], acc2, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfb[s2][0], slo[1][s2], acc2, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfb[s2][0], shi[1][s2], acc2, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                // output scale / bias of the lane's rows from the LDS stash (rows beyond cout hold 1 / 0).
                // Reading the first register group closes the MFMA chain -- only then are the fragment
                // registers re-used for the next block's loads (they land under this block's stores)
                const int co0 = jb * 32 + 4 * lh;
                auto out_group = [&](int gq) {
                    const cn_f32x4 os = *reinterpret_cast<const cn_f32x4 *>(smem + P_SSOFF + 512 + (co0 + 8 * gq) * 4);
                    const cn_f32x4 bs = *reinterpret_cast<const cn_f32x4 *>(smem + P_SSOFF + 896 + (co0 + 8 * gq) * 4);
                    cn_f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc2[4 * gq + e] * os[e] + bs[e];
                    return v;
                };
                auto store_group = [&](int gq, cn_f32x4 v) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int co = co0 + 8 * gq + e;
                        if (ok && co < cout2) ypix[(size_t)co * HWp] = v[e];
                    }
                };
                const cn_f32x4 v0 = out_group(0);
                __builtin_amdgcn_sched_barrier(0);
                if (jb + 1 < nblk2) {
                    load_w2(wfa, w2, jb + 1, 0, ln);
                    load_w2(wfb, w2, jb + 1, 1, ln);
                }
                store_group(0, v0);
#pragma unroll
                for (int gq = 1; gq < 4; ++gq) store_group(gq, out_group(gq));
            }
        };

#define P3_IC(v) std::integral_constant<int, (v)>{}
        int g = 0, sidx = 0;
        // one stage = the nine taps of one 32-channel chunk.  LAST = the item's last chunk: its copy of
        // the code requests the first 1x1 fragments two steps before the end (a separate copy so that
        // their 32 registers are live over those two steps and the epilogue only)
        auto stage = [&](auto LAST, int c, const P3Item &cur) {
            constexpr bool last = decltype(LAST)::value;
            p3_barrier();
            const int hb = (sidx & 1) * P_HBYTES;
            // per item: scale (prescale factor * exponents) and bias1 of the head's 64 hidden channels and
            // the output scale / bias of its 1x1 rows, requested at the first step, parked in the
            // (shared: every wave writes the same values) LDS stash two steps later
            float sv = 1.f, hv = 0.f, ov[2] = {1.f, 1.f}, bv[2] = {0.f, 0.f};
            if (c == 0) {
                int ln = lane;
                asm volatile("" : "+v"(ln));
                const unsigned o = (unsigned)(64 * cur.nb + ln) * 4u;
                if (a.scale) sv = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(a.scale) + o);
                if (a.shift) hv = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(a.shift) + o);
                const float *os2 = hd.oscale[cur.nb], *b2 = hd.bias[cur.nb];
                const int cout2 = hd.cout[cur.nb];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int co = ln + 64 * u;     // rows that do not exist: 1 / 0
                    if (co < cout2) {
                        if (os2) ov[u] = os2[co];
                        if (b2) bv[u] = b2[co];
                    }
                }
            }
            step(P3_IC(0), hb, P_WOFF + ((g + 0) & 3) * P_WSLOT);
            p3_barrier(); step(P3_IC(1), hb, P_WOFF + ((g + 1) & 3) * P_WSLOT);
            p3_barrier(); step(P3_IC(2), hb, P_WOFF + ((g + 2) & 3) * P_WSLOT);
            if (c == 0) {
                *reinterpret_cast<float *>(smem + P_SSOFF + lane * 4) = sv;
                *reinterpret_cast<float *>(smem + P_SSOFF + 256 + lane * 4) = hv;
                *reinterpret_cast<float *>(smem + P_SSOFF + 512 + lane * 4) = ov[0];
                *reinterpret_cast<float *>(smem + P_SSOFF + 896 + lane * 4) = bv[0];
                if (lane < 32) {
                    *reinterpret_cast<float *>(smem + P_SSOFF + 512 + (64 + lane) * 4) = ov[1];
                    *reinterpret_cast<float *>(smem + P_SSOFF + 896 + (64 + lane) * 4) = bv[1];
                }
            }
            p3_barrier(); step(P3_IC(3), hb, P_WOFF + ((g + 3) & 3) * P_WSLOT);
            p3_barrier(); step(P3_IC(4), hb, P_WOFF + ((g + 4) & 3) * P_WSLOT);
            p3_barrier(); step(P3_IC(5), hb, P_WOFF + ((g + 5) & 3) * P_WSLOT);
            p3_barrier(); step(P3_IC(6), hb, P_WOFF + ((g + 6) & 3) * P_WSLOT);
            if constexpr (last) {
                int ln = lane;
                asm volatile("" : "+v"(ln));
                __builtin_amdgcn_sched_barrier(0);
                load_w2(wfa, hd.wf[cur.nb], 0, 0, ln);
                __builtin_amdgcn_sched_barrier(0);
            }
            p3_barrier(); step(P3_IC(7), hb, P_WOFF + ((g + 7) & 3) * P_WSLOT);
            p3_barrier(); step(P3_IC(8), hb, P_WOFF + ((g + 8) & 3) * P_WSLOT);
            g += 9;
            ++sidx;
        };
        if constexpr (PIPE) {
            // pipelined fragment schedule (see the plain consumers below): two sets of six fragments, the
            // next block's reads behind the fourth MFMA of the current one, the barrier of step t + 1 in
            // the middle of block (t, 1); nothing carried across an item boundary
            struct HFrag { p3_f16x8 wh[2], wl[2], xh, xl; };
            HFrag F0, F1;
            auto hload = [&](auto T, auto KH, int hb, int wb, HFrag &F) {
                constexpr int t = decltype(T)::value, kh = decltype(KH)::value;
                constexpr int ky = t / 3, kx = t % 3;
                constexpr int tapoff = (ky * P_HW + kx) * 128;
                constexpr int blk1 = 32 * 128;
                int ax = arow + ((aswz >> (8 * kx)) & 0xff), bx = b0;
                asm volatile("" : "+v"(ax), "+v"(bx));
#pragma unroll
                for (int j = 0; j < 2; ++j) F.wh[j] = lds128(wb + j * blk1 + (bx ^ (kh << 5)));
                F.xl = lds128(hb + tapoff + (ax ^ ((kh << 5) | 64)));
                F.xh = lds128(hb + tapoff + (ax ^ (kh << 5)));
#pragma unroll
                for (int j = 0; j < 2; ++j) F.wl[j] = lds128(wb + j * blk1 + (bx ^ ((kh << 5) | 64)));
            };
            auto hfront = [&](const HFrag &F) {
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.wh[j], F.xl, acc[j], 0, 0, 0);
                if constexpr (P3_SPLIT >= 4) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.wl[j], F.xh, acc[j], 0, 0, 0);
                }
                if constexpr (P3_SPLIT >= 6) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.wh[j], F.xh, acc[j], 0, 0, 0);
                }
            };
            auto hback = [&](const HFrag &F) {
                if constexpr (P3_SPLIT < 4) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.wl[j], F.xh, acc[j], 0, 0, 0);
                }
                if constexpr (P3_SPLIT < 6) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.wh[j], F.xh, acc[j], 0, 0, 0);
                }
            };
            auto hkeep = [&](const HFrag &F) {
                asm volatile("" :: "v"(F.wh[0]), "v"(F.wh[1]), "v"(F.wl[0]), "v"(F.wl[1]), "v"(F.xh), "v"(F.xl));
            };
            auto pstage = [&](auto LAST, int c, const P3Item &cur) {
                constexpr bool last = decltype(LAST)::value;
                const int hb = (sidx & 1) * P_HBYTES, hb1 = ((sidx + 1) & 1) * P_HBYTES;
                auto WB = [&](int i) { return P_WOFF + ((g + i) & 3) * P_WSLOT; };
                float sv = 1.f, hv = 0.f, ov[2] = {1.f, 1.f}, bv[2] = {0.f, 0.f};
                if (c == 0) {
                    int ln = lane;
                    asm volatile("" : "+v"(ln));
                    const unsigned o = (unsigned)(64 * cur.nb + ln) * 4u;
                    if (a.scale) sv = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(a.scale) + o);
                    if (a.shift) hv = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(a.shift) + o);
                    const float *os2 = hd.oscale[cur.nb], *b2 = hd.bias[cur.nb];
                    const int cout2 = hd.cout[cur.nb];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int co = ln + 64 * u;
                        if (co < cout2) {
                            if (os2) ov[u] = os2[co];
                            if (b2) bv[u] = b2[co];
                        }
                    }
                }
                auto pstep = [&](auto T) {
                    constexpr int t = decltype(T)::value;
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_setprio(1);
                    hfront(F0);
                    __builtin_amdgcn_sched_barrier(0);
                    hload(T, P3_IC(1), hb, WB(t), F1);
                    __builtin_amdgcn_sched_barrier(0);
                    hkeep(F0);
                    hback(F0);
                    __builtin_amdgcn_sched_barrier(0);
                    hfront(F1);
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_setprio(0);
                    p3_lds_fence();              // (every read of tile t complete before its slot is handed back)
                    p3_barrier();
                    if constexpr (t + 1 < 9) hload(P3_IC(t + 1), P3_IC(0), hb, WB(t + 1), F0);
                    else if constexpr (!last) hload(P3_IC(0), P3_IC(0), hb1, WB(9), F0);
                    __builtin_amdgcn_sched_barrier(0);
                    hkeep(F1);
                    hback(F1);
                    __builtin_amdgcn_sched_barrier(0);
                };
                pstep(P3_IC(0));
                pstep(P3_IC(1));
                pstep(P3_IC(2));
                if (c == 0) {
                    *reinterpret_cast<float *>(smem + P_SSOFF + lane * 4) = sv;
                    *reinterpret_cast<float *>(smem + P_SSOFF + 256 + lane * 4) = hv;
                    *reinterpret_cast<float *>(smem + P_SSOFF + 512 + lane * 4) = ov[0];
                    *reinterpret_cast<float *>(smem + P_SSOFF + 896 + lane * 4) = bv[0];
                    if (lane < 32) {
                        *reinterpret_cast<float *>(smem + P_SSOFF + 512 + (64 + lane) * 4) = ov[1];
                        *reinterpret_cast<float *>(smem + P_SSOFF + 896 + (64 + lane) * 4) = bv[1];
                    }
                }
                pstep(P3_IC(3));
                pstep(P3_IC(4));
                pstep(P3_IC(5));
                pstep(P3_IC(6));
                if constexpr (last) {
                    int ln = lane;
                    asm volatile("" : "+v"(ln));
                    __builtin_amdgcn_sched_barrier(0);
                    load_w2(wfa, hd.wf[cur.nb], 0, 0, ln);
                    __builtin_amdgcn_sched_barrier(0);
                }
                pstep(P3_IC(7));
                pstep(P3_IC(8));
                g += 9;
                ++sidx;
            };
            p3_barrier();                        // barrier 0 of the first stage
            for (int k = 0; k < nit; ++k) {
                const P3Item cur = p3_decode(a, first + k * nx);
                __builtin_amdgcn_sched_barrier(0);
                hload(P3_IC(0), P3_IC(0), (sidx & 1) * P_HBYTES, P_WOFF + (g & 3) * P_WSLOT, F0);
                __builtin_amdgcn_sched_barrier(0);
                for (int c = 0; c < a.nchunk - 1; ++c) pstage(std::false_type{}, c, cur);
                pstage(std::true_type{}, a.nchunk - 1, cur);
                // (behind barrier 0 of the next item: the stash is rewritten behind its barrier 3 at the
                // earliest, which no wave passes before every wave has left this epilogue)
                if (a.nchunk == 1) p3_lds_fence();
                epilogue(cur);
                zero_acc();
                lane_consts();
            }
        } else {
        for (int k = 0; k < nit; ++k) {
            const P3Item cur = p3_decode(a, first + k * nx);
            for (int c = 0; c < a.nchunk - 1; ++c) stage(std::false_type{}, c, cur);
            stage(std::true_type{}, a.nchunk - 1, cur);
            // (the stash was written in this item's first stage, two barriers ago at least; nothing the
            // epilogue reads is touched by the loaders)
            if (a.nchunk == 1) p3_lds_fence();
            epilogue(cur);
            zero_acc();
            lane_consts();
        }
        p3_barrier();
        }
#undef P3_IC
        if (a.range) cn_rng_commit(a.range, 0, rng_out);
        return;
    }

    // =============================== CONSUMERS ===============================
    const int wm = wave >> 1, wn = wave & 1;
    // Fragment addresses of the lane.  A (input pixels = the MFMA's B operand): block i = pixels
    // 64 wm + 32 i + l31 = tile rows 4 wm + 2 i + (l31 >> 4), column l31 & 15; tap (ky, kx) adds
    // (ky * 18 + kx) rows (an immediate) and changes the swizzle key with kx only: one register
    // for the row base of block 0 (block 1 = two tile rows = 2 * 18 halo rows further: an immediate),
    // one for the three swizzle terms (8 bits each).  B (weights = the MFMA's A operand): row
    // 32 wn + l31 of the tile.  They are recomputed per item from an opaque copy of the lane id
    // so that nothing of them is live (or spilled and reloaded behind the output stores) across
    // an epilogue.
    int arow, aswz, b0;
    auto lane_consts = [&]() {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int l31 = ln & 31, lh = ln >> 5, px = ln & 15;
        arow = ((4 * wm + (l31 >> 4)) * P_HW + px) * 128;
        aswz = 0;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) aswz |= (((lh ^ (((px + kx) >> 1) & 7)) & 7) << 4) << (8 * kx);
        b0 = (32 * wn + l31) * 128 + (((lh ^ ((l31 >> 1) & 7)) & 7) << 4);
    };
    lane_consts();

    cn_f32x16 acc[2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    };
    zero_acc();
    float rng_out = 0.f;
    cn_f32x4 resv[2][4];   // residual rows of the item (row layout), requested at the end of its last step

    auto lds128 = [&](int off) { return *reinterpret_cast<const p3_f16x8 *>(smem + off); };

    // one (tap, chunk) step: 12 fragment reads, 12 MFMAs.  Tap t sits at (t / 3, t % 3) of the 3x3
    // neighbourhood; in the parity form of the transposed convolution at (t / 2 + py, t % 2 + px)
    int par_y = 0, par_x = 0;      // parity of the current item (DECONV)
    auto step = [&](auto T, int hb, int wb) {
        constexpr int t = decltype(T)::value;
        const int ky = S2 ? p3_s2_oy(t) : (DECONV ? t / 2 + par_y : t / 3);
        const int kx = S2 ? p3_s2_ox(t) : (DECONV ? t % 2 + par_x : t % 3);
        const int tapoff = (ky * P_HW + kx) * 128;
        // quarter q: 0 = high k 0-15, 1 = high k 16-31, 2 = low k 0-15, 3 = low k 16-31
        if (DBG && (a.dbg & 2)) return;
        p3_f16x8 xf[4][2], wf[4];
        // the lane's row addresses pass through an opaque copy: otherwise the compiler keeps every
        // (tap, quarter, ring slot) address variant of the unrolled stage in its own register (~70)
        int ax = arow + ((aswz >> (8 * kx)) & 0xff) + (DECONV ? tapoff : 0), bx = b0;
        asm volatile("" : "+v"(ax), "+v"(bx));
        const int tapimm = DECONV ? 0 : tapoff;   // (a compile-time immediate in the 9-tap form)
        constexpr int blk1 = 2 * P_HW * 128;   // block 1 of the wave: two tile rows below block 0
        if (DBG && (a.dbg & 64)) {
            // ablation: half of the fragment reads (the low parts re-use the high parts' registers)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                wf[kh] = lds128(wb + (bx ^ (kh << 5)));
#pragma unroll
                for (int i = 0; i < 2; ++i) xf[kh][i] = lds128(hb + tapimm + i * blk1 + (ax ^ (kh << 5)));
                wf[2 + kh] = wf[kh];
#pragma unroll
                for (int i = 0; i < 2; ++i) xf[2 + kh][i] = xf[kh][i];
            }
        } else
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            // (the operands of the first products first)
            wf[kh] = lds128(wb + (bx ^ (kh << 5)));
#pragma unroll
            for (int i = 0; i < 2; ++i) xf[2 + kh][i] = lds128(hb + tapimm + i * blk1 + (ax ^ ((kh << 5) | 64)));
#pragma unroll
            for (int i = 0; i < 2; ++i) xf[kh][i] = lds128(hb + tapimm + i * blk1 + (ax ^ (kh << 5)));
            wf[2 + kh] = lds128(wb + (bx ^ ((kh << 5) | 64)));
        }
        // every fragment read is issued before the first MFMA (cn_conv3x3.hip: a ds_read sunk behind
        // an MFMA into that MFMA's operand registers can overwrite them before a queued MFMA reads them)
        __builtin_amdgcn_sched_barrier(0);
        if (DBG && (a.dbg & 1)) {   // keep the fragments live
#pragma unroll
            for (int q = 0; q < 4; ++q) asm volatile("" :: "v"(wf[q]), "v"(xf[q][0]), "v"(xf[q][1]));
            return;
        }
        if (!(a.knobs & 1)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            // smallest terms first: w_hi * x_lo, w_lo * x_hi, then w_hi * x_hi
#pragma unroll
            for (int i = 0; i < 2; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[kh], xf[2 + kh][i], acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[2 + kh], xf[kh][i], acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[kh], xf[kh][i], acc[i], 0, 0, 0);
        }
        if (!(a.knobs & 1)) __builtin_amdgcn_s_setprio(0);
    };

    // pixel of the row-layout lane: block i, pass k -> row 8k + (lane >> 3) of the wave's block
    // (lane-derived indices of the epilogue and the residual request are recomputed from an opaque
    // copy of the lane id where they are used: kept as loop invariants they cost ~20 registers)
    auto opaque_lane = [&]() {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        return ln;
    };
    auto row_pixel = [&](const P3Item &it, int i, int k, int rrow, bool &ok) {
        const int p = 64 * wm + 32 * i + 8 * k + rrow;
        const int oy = it.ty0 + (p >> 4), ox = it.tx0 + (p & 15);
        ok = oy < a.H && ox < a.W;
        if constexpr (DECONV)     // output pixel (2 y + py, 2 x + px) of the (2 H, 2 W) map
            return (it.b * 2 * a.H + 2 * oy + (it.par >> 1)) * 2 * a.W + 2 * ox + (it.par & 1);
        return (it.b * a.H + oy) * a.W + ox;
    };
    // raw buffer descriptor of the residual tensor (dword 3: 32-bit data format, gfx9 family)
    const __amdgpu_buffer_rsrc_t res_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(a.residual ? a.residual : a.x), 0, a.res_bytes, 0x00020000);
    auto load_residual = [&](const P3Item &it) {
        const int ln = opaque_lane();
        const int rrow = ln >> 3, rcol = ln & 7;
        const int grp = 2 * it.nb + wn;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                bool ok;
                const int pix = row_pixel(it, i, k, rrow, ok);
                const bool take = ok && grp < a.ngroups;
                // (buffer form: SGPR descriptor + 32-bit lane offset, no 64-bit address registers)
                const unsigned off = take ? (unsigned)pix * (unsigned)a.res_pitchB + (unsigned)(grp * 128 + rcol * 16) : 0u;
                resv[i][k] = __builtin_bit_cast(cn_f32x4, __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, off, 0, 0));
            }
    };

    // epilogue of one item; `sb` = LDS offset of the (dead) halo buffer its strips alias
    const float relu_floor = a.relu ? 0.f : -__builtin_inff();
    auto epilogue = [&](const P3Item &it, int sb) {
        const int grp = 2 * it.nb + wn;
        if (grp >= a.ngroups) return;          // wave-uniform: this wave's 32 channels do not exist
        const int stg = sb + wave * P_STG;
        const int ln = opaque_lane();
        const int rrow = ln >> 3, rcol = ln & 7, l31 = ln & 31, lh = ln >> 5;
        // scale / shift of the lane's channels 8g + 4 lh .. + 3 of the wave's group: from the wave's
        // LDS stash (requested at the item's first step, stored two steps later)
        const int rowb = stg + l31 * P_STG_ROW + 8 * lh;    // + 16 g (+ 64): f32s pieces; plain: 2x
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if constexpr (RES != 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    *reinterpret_cast<cn_f32x4 *>(smem + stg + (8 * k + rrow) * P_STG_ROW + rcol * 16) = resv[i][k];
                // (the strip is exchanged between lanes: the LDS serves a wave's accesses in order,
                // the compiler must keep them in order too)
                asm volatile("" ::: "memory");
            }
            // every residual piece of the wave is read before any output piece is written: with
            // different formats on the two sides a lane's output bytes are other lanes' residual bytes
            cn_f32x4 rr[4];
            if constexpr (RES != 0) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if constexpr (RES == 2)
                        rr[g] = *reinterpret_cast<const cn_f32x4 *>(smem + rowb + 8 * lh + 32 * g);
                    else
                        rr[g] = cn_join4(*reinterpret_cast<const cn_f16x4v *>(smem + rowb + 16 * g),
                                         *reinterpret_cast<const cn_f16x4v *>(smem + rowb + 64 + 16 * g));
                }
                asm volatile("" ::: "memory");
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const cn_f32x4 sc = *reinterpret_cast<const cn_f32x4 *>(smem + P_SSOFF + wave * 256 + (8 * g + 4 * lh) * 4);
                const cn_f32x4 sh = *reinterpret_cast<const cn_f32x4 *>(smem + P_SSOFF + wave * 256 + 128 + (8 * g + 4 * lh) * 4);
                cn_f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][4 * g + e] * sc[e] + sh[e];
                if constexpr (RES != 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaf(rr[g][e], a.res_mul, v[e]);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], relu_floor);
                if constexpr (OUT_PLAIN) {
                    *reinterpret_cast<cn_f32x4 *>(smem + rowb + 8 * lh + 32 * g) = v;
                } else {
                    cn_rng_upd4(rng_out, v);
                    cn_f16x4v h4, l4;
                    cn_split4(v, h4, l4);
                    *reinterpret_cast<cn_f16x4v *>(smem + rowb + 16 * g) = h4;
                    *reinterpret_cast<cn_f16x4v *>(smem + rowb + 64 + 16 * g) = l4;
                }
            }
            // rows back in row layout: 8 lanes = one 128-byte output row
            asm volatile("" ::: "memory");
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                bool ok;
                const int pix = row_pixel(it, i, k, rrow, ok);
                const cn_f32x4 v = *reinterpret_cast<const cn_f32x4 *>(smem + stg + (8 * k + rrow) * P_STG_ROW + rcol * 16);
                if (ok && !(DBG && (a.dbg & 32)))
                    *reinterpret_cast<cn_f32x4 *>(a.y + ((unsigned)pix * (unsigned)a.out_pitchB + (unsigned)(grp * 128 + rcol * 16))) = v;
            }
            asm volatile("" ::: "memory");
        }
    };

    P3Item cur = p3_decode(a, first), prev = cur;
    par_y = cur.par >> 1; par_x = cur.par & 1;
    int k = 0, c = 0, g = 0;
    unsigned long long pf_bar = 0, pf_epi = 0, pf_t0 = 0;
    auto now = [&]() { return DBG ? (unsigned long long)__builtin_readcyclecounter() : 0ull; };
    if (DBG) pf_t0 = now();
    auto bar = [&]() {
        const unsigned long long c0 = now();
        p3_barrier();
        if (DBG) pf_bar += now() - c0;
    };
#define P3_IC(v) std::integral_constant<int, (v)>{}
    if constexpr (PIPE) {
        // ---- pipelined fragment schedule.  A step is two blocks of six MFMAs (K halves 0 and 1), each on
        // its own set of six fragments.  The reads of the NEXT block go out in the middle of the current
        // one -- behind its fourth MFMA, into the set of the block BEFORE it (whose MFMAs have all left
        // the queue by then: the operand hazard of DESIGN 3.0 wants four MFMAs between a set's last use
        // and its refill) -- so the matrix pipe has work queued while fragments are in flight, and the
        // wave reaches a barrier with four MFMAs still to run.  The barrier of step t + 1 sits in the
        // middle of block (t, 1), in front of the reads of (t + 1, 0): the loaders' protocol (tile t + 1
        // and the stage's halo landed before it; every read of tile t - 1 complete before it) is the
        // unpipelined one.  Nothing is carried across an item boundary (the epilogue wants the registers).
        struct Frag { p3_f16x8 wh, wl, xh[2], xl[2]; };
        Frag F0, F1;
        auto load_half = [&](auto T, auto KH, int hb, int wb, Frag &F) {
            constexpr int t = decltype(T)::value, kh = decltype(KH)::value;
            const int ky = S2 ? p3_s2_oy(t) : (DECONV ? t / 2 + par_y : t / 3);
            const int kx = S2 ? p3_s2_ox(t) : (DECONV ? t % 2 + par_x : t % 3);
            const int tapoff = (ky * P_HW + kx) * 128;
            int ax = arow + ((aswz >> (8 * kx)) & 0xff) + (DECONV ? tapoff : 0), bx = b0;
            asm volatile("" : "+v"(ax), "+v"(bx));
            const int tapimm = DECONV ? 0 : tapoff;
            constexpr int blk1 = 2 * P_HW * 128;
            F.wh = lds128(wb + (bx ^ (kh << 5)));
#pragma unroll
            for (int i = 0; i < 2; ++i) F.xl[i] = lds128(hb + tapimm + i * blk1 + (ax ^ ((kh << 5) | 64)));
#pragma unroll
            for (int i = 0; i < 2; ++i) F.xh[i] = lds128(hb + tapimm + i * blk1 + (ax ^ (kh << 5)));
            F.wl = lds128(wb + (bx ^ ((kh << 5) | 64)));
        };
        // (same order of the products as the unpipelined step: smallest terms first)
        auto mfma_front = [&](const Frag &F) {
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.wh, F.xl[i], acc[i], 0, 0, 0);
            if constexpr (P3_SPLIT >= 4) {
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.wl, F.xh[i], acc[i], 0, 0, 0);
            }
            if constexpr (P3_SPLIT >= 6) {
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.wh, F.xh[i], acc[i], 0, 0, 0);
            }
        };
        auto mfma_back = [&](const Frag &F) {
            if constexpr (P3_SPLIT < 4) {
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.wl, F.xh[i], acc[i], 0, 0, 0);
            }
            if constexpr (P3_SPLIT < 6) {
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.wh, F.xh[i], acc[i], 0, 0, 0);
            }
        };
        auto keep = [&](const Frag &F) {
            asm volatile("" :: "v"(F.wh), "v"(F.wl), "v"(F.xh[0]), "v"(F.xh[1]), "v"(F.xl[0]), "v"(F.xl[1]));
        };
        bar();                                   // barrier 0 of stage 0
        int s = 0;                               // stage counter of the workgroup (halo buffer parity)
        for (k = 0; k < nit; ++k) {              // an item is entered behind barrier 0 of its first stage
            cur = p3_decode(a, first + k * nx);
            par_y = cur.par >> 1; par_x = cur.par & 1;
            // this lane's scale (lanes 0-31) or shift (32-63) value of the wave's 32 channels: requested
            // here, parked in the wave's LDS stash three steps later
            float ssv;
            {
                const int grp = 2 * cur.nb + wn;
                const int ln = opaque_lane();
                const unsigned o = (unsigned)(32 * grp + (ln & 31)) * 4u;
                float sv = 1.f, hv = 0.f;
                if (grp < a.ngroups) {
                    sv = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(a.scale) + o);
                    if (a.shift) hv = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(a.shift) + o);
                }
                ssv = (ln >> 5) ? hv : sv;
            }
            __builtin_amdgcn_sched_barrier(0);
            load_half(P3_IC(0), P3_IC(0), S2 ? 0 : (s & 1) * P_HBYTES, P_WOFF + (g & 3) * P_WSLOT, F0);
            __builtin_amdgcn_sched_barrier(0);
            for (c = 0; c < a.nchunk; ++c, ++s) {
                const int hb0 = (s & 1) * P_HBYTES, hb1 = ((s + 1) & 1) * P_HBYTES;
                auto HB = [&](int i) { return S2 ? p3_s2_buf(i) * P_HBYTES : hb0; };
                auto WB = [&](int i) { return P_WOFF + ((g + i) & 3) * P_WSLOT; };
                const bool more = (c + 1 < a.nchunk);     // another stage of this item follows
                auto pstep = [&](auto T) {
                    constexpr int t = decltype(T)::value;
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_setprio(1);
                    mfma_front(F0);
                    __builtin_amdgcn_sched_barrier(0);
                    load_half(T, P3_IC(1), HB(t), WB(t), F1);
                    __builtin_amdgcn_sched_barrier(0);
                    keep(F0);
                    mfma_back(F0);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_front(F1);
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_setprio(0);
                    // every fragment read of tile t is complete before its slot is handed back (the front
                    // MFMAs reference all six fragments, so this wait is already behind us: it is here to
                    // say so -- a front part that left fragments to the back part, P3_SPLIT = 2, was measured
                    // losing words of tile t to the DMA of tile t + 4 on the layers with the hottest weights)
                    p3_lds_fence();
                    p3_barrier();                // barrier t + 1 (t = NTAP - 1: barrier 0 of the next stage)
                    if constexpr (t + 1 < NTAP) {
                        load_half(P3_IC(t + 1), P3_IC(0), HB(t + 1), WB(t + 1), F0);
                    } else {
                        if (more) load_half(P3_IC(0), P3_IC(0), S2 ? 0 : hb1, WB(NTAP), F0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    keep(F1);
                    mfma_back(F1);
                    __builtin_amdgcn_sched_barrier(0);
                };
                pstep(P3_IC(0));
                pstep(P3_IC(1));
                pstep(P3_IC(2));
                if (c == 0) *reinterpret_cast<float *>(smem + P_SSOFF + wave * 256 + lane * 4) = ssv;
                pstep(P3_IC(3));
                if constexpr (!DECONV) {
                    pstep(P3_IC(4));
                    pstep(P3_IC(5));
                    pstep(P3_IC(6));
                    pstep(P3_IC(7));
                    pstep(P3_IC(8));
                }
                g += NTAP;
            }
            // behind barrier 0 of the next item's first stage (or the last barrier of all): the item's
            // last stage used buffer (s - 1) & 1, dead until the loader refills it behind the next barrier
            if constexpr (RES != 0) {
                __builtin_amdgcn_sched_barrier(0);
                load_residual(cur);
                __builtin_amdgcn_sched_barrier(0);
            }
            epilogue(cur, S2 ? P_HBYTES : ((s - 1) & 1) * P_HBYTES);
            zero_acc();
            lane_consts();
        }
    } else
    for (int s = 0;; ++s) {
        bar();
        if (s > 0 && c == 0) {
            // the previous item is complete: its last stage used buffer (s - 1) & 1, dead until the
            // loader refills it after this step's successor barrier
            const unsigned long long c0 = now();
            if (!(DBG && (a.dbg & 16))) epilogue(prev, S2 ? P_HBYTES : ((s - 1) & 1) * P_HBYTES);
            zero_acc();
            lane_consts();
            if (DBG) pf_epi += now() - c0;
        }
        if (s == S) break;
        const int hb0 = (s & 1) * P_HBYTES;
        auto HB = [&](int i) { return S2 ? p3_s2_buf(i) * P_HBYTES : hb0; };   // halo buffer of step i
        // per item: this lane's scale (lanes 0-31) or shift (32-63) value of the wave's 32 channels,
        // requested at the first step and parked in the wave's LDS stash two steps later
        float ssv = 0.f;
        if (c == 0) {
            const int grp = 2 * cur.nb + wn;
            const int ln = opaque_lane();
            const unsigned o = (unsigned)(32 * grp + (ln & 31)) * 4u;
            float sv = 1.f, hv = 0.f;
            if (grp < a.ngroups) {
                sv = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(a.scale) + o);
                if (a.shift) hv = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(a.shift) + o);
            }
            ssv = (ln >> 5) ? hv : sv;
        }
        step(P3_IC(0), HB(0), P_WOFF + ((g + 0) & 3) * P_WSLOT);
        bar(); step(P3_IC(1), HB(1), P_WOFF + ((g + 1) & 3) * P_WSLOT);
        bar(); step(P3_IC(2), HB(2), P_WOFF + ((g + 2) & 3) * P_WSLOT);
        if (c == 0) *reinterpret_cast<float *>(smem + P_SSOFF + wave * 256 + lane * 4) = ssv;
        bar(); step(P3_IC(3), HB(3), P_WOFF + ((g + 3) & 3) * P_WSLOT);
        if constexpr (!DECONV) {
        bar(); step(P3_IC(4), HB(4), P_WOFF + ((g + 4) & 3) * P_WSLOT);
        bar(); step(P3_IC(5), HB(5), P_WOFF + ((g + 5) & 3) * P_WSLOT);
        // the item's residual rows: requested RES_AT steps before the end of its last stage (probe:
        // requested behind the last step, the epilogue of a 64 -> 64 item waits ~3500 cycles for
        // them).  Held over the whole item the 32 registers push the kernel past 128 -- and two
        // 6-wave workgroups at 3 waves per SIMD leave the dispatcher no slack: measured, half of the
        // second workgroups then start only when a first one has finished
        auto maybe_residual = [&](int at) {
            if constexpr (RES != 0) {
                if (at == RES_AT && c == a.nchunk - 1) {
                    __builtin_amdgcn_sched_barrier(0);
                    load_residual(cur);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        maybe_residual(3);
        bar(); step(P3_IC(6), HB(6), P_WOFF + ((g + 6) & 3) * P_WSLOT);
        maybe_residual(2);
        bar(); step(P3_IC(7), HB(7), P_WOFF + ((g + 7) & 3) * P_WSLOT);
        maybe_residual(1);
        bar(); step(P3_IC(8), HB(8), P_WOFF + ((g + 8) & 3) * P_WSLOT);
        maybe_residual(0);
        }
        g += NTAP;
        if (++c == a.nchunk) {
            c = 0;
            prev = cur;
            ++k;
            if (k < nit) cur = p3_decode(a, first + k * nx);
            par_y = cur.par >> 1; par_x = cur.par & 1;
        }
    }
    if (DBG && a.prof && lane == 0) {
        unsigned long long *o = a.prof + ((size_t)blockIdx.x * 6 + wave) * 8;
        o[0] = now() - pf_t0; o[1] = pf_epi; o[2] = pf_bar; o[3] = (unsigned long long)S;
        o[4] = pf_rt0; o[5] = __builtin_amdgcn_s_memrealtime();
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        o[6] = xcc; o[7] = hwid;
    }
#undef P3_IC
    if constexpr (!OUT_PLAIN) {
        if (a.range) cn_rng_commit(a.range, 0, rng_out);
    }
}

}  // namespace

// Does the persistent kernel take this layer?  f32s input (whole 128-byte groups per pixel), output
// and residual as whole 32-channel groups (f32s, or plain fp32 at a pitch that is a multiple of 32).
bool cn_conv3x3p_takes(int B, int H, int W, int Cin, int Cout, int in_pitch, int out_pitch, int res_pitch,
                       bool in_plain, bool has_res)
{
    if (!cn_tune_c3p || in_plain) return false;   // (the caller also guarantees a non-null scale)
    if ((in_pitch & 31) || (out_pitch & 31) || (has_res && (res_pitch & 31))) return false;
    if (Cout % 32) return false;
    const long items = (long)B * cn_cdiv(H, P_TH) * cn_cdiv(W, P_TW) * cn_cdiv(Cout, 64);
    // enough items that the persistent grid fills the chip (key 28 = 2: every shape, for tests)
    if (cn_tune_c3p < 2 && items < 256) return false;
    if ((long)B * H * W * (long)max(in_pitch, max(out_pitch, res_pitch)) * 4 >= (1L << 31)) return false;
    (void)Cin;
    return true;
}

// ConvTranspose2d(4, stride 2, pad 1) in parity form on the persistent kernel: f32s input, output as
// whole 32-channel groups (f32s or plain), enough items to fill the chip
bool cn_deconv4x4s2p_takes(int B, int H, int W, int Cin, int Cout, int in_pitch, int out_pitch, bool in_plain)
{
    if (!cn_tune_c3p || !(cn_tune_c3p_deconv) || in_plain) return false;
    if ((in_pitch & 31) || (out_pitch & 31) || (Cout % 32)) return false;
    const long items = 4L * B * cn_cdiv(H, P_TH) * cn_cdiv(W, P_TW) * cn_cdiv(Cout, 64);
    if (cn_tune_c3p < 2 && items < 256) return false;
    if (4L * B * H * W * (long)max(in_pitch, out_pitch) * 4 >= (1L << 31)) return false;
    (void)Cin;
    return true;
}

int cn_deconv4x4s2_persist(const void *x, const void *w_packed, const float *scale, const float *shift, void *y,
                           int B, int H, int W, int Cin, int Cout, int in_pitch, int out_pitch, int relu,
                           int out_plain, const cn_f32s_ctl *ctl, hipStream_t st)
{
    P3Args a = {};
    a.x = (const char *)x; a.w = (const char *)w_packed; a.scale = scale; a.shift = shift;
    a.y = (char *)y;
    a.H = H; a.W = W;
    a.Hi = H; a.Wi = W;
    a.in_pitchB = in_pitch * 4; a.out_pitchB = out_pitch * 4;
    const int cin_pad = (Cin + 31) / 32 * 32;
    a.cin_padB = cin_pad * 4;
    a.cout_pad = (Cout + 31) / 32 * 32;
    a.ngroups = a.cout_pad / 32;
    a.nchunk = cin_pad / 32;
    a.nblk = cn_cdiv(Cout, 64);
    a.npar = 4;
    a.tiles_x = cn_cdiv(W, P_TW);
    a.tiles_y = cn_cdiv(H, P_TH);
    a.items = B * a.tiles_y * a.tiles_x * a.npar * a.nblk;
    a.relu = relu; a.out_plain = out_plain;
    a.res_mul = 1.f;
    a.range = ctl ? ctl->range : nullptr;
    a.stagger = cn_tune_c3p_stagger;
    a.knobs = cn_tune_c3p_knobs;
    int per_xcd = cn_cdiv(a.items, 8);
    if (per_xcd > 64) per_xcd = 64;
    const dim3 grid(8 * per_xcd), block(p3_threads(4, false));
#define P3_LAUNCH4(OP, PIPE)                                                                                     \
    do {                                                                                                         \
        CN_SET_MAX_LDS_ONCE((conv3x3p_kernel<0, OP, false, false, 4, false, PIPE>), P_LDS);                       \
        hipLaunchKernelGGL((conv3x3p_kernel<0, OP, false, false, 4, false, PIPE>), grid, block, P_LDS, st, a, P3Heads{}); \
    } while (0)
    if (a.knobs & 2) {
        if (out_plain) P3_LAUNCH4(true, true); else P3_LAUNCH4(false, true);
    } else {
        if (out_plain) P3_LAUNCH4(true, false); else P3_LAUNCH4(false, false);
    }
#undef P3_LAUNCH4
    CN_CHECK_LAUNCH();
    return CN_OK;
}

// 3x3 / stride 2 / pad 1 on the persistent kernel (parity-plane form): f32s tensors on both sides,
// whole 32-channel output groups, no residual
bool cn_conv3x3s2p_takes(int B, int Hi, int Wi, int Cin, int Cout, int in_pitch, int out_pitch)
{
    if (!cn_tune_c3p || !cn_tune_c3p_s2) return false;
    if ((in_pitch & 31) || (out_pitch & 31) || (Cout % 32)) return false;
    const int Ho = (Hi - 1) / 2 + 1, Wo = (Wi - 1) / 2 + 1;
    const long items = (long)B * cn_cdiv(Ho, P_TH) * cn_cdiv(Wo, P_TW) * cn_cdiv(Cout, 64);
    if (cn_tune_c3p < 2 && items < 256) return false;
    if ((long)B * Hi * Wi * (long)in_pitch * 4 >= (1L << 31) || (long)B * Ho * Wo * (long)out_pitch * 4 >= (1L << 31))
        return false;
    (void)Cin;
    return true;
}

int cn_conv3x3s2_persist(const void *x, const void *w_packed, const float *scale, const float *shift, void *y,
                         int B, int Hi, int Wi, int Cin, int Cout, int in_pitch, int out_pitch, int relu,
                         int out_plain, const cn_f32s_ctl *ctl, hipStream_t st)
{
    P3Args a = {};
    a.x = (const char *)x; a.w = (const char *)w_packed; a.scale = scale; a.shift = shift;
    a.y = (char *)y;
    a.H = (Hi - 1) / 2 + 1; a.W = (Wi - 1) / 2 + 1;
    a.Hi = Hi; a.Wi = Wi;
    a.in_pitchB = in_pitch * 4; a.out_pitchB = out_pitch * 4;
    const int cin_pad = (Cin + 31) / 32 * 32;
    a.cin_padB = cin_pad * 4;
    a.cout_pad = (Cout + 31) / 32 * 32;
    a.ngroups = a.cout_pad / 32;
    a.nchunk = cin_pad / 32;
    a.nblk = cn_cdiv(Cout, 64);
    a.npar = 1;
    a.tiles_x = cn_cdiv(a.W, P_TW);
    a.tiles_y = cn_cdiv(a.H, P_TH);
    a.items = B * a.tiles_y * a.tiles_x * a.nblk;
    a.relu = relu; a.out_plain = out_plain;
    a.res_mul = 1.f;
    a.range = ctl ? ctl->range : nullptr;
    a.stagger = cn_tune_c3p_stagger;
    a.knobs = cn_tune_c3p_knobs;
    int per_xcd = cn_cdiv(a.items, 8);
    if (per_xcd > 64) per_xcd = 64;
    const dim3 grid(8 * per_xcd), block(p3_threads(9, true));
#define P3_LAUNCHS2(OP, PIPE)                                                                                    \
    do {                                                                                                         \
        CN_SET_MAX_LDS_ONCE((conv3x3p_kernel<0, OP, false, false, 9, true, PIPE>), P_LDS);                        \
        hipLaunchKernelGGL((conv3x3p_kernel<0, OP, false, false, 9, true, PIPE>), grid, block, P_LDS, st, a, P3Heads{}); \
    } while (0)
    if (a.knobs & 2) {
        if (out_plain) P3_LAUNCHS2(true, true); else P3_LAUNCHS2(false, true);
    } else {
        if (out_plain) P3_LAUNCHS2(true, false); else P3_LAUNCHS2(false, false);
    }
#undef P3_LAUNCHS2
    CN_CHECK_LAUNCH();
    return CN_OK;
}

static int p3_probe_dbg = 0;
static unsigned long long *p3_probe_prof = nullptr;
// instrumented launches (tools/bench_c3p.py): dbg != 0 or prof != null route cn_conv3x3s1_persist to the
// DBG instantiation (no residual, f32s output) with these switches / counters
// workgroups of the given instantiation the runtime expects to be resident per CU
extern "C" int cn_conv3x3p_occupancy(int res, int out_plain)
{
    int n = -1;
    hipError_t e;
#define P3_OCC(R, OP)                                                                             \
    do {                                                                                          \
        CN_SET_MAX_LDS_ONCE((conv3x3p_kernel<R, OP>), P_LDS);                                      \
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv3x3p_kernel<R, OP>, 384, P_LDS);  \
    } while (0)
    if (out_plain) {
        if (res == 0) P3_OCC(0, true); else if (res == 1) P3_OCC(1, true); else P3_OCC(2, true);
    } else {
        if (res == 0) P3_OCC(0, false); else if (res == 1) P3_OCC(1, false); else P3_OCC(2, false);
    }
#undef P3_OCC
    return e == hipSuccess ? n : -(int)e;
}

extern "C" int cn_conv3x3p_probe(int dbg, void *prof)
{
    p3_probe_dbg = dbg;
    p3_probe_prof = (unsigned long long *)prof;
    return CN_OK;
}

int cn_conv3x3s1_persist(const void *x, const void *w_packed, const float *scale, const float *shift,
                         const void *residual, void *y, int B, int H, int W, int Cin, int Cout,
                         int in_pitch, int out_pitch, int res_pitch, int relu, int out_plain, int res_plain,
                         const cn_f32s_ctl *ctl, hipStream_t st)
{
    P3Args a = {};
    a.x = (const char *)x; a.w = (const char *)w_packed; a.scale = scale; a.shift = shift;
    a.residual = (const char *)residual; a.y = (char *)y;
    a.H = H; a.W = W;
    a.Hi = H; a.Wi = W;
    a.in_pitchB = in_pitch * 4; a.out_pitchB = out_pitch * 4; a.res_pitchB = res_pitch * 4;
    a.res_bytes = (int)((long)B * H * W * res_pitch * 4);   // < 2^31 (cn_conv3x3p_takes)
    const int cin_pad = (Cin + 31) / 32 * 32;
    a.cin_padB = cin_pad * 4;
    a.cout_pad = (Cout + 31) / 32 * 32;
    a.ngroups = a.cout_pad / 32;
    a.nchunk = cin_pad / 32;
    a.nblk = cn_cdiv(Cout, 64);
    a.npar = 1;
    a.tiles_x = cn_cdiv(W, P_TW);
    a.tiles_y = cn_cdiv(H, P_TH);
    a.items = B * a.tiles_y * a.tiles_x * a.nblk;
    a.relu = relu; a.out_plain = out_plain; a.res_plain = res_plain;
    a.res_mul = (ctl && ctl->res_mul != 0.f) ? ctl->res_mul : 1.f;
    a.range = ctl ? ctl->range : nullptr;
    a.stagger = cn_tune_c3p_stagger;
    a.knobs = cn_tune_c3p_knobs;
    // two workgroups per CU, a multiple of 8 (one share per XCD), never more than one per item
    int per_xcd = cn_cdiv(a.items, 8);
    if (per_xcd > 64) per_xcd = 64;
    const dim3 grid(8 * per_xcd), block(384);
#define P3_LAUNCH_(R, OP, PIPE)                                                                                  \
    do {                                                                                                         \
        CN_SET_MAX_LDS_ONCE((conv3x3p_kernel<R, OP, false, false, 9, false, PIPE>), P_LDS);                       \
        hipLaunchKernelGGL((conv3x3p_kernel<R, OP, false, false, 9, false, PIPE>), grid, block, P_LDS, st, a, P3Heads{}); \
    } while (0)
#define P3_LAUNCH(R, OP)                                                    \
    do {                                                                    \
        if (a.knobs & 2) P3_LAUNCH_(R, OP, true); else P3_LAUNCH_(R, OP, false); \
    } while (0)
    const int rmode = !residual ? 0 : (res_plain ? 2 : 1);
    if ((p3_probe_dbg || p3_probe_prof) && !out_plain && rmode < 2) {
        a.dbg = p3_probe_dbg;
        a.prof = p3_probe_prof;
        if (rmode == 0) {
            CN_SET_MAX_LDS_ONCE((conv3x3p_kernel<0, false, true>), P_LDS);
            hipLaunchKernelGGL((conv3x3p_kernel<0, false, true>), grid, block, P_LDS, st, a, P3Heads{});
        } else {
            CN_SET_MAX_LDS_ONCE((conv3x3p_kernel<1, false, true>), P_LDS);
            hipLaunchKernelGGL((conv3x3p_kernel<1, false, true>), grid, block, P_LDS, st, a, P3Heads{});
        }
        CN_CHECK_LAUNCH();
        return CN_OK;
    }
    if (out_plain) {
        if (rmode == 0) P3_LAUNCH(0, true); else if (rmode == 1) P3_LAUNCH(1, true); else P3_LAUNCH(2, true);
    } else {
        if (rmode == 0) P3_LAUNCH(0, false); else if (rmode == 1) P3_LAUNCH(1, false); else P3_LAUNCH(2, false);
    }
#undef P3_LAUNCH
#undef P3_LAUNCH_
    CN_CHECK_LAUNCH();
    return CN_OK;
}

// The fused heads on the persistent kernel: f32s feature map (whole 128-byte groups per pixel), one
// 64-wide hidden layer per head, every head <= 96 outputs, enough items to fill the chip.
bool cn_heads3x3p_takes(int B, int H, int W, int in_pitch, int head_conv, int n_heads, const cn_head_out *heads,
                        bool in_plain)
{
    if (!cn_tune_c3p || !cn_tune_c3p_heads || in_plain || head_conv != 64 || n_heads > P_MAXH) return false;
    if (in_pitch & 31) return false;
    for (int h = 0; h < n_heads; ++h)
        if (heads[h].cout > 96 || !heads[h].w_frag || !cn_aligned16(heads[h].w_frag)) return false;
    const long items = (long)B * cn_cdiv(H, P_TH) * cn_cdiv(W, P_TW) * n_heads;
    if (cn_tune_c3p < 2 && items < 256) return false;
    if ((long)B * H * W * (long)in_pitch * 4 >= (1L << 31)) return false;
    for (int h = 0; h < n_heads; ++h)
        if ((long)B * heads[h].cout * H * W >= (1L << 31)) return false;
    return true;
}

int cn_heads3x3p(const void *x, int B, int H, int W, int Cin, int in_pitch, const void *w1_packed,
                 const float *scale1, const float *bias1, int n_heads, const cn_head_out *heads,
                 const cn_f32s_ctl *ctl, hipStream_t st)
{
    P3Args a = {};
    P3Heads hd = {};
    for (int h = 0; h < n_heads; ++h) {
        hd.wf[h] = (const char *)heads[h].w_frag; hd.bias[h] = heads[h].bias; hd.oscale[h] = heads[h].oscale;
        hd.y[h] = heads[h].y; hd.cout[h] = heads[h].cout;
    }
    a.x = (const char *)x; a.w = (const char *)w1_packed; a.scale = scale1; a.shift = bias1;
    a.H = H; a.W = W;
    a.Hi = H; a.Wi = W;
    a.in_pitchB = in_pitch * 4;
    const int cin_pad = (Cin + 31) / 32 * 32;
    a.cin_padB = cin_pad * 4;
    a.cout_pad = 64 * n_heads;
    a.ngroups = a.cout_pad / 32;
    a.nchunk = cin_pad / 32;
    a.nblk = n_heads;
    a.npar = 1;
    a.tiles_x = cn_cdiv(W, P_TW);
    a.tiles_y = cn_cdiv(H, P_TH);
    a.items = B * a.tiles_y * a.tiles_x * a.nblk;
    a.relu = 1;
    a.res_mul = 1.f;
    a.range = ctl ? ctl->range : nullptr;
    a.stagger = cn_tune_c3p_stagger;
    a.knobs = cn_tune_c3p_knobs;
    int per_xcd = cn_cdiv(a.items, 8);
    if (per_xcd > 64) per_xcd = 64;
    const dim3 grid(8 * per_xcd), block(384);
    // the heads exist in the pipelined schedule only: their unpipelined instantiation kept three barriers
    // reachable with an LDS read in flight (tools/audit_barriers.py) and is no longer built; key 30 bit 1
    // does not reach this launch
    a.knobs |= 2;
    CN_SET_MAX_LDS_ONCE((conv3x3p_kernel<0, false, false, true, 9, false, true>), P_LDS);
    hipLaunchKernelGGL((conv3x3p_kernel<0, false, false, true, 9, false, true>), grid, block, P_LDS, st, a, hd);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

// ---- 1x1 matrix of a head as MFMA-ready fragments for the kernel above
namespace {
__global__ void pack_head_w2_kernel(const float *__restrict__ w, char *__restrict__ out, int cout, int nblk2)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;      // (jb, j, s2, lane)
    if (t >= nblk2 * 4 * 64) return;
    const int ln = t & 63, s2 = (t >> 6) & 1, j = (t >> 7) & 1, jb = t >> 8;
    const int l31 = ln & 31, lh = ln >> 5;
    const int row = jb * 32 + l31;
    _Float16 hi[8], lo[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        // K order of step s2 of hidden block j: lane half lh holds 32 j + 16 s2 + 4 lh + {0..3}, then + 8
        const int k = 32 * j + 16 * s2 + 8 * (q >> 2) + 4 * lh + (q & 3);
        const float v = row < cout ? w[(size_t)row * 64 + k] : 0.f;
        const float c = fminf(fmaxf(v, -65504.0f), 65504.0f);
        hi[q] = (_Float16)c;
        lo[q] = (_Float16)(c - (float)hi[q]);
    }
    char *o = out + ((size_t)((jb * 2 + j) * 2 + s2) * 2 * 64 + ln) * 16;
    *reinterpret_cast<p3_f16x8 *>(o) = *reinterpret_cast<const p3_f16x8 *>(hi);
    *reinterpret_cast<p3_f16x8 *>(o + 64 * 16) = *reinterpret_cast<const p3_f16x8 *>(lo);
}
}  // namespace

extern "C" size_t cn_packed_head_w2_bytes(int cout)
{
    return cout > 0 ? (size_t)((cout + 31) / 32) * 8192 : 0;
}

extern "C" int cn_pack_head_w2_f32s(const float *w, void *out, int cout, void *stream)
{
    if (!w || !out) return CN_ERR_NULL;
    if (cout <= 0) return CN_ERR_SHAPE;
    if (!cn_aligned16(out)) return CN_ERR_ALIGN;
    const int nblk2 = (cout + 31) / 32;
    hipLaunchKernelGGL(pack_head_w2_kernel, dim3(nblk2), dim3(256), 0, (hipStream_t)stream, w, (char *)out, cout, nblk2);
    CN_CHECK_LAUNCH();
    return CN_OK;
}
