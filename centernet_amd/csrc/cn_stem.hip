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
#include <type_traits>

int cn_tune_stem_stagger = 0;   // cn_set_tuning key 44: start delay of the second resident workgroup of the stem + max-pool kernel, units of 256 cycles
int cn_tune_stem_dbg = 0;   // cn_set_tuning key 43: probe switches of the stem + max-pool kernel (StemArgs.dbg)

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
    float x_mul;      // f32s kernels: the image is multiplied by this (2^-e) before it is split
    uint32_t *range;  // f32s kernels: [1] receives max |x * x_mul| (cn_f32s_ctl), may be null
    int y_f32s;       // stem + max-pool kernel: y is an f32s tensor (CN_CONV_STEM_Y_F32S), range side 0 = max |y|
    int stagger;      // stem + max-pool kernel: start delay of workgroups >= 256 (the second occupant of every CU), units of 256 cycles
    int dbg;          // stem + max-pool kernel, probe instantiation (cn_set_tuning key 43): 1 = no MFMAs, 2 = no window
                      // stores, 4 = no image loads, 8 = no pooling / output stores, 16 = no barriers
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


// ---------------------------------------------------------------------------------------
// Persistent variant for 7x7 stems whose output rows are a multiple of 128 pixels wide (every
// network at 512x512).  gridDim.x workgroups (two per CU) walk the tile list:
//   * weights are staged ONCE per workgroup;
//   * the input window of the NEXT tile is fetched into registers while the current tile's
//     MFMAs run (the zero-fill select is deferred to the LDS store, so nothing in between
//     waits on the loads);
//   * K is ordered as (c, ky-pair, kx): the two lane halves of v_mfma_f32_32x32x2_f32 take
//     ky = 2p and ky = 2p+1 of the same (c, kx), i.e. LDS addresses exactly one window row
//     apart.  With stride and window pitch compile-time, every A-fragment read is
//     `ds_read_b32 base offset:imm` -- no offset table, no address arithmetic in the K loop.
//     ky = 7 (the pad of the 4th pair) has zero weight and reads the next plane's first row
//     (finite data) or, for the last plane, a zeroed spare row.
// A tile = 128 consecutive output pixels of one output row: window = 7 rows x WX columns
// x 3 planes, WX = 127*stride + 7.
constexpr int PKH = 7, PKW = 7, PROWS = 3 * PKH;
constexpr int PPE = 3 * 4 * PKW;   // 84 (c, ky-pair, kx) elements
constexpr int PKP = 2 * PPE;       // 168 K values
constexpr int PLDW = PKP + 4;      // LDS weight row pitch (floats)

template <int BN, int S>
__global__ __launch_bounds__(NT) void stem_persist_f32_kernel(const StemArgs a, int total_tiles)
{
    // BN == 16 (DLA's base_layer, 3 -> 16): v_mfma_f32_16x16x4_f32 instead of the 32x32x2 form,
    // so that no half of the N tile is padding.  K is then ordered (c, ky-quad, kx): the four
    // lane quarters take ky = 4g .. 4g+3 of the same (c, kx) -- LDS addresses one window row
    // apart, again `ds_read_b32 base offset:imm` -- and the lane's 42 weights live in registers
    // for the whole persistent loop (no weight tile in LDS at all).
    constexpr bool N16 = (BN == 16);
    constexpr int WN = N16 ? 1 : BN / 32;
    constexpr int WM = 4 / WN;
    constexpr int TM = BM / WM;
    constexpr int MB = N16 ? TM / 16 : TM / 32;
    constexpr int NE16 = 3 * 2 * PKW;  // 42 (c, ky-quad, kx) elements of the N16 form
    constexpr int WX = (BM - 1) * S + PKW;
    constexpr int WXP = WX | 1;
    constexpr int WIN_FLOATS = ((PROWS + 1) * WXP + 3) & ~3;  // + one zero row
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *win = reinterpret_cast<float *>(smem);   // [22][WXP]
    float *dummy = win + WIN_FLOATS;                 // [NT] sink of masked-off window stores
    float *Ws = dummy + NT;                          // [BN][PLDW]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int n0 = blockIdx.y * BN;
    const int tpr = a.Wo / BM;  // tiles per output row

    const int l15 = lane & 15, lq = lane >> 4;
    float wreg[N16 ? NE16 : 1];
    if constexpr (N16) {
#pragma unroll
        for (int e = 0; e < NE16; ++e) {
            const int c = e / (2 * PKW), g = (e / PKW) % 2, kx = e % PKW;
            const int ky = 4 * g + lq;
            const int nn = min(n0 + l15, a.cout_pad - 1);
            wreg[e] = (ky < PKH && n0 + l15 < a.Cout)
                          ? a.w[(size_t)nn * a.KP + (ky * PKW + kx) * 3 + c] : 0.f;
        }
    }
    // ---- once per workgroup: weights of this N tile, permuted from the packed
    // [cout_pad][KP] (k = tap*3 + c) layout into the (group, half, element) K order
    if constexpr (!N16)
    for (int i = tid; i < BN * PKP; i += NT) {
        const int nrow = i / PKP, k = i - nrow * PKP;
        const int g = k >> 3, half = (k >> 2) & 1, e = k & 3;
        const int pe = g * 4 + e;
        const int c = pe / (4 * PKW), rem = pe - c * (4 * PKW);
        const int kyp = rem / PKW, kx = rem - kyp * PKW;
        const int ky = 2 * kyp + half;
        const int nn = min(n0 + nrow, a.cout_pad - 1);  // rows past the padded Cout: never stored
        float w = 0.f;
        if (ky < PKH) w = a.w[(size_t)nn * a.KP + (ky * PKW + kx) * 3 + c];
        Ws[nrow * PLDW + k] = w;
    }
    for (int i = tid; i < WXP; i += NT) win[PROWS * WXP + i] = 0.f;  // spare zero row

    // tail columns (>= 256) of all 21 window rows ride in one extra load per thread
    const int trow = tid >> 3, tcol = NT + (tid & 7);
    const bool tail_ok = trow < PROWS && tcol < WX;
    const int tc = trow / PKH, twy = trow - tc * PKH;
    const int tail_dst = tail_ok ? trow * WXP + tcol : WIN_FLOATS + tid;
    const bool col_ok = tid < WX;

    float v[PROWS], vt;
    unsigned vmask = 0;
    auto prefetch = [&](int tile) {
        const int xt = tile % tpr;
        const int rowid = tile / tpr;  // b*Ho + oy
        const int b = rowid / a.Ho, oy = rowid - b * a.Ho;
        const int iy_min = oy * S - a.pad, ix_min = xt * BM * S - a.pad;
        const float *xb = a.x + (size_t)b * 3 * a.H * a.W;
        const int ix = ix_min + tid;
        const bool cok = col_ok && ix >= 0 && ix < a.W;
        unsigned mk = 0;
#pragma unroll
        for (int u = 0; u < PROWS; ++u) {
            const int c = u / PKH, wy = u % PKH;
            const int iy = iy_min + wy;
            const bool ok = cok && iy >= 0 && iy < a.H;
            v[u] = xb[ok ? ((size_t)(c * a.H + iy) * a.W + ix) : 0];
            mk |= ok ? (1u << u) : 0u;
        }
        {
            const int iy = iy_min + twy, jx = ix_min + tcol;
            const bool ok = tail_ok && iy >= 0 && iy < a.H && jx >= 0 && jx < a.W;
            vt = xb[ok ? ((size_t)(tc * a.H + iy) * a.W + jx) : 0];
            mk |= ok ? (1u << PROWS) : 0u;
        }
        vmask = mk;
    };
    auto store_window = [&]() {  // branch-free: masked-off lanes write to the sink
#pragma unroll
        for (int u = 0; u < PROWS; ++u)
            win[col_ok ? u * WXP + tid : WIN_FLOATS + tid] = ((vmask >> u) & 1u) ? v[u] : 0.f;
        win[tail_dst] = ((vmask >> PROWS) & 1u) ? vt : 0.f;
    };

    const float *abase[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
        abase[i] = N16 ? win + (wm * TM + i * 16 + l15) * S + lq * WXP
                       : win + (wm * TM + i * 32 + l31) * S + lh * WXP;
    const float *wrow = Ws + (wn * 32 + l31) * PLDW + 4 * lh;
    const int n = N16 ? n0 + l15 : n0 + wn * 32 + l31;
    float sc = (a.scale && n < a.Cout) ? a.scale[n] : 1.f;
    float sf = (a.shift && n < a.Cout) ? a.shift[n] : 0.f;
    asm volatile("" : "+v"(sc), "+v"(sf));  // settle these loads before the tile loop

    int tile = blockIdx.x;
    if (tile < total_tiles) prefetch(tile);
    for (; tile < total_tiles; tile += gridDim.x) {
        store_window();
        __syncthreads();  // window (and, first time, weights) visible
        const int next = tile + gridDim.x;
        if (next < total_tiles) prefetch(next);  // in flight during the MFMAs below

        if constexpr (N16) {
            cn_f32x4 acc4[MB];
#pragma unroll
            for (int i = 0; i < MB; ++i) acc4[i] = cn_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < NE16; ++e) {
                const int c = e / (2 * PKW), g = (e / PKW) % 2, kx = e % PKW;
                const int off = (c * PKH + 4 * g) * WXP + kx;  // compile-time immediate
#pragma unroll
                for (int i = 0; i < MB; ++i)
                    acc4[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(abase[i][off], wreg[e], acc4[i],
                                                                   0, 0, 0);
            }
            // D of the 16x16 MFMA: col = lane & 15 (cout), rows 4*(lane >> 4) + r (pixels)
            if (n < a.Cout) {
                const int xt = tile % tpr;
                const int rowid = tile / tpr;
                float *yb = a.y + ((size_t)rowid * a.Wo + (size_t)xt * BM) * a.out_pitch + n;
#pragma unroll
                for (int i = 0; i < MB; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m = wm * TM + i * 16 + 4 * lq + r;
                        float t = acc4[i][r] * sc + sf;
                        if (a.relu) t = fmaxf(t, 0.f);
                        yb[(size_t)m * a.out_pitch] = t;
                    }
            }
            __syncthreads();  // every wave is done reading the window
            continue;
        }
        cn_f32x16 acc[MB];
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
        for (int g = 0; g < PKP / 8; ++g) {
            const cn_f32x4 bf = *reinterpret_cast<const cn_f32x4 *>(wrow + g * 8);
            float af[MB][4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int pe = g * 4 + e;
                const int c = pe / (4 * PKW), rem = pe % (4 * PKW);
                const int kyp = rem / PKW, kx = rem % PKW;
                const int off = (c * PKH + 2 * kyp) * WXP + kx;  // compile-time immediate
#pragma unroll
                for (int i = 0; i < MB; ++i) af[i][e] = abase[i][off];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MB; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[e], acc[i], 0, 0, 0);
        }

        // epilogue: BN (scale/shift) + ReLU, NHWC; lanes run along Cout (128-byte rows)
        if (n < a.Cout) {
            const int xt = tile % tpr;
            const int rowid = tile / tpr;
            float *yb = a.y + ((size_t)rowid * a.Wo + (size_t)xt * BM) * a.out_pitch + n;
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    float t = acc[i][r] * sc + sf;
                    if (a.relu) t = fmaxf(t, 0.f);
                    yb[(size_t)m * a.out_pitch] = t;
                }
        }
        __syncthreads();  // every wave is done reading the window
    }
}

// ---------------------------------------------------------------------------------------
// f32s form of the persistent stem (7x7 / stride 2, 32 or 64 output channels per tile): the
// same tile walk and register prefetch, but the window lives in LDS as two fp16 planes (high
// and low parts of the fp32 pixels, split while it is staged) and K is consumed 16 at a time
// by v_mfma_f32_32x32x16_f16 -- three per step (hi*hi + hi*lo + lo*hi), 33 per 32 x 32 block
// instead of 84 fp32 matrix instructions of twice the latency.  K order: step s, lane half h,
// element j  <->  window row r = 2s + h (= c*7 + ky), kx = j; j = 7 and r = 21 carry zero
// weights (and read the next pixel / the zeroed spare row).  A lane's eight kx of a row are
// eight consecutive fp16 of the window -- 4-byte aligned at stride 2 -- fetched as four
// ds_read_b32 with immediate offsets.  Output stays plain fp32 (the max-pool reads it).
constexpr int SKS = 11;            // 16-deep K steps (22 (c, kx) groups of 8 incl. the zero group)
constexpr int SKP = 16 * SKS;      // 176 K values per part
constexpr int SLDW = SKP + 8;      // fp16 per LDS weight row (368 bytes: conflict-free b128 reads)
// Round 6: the window is COLUMN-major.  K order: step s, lane half h, element j  <->  group g = 2s + h =
// (c, kx) = (g / 7, g % 7), ky = j; j = 7 and g = 21 carry zero weights.  A plane holds, per channel and
// window column, the column's seven ky values + a zero as one 16-byte group: a lane's A fragment of a step
// is ONE ds_read_b128 per plane (was four ds_read_b32), a staging thread writes its column's channel as
// ONE ds_write_b128 per plane (was seven ds_write_b16), conversions run on pairs.  Columns are XOR-swizzled
// (slot = col ^ ((col >> 4) & 1)) so that the 16-lane groups of a ds_read_b128 -- lanes two columns
// apart at stride 2 -- hit 16 distinct bank groups.  Counters of the row-major form at B = 32: 14.7 VALU
// instructions per MFMA, matrix pipe 26 % busy (profiles/r05_sq_counters_v3_cfg1.txt).
constexpr int CM_WXC = 264;                 // window columns per channel (261 carry data)
constexpr int CM_CH = CM_WXC * 16;          // bytes per channel of a plane
constexpr int CM_PLANE = 3 * CM_CH;         // 12672 bytes per plane (high / low)
__device__ __forceinline__ int cm_swz(int col) { return col ^ ((col >> 4) & 1); }
typedef _Float16 st_f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t st_u32x4 __attribute__((ext_vector_type(4)));

template <int BN>
__global__ __launch_bounds__(NT) void stem_persist_f32s_kernel(const StemArgs a, int total_tiles)
{
    constexpr int S = 2;
    constexpr int WN = BN / 32;
    constexpr int WM = 4 / WN;
    constexpr int TM = BM / WM;
    constexpr int MB = TM / 32;
    constexpr int WX = (BM - 1) * S + PKW;           // 261 columns carry data
    static_assert(WX <= CM_WXC - 3, "window columns");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *winH = smem;                                               // [3][CM_WXC][8] fp16: high parts
    char *winL = smem + CM_PLANE;                                    // low parts
    _Float16 *WsH = reinterpret_cast<_Float16 *>(smem + 2 * CM_PLANE);   // [BN][SLDW], 16-byte aligned
    _Float16 *WsL = WsH + BN * SLDW;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int n0 = blockIdx.y * BN;
    const int tpr = a.Wo / BM;

    // ---- once per workgroup: zero both planes (the spare row and column 261 stay zero), and
    // the weights of this N tile, split and permuted into the (step, half, kx) K order
    for (int i = tid; i < 2 * CM_PLANE / 4; i += NT) reinterpret_cast<uint32_t *>(smem)[i] = 0u;
    for (int i = tid; i < BN * SKP; i += NT) {
        const int nrow = i / SKP, k = i - nrow * SKP;
        const int st = k >> 4, h = (k >> 3) & 1, j = k & 7;
        const int r = 2 * st + h;                  // group (c, kx); element j = ky
        const int c = r / PKW, kx = r - c * PKW;
        const int nn = min(n0 + nrow, a.cout_pad - 1);
        float w = 0.f;
        if (r < 3 * PKW && j < PKH) w = a.w[(size_t)nn * a.KP + (j * PKW + kx) * 3 + c];
        const _Float16 hi = (_Float16)w;
        WsH[nrow * SLDW + k] = hi;
        WsL[nrow * SLDW + k] = (_Float16)(w - (float)hi);
    }
    __syncthreads();

    // staging: thread t owns window column t (all 21 (c, ky) values); columns 256 .. 260 are taken by
    // threads 128 .. 142 as one (c, column) group of seven rows each
    constexpr int NTAIL = 3 * (WX - NT);
    const int tt = tid - 128;
    const bool tail_thr = tt >= 0 && tt < NTAIL;
    const int tc = tail_thr ? tt / (WX - NT) : 0, tcol = NT + (tail_thr ? tt % (WX - NT) : 0);

    float v[PROWS], vt[PKH];
    unsigned vrow = 0;          // bit u: window row u = (c, ky) lies inside the image (uniform)
    unsigned vcol = 0, vtcol = 0;   // all ones: this thread's column / tail column lies inside the image
    auto prefetch = [&](int tile) {
        const int xt = tile % tpr;
        const int rowid = tile / tpr;  // b*Ho + oy
        const int b = rowid / a.Ho, oy = rowid - b * a.Ho;
        const int iy_min = oy * S - a.pad, ix_min = xt * BM * S - a.pad;
        const float *xb = a.x + (size_t)b * 3 * a.H * a.W;
        const int ix = ix_min + tid, jx = ix_min + tcol;
        const bool cok = ix >= 0 && ix < a.W;
        const bool tok = tail_thr && jx >= 0 && jx < a.W;
        const int ixc = cok ? ix : 0, jxc = tok ? jx : 0;
        unsigned mk = 0;
#pragma unroll
        for (int u = 0; u < PROWS; ++u) {
            const int c = u / PKH, wy = u % PKH;
            const int iy = iy_min + wy;
            const bool rok = iy >= 0 && iy < a.H;            // uniform
            const float *rowp = xb + (size_t)(c * a.H + (rok ? iy : 0)) * a.W;
            v[u] = rowp[ixc];
            mk |= rok ? (1u << u) : 0u;
        }
        if (tail_thr) {
#pragma unroll
            for (int wy = 0; wy < PKH; ++wy) {
                const int iy = iy_min + wy;
                const bool rok = iy >= 0 && iy < a.H;
                vt[wy] = xb[(size_t)(tc * a.H + (rok ? iy : 0)) * a.W + jxc];
            }
        }
        vrow = mk;
        vcol = cok ? 0xffffffffu : 0u;
        vtcol = tok ? 0xffffffffu : 0u;
    };
    uint32_t rng_bits = 0;      // max |x| as a bit pattern; a NaN reads above +inf's pattern and stays
    // one (channel, column) group: seven rows -> (x * x_mul) split into 8 + 8 fp16 (ky = 7: zero), two 16-byte stores
    auto put7 = [&](int c, int col, const float *src, unsigned cmask, unsigned rowbits) {
        float x[8];
#pragma unroll
        for (int ky = 0; ky < PKH; ++ky) {
            const uint32_t keep = cmask & (((rowbits >> ky) & 1u) ? 0xffffffffu : 0u);
            const float xs = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, src[ky]) & keep) * a.x_mul;
            const uint32_t ab = __builtin_bit_cast(uint32_t, xs) & 0x7fffffffu;
            rng_bits = rng_bits > ab ? rng_bits : ab;
            x[ky] = __builtin_amdgcn_fmed3f(xs, -65504.0f, 65504.0f);
        }
        x[7] = 0.f;
        st_u32x4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            uint32_t h2, l2;
            cn_split2_bits(x[2 * e], x[2 * e + 1], h2, l2);
            hi[e] = h2; lo[e] = l2;
        }
        const int off = c * CM_CH + cm_swz(col) * 16;
        *reinterpret_cast<st_u32x4 *>(winH + off) = hi;
        *reinterpret_cast<st_u32x4 *>(winL + off) = lo;
    };
    auto store_window = [&]() {
#pragma unroll
        for (int c = 0; c < 3; ++c) put7(c, tid, v + c * PKH, vcol, vrow >> (c * PKH));
        if (tail_thr) put7(tc, tcol, vt, vtcol, vrow >> (tc * PKH));
    };

    // A fragment of step st, block i: group g = 2 st + lh = (c, kx), this lane's pixel's window column
    // 2 * pixel + kx -- one 16-byte read per plane; the byte offsets do not depend on the tile
    int aoff[SKS][MB];
#pragma unroll
    for (int st = 0; st < SKS; ++st) {
        const int g = min(2 * st + lh, 3 * PKW - 1);      // group 21 has zero weights: any valid address
        const int c = g / PKW, kx = g - c * PKW;
#pragma unroll
        for (int i = 0; i < MB; ++i)
            aoff[st][i] = c * CM_CH + cm_swz((wm * TM + i * 32 + l31) * S + kx) * 16;
    }
    const _Float16 *wH = WsH + (wn * 32 + l31) * SLDW + 8 * lh;
    const _Float16 *wL = WsL + (wn * 32 + l31) * SLDW + 8 * lh;
    const int n = n0 + wn * 32 + l31;
    float sc = (a.scale && n < a.Cout) ? a.scale[n] : 1.f;
    float sf = (a.shift && n < a.Cout) ? a.shift[n] : 0.f;
    asm volatile("" : "+v"(sc), "+v"(sf));

    int tile = blockIdx.x;
    if (tile < total_tiles) prefetch(tile);
    for (; tile < total_tiles; tile += gridDim.x) {
        store_window();
        __syncthreads();  // window visible
        const int next = tile + gridDim.x;
        if (next < total_tiles) prefetch(next);  // in flight during the MFMAs below

        cn_f32x16 acc[MB];
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        // operand sets alternate by step parity; the set a load targets was last read by MFMAs
        // issued a whole step earlier (see the operand hazard note in cn_conv.hip)
        st_u32x4 fa[2][2][MB];      // [set][hi / lo][block]
        st_f16x8 fb[2][2];          // [set][hi / lo]
        auto load_step = [&](int set, int st) {
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                fa[set][0][i] = *reinterpret_cast<const st_u32x4 *>(winH + aoff[st][i]);
                fa[set][1][i] = *reinterpret_cast<const st_u32x4 *>(winL + aoff[st][i]);
            }
            fb[set][0] = *reinterpret_cast<const st_f16x8 *>(wH + 16 * st);
            fb[set][1] = *reinterpret_cast<const st_f16x8 *>(wL + 16 * st);
        };
        load_step(0, 0);
#pragma unroll
        for (int st = 0; st < SKS; ++st) {
            const int cur = st & 1;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MB; ++i)    // lo * hi
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                    __builtin_bit_cast(st_f16x8, fa[cur][1][i]), fb[cur][0], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (st + 1 < SKS) load_step(cur ^ 1, st + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MB; ++i)    // hi * lo
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                    __builtin_bit_cast(st_f16x8, fa[cur][0][i]), fb[cur][1], acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MB; ++i)    // hi * hi
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                    __builtin_bit_cast(st_f16x8, fa[cur][0][i]), fb[cur][0], acc[i], 0, 0, 0);
        }

        // epilogue: BN (scale/shift) + ReLU, NHWC plain fp32; lanes run along Cout
        if (n < a.Cout) {
            const int xt = tile % tpr;
            const int rowid = tile / tpr;
            float *yb = a.y + ((size_t)rowid * a.Wo + (size_t)xt * BM) * a.out_pitch + n;
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    float t = acc[i][r] * sc + sf;
                    if (a.relu) t = fmaxf(t, 0.f);
                    yb[(size_t)m * a.out_pitch] = t;
                }
        }
        __syncthreads();  // every wave is done reading the window
    }
    if (a.range) cn_rng_commit(a.range, 1, rng_bits > 0x7f800000u ? __builtin_inff() : __builtin_bit_cast(float, rng_bits));
}

// ---------------------------------------------------------------------------------------
// f32s stem + MaxPool2d(3, stride 2, padding 1) in one kernel (resnet_dcn.py:138-141,
// msra_resnet.py: conv1 -> bn1 -> relu -> maxpool): the 7x7 output (537 MB at B = 32, 512^2) is
// never written.  A workgroup owns a STRIP of the image -- full rows, R pooled rows high -- and
// walks its stem rows top to bottom, the TPR 128-pixel tiles of a row left to right.  Per tile:
//   * BN + ReLU on the accumulators; a lane holds, for one channel, pixel groups {4g .. 4g+3};
//   * horizontal 3-max at stride 2: pooled column 2g+1 = max(4g+1, 4g+2, 4g+3) is lane-local,
//     pooled column 2g = max(4g-1, 4g, 4g+1) takes pixel 4g-1 from the partner lane (other half
//     of the wave), from the other wave pair (pixel 63) or from the previous tile (pixel 127 of
//     the tile to the left) through a 1 KB LDS line;
//   * vertical 3-max as a running maximum in registers (16 per tile column): an even stem row
//     joins it, an odd row 2p+1 completes pooled row p -- which is stored -- and starts p+1.
// The strip's first row (2*p0 - 1) is computed only to start the running maximum: one extra row
// per 2R (6 % at R = 8).  max() is exact, so the result equals stem -> max-pool bit for bit.
template <int TPR, bool DBG>
__global__ __launch_bounds__(NT) void stem_pool_f32s_kernel(const StemArgs a, int nstrips, int R)
{
    const int dbg = DBG ? (a.dbg & 255) : 0;
    if (a.stagger && blockIdx.x >= 256u) {
        // two workgroups share a CU and walk identical strips: started together they sit in their MFMA
        // loops -- and in their staging / pooling phases -- at the same time and nothing overlaps.  One-off
        // phase shift of the second occupant.
        const unsigned long long t0 = __builtin_readcyclecounter();
        while (__builtin_readcyclecounter() - t0 < (unsigned long long)a.stagger * 256u) __builtin_amdgcn_s_sleep(32);
    }
    constexpr int BN = 64;
    constexpr int S = 2;
    constexpr int WN = 2, WM = 2, TM = BM / WM, MB = TM / 32;
    constexpr int WX = (BM - 1) * S + PKW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *winH = smem;                                               // column-major planes, see stem_persist_f32s_kernel
    char *winL = smem + CM_PLANE;
    _Float16 *WsH = reinterpret_cast<_Float16 *>(smem + 2 * CM_PLANE);
    _Float16 *WsL = WsH + BN * SLDW;
    float *bnd = reinterpret_cast<float *>(WsL + BN * SLDW);   // [2 parity][2 wm][64]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int PH = a.Ho / 2, PW = a.Wo / 2;
    const int spi = PH / R;                  // strips per image
    const float NEG_INF = -__builtin_huge_valf();

    for (int i = tid; i < 2 * CM_PLANE / 4; i += NT) reinterpret_cast<uint32_t *>(smem)[i] = 0u;
    if (!(dbg & 1 && dbg & 2 && dbg & 4 && dbg & 8 && dbg & 16))
    for (int i = tid; i < BN * SKP; i += NT) {
        const int nrow = i / SKP, k = i - nrow * SKP;
        const int st = k >> 4, h = (k >> 3) & 1, j = k & 7;
        const int r = 2 * st + h;                  // group (c, kx); element j = ky
        const int c = r / PKW, kx = r - c * PKW;
        const int nn = min(nrow, a.cout_pad - 1);
        float w = 0.f;
        if (r < 3 * PKW && j < PKH) w = a.w[(size_t)nn * a.KP + (j * PKW + kx) * 3 + c];
        const _Float16 hi = (_Float16)w;
        WsH[nrow * SLDW + k] = hi;
        WsL[nrow * SLDW + k] = (_Float16)(w - (float)hi);
    }
    __syncthreads();

    // staging: thread t owns window column t (all 21 (c, ky) values); columns 256 .. 260 are taken by
    // threads 128 .. 142 as one (c, column) group of seven rows each
    constexpr int NTAIL = 3 * (WX - NT);
    const int tt = tid - 128;
    const bool tail_thr = tt >= 0 && tt < NTAIL;
    const int tc = tail_thr ? tt / (WX - NT) : 0, tcol = NT + (tail_thr ? tt % (WX - NT) : 0);

    float v[PROWS], vt[PKH];
    unsigned vrow = 0;          // bit u: window row u = (c, ky) lies inside the image (uniform)
    unsigned vcol = 0, vtcol = 0;   // all ones: this thread's column / tail column lies inside the image
    auto prefetch = [&](int b, int oy, int xt) {
        const int iy_min = oy * S - a.pad, ix_min = xt * BM * S - a.pad;
        const float *xb = a.x + (size_t)b * 3 * a.H * a.W;
        const int ix = ix_min + tid, jx = ix_min + tcol;
        const bool cok = ix >= 0 && ix < a.W;
        const bool tok = tail_thr && jx >= 0 && jx < a.W;
        const int ixc = cok ? ix : 0, jxc = tok ? jx : 0;
        unsigned mk = 0;
#pragma unroll
        for (int u = 0; u < PROWS; ++u) {
            const int c = u / PKH, wy = u % PKH;
            const int iy = iy_min + wy;
            const bool rok = iy >= 0 && iy < a.H;            // uniform
            const float *rowp = xb + (size_t)(c * a.H + (rok ? iy : 0)) * a.W;
            v[u] = rowp[ixc];
            mk |= rok ? (1u << u) : 0u;
        }
        if (tail_thr) {
#pragma unroll
            for (int wy = 0; wy < PKH; ++wy) {
                const int iy = iy_min + wy;
                const bool rok = iy >= 0 && iy < a.H;
                vt[wy] = xb[(size_t)(tc * a.H + (rok ? iy : 0)) * a.W + jxc];
            }
        }
        vrow = mk;
        vcol = cok ? 0xffffffffu : 0u;
        vtcol = tok ? 0xffffffffu : 0u;
    };
    uint32_t rng_bits = 0;      // max |x| as a bit pattern; a NaN reads above +inf's pattern and stays
    float rng_out = 0.f;
    // one (channel, column) group: seven rows -> (x * x_mul) split into 8 + 8 fp16 (ky = 7: zero), two 16-byte stores
    auto put7 = [&](int c, int col, const float *src, unsigned cmask, unsigned rowbits) {
        float x[8];
#pragma unroll
        for (int ky = 0; ky < PKH; ++ky) {
            const uint32_t keep = cmask & (((rowbits >> ky) & 1u) ? 0xffffffffu : 0u);
            const float xs = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, src[ky]) & keep) * a.x_mul;
            const uint32_t ab = __builtin_bit_cast(uint32_t, xs) & 0x7fffffffu;
            rng_bits = rng_bits > ab ? rng_bits : ab;
            x[ky] = __builtin_amdgcn_fmed3f(xs, -65504.0f, 65504.0f);
        }
        x[7] = 0.f;
        st_u32x4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            uint32_t h2, l2;
            cn_split2_bits(x[2 * e], x[2 * e + 1], h2, l2);
            hi[e] = h2; lo[e] = l2;
        }
        const int off = c * CM_CH + cm_swz(col) * 16;
        *reinterpret_cast<st_u32x4 *>(winH + off) = hi;
        *reinterpret_cast<st_u32x4 *>(winL + off) = lo;
    };
    auto store_window = [&]() {
#pragma unroll
        for (int c = 0; c < 3; ++c) put7(c, tid, v + c * PKH, vcol, vrow >> (c * PKH));
        if (tail_thr) put7(tc, tcol, vt, vtcol, vrow >> (tc * PKH));
    };

    // A fragment of step st, block i: group g = 2 st + lh = (c, kx), this lane's pixel's window column
    // 2 * pixel + kx -- one 16-byte read per plane; the byte offsets do not depend on the tile
    int aoff[SKS][MB];
#pragma unroll
    for (int st = 0; st < SKS; ++st) {
        const int g = min(2 * st + lh, 3 * PKW - 1);      // group 21 has zero weights: any valid address
        const int c = g / PKW, kx = g - c * PKW;
#pragma unroll
        for (int i = 0; i < MB; ++i)
            aoff[st][i] = c * CM_CH + cm_swz((wm * TM + i * 32 + l31) * S + kx) * 16;
    }
    const _Float16 *wH = WsH + (wn * 32 + l31) * SLDW + 8 * lh;
    const _Float16 *wL = WsL + (wn * 32 + l31) * SLDW + 8 * lh;
    const int n = wn * 32 + l31;
    float sc = (a.scale && n < a.Cout) ? a.scale[n] : 1.f;
    float sf = (a.shift && n < a.Cout) ? a.shift[n] : 0.f;
    asm volatile("" : "+v"(sc), "+v"(sf));

    // strip s: image b = s / spi, pooled rows p0 .. p0 + R - 1, stem rows 2*p0 - 1 .. 2*(p0+R) - 1
    // (row -1 of the first strip does not exist: the running maximum starts at -inf)
    auto first_row = [&](int s) { const int p0 = (s % spi) * R; return p0 == 0 ? 0 : 2 * p0 - 1; };
    int s = blockIdx.x;
    if (s >= nstrips) return;
    int y = first_row(s), cb = 0, par = 0;
    prefetch(s / spi, y, 0);
    float cur[TPR][8 * MB];
    while (true) {
        const int b = s / spi, p0 = (s % spi) * R;
        const int ylast = 2 * (p0 + R) - 1;
        if (cb == 0 && y == first_row(s)) {
#pragma unroll
            for (int t = 0; t < TPR; ++t)
#pragma unroll
                for (int e = 0; e < 8 * MB; ++e) cur[t][e] = NEG_INF;
        }
        if (!(dbg & 2)) store_window();
        if (!(dbg & 16)) __syncthreads();  // window visible
        // the tile after this one (possibly the first of the next strip): in flight during the MFMAs
        int ns = s, ny = y, ncb = cb + 1;
        if (ncb == TPR) { ncb = 0; ny = y + 1; }
        if (ny > ylast) { ns = s + gridDim.x; ny = ns < nstrips ? first_row(ns) : 0; }
        const bool more = ns < nstrips;
        if (more && !(dbg & 4)) prefetch(ns / spi, ny, ncb);

        cn_f32x16 acc[MB];
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        st_u32x4 fa[2][2][MB];
        st_f16x8 fb[2][2];
        auto load_step = [&](int set, int st) {
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                fa[set][0][i] = *reinterpret_cast<const st_u32x4 *>(winH + aoff[st][i]);
                fa[set][1][i] = *reinterpret_cast<const st_u32x4 *>(winL + aoff[st][i]);
            }
            fb[set][0] = *reinterpret_cast<const st_f16x8 *>(wH + 16 * st);
            fb[set][1] = *reinterpret_cast<const st_f16x8 *>(wL + 16 * st);
        };
        if (!(dbg & 1)) load_step(0, 0);
#pragma unroll
        for (int st = 0; st < SKS; ++st) {
            if (dbg & 1) break;
            const int cs = st & 1;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MB; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                    __builtin_bit_cast(st_f16x8, fa[cs][1][i]), fb[cs][0], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (st + 1 < SKS) load_step(cs ^ 1, st + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MB; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                    __builtin_bit_cast(st_f16x8, fa[cs][0][i]), fb[cs][1], acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MB; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                    __builtin_bit_cast(st_f16x8, fa[cs][0][i]), fb[cs][0], acc[i], 0, 0, 0);
        }

        // ---- BN + ReLU, then the pooling; acc[i][4q + j] = pixel wm*64 + i*32 + 8q + 4lh + j
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float t = acc[i][r] * sc + sf;
                acc[i][r] = a.relu ? fmaxf(t, 0.f) : t;
            }
        // this wave pair's last pixel (63 / 127) goes to the LDS line of this tile's parity
        if (lh == 1) bnd[(par * 2 + wm) * BN + n] = acc[MB - 1][15];
        if (!(dbg & 16)) __syncthreads();   // also: every wave is done reading the window
        if (dbg & 8) {
            if (!more) break;
            par ^= 1;
            s = ns; y = ny; cb = ncb;
            continue;
        }
        float lastp[MB][4];   // the partner lane's pixel 4g' + 3 of every group it holds
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) lastp[i][q] = __shfl_xor(acc[i][4 * q + 3], 32);
        // pixel -1 of this wave pair's 64 pixels
        float edge = NEG_INF;
        if (wm == 1) edge = bnd[(par * 2 + 0) * BN + n];
        else if (cb > 0) edge = bnd[((par ^ 1) * 2 + 1) * BN + n];
        float hp[MB][4][2];
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // group g = 2q + lh (+ 8i): pixel 4g - 1 is the last pixel of group g - 1
                float left;
                if (lh == 1) left = lastp[i][q];
                else left = q > 0 ? lastp[i][q - 1] : (i > 0 ? lastp[i - 1][3] : edge);
                hp[i][q][0] = fmaxf(fmaxf(left, acc[i][4 * q]), acc[i][4 * q + 1]);
                hp[i][q][1] = fmaxf(fmaxf(acc[i][4 * q + 1], acc[i][4 * q + 2]), acc[i][4 * q + 3]);
            }
        const bool odd = (y & 1) != 0;
#pragma unroll
        for (int t = 0; t < TPR; ++t) {
            if (t != cb) continue;
            if (odd && y > first_row(s)) {
                // completes pooled row (y - 1) / 2
                const int pr = (y - 1) >> 1;
                if (a.y_f32s) {
                    // f32s output: a pooled pixel's 32 channels are one 128-byte group (32 high halves, 32 low
                    // halves) and a lane holds ONE channel of 16 pooled pixels -- 2-byte stores and a 64-bit address
                    // per value were the largest item of this kernel (probe: 55 of 205 us).  The wave's 32 pixels x
                    // 128 bytes go through a wave-private LDS strip instead -- four 1 KiB chunks of the window
                    // columns this wave itself stages (the window is dead behind the barrier above; no other
                    // wave writes them) -- and leave as 16-byte stores of whole groups.
                    const uint32_t nkeep = n < a.Cout ? 0xffffffffu : 0u;
                    char *strip[4] = {winH + wave * 1024, winH + CM_CH + wave * 1024, winH + 2 * CM_CH + wave * 1024,
                                      winL + wave * 1024};
#pragma unroll
                    for (int i = 0; i < MB; ++i)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            float v0 = fmaxf(cur[t][(i * 4 + q) * 2], hp[i][q][0]);
                            float v1 = fmaxf(cur[t][(i * 4 + q) * 2 + 1], hp[i][q][1]);
                            cn_rng_upd1(rng_out, v0);
                            cn_rng_upd1(rng_out, v1);
                            uint32_t hb, lb;
                            cn_split2_bits(__builtin_amdgcn_fmed3f(v0, -65504.0f, 65504.0f),
                                           __builtin_amdgcn_fmed3f(v1, -65504.0f, 65504.0f), hb, lb);
                            hb &= nkeep;             // pad channels of the last group: zero
                            lb &= nkeep;
                            // local pooled pixel 16 i + 4 q + 2 lh + h2: chunk 2 i + (q >> 1), row 4 (q & 1) + 2 lh + h2
                            char *dst = strip[2 * i + (q >> 1)] + (4 * (q & 1) + 2 * lh) * 128 + l31 * 2;
                            *reinterpret_cast<uint16_t *>(dst) = (uint16_t)hb;
                            *reinterpret_cast<uint16_t *>(dst + 64) = (uint16_t)lb;
                            *reinterpret_cast<uint16_t *>(dst + 128) = (uint16_t)(hb >> 16);
                            *reinterpret_cast<uint16_t *>(dst + 192) = (uint16_t)(lb >> 16);
                        }
                    // LDS operations of one wave complete in order: the reads below see the writes above
                    const size_t pitchB = (size_t)a.out_pitch * 4;
                    char *rowp = reinterpret_cast<char *>(a.y) + ((size_t)(b * PH + pr) * PW + (size_t)cb * (BM / 2)) * pitchB;
                    const unsigned lane_off = (unsigned)(32 * wm + (lane >> 3)) * (unsigned)pitchB + (unsigned)wn * 128u + (unsigned)(lane & 7) * 16u;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const st_u32x4 g = *reinterpret_cast<const st_u32x4 *>(strip[k] + lane * 16);
                        *reinterpret_cast<st_u32x4 *>(rowp + lane_off + (unsigned)(8 * k) * (unsigned)pitchB) = g;
                    }
                } else {
                    float *yb = a.y + ((size_t)(b * PH + pr) * PW + (size_t)cb * (BM / 2)) * a.out_pitch + n;
#pragma unroll
                    for (int i = 0; i < MB; ++i)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
#pragma unroll
                            for (int h2 = 0; h2 < 2; ++h2) {
                                const int pc = 2 * (2 * q + lh + 8 * i + 16 * wm) + h2;
                                if (n < a.Cout) yb[(size_t)pc * a.out_pitch] = fmaxf(cur[t][(i * 4 + q) * 2 + h2], hp[i][q][h2]);
                            }
                }
            }
            if (odd) {      // uniform: a branch, not sixteen selects
#pragma unroll
                for (int e = 0; e < 8 * MB; ++e) cur[t][e] = hp[e >> 3][(e >> 1) & 3][e & 1];
            } else {
#pragma unroll
                for (int e = 0; e < 8 * MB; ++e) cur[t][e] = fmaxf(cur[t][e], hp[e >> 3][(e >> 1) & 3][e & 1]);
            }
        }
        if (!more) break;
        par ^= 1;
        s = ns; y = ny; cb = ncb;
    }
    if (a.range) {
        cn_rng_commit(a.range, 1, rng_bits > 0x7f800000u ? __builtin_inff() : __builtin_bit_cast(float, rng_bits));
        if (a.y_f32s) cn_rng_commit(a.range, 0, rng_out);
    }
}

template <int TPR>
int launch_stem_pool_f32s(const StemArgs &a0, int B, int R, hipStream_t st)
{
    constexpr size_t lds = (size_t)2 * CM_PLANE + (size_t)2 * 64 * SLDW * 2 + 2 * 2 * 64 * 4;
    StemArgs a = a0;
    a.dbg = cn_tune_stem_dbg & 255;
    a.stagger = cn_tune_stem_stagger;
    const int nstrips = B * (a.Ho / 2 / R);
    int wgs = nstrips < 512 ? nstrips : 512;  // two resident workgroups per CU
    if (a.stagger == 1000) { wgs = nstrips < 256 ? nstrips : 256; a.stagger = 0; }   // probe: one workgroup per CU
    if (a.stagger == 1001) { wgs = nstrips < 768 ? nstrips : 768; a.stagger = 0; }
    if (a.dbg) {
        CN_SET_MAX_LDS_ONCE((stem_pool_f32s_kernel<TPR, true>), lds);
        hipLaunchKernelGGL((stem_pool_f32s_kernel<TPR, true>), dim3(wgs), dim3(NT), lds, st, a, nstrips, R);
    } else {
        CN_SET_MAX_LDS_ONCE((stem_pool_f32s_kernel<TPR, false>), lds);
        hipLaunchKernelGGL((stem_pool_f32s_kernel<TPR, false>), dim3(wgs), dim3(NT), lds, st, a, nstrips, R);
    }
    CN_CHECK_LAUNCH();
    return CN_OK;
}

template <int BN>
int launch_stem_persist_f32s(const StemArgs &a, int B, hipStream_t st)
{
    constexpr size_t lds = (size_t)2 * CM_PLANE + (size_t)2 * BN * SLDW * 2;
    const long total = (long)B * a.tiles_per_image;
    const int wgs = (int)(total < 512 ? total : 512);  // two resident workgroups per CU
    dim3 grid(wgs, cn_cdiv(a.Cout, BN));
    CN_SET_MAX_LDS_ONCE((stem_persist_f32s_kernel<BN>), lds);
    hipLaunchKernelGGL((stem_persist_f32s_kernel<BN>), grid, dim3(NT), lds, st, a, (int)total);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

// ---------------------------------------------------------------------------------------
// f32s form of the 7x7 / stride 1 / pad 3 stem with <= 16 output channels (DLA base_layer,
// pose_dla_dcn.py:229-233) on v_mfma_f32_16x16x32_f16.  The fp32 form (stem_persist_f32_kernel
// <16, 1>) spends 37 x 32-cycle matrix instructions per 16 x 16 block (0.58 ms at B = 32, 512^2, twice
// the fp32 matrix roof's 0.25 ms); here K = 21 window rows (c, ky) x 8 (kx = 0..6 + a zero-weight
// pad) is walked as six steps of four rows -- lane group q = lane >> 4 holds k = 8q..8q+7 = row
// 4s + q, kx = 0..7 -- i.e. 18 instructions of 16 cycles per block.
// A lane's eight kx of a row are eight CONSECUTIVE fp16 of the window starting at its own pixel:
// 2-byte granularity at stride 1.  Every plane is therefore kept twice, the second copy shifted
// by one element, so that even and odd start columns both read four aligned dwords with
// immediate offsets (copy E: column e at index e; copy O: column e at index e + 1).
// Window = image columns x0 - 4 .. x0 + 131 (origin even: pairs of columns are 8-byte loads),
// rows (c, ky) 0..20 + three zero rows for the zero-weight tail of step 5; high and low planes;
// split ONCE while staged (image * x_mul, range word fed there).  Weights: the fp32 stem pack
// [cout_pad][KP], k = (ky*7 + kx)*3 + c, split into registers at kernel start.  Output plain fp32.
constexpr int S16_ROWB = 288;                    // bytes per window row of one copy (136 + 1 columns, padded)
constexpr int S16_ROWS = 24;
constexpr int S16_COPY = S16_ROWS * S16_ROWB;    // 6912 bytes per copy
constexpr int S16_PAIRS = 68;                    // column pairs per row
typedef _Float16 s16_f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t s16_u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(NT, 2) void stem16s_kernel(const StemArgs a, int total_tiles)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [hi E | hi O | lo E | lo O]
    char *hiE = smem, *hiO = smem + S16_COPY, *loE = smem + 2 * S16_COPY, *loO = smem + 3 * S16_COPY;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int tpr = a.Wo / BM;
    const float x_mul = a.x_mul;

    // zero rows 21..23 of every copy (finite data under the zero weights of step 5)
    for (int i = tid; i < 4 * 3 * (S16_ROWB / 4); i += NT) {
        const int cp = i / (3 * (S16_ROWB / 4)), r = i - cp * (3 * (S16_ROWB / 4));
        *reinterpret_cast<uint32_t *>(smem + cp * S16_COPY + 21 * S16_ROWB + r * 4) = 0u;
    }

    // weights of this lane: output channel l15, step s: window row 4s + lq = (c, ky), kx = 0..6 (+ zero)
    s16_f16x8 wh[6], wl[6];
#pragma unroll
    for (int s2 = 0; s2 < 6; ++s2) {
        const int r = 4 * s2 + lq;
        const int c = r / 7, ky = r - c * 7;
        float w8[8];
#pragma unroll
        for (int kx = 0; kx < 8; ++kx) {
            const bool ok = r < 21 && kx < 7 && l15 < a.Cout;
            w8[kx] = ok ? a.w[(size_t)l15 * a.KP + ((ky * 7 + (kx < 7 ? kx : 0)) * 3 + c)] : 0.f;
        }
        uint32_t h[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) cn_split2_bits(w8[2 * e], w8[2 * e + 1], h[e], l[e]);
        wh[s2] = __builtin_bit_cast(s16_f16x8, s16_u32x4{h[0], h[1], h[2], h[3]});
        wl[s2] = __builtin_bit_cast(s16_f16x8, s16_u32x4{l[0], l[1], l[2], l[3]});
    }
    const float sc = (a.scale && l15 < a.Cout) ? a.scale[l15] : 1.f;
    const float sf = (a.shift && l15 < a.Cout) ? a.shift[l15] : 0.f;

    // staging: 21 rows x 68 column pairs over 256 threads; per pair the two columns and the one to
    // their left (the shifted copy's pair)
    constexpr int NPR = (21 * S16_PAIRS + NT - 1) / NT;   // 6
    // TWO tiles ahead: a tile is ~1 us of work, a window row comes from HBM in 2-3 us -- one tile
    // of prefetch distance left every tile waiting for its loads (0.43 ms)
    float v0[2][NPR], v1[2][NPR], vm[2][NPR];
    float rng_in = 0.f;
    auto prefetch = [&](auto SET, int tile) {
        constexpr int st = decltype(SET)::value;
        const int xt = tile % tpr;
        const int rowid = tile / tpr;                    // b*Ho + oy
        const int b = rowid / a.Ho, oy = rowid - b * a.Ho;
        const int ix0 = xt * BM - 4;
        const float *img = a.x + (size_t)b * 3 * a.H * a.W;
#pragma unroll
        for (int u = 0; u < NPR; ++u) {
            const int i = tid + u * NT;
            const int row = min(i / S16_PAIRS, 20), pp = i % S16_PAIRS;   // (idle slots of the last trip: clamped, unused)
            const int c = row / 7, ky = row - c * 7;
            const int iy = oy - 3 + ky, ix = ix0 + 2 * pp;
            const bool rok = i < 21 * S16_PAIRS && iy >= 0 && iy < a.H;
            const bool ok = rok && ix >= 0 && ix < a.W;            // (W even: the pair is in or out as a whole)
            const bool okm = rok && ix - 1 >= 0 && ix - 1 < a.W;
            const float *rowp = img + ((size_t)c * a.H + (rok ? iy : 0)) * a.W;
            const float2 pr = *reinterpret_cast<const float2 *>(rowp + (ok ? ix : 0));
            const float lm = rowp[okm ? ix - 1 : 0];
            v0[st][u] = ok ? pr.x : 0.f;
            v1[st][u] = ok ? pr.y : 0.f;
            vm[st][u] = okm ? lm : 0.f;
        }
    };
    auto store_window = [&](auto SET) {
        constexpr int st = decltype(SET)::value;
#pragma unroll
        for (int u = 0; u < NPR; ++u) {
            const int i = tid + u * NT;
            if (i < 21 * S16_PAIRS) {
                const int row = i / S16_PAIRS, pp = i - row * S16_PAIRS;
                const float s0 = v0[st][u] * x_mul, s1 = v1[st][u] * x_mul, sm = vm[st][u] * x_mul;   // real -> stored units
                cn_rng_upd1_in(rng_in, s0);
                cn_rng_upd1_in(rng_in, s1);
                const float c0 = fminf(fmaxf(s0, -65504.f), 65504.f), c1 = fminf(fmaxf(s1, -65504.f), 65504.f);
                const float cm = fminf(fmaxf(sm, -65504.f), 65504.f);
                uint32_t he, le, ho, lo;
                cn_split2_bits(c0, c1, he, le);       // copy E: columns (2p, 2p+1) at indices (2p, 2p+1)
                cn_split2_bits(cm, c0, ho, lo);       // copy O: columns (2p-1, 2p) at indices (2p, 2p+1)
                const int o = row * S16_ROWB + 4 * pp;
                *reinterpret_cast<uint32_t *>(hiE + o) = he;
                *reinterpret_cast<uint32_t *>(loE + o) = le;
                *reinterpret_cast<uint32_t *>(hiO + o) = ho;
                *reinterpret_cast<uint32_t *>(loO + o) = lo;
            }
        }
    };

    // A fragment base of (16-pixel block mb): start column x + 1 of the lane's pixel, row lq
    unsigned abase[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        const int sc0 = wave * 32 + mb * 16 + l15 + 1;          // window column of kx = 0
        abase[mb] = (unsigned)(((sc0 & 1) ? S16_COPY : 0) + lq * S16_ROWB + 2 * (sc0 + (sc0 & 1)));
    }

    auto one_tile = [&](auto SET, int tile) {   // SET: the register set that holds this tile's window
        store_window(SET);
        __syncthreads();
        const int next = tile + 2 * (int)gridDim.x;
        if (next < total_tiles) prefetch(SET, next);

        // three phases of two K steps; fragments in two register sets: a set is reloaded only when
        // the MFMAs that read it are a full phase back (operand hazard note, cn_conv.hip)
        s16_u32x4 fh[2][2][2], fl[2][2][2];    // [set][step of the phase][mb]
        auto load_phase = [&](auto SET, int ph) {
            constexpr int st = decltype(SET)::value;
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    const char *pa = smem + abase[mb] + (2 * ph + q) * 4 * S16_ROWB;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        fh[st][q][mb][e] = *reinterpret_cast<const uint32_t *>(pa + 4 * e);
                        fl[st][q][mb][e] = *reinterpret_cast<const uint32_t *>(pa + 2 * S16_COPY + 4 * e);
                    }
                }
        };
        cn_f32x4 acc[2] = {cn_f32x4{0.f, 0.f, 0.f, 0.f}, cn_f32x4{0.f, 0.f, 0.f, 0.f}};
        auto mfma_phase = [&](auto SET, auto PH) {
            constexpr int st = decltype(SET)::value;
            constexpr int ph = decltype(PH)::value;
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    const s16_f16x8 xh = __builtin_bit_cast(s16_f16x8, fh[st][q][mb]);
                    const s16_f16x8 xl = __builtin_bit_cast(s16_f16x8, fl[st][q][mb]);
                    acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl, wh[2 * ph + q], acc[mb], 0, 0, 0);
                    acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, wl[2 * ph + q], acc[mb], 0, 0, 0);
                    acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, wh[2 * ph + q], acc[mb], 0, 0, 0);
                }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        load_phase(I0{}, 0);
        load_phase(I1{}, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_phase(I0{}, I0{});
        mfma_phase(I1{}, I1{});
        __builtin_amdgcn_sched_barrier(0);
        load_phase(I0{}, 2);
        __builtin_amdgcn_sched_barrier(0);
        mfma_phase(I0{}, I2{});
        // D: col = lane & 15 (cout), rows 4*(lane >> 4) + r (pixels)
        const int xt = tile % tpr;
        const int rowid = tile / tpr;
        float *yb = a.y + ((size_t)rowid * a.Wo + (size_t)xt * BM) * a.out_pitch;
        if (l15 < a.Cout) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = wave * 32 + mb * 16 + 4 * lq + r;
                    float t = acc[mb][r] * sc + sf;
                    if (a.relu) t = fmaxf(t, 0.f);
                    yb[(size_t)m * a.out_pitch + l15] = t;
                }
        }
        __syncthreads();
    };
    {
        const int g = (int)gridDim.x;
        int tile = blockIdx.x;
        if (tile < total_tiles) prefetch(std::integral_constant<int, 0>{}, tile);
        if (tile + g < total_tiles) prefetch(std::integral_constant<int, 1>{}, tile + g);
        for (; tile < total_tiles; tile += 2 * g) {
            one_tile(std::integral_constant<int, 0>{}, tile);
            if (tile + g < total_tiles) one_tile(std::integral_constant<int, 1>{}, tile + g);
        }
    }
    if (a.range) cn_rng_commit(a.range, 1, rng_in);
}

int launch_stem16s(const StemArgs &a, int B, hipStream_t st)
{
    constexpr size_t lds = (size_t)4 * S16_COPY;
    const long total = (long)B * a.tiles_per_image;
    const int wgs = (int)(total < 512 ? total : 512);  // two resident workgroups per CU
    hipLaunchKernelGGL(stem16s_kernel, dim3(wgs), dim3(NT), lds, st, a, (int)total);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

template <int BN, int S>
int launch_stem_persist(const StemArgs &a, int B, hipStream_t st)
{
    constexpr int WXP = ((BM - 1) * S + PKW) | 1;
    constexpr size_t lds = (size_t)((((PROWS + 1) * WXP + 3) & ~3) + NT + (BN == 16 ? 0 : BN * PLDW)) * 4;
    const long total = (long)B * a.tiles_per_image;
    const int wgs = (int)(total < 512 ? total : 512);  // two resident workgroups per CU
    dim3 grid(wgs, cn_cdiv(a.Cout, BN));
    CN_SET_MAX_LDS_ONCE((stem_persist_f32_kernel<BN, S>), lds);
    hipLaunchKernelGGL((stem_persist_f32_kernel<BN, S>), grid, dim3(NT), lds, st, a, (int)total);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

}  // namespace

// Pooled rows per strip of the fused stem + max-pool kernel, or 0 when the shape is outside it:
// 7x7 / stride 2, 33..64 output channels, rows of 1..4 whole 128-pixel tiles, an even number of
// rows, and enough strips to fill the chip (shorter strips recompute more rows: 1 in 2R).
int cn_stem_pool_rows(int B, int Ho, int Wo, int Cout, int KH, int KW, int stride, int KP)
{
    if (KH != PKH || KW != PKW || stride != 2 || Cout <= 32 || Cout > 64) return 0;
    if ((KP & 7) || KP < PKH * PKW * 3) return 0;
    if (Wo % BM || Wo / BM > 4 || (Ho & 1)) return 0;
    const int PH = Ho / 2;
    if (PH % 8 == 0 && (long)B * (PH / 8) >= 384) return 8;
    if (PH % 4 == 0 && (long)B * (PH / 4) >= 192) return 4;
    return 0;
}

int cn_stem_pool_f32s(const float *x, const float *w_packed, const float *scale, const float *shift,
                      float *y, int B, int H, int W, int Ho, int Wo, int Cout, int KH, int KW,
                      int stride, int pad, int relu, int out_pitch, int KP, int y_f32s, const cn_f32s_ctl *ctl,
                      hipStream_t st)
{
    const int R = cn_stem_pool_rows(B, Ho, Wo, Cout, KH, KW, stride, KP);
    if (!R) return CN_ERR_UNSUPPORTED;
    if (y_f32s && ((out_pitch & 31) || (((uintptr_t)y) & 127u))) return CN_ERR_ALIGN;
    StemArgs a;
    a.y_f32s = y_f32s;
    a.x_mul = (ctl && ctl->x_mul != 0.f) ? ctl->x_mul : 1.f;
    a.range = ctl ? ctl->range : nullptr;
    a.x = x; a.w = w_packed; a.scale = scale; a.shift = shift; a.y = y;
    a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout; a.KH = KH; a.KW = KW;
    a.stride = stride; a.pad = pad; a.relu = relu; a.out_pitch = out_pitch; a.KP = KP;
    a.tiles_per_image = Ho * (Wo / BM);
    a.cout_pad = (Cout + 31) / 32 * 32;
    switch (Wo / BM) {
    case 1: return launch_stem_pool_f32s<1>(a, B, R, st);
    case 2: return launch_stem_pool_f32s<2>(a, B, R, st);
    case 3: return launch_stem_pool_f32s<3>(a, B, R, st);
    default: return launch_stem_pool_f32s<4>(a, B, R, st);
    }
}

// Returns CN_ERR_UNSUPPORTED when the shape does not fit this kernel (the caller then uses
// the generic implicit-GEMM stem).
int cn_stem_conv_f32(const float *x, const float *w_packed, const float *scale, const float *shift,
                     float *y, int B, int H, int W, int Ho, int Wo, int Cout, int KH, int KW,
                     int stride, int pad, int relu, int out_pitch, int KP, int persistent,
                     const cn_f32s_ctl *ctl, hipStream_t st)
{
    if (KP & 7) return CN_ERR_UNSUPPORTED;
    const float ctl_x_mul = (ctl && ctl->x_mul != 0.f) ? ctl->x_mul : 1.f;
    uint32_t *const ctl_range = ctl ? ctl->range : nullptr;
    if (persistent && KH == PKH && KW == PKW && Wo % BM == 0 && (stride == 1 || stride == 2) &&
        KP >= PKH * PKW * 3) {
        StemArgs a;
        a.y_f32s = 0;
        a.x_mul = ctl_x_mul; a.range = ctl_range;
        a.x = x; a.w = w_packed; a.scale = scale; a.shift = shift; a.y = y;
        a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout; a.KH = KH; a.KW = KW;
        a.stride = stride; a.pad = pad; a.relu = relu; a.out_pitch = out_pitch; a.KP = KP;
        a.tiles_per_image = Ho * (Wo / BM);
        a.cout_pad = (Cout + 31) / 32 * 32;
        // bit 1 of `persistent`: f32s arithmetic (three fp16 MFMAs per product), stride-2 stems
        // of more than 16 output channels
        if ((persistent & 2) && stride == 2 && Cout > 16)
            return Cout > 32 ? launch_stem_persist_f32s<64>(a, B, st)
                             : launch_stem_persist_f32s<32>(a, B, st);
        // ... and the stride-1 stem of <= 16 output channels (DLA base_layer)
        if ((persistent & 2) && stride == 1 && Cout <= 16 && pad == 3 && (W & 3) == 0)
            return launch_stem16s(a, B, st);
        if (Cout > 32)
            return stride == 2 ? launch_stem_persist<64, 2>(a, B, st)
                               : launch_stem_persist<64, 1>(a, B, st);
        if (Cout <= 16)
            return stride == 2 ? launch_stem_persist<16, 2>(a, B, st)
                               : launch_stem_persist<16, 1>(a, B, st);
        return stride == 2 ? launch_stem_persist<32, 2>(a, B, st)
                           : launch_stem_persist<32, 1>(a, B, st);
    }
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
    a.y_f32s = 0;
    a.x_mul = 1.f; a.range = nullptr;   // fp32 kernel: nothing is split
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
