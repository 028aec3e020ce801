// cn_dcn.hip -- fused modulated deformable convolution (DCNv2) forward with LDS-staged
// input tiles.
//
// Replaces: DCN.forward -> DCNv2Function.forward -> dcn_v2_cuda_forward
//   (DCNv2/dcn_v2.py:64-70, dcn_v2_func.py:22-38, src/dcn_v2_cuda.c:10-102): per sample, a
//   bias SGEMM, modulated_deformable_im2col_gpu_kernel (src/cuda/dcn_v2_im2col_cuda.cu:118-180,
//   bilinear sampler :18-47) writing a Cin*9*HW column buffer, and the main SGEMM.
//
// One workgroup = an 8x8 block of output pixels of one image x up to 128 output channels.
//   * Per 32-channel chunk the (8 + 2 + 2R) x (8 + 2 + 2R) input WINDOW around the block
//     (R = 4 pixels of offset reach) is copied once into LDS with 128-byte row reads
//     (NHWC: one pixel's 32 channels are one line, so the copy is coalesced whatever the
//     offsets are).
//   * For each of the 9 taps, every thread forms (w1*v1 + w2*v2 + w3*v3 + w4*v4) * mask for
//     its (pixel, 4 channels) from the window -- four ds_read_b128 instead of four global
//     gathers -- and writes it into the LDS A tile of the MFMA contraction.  A corner that
//     falls outside the window (offset beyond R) is fetched from global memory instead, so
//     unbounded offsets (dcn_v2.py:65-67: raw conv output) stay exact.
//   * Sampling records (clamped corner coordinates, corner weights zeroed by the reference's
//     corner rule :30-41 and window rule :165, mask with the sigmoid of dcn_v2.py:67 fused)
//     are computed once per (pixel, tap), two taps ahead of their use (three LDS buffers).
//   * Weight tile per (tap, chunk) through LDS with register prefetch; fp32 MFMA
//     (v_mfma_f32_32x32x2_f32); epilogue y = relu?((acc + bias) * scale + shift), i.e. the
//     BatchNorm + ReLU that follows every DCN in CenterNet (resnet_dcn.py:237-239,
//     pose_dla_dcn.py:345-357).  No column buffer, no per-sample host loop.
#include "cn_common.h"

namespace {

constexpr int NT = 256;
constexpr int LDT = 36;            // floats per LDS row (128 B data + 16 B pad)
constexpr int TS = 8;              // output tile is TS x TS pixels
constexpr int BM = TS * TS;        // 64
constexpr int REACH = 4;           // offset reach covered by the LDS window
constexpr int WD = TS + 2 + 2 * REACH;  // window side (18)
constexpr int WPIX = WD * WD;           // 324 pixels
constexpr int NPW = (WPIX + 31) / 32;   // window load passes per thread (11)

struct DcnArgs {
    const float *x, *w, *bias, *scale, *shift, *om;
    float *y;
    int B, H, W, Cin, Cout, om_pitch, mask_sigmoid, relu;
    int cin_pad, cout_pad, nchunk, tiles_x, tiles_y, vec_out, setprio;
};

__device__ __forceinline__ float dcn_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

template <int BN>
__global__ __launch_bounds__(NT) void dcn_window_kernel(const DcnArgs a)
{
    constexpr int WM = 2, WN = 2;
    constexpr int TM = BM / WM, TN = BN / WN;  // 32 x BN/2 per wave
    constexpr int NB = TN / 32;
    constexpr int PB = BN / 32;
    static_assert(TN % 32 == 0, "wave tile");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *Win = reinterpret_cast<float *>(smem);       // [WPIX][LDT]
    float *As = Win + WPIX * LDT;                        // [BM][LDT]
    float *Bs = As + BM * LDT;                           // [BN][LDT]
    // records of tap t live in buffer t % 3: the writer (two taps ahead) never touches the
    // buffers of the tap being read or the next one, also across the 8 -> 0 wrap
    int *rec_i = reinterpret_cast<int *>(Bs + BN * LDT); // [3][BM][2]  packed clamped corners
    float *rec_w = reinterpret_cast<float *>(rec_i + 3 * BM * 2);  // [3][BM][4] corner weights
    float *rec_m = rec_w + 3 * BM * 4;                   // [3][BM] mask
    int *rowoff = reinterpret_cast<int *>(rec_m + 3 * BM);  // [BM]
    // epilogue staging reuses As + Bs: TM x (BN + 4) floats <= (BM + BN) * LDT

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = tid >> 3, q = tid & 7;
    const int tiles = a.tiles_x * a.tiles_y;
    const int b = blockIdx.x / tiles;
    const int tr = blockIdx.x - b * tiles;
    const int ty0 = (tr / a.tiles_x) * TS, tx0 = (tr % a.tiles_x) * TS;
    const int wy0 = ty0 - 1 - REACH, wx0 = tx0 - 1 - REACH;  // window origin in the image
    const int n0 = blockIdx.y * BN;
    const int H = a.H, W = a.W;
    const cn_f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // ---- window pixels handled by this thread: image pixel offset or -1
    int woff[NPW];
#pragma unroll
    for (int p = 0; p < NPW; ++p) {
        const int wp = p * 32 + lrow;
        const int wy = wp / WD, wx = wp - wy * WD;
        const int iy = wy0 + wy, ix = wx0 + wx;
        woff[p] = (wp < WPIX && iy >= 0 && iy < H && ix >= 0 && ix < W) ? (b * H + iy) * W + ix : -1;
    }
    for (int m = tid; m < BM; m += NT) {
        const int oy = ty0 + m / TS, ox = tx0 + m % TS;
        rowoff[m] = (oy < H && ox < W) ? (b * H + oy) * W + ox : -1;
    }

    // dcn_v2_im2col_cuda.cu:151-176 and :18-47, evaluated once per (pixel, tap)
    auto records = [&](int tap) {
        const int pb = tap % 3;
        if (tid < BM) {
            const int m = tid;
            const int oy = ty0 + m / TS, ox = tx0 + m % TS;
            int cy = 0, cx = 0;
            float w1 = 0.f, w2 = 0.f, w3 = 0.f, w4 = 0.f, mk = 0.f;
            if (oy < H && ox < W) {
                const float *om = a.om + (size_t)((b * H + oy) * W + ox) * a.om_pitch;
                const float off_h = om[2 * tap];
                const float off_w = om[2 * tap + 1];
                mk = om[18 + tap];
                if (a.mask_sigmoid) mk = dcn_sigmoid(mk);  // dcn_v2.py:67
                const int ki = tap / 3, kj = tap - ki * 3;
                const float h_im = (float)(oy - 1 + ki) + off_h;
                const float w_im = (float)(ox - 1 + kj) + off_w;
                if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
                    const float hf = floorf(h_im), wf = floorf(w_im);
                    const int h_low = (int)hf, w_low = (int)wf;
                    const int h_high = h_low + 1, w_high = w_low + 1;
                    const float lhh = h_im - hf, lww = w_im - wf;
                    const float hh = 1.f - lhh, hw = 1.f - lww;
                    const bool hl_ok = h_low >= 0, wl_ok = w_low >= 0;
                    const bool hh_ok = h_high <= H - 1, wh_ok = w_high <= W - 1;
                    w1 = (hl_ok && wl_ok) ? hh * hw : 0.f;
                    w2 = (hl_ok && wh_ok) ? hh * lww : 0.f;
                    w3 = (hh_ok && wl_ok) ? lhh * hw : 0.f;
                    w4 = (hh_ok && wh_ok) ? lhh * lww : 0.f;
                    const int yl = max(h_low, 0), yh = min(h_high, H - 1);
                    const int xl = max(w_low, 0), xh = min(w_high, W - 1);
                    cy = yl | (yh << 16);
                    cx = xl | (xh << 16);
                }
            }
            rec_i[(pb * BM + m) * 2 + 0] = cy;
            rec_i[(pb * BM + m) * 2 + 1] = cx;
            float *sw = rec_w + (pb * BM + m) * 4;
            sw[0] = w1; sw[1] = w2; sw[2] = w3; sw[3] = w4;
            rec_m[pb * BM + m] = mk;
        }
    };

    cn_f32x16 acc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    cn_f32x4 rw[NPW], rb[PB];
    auto load_win = [&](int chunk) {
        const int c = chunk * 32 + 4 * q;
#pragma unroll
        for (int p = 0; p < NPW; ++p) {
            const bool ok = woff[p] >= 0 && c < a.Cin;
            const cn_f32x4 v = *reinterpret_cast<const cn_f32x4 *>(
                a.x + (ok ? ((size_t)woff[p] * a.Cin + c) : 0));
            rw[p] = ok ? v : zero4;
        }
    };
    auto store_win = [&]() {
#pragma unroll
        for (int p = 0; p < NPW; ++p) {
            const int wp = p * 32 + lrow;
            if (wp < WPIX) *reinterpret_cast<cn_f32x4 *>(Win + wp * LDT + 4 * q) = rw[p];
        }
    };
    auto load_B = [&](int chunk, int tap) {
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const int n = min(n0 + p * 32 + lrow, a.cout_pad - 1);
            rb[p] = *reinterpret_cast<const cn_f32x4 *>(
                a.w + ((size_t)(tap * a.cout_pad + n) * a.cin_pad + chunk * 32 + 4 * q));
        }
    };
    auto store_B = [&]() {
#pragma unroll
        for (int p = 0; p < PB; ++p)
            *reinterpret_cast<cn_f32x4 *>(Bs + (p * 32 + lrow) * LDT + 4 * q) = rb[p];
    };
    // one corner: from the LDS window when it lies inside, else from global memory
    auto corner = [&](int y, int x, int c, bool cok) -> cn_f32x4 {
        const int wy = y - wy0, wx = x - wx0;
        if ((unsigned)wy < (unsigned)WD && (unsigned)wx < (unsigned)WD)
            return *reinterpret_cast<const cn_f32x4 *>(Win + (wy * WD + wx) * LDT + 4 * q);
        const cn_f32x4 v = *reinterpret_cast<const cn_f32x4 *>(
            a.x + ((size_t)((b * H + y) * W + x) * a.Cin + (cok ? c : 0)));
        return cok ? v : zero4;
    };
    // A tile of one tap: (w1*v1 + w2*v2 + w3*v3 + w4*v4) * mask  (dcn_v2_im2col_cuda.cu:43-45,174)
    auto build_A = [&](int chunk, int tap) {
        const int pb = tap % 3;
        const int c = chunk * 32 + 4 * q;
        const bool cok = c < a.Cin;
#pragma unroll
        for (int p = 0; p < BM / 32; ++p) {
            const int r = p * 32 + lrow;
            const int cy = rec_i[(pb * BM + r) * 2 + 0], cx = rec_i[(pb * BM + r) * 2 + 1];
            const float *wt = rec_w + (pb * BM + r) * 4;
            const float mk = rec_m[pb * BM + r];
            const int yl = cy & 0xffff, yh = cy >> 16, xl = cx & 0xffff, xh = cx >> 16;
            const cn_f32x4 v1 = corner(yl, xl, c, cok), v2 = corner(yl, xh, c, cok);
            const cn_f32x4 v3 = corner(yh, xl, c, cok), v4 = corner(yh, xh, c, cok);
            cn_f32x4 v = v1 * wt[0] + v2 * wt[1] + v3 * wt[2] + v4 * wt[3];
            v = v * mk;
            *reinterpret_cast<cn_f32x4 *>(As + r * LDT + 4 * q) = v;
        }
    };
    auto compute = [&]() {
        const float *Ab = As + (wm * TM + l31) * LDT + 4 * lh;
        const float *Bb = Bs + (wn * TN + l31) * LDT + 4 * lh;
        if (a.setprio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const cn_f32x4 af = *reinterpret_cast<const cn_f32x4 *>(Ab + kk * 8);
            cn_f32x4 bf[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j)
                bf[j] = *reinterpret_cast<const cn_f32x4 *>(Bb + j * 32 * LDT + kk * 8);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], bf[j][s], acc[j], 0, 0, 0);
        }
        if (a.setprio) __builtin_amdgcn_s_setprio(0);
    };

    // ---- prologue
    load_win(0);
    load_B(0, 0);
    records(0);
    records(1);
    store_win();
    __syncthreads();

    // ---- main loop: chunk-major, 9 taps inner, two barriers per tap
    for (int c = 0; c < a.nchunk; ++c) {
#pragma unroll 1
        for (int t = 0; t < 9; ++t) {
            const bool last = (c + 1 == a.nchunk) && (t == 8);
            const bool new_win = (t == 8) && (c + 1 < a.nchunk);
            build_A(c, t);   // LDS window -> LDS A tile (reads records of tap t)
            store_B();       // weight tile prefetched during the previous tap
            __syncthreads();
            if (!last) load_B(t == 8 ? c + 1 : c, t == 8 ? 0 : t + 1);
            if (new_win) load_win(c + 1);
            compute();
            // records of tap t+2 (mod 9): they do not depend on the chunk, so after the first
            // chunk they are simply rewritten with the same values
            records((t + 2) % 9);
            __syncthreads();
            if (new_win) {
                store_win();
                __syncthreads();
            }
        }
    }

    // ---- epilogue: stage one wave-row (32 pixels) of the tile, 16-byte stores along Cout
    constexpr int LDC = BN + 4;
    float *Cs = As;
    constexpr int C4 = BN / 4;
    constexpr int RPI = NT / C4;
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
#pragma unroll 1
    for (int pass = 0; pass < WM; ++pass) {
        if (pass) __syncthreads();
        if (wm == pass) {
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                    Cs[row * LDC + wn * TN + j * 32 + l31] = acc[j][r];
                }
        }
        __syncthreads();
        const int rbase = pass * TM;
#pragma unroll
        for (int k = 0; k < ITERS; ++k) {
            const int lr = k * RPI + r0;
            if (lr >= TM) continue;
            const int off = rowoff[rbase + lr];
            if (off < 0 || n >= a.Cout) continue;
            if (vec) {
                cn_f32x4 v = *reinterpret_cast<const cn_f32x4 *>(Cs + lr * LDC + c4 * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = (v[e] + bs[e]) * sc[e] + sf[e];
                    v[e] = a.relu ? fmaxf(t, 0.f) : t;
                }
                *reinterpret_cast<cn_f32x4 *>(a.y + (size_t)off * a.Cout + n) = v;
            } else {
                for (int e = 0; e < 4 && (n + e) < a.Cout; ++e) {
                    const float t = (Cs[lr * LDC + c4 * 4 + e] + bs[e]) * sc[e] + sf[e];
                    a.y[(size_t)off * a.Cout + n + e] = a.relu ? fmaxf(t, 0.f) : t;
                }
            }
        }
    }
}

template <int BN>
int launch_dcn(const DcnArgs &a, hipStream_t st)
{
    constexpr size_t lds = (size_t)(WPIX * LDT + BM * LDT + BN * LDT) * 4 +
                           (size_t)3 * BM * (2 + 4 + 1) * 4 + BM * 4;
    static_assert((size_t)(BM / 2) * (BN + 4) <= (size_t)(BM + BN) * LDT, "epilogue staging fits");
    CN_SET_MAX_LDS_ONCE(dcn_window_kernel<BN>, lds);
    dim3 grid((unsigned)(a.B * a.tiles_x * a.tiles_y), cn_cdiv(a.Cout, BN));
    hipLaunchKernelGGL(dcn_window_kernel<BN>, grid, dim3(NT), lds, st, a);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

}  // namespace

// Returns CN_ERR_UNSUPPORTED for shapes this kernel does not take (the caller falls back to
// the global-gather implicit-GEMM form in cn_conv.hip).
int cn_dcn_window_f32(const float *x, const float *w_packed, const float *bias, const float *om,
                      int om_pitch, const float *scale, const float *shift, float *y, int B, int Cin,
                      int H, int W, int Cout, int mask_sigmoid, int relu, int setprio, hipStream_t st)
{
    if (H >= 65536 || W >= 65536) return CN_ERR_UNSUPPORTED;  // corners are packed in 16 bits
    DcnArgs a = {};
    a.x = x; a.w = w_packed; a.bias = bias; a.scale = scale; a.shift = shift; a.om = om; a.y = y;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.om_pitch = om_pitch;
    a.mask_sigmoid = mask_sigmoid; a.relu = relu; a.setprio = setprio;
    a.cin_pad = (Cin + 31) / 32 * 32;
    a.cout_pad = (Cout + 31) / 32 * 32;
    a.nchunk = a.cin_pad / 32;
    a.tiles_x = cn_cdiv(W, TS);
    a.tiles_y = cn_cdiv(H, TS);
    a.vec_out = ((Cout & 3) == 0 && cn_aligned16(y)) ? 1 : 0;
    if (Cout > 64) return launch_dcn<128>(a, st);
    return launch_dcn<64>(a, st);
}
