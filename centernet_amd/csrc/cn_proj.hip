// cn_proj.hip -- 1x1 convolution (stride 1 or 2) on f32s tensors: the `downsample` projection of a
// residual block (resnet_dcn.py:179-195 `nn.Conv2d(inplanes, planes, 1, stride) + BatchNorm`, applied in
// BasicBlock.forward :52-56 / Bottleneck :99-103) and any other 1x1 layer without a residual operand.
//
// Why its own kernel.  On the implicit-GEMM kernel the three projections of resdcn_18 took 40 / 28 / 27 us at
// B = 32 (54-82 TFLOP/s): K is 64 .. 256, so a 128 x 128 tile is two to eight K steps between a prologue, two
// barriers per step and an epilogue -- the launch is all fixed cost.  The layer moves 100 / 50 / 25 MB (every
// second pixel of every second row in, all of the output out): it is bound by HBM, not by the matrix pipe.  Here
//   * a wave owns 32 output pixels x 128 output channels and never meets a barrier in its K loop: its 32 input
//     pixels of a chunk (32 x 128 bytes) come by LDS-DMA into a wave-PRIVATE strip -- whole 128-byte rows, eight
//     lanes per pixel, XOR-swizzled through the source address (piece ^ ((pixel >> 1) & 7)) so that the fragment
//     reads of lanes 128 bytes apart stay conflict-free -- all chunks of a short K requested up front;
//   * the weights' A fragments come straight from the fragment-ordered copy of the matrix (one contiguous 1 KiB
//     per wave-load, L2-resident, shared by the four waves of a workgroup through L1).  (A first form that read
//     both operands as per-lane 16-byte fragments from the row forms -- 32 cache lines per load instruction --
//     took 70 / 35 / 31 us: the texture path, not HBM, was the bound.)
//   * bias / BatchNorm as the usual (scale, shift) epilogue, then the wave's 32 x 128-byte rows go through
//     the same strip and leave as whole 128-byte groups (16-byte stores).
// f32s arithmetic: acc += Al*Bh + Ah*Bl + Ah*Bh per K half (cn_common.h).
#include "cn_common.h"

int cn_tune_proj = 1;   // cn_set_tuning key 46: 1 = f32s 1x1 layers without residual on this kernel (default), 0 = implicit GEMM

namespace {

constexpr int J_NT = 256;                 // four waves: four blocks of 32 pixels
constexpr int J_NB = 4;                   // blocks of 32 output channels per wave
constexpr int J_ROW = 144;                // bytes per staged OUTPUT pixel row (128 + 16: conflict-free 8-byte writes)
constexpr int J_STRIP = 32 * J_ROW;       // 4608 bytes per wave (epilogue; aliases the wave's input slots)
constexpr int J_SLOT = 32 * 128;          // one chunk of the wave's 32 input pixels
// J_PRE input slots per wave (chunks in flight): 2 for K <= 64 (33 KiB of LDS: the register budget, three workgroups
// per CU, is the limit), 4 otherwise (65 KiB: two workgroups per CU)
static_assert(J_STRIP <= 2 * J_SLOT, "the epilogue strip aliases the input slots");

struct JArgs {
    const char *x;            // f32s NHWC input
    const char *w;            // f32s fragment-ordered weights [chunk][block][quarter][lane] x 16 bytes (behind the row form)
    int ncb;                  // 32-channel blocks of the padded Cout
    const float *scale, *shift;
    char *y;
    int M, Ho, Wo, H, W, stride;
    int in_pitchB, out_pitchB, cout_pad, nchunk, Cout, relu;
    uint32_t *range;
};

typedef _Float16 j_f16x8 __attribute__((ext_vector_type(8)));

typedef __attribute__((address_space(3))) void j_lds_void;
typedef __attribute__((address_space(1))) const void j_glb_void;

template <bool OUT_PLAIN, int J_PRE>
__global__ __launch_bounds__(J_NT, 2) void proj1x1_kernel(const JArgs a)
{
    constexpr int J_WAVE = J_PRE * J_SLOT;    // bytes per wave
    constexpr int J_SS = 4 * J_WAVE;          // float [2][128]: scale, shift of the workgroup's channels
    extern __shared__ __attribute__((aligned(128))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int n0 = (int)blockIdx.y * (32 * J_NB);
    // bias / BN of the workgroup's 128 channels
    {
        const int which = tid >> 7, c = tid & 127;
        const float *src = which ? a.shift : a.scale;
        float v = which ? 0.f : 1.f;
        if (src && n0 + c < a.Cout) v = src[n0 + c];
        reinterpret_cast<float *>(smem + J_SS)[tid] = v;
    }
    // ---- the wave's 32 input pixels by LDS-DMA: instruction q moves pixels 8 q .. 8 q + 7, lane j = (pixel j >> 3,
    // physical 16-byte piece j & 7), which receives the pixel's logical piece (j & 7) ^ ((pixel >> 1) & 7)
    const int m_base = (int)blockIdx.x * 128 + wave * 32;
    unsigned src_off[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int p = 8 * q + (lane >> 3);
        const int mc = min(m_base + p, a.M - 1);
        const int ox = mc % a.Wo, t1 = mc / a.Wo, oy = t1 % a.Ho, b = t1 / a.Ho;
        const unsigned pin = (unsigned)((b * a.H + oy * a.stride) * a.W + ox * a.stride);
        src_off[q] = pin * (unsigned)a.in_pitchB + (unsigned)(((lane & 7) ^ ((p >> 1) & 7)) * 16);
    }
    char *wbase = smem + wave * J_WAVE;
    auto dma = [&](int slot, int c) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_global_load_lds((j_glb_void *)(a.x + src_off[q] + (unsigned)c * 128u),
                                             (j_lds_void *)(wbase + slot * J_SLOT + q * 1024), 16, 0, 0);
    };
#pragma unroll
    for (int c = 0; c < J_PRE; ++c)
        if (c < a.nchunk) dma(c, c);
    // fragment reads: pixel l31's row, logical piece 2 kh + h (high) / 4 + 2 kh + h (low), swizzled
    const int sw = (l31 >> 1) & 7;
    const int rd = l31 * 128;
    // A fragments: block (n0 / 32 + nb) of chunk c, quarter kh (high) / 2 + kh (low): wfrag + ((c * ncb + block) * 4 + quarter) * 1024
    const char *wl = a.w + (size_t)(n0 >> 5) * 4096 + (unsigned)lane * 16u;
    const size_t chunkB = (size_t)a.ncb * 4096;
    const int nbv = min(J_NB, a.ncb - (n0 >> 5));       // blocks of this workgroup that exist (uniform)

    cn_f32x16 acc[J_NB];
#pragma unroll
    for (int nb = 0; nb < J_NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
    __syncthreads();   // scale / shift visible (the only barrier)

    for (int c = 0; c < a.nchunk; ++c) {
        const int slot = c & (J_PRE - 1);
        j_f16x8 af[2][4];
        auto load_a = [&](int set, int nb) {
            const char *g = wl + (size_t)c * chunkB + (size_t)min(nb, nbv - 1) * 4096;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) af[set][qd] = *reinterpret_cast<const j_f16x8 *>(g + qd * 1024);
        };
        load_a(0, 0);
        // the chunk's DMA pieces are this wave's own: wait until at most the younger chunks' are outstanding
        // (vmcnt counts in order; the A loads just issued are younger too -- drain to the number that may remain)
        {
            const int younger = min(a.nchunk - 1 - c, J_PRE - 1);     // chunks requested behind this one
            if (J_PRE > 2 && younger >= 3) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");       // 3 x 4 DMA + 4 A loads
            else if (J_PRE > 2 && younger == 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
        j_f16x8 bf[4];   // [kh 0 high, kh 1 high, kh 0 low, kh 1 low]
        {
            const char *sp = wbase + slot * J_SLOT + rd;
            bf[0] = *reinterpret_cast<const j_f16x8 *>(sp + (((0 + h) ^ sw) << 4));
            bf[1] = *reinterpret_cast<const j_f16x8 *>(sp + (((2 + h) ^ sw) << 4));
            bf[2] = *reinterpret_cast<const j_f16x8 *>(sp + (((4 + h) ^ sw) << 4));
            bf[3] = *reinterpret_cast<const j_f16x8 *>(sp + (((6 + h) ^ sw) << 4));
        }
#pragma unroll
        for (int nb = 0; nb < J_NB; ++nb) {
            const int set = nb & 1;
            if (nb + 1 < J_NB) load_a(set ^ 1, nb + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[set][2 + kh], bf[kh], acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[set][kh], bf[2 + kh], acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[set][kh], bf[kh], acc[nb], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // this slot's next chunk: its fragment reads are complete (the MFMAs above consumed them)
        if (c + J_PRE < a.nchunk) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            dma(slot, c + J_PRE);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

    // ---- epilogue: acc[nb][4 g + e] = channel n0 + 32 nb + 8 g + 4 h + e of pixel l31.  y = relu?(acc * scale + shift),
    // one 32-channel block at a time through the wave's strip (LDS serves a wave's accesses in order)
    char *strip = smem + wave * J_WAVE;
    const float *ss = reinterpret_cast<const float *>(smem + J_SS);
    float rng_out = 0.f;
    const int rrow = lane >> 3, rcol = lane & 7;
    const float floor_v = a.relu ? 0.f : -__builtin_inff();
#pragma unroll
    for (int nb = 0; nb < J_NB; ++nb) {
        if (n0 + 32 * nb >= a.cout_pad) break;          // uniform
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const cn_f32x4 sc = *reinterpret_cast<const cn_f32x4 *>(ss + 32 * nb + 8 * g + 4 * h);
            const cn_f32x4 sh = *reinterpret_cast<const cn_f32x4 *>(ss + 128 + 32 * nb + 8 * g + 4 * h);
            cn_f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[nb][4 * g + e] * sc[e] + sh[e], floor_v);
            if (OUT_PLAIN) {
                *reinterpret_cast<cn_f32x4 *>(strip + l31 * J_ROW + (8 * g + 4 * h) * 4) = v;
            } else {
                cn_rng_upd4(rng_out, v);
                cn_f16x4v h4, l4;
                cn_split4(v, h4, l4);
                *reinterpret_cast<cn_f16x4v *>(strip + l31 * J_ROW + (8 * g + 4 * h) * 2) = h4;
                *reinterpret_cast<cn_f16x4v *>(strip + l31 * J_ROW + 64 + (8 * g + 4 * h) * 2) = l4;
            }
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int row = 8 * k + rrow;
            const cn_f32x4 v = *reinterpret_cast<const cn_f32x4 *>(strip + row * J_ROW + rcol * 16);
            const int mo = (int)blockIdx.x * 128 + wave * 32 + row;
            if (mo < a.M)
                *reinterpret_cast<cn_f32x4 *>(a.y + (size_t)mo * a.out_pitchB + (size_t)(n0 + 32 * nb) * 4 + rcol * 16) = v;
        }
        asm volatile("" ::: "memory");
    }
    if (!OUT_PLAIN) cn_rng_commit(a.range, 0, rng_out);
}

}  // namespace

// Shapes this kernel takes: f32s NHWC input (pitch = whole 32-channel groups), 1x1, stride 1 or 2, no padding,
// no residual; f32s or plain NHWC output whose pitch covers whole 32-channel groups of the padded Cout.
bool cn_proj1x1_takes(int B, int H, int W, int Cin, int Cout, int stride, int in_pitch, int out_pitch)
{
    if (!cn_tune_proj) return false;
    if ((Cin & 31) || (in_pitch & 31) || (out_pitch & 31) || (stride != 1 && stride != 2)) return false;
    if (out_pitch < (Cout + 31) / 32 * 32) return false;
    if ((size_t)B * H * W * in_pitch * 4 >= ((size_t)1 << 32)) return false;
    return Cin <= 1024 && Cout >= 32;
}

int cn_proj1x1_f32s(const void *x, const void *w_packed, const float *scale, const float *shift, void *y,
                    int B, int H, int W, int Cin, int Cout, int stride, int in_pitch, int out_pitch, int relu,
                    int out_plain, const cn_f32s_ctl *ctl, hipStream_t st)
{
    JArgs a = {};
    // the fragment-ordered copy sits behind the row form ([1][cout_pad][cin_pad] x 4 bytes)
    a.x = (const char *)x; a.w = (const char *)w_packed + (size_t)((Cout + 31) / 32 * 32) * Cin * 4; a.scale = scale; a.shift = shift; a.y = (char *)y;
    a.H = H; a.W = W; a.stride = stride;
    a.Ho = (H - 1) / stride + 1; a.Wo = (W - 1) / stride + 1;
    a.M = B * a.Ho * a.Wo;
    a.in_pitchB = in_pitch * 4; a.out_pitchB = out_pitch * 4;
    a.cout_pad = (Cout + 31) / 32 * 32;
    a.ncb = a.cout_pad / 32;
    a.nchunk = Cin / 32;
    a.Cout = Cout; a.relu = relu;
    a.range = ctl ? ctl->range : nullptr;
    const dim3 grid((unsigned)cn_cdiv(a.M, 128), (unsigned)cn_cdiv(a.cout_pad, 32 * J_NB));
    const int pre = a.nchunk <= 2 ? 2 : 4;
    const int lds = 4 * pre * J_SLOT + 2 * 128 * 4;
    if (out_plain && pre == 2) {
        CN_SET_MAX_LDS_ONCE((proj1x1_kernel<true, 2>), 4 * 2 * J_SLOT + 1024);
        hipLaunchKernelGGL((proj1x1_kernel<true, 2>), grid, dim3(J_NT), lds, st, a);
    } else if (out_plain) {
        CN_SET_MAX_LDS_ONCE((proj1x1_kernel<true, 4>), 4 * 4 * J_SLOT + 1024);
        hipLaunchKernelGGL((proj1x1_kernel<true, 4>), grid, dim3(J_NT), lds, st, a);
    } else if (pre == 2) {
        CN_SET_MAX_LDS_ONCE((proj1x1_kernel<false, 2>), 4 * 2 * J_SLOT + 1024);
        hipLaunchKernelGGL((proj1x1_kernel<false, 2>), grid, dim3(J_NT), lds, st, a);
    } else {
        CN_SET_MAX_LDS_ONCE((proj1x1_kernel<false, 4>), 4 * 4 * J_SLOT + 1024);
        hipLaunchKernelGGL((proj1x1_kernel<false, 4>), grid, dim3(J_NT), lds, st, a);
    }
    CN_CHECK_LAUNCH();
    return CN_OK;
}
