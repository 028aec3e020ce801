/*
 * centernet_amd.h -- C ABI of libcenternet_amd.so (gfx950 / MI355X only).
 *
 * This is the drop-in boundary for the CenterNet inference hot path
 * (SURVEY.md section 8b).  Every entry point:
 *   - takes plain device pointers, sizes and a hipStream_t passed as void*;
 *   - never allocates, frees or synchronises: the caller owns every buffer,
 *     including the workspace, and the call is stream-ordered and asynchronous;
 *   - returns CN_OK (0) or a negative cn_status; nothing longjmps or prints
 *     (the reference raises THError, DCNv2/src/dcn_v2_cuda.c:33-38, and only
 *     printf()s launch errors, dcn_v2_im2col_cuda.cu:331-335);
 *   - is thread-safe per stream (no hidden global scratch, unlike the
 *     per-Function `ones`/`columns` buffers of DCNv2/dcn_v2_func.py:28).
 *
 * All tensors are fp32.  "NCHW" is the reference's layout; "NHWC" is the
 * native layout of this library (channel vectors contiguous so that the four
 * bilinear taps of the deformable gather are 128-byte coalesced reads).
 *
 * Citations are relative to /root/reference.
 */
#ifndef CENTERNET_AMD_H
#define CENTERNET_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum cn_status {
    CN_OK = 0,
    CN_ERR_SHAPE = -1,     /* shapes / kernel sizes do not match (THError in the reference) */
    CN_ERR_UNSUPPORTED = -2,
    CN_ERR_WORKSPACE = -3, /* workspace too small; query with the *_workspace_bytes call */
    CN_ERR_LAUNCH = -4,    /* hipGetLastError() != hipSuccess after a launch */
    CN_ERR_NULL = -5,
    CN_ERR_ALIGN = -6      /* a pointer is not 16-byte aligned */
} cn_status;

#define CN_LAYOUT_NCHW 0
#define CN_LAYOUT_NHWC 1

/* Element type of activations / packed weights.  fp32 is the parity path; fp16 (fp32
 * accumulate, v_mfma_f32_32x32x16_f16) exists for BASELINE configs[4] (Hourglass-104 fp16),
 * a surface the reference does not have. */
#define CN_DTYPE_F32 0
#define CN_DTYPE_F16 1
/* fp32 values stored as fp16 (high, low) pairs, 32-channel groups of 128 bytes; computed
 * with three fp16 MFMAs per product at fp32-level accuracy (csrc/cn_common.h) */
#define CN_DTYPE_F32S 2

/* f32s range control (csrc/cn_common.h "Range").  An f32s tensor holds  stored = real * 2^-e  with a
 * per-tensor exponent e owned by the caller (centernet_amd/engine.py picks it so that the tensor's
 * largest magnitude sits near 2^9; the fp16 pair then keeps its 22 bits over 12 binades below
 * and has 2^6 of head-room above).  The kernels never see e: its effect is folded by the caller
 * into the per-channel epilogue `scale` / `shift` they apply anyway, and into the two multipliers
 * below (powers of two, hence exact).  `range`, when not NULL, points to CN_RANGE_WORDS device
 * words the caller zeroed: the launch max-es into them, as float bit patterns, the largest
 * |value| it split -- side 0 = the output side (the stored tensor / the hidden tile of the fused
 * heads), side 1 = the input side (a plain x, the blended deformable samples); word
 * [(side * CN_RANGE_SLOTS + slot) * CN_RANGE_STRIDE], the maximum over the slots is the
 * launch's (cn_range_fold reduces them).  A maximum > 65504 means a
 * value was clamped: the result is invalid and the caller must re-scale (the reference's plain
 * fp32 never saturates; neither does this library silently).  NULL ctl = all defaults. */
#define CN_RANGE_SLOTS 64   /* words per side a launch spreads its per-wave maxima over ... */
#define CN_RANGE_STRIDE 16  /* ... one per 64-byte line (atomics on one address serialise) */
#define CN_RANGE_WORDS (2 * CN_RANGE_SLOTS * CN_RANGE_STRIDE)  /* words of one launch's `range` */
typedef struct cn_f32s_ctl {
    float x_mul;     /* plain fp32 x (CN_CONV_X_PLAIN, deformable input, stem image): multiplied by
                        this before it is split; 0 = 1 */
    float res_mul;   /* the residual is multiplied by this before it is added; 0 = 1 */
    uint32_t *range; /* device, CN_RANGE_WORDS words, or NULL */
} cn_f32s_ctl;

/* Library / ABI version (major*10000 + minor*100 + patch). */
int cn_version(void);
/* Human-readable text for a cn_status. */
const char *cn_status_string(int status);
/* Architecture the device code was compiled for ("gfx950"). */
const char *cn_arch(void);
/* Kernel-selection knobs for benchmarking (not needed for correctness).
 * THREADING: the knobs are plain process-global integers read by every launch.  cn_set_tuning is NOT
 * thread-safe against concurrent launches: the "thread-safe per stream" rule of the entry points holds only
 * while nobody calls cn_set_tuning.  Set knobs once, before the first launch (as bench.py --tune does), or
 * with all streams of the process idle; a value changed under a running launch list gives a mix of forms
 * (every form computes the same function, so results stay valid, timings do not).
 * key 1: LDS tile buffers of the dense implicit-GEMM kernels, 0 = default, 1 or 2.
 * key 2: 1 = never pick 64-wide N tiles for Cout > 64 (default 0 = pick them when they
 *        avoid a half-empty 128-wide tile).
 * key 3: (retired) pixel tile of the deformable kernel; 64-pixel tiles are the only form built.
 * key 4: pixel tile of the dense kernels for Cout > 64, 0 = default, 64 or 128.
 * key 5: 1 = never split K.
 * key 6: 1 = run the 3-channel stem on the generic implicit-GEMM kernel instead of the
 *        LDS-window kernel (cn_stem.hip).
 * key 8: s_setprio(1) around the MFMA clusters (default 1).  key 9: ablation only.
 * key 10: 1 = run 3x3/stride-1 layers on the generic implicit GEMM instead of the LDS-halo
 *         kernel (cn_conv3x3.hip).
 * key 11: 1 = LDS-window deformable kernel (cn_dcn.hip) instead of the default global-gather
 *         form (cn_conv.hip); kept for A/B: it measured 20-30 % slower.
 * key 7: XCD-aware tile order: 0 = deformable kernel only (default, +5-10 % there), 1 = also
 *        the dense implicit-GEMM kernels (no gain measured), 2 = nowhere.
 * key 19: 64-wide LDS-halo tiles at four workgroups per CU (single-buffered weight tile,
 *         <= 128 registers): 0 = when that needs fewer dispatch rounds (default), 1 = always,
 *         2 = never.
 * key 18: phase shift of the workgroups that share a CU in the LDS-halo kernel, in percent of
 *         one tile's MFMA time (default 100, 0 = off); applied to launches of >= 4 dispatch rounds.
 * key 16 / 17: split-K: least K chunks per slice (default 8) / most slices (default 16).
 * key 15: fp32 / fp16 kernels: 0 = 4-wave instead of 8-wave workgroups for the 128-wide tiles of
 *         the LDS-halo kernel (default 1: +1 %, measured).
 * key 14: 64-wide layers in the LDS-halo kernel as 256-pixel tiles: 1 = four waves, 2 / 3 = eight
 *         waves (f32s; one / three taps of weight prefetch); default 0 (measured, no gain).
 * key 21: f32s LDS-halo kernel (A/B switches): bit 0 = 128-wide tiles as eight waves of 32 x 64
 *         with weights three taps ahead instead of four waves of 64 x 64 two taps ahead
 *         (default 0: +1 % on resdcn_18); bit 1 = every 64-wide layer two taps ahead (default:
 *         one tap for two-chunk layers, three for longer K; measured); bit 2 = 8 x 16 pixel tiles
 *         on wide maps too (no difference, measured).
 * key 13: tap split of the deformable kernel, 0 = auto, 1 = never, 3 or 9 = force.
 * key 20: f32s LDS-halo kernel, 128-wide tiles: 1 = per-tap weight tile in LDS (default),
 *         0 = weights streamed into registers from the fragment-ordered copy (+5-20 % in a
 *         back-to-back micro-benchmark, no gain inside the network: measured).
 * key 12: 0 = one-tile-per-workgroup stem kernel instead of the persistent, prefetching one
 *         (default 1; both in cn_stem.hip).
 * key 22: deformable kernel: 1 = a tile is an 8-wide BLOCK of pixels (8 x 8 / 8 x 16) when the map
 *         divides into them (default: the nine taps of a block sample a compact neighbourhood that
 *         stays in L1 / L2), 0 = BM consecutive pixels of a row.
 * key 23: f32s deformable kernel: 0 = the register-sampling LDS-window form (cn_dcn2.hip: every
 *         lane samples its own MFMA operand from a window of the input in LDS) for the shapes it
 *         takes when the grid has >= 192 workgroups (default), 1 = the global-gather form always,
 *         2 = the register-sampling form for every shape it takes (tests), 3 = the earlier
 *         wave-specialised window form for every shape it takes (kept for comparison: slower).
 * key 27: 7x7 / stride 1 stem of <= 16 output channels with CN_CONV_STEM_F32S: 1 = f32s kernel
 *         (default), 0 = the fp32 16x16x4 kernel (then cn_stem_f32s_supported answers 0 for it).
 * key 26: fused f32s heads with a hidden layer wider than 64: 1 = hidden layer kept in registers
 *         (default), 0 = staged through LDS; 2 = 128-wide slices (register-bound, A/B only),
 *         3 = the register form for 64-wide hidden layers too (slower there, A/B only).
 * key 24: fused heads: 1 = 1-D grid with the heads of a pixel tile dispatched together on one XCD
 *         (default), 0 = one grid row per head.
 * key 23 (round 5 / 6 values): 4 / 5 = the team form (cn_dcn3.hip) in T / N mode, 6 / 7 = the wide form
 *         (cn_dcn4.hip: a workgroup owns ALL output channels of its tile; 7 = four blocks per workgroup)
 *         for every shape they take.
 * key 41: 1 = layers with Cout % 128 == 0 take the wide deformable form (default), 0 = the team form.
 * key 42: wide form: K split until a launch has this many workgroups (default 256).
 * key 43 / 44: stem + max-pool kernel: probe switches of its instrumented instantiation / start delay of
 *         the second resident workgroup (measurement only; defaults 0). */
int cn_set_tuning(int key, int value);



/* ------------------------------------------------------------------------
 * Deformable convolution v2, forward.
 *
 * Replaces: void dcn_v2_cuda_forward(THCudaTensor *input, *weight, *bias,
 *   *ones, *offset, *mask, *output, *columns, int kernel_h, int kernel_w,
 *   stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w,
 *   deformable_group)          -- DCNv2/src/dcn_v2_cuda.h:9-17,
 *                                 impl DCNv2/src/dcn_v2_cuda.c:10-102,
 *                                 called from DCNv2/dcn_v2_func.py:29-37.
 * Differences at the boundary: no `ones`/`columns` scratch (the column buffer
 * is never materialised; bias is an epilogue), the batch is one launch (no
 * per-sample host loop, dcn_v2_cuda.c:61), errors are return codes.
 *
 *   y[b,o,h,w] = bias[o] + sum_{c,k} W[o,c,k] * m[b,k,h,w] *
 *                bilinear(x[b,c], h*s-p+i*d+dy[b,k,h,w], w*s-p+j*d+dx[b,k,h,w])
 * with the reference's sampling-window rule (dcn_v2_im2col_cuda.cu:165) and
 * per-corner zeroing (:30-41).  offset channel 2k = dy, 2k+1 = dx of tap k.
 *
 * layout = CN_LAYOUT_NCHW: input (B,Cin,H,W), offset (B,dg*2*kh*kw,Ho,Wo),
 *   mask (B,dg*kh*kw,Ho,Wo), output (B,Cout,Ho,Wo) -- exactly the reference.
 * layout = CN_LAYOUT_NHWC: input (B,H,W,Cin), offset_mask (B,Ho,Wo,om_pitch)
 *   holding [2*kh*kw offsets | kh*kw mask] per pixel (mask pointer ignored,
 *   om_pitch >= 3*kh*kw given through cn_dcn_v2_forward_nhwc_f32 below).
 *
 * weight is always the reference's (Cout,Cin,kh,kw) tensor here; use
 * cn_pack_conv_weight_f32 + the *_packed entry points to skip the repack.
 * Domain: the reference operator's -- any kernel size, stride, padding, dilation,
 * deformable_group (Cin % deformable_group == 0) and channel count.  The configuration
 * CenterNet instantiates (3x3, stride 1, pad 1, dilation 1, one group, Cin % 4 == 0:
 * resnet_dcn.py:221-223, pose_dla_dcn.py:352) runs on the NHWC MFMA kernel and needs
 * cn_dcn_v2_forward_workspace_bytes() of 16-byte aligned scratch (layout change + tap
 * split); every other configuration (e.g. DCNv2/test.py:16-19 inC = 2, :169-179
 * deformable_groups = 2) runs on the general-domain NCHW kernel (cn_dcn_general.hip),
 * which takes no workspace (NULL / 0 accepted).  Invalid geometry -> CN_ERR_SHAPE.
 * ------------------------------------------------------------------------ */
size_t cn_dcn_v2_forward_workspace_bytes(int B, int Cin, int H, int W, int Cout,
                                         int kernel_h, int kernel_w, int layout);

int cn_dcn_v2_forward_f32(const float *input, const float *weight, const float *bias,
                          const float *offset, const float *mask, float *output,
                          int B, int Cin, int H, int W, int Cout,
                          int kernel_h, int kernel_w, int stride_h, int stride_w,
                          int pad_h, int pad_w, int dilation_h, int dilation_w,
                          int deformable_group, int apply_mask_sigmoid,
                          void *workspace, size_t workspace_bytes, void *stream);

/*
 * Native form used by the network plan: NHWC activations, packed weight
 * ([tap][Cout_pad][Cin], see cn_pack_conv_weight_f32), offsets and mask logits
 * interleaved per pixel as produced by the conv_offset_mask convolution
 * (DCNv2/dcn_v2.py:64-68: channels 0..17 offsets, 18..26 mask logits; the
 * sigmoid of :67 is applied inside when mask_sigmoid != 0), and a fused
 * epilogue  y = relu?( (acc + bias) * scale + shift )  that folds the
 * BatchNorm + ReLU following every DCN in resnet_dcn.py:237-239 /
 * pose_dla_dcn.py:345-357.  scale/shift may be NULL (identity).
 * workspace: small maps split the nine taps over 3 or 9 workgroups per output tile (fp32
 * partial sums, deterministic reduce + epilogue in a second launch) so that the
 * latency-bound gather still fills the 256 CUs; that needs
 * cn_dcn_v2_forward_nhwc_workspace_bytes() of 16-byte aligned scratch.  With a NULL or
 * too small workspace the layer runs unsplit (same result up to fp32 summation order).
 */
size_t cn_dcn_v2_forward_nhwc_workspace_bytes(int B, int Cin, int H, int W, int Cout);
/* dtype-generic form: the input stays a PLAIN fp32 NHWC tensor in every mode (one 16-byte gather
 * per bilinear corner); dtype = CN_DTYPE_F32S takes an f32s-packed weight, contracts on the
 * fp16 matrix instruction (the blended sample is split as it enters the LDS A tile) and writes
 * y as f32s (pitch % 32 == 0) or, with CN_CONV_Y_PLAIN, as plain fp32. */
int cn_dcn_v2_forward_nhwc(const float *input_nhwc, const void *weight_packed, const float *bias,
                           const float *offset_mask_nhwc, int om_pitch, const float *scale,
                           const float *shift, void *output_nhwc, int out_pitch, int B, int Cin,
                           int H, int W, int Cout, int mask_sigmoid, int relu, int dtype, int flags,
                           const cn_f32s_ctl *ctl, void *workspace, size_t workspace_bytes,
                           void *stream);
int cn_dcn_v2_forward_nhwc_f32(const float *input_nhwc, const float *weight_packed,
                               const float *bias, const float *offset_mask_nhwc,
                               int om_pitch, const float *scale, const float *shift,
                               float *output_nhwc, int B, int Cin, int H, int W,
                               int Cout, int mask_sigmoid, int relu, void *workspace,
                               size_t workspace_bytes, void *stream);

/* Plain fp32 NHWC <-> f32s NHWC (CN_DTYPE_F32S: fp16 high / low pairs in 128-byte groups of 32
 * channels).  Pitches in channels; f32s pitches are multiples of 32; channels of the last
 * group beyond C are written as zeros.  New surface (the reference has one tensor format). */
int cn_f32_to_f32s(const float *x, void *y, size_t npix, int C, int in_pitch, int out_pitch,
                   void *stream);
int cn_f32s_to_f32(const void *x, float *y, size_t npix, int C, int in_pitch, int out_pitch,
                   void *stream);
/* the same with the tensor's exponent: y = split(x * mul) (side 1 of `range`, CN_RANGE_WORDS words, receives max |x * mul|) and
 * y = join(x) * mul; mul is a power of two (0 = 1), range may be NULL */
int cn_f32_to_f32s_scaled(const float *x, void *y, size_t npix, int C, int in_pitch, int out_pitch,
                          float mul, uint32_t *range, void *stream);
int cn_f32s_to_f32_scaled(const void *x, float *y, size_t npix, int C, int in_pitch, int out_pitch,
                          float mul, void *stream);
/* max |x| over the C channels of npix pixels of a plain fp32 NHWC tensor, max-ed into *word as a
 * float bit pattern (the caller zeroes it): the calibration pass of the f32s exponents */
int cn_absmax_f32(const float *x, size_t npix, int C, int pitch, uint32_t *word, void *stream);
/* end-of-forward bookkeeping of the range words of n_launches launches (cur: n_launches x
 * CN_RANGE_WORDS; hi, lo: n_launches x 2 sides): m = max over the slots, hi = max(hi, m),
 * lo = min(lo, m) over the non-zero m, slots = 0 -- the host reads hi / lo whenever it synchronises anyway (largest
 * and smallest per-forward maximum since its last look) instead of after every forward */
int cn_range_fold(uint32_t *cur, uint32_t *hi, uint32_t *lo, int n_launches, void *stream);
/* The same, plus a sticky two-word digest for a cheap host check: summary[0] = max over all
 * (launch, side) values seen so far (float bits; > 65504 or NaN bits = something was clamped),
 * summary[1] = the smallest NON-ZERO per-forward maximum so far (initialise to 0x7f800000).  The
 * host reads 8 bytes per forward and walks the hi / lo tables only when the digest is out of bounds. */
int cn_range_fold_digest(uint32_t *cur, uint32_t *hi, uint32_t *lo, uint32_t *summary,
                         int n_launches, void *stream);

/* Flip-test averaging (detectors/ctdet.py:34-37, detectors/multi_pose.py:44-55, models/utils.py:28-50).
 * x_pair (2, C, H, W): image 1 is the horizontally mirrored frame; out (1, C, H, W) =
 * (f(x0[c, y, x]) + sign[c] * f(x1[src[c], y, W-1-x])) / 2.  chan_src (C) = left / right joint permutation
 * (flip_lr / flip_lr_off), NULL = identity; chan_sign (C), NULL = +1 (-1 on the x components of joint
 * offsets); apply_sigmoid: f = logistic, written back to x_pair in place (the reference's sigmoid_()),
 * chan_src must then be a permutation. */
int cn_flip_average_f32(float *x_pair, float *out, int C, int H, int W, const int32_t *chan_src,
                        const float *chan_sign, int apply_sigmoid, void *stream);
/* Box calibration (bench.py `box_calibration`; measurement aid, not on the product path; no
 * reference counterpart).  cn_calib_mfma_f16: a register-only v_mfma_f32_32x32x16_f16 loop on
 * every SIMD (1024 workgroups x 4 waves, `iters` x 16 instructions per wave); returns the FLOPs
 * of the launch (0 on error) -- the caller times it with HIP events.  cn_calib_copy: a float4
 * grid-stride copy of `bytes` (multiple of 16) from src to dst. */
double cn_calib_mfma_f16(float *sink, int iters, void *stream);
int cn_calib_copy(const void *src, void *dst, size_t bytes, void *stream);
/* cn_calib_latency: ONE lane walks `steps` dependent loads through `chain` (chain[i] = index of the
 * next element; the caller lays the walk out: a buffer beyond the caches for the HBM latency, a small
 * one for L2), then `steps` dependent device-scope atomic additions on *atom.  out[0], out[1]: ticks of
 * the constant 100 MHz clock each chain took; out[2]: sink. */
int cn_calib_latency(const uint32_t *chain, uint32_t start, int steps, uint32_t *atom,
                     unsigned long long *out, void *stream);
/* cn_calib_clock: one lane counts shader-core cycles over `ticks` ticks of the constant 100 MHz clock:
 * out[0] = core cycles, out[1] = ticks -- the core clock at the moment the launch runs (enqueue it right
 * behind the work whose clock is asked for). */
int cn_calib_clock(unsigned long long *out, int ticks, void *stream);

/* ------------------------------------------------------------------------
 * Dense convolution as an implicit GEMM on fp32 MFMA (no im2col buffer).
 *
 * Replaces the torch.nn.Conv2d / BatchNorm2d(eval) / ReLU / residual-add call
 * sites of the backbones (resnet_dcn.py:38-67,138-142,155-177;
 * msra_resnet.py; pose_dla_dcn.py:147-221; large_hourglass.py:17-74) and, with
 * the output-scatter arguments, the four parity classes of
 * ConvTranspose2d(k=4,s=2,p=1) (resnet_dcn.py:228-235).
 *
 *   y = relu?( conv(x, w) * scale + shift + residual? )
 *
 * x: NHWC with pixel pitch in_pitch floats (>= Cin; lets a layer read a channel
 * slice of a wider tensor), or NCHW3 for the stem (in_layout = NCHW, Cin = 3).
 * w: packed by cn_pack_conv_weight_f32.  y: NHWC (pitch out_pitch) or NCHW.
 * Output pixel (oy,ox) of the conv grid (Ho,Wo) is written to
 * (oy*oy_mul+oy_add, ox*ox_mul+ox_add) of a (B,OH,OW) map.
 * ------------------------------------------------------------------------ */
typedef struct cn_conv_desc {
    int B, H, W, Cin;       /* input */
    int Ho, Wo, Cout;       /* conv output grid */
    int KH, KW;
    int stride, pad_h, pad_w, dil;
    int in_layout, in_pitch; /* pitch in floats per pixel (NHWC) */
    int out_layout, out_pitch;
    int OH, OW, oy_mul, oy_add, ox_mul, ox_add;
    int relu;
    int dtype;              /* CN_DTYPE_F32 / CN_DTYPE_F16 / CN_DTYPE_F32S (x, w, residual, NHWC y) */
    int flags;              /* CN_CONV_* bits, 0 by default */
    int res_pitch;          /* pixel pitch of the residual in floats; 0 = out_pitch.  A pitch that differs
                               from out_pitch (y or the residual is a channel slice of a wider tensor, e.g. a
                               concatenation buffer: pose_dla_dcn.py:157-165) is taken by the 3x3 / stride 1
                               / pad 1 layers only: ask cn_conv2d_res_pitch_supported() */
    cn_f32s_ctl ctl;        /* dtype = CN_DTYPE_F32S (and the CN_CONV_STEM_F32S stem): range control */
} cn_conv_desc;
/* 1 when cn_conv2d honours d->res_pitch != d->out_pitch for this descriptor (host-only) */
int cn_conv2d_res_pitch_supported(const cn_conv_desc *d);
/* dtype = CN_DTYPE_F32S only: x (and the residual) / y are plain fp32 NHWC tensors; the kernel
 * converts while staging / storing (e.g. the offset maps the deformable kernel reads). */
#define CN_CONV_X_PLAIN 1
#define CN_CONV_Y_PLAIN 2
#define CN_CONV_R_PLAIN 4
/* dtype = CN_DTYPE_F32, 3-channel NCHW stem only: compute with three fp16 MFMAs per product
 * (the image and the packed weight stay fp32 and are split inside the kernel) */
#define CN_CONV_STEM_F32S 8
/* with CN_CONV_STEM_F32S: the MaxPool2d(kernel 3, stride 2, padding 1) that follows the stem in the
 * ResNet backbones (resnet_dcn.py:138-141, msra_resnet.py:116-119) is applied inside the kernel;
 * Ho / Wo stay the convolution's output size, y is the pooled (B, Ho/2, Wo/2, out_pitch) tensor.
 * Only for shapes cn_stem_maxpool_supported() accepts, CN_ERR_UNSUPPORTED otherwise. */
#define CN_CONV_STEM_MAXPOOL 16
/* with CN_CONV_STEM_MAXPOOL: y (the pooled map) is written as an f32s tensor (out_pitch % 32 == 0,
 * 128-byte aligned) holding y * 2^-e -- the exponent folded into scale / shift by the caller -- and
 * side 0 of ctl.range receives max |y|: the consumers of the ResNet stem are f32s layers */
#define CN_CONV_STEM_Y_F32S 32

/* 1 when cn_conv2d accepts CN_CONV_STEM_MAXPOOL for this stem descriptor (7x7 / stride 2, 33..64
 * output channels, rows of 1..4 whole 128-pixel tiles, even Ho, batch * Ho large enough to fill
 * the chip with row strips), else 0.  Host-only, no GPU needed. */
int cn_stem_maxpool_supported(const cn_conv_desc *d);
/* 1 when cn_conv2d runs this stem descriptor with CN_CONV_STEM_F32S arithmetic (only then are
 * ctl.x_mul / ctl.range used: the fp32 stem kernels split nothing), else 0.  Host-only. */
int cn_stem_f32s_supported(const cn_conv_desc *d);

/* Number of floats of the packed weight for (Cout,Cin,KH,KW). */
size_t cn_packed_conv_weight_floats(int Cout, int Cin, int KH, int KW);
/* (Cout,Cin,KH,KW) fp32 device tensor -> packed [KH*KW][Cout_pad32][Cin_pad] */
int cn_pack_conv_weight_f32(const float *w_oihw, float *w_packed, int Cout, int Cin,
                            int KH, int KW, void *stream);

int cn_conv2d_f32(const cn_conv_desc *desc, const float *x, const float *w_packed,
                  const float *scale, const float *shift, const float *residual,
                  float *y, void *stream);
/* dtype-generic forms (desc->dtype selects fp32 / fp16; NCHW head outputs and the stem's
 * NCHW image are always fp32; scale/shift are always fp32).
 * CN_DTYPE_F32S: 3x3 and (round 6) 1x1 kernels carry a second, fragment-ordered copy of the matrix behind
 * the row-ordered one ([tap][chunk][32-channel block][quarter][lane] x 16 bytes: what a lane of the
 * 32x32x16 MFMA holds, one contiguous 1 KiB per wave-load); cn_packed_conv_weight_elems counts both. */
size_t cn_packed_conv_weight_elems(int Cout, int Cin, int KH, int KW, int dtype);
int cn_pack_conv_weight(const float *w_oihw, void *w_packed, int Cout, int Cin, int KH, int KW,
                        int dtype, void *stream);
/* workspace: optional scratch for split-K (layers whose plain grid would leave most of the
 * 256 CUs idle: small maps with deep K); query with cn_conv2d_workspace_bytes, NULL/0 = never
 * split.  The split is deterministic (fixed partition, second-stage reduce, no atomics). */
size_t cn_conv2d_workspace_bytes(const cn_conv_desc *desc);
int cn_conv2d(const cn_conv_desc *desc, const void *x, const void *w_packed, const float *scale,
              const float *shift, const void *residual, void *y, void *workspace,
              size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------
 * ConvTranspose2d(kernel 4, stride 2, padding 1, bias=False) + BN(eval) + ReLU.
 * Replaces the up-sampling layers of resnet_dcn.py:228-235 / msra_resnet.py
 * (_make_deconv_layer).  Executed as the four output-parity 2x2 convolutions in
 * ONE launch.  w_iohw is torch's (Cin,Cout,4,4) ConvTranspose2d weight.
 * x (B,H,W,in_pitch) NHWC -> y (B,2H,2W,out_pitch) NHWC.
 * ------------------------------------------------------------------------ */
size_t cn_packed_deconv4x4s2_weight_floats(int Cin, int Cout);
/* dtype-generic forms (CN_DTYPE_F32 / CN_DTYPE_F32S; `flags`: CN_CONV_X_PLAIN / CN_CONV_Y_PLAIN) */
int cn_pack_deconv4x4s2_weight(const float *w_iohw, void *w_packed, int Cin, int Cout, int dtype,
                               void *stream);
int cn_conv_transpose4x4s2(const void *x_nhwc, const void *w_packed, const float *scale,
                           const float *shift, void *y_nhwc, int B, int H, int W, int Cin, int Cout,
                           int in_pitch, int out_pitch, int relu, int dtype, int flags,
                           const cn_f32s_ctl *ctl, void *stream);
int cn_pack_deconv4x4s2_weight_f32(const float *w_iohw, float *w_packed, int Cin, int Cout,
                                   void *stream);
int cn_conv_transpose4x4s2_f32(const float *x_nhwc, const float *w_packed, const float *scale,
                               const float *shift, float *y_nhwc, int B, int H, int W, int Cin,
                               int Cout, int in_pitch, int out_pitch, int relu, void *stream);

/* 3x3 / stride-2 / pad-1 max pooling, NHWC (resnet_dcn.py:142). */
int cn_maxpool3x3s2_nhwc_f32(const float *x, float *y, int B, int H, int W, int C,
                             void *stream);
/* generic k x k / stride s max pooling NHWC (pose_dla_dcn.py:200 uses 2x2 s2) */
int cn_maxpool_nhwc_f32(const float *x, float *y, int B, int H, int W, int C, int k,
                        int s, int pad, void *stream);
/* the same with an f32s input tensor (in_dtype = CN_DTYPE_F32S, C % 32 == 0; CN_DTYPE_F32: as
 * above); the output is plain fp32 */
int cn_maxpool_nhwc(const void *x_nhwc, float *y_nhwc, int B, int H, int W, int C, int k, int s,
                    int pad, int in_dtype, void *stream);
/* f32s input of exponent e: out_mul = 2^e brings the pooled values back to real units */
int cn_maxpool_nhwc_scaled(const void *x_nhwc, float *y_nhwc, int B, int H, int W, int C, int k,
                           int s, int pad, int in_dtype, float out_mul, void *stream);
/* f32s in (pixel pitch in_pitch), f32s out (pixel pitch out_pitch -- y may be a channel slice of a wider
 * tensor, e.g. the concatenation buffer of pose_dla_dcn.py:157-165): y = split(max(window) * mul), mul a
 * power of two (the ratio of the two tensors' exponents); range (CN_RANGE_WORDS words, may be NULL)
 * side 0 receives max |y| */
int cn_maxpool_nhwc_f32s(const void *x, void *y, int B, int H, int W, int C, int in_pitch, int out_pitch,
                         int k, int s, int pad, float mul, uint32_t *range, void *stream);

/* Depthwise ConvTranspose2d(C, C, kernel 2f, stride f, padding f/2, groups=C, bias=False)
 * -- the up-sampling of IDAUp (pose_dla_dcn.py:370-373) -- fused with the element-wise
 * add that follows it (IDAUp.forward, :385-386: node(layers[i] + layers[i-1])).
 * x (B,H,W,C) NHWC; w_taps [(2f)*(2f)][C] (tap-major: w_taps[ky*2f+kx][c] = w[c,0,ky,kx]);
 * add (B,fH,fW,C) or NULL; y (B,fH,fW,C). */
int cn_dw_conv_transpose_f32(const float *x, const float *w_taps, const float *add, float *y,
                             int B, int H, int W, int C, int f, void *stream);
/* Channel-slice copy between NHWC tensors of different pitch (torch.cat(.., 1) of
 * Root.forward, pose_dla_dcn.py:159). */
int cn_copy_channels_f32(const float *src, int src_pitch, float *dst, int dst_pitch,
                         size_t npix, int C, void *stream);
/* Nearest x2 up-sampling + add of the skip branch (large_hourglass.py:102-109, 163-174). */
int cn_upsample2x_add_f32(const float *x, const float *add, float *y, int B, int H, int W,
                          int C, void *stream);
int cn_upsample2x_add_f16(const void *x, const void *add, void *y, int B, int H, int W, int C,
                          void *stream);

/* Fused detection heads.  The reference builds every head as
 *   nn.Sequential(Conv2d(F, head_conv, 3, padding=1, bias=True), ReLU, Conv2d(head_conv, C, 1))
 * (resnet_dcn.py:155-177, msra_resnet.py) and runs them one after the other on the same
 * feature map.  Here all heads are ONE launch: w1_packed is cn_pack_conv_weight_f32 of the
 * first convolutions concatenated along Cout, (n_heads*head_conv, F, 3, 3); bias1 the
 * concatenated biases; the hidden activations stay in LDS and each head's 1x1 convolution
 * runs as a second matrix-core GEMM in the same workgroup, writing the NCHW maps
 * (B, cout, H, W) that cn_ctdet_decode_f32 / cn_multi_pose_decode_f32 consume.
 * x is NHWC with row pitch in_pitch.  Built for n_heads <= 8 and head_conv = 64 (resnet_dcn.py),
 * or 128 / 192 / 256 (pose_dla_dcn.py:456-468, large_hourglass.py) with every cout <= 96: the
 * hidden layer is then processed in 64-channel slices whose 1x1 products accumulate in registers.
 * Anything else returns CN_ERR_UNSUPPORTED (the caller then issues cn_conv2d per convolution). */
typedef struct cn_head_out {
    const float *w;    /* (cout, head_conv) row-major == Conv2d(head_conv, cout, 1).weight */
    const float *bias; /* (cout) or NULL */
    float *y;          /* (B, cout, H, W) */
    int cout;
    int reserved;
    const float *oscale; /* (cout) or NULL: y = acc * oscale + bias -- undoes a per-row pre-scale of w
                            and the exponent of the hidden tile (f32s; see cn_f32s_ctl) */
    const void *w_frag;  /* optional (f32s, head_conv = 64, cout <= 96): the same matrix w as (high, low)
                            fp16 fragments, cn_pack_head_w2_f32s -- with it the heads run on the
                            persistent kernel (cn_conv3x3p.hip); NULL = the one-tile-per-workgroup kernel */
} cn_head_out;
/* w (cout, 64) row-major -> out (cn_packed_head_w2_bytes(cout) bytes, 16-byte aligned) */
size_t cn_packed_head_w2_bytes(int cout);
int cn_pack_head_w2_f32s(const float *w, void *out, int cout, void *stream);
int cn_heads3x3_1x1_f32(const float *x, int B, int H, int W, int Cin, int in_pitch,
                        const float *w1_packed, const float *bias1, int head_conv, int n_heads,
                        const cn_head_out *heads, void *stream);
/* dtype-generic form.  CN_DTYPE_F32S: x is an f32s tensor (or plain fp32 with CN_CONV_X_PLAIN),
 * w1_packed an f32s-packed weight whose per-row prescale factors come in scale1 (NULL = 1);
 * the hidden tile and the 1x1 weights are split inside the kernel; outputs are fp32 NCHW. */
int cn_heads3x3_1x1(const void *x, int B, int H, int W, int Cin, int in_pitch,
                    const void *w1_packed, const float *scale1, const float *bias1, int head_conv,
                    int n_heads, const cn_head_out *heads, int dtype, int flags,
                    const cn_f32s_ctl *ctl, void *stream);

/* ------------------------------------------------------------------------
 * Pre-process on the device: BaseDetector.pre_process
 * (src/lib/detectors/base_detector.py:37-65) = cv2.resize (scale != 1) +
 * cv2.warpAffine(INTER_LINEAR, zero border) + (x/255 - mean)/std + HWC->CHW (+ flip concat).
 *
 * cn_warp_normalize_u8_f32: image (H, W, 3) uint8 on the device (row pitch in bytes),
 *   dst_to_src_2x3 = six HOST doubles: trans_input inverted the way cv::warpAffine inverts its M
 *   (centernet_amd/image.py invert_affine), out = (1 + flip_concat, 3, out_h, out_w) fp32 NCHW
 *   with out[1] = out[0] flipped along x (base_detector.py:59-60).  mean3 / std3: HOST floats.
 * cn_resize_bilinear_u8: cv2.resize(image, (out_w, out_h)), uint8 HWC -> dense uint8 HWC.
 * Arithmetic: OpenCV's uint8 INTER_LINEAR path -- fixed point, not float bilinear: warpAffine
 * samples at 1/32-pixel positions (AB_BITS = 10, INTER_BITS = 5) with 15-bit weights
 * (INTER_REMAP_COEF_BITS) and round-half-up, zero border; resize is separable with 11-bit
 * coefficients and its own uint8 vertical pass, a copy at equal size and the 2 x 2 mean at
 * exactly half size.  The published algorithm is restated with its constants in
 * oracle/pre_oracle.py (OpenCV is a third-party dependency that is absent here); these entry
 * points equal that restatement bit for bit.  Normalisation: float64, rounded once to fp32.
 * ------------------------------------------------------------------------ */
int cn_warp_normalize_u8_f32(const uint8_t *image_hwc, int H, int W, int pitch_bytes,
                             const double *dst_to_src_2x3, int out_h, int out_w,
                             const float *mean3, const float *std3, int flip_concat,
                             float *out_nchw, void *stream);
/* the same for N images of one geometry in ONE launch (frames of a video, a batch from the loader):
 * images image_stride_bytes apart, outputs dense (N, 3 | 6, out_h, out_w) */
int cn_warp_normalize_u8_f32_batch(const uint8_t *images_hwc, int N, size_t image_stride_bytes, int H, int W,
                                   int pitch_bytes, const double *dst_to_src_2x3, int out_h, int out_w,
                                   const float *mean3, const float *std3, int flip_concat,
                                   float *out_nchw, void *stream);
/* ctdet_post_process + the per-class split (utils/post_process.py:83-100, utils/image.py:19-24,63-66,
 * detectors/ctdet.py:47-56) on the device.  dets (B, K, 6) raw detections in output-grid units (K <= 128);
 * to_source_2x3: the float64 inverse map of get_affine_transform(c, s, 0, out_size, inv=1) -- one for
 * all images (per_image = 0) or B of them on the device (per_image = 1; device pointer either way);
 * rows (B, K, 5): [x1, y1, x2, y2, score] in source pixels / scale, grouped by class, inside a class
 * in their original order; bounds (B, num_classes + 1): class j of image b = rows[b, bounds[b, j] :
 * bounds[b, j + 1]] (rows with a class outside [0, num_classes) lie behind bounds[b, num_classes]).
 * Bit-identical to the reference's float64 affine + float32 rounding. */
int cn_ctdet_post_process_f32(const float *dets, int B, int K, int num_classes, const double *to_source_2x3,
                              int per_image, float scale, float *rows, int32_t *bounds, void *stream);
int cn_resize_bilinear_u8(const uint8_t *image_hwc, int H, int W, int pitch_bytes, int out_h,
                          int out_w, uint8_t *out_hwc, void *stream);

/* The same two operations on the HOST, for callers that keep BaseDetector.pre_process on host
 * cores (DataLoader workers; base_detector.py:37-65): identical integer arithmetic, results
 * equal the device kernels bit for bit.  channels <= 4; dst_to_src_2x3 as above. */
int cn_warp_affine_u8_host(const uint8_t *image_hwc, int h_in, int w_in, int channels,
                           const double *dst_to_src_2x3, int h_out, int w_out, uint8_t *out_hwc);
int cn_resize_linear_u8_host(const uint8_t *image_hwc, int h_in, int w_in, int channels, int h_out,
                             int w_out, uint8_t *out_hwc);

/* ((image / 255. - mean) / std).astype(float32) + HWC -> CHW of a 3-channel uint8 image on the HOST
 * (base_detector.py:56-58), numpy's float64 arithmetic, one rounding to float32. */
int cn_normalize_u8_chw_f32_host(const uint8_t *image_hwc, int h, int w, const float *mean3,
                                 const float *std3, float *out_chw);

/* Soft-NMS on a HOST array, in place (rows of `stride` floats: x1,y1,x2,y2,score,...).
 * Replaces external.nms.soft_nms / soft_nms_39 (src/lib/external/nms.pyx:77-275), used by
 * merge_outputs when --nms or multi-scale testing is on (detectors/ctdet.py:63-64).
 * method 0 = hard NMS, 1 = linear, 2 = gaussian.  Returns the kept count (>= 0) or < 0. */
int cn_soft_nms_f32(float *boxes_host, int n, int stride, float sigma, float Nt,
                    float threshold, int method);

/* Layout conversion at the API edge. */
int cn_nchw_to_nhwc_f32(const float *x, float *y, int B, int C, int H, int W,
                        int out_pitch, void *stream);
int cn_nhwc_to_nchw_f32(const float *x, float *y, int B, int C, int H, int W,
                        int in_pitch, void *stream);

/* ------------------------------------------------------------------------
 * ctdet decode: sigmoid -> 3x3 peak test -> per-class top-K -> global top-K
 *               -> wh/reg gather -> box assembly.
 *
 * Replaces: ctdet_decode(heat, wh, reg=None, cat_spec_wh=False, K=100)
 *   (models/decode.py:464-495) together with hm.sigmoid_() of
 *   detectors/ctdet.py:31 when apply_sigmoid != 0, i.e. _nms (decode.py:9-15),
 *   _topk (:103-119) and _transpose_and_gather_feat (models/utils.py:22-26).
 *
 * heat (B,C,H,W) NCHW: logits when apply_sigmoid, else post-sigmoid scores
 *   (what the reference function receives).
 * wh (B,2,H,W) or (B,2C,H,W) if cat_spec_wh; reg (B,2,H,W) or NULL.
 * dets (B,K,6) = [x1,y1,x2,y2,score,cls] sorted by score descending.
 * inds (B,K) int32 spatial index of each detection, may be NULL.
 * Tie order (unspecified by torch.topk): score desc, class asc, index asc.
 * Requires K <= 128 and K <= H*W (torch.topk raises for K > H*W).
 * ------------------------------------------------------------------------ */
size_t cn_ctdet_decode_workspace_bytes(int B, int C, int H, int W, int K);

int cn_ctdet_decode_f32(const float *heat, const float *wh, const float *reg,
                        int B, int C, int H, int W, int K, int cat_spec_wh,
                        int apply_sigmoid, float *dets, int32_t *inds,
                        void *workspace, size_t workspace_bytes, void *stream);

/* Flag bits of the `apply_sigmoid` argument of cn_nms_topk_channel_f32 / cn_topk_f32:
 * bit 0 = apply the logistic first (detectors/ctdet.py:31); CN_DECODE_NO_PEAK_TEST = rank every
 * cell, i.e. the plain _topk_channel / _topk without the _nms in front. */
#define CN_DECODE_SIGMOID 1
#define CN_DECODE_NO_PEAK_TEST 512
/* Image-level top-K (cn_ctdet_decode_f32, cn_topk_f32) on planes of <= 128 x 128 cells is ONE kernel
 * launch that reads the heat-map once.  Its per-image state words live in the workspace, must be
 * zero on entry and are left at zero on exit; by default the call zeroes them itself (one small
 * fill in front of the kernel).  CN_DECODE_STATE_CLEAN: the caller owns this workspace exclusively,
 * zeroed it once after allocation (all of it) and has used it for nothing but calls of the same
 * entry point with the same (B, C, H, W, K) since -- the fill is skipped.
 * CN_DECODE_TWO_LAUNCHES selects the two-launch form (group maxima + threshold-pruned collect, the
 * form for larger planes) and 2048 the per-(class, band) select, for comparison; all forms are
 * bit-identical. */
#define CN_DECODE_STATE_CLEAN 4096
/* Byte offset and size, inside the workspace of cn_ctdet_decode_f32 / cn_topk_f32, of the per-image state
 * words of the one-launch form (0 bytes when the shape takes another form).  A CN_DECODE_STATE_CLEAN caller
 * re-zeroes (or checks) exactly this region after a failed or aborted call instead of trusting it. */
int cn_decode_state_region(int B, int C, int H, int W, int K, size_t *offset, size_t *bytes);
#define CN_DECODE_TWO_LAUNCHES 8192
#define CN_DECODE_PER_BAND 2048

/* _nms + _topk_channel (models/decode.py:9-15, 92-101) as one kernel: per
 * (b,c) plane the K best peaks; scores (B,C,K) desc, inds (B,C,K) int32. */
int cn_nms_topk_channel_f32(const float *heat, int B, int C, int H, int W, int K,
                            int apply_sigmoid, float *scores, int32_t *inds,
                            void *workspace, size_t workspace_bytes, void *stream);

/* _nms + _topk (models/decode.py:9-15, 103-119): global top-K over all classes of an image.
 * scores (B,K) desc, inds (B,K) = y*W+x, clses (B,K); ys = inds / W, xs = inds % W.
 * Workspace: cn_ctdet_decode_workspace_bytes. */
int cn_topk_f32(const float *heat, int B, int C, int H, int W, int K, int apply_sigmoid,
                float *scores, int32_t *inds, int32_t *clses, void *workspace,
                size_t workspace_bytes, void *stream);

/* _transpose_and_gather_feat (models/utils.py:12-26): out[b,k,c] = feat[b,c,inds[b,k]] for an
 * NCHW map, without the reference's full-tensor permute().contiguous(). */
int cn_gather_feat_f32(const float *feat, const int32_t *inds, float *out, int B, int C, int H,
                       int W, int K, void *stream);

/* ddd_decode(heat, rot, depth, dim, wh=None, reg=None, K=40) (models/decode.py:426-462):
 * dets (B,K,16) = [xs, ys, score, rot x8, depth, dim x3, cls], or (B,K,18) with wh x2 before
 * cls when wh != NULL.  heat is post-sigmoid unless apply_sigmoid; depth is passed as the
 * caller prepared it (the detector applies 1/(sigmoid(dep)+1e-6)-1 first, detectors/ddd.py:55). */
size_t cn_ddd_decode_workspace_bytes(int B, int C, int H, int W, int K);
int cn_ddd_decode_f32(const float *heat, const float *rot, const float *depth, const float *dim,
                      const float *wh, const float *reg, int B, int C, int H, int W, int K,
                      int apply_sigmoid, float *dets, void *workspace, size_t workspace_bytes,
                      void *stream);

/* Edge aggregation in front of exct_decode when aggr_weight > 0 (models/decode.py:17-90, :136-140):
 * out = _h_aggregate(heat, aggr_weight) (horizontal = 1: t_heat, b_heat) or _v_aggregate (horizontal =
 * 0: l_heat, r_heat); heat, out (B, C, H, W), out != heat.  Bit-identical to the reference's float32
 * arithmetic. */
int cn_exct_aggregate_f32(const float *heat, float *out, int B, int C, int H, int W, int horizontal,
                          float aggr_weight, void *stream);
/* exct_decode(t_heat, l_heat, b_heat, r_heat, ct_heat, t_regr.., K=40, scores_thresh=0.1,
 *             center_thresh=0.1, aggr_weight=0.0, num_dets=1000)   (models/decode.py:273-424)
 * for aggr_weight == 0: dets (B, num_dets, 14) = [l_x, t_y, r_x, b_y, score, t_x, t_y, l_x, l_y,
 * b_x, b_y, r_x, r_y, cls].  The four regression maps are used only when all are given
 * (decode.py:372-373).  Heat-maps are post-sigmoid.  K <= 64, num_dets <= 1024.
 * Tie order of equal scores (unspecified by torch.topk): candidate index ascending.
 * The `apply_sigmoid` argument carries flags: bit 0 (the logistic) is not supported here;
 * CN_EXCT_CLAMP_ONE = the edge maps may exceed 1 (they are aggregated maps, cn_exct_aggregate_f32):
 * after the 3x3 peak test the surviving values are clamped to 1 before the top-K, as the reference
 * does (decode.py:302-305: `t_heat[t_heat > 1] = 1` behind `_nms`); runs one more pass per edge map. */
#define CN_EXCT_CLAMP_ONE 2
size_t cn_exct_decode_workspace_bytes(int B, int C, int H, int W, int K);
int cn_exct_decode_f32(const float *t_heat, const float *l_heat, const float *b_heat,
                       const float *r_heat, const float *ct_heat, const float *t_regr,
                       const float *l_regr, const float *b_regr, const float *r_regr, int B, int C,
                       int H, int W, int K, float scores_thresh, float center_thresh, int num_dets,
                       int apply_sigmoid, float *dets, void *workspace, size_t workspace_bytes,
                       void *stream);
/* agnex_ct_decode(t_heat, l_heat, b_heat, r_heat, ct_heat, t_regr.., K=40, scores_thresh=0.1,
 *                 center_thresh=0.1, aggr_weight=0.0, num_dets=1000)   (models/decode.py:121-271), the
 * class-agnostic form behind --agnostic_ex (detectors/exdet.py:26): the four edge maps are (B, 1, H, W), the
 * centre map (B, C, H, W); a grouping is scored against the per-cell MAXIMUM of the centre map over the classes
 * (torch.max(ct_heat, dim=1), decode.py:164), there is no class rule, and a detection's class is the arg-max
 * (first on ties) at the box centre.  Same row layout, limits, tie order and `flags` (CN_EXCT_CLAMP_ONE) as
 * cn_exct_decode_f32; the scores are that function's over single-channel maps, bit for bit. */
size_t cn_agnex_ct_decode_workspace_bytes(int B, int C, int H, int W, int K);
int cn_agnex_ct_decode_f32(const float *t_heat, const float *l_heat, const float *b_heat,
                           const float *r_heat, const float *ct_heat, const float *t_regr,
                           const float *l_regr, const float *b_regr, const float *r_regr, int B, int C,
                           int H, int W, int K, float scores_thresh, float center_thresh, int num_dets,
                           int flags, float *dets, void *workspace, size_t workspace_bytes,
                           void *stream);

/* ------------------------------------------------------------------------
 * multi_pose decode.
 * Replaces: multi_pose_decode(heat, wh, kps, reg, hm_hp, hp_offset, K)
 *   (models/decode.py:497-571).  dets (B,K,4+1+2J+1).
 * ------------------------------------------------------------------------ */
size_t cn_multi_pose_decode_workspace_bytes(int B, int C, int H, int W, int J, int K);

int cn_multi_pose_decode_f32(const float *heat, const float *wh, const float *kps,
                             const float *reg, const float *hm_hp,
                             const float *hp_offset, int B, int C, int H, int W,
                             int J, int K, int apply_sigmoid, float *dets,
                             void *workspace, size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CENTERNET_AMD_H */
