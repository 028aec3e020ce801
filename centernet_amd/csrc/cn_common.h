// cn_common.h -- shared host/device helpers for libcenternet_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/centernet_amd.h"

#include <mutex>
#define CN_WAVE 64

// Raise a kernel's dynamic-LDS limit exactly once per process (thread-safe).
#define CN_SET_MAX_LDS_ONCE(kernel, bytes)                                                   \
    do {                                                                                     \
        static std::once_flag cn_once__;                                                     \
        std::call_once(cn_once__, [] {                                                       \
            (void)hipFuncSetAttribute((const void *)(kernel),                                \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
        });                                                                                  \
    } while (0)

#define CN_CHECK_LAUNCH()                                  \
    do {                                                   \
        hipError_t e__ = hipGetLastError();                \
        if (e__ != hipSuccess) return CN_ERR_LAUNCH;       \
    } while (0)

static inline bool cn_aligned16(const void *p) { return (((uintptr_t)p) & 15u) == 0; }
static inline int cn_cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t cn_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

typedef float cn_f32x4 __attribute__((ext_vector_type(4)));
typedef int cn_i32x4 __attribute__((ext_vector_type(4)));
typedef float cn_f32x16 __attribute__((ext_vector_type(16)));

// ---- "f32s": fp32 values stored as a pair of fp16 (CN_DTYPE_F32S) ------------------------
// gfx950's fp32 matrix instruction runs at 1/16 of the fp16 rate, so the dense layers compute
//     a*b  ~=  ah*bh + ah*bl + al*bh          (a = ah + al, b = bh + bl; fp32 accumulate)
// with three v_mfma_f32_32x32x16_f16 per product instead of one v_mfma_f32_32x32x2_f32: 5.3x
// the matrix throughput at fp32-level accuracy (ah = fp16(a) keeps 11 bits, al = fp16(a - ah)
// the next 11: |a - ah - al| <= 2^-22 |a| and the dropped al*bl term is <= 2^-22 |ab|, the
// size of fp32's own rounding error in a K-long accumulation).
// Storage: a group of 32 channels occupies the same 128 bytes as 32 floats -- 32 fp16 high
// parts, then 32 fp16 low parts -- so tensors keep their fp32 byte size, pitch and addressing
// per (pixel, 32-channel group); pitches of f32s tensors are multiples of 32 channels.
// A 128-byte group IS the LDS row the MFMA loop reads: bytes [32*kk + 16*h, +16) of a row are
// k = 16*kk + 8h .. +7 of the high parts (kk = 0, 1) or of the low parts (kk = 2, 3).
struct cn_f32s { float raw; };   // element tag of f32s tensors in kernel templates (4 bytes)
typedef _Float16 cn_f16x4v __attribute__((ext_vector_type(4)));
typedef _Float16 cn_f16x8v __attribute__((ext_vector_type(8)));

//
// Range.  An fp16 pair carries 22 bits only while the high part is a normal fp16 number and
// the low part has not gone subnormal, i.e. for 2^-3 <~ |a| <= 65504; below that the absolute
// error floor is 2^-25 whatever the magnitude.  The library therefore never splits a raw value:
// every f32s tensor holds   stored = real * 2^-e   with a per-tensor exponent e the caller picks
// so that the tensor's largest magnitude sits near 2^9 (centernet_amd/engine.py: measured once
// per network on a plain-fp32 calibration pass), weights are pre-scaled per output row to
// [2^13, 2^14), and the factors are undone -- exactly, they are powers of two -- through the
// per-channel epilogue scale the kernels apply anyway.  Kernels take the remaining multipliers
// in cn_f32s_ctl (x_mul for a plain input that is split while it is staged, res_mul for the
// residual).  Nothing saturates silently: every split site keeps the running max |value| it was
// asked to split and the launch max-es it into the caller's `range` words (cn_rng_* below); a
// value beyond 65504 is still clamped (its low part would be NaN) but the word then reads
// > 65504 and the host re-calibrates and re-runs (engine.Plan.check_range).
// Two values: high parts by one v_cvt_pk_f16_f32, low parts by v_fma_mixlo/mixhi_f16 -- fp16(fma(high
// as fp16 operand, -1, c)) in ONE instruction each.  c - high is exact in fp32 (high is c rounded
// to 11 bits), so this is bit for bit the (_Float16)(c - (float)high) it replaces, in 3 instructions
// per pair instead of 5 (cvt_pk, 2 x cvt_f32_f16, pk_add, cvt_pk): the split is the largest
// single item of VALU work in every f32s staging loop and epilogue.
typedef _Float16 cn_f16x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void cn_split2_bits(float c0, float c1, uint32_t &hi, uint32_t &lo)
{
    const cn_f16x2v h = {(_Float16)c0, (_Float16)c1};
    hi = __builtin_bit_cast(uint32_t, h);
    const float m1 = -1.0f;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "s"(m1), "v"(c0));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "s"(m1), "v"(c1));
}
// CLAMP = false only where the values are known to lie inside the fp16 range already
template <bool CLAMP = true>
__device__ __forceinline__ void cn_split4(cn_f32x4 v, cn_f16x4v &hi, cn_f16x4v &lo)
{
    float c[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
        // clamp to the fp16 range: an overflow would make the low part NaN (inf - inf)
        c[e] = CLAMP ? __builtin_fminf(__builtin_fmaxf(v[e], -65504.0f), 65504.0f) : v[e];
    uint32_t h[2], l[2];
    cn_split2_bits(c[0], c[1], h[0], l[0]);
    cn_split2_bits(c[2], c[3], h[1], l[1]);
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    hi = __builtin_bit_cast(cn_f16x4v, u32x2{h[0], h[1]});
    lo = __builtin_bit_cast(cn_f16x4v, u32x2{l[0], l[1]});
}
// The logistic of the detectors (hm.sigmoid_(), detectors/ctdet.py:31, multi_pose.py:33-35): ONE definition
// for every kernel that applies it (all decode forms, the flip-test average), so that the paths stay
// bit-identical among themselves: 1 / (1 + exp(-x)) as v_exp_f32 + v_rcp_f32 (1 ulp each), within 2e-7
// of torch's value.  The reciprocal is the hardware approximation, not the IEEE division (11 instructions
// per cell, the largest single item of the one-launch decode).
__device__ __forceinline__ float sigmoidf_ref(float x)
{
    return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}

// ---- range words: running max |v| of everything a launch split, as float bit patterns
// (non-negative floats order like unsigned integers).  One conditional atomic per wave, spread
// over CN_RANGE_SLOTS words in separate 64-byte lines per side: the waves of a launch finish in
// bursts (a whole dispatch round at once), and thousands of atomics on ONE address serialise
// at ~12 ns each (measured: +30 us on a 38 us launch); cn_range_fold reduces the slots.
__device__ __forceinline__ void cn_rng_upd4(float &m, cn_f32x4 v)
{
    m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fmaxf(__builtin_fabsf(v[0]), __builtin_fabsf(v[1]))),
                        __builtin_fmaxf(__builtin_fabsf(v[2]), __builtin_fabsf(v[3])));
}
__device__ __forceinline__ void cn_rng_upd1(float &m, float v) { m = __builtin_fmaxf(m, __builtin_fabsf(v)); }
// the same, NaN-sticky: for the sites where USER data enters (the network input in the stem kernels,
// the plain -> f32s converter).  fmaxf drops a NaN and the clamp in front of the split turns it into
// -65504, so a NaN image would give finite garbage where the reference propagates NaN; here it makes
// the range word read +inf: the forward is reported as invalid and the re-calibration that follows
// refuses non-finite activations loudly.  (Inside the network values stay finite: every split site
// clamps, so a NaN cannot be produced from finite inputs and weights.)
__device__ __forceinline__ void cn_rng_upd1_in(float &m, float v)
{
    m = (v == v) ? __builtin_fmaxf(m, __builtin_fabsf(v)) : __builtin_inff();
}
// wave-wide maximum of non-negative floats, returned (uniform) as a bit pattern: integer max
// over the bit patterns -- four DPP steps inside each row of 16 lanes, then the four rows
// through SGPRs.  ~12 instructions, no LDS traffic (a __shfl_xor ladder is six dependent
// ds_bpermute round trips at the very end of every workgroup).  All 64 lanes must be active.
__device__ __forceinline__ uint32_t cn_wave_max_bits(float m)
{
    int v = __builtin_bit_cast(int, m);
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false));   // row_ror:4
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false));   // row_ror:8
    const int r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
    const int r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
    return (uint32_t)max(max(r0, r1), max(r2, r3));
}
// single-word form (cn_absmax_f32)
__device__ __forceinline__ void cn_rng_commit1(uint32_t *word, float m)
{
    const uint32_t bits = cn_wave_max_bits(m);
    if ((threadIdx.x & 63) == 0 && bits) atomicMax(word, bits);
}
// call with the whole wave active (no early-returned lanes); `range` may be null;
// side 0 = output side, 1 = input side (cn_f32s_ctl.range).  The atomic returns nothing: the
// wave does not wait for it.
__device__ __forceinline__ void cn_rng_commit(uint32_t *range, int side, float m)
{
    if (!range) return;   // uniform
    const uint32_t bits = cn_wave_max_bits(m);
    if ((threadIdx.x & 63) == 0 && bits) {
        const unsigned slot = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6) + blockIdx.y * 7u +
                               blockIdx.z * 13u) & (CN_RANGE_SLOTS - 1);
        atomicMax(range + ((size_t)side * CN_RANGE_SLOTS + slot) * CN_RANGE_STRIDE, bits);
    }
}
__device__ __forceinline__ cn_f32x4 cn_join4(cn_f16x4v hi, cn_f16x4v lo)
{
    cn_f32x4 r = {(float)hi[0] + (float)lo[0], (float)hi[1] + (float)lo[1],
                  (float)hi[2] + (float)lo[2], (float)hi[3] + (float)lo[3]};
    return r;
}
// 4 consecutive channels n..n+3 (n % 4 == 0) of pixel `pix` of an f32s tensor
__device__ __forceinline__ void cn_store4_f32s(void *base, size_t pix, int pitch, int n, cn_f32x4 v)
{
    char *g = reinterpret_cast<char *>(base) + (pix * (size_t)pitch + (size_t)(n & ~31)) * 4;
    cn_f16x4v hi, lo;
    cn_split4(v, hi, lo);
    *reinterpret_cast<cn_f16x4v *>(g + (n & 31) * 2) = hi;
    *reinterpret_cast<cn_f16x4v *>(g + 64 + (n & 31) * 2) = lo;
}
__device__ __forceinline__ cn_f32x4 cn_load4_f32s(const void *base, size_t pix, int pitch, int n)
{
    const char *g = reinterpret_cast<const char *>(base) + (pix * (size_t)pitch + (size_t)(n & ~31)) * 4;
    return cn_join4(*reinterpret_cast<const cn_f16x4v *>(g + (n & 31) * 2),
                    *reinterpret_cast<const cn_f16x4v *>(g + 64 + (n & 31) * 2));
}
__device__ __forceinline__ void cn_store1_f32s(void *base, size_t pix, int pitch, int n, float v)
{
    char *g = reinterpret_cast<char *>(base) + (pix * (size_t)pitch + (size_t)(n & ~31)) * 4;
    const float c = __builtin_fminf(__builtin_fmaxf(v, -65504.0f), 65504.0f);
    const _Float16 hi = (_Float16)c;
    *reinterpret_cast<_Float16 *>(g + (n & 31) * 2) = hi;
    *reinterpret_cast<_Float16 *>(g + 64 + (n & 31) * 2) = (_Float16)(c - (float)hi);
}
__device__ __forceinline__ float cn_load1_f32s(const void *base, size_t pix, int pitch, int n)
{
    const char *g = reinterpret_cast<const char *>(base) + (pix * (size_t)pitch + (size_t)(n & ~31)) * 4;
    return (float)*reinterpret_cast<const _Float16 *>(g + (n & 31) * 2) +
           (float)*reinterpret_cast<const _Float16 *>(g + 64 + (n & 31) * 2);
}
