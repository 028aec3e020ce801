// cn_decode.hip -- heat-map decode for gfx950: sigmoid -> 3x3 peak test ->
// exact top-K -> gather -> box assembly.
//
// Reference behaviour being replaced (paths relative to /root/reference/src/lib):
//   hm.sigmoid_()                      detectors/ctdet.py:31
//   _nms                               models/decode.py:9-15
//   _topk_channel / _topk              models/decode.py:92-119
//   _transpose_and_gather_feat         models/utils.py:22-26
//   ctdet_decode                       models/decode.py:464-495
//
// Design (HBM-bound: the heat-map is read exactly once):
//   kernel 1  nms_topk_kernel   one workgroup per (image, class, row band).  The
//             band (+1 halo row each side) is staged in LDS after the sigmoid,
//             every thread keeps its peak-tested values in registers, and an
//             exact wave/LDS radix select over a 64-bit key (score, ~index)
//             leaves the band's K best peaks, sorted, as candidates.
//   kernel 2  merge_topk_kernel one workgroup per image: the same radix select
//             over the C*bands*K candidates (== torch's second topk over C*K,
//             decode.py:112), then each of the K winners gathers wh/reg straight
//             from the NCHW maps (no permute().contiguous() copy, utils.py:23)
//             and writes its box.
//
// Order of equal scores (torch.topk leaves it unspecified): score descending,
// then class ascending, then spatial index ascending -- the key below makes
// that a strict total order, so the result is deterministic.
#include "cn_common.h"

// Decode arithmetic must round exactly like the reference's torch ops (bit-exact parity):
// no mul+add fusion anywhere in this file (hipcc's default is -ffp-contract=fast-honor-pragmas
// and HIP's __fmul_rn/__fadd_rn are plain operators).
#pragma clang fp contract(off)
// Rounded-once helpers compiled under contract(off): unlike HIP's header-inline __fmul_rn /
// __fadd_rn their results can never be fused into an FMA after inlining.
static __device__ __forceinline__ float mul_rn(float a, float b) { return a * b; }
static __device__ __forceinline__ float add_rn(float a, float b) { return a + b; }

namespace {

constexpr int NT = 256;      // threads per workgroup (4 waves)
constexpr int KMAX = 128;    // largest supported K
constexpr int HBINS = 2048;  // radix-select histogram bins (11-bit digits)

typedef unsigned long long u64;

// Order-preserving float -> uint32 map (larger float -> larger key).
__device__ __forceinline__ uint32_t f2key(float v)
{
    const uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k)
{
    const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}
constexpr uint32_t KEY_ZERO = 0x80000000u;  // f2key(+0.0f)

constexpr int MAXW = 16;  // waves of the largest workgroup that uses SelShared (1024 threads)

struct SelShared {
    uint32_t hist[HBINS];
    u64 sel[KMAX];
    uint32_t wsum[MAXW];
    uint32_t cnt;
    uint32_t digit, need, bincount;
    uint32_t ncand;
    uint32_t pad_[3];
};
static_assert(sizeof(SelShared) % 16 == 0, "keep the LDS carve 16-byte aligned");

__device__ __forceinline__ int pass_shift(int p)
{
    // 64-bit key split into digits of 11,11,10 | 11,11,10 bits (msb first)
    return p == 0 ? 53 : p == 1 ? 42 : p == 2 ? 32 : p == 3 ? 21 : p == 4 ? 10 : 0;
}
__device__ __forceinline__ uint32_t pass_dmask(int p) { return (p == 2 || p == 5) ? 1023u : 2047u; }

// Exact selection of the `need0` largest 64-bit keys among the elements that
// `for_each(f)` enumerates (f(key64, is_plain_zero)); keys must be distinct.
// On return every thread holds (prefix, mask): an element is selected iff
// (key & mask) >= prefix.  Elements whose score is exactly +0.0 (the ~89 % of a
// heat-map that the peak test suppressed) are counted per wave instead of one
// LDS atomic each, otherwise they would serialise on a single histogram bin.
template <int TB, class ForEach>
__device__ __forceinline__ void radix_select(ForEach &&for_each, uint32_t need0, SelShared &sh,
                                             u64 &out_prefix, u64 &out_mask)
{
    static_assert(HBINS % TB == 0 && TB / CN_WAVE <= MAXW, "block size");
    constexpr int BPT = HBINS / TB;  // histogram bins owned by a thread
    const int tid = threadIdx.x;
    const int lane = tid & (CN_WAVE - 1);
    const int wave = tid / CN_WAVE;
    u64 prefix = 0, mask = 0;
    uint32_t need = need0;
    const u64 zkey = (u64)KEY_ZERO << 32;
#pragma unroll 1
    for (int pass = 0; pass < 6; ++pass) {
        const int shift = pass_shift(pass);
        const uint32_t dmask = pass_dmask(pass);
        for (int i = tid; i < HBINS; i += TB) sh.hist[i] = 0;
        __syncthreads();
        uint32_t zc = 0;
        for_each([&](u64 k, bool plain_zero) {
            if ((k & mask) == prefix) {
                if (plain_zero && pass < 3)
                    ++zc;
                else
                    atomicAdd(&sh.hist[(uint32_t)(k >> shift) & dmask], 1u);
            }
        });
        if (pass < 3) {
            for (int o = CN_WAVE / 2; o > 0; o >>= 1) zc += __shfl_down(zc, o);
            if (lane == 0 && zc) atomicAdd(&sh.hist[(uint32_t)(zkey >> shift) & dmask], zc);
        }
        __syncthreads();
        // suffix sums over bins: thread t owns bins [BPT*t, BPT*t + BPT)
        uint32_t p = 0;
#pragma unroll
        for (int j = 0; j < BPT; ++j) p += sh.hist[tid * BPT + j];
        uint32_t s = p;
#pragma unroll
        for (int o = 1; o < CN_WAVE; o <<= 1) {
            const uint32_t t = __shfl_down(s, o);
            if (lane + o < CN_WAVE) s += t;
        }
        if (lane == 0) sh.wsum[wave] = s;
        __syncthreads();
        for (int w = wave + 1; w < TB / CN_WAVE; ++w) s += sh.wsum[w];
        const uint32_t above = s - p;  // elements in strictly higher bins
        if (above < need && s >= need) {
            uint32_t run = above;
            for (int j = BPT - 1; j >= 0; --j) {
                const uint32_t h = sh.hist[tid * BPT + j];
                if (run + h >= need) {
                    sh.digit = tid * BPT + j;
                    sh.need = need - run;
                    sh.bincount = h;
                    break;
                }
                run += h;
            }
        }
        __syncthreads();
        prefix |= (u64)sh.digit << shift;
        mask |= (u64)dmask << shift;
        need = sh.need;
        if (sh.bincount == need) break;  // the whole bin is taken: done
    }
    out_prefix = prefix;
    out_mask = mask;
}

// Collect the selected keys into sh.sel[0..KMAX) and sort them descending.  The sort is a
// 128-element bitonic network run by ONE wave (two keys per lane, cross-lane exchange by
// shuffles): no workgroup barriers inside the network.
template <int TB, class ForEach>
__device__ __forceinline__ void collect_and_sort(ForEach &&for_each, u64 prefix, u64 mask,
                                                 SelShared &sh)
{
    const int tid = threadIdx.x;
    __syncthreads();
    if (tid == 0) sh.cnt = 0;
    for (int i = tid; i < KMAX; i += TB) sh.sel[i] = 0;  // pad keys sort last
    __syncthreads();
    for_each([&](u64 k, bool) {
        if ((k & mask) >= prefix) {
            const uint32_t pos = atomicAdd(&sh.cnt, 1u);
            if (pos < (uint32_t)KMAX) sh.sel[pos] = k;
        }
    });
    __syncthreads();
    if (tid < CN_WAVE) {
        const int lane = tid;
        u64 v[2] = {sh.sel[lane], sh.sel[lane + CN_WAVE]};
        for (int k = 2; k <= KMAX; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                if (j == CN_WAVE) {  // partner = the lane's other slot (only when k == 128)
                    const u64 hi = v[0] > v[1] ? v[0] : v[1];
                    const u64 lo = v[0] > v[1] ? v[1] : v[0];
                    v[0] = hi;
                    v[1] = lo;
                } else {
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const int i = s * CN_WAVE + lane;
                        const uint32_t olo = __shfl_xor((uint32_t)v[s], j);
                        const uint32_t ohi = __shfl_xor((uint32_t)(v[s] >> 32), j);
                        const u64 o = ((u64)ohi << 32) | olo;
                        const bool desc = (i & k) == 0;
                        const bool lower = (i & j) == 0;
                        const bool take_max = (lower == desc);
                        v[s] = take_max ? (v[s] > o ? v[s] : o) : (v[s] > o ? o : v[s]);
                    }
                }
            }
        }
        sh.sel[lane] = v[0];
        sh.sel[lane + CN_WAVE] = v[1];
    }
    __syncthreads();
}

// (sigmoidf_ref: cn_common.h -- one definition of the logistic for every decode form and the flip average)

constexpr int CAND_CAP = 4096;  // compacted positive peaks per band (uint16 tile offsets)

// ---------------------------------------------------------------------------
// kernel 1: per (image, class, row-band): sigmoid + 3x3 peak test + top-K
// grid (nbands, C, B), block NT, dynamic LDS = SelShared | tile (R+2)*W floats | peak list
//
// Fast path: the peak test (four cells per step, horizontal 3-max shared between the three
// rows) appends the tile offsets of the POSITIVE peaks -- about a ninth of the cells -- to a
// compact LDS list with one wave-aggregated atomic per step; the exact radix select then
// touches only that list.  If a band holds fewer than K positive peaks, or more than the
// list can hold (constant maps, tiny maps, negative "heat"), the kernel falls back to a
// full scan that re-evaluates the peak test per element, so every input is still exact.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void nms_topk_kernel(const float *__restrict__ heat, int C, int H,
                                                      int W, int K, int R, int apply_sigmoid,
                                                      float *__restrict__ cand_score,
                                                      int32_t *__restrict__ cand_idx)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    SelShared &sh = *reinterpret_cast<SelShared *>(smem);
    float *tile = reinterpret_cast<float *>(smem + sizeof(SelShared));

    const int tid = threadIdx.x;
    const int lane = tid & (CN_WAVE - 1);
    const int band = blockIdx.x, c = blockIdx.y, b = blockIdx.z;
    const int nbands = gridDim.x;
    const int r0 = band * R;
    const int rows = min(R, H - r0);  // rows of this band
    const int n_band = rows * W;
    const int trows = rows + 2;
    uint16_t *clist = reinterpret_cast<uint16_t *>(tile + (size_t)(R + 2) * W);
    const float *plane = heat + ((size_t)b * C + c) * (size_t)H * W;
    const float NEG_INF = -__builtin_huge_valf();
    // ablation bits for tools/bench_decode.py (never set through the Python API)
    const bool dbg_noselect = (apply_sigmoid & 256) != 0;
    const bool dbg_nonms = (apply_sigmoid & CN_DECODE_NO_PEAK_TEST) != 0;  // plain topk
    const bool dbg_slow = (apply_sigmoid & 1024) != 0;
    apply_sigmoid &= 1;
    if (tid == 0) sh.ncand = 0;

    // ---- stage rows r0-1 .. r0+rows (rows+2 rows) into LDS, sigmoid applied once
    const bool vec = (W & 3) == 0;
    if (vec) {
        const int w4 = W >> 2;
        const int nq = trows * w4;
        constexpr int U = 8;  // global loads in flight per thread before the first use
        for (int q0 = tid; q0 < nq; q0 += U * NT) {
            cn_f32x4 v[U];
            int dst[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = q0 + u * NT;
                dst[u] = -1;
                v[u].x = v[u].y = v[u].z = v[u].w = NEG_INF;
                if (q < nq) {
                    const int tr = q / w4;
                    const int c4 = q - tr * w4;
                    const int gy = r0 - 1 + tr;
                    dst[u] = tr * W + c4 * 4;
                    if (gy >= 0 && gy < H) {
                        v[u] = *reinterpret_cast<const cn_f32x4 *>(plane + (size_t)gy * W + c4 * 4);
                    } else {
                        dst[u] |= 0x40000000;  // halo row outside the image: keep -inf
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (dst[u] < 0) continue;
                cn_f32x4 t = v[u];
                if (apply_sigmoid && !(dst[u] & 0x40000000)) {
                    t.x = sigmoidf_ref(t.x);
                    t.y = sigmoidf_ref(t.y);
                    t.z = sigmoidf_ref(t.z);
                    t.w = sigmoidf_ref(t.w);
                }
                *reinterpret_cast<cn_f32x4 *>(tile + (dst[u] & 0x3FFFFFFF)) = t;
            }
        }
    } else {
        const int ne = trows * W;
        for (int q = tid; q < ne; q += NT) {
            const int tr = q / W;
            const int x = q - tr * W;
            const int gy = r0 - 1 + tr;
            float v = NEG_INF;
            if (gy >= 0 && gy < H) {
                v = plane[(size_t)gy * W + x];
                if (apply_sigmoid) v = sigmoidf_ref(v);
            }
            tile[tr * W + x] = v;
        }
    }
    __syncthreads();

    // peak test of one cell (decode.py:9-15); t = offset of the cell in the tile
    auto nms_val = [&](int t, int x) -> float {
        const float *row = tile + t;
        const float v = row[0];
        float m = fmaxf(fmaxf(row[-W], row[W]), v);
        if (x > 0) m = fmaxf(fmaxf(m, row[-1]), fmaxf(row[-W - 1], row[W - 1]));
        if (x < W - 1) m = fmaxf(fmaxf(m, row[1]), fmaxf(row[-W + 1], row[W + 1]));
        if (dbg_nonms) m = v;
        return ((m == v) ? v : 0.0f) + 0.0f;  // heat * keep, -0.0 -> +0.0
    };
    // append the tile offset of a positive peak; one LDS atomic per wave per call
    auto append = [&](bool is, int t) {
        const u64 bal = __ballot(is);
        if (bal) {
            const int leader = __ffsll((long long)bal) - 1;
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(&sh.ncand, (uint32_t)__popcll(bal));
            base = __shfl(base, leader);
            if (is) {
                const uint32_t pos =
                    base + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32),
                                                     __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                if (pos < (uint32_t)CAND_CAP) clist[pos] = (uint16_t)t;
            }
        }
    };

    if (vec) {
        const int w4 = W >> 2;
        const int nq = rows * w4;
        const int nqr = (nq + NT - 1) / NT * NT;  // whole waves take part in the ballots
        for (int q = tid; q < nqr; q += NT) {
            bool pk[4] = {false, false, false, false};
            int t0 = 0;
            if (q < nq) {
                const int y = q / w4;
                const int x4 = q - y * w4;
                t0 = (y + 1) * W + x4 * 4;
                float hm[4] = {NEG_INF, NEG_INF, NEG_INF, NEG_INF};
                cn_f32x4 ctr = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int dr = -1; dr <= 1; ++dr) {
                    const float *row = tile + t0 + dr * W;
                    const cn_f32x4 cv = *reinterpret_cast<const cn_f32x4 *>(row);
                    const float l = (x4 > 0) ? row[-1] : NEG_INF;
                    const float r = (x4 < w4 - 1) ? row[4] : NEG_INF;
                    hm[0] = fmaxf(hm[0], fmaxf(fmaxf(l, cv.x), cv.y));
                    hm[1] = fmaxf(hm[1], fmaxf(fmaxf(cv.x, cv.y), cv.z));
                    hm[2] = fmaxf(hm[2], fmaxf(fmaxf(cv.y, cv.z), cv.w));
                    hm[3] = fmaxf(hm[3], fmaxf(fmaxf(cv.z, cv.w), r));
                    if (dr == 0) ctr = cv;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) pk[e] = (dbg_nonms || hm[e] == ctr[e]) && ctr[e] > 0.0f;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) append(pk[e], t0 + e);
        }
    } else {
        const int nr = (n_band + NT - 1) / NT * NT;
        for (int e = tid; e < nr; e += NT) {
            bool is = false;
            int t = 0;
            if (e < n_band) {
                const int y = e / W, x = e - y * W;
                t = (y + 1) * W + x;
                is = nms_val(t, x) > 0.0f;
            }
            append(is, t);
        }
    }
    __syncthreads();

    const uint32_t base = (uint32_t)(r0 * W);
    const int kb = min(K, n_band);
    const uint32_t npos = sh.ncand;
    const bool fast = !dbg_slow && npos >= (uint32_t)kb && npos <= (uint32_t)CAND_CAP;
    u64 prefix, mask;
    if (dbg_noselect) {
        if (tid < KMAX) sh.sel[tid] = npos;
        __syncthreads();
    } else if (fast) {
        auto for_each = [&](auto &&f) {
            for (uint32_t j = tid; j < npos; j += NT) {
                const uint32_t t = clist[j];
                f(((u64)f2key(tile[t]) << 32) | (u64)(0xFFFFFFFFu - (base + t - (uint32_t)W)), false);
            }
        };
        radix_select<NT>(for_each, (uint32_t)kb, sh, prefix, mask);
        collect_and_sort<NT>(for_each, prefix, mask, sh);
    } else {
        // exact fallback: every cell takes part (zeros from suppressed cells included)
        auto for_each = [&](auto &&f) {
            for (int e = tid; e < n_band; e += NT) {
                const int y = e / W, x = e - y * W;
                const uint32_t kk = f2key(nms_val((y + 1) * W + x, x));
                f(((u64)kk << 32) | (u64)(0xFFFFFFFFu - (base + (uint32_t)e)), kk == KEY_ZERO);
            }
        };
        radix_select<NT>(for_each, (uint32_t)kb, sh, prefix, mask);
        collect_and_sort<NT>(for_each, prefix, mask, sh);
    }

    if (tid < K) {
        const size_t o = ((((size_t)b * C + c) * nbands) + band) * K + tid;
        if (tid < kb) {
            const u64 k = sh.sel[tid];
            cand_score[o] = key2f((uint32_t)(k >> 32));
            cand_idx[o] = (int32_t)(0xFFFFFFFFu - (uint32_t)k);
        } else {
            cand_score[o] = NEG_INF;
            cand_idx[o] = -1;
        }
    }
}

// ---------------------------------------------------------------------------
// Image-level top-K without a per-band select (ctdet / _topk: the K best peaks of ALL classes of
// an image, decode.py:103-119).  Only ~K of the C*H*W cells matter, so the exact select runs on
// a threshold-pruned candidate list instead of in every (class, band) -- in TWO launches:
//   launch 1  group_max_kernel: ONE streaming pass over the raw map (logits when the sigmoid is
//             fused: the logistic is monotone, no transcendental per cell).  A GROUP is 8 rows x 128
//             columns of a plane; per group it records the largest raw value of all cells and the
//             largest raw value among the cells that pass the 3x3 peak test on RAW values (a raw
//             peak is also a peak of the sigmoid values).
//   launch 2  collect_merge_kernel, one workgroup per plane:
//             image_threshold: T_b = K-th largest group peak-maximum of image b (as a score),
//               computed by every plane of the image from the same ~1 k keys (cheaper than a launch
//               boundary, and no workgroup ever waits for another).  At least K distinct peaks reach
//               T_b, so every cell of the exact top-K has a peak value >= T_b -- ties at T_b
//               included -- and nothing below can be in it.
//             plane_collect: only the groups whose all-cell maximum reaches T_b are looked at again
//               (L2 / Infinity-Cache resident); their cells get the EXACT treatment -- sigmoid, 3x3
//               equality test on sigmoid values (decode.py:9-15) -- and are appended to the image's
//               candidate list as 64-bit keys (score, ~flat index).
//             the image's LAST plane to arrive (one atomic counter per image): exact radix select +
//               sort of the few hundred candidates, gather, box assembly.
// A plane hands on at most its own K best keys (exact select inside the plane when it holds more:
// saturated plateaus), so the image's list of C * K keys cannot overflow; images with T_b <= 0 (fewer
// than K groups with a positive peak; constant maps: zeros of suppressed cells take part) run that
// per-plane select over all cells of every plane.  Every input is handled exactly by the same two
// launches; both forms are bit-identical to the per-(class, band) select of round 1.
// ---------------------------------------------------------------------------

// A GROUP = 8 rows x 128 columns of one (image, class) plane.  One half-wave owns a 16-row x
// 128-column unit (two groups): lane l holds the 4-cell quad l of a row and walks down the
// rows with the previous / current / next row in registers, so every cell is loaded from global
// memory exactly once (18 row loads for 16 rows, all issued up front), horizontal neighbours come
// from the adjacent lanes by shuffle, and nothing goes through LDS.
constexpr int GROWS = 8;     // rows per group
constexpr int GUNIT = 16;    // rows per half-wave unit

__global__ __launch_bounds__(NT) void group_max_kernel(const float *__restrict__ heat, int H, int W,
                                                       int nrg, int ncb, int flags,
                                                       uint32_t *__restrict__ gpeak,
                                                       uint32_t *__restrict__ gall, int C,
                                                       int32_t *__restrict__ counts,
                                                       int32_t *__restrict__ done)
{
    const int tid = threadIdx.x;
    // the second launch's per-image candidate count and arrival counter start at zero (stream order)
    if (tid == 0 && blockIdx.x % (unsigned)C == 0) {
        counts[blockIdx.x / (unsigned)C] = 0;
        done[blockIdx.x / (unsigned)C] = 0;
    }
    const int lane = tid & (CN_WAVE - 1);
    const int hl = lane & 31;                 // lane inside the half-wave
    const int hw = tid >> 5;                  // half-wave of the workgroup (0..7)
    const int nunit_r = (nrg + 1) / 2;        // 16-row units per plane
    const int units = nunit_r * ncb;          // units per plane
    const int w4 = W >> 2;
    const float NEG_INF = -__builtin_huge_valf();
    const bool nonms = (flags & CN_DECODE_NO_PEAK_TEST) != 0;
    const size_t plane_id = blockIdx.x;       // b * C + c
    const float *plane = heat + plane_id * (size_t)H * W;
    for (int u = hw; u < units; u += NT / 32) {   // uniform per half-wave
        const int ur = u / ncb, cb = u - ur * ncb;
        const int y0 = ur * GUNIT;
        const int x4 = cb * 32 + hl;              // this lane's quad column
        const bool col_ok = x4 < w4;
        // rows y0-1 .. y0+16: all loads issued before the first use
        cn_f32x4 rowv[GUNIT + 2];
        float lft[GUNIT + 2], rgt[GUNIT + 2];     // cells just outside the 128-column block
#pragma unroll
        for (int r = 0; r < GUNIT + 2; ++r) {
            const int y = y0 - 1 + r;
            const bool ok = col_ok && y >= 0 && y < H;
            rowv[r].x = rowv[r].y = rowv[r].z = rowv[r].w = NEG_INF;
            lft[r] = rgt[r] = NEG_INF;
            if (ok) {
                const float *p = plane + (size_t)y * W + x4 * 4;
                rowv[r] = *reinterpret_cast<const cn_f32x4 *>(p);
                if (hl == 0 && x4 > 0) lft[r] = p[-1];
                if (hl == 31 && x4 + 1 < w4) rgt[r] = p[4];
            }
        }
        // horizontal 3-max of every row, then the vertical combination
        cn_f32x4 hmax[GUNIT + 2];
#pragma unroll
        for (int r = 0; r < GUNIT + 2; ++r) {
            float l = __shfl_up(rowv[r].w, 1, 32);
            float rr = __shfl_down(rowv[r].x, 1, 32);
            if (hl == 0) l = lft[r];
            if (hl == 31) rr = rgt[r];
            if (!col_ok) { l = NEG_INF; rr = NEG_INF; }
            // a lane beyond the map's last quad hands -inf to its left neighbour
            hmax[r].x = fmaxf(fmaxf(l, rowv[r].x), rowv[r].y);
            hmax[r].y = fmaxf(fmaxf(rowv[r].x, rowv[r].y), rowv[r].z);
            hmax[r].z = fmaxf(fmaxf(rowv[r].y, rowv[r].z), rowv[r].w);
            hmax[r].w = fmaxf(fmaxf(rowv[r].z, rowv[r].w), rr);
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            uint32_t kp = 0u, ka = 0u;
#pragma unroll
            for (int r = 1 + half * GROWS; r < 1 + (half + 1) * GROWS; ++r) {
                const int y = y0 - 1 + r;
                if (!(col_ok && y < H)) continue;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = rowv[r][e];
                    const float m = fmaxf(fmaxf(hmax[r - 1][e], hmax[r][e]), hmax[r + 1][e]);
                    const uint32_t k = f2key(v + 0.0f);
                    ka = max(ka, k);
                    if (nonms || m == v) kp = max(kp, k);
                }
            }
            for (int o = 16; o > 0; o >>= 1) {
                kp = max(kp, (uint32_t)__shfl_xor((int)kp, o, 32));
                ka = max(ka, (uint32_t)__shfl_xor((int)ka, o, 32));
            }
            const int rg = ur * 2 + half;
            if (hl == 0 && rg < nrg) {
                const size_t gi = (plane_id * nrg + rg) * ncb + cb;
                gpeak[gi] = kp;
                gall[gi] = ka;
            }
        }
    }
}

// phase 2 (device function, run by EVERY plane's workgroup of the second launch -- the select over
// an image's ~1 k group maxima is cheaper than a launch boundary and needs no inter-workgroup
// wait): the K-th largest group peak-maximum of image b (exact).  Returns, to every thread, the
// SCORE key of the threshold (the logistic applied when it is fused; <= KEY_ZERO = degenerate
// image: fewer than K groups with a peak, or a non-positive threshold -- zeros of suppressed cells
// would take part) and a conservative lower bound, in raw units, of every cell whose score can
// reach it.
// the `need`-th largest of the 32-bit keys `for_each(f)` enumerates (f(key32)), exactly: three
// digit passes (11, 11, 10 bits, msb first) over a histogram in LDS, no early exit, no sort
// The bin of sh.hist that holds the `need`-th largest element (bins counted from the top): sh.digit =
// that bin, sh.need = the rank still to find inside it, sh.bincount = elements in that bin and above.
// All threads call it behind a barrier that closes the histogram; it ends with a barrier.
template <int TB>
__device__ __forceinline__ void suffix_bin(uint32_t need, SelShared &sh)
{
    constexpr int BPT = HBINS / TB;
    const int tid = threadIdx.x;
    const int lane = tid & (CN_WAVE - 1), wave = tid / CN_WAVE;
    uint32_t p = 0;
#pragma unroll
    for (int j = 0; j < BPT; ++j) p += sh.hist[tid * BPT + j];
    uint32_t s = p;
#pragma unroll
    for (int o = 1; o < CN_WAVE; o <<= 1) {
        const uint32_t t = __shfl_down(s, o);
        if (lane + o < CN_WAVE) s += t;
    }
    if (lane == 0) sh.wsum[wave] = s;
    __syncthreads();
    for (int w = wave + 1; w < TB / CN_WAVE; ++w) s += sh.wsum[w];
    const uint32_t above = s - p;  // elements in strictly higher bins
    if (above < need && s >= need) {
        uint32_t run = above;
        for (int j = BPT - 1; j >= 0; --j) {
            const uint32_t h = sh.hist[tid * BPT + j];
            if (run + h >= need) {
                sh.digit = tid * BPT + j;
                sh.need = need - run;
                sh.bincount = run + h;
                break;
            }
            run += h;
        }
    }
    __syncthreads();
}

template <int TB, class ForEach>
__device__ __forceinline__ uint32_t kth_largest_key32(ForEach &&for_each, uint32_t need, SelShared &sh)
{
    constexpr int BPT = HBINS / TB;
    const int tid = threadIdx.x;
    const int lane = tid & (CN_WAVE - 1), wave = tid / CN_WAVE;
    uint32_t prefix = 0, mask = 0;
#pragma unroll 1
    for (int pass = 0; pass < 3; ++pass) {
        const int shift = pass == 0 ? 21 : pass == 1 ? 10 : 0;
        const uint32_t dmask = pass == 2 ? 1023u : 2047u;
        for (int i = tid; i < HBINS; i += TB) sh.hist[i] = 0;
        __syncthreads();
        for_each([&](uint32_t k) {
            if ((k & mask) == prefix) atomicAdd(&sh.hist[(k >> shift) & dmask], 1u);
        });
        __syncthreads();
        uint32_t p = 0;
#pragma unroll
        for (int j = 0; j < BPT; ++j) p += sh.hist[tid * BPT + j];
        uint32_t s = p;
#pragma unroll
        for (int o = 1; o < CN_WAVE; o <<= 1) {
            const uint32_t t = __shfl_down(s, o);
            if (lane + o < CN_WAVE) s += t;
        }
        if (lane == 0) sh.wsum[wave] = s;
        __syncthreads();
        for (int w = wave + 1; w < TB / CN_WAVE; ++w) s += sh.wsum[w];
        const uint32_t above = s - p;  // keys in strictly higher bins
        if (above < need && s >= need) {
            uint32_t run = above;
            for (int j = BPT - 1; j >= 0; --j) {
                const uint32_t h = sh.hist[tid * BPT + j];
                if (run + h >= need) {
                    sh.digit = tid * BPT + j;
                    sh.need = need - run;
                    break;
                }
                run += h;
            }
        }
        __syncthreads();
        prefix |= sh.digit << shift;
        mask |= dmask << shift;
        need = sh.need;
        __syncthreads();       // (sh.digit / sh.need are rewritten by the next pass)
    }
    return prefix;
}

__device__ __forceinline__ uint32_t image_threshold(const uint32_t *__restrict__ g, int ng, int K, int flags,
                                                    SelShared &sh, float &rawthr)
{
    const int tid = threadIdx.x;
    rawthr = 0.f;
    if (ng < K) return 0u;
    // the image's group maxima: fetched ONCE into registers (a select pass over global memory is a
    // load round trip per pass: ~10 us per workgroup, and every plane's workgroup does this)
    constexpr int RK = 8;
    uint32_t kraw;
    if (ng <= RK * NT) {
        uint32_t kr[RK];
#pragma unroll
        for (int u = 0; u < RK; ++u) kr[u] = (tid + u * NT < ng) ? g[tid + u * NT] : 0u;
        auto for_each = [&](auto &&f) {
#pragma unroll
            for (int u = 0; u < RK; ++u)
                if (tid + u * NT < ng) f(kr[u]);
        };
        kraw = kth_largest_key32<NT>(for_each, (uint32_t)K, sh);
    } else {
        auto for_each = [&](auto &&f) {
            for (int j = tid; j < ng; j += NT) f(g[j]);
        };
        kraw = kth_largest_key32<NT>(for_each, (uint32_t)K, sh);
    }
    if (kraw == 0u) return 0u;    // fewer than K groups hold a peak at all
    const float raw = key2f(kraw);
    const bool sig = (flags & 1) != 0;
    const float score = sig ? sigmoidf_ref(raw) : raw;
    // the device logistic is monotone only up to its last bit: admit a margin of raw values
    // below the K-th one; the exact test in phase 3 sorts them out
    rawthr = sig ? raw - (1e-3f + 1e-3f * fabsf(raw)) : raw;
    return f2key(score + 0.0f);
}

// phase 3 (device function): the candidate keys of ONE plane.  The workgroup's eight half-waves share
// the plane's live groups (all-cell maximum >= the raw threshold).  A half-wave re-reads its 8 x 128
// group with the same rolling register window as phase 1 (10 row loads, neighbours by shuffle),
// applies the logistic to the whole window and runs the reference's test on the SCORES -- 3x3
// maximum, exact equality (decode.py:9-15) -- in registers; qualifying cells are counted per lane,
// placed by a half-wave prefix sum into a list in LDS.  The threshold often falls INTO the noise floor
// of a real heat-map (a few confident objects, K = 100): then most groups are live and this pass costs
// about what phase 1 does plus the logistics -- not nine dependent loads per cell over the threshold.
// A plane never has to hand on more than its own K best keys (the image's K best are among them), so
// the image's list -- capacity C * K -- cannot overflow whatever the map holds:
//   n <= K qualifying cells        -> all of them
//   K < n <= PLCAP                 -> exact select of the K best from the LDS list
//   n > PLCAP                      -> (class planes with a high bias hold most of an image's candidates)
//                                     the plane raises ITS threshold to the K-th best of the PLCAP keys
//                                     it did keep -- the plane's own K best all reach it -- and
//                                     collects again; only if that changes nothing (a plateau of more
//                                     than PLCAP equal scores)
//                                  -> exact select over the cells of the plane, keys recomputed per
//                                     pass from global memory (slow: hundreds of microseconds; rare)
// and a DEGENERATE image (threshold <= 0: fewer than K groups with a positive peak, constant maps --
// zeros of suppressed cells take part in the top K) takes the last route for every plane with all
// cells admitted: every input is handled exactly, by the same two launches.
constexpr int PLCAP = 1024;   // candidate keys of a plane kept in LDS

__device__ __forceinline__ void plane_collect(const float *__restrict__ heat, int C, int H, int W, int nrg,
                                              int ncb, int flags, int K, const uint32_t *__restrict__ gall,
                                              uint32_t tkey, float rthr, SelShared &sh, u64 *__restrict__ keys,
                                              int cap, int32_t *__restrict__ counts)
{
    __shared__ u64 pl_keys[PLCAP];
    __shared__ int pl_cnt, pl_base;
    const int tid = threadIdx.x;
    const int lane = tid & (CN_WAVE - 1);
    const int hl = lane & 31, hw = tid >> 5;
    const size_t plane_id = blockIdx.x;
    const int b = (int)(plane_id / C), c = (int)(plane_id - (size_t)b * C);
    const bool sig = (flags & 1) != 0;
    const bool nonms = (flags & CN_DECODE_NO_PEAK_TEST) != 0;
    const bool degenerate = tkey <= KEY_ZERO;
    const uint32_t rkey = f2key(rthr + 0.0f);   // group maxima are RAW keys
    const int HW = H * W;
    const float *plane = heat + plane_id * (size_t)HW;
    const uint32_t base = (uint32_t)c * (uint32_t)HW;
    u64 *kimg = keys + (size_t)b * cap;
    const int w4 = W >> 2;
    const float NEG_INF = -__builtin_huge_valf();
    uint32_t tcur = tkey;          // the plane's current threshold (raised when its list overflows)

    auto collect_group = [&](int g) {
        const int rg = g / ncb, cb = g - rg * ncb;
        const int y0 = rg * GROWS;
        const int x4 = cb * 32 + hl;
        const bool col_ok = x4 < w4;
        cn_f32x4 rowv[GROWS + 2];
        float lft[GROWS + 2], rgt[GROWS + 2];
#pragma unroll
        for (int r = 0; r < GROWS + 2; ++r) {
            const int y = y0 - 1 + r;
            const bool ok = col_ok && y >= 0 && y < H;
            rowv[r].x = rowv[r].y = rowv[r].z = rowv[r].w = NEG_INF;
            lft[r] = rgt[r] = NEG_INF;
            if (ok) {
                const float *p = plane + (size_t)y * W + x4 * 4;
                rowv[r] = *reinterpret_cast<const cn_f32x4 *>(p);
                if (hl == 0 && x4 > 0) lft[r] = p[-1];
                if (hl == 31 && x4 + 1 < w4) rgt[r] = p[4];
            }
        }
        if (sig) {   // scores of the whole window (cells outside the map stay -inf)
#pragma unroll
            for (int r = 0; r < GROWS + 2; ++r) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (rowv[r][e] != NEG_INF) rowv[r][e] = sigmoidf_ref(rowv[r][e]);
                if (lft[r] != NEG_INF) lft[r] = sigmoidf_ref(lft[r]);
                if (rgt[r] != NEG_INF) rgt[r] = sigmoidf_ref(rgt[r]);
            }
        }
        cn_f32x4 hmax[GROWS + 2];
#pragma unroll
        for (int r = 0; r < GROWS + 2; ++r) {
            float l = __shfl_up(rowv[r].w, 1, 32);
            float rr = __shfl_down(rowv[r].x, 1, 32);
            if (hl == 0) l = lft[r];
            if (hl == 31) rr = rgt[r];
            if (!col_ok) { l = NEG_INF; rr = NEG_INF; }
            hmax[r].x = fmaxf(fmaxf(l, rowv[r].x), rowv[r].y);
            hmax[r].y = fmaxf(fmaxf(rowv[r].x, rowv[r].y), rowv[r].z);
            hmax[r].z = fmaxf(fmaxf(rowv[r].y, rowv[r].z), rowv[r].w);
            hmax[r].w = fmaxf(fmaxf(rowv[r].z, rowv[r].w), rr);
        }
        // this lane's 32 cells: bit (r-1)*4+e set when the cell's key reaches the threshold
        auto cell_key = [&](int r, int e) -> uint32_t {
            const float v = rowv[r][e];
            const float m = fmaxf(fmaxf(hmax[r - 1][e], hmax[r][e]), hmax[r + 1][e]);
            return f2key(((nonms || m == v) ? v : 0.0f) + 0.0f);
        };
        uint32_t tmask = 0u;
#pragma unroll
        for (int r = 1; r <= GROWS; ++r) {
            const int y = y0 - 1 + r;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool take = col_ok && y < H && cell_key(r, e) >= tcur;
                tmask |= take ? (1u << ((r - 1) * 4 + e)) : 0u;
            }
        }
        // half-wave inclusive prefix sum of the per-lane counts, one LDS atomic for the group
        const int mine = __popc(tmask);
        int incl = mine;
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
            const int t = __shfl_up(incl, o, 32);
            if (hl >= o) incl += t;
        }
        const int total = __shfl(incl, 31, 32);
        if (total == 0) return;
        int pos0 = 0;
        if (hl == 31) pos0 = atomicAdd(&pl_cnt, total);
        int pos = __shfl(pos0, 31, 32) + incl - mine;
#pragma unroll
        for (int r = 1; r <= GROWS; ++r) {
            const int y = y0 - 1 + r;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (tmask & (1u << ((r - 1) * 4 + e))) {
                    const uint32_t cell = (uint32_t)(y * W + x4 * 4 + e);
                    if (pos < PLCAP)
                        pl_keys[pos] = ((u64)cell_key(r, e) << 32) | (u64)(0xFFFFFFFFu - (base + cell));
                    ++pos;
                }
            }
        }
    };
    bool plateau = false;
    if (!degenerate)
    for (int round = 0;; ++round) {
        if (tid == 0) pl_cnt = 0;
        __syncthreads();
        const int ng = nrg * ncb;
        // which groups can hold a qualifying cell: all group maxima are fetched at once (a chain of
        // dependent loads here cost more than the whole streaming pass)
        // and the live ones compacted into a list, so that the two halves of a wave always work on
        // two groups at the same time
        __shared__ u64 live[NT / 64];
        __shared__ int glist[NT];
        for (int g0 = 0; g0 < ng; g0 += NT) {
            const int g = g0 + tid;
            const bool q = g < ng && gall[plane_id * ng + g] >= rkey;
            const u64 bal = __ballot(q);
            if (lane == 0) live[tid >> 6] = bal;
            __syncthreads();
            int before = 0, nlive = 0;
#pragma unroll
            for (int w = 0; w < NT / 64; ++w) {
                const int n = __popcll(live[w]);
                if (w < (tid >> 6)) before += n;
                nlive += n;
            }
            if (q) glist[before + __popcll(bal & ((1ull << lane) - 1ull))] = g;
            __syncthreads();
            for (int i = hw; i < nlive; i += NT / 32) collect_group(glist[i]);
            __syncthreads();
        }
        if (pl_cnt <= PLCAP) break;            // (uniform: read behind the barrier)
        // more qualifying cells than the list holds: raise the plane's threshold to the K-th best of the
        // PLCAP keys that were kept (its own K best all reach that) and collect again
        auto kept = [&](auto &&f) {
            for (int j = tid; j < PLCAP; j += NT) f((uint32_t)(pl_keys[j] >> 32));
        };
        const uint32_t tnew = kth_largest_key32<NT>(kept, (uint32_t)K, sh);
        if (tnew == tcur || round == 3) {      // a plateau of equal scores (or no progress): exact select below
            plateau = true;
            break;
        }
        tcur = tnew;
    }
    // ---- hand the plane's keys on: at most its K best
    const int n = (degenerate || plateau) ? PLCAP + 1 : pl_cnt;      // (uniform; the last loop round ended with a barrier)
    int m;                                              // keys this plane emits
    const u64 *src;
    if (n <= K) {
        m = n;
        src = pl_keys;
    } else {
        m = min(K, HW);
        u64 prefix, mask;
        if (n <= PLCAP) {
            auto for_each = [&](auto &&f) {
                for (int j = tid; j < n; j += NT) f(pl_keys[j], false);
            };
            radix_select<NT>(for_each, (uint32_t)m, sh, prefix, mask);
            collect_and_sort<NT>(for_each, prefix, mask, sh);
        } else {
            // every cell of the plane (degenerate image), or every cell that reaches the threshold (more
            // of them than the LDS list holds): keys recomputed from global memory in every pass
            auto for_each = [&](auto &&f) {
                for (int e = tid; e < HW; e += NT) {
                    const int y = e / W, x = e - y * W;
                    auto sc = [&](int yy, int xx) {
                        const float v = plane[yy * W + xx];
                        return sig ? sigmoidf_ref(v) : v;
                    };
                    const float v = sc(y, x);
                    float mx = v;
                    if (!nonms) {
                        for (int dy = -1; dy <= 1; ++dy) {
                            const int yy = y + dy;
                            if (yy < 0 || yy >= H) continue;
                            for (int dx = -1; dx <= 1; ++dx) {
                                const int xx = x + dx;
                                if (xx < 0 || xx >= W || (dy == 0 && dx == 0)) continue;
                                mx = fmaxf(mx, sc(yy, xx));
                            }
                        }
                    }
                    const uint32_t kk = f2key(((mx == v) ? v : 0.0f) + 0.0f);   // heat * keep, -0.0 -> +0.0
                    if (degenerate || kk >= tcur)
                        f(((u64)kk << 32) | (u64)(0xFFFFFFFFu - (base + (uint32_t)e)), kk == KEY_ZERO);
                }
            };
            radix_select<NT>(for_each, (uint32_t)m, sh, prefix, mask);
            collect_and_sort<NT>(for_each, prefix, mask, sh);
        }
        src = sh.sel;
    }
    if (m > 0) {
        // one returning atomic per plane reserves its slice of the image's list; the keys go out as
        // device-scope stores (read by the image's last workgroup, possibly on another XCD, without any
        // cache maintenance in between -- see collect_merge_kernel)
        if (tid == 0) pl_base = atomicAdd(&counts[b], m);
        __syncthreads();
        const int o = pl_base;
        for (int j = tid; j < m; j += NT)
            if (o + j < cap) __hip_atomic_store(kimg + o + j, src[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------
// kernel 2: merge candidate lists of one group into its K best, sorted.
//   CTDET : group = image; N = C*per_class candidates, class = j / per_class;
//           output = gathered boxes (decode.py:472-493)
//   !CTDET: group = (image, channel); output = (scores, inds)  (_topk_channel)
// ---------------------------------------------------------------------------
enum { MODE_CTDET = 0, MODE_CHANNEL = 1, MODE_POSE = 2, MODE_TOPK = 3 };

constexpr int NTM = 1024;  // the merge runs one workgroup per image: make it a big one


// rows 0 .. K-1 of the sorted selection sh.sel -> outputs of group g (see merge_topk_kernel)
template <int MODE>
__device__ __forceinline__ void emit_rows(const SelShared &sh, int g, int H, int W, int K, int C,
                                          const float *__restrict__ wh, const float *__restrict__ reg,
                                          int cat_spec_wh, float *__restrict__ dets, int det_dim,
                                          int32_t *__restrict__ inds_out, float *__restrict__ out_scores,
                                          const float *__restrict__ kps_map, int J,
                                          int32_t *__restrict__ cls_out)
{
    constexpr bool CTDET = (MODE != MODE_CHANNEL);
    const int tid = threadIdx.x;
    const int HW = H * W;
    if (tid < K) {
        const u64 k = sh.sel[tid];
        const float score = key2f((uint32_t)(k >> 32));
        const uint32_t fid = 0xFFFFFFFFu - (uint32_t)k;
        if (CTDET) {
            const int cls = (int)(fid / (uint32_t)HW);
            const int ind = (int)(fid - (uint32_t)cls * (uint32_t)HW);
            const int yi = ind / W, xi = ind - yi * W;
            float xs = (float)xi, ys = (float)yi;
            const int b = g;
            if (MODE == MODE_TOPK) {  // _topk (decode.py:103-119): scores, inds, clses only
                out_scores[(size_t)b * K + tid] = score;
                inds_out[(size_t)b * K + tid] = ind;
                cls_out[(size_t)b * K + tid] = cls;
                return;
            }
            if (reg) {  // decode.py:472-476
                xs = xs + reg[((size_t)b * 2 + 0) * HW + ind];
                ys = ys + reg[((size_t)b * 2 + 1) * HW + ind];
            } else {  // decode.py:477-479
                xs = xs + 0.5f;
                ys = ys + 0.5f;
            }
            const int whC = cat_spec_wh ? 2 * C : 2;
            const int wc = cat_spec_wh ? 2 * cls : 0;  // decode.py:481-486
            const float w = wh[((size_t)b * whC + wc + 0) * HW + ind];
            const float h = wh[((size_t)b * whC + wc + 1) * HW + ind];
            float *d = dets + ((size_t)b * K + tid) * det_dim;
            d[0] = xs - w / 2;  // decode.py:489-492
            d[1] = ys - h / 2;
            d[2] = xs + w / 2;
            d[3] = ys + h / 2;
            d[4] = score;
            d[det_dim - 1] = (float)cls;
            if (MODE == MODE_POSE) {
                // decode.py:506-509: kps = hps[ind] + (xs, ys) with the un-offset centre
                const float x0 = (float)xi, y0 = (float)yi;
                for (int j = 0; j < J; ++j) {
                    d[5 + 2 * j] = kps_map[((size_t)b * 2 * J + 2 * j) * HW + ind] + x0;
                    d[5 + 2 * j + 1] = kps_map[((size_t)b * 2 * J + 2 * j + 1) * HW + ind] + y0;
                }
            }
            if (inds_out) inds_out[(size_t)b * K + tid] = ind;
        } else {
            out_scores[(size_t)g * K + tid] = score;
            inds_out[(size_t)g * K + tid] = (int32_t)fid;
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(NTM) void merge_topk_kernel(
    const float *__restrict__ cand_score, const int32_t *__restrict__ cand_idx, int N,
    int per_class, int H, int W, int K, int C, const float *__restrict__ wh,
    const float *__restrict__ reg, int cat_spec_wh, float *__restrict__ dets, int det_dim,
    int32_t *__restrict__ inds_out, float *__restrict__ out_scores,
    const float *__restrict__ kps_map, int J, int32_t *__restrict__ cls_out)
{
    constexpr bool CTDET = (MODE != MODE_CHANNEL);  // group = image, class from position
    extern __shared__ __attribute__((aligned(16))) char smem[];
    SelShared &sh = *reinterpret_cast<SelShared *>(smem);
    const int tid = threadIdx.x;
    const int g = blockIdx.x;
    const int HW = H * W;
    const float *cs = cand_score + (size_t)g * N;
    const int32_t *ci = cand_idx + (size_t)g * N;

    u64 prefix, mask;
    auto for_each = [&](auto &&f) {
        for (int j = tid; j < N; j += NTM) {
            const int32_t idx = ci[j];
            if (idx < 0) continue;
            const uint32_t kk = f2key(cs[j] + 0.0f);
            const uint32_t fid = CTDET ? (uint32_t)((j / per_class) * HW + idx) : (uint32_t)idx;
            f(((u64)kk << 32) | (u64)(0xFFFFFFFFu - fid), kk == KEY_ZERO);
        }
    };
    radix_select<NTM>(for_each, (uint32_t)K, sh, prefix, mask);
    collect_and_sort<NTM>(for_each, prefix, mask, sh);
    emit_rows<MODE>(sh, g, H, W, K, C, wh, reg, cat_spec_wh, dets, det_dim, inds_out, out_scores, kps_map, J,
                    cls_out);
}

// ---------------------------------------------------------------------------
// Second (and last) launch of the image-level decode: one workgroup per (image, class) plane
//   1. image_threshold  -- every plane of an image derives the same threshold from the group maxima
//                          of launch 1 (no wait between workgroups);
//   2. plane_collect    -- the plane's candidate keys;
//   3. the LAST plane of an image to arrive (one atomic counter per image; nobody waits for anybody)
//      selects and sorts the image's K best candidates and writes its detections.
// Degenerate images and saturated plateaus are dealt with plane by plane in plane_collect (every
// plane hands on at most its K best keys): the last arriver always finds between K and C * K keys.
// ---------------------------------------------------------------------------
struct EmitArgs {
    const float *wh, *reg;
    int cat_spec_wh;
    float *dets;
    int det_dim;
    int32_t *inds_out;
    float *out_scores;
    int32_t *cls_out;
};

template <int MODE>
__global__ __launch_bounds__(NT) void collect_merge_kernel(const float *__restrict__ heat, int C, int H,
                                                           int W, int nrg, int ncb, int flags, int K,
                                                           const uint32_t *__restrict__ gpeak,
                                                           const uint32_t *__restrict__ gall,
                                                           u64 *__restrict__ keys, int cap,
                                                           int32_t *__restrict__ counts,
                                                           int32_t *__restrict__ done, const EmitArgs ea)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    SelShared &sh = *reinterpret_cast<SelShared *>(smem);
    __shared__ int s_last;
    const int tid = threadIdx.x;
    const int b = (int)(blockIdx.x / (unsigned)C);
    const int ng = C * nrg * ncb;
    float rthr;
    const uint32_t tkey = image_threshold(gpeak + (size_t)b * ng, ng, K, flags, sh, rthr);
    plane_collect(heat, C, H, W, nrg, ncb, flags, K, gall, tkey, rthr, sh, keys, cap, counts);
    // ---- arrival.  What the image's last workgroup reads from the others -- candidate keys and their
    // count -- is written with device-scope atomics (write-through to the coherence point) and read
    // with device-scope loads: no __threadfence() on either side (on this part a device-scope release /
    // acquire is a write-back / invalidate of the whole L2 of the XCD; 2560 workgroups doing that cost
    // 0.18 ms).  Order: every wave waits for the acknowledgement of its stores, the workgroup
    // meets, then ONE thread bumps the image's counter.
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0)
        s_last = (__hip_atomic_fetch_add(&done[b], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == C - 1) ? 1 : 0;
    __syncthreads();
    if (!s_last) return;
    // every plane handed on its K best (or all it had): between K and C * K keys, the image's K best
    // among them (fewer than K only when the whole image has fewer cells: excluded by the caller)
    const int cnt = min(__hip_atomic_load(&counts[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), cap);
    const u64 *kg = keys + (size_t)b * cap;
    u64 prefix, mask;
    constexpr int RK = 16;
    if (cnt <= RK * NT) {
        // fetched once into registers (16 per thread), device-scope loads
        u64 kr[RK];
#pragma unroll
        for (int u = 0; u < RK; ++u)
            kr[u] = (tid + u * NT < cnt) ? __hip_atomic_load(kg + tid + u * NT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                         : 0ull;
        auto for_each = [&](auto &&f) {
#pragma unroll
            for (int u = 0; u < RK; ++u)
                if (tid + u * NT < cnt) f(kr[u], (uint32_t)(kr[u] >> 32) == KEY_ZERO);
        };
        radix_select<NT>(for_each, (uint32_t)K, sh, prefix, mask);
        collect_and_sort<NT>(for_each, prefix, mask, sh);
    } else {
        auto for_each = [&](auto &&f) {
            for (int j = tid; j < cnt; j += NT) {
                const u64 k = __hip_atomic_load(kg + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                f(k, (uint32_t)(k >> 32) == KEY_ZERO);
            }
        };
        radix_select<NT>(for_each, (uint32_t)K, sh, prefix, mask);
        collect_and_sort<NT>(for_each, prefix, mask, sh);
    }
    emit_rows<MODE>(sh, b, H, W, K, C, ea.wh, ea.reg, ea.cat_spec_wh, ea.dets, ea.det_dim, ea.inds_out,
                    ea.out_scores, (const float *)nullptr, 0, ea.cls_out);
}

// ---------------------------------------------------------------------------
// ONE-launch form of the image-level decode (round 4): planes of at most 128 x 128 cells -- the
// ctdet map of a 512 x 512 input and everything smaller.  The heat-map is read exactly ONCE: a
// workgroup of eight waves owns one (image, class) plane and holds it in REGISTERS -- half-wave u
// keeps rows 8u .. 8u+7 (+ one halo row above and below, from L2), lane l the 4-cell quad l of every
// row -- so the logistic, the 3x3 peak test on the scores (decode.py:9-15) and every later look at a
// cell are register work; horizontal neighbours come from the adjacent lanes by shuffle.  Then
//   1. every lane takes the largest key of its 32 cells; T0 = the K-th largest of the 512 lane maxima
//      to OP_TBITS bits (ONE wave, the maxima in its registers, a binary search on wave ballots: no
//      histogram, no barrier inside).  At least K cells of the plane reach T0, so nothing below it
//      can be among the plane's -- or the image's -- K best.  T0 is folded into the image's FLOOR (a
//      device-scope atomic maximum): planes that start later skip every cell below the best T0
//      published so far.  Workgroups run class-major (all images' plane 0, then plane 1, ...): an
//      image's planes are spread over the launch and most of them find a floor;
//   2. the cells that reach max(T0, floor) go into a list in LDS; up to OP_PE keys go as they are
//      into the plane's OWN slot of the image's key buffer (no position to fetch from a counter: one
//      device-scope round trip less), a longer list goes through the exact select of the plane's K
//      best, and a list that overflows LDS (plateaus, tiny or constant maps: T0 useless) through the
//      same select over the register cells, zeros of suppressed cells included;
//   3. arrival as in collect_merge_kernel; the last arriver reads the planes' key counts, keeps the
//      keys that reach the image's FINAL floor (a few hundred: in LDS), selects and sorts the K best
//      and writes the detections.  It leaves the image's state words (arrival counter, floor) at zero.
// Every input is handled exactly inside this launch; results are bit-identical to the two-launch
// and the per-band forms (tests/test_gpu_decode.py).
// ---------------------------------------------------------------------------
constexpr int OP_NT = 512;        // threads (8 waves)
constexpr int OP_ROWS = 8;        // rows a half-wave owns
constexpr int OP_CELLS = OP_ROWS * 4;
constexpr int OP_PE = 256;        // key slots of a plane in the image's buffer (>= KMAX)
constexpr int OP_RK = 40;         // key slots per thread the last arriver keeps in registers (80 planes)
constexpr int OP_CMAX = 1024;     // classes (the planes' key counts sit in LDS)
constexpr int OP_TBITS = 16;      // leading key bits T0 is resolved to
static_assert(OP_PE >= KMAX && OP_PE <= PLCAP, "plane emit cap");
static_assert((OP_PE & (OP_PE - 1)) == 0 && OP_NT * 4 % OP_PE == 0, "slot arithmetic");

#ifdef CN_ABLATE_DECODE       // variant builds only (tools/build_variant.sh): leave after stage (flags >> 16) & 15
#define PSM_STAGE_EXIT(n, keep)                                                         \
    if (((flags >> 16) & 15) == (n)) {                                                  \
        if ((keep) == 0xdeadbeefu) floorv[b] = 1u;                                      \
        return;                                                                         \
    }
#else
#define PSM_STAGE_EXIT(n, keep)
#endif

// The image's last arriver of plane_select_merge_kernel (its own function, not inlined: 80 registers of
// keys must not weigh on the allocation of the plane part every workgroup runs).
template <int MODE>
__device__ __attribute__((noinline)) void psm_last_arriver(SelShared &sh, int32_t *s_pc, uint32_t *s_kmax_p, int b,
                                                           int C, int H, int W, int K,
                                                           const u64 *__restrict__ keys,
                                                           const int32_t *__restrict__ pcount,
                                                           const uint32_t *__restrict__ floorv, const EmitArgs ea)
{
    const int tid = threadIdx.x;
    const int lane = tid & (CN_WAVE - 1);
    uint32_t &s_kmax = *s_kmax_p;
    // every plane's floor update precedes its arrival: this is the image's final floor, and the
    // image's K best all reach it (the plane that set it has K cells there and handed them on)
    const uint32_t F = max(__hip_atomic_load(&floorv[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), 1u);
    for (int p = tid; p < C; p += OP_NT)
        s_pc[p] = __hip_atomic_load(&pcount[(size_t)b * C + p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) s_kmax = 0u;
    for (int i = tid; i < HBINS; i += OP_NT) sh.hist[i] = 0;
    __syncthreads();
    const u64 *kimg = keys + (size_t)b * C * OP_PE;
    const int slots = C * OP_PE;
    u64 prefix, mask;
    if (slots <= OP_NT * OP_RK) {
        // Every key of the image in registers, ONE round trip to the coherence point.  Then a LINEAR
        // histogram of the score keys over [F, largest key] (2048 bins: order-preserving integers, so
        // the bin that holds the K-th largest is exact) leaves the keys of that bin and above -- K plus
        // a few -- which are sorted as they are; a plateau that fills the bin goes through the digit
        // select.
        u64 kr[OP_RK];
#pragma unroll
        for (int u = 0; u < OP_RK; ++u) {
            const int i = tid + u * OP_NT;
            const bool ok = i < slots && (i & (OP_PE - 1)) < s_pc[min(i, slots - 1) / OP_PE];
            kr[u] = ok ? __hip_atomic_load(kimg + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        }
        uint32_t mx = 0u;
#pragma unroll
        for (int u = 0; u < OP_RK; ++u) {
            const uint32_t sk = (uint32_t)(kr[u] >> 32);
            if (sk < F) kr[u] = 0ull;                            // (and the empty slots: key 0)
            else mx = max(mx, sk);
        }
#pragma unroll
        for (int o = CN_WAVE / 2; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
        if (lane == 0 && mx) atomicMax(&s_kmax, mx);
        __syncthreads();
        const uint32_t range = max(s_kmax, F) - F;
        const int shift = max(0, 21 - __clz((int)(range | 1u)));      // (range >> shift) < 2048
#pragma unroll
        for (int u = 0; u < OP_RK; ++u)
            if (kr[u]) atomicAdd(&sh.hist[((uint32_t)(kr[u] >> 32) - F) >> shift], 1u);
        __syncthreads();
        suffix_bin<OP_NT>((uint32_t)K, sh);                      // sh.digit: the bin, sh.bincount: keys in it and above
        const uint32_t F1 = F + (sh.digit << shift);
        const uint32_t kept = sh.bincount;
        auto for_each = [&](auto &&f) {
#pragma unroll
            for (int u = 0; u < OP_RK; ++u) {
                const uint32_t sk = (uint32_t)(kr[u] >> 32);
                if (sk >= F1) f(kr[u], sk == KEY_ZERO);
            }
        };
        if (kept <= (uint32_t)KMAX) {
            collect_and_sort<OP_NT>(for_each, 0ull, 0ull, sh);   // all of them
        } else {
            radix_select<OP_NT>(for_each, (uint32_t)K, sh, prefix, mask);
            collect_and_sort<OP_NT>(for_each, prefix, mask, sh);
        }
    } else {                                 // (more planes than the registers take: straight from the slots)
        auto for_each = [&](auto &&f) {
            for (int i = tid; i < slots; i += OP_NT) {
                if ((i & (OP_PE - 1)) < s_pc[i / OP_PE]) {
                    const u64 k = __hip_atomic_load(kimg + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((uint32_t)(k >> 32) >= F) f(k, (uint32_t)(k >> 32) == KEY_ZERO);
                }
            }
        };
        radix_select<OP_NT>(for_each, (uint32_t)K, sh, prefix, mask);
        collect_and_sort<OP_NT>(for_each, prefix, mask, sh);
    }
    emit_rows<MODE>(sh, b, H, W, K, C, ea.wh, ea.reg, ea.cat_spec_wh, ea.dets, ea.det_dim, ea.inds_out,
                    ea.out_scores, (const float *)nullptr, 0, ea.cls_out);
}

#ifndef CN_PSM_WAVES
#define CN_PSM_WAVES 4        // waves per SIMD the register budget is cut for (two workgroups per CU)
#endif
template <int MODE>
__global__ __launch_bounds__(OP_NT, CN_PSM_WAVES) void plane_select_merge_kernel(const float *__restrict__ heat, int B, int C,
                                                                      int H, int W, int flags, int K,
                                                                      u64 *__restrict__ keys,
                                                                      int32_t *__restrict__ pcount,
                                                                      int32_t *__restrict__ done,
                                                                      uint32_t *__restrict__ floorv, const EmitArgs ea)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    SelShared &sh = *reinterpret_cast<SelShared *>(smem);
    __shared__ u64 kbuf[PLCAP];             // the plane's list
    __shared__ uint32_t s_lmax[OP_NT / 2];
    __shared__ int32_t s_pc[OP_CMAX];
    __shared__ int pl_cnt, s_last;
    __shared__ uint32_t s_thr, s_kmax;
    const int tid = threadIdx.x;
    const int lane = tid & (CN_WAVE - 1);
    const int hl = lane & 31, hw = tid >> 5;
    int b, c;
    // planes class by class (all images' plane 0, then plane 1, ...): an image's planes are spread over the
    // launch and most of them find a floor.  (Round 4 A/B, removed in round 5: image by image +40 % time;
    // class-major in blocks of eight classes +9 % warm, -19 % cold -- DESIGN.md 3.4.)
    c = (int)(blockIdx.x / (unsigned)B);
    b = (int)blockIdx.x - c * B;
    const size_t plane_id = (size_t)b * C + c;
    const bool sig = (flags & 1) != 0;
    const bool nonms = (flags & CN_DECODE_NO_PEAK_TEST) != 0;
    const int HW = H * W;
    const float *plane = heat + plane_id * (size_t)HW;
    const uint32_t base = (uint32_t)c * (uint32_t)HW;
    const int w4 = W >> 2;
    // the image's floor so far: requested first, lands under the map loads
    uint32_t fl = 0u;
    if (tid == 0) fl = __hip_atomic_load(&floorv[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // ---- the plane into registers: rows y0-1 .. y0+8 of this half-wave, all loads issued up front
    const int y0 = hw * OP_ROWS;
    const bool col_ok = hl < w4;
    const int nv = col_ok ? min(max(H - y0, 0), OP_ROWS) : 0;    // valid rows of this lane's 8
    // (branch-free: rows / quads outside the map read a clamped address and are replaced below)
    cn_f32x4 rowv[OP_ROWS + 2];
    {
        const uint32_t xoff = col_ok ? (uint32_t)hl * 4u : 0u;
#pragma unroll
        for (int r = 0; r < OP_ROWS + 2; ++r) {
            const int yc = min(max(y0 - 1 + r, 0), H - 1);
            rowv[r] = *reinterpret_cast<const cn_f32x4 *>(plane + ((uint32_t)yc * (uint32_t)W + xoff));
        }
    }
#ifdef CN_ABLATE_DECODE
    if (((flags >> 16) & 15) == 1) {
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < OP_ROWS + 2; ++r) acc += rowv[r].x + rowv[r].y + rowv[r].z + rowv[r].w;
        if (acc == 123.456f) floorv[b] = 1u;
        return;
    }
#endif
    // (selections below are written as bit masks: left as ?: the compiler wraps every transcendental
    // and every shuffle fix-up in its own branch)
    const uint32_t NINF = 0xff800000u;
    auto sel = [](uint32_t msk, float a, float bfl) -> float {   // msk ? a : b, bitwise
        return __uint_as_float((__float_as_uint(a) & msk) | (__float_as_uint(bfl) & ~msk));
    };
    // scores of a window row; cells outside the map stay -inf (the peak test's padding, decode.py:12)
    auto score_row = [&](int r) {
        const int y = y0 - 1 + r;
        const uint32_t in_map = (col_ok && y >= 0 && y < H) ? 0xffffffffu : 0u;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x = rowv[r][e];
            rowv[r][e] = sel(in_map, sig ? sigmoidf_ref(x) : x, __uint_as_float(NINF));
        }
    };
    // horizontal 3-max of a row (neighbour quads by shuffle inside the half-wave)
    const uint32_t has_l = (hl != 0 && col_ok) ? 0xffffffffu : 0u;
    const uint32_t has_r = (hl != 31 && col_ok) ? 0xffffffffu : 0u;
    auto hrow = [&](int r) -> cn_f32x4 {
        const float l = sel(has_l, __shfl_up(rowv[r].w, 1, 32), __uint_as_float(NINF));
        const float rr = sel(has_r, __shfl_down(rowv[r].x, 1, 32), __uint_as_float(NINF));
        cn_f32x4 h;
        h.x = fmaxf(fmaxf(l, rowv[r].x), rowv[r].y);
        h.y = fmaxf(fmaxf(rowv[r].x, rowv[r].y), rowv[r].z);
        h.z = fmaxf(fmaxf(rowv[r].y, rowv[r].z), rowv[r].w);
        h.w = fmaxf(fmaxf(rowv[r].z, rowv[r].w), rr);
        return h;
    };
    // the 32 cells of this lane as keys of heat * keep (decode.py:14; -0.0 -> +0.0); cell i = row
    // y0 + (i >> 2), column 4 hl + (i & 3).  Cells outside the map get key 0, which no threshold
    // admits (thresholds are >= 1; a cell of the map has key 0 only for one NaN pattern).  A row's
    // registers die as soon as the row below it has been combined.
    uint32_t kreg[OP_CELLS];
    {
        score_row(0);
        score_row(1);
        cn_f32x4 h0 = hrow(0), h1 = hrow(1);
#pragma unroll
        for (int r = 1; r <= OP_ROWS; ++r) {
            score_row(r + 1);
            const cn_f32x4 h2 = hrow(r + 1);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = rowv[r][e];
                const float m = fmaxf(fmaxf(h0[e], h1[e]), h2[e]);
                const uint32_t k = f2key(((nonms || m == v) ? v : 0.0f) + 0.0f);
                kreg[(r - 1) * 4 + e] = (r - 1 < nv) ? k : 0u;
            }
            h0 = h1;
            h1 = h2;
        }
    }
    auto key64 = [&](int i) -> u64 {
        const uint32_t cell = (uint32_t)((y0 + (i >> 2)) * W + hl * 4 + (i & 3));
        return ((u64)kreg[i] << 32) | (u64)(0xFFFFFFFFu - (base + cell));
    };

    // ---- 1. T0 = K-th largest lane maximum, to OP_TBITS bits (0: fewer than K lanes hold a cell --
    // small maps: every cell of the map is admitted)
    uint32_t lmax = 0u;
#pragma unroll
    for (int i = 0; i < OP_CELLS; ++i) lmax = max(lmax, kreg[i]);
    PSM_STAGE_EXIT(2, lmax)
    // (a lane and its partner in the other half-wave count as one: 256 values, each the key of one cell)
    const uint32_t pmax = max(lmax, (uint32_t)__shfl_xor((int)lmax, 32));
    if (lane < 32) s_lmax[(tid >> 6) * 32 + lane] = pmax;
    if (tid == 0) pl_cnt = 0;
    __syncthreads();
    if (tid < CN_WAVE) {
        // ONE wave searches while seven wait: it goes first on its SIMD
        __builtin_amdgcn_s_setprio(3);
        constexpr int NV = OP_NT / 2 / CN_WAVE;
        uint32_t v[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j] = s_lmax[lane + j * CN_WAVE];
        uint32_t T0 = 0u;
#pragma unroll
        for (int bit = 31; bit >= 32 - OP_TBITS; --bit) {
            const uint32_t cand = T0 | (1u << bit);
            int cnt = 0;
#pragma unroll
            for (int j = 0; j < NV; ++j) cnt += __popcll(__ballot(v[j] >= cand));
            if (cnt >= K) T0 = cand;
        }
        if (tid == 0) {
            s_thr = max(max(T0, fl), 1u);
            if (T0 > fl) __hip_atomic_fetch_max(&floorv[b], T0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __builtin_amdgcn_s_setprio(0);
    }
    __syncthreads();
    const uint32_t thr = s_thr;
    PSM_STAGE_EXIT(3, thr)

    // ---- 2. cells that reach thr -> LDS list (half-wave prefix sums, one LDS atomic per half-wave)
    uint32_t m0 = 0u;
#pragma unroll
    for (int i = 0; i < OP_CELLS; ++i) m0 |= (kreg[i] >= thr) ? (1u << i) : 0u;
    {
        const int mine = __popc(m0);
        int incl = mine;
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
            const int t = __shfl_up(incl, o, 32);
            if (hl >= o) incl += t;
        }
        const int total = __shfl(incl, 31, 32);
        int pos0 = 0;
        if (hl == 31 && total) pos0 = atomicAdd(&pl_cnt, total);
        int pos = __shfl(pos0, 31, 32) + incl - mine;
        if (mine) {
#pragma unroll
            for (int i = 0; i < OP_CELLS; ++i) {
                if ((m0 >> i) & 1u) {
                    if (pos < PLCAP) kbuf[pos] = key64(i);
                    ++pos;
                }
            }
        }
    }
    __syncthreads();
    const int n = pl_cnt;                    // (uniform)
    PSM_STAGE_EXIT(4, (uint32_t)n)
    int m;
    const u64 *src;
    if (n <= OP_PE) {
        m = n;
        src = kbuf;
    } else {
        m = K;                               // n > OP_PE >= K cells reach thr: the plane's K best are among them
        u64 prefix, mask;
        if (n <= PLCAP) {
            auto for_each = [&](auto &&f) {
                for (int j = tid; j < n; j += OP_NT) f(kbuf[j], false);
            };
            radix_select<OP_NT>(for_each, (uint32_t)m, sh, prefix, mask);
            collect_and_sort<OP_NT>(for_each, prefix, mask, sh);
        } else {
            auto for_each = [&](auto &&f) {
#pragma unroll
                for (int i = 0; i < OP_CELLS; ++i)
                    if (kreg[i] >= thr) f(key64(i), kreg[i] == KEY_ZERO);
            };
            radix_select<OP_NT>(for_each, (uint32_t)m, sh, prefix, mask);
            collect_and_sort<OP_NT>(for_each, prefix, mask, sh);
        }
        src = sh.sel;
    }
    // the plane's own slot: nothing to fetch before the stores can go
    u64 *kslot = keys + plane_id * OP_PE;
    for (int j = tid; j < m; j += OP_NT) __hip_atomic_store(kslot + j, src[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) __hip_atomic_store(&pcount[plane_id], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    PSM_STAGE_EXIT(5, (uint32_t)m)

    // ---- 3. arrival (see collect_merge_kernel for the memory-ordering argument)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0)
        s_last = (__hip_atomic_fetch_add(&done[b], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == C - 1) ? 1 : 0;
    __syncthreads();
    if (!s_last) return;
    PSM_STAGE_EXIT(6, (uint32_t)m)
    psm_last_arriver<MODE>(sh, s_pc, &s_kmax, b, C, H, W, K, keys, pcount, floorv, ea);
    // the image's state words go back to zero: the next call on this workspace needs no fill
    // (device-scope stores: the words of other images share these cache lines and are being updated
    // by atomics from other XCDs)
    if (tid == 0) {
        __hip_atomic_store(&done[b], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&floorv[b], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}


struct BandPlan {
    int R;       // rows per band
    int nbands;
    size_t lds;  // dynamic LDS bytes of kernel 1
};

// Deterministic function of the shapes only (workspace query == launch).
// Bands of <= 8192 cells: tile + list + select state stay under ~52 KB, so three
// workgroups share a CU and their load / peak-test / select phases overlap.
bool make_band_plan(int B, int C, int H, int W, int K, BandPlan *bp)
{
    (void)B; (void)C;
    if (W <= 0 || H <= 0 || W > 4096) return false;
    int R = 8192 / W;
    if (R < 1) R = 1;
    if (R > H) R = H;
    while (R < H && R * W < K) ++R;  // a band should be able to supply K candidates
    if ((long)R * W > 16384) return false;
    bp->R = R;
    bp->nbands = cn_cdiv(H, R);
    bp->lds = sizeof(SelShared) + (size_t)(R + 2) * W * sizeof(float) + CAND_CAP * sizeof(uint16_t);
    return bp->lds <= 160 * 1024;
}

int launch_nms_topk(const float *heat, int B, int C, int H, int W, int K, int apply_sigmoid,
                    const BandPlan &bp, float *cand_score, int32_t *cand_idx, hipStream_t st)
{
    dim3 grid(bp.nbands, C, B), block(NT);
    CN_SET_MAX_LDS_ONCE(nms_topk_kernel, 160 * 1024);
    hipLaunchKernelGGL(nms_topk_kernel, grid, block, bp.lds, st, heat, C, H, W, K, bp.R,
                       apply_sigmoid, cand_score, cand_idx);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

}  // namespace

namespace {
// image-level (threshold-pruned) top-K: layout of its scratch behind the per-band candidate
// arrays, and whether a shape takes it (enough groups per image for a meaningful threshold;
// single-class maps such as the pose centre map keep the per-band select)
struct ImgPlan {
    bool use;
    bool one_pass;                 // planes of <= 128 x 128 cells: plane_select_merge_kernel (ONE launch)
    int nrg, ncb;                  // 8-row groups per plane, 128-column blocks per row
    int cap;                       // candidate keys per image: every plane hands on at most its K best
    size_t gpeak, gall, counts, done, floorv, pcount, keys, total;   // byte offsets in the workspace
    size_t state_bytes;            // counts | done | floorv: contiguous, zero between calls of the one-launch form
};
ImgPlan make_img_plan(int B, int C, int H, int W, int K, const BandPlan &bp)
{
    ImgPlan p = {};
    const size_t n = (size_t)B * C * bp.nbands * K;
    size_t o = cn_align_up(n * sizeof(float), 256) + cn_align_up(n * sizeof(int32_t), 256);
    p.nrg = cn_cdiv(H, GROWS);
    p.ncb = cn_cdiv(W, 128);
    // enough groups per image for a meaningful threshold; rows of whole quads
    p.use = (W & 3) == 0 && (long)C * p.nrg * p.ncb >= 4L * K;
    p.one_pass = p.use && H <= (OP_NT / 32) * OP_ROWS && W <= 128 && C <= OP_CMAX;
    const size_t ng = (size_t)B * C * p.nrg * p.ncb;
    p.gpeak = o;  o += cn_align_up(ng * 4, 256);
    p.gall = o;   o += cn_align_up(ng * 4, 256);
    p.counts = o; o += cn_align_up((size_t)B * 4, 256);
    p.done = o;   o += cn_align_up((size_t)B * 4, 256);
    p.floorv = o; o += cn_align_up((size_t)B * 4, 256);
    p.state_bytes = o - p.counts;
    p.pcount = o; o += cn_align_up((size_t)B * C * 4, 256);   // keys in every plane's slot (one-launch form)
    p.cap = C * (p.one_pass ? OP_PE : K);
    p.keys = o;   o += cn_align_up((size_t)B * p.cap * 8, 256);
    p.total = o;
    return p;
}

}  // namespace
// Where the one-launch image-level decode keeps its per-image state words inside the workspace of
// cn_ctdet_decode_f32 / cn_topk_f32 (byte offset and size; 0 bytes when the shape takes another form).
// They are zero between calls: a caller that owns its workspace (CN_DECODE_STATE_CLEAN) can check or
// re-zero exactly this region after a failed call instead of trusting it.
extern "C" int cn_decode_state_region(int B, int C, int H, int W, int K, size_t *offset, size_t *bytes)
{
    if (!offset || !bytes) return CN_ERR_NULL;
    *offset = 0; *bytes = 0;
    BandPlan bp;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || K <= 0) return CN_ERR_SHAPE;
    if (!make_band_plan(B, C, H, W, K, &bp)) return CN_ERR_UNSUPPORTED;
    const ImgPlan ip = make_img_plan(B, C, H, W, K, bp);
    if (ip.use && ip.one_pass) { *offset = ip.counts; *bytes = ip.state_bytes; }
    return CN_OK;
}
namespace {
// the image-level decode: TWO launches (group maxima; threshold + candidates + select + outputs)
template <int MODE>
int launch_image_topk(const float *heat, int B, int C, int H, int W, int K, int flags, const ImgPlan &ip,
                      char *ws, const EmitArgs &ea, hipStream_t st)
{
    uint32_t *gpeak = (uint32_t *)(ws + ip.gpeak);
    uint32_t *gall = (uint32_t *)(ws + ip.gall);
    int32_t *counts = (int32_t *)(ws + ip.counts);
    int32_t *done = (int32_t *)(ws + ip.done);
    u64 *keys = (u64 *)(ws + ip.keys);
    dim3 grid((unsigned)(B * C)), block(NT);
    if (ip.one_pass && !(flags & CN_DECODE_TWO_LAUNCHES)) {
        // ONE launch.  The image state words (arrival counter, floor) must be zero on entry and are
        // left at zero by the kernel: a caller that keeps the workspace to itself says so
        // (CN_DECODE_STATE_CLEAN) after zeroing it once; anyone else pays a small fill.
        if (!(flags & CN_DECODE_STATE_CLEAN)) {
            if (hipMemsetAsync(ws + ip.counts, 0, ip.state_bytes, st) != hipSuccess) return CN_ERR_LAUNCH;
        }
        hipLaunchKernelGGL(plane_select_merge_kernel<MODE>, grid, dim3(OP_NT), sizeof(SelShared), st, heat, B, C,
                           H, W, flags, K, keys, (int32_t *)(ws + ip.pcount), done, (uint32_t *)(ws + ip.floorv),
                           ea);
        CN_CHECK_LAUNCH();
        return CN_OK;
    }
    hipLaunchKernelGGL(group_max_kernel, grid, block, 0, st, heat, H, W, ip.nrg, ip.ncb, flags, gpeak,
                       gall, C, counts, done);
    CN_CHECK_LAUNCH();
    hipLaunchKernelGGL(collect_merge_kernel<MODE>, grid, block, sizeof(SelShared), st, heat, C, H, W,
                       ip.nrg, ip.ncb, flags, K, gpeak, gall, keys, ip.cap, counts, done, ea);
    CN_CHECK_LAUNCH();
    return CN_OK;
}
}  // namespace

extern "C" size_t cn_ctdet_decode_workspace_bytes(int B, int C, int H, int W, int K)
{
    BandPlan bp;
    if (B <= 0 || C <= 0 || K <= 0 || !make_band_plan(B, C, H, W, K, &bp)) return 0;
    return make_img_plan(B, C, H, W, K, bp).total;
}

static int decode_checks(const void *heat, int B, int C, int H, int W, int K, BandPlan *bp)
{
    if (!heat) return CN_ERR_NULL;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || K <= 0) return CN_ERR_SHAPE;
    if ((long)K > (long)H * W) return CN_ERR_SHAPE;  // torch.topk: k out of range
    if (K > KMAX) return CN_ERR_UNSUPPORTED;
    if ((long)C * H * W >= (1L << 31)) return CN_ERR_UNSUPPORTED;
    if (!make_band_plan(B, C, H, W, K, bp)) return CN_ERR_UNSUPPORTED;
    if (((W & 3) == 0) && !cn_aligned16(heat)) return CN_ERR_ALIGN;
    return CN_OK;
}

extern "C" int cn_ctdet_decode_f32(const float *heat, const float *wh, const float *reg, int B,
                                   int C, int H, int W, int K, int cat_spec_wh, int apply_sigmoid,
                                   float *dets, int32_t *inds, void *workspace,
                                   size_t workspace_bytes, void *stream)
{
    BandPlan bp;
    int rc = decode_checks(heat, B, C, H, W, K, &bp);
    if (rc != CN_OK) return rc;
    if (!wh || !dets || !workspace) return CN_ERR_NULL;
    const size_t need = cn_ctdet_decode_workspace_bytes(B, C, H, W, K);
    if (workspace_bytes < need) return CN_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)B * C * bp.nbands * K;
    float *cand_score = (float *)workspace;
    int32_t *cand_idx = (int32_t *)((char *)workspace + cn_align_up(n * sizeof(float), 256));
    const ImgPlan ip = make_img_plan(B, C, H, W, K, bp);
    if (ip.use && !(apply_sigmoid & 2048)) {   // bit 11: force the per-band select (tests / A-B)
        const EmitArgs ea = {wh, reg, cat_spec_wh, dets, 6, inds, nullptr, nullptr};
        return launch_image_topk<MODE_CTDET>(heat, B, C, H, W, K, apply_sigmoid, ip, (char *)workspace, ea, st);
    }
    rc = launch_nms_topk(heat, B, C, H, W, K, apply_sigmoid, bp, cand_score, cand_idx, st);
    if (rc != CN_OK) return rc;
    hipLaunchKernelGGL(merge_topk_kernel<MODE_CTDET>, dim3(B), dim3(NTM), sizeof(SelShared), st,
                       cand_score, cand_idx, C * bp.nbands * K, bp.nbands * K, H, W, K, C, wh, reg,
                       cat_spec_wh, dets, 6, inds, (float *)nullptr, (const float *)nullptr, 0,
                       (int32_t *)nullptr);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

extern "C" int cn_nms_topk_channel_f32(const float *heat, int B, int C, int H, int W, int K,
                                       int apply_sigmoid, float *scores, int32_t *inds,
                                       void *workspace, size_t workspace_bytes, void *stream)
{
    BandPlan bp;
    int rc = decode_checks(heat, B, C, H, W, K, &bp);
    if (rc != CN_OK) return rc;
    if (!scores || !inds) return CN_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    if (bp.nbands == 1) {
        // one band per plane: kernel 1 already emits the final sorted (B,C,K) lists
        return launch_nms_topk(heat, B, C, H, W, K, apply_sigmoid, bp, scores, inds, st);
    }
    if (!workspace) return CN_ERR_NULL;
    const size_t need = cn_ctdet_decode_workspace_bytes(B, C, H, W, K);
    if (workspace_bytes < need) return CN_ERR_WORKSPACE;
    const size_t n = (size_t)B * C * bp.nbands * K;
    float *cand_score = (float *)workspace;
    int32_t *cand_idx = (int32_t *)((char *)workspace + cn_align_up(n * sizeof(float), 256));
    rc = launch_nms_topk(heat, B, C, H, W, K, apply_sigmoid, bp, cand_score, cand_idx, st);
    if (rc != CN_OK) return rc;
    hipLaunchKernelGGL(merge_topk_kernel<MODE_CHANNEL>, dim3(B * C), dim3(NTM), sizeof(SelShared),
                       st, cand_score, cand_idx, bp.nbands * K, bp.nbands * K, H, W, K, C,
                       (const float *)nullptr, (const float *)nullptr, 0, (float *)nullptr, 0, inds,
                       scores, (const float *)nullptr, 0, (int32_t *)nullptr);
    CN_CHECK_LAUNCH();
    return CN_OK;
}


// ---------------------------------------------------------------------------
// multi_pose decode (models/decode.py:497-571)
//   stage A  nms_topk + merge<MODE_POSE>: boxes, scores, regression keypoints
//   stage B  nms_topk (+ channel merge) on hm_hp: K candidates per joint (_topk_channel)
//   stage C  pose_match_kernel: nearest candidate per (detection, joint), reject rule, blend
// Arithmetic is kept in the reference's association (no FMA contraction) so the result
// is bit-identical to torch on CPU: the argmin and the threshold tests are discontinuous.
// ---------------------------------------------------------------------------
namespace {

__global__ __launch_bounds__(KMAX) void pose_match_kernel(
    const float *__restrict__ hp_scores, const int32_t *__restrict__ hp_inds,
    const float *__restrict__ hp_offset, float *__restrict__ dets, int J, int K, int H, int W,
    int det_dim)
{
    __shared__ float cs[KMAX], cx[KMAX], cy[KMAX];
    const int j = blockIdx.x, b = blockIdx.y;
    const int q = threadIdx.x;
    const int HW = H * W;
    const float thresh = 0.1f;
    if (q < K) {
        const size_t o = ((size_t)b * J + j) * K + q;
        float s = hp_scores[o];
        const int ind = hp_inds[o];
        const int yi = ind / W, xi = ind - yi * W;
        float x = (float)xi, y = (float)yi;
        if (hp_offset) {  // decode.py:534-539
            x = add_rn(x, hp_offset[((size_t)b * 2 + 0) * HW + ind]);
            y = add_rn(y, hp_offset[((size_t)b * 2 + 1) * HW + ind]);
        } else {
            x = add_rn(x, 0.5f);
            y = add_rn(y, 0.5f);
        }
        const float m = (s > thresh) ? 1.0f : 0.0f;  // decode.py:544-547
        const float im = __fsub_rn(1.0f, m);
        cs[q] = add_rn(mul_rn(im, -1.0f), mul_rn(m, s));
        cy[q] = add_rn(mul_rn(im, -10000.0f), mul_rn(m, y));
        cx[q] = add_rn(mul_rn(im, -10000.0f), mul_rn(m, x));
    }
    __syncthreads();
    if (q < K) {
        float *d = dets + ((size_t)b * K + q) * det_dim;
        const float rx = d[5 + 2 * j], ry = d[5 + 2 * j + 1];
        float best = 0.f;
        int bi = -1;
        for (int c = 0; c < K; ++c) {  // decode.py:550-551, first minimum
            const float dx = __fsub_rn(rx, cx[c]);
            const float dy = __fsub_rn(ry, cy[c]);
            const float dist = __fsqrt_rn(add_rn(mul_rn(dx, dx), mul_rn(dy, dy)));
            if (bi < 0 || dist < best) {
                best = dist;
                bi = c;
            }
        }
        const float sc = cs[bi], kx = cx[bi], ky = cy[bi];
        const float l = d[0], t = d[1], r = d[2], bt = d[3];
        const float bh = __fsub_rn(bt, t), bw = __fsub_rn(r, l);
        const float mx = mul_rn(fmaxf(bh, bw), 0.3f);
        const bool reject = (kx < l) || (kx > r) || (ky < t) || (ky > bt) || (sc < thresh) ||
                            (best > mx);  // decode.py:562-565
        const float m = reject ? 1.0f : 0.0f;
        const float im = __fsub_rn(1.0f, m);
        d[5 + 2 * j] = add_rn(mul_rn(im, kx), mul_rn(m, rx));  // decode.py:566
        d[5 + 2 * j + 1] = add_rn(mul_rn(im, ky), mul_rn(m, ry));
    }
}

struct PoseWs {
    size_t cand_s, cand_i, hp_s, hp_i, total;
};
bool pose_ws_plan(int B, int C, int H, int W, int J, int K, PoseWs *p, BandPlan *bp_hm,
                  BandPlan *bp_hp)
{
    if (!make_band_plan(B, C, H, W, K, bp_hm) || !make_band_plan(B, J > 0 ? J : 1, H, W, K, bp_hp))
        return false;
    const size_t n_hm = (size_t)B * C * bp_hm->nbands * K;
    const size_t n_hp = (size_t)B * J * bp_hp->nbands * K;
    const size_t n = n_hm > n_hp ? n_hm : n_hp;  // candidate buffers are reused by both stages
    size_t o = 0;
    p->cand_s = o; o += cn_align_up(n * 4, 256);
    p->cand_i = o; o += cn_align_up(n * 4, 256);
    p->hp_s = o;   o += cn_align_up((size_t)B * J * K * 4, 256);
    p->hp_i = o;   o += cn_align_up((size_t)B * J * K * 4, 256);
    p->total = o;
    return true;
}

}  // namespace

extern "C" size_t cn_multi_pose_decode_workspace_bytes(int B, int C, int H, int W, int J, int K)
{
    PoseWs p;
    BandPlan a, b;
    if (B <= 0 || C <= 0 || J <= 0 || K <= 0 || !pose_ws_plan(B, C, H, W, J, K, &p, &a, &b)) return 0;
    return p.total;
}

extern "C" int cn_multi_pose_decode_f32(const float *heat, const float *wh, const float *kps,
                                        const float *reg, const float *hm_hp,
                                        const float *hp_offset, int B, int C, int H, int W, int J,
                                        int K, int apply_sigmoid, float *dets, void *workspace,
                                        size_t workspace_bytes, void *stream)
{
    BandPlan bp, bph;
    int rc = decode_checks(heat, B, C, H, W, K, &bp);
    if (rc != CN_OK) return rc;
    if (!wh || !kps || !dets || !workspace) return CN_ERR_NULL;
    if (J <= 0) return CN_ERR_SHAPE;
    PoseWs p;
    if (!pose_ws_plan(B, C, H, W, J, K, &p, &bp, &bph)) return CN_ERR_UNSUPPORTED;
    if (workspace_bytes < p.total) return CN_ERR_WORKSPACE;
    if (hm_hp && ((W & 3) == 0) && !cn_aligned16(hm_hp)) return CN_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)workspace;
    float *cand_s = (float *)(ws + p.cand_s);
    int32_t *cand_i = (int32_t *)(ws + p.cand_i);
    float *hp_s = (float *)(ws + p.hp_s);
    int32_t *hp_i = (int32_t *)(ws + p.hp_i);
    const int D = 4 + 1 + 2 * J + 1;
    // stage A
    rc = launch_nms_topk(heat, B, C, H, W, K, apply_sigmoid, bp, cand_s, cand_i, st);
    if (rc != CN_OK) return rc;
    hipLaunchKernelGGL(merge_topk_kernel<MODE_POSE>, dim3(B), dim3(NTM), sizeof(SelShared), st,
                       cand_s, cand_i, C * bp.nbands * K, bp.nbands * K, H, W, K, C, wh, reg, 0,
                       dets, D, (int32_t *)nullptr, (float *)nullptr, kps, J, (int32_t *)nullptr);
    CN_CHECK_LAUNCH();
    if (!hm_hp) return CN_OK;
    // stage B: per-joint top-K of the keypoint heat-map
    if (bph.nbands == 1) {
        rc = launch_nms_topk(hm_hp, B, J, H, W, K, apply_sigmoid, bph, hp_s, hp_i, st);
        if (rc != CN_OK) return rc;
    } else {
        rc = launch_nms_topk(hm_hp, B, J, H, W, K, apply_sigmoid, bph, cand_s, cand_i, st);
        if (rc != CN_OK) return rc;
        hipLaunchKernelGGL(merge_topk_kernel<MODE_CHANNEL>, dim3(B * J), dim3(NTM),
                           sizeof(SelShared), st, cand_s, cand_i, bph.nbands * K, bph.nbands * K, H,
                           W, K, J, (const float *)nullptr, (const float *)nullptr, 0,
                           (float *)nullptr, 0, hp_i, hp_s, (const float *)nullptr, 0,
                           (int32_t *)nullptr);
        CN_CHECK_LAUNCH();
    }
    // stage C
    hipLaunchKernelGGL(pose_match_kernel, dim3(J, B), dim3(KMAX), 0, st, hp_s, hp_i, hp_offset,
                       dets, J, K, H, W, D);
    CN_CHECK_LAUNCH();
    return CN_OK;
}


// ---------------------------------------------------------------------------
// _topk / _transpose_and_gather_feat / ddd_decode (models/decode.py:103-119,
// models/utils.py:12-26, models/decode.py:426-462)
// ---------------------------------------------------------------------------
namespace {

// out[b,k,c] = feat[b,c,ind[b,k]]  (NCHW map -> (B,K,C); replaces permute+contiguous+gather)
__global__ void gather_feat_kernel(const float *__restrict__ feat, const int32_t *__restrict__ inds,
                                   float *__restrict__ out, int C, int HW, int K, size_t total)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    const size_t bk = i / C;
    const size_t b = bk / K;
    // an index outside the map never reads out of bounds (the Python wrapper raises for it, as
    // torch.gather does; a raw C-ABI caller gets NaN in that row)
    const int32_t ind = inds[bk];
    out[i] = ((uint32_t)ind < (uint32_t)HW) ? feat[(b * C + c) * HW + ind]
                                             : __builtin_nanf("");
}

// one thread per (b,k): [xs, ys, score, rot(8), depth, dim(3), (wh(2),) cls]  (decode.py:433-460)
__global__ void ddd_assemble_kernel(const float *__restrict__ scores, const int32_t *__restrict__ inds,
                                    const int32_t *__restrict__ clses, const float *__restrict__ rot,
                                    const float *__restrict__ depth, const float *__restrict__ dim,
                                    const float *__restrict__ wh, const float *__restrict__ reg,
                                    float *__restrict__ dets, int B, int K, int H, int W)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * K) return;
    const int b = i / K;
    const int HW = H * W;
    const int ind = inds[i];
    const int yi = ind / W, xi = ind - yi * W;
    float xs = (float)xi, ys = (float)yi;
    if (reg) {
        xs = xs + reg[((size_t)b * 2 + 0) * HW + ind];
        ys = ys + reg[((size_t)b * 2 + 1) * HW + ind];
    } else {
        xs = xs + 0.5f;
        ys = ys + 0.5f;
    }
    const int D = wh ? 18 : 16;
    float *d = dets + (size_t)i * D;
    d[0] = xs; d[1] = ys; d[2] = scores[i];
    for (int c = 0; c < 8; ++c) d[3 + c] = rot[((size_t)b * 8 + c) * HW + ind];
    d[11] = depth[(size_t)b * HW + ind];
    for (int c = 0; c < 3; ++c) d[12 + c] = dim[((size_t)b * 3 + c) * HW + ind];
    if (wh) {
        d[15] = wh[((size_t)b * 2 + 0) * HW + ind];
        d[16] = wh[((size_t)b * 2 + 1) * HW + ind];
    }
    d[D - 1] = (float)clses[i];
}

}  // namespace

extern "C" int cn_topk_f32(const float *heat, int B, int C, int H, int W, int K, int apply_sigmoid,
                           float *scores, int32_t *inds, int32_t *clses, void *workspace,
                           size_t workspace_bytes, void *stream)
{
    BandPlan bp;
    int rc = decode_checks(heat, B, C, H, W, K, &bp);
    if (rc != CN_OK) return rc;
    if (!scores || !inds || !clses || !workspace) return CN_ERR_NULL;
    const size_t need = cn_ctdet_decode_workspace_bytes(B, C, H, W, K);
    if (workspace_bytes < need) return CN_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)B * C * bp.nbands * K;
    float *cand_score = (float *)workspace;
    int32_t *cand_idx = (int32_t *)((char *)workspace + cn_align_up(n * sizeof(float), 256));
    const ImgPlan ip = make_img_plan(B, C, H, W, K, bp);
    if (ip.use && !(apply_sigmoid & 2048)) {
        const EmitArgs ea = {nullptr, nullptr, 0, nullptr, 0, inds, scores, clses};
        return launch_image_topk<MODE_TOPK>(heat, B, C, H, W, K, apply_sigmoid, ip, (char *)workspace, ea, st);
    }
    rc = launch_nms_topk(heat, B, C, H, W, K, apply_sigmoid, bp, cand_score, cand_idx, st);
    if (rc != CN_OK) return rc;
    hipLaunchKernelGGL(merge_topk_kernel<MODE_TOPK>, dim3(B), dim3(NTM), sizeof(SelShared), st,
                       cand_score, cand_idx, C * bp.nbands * K, bp.nbands * K, H, W, K, C,
                       (const float *)nullptr, (const float *)nullptr, 0, (float *)nullptr, 0, inds,
                       scores, (const float *)nullptr, 0, clses);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

extern "C" int cn_gather_feat_f32(const float *feat, const int32_t *inds, float *out, int B, int C,
                                  int H, int W, int K, void *stream)
{
    if (!feat || !inds || !out) return CN_ERR_NULL;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || K <= 0) return CN_ERR_SHAPE;
    const size_t total = (size_t)B * K * C;
    hipLaunchKernelGGL(gather_feat_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, feat, inds, out, C, H * W, K, total);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

extern "C" size_t cn_ddd_decode_workspace_bytes(int B, int C, int H, int W, int K)
{
    const size_t base = cn_ctdet_decode_workspace_bytes(B, C, H, W, K);
    if (!base) return 0;
    return base + 3 * cn_align_up((size_t)B * K * 4, 256);
}

extern "C" int cn_ddd_decode_f32(const float *heat, const float *rot, const float *depth,
                                 const float *dim, const float *wh, const float *reg, int B, int C,
                                 int H, int W, int K, int apply_sigmoid, float *dets,
                                 void *workspace, size_t workspace_bytes, void *stream)
{
    if (!rot || !depth || !dim || !dets || !workspace) return CN_ERR_NULL;
    const size_t base = cn_ctdet_decode_workspace_bytes(B, C, H, W, K);
    if (!base) return CN_ERR_UNSUPPORTED;
    if (workspace_bytes < cn_ddd_decode_workspace_bytes(B, C, H, W, K)) return CN_ERR_WORKSPACE;
    const size_t slot = cn_align_up((size_t)B * K * 4, 256);
    float *scores = (float *)((char *)workspace + base);
    int32_t *inds = (int32_t *)((char *)workspace + base + slot);
    int32_t *clses = (int32_t *)((char *)workspace + base + 2 * slot);
    int rc = cn_topk_f32(heat, B, C, H, W, K, apply_sigmoid, scores, inds, clses, workspace, base,
                         stream);
    if (rc != CN_OK) return rc;
    hipLaunchKernelGGL(ddd_assemble_kernel, dim3(cn_cdiv(B * K, 128)), dim3(128), 0,
                       (hipStream_t)stream, scores, inds, clses, rot, depth, dim, wh, reg, dets, B, K,
                       H, W);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

// ---------------------------------------------------------------------------
// exct_decode (models/decode.py:273-424; aggr_weight > 0: cn_exct_aggregate_f32 in front): ExtremeNet-style grouping of the
// K best top / left / bottom / right extreme points into boxes.
//   stage A  cn_topk_f32 x4 (_nms + _topk of the four extreme-point heat-maps)
//   stage B  exct_score_kernel: the K^4 candidate scores (decode.py:316-366), written once
//   stage C  exct_select_kernel: exact top-num_dets (radix select + 1024-key bitonic sort in
//            LDS) and the box / extreme-point assembly (decode.py:372-420)
// Candidate index = ((t*K + l)*K + b)*K + r, as the reference's view(batch, -1).
// Tie order (unspecified by torch.topk): score desc, candidate index asc.
// ---------------------------------------------------------------------------
namespace {

constexpr int EXCT_MAX_DETS = 1024;

struct ExctLists {  // (B, K) arrays of the four _topk calls, order t, l, b, r
    const float *score[4];
    const int32_t *ind[4];
    const int32_t *cls[4];
};

__device__ __forceinline__ float exct_score(const ExctLists &L, const float *__restrict__ ct_heat,
                                            int b, int K, int H, int W, int C, int it, int il,
                                            int ib, int ir, float scores_thresh, float center_thresh)
{
    const int o = b * K;
    const float ts = L.score[0][o + it], ls = L.score[1][o + il], bs = L.score[2][o + ib],
                rs = L.score[3][o + ir];
    const int ti = L.ind[0][o + it], li = L.ind[1][o + il], bi = L.ind[2][o + ib],
              ri = L.ind[3][o + ir];
    const int tc = L.cls[0][o + it], lc = L.cls[1][o + il], bc = L.cls[2][o + ib],
              rc = L.cls[3][o + ir];
    const float t_y = (float)(ti / W), t_x = (float)(ti % W);
    const float l_y = (float)(li / W), l_x = (float)(li % W);
    const float b_y = (float)(bi / W), b_x = (float)(bi % W);
    const float r_y = (float)(ri / W), r_x = (float)(ri % W);
    // decode.py:331-336: centre of the box -> centre heat-map of the TOP point's class
    const int cx = (int)((l_x + r_x + 0.5f) / 2.0f);
    const int cy = (int)((t_y + b_y + 0.5f) / 2.0f);
    const float cs = ct_heat[((size_t)b * C + tc) * H * W + (size_t)cy * W + cx];
    // decode.py:343: (t + l + b + r + 2*ct) / 6, left to right
    float s = ((((ts + ls) + bs) + rs) + 2.0f * cs) / 6.0f;
    // decode.py:346-366: one unit off per violated rule, in the reference's order
    const bool sc_bad = (ts < scores_thresh) || (ls < scores_thresh) || (bs < scores_thresh) ||
                        (rs < scores_thresh) || (cs < center_thresh);
    const bool cls_bad = (tc != lc) || (tc != bc) || (tc != rc);
    const bool top_bad = (t_y > l_y) || (t_y > b_y) || (t_y > r_y);
    const bool left_bad = (l_x > t_x) || (l_x > b_x) || (l_x > r_x);
    const bool bottom_bad = (b_y < t_y) || (b_y < l_y) || (b_y < r_y);
    const bool right_bad = (r_x < t_x) || (r_x < l_x) || (r_x < b_x);
    s = s - (sc_bad ? 1.0f : 0.0f);
    s = s - (cls_bad ? 1.0f : 0.0f);
    s = s - (top_bad ? 1.0f : 0.0f);
    s = s - (left_bad ? 1.0f : 0.0f);
    s = s - (bottom_bad ? 1.0f : 0.0f);
    s = s - (right_bad ? 1.0f : 0.0f);
    return s;
}

__global__ void exct_score_kernel(const ExctLists L, const float *__restrict__ ct_heat,
                                  float *__restrict__ cand, int K, int H, int W, int C,
                                  float scores_thresh, float center_thresh)
{
    const int b = blockIdx.y;
    const size_t n = (size_t)K * K * K * K;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        const int ir = (int)(i % K), ib = (int)((i / K) % K), il = (int)((i / ((size_t)K * K)) % K),
                  it = (int)(i / ((size_t)K * K * K));
        cand[(size_t)b * n + i] =
            exct_score(L, ct_heat, b, K, H, W, C, it, il, ib, ir, scores_thresh, center_thresh);
    }
}

struct ExctRegr {
    const float *r[4];  // t, l, b, r regression maps (B,2,H,W) or all null
};

// one 1024-thread workgroup per image
// ``cls_map`` (B, H, W) or null: the class-agnostic form (agnex_ct_decode, decode.py:171-187,262-263) takes a
// detection's class from the centre map's arg-max at the box centre instead of from the top point's list
__global__ __launch_bounds__(NTM) void exct_select_kernel(
    const float *__restrict__ cand, const ExctLists L, const ExctRegr R, float *__restrict__ dets,
    int K, int H, int W, int num_dets, const int32_t *__restrict__ cls_map)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    SelShared &sh = *reinterpret_cast<SelShared *>(smem);
    u64 *big = reinterpret_cast<u64 *>(smem + sizeof(SelShared));  // [EXCT_MAX_DETS]
    const int tid = threadIdx.x, b = blockIdx.x;
    const uint32_t n = (uint32_t)K * K * K * K;
    const float *cs = cand + (size_t)b * n;

    auto for_each = [&](auto &&f) {
        for (uint32_t j = tid; j < n; j += NTM) {
            const uint32_t kk = f2key(cs[j] + 0.0f);
            f(((u64)kk << 32) | (u64)(0xFFFFFFFFu - j), false);
        }
    };
    u64 prefix, mask;
    radix_select<NTM>(for_each, (uint32_t)num_dets, sh, prefix, mask);
    // collect the selected keys (exactly num_dets of them) and sort all 1024 slots descending
    __syncthreads();
    if (tid == 0) sh.cnt = 0;
    big[tid] = 0;  // pad keys sort last (NTM == EXCT_MAX_DETS)
    __syncthreads();
    for_each([&](u64 k, bool) {
        if ((k & mask) >= prefix) {
            const uint32_t pos = atomicAdd(&sh.cnt, 1u);
            if (pos < (uint32_t)EXCT_MAX_DETS) big[pos] = k;
        }
    });
    __syncthreads();
    for (int k = 2; k <= EXCT_MAX_DETS; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int partner = tid ^ j;
            if (partner > tid) {
                const u64 x = big[tid], y = big[partner];
                const bool desc = (tid & k) == 0;
                if (desc ? (x < y) : (x > y)) {
                    big[tid] = y;
                    big[partner] = x;
                }
            }
            __syncthreads();
        }
    }
    if (tid >= num_dets) return;
    const u64 key = big[tid];
    const float score = key2f((uint32_t)(key >> 32));
    const uint32_t idx = 0xFFFFFFFFu - (uint32_t)key;
    const int ir = (int)(idx % K), ib = (int)((idx / K) % K), il = (int)((idx / (K * K)) % K),
              it = (int)(idx / (K * K * K));
    const int o = b * K;
    const int HW = H * W;
    const int sel[4] = {it, il, ib, ir};
    float xs[4], ys[4], gx[4], gy[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int ind = L.ind[e][o + sel[e]];
        float x = (float)(ind % W), y = (float)(ind / W);
        gx[e] = x;
        gy[e] = y;
        if (R.r[0]) {  // decode.py:372-390 (all four regression maps given)
            x = x + R.r[e][((size_t)b * 2 + 0) * HW + ind];
            y = y + R.r[e][((size_t)b * 2 + 1) * HW + ind];
        } else {      // decode.py:391-399
            x = x + 0.5f;
            y = y + 0.5f;
        }
        xs[e] = x;
        ys[e] = y;
    }
    float *d = dets + ((size_t)b * num_dets + tid) * 14;
    d[0] = xs[1]; d[1] = ys[0]; d[2] = xs[3]; d[3] = ys[2];  // bboxes = (l_x, t_y, r_x, b_y)
    d[4] = score;
    d[5] = xs[0]; d[6] = ys[0]; d[7] = xs[1]; d[8] = ys[1];
    d[9] = xs[2]; d[10] = ys[2]; d[11] = xs[3]; d[12] = ys[3];
    if (cls_map) {   // decode.py:171-172: the box centre on the grid, from the points BEFORE their offsets
        const int cx = (int)((gx[1] + gx[3] + 0.5f) / 2.0f);
        const int cy = (int)((gy[0] + gy[2] + 0.5f) / 2.0f);
        d[13] = (float)cls_map[(size_t)b * HW + (size_t)cy * W + cx];
    } else {
        d[13] = (float)L.cls[0][o + it];                      // clses = t_clses
    }
}

// agnex_ct_decode's `torch.max(ct_heat, dim=1)` (decode.py:164): per cell the largest value over the classes and
// the FIRST class that holds it
__global__ void channel_max_kernel(const float *__restrict__ heat, float *__restrict__ vmax,
                                   int32_t *__restrict__ imax, int C, int HW, size_t cells)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / HW, p = i - b * HW;
        const float *src = heat + b * (size_t)C * HW + p;
        float best = src[0];
        int arg = 0;
        for (int c = 1; c < C; ++c) {
            const float v = src[(size_t)c * HW];
            if (v > best) {
                best = v;
                arg = c;
            }
        }
        vmax[i] = best;
        imax[i] = arg;
    }
}

}  // namespace

// ---------------------------------------------------------------------------
// Edge aggregation of exct_decode (models/decode.py:17-90, aggr_weight > 0): along a row (_h_aggregate)
// or a column (_v_aggregate) of every plane, from both ends, a running sum that continues while the
// values do not fall -- ret[i] = heat[i] + ret[i-1] * (heat[i] >= heat[i-1]) -- minus the value itself;
// out = (w * first + w * second) + heat in the reference's order of float32 operations.  One thread per
// line, two sequential passes (the recurrence is serial by definition; the maps are small).
// ---------------------------------------------------------------------------
namespace {
__global__ void exct_aggregate_kernel(const float *__restrict__ heat, float *__restrict__ out, int planes,
                                      int H, int W, int horizontal, float w)
{
    const int nline = horizontal ? H : W, len = horizontal ? W : H;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)planes * nline) return;
    const int plane = (int)(t / nline), line = (int)(t - (long)plane * nline);
    const size_t base = (size_t)plane * H * W + (horizontal ? (size_t)line * W : (size_t)line);
    const size_t step = horizontal ? 1 : (size_t)W;
    // first pass (from the start: _left / _top): out holds ret - heat
    float prev_h = heat[base], prev_r = prev_h;
    out[base] = add_rn(prev_r, -prev_h);
    for (int i = 1; i < len; ++i) {
        const float h = heat[base + i * step];
        const float r = add_rn(h, mul_rn(prev_r, (h >= prev_h) ? 1.0f : 0.0f));
        out[base + i * step] = add_rn(r, -h);
        prev_h = h;
        prev_r = r;
    }
    // second pass (from the end: _right / _bottom) and the combination
    prev_h = heat[base + (size_t)(len - 1) * step];
    prev_r = prev_h;
    {
        const size_t o = base + (size_t)(len - 1) * step;
        out[o] = add_rn(add_rn(mul_rn(w, out[o]), mul_rn(w, add_rn(prev_r, -prev_h))), prev_h);
    }
    for (int i = len - 2; i >= 0; --i) {
        const size_t o = base + i * step;
        const float h = heat[o];
        const float r = add_rn(h, mul_rn(prev_r, (h >= prev_h) ? 1.0f : 0.0f));
        out[o] = add_rn(add_rn(mul_rn(w, out[o]), mul_rn(w, add_rn(r, -h))), h);
        prev_h = h;
        prev_r = r;
    }
}
}  // namespace

extern "C" int cn_exct_aggregate_f32(const float *heat, float *out, int B, int C, int H, int W,
                                     int horizontal, float aggr_weight, void *stream)
{
    if (!heat || !out) return CN_ERR_NULL;
    if (heat == out) return CN_ERR_UNSUPPORTED;      // two passes read the input
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return CN_ERR_SHAPE;
    const long lines = (long)B * C * (horizontal ? H : W);
    hipLaunchKernelGGL(exct_aggregate_kernel, dim3((unsigned)((lines + 63) / 64)), dim3(64), 0,
                       (hipStream_t)stream, heat, out, B * C, H, W, horizontal ? 1 : 0, aggr_weight);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

namespace {
// decode.py:297-305 for one edge map: heat * (max_pool2d(heat, 3, 1, 1) == heat), then values > 1 set
// to 1.  (Clamping BEFORE the peak test would turn every plateau of clamped cells into peaks.)
__global__ void exct_nms_clamp_kernel(const float *__restrict__ heat, float *__restrict__ out, int H, int W,
                                      size_t total)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        const float v = heat[i];
        float m = v;
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int yy = y + dy, xx = x + dx;
                if (yy >= 0 && yy < H && xx >= 0 && xx < W) m = fmaxf(m, heat[i + (long)dy * W + dx]);
            }
        out[i] = (m == v) ? fminf(v, 1.0f) : 0.0f;
    }
}
}  // namespace

extern "C" size_t cn_exct_decode_workspace_bytes(int B, int C, int H, int W, int K)
{
    const size_t base = cn_ctdet_decode_workspace_bytes(B, C, H, W, K);
    if (!base) return 0;
    const size_t lists = 4 * 3 * cn_align_up((size_t)B * K * 4, 256);
    const size_t cand = cn_align_up((size_t)B * K * K * K * K * sizeof(float), 256);
    const size_t map = cn_align_up((size_t)B * C * H * W * sizeof(float), 256);   // CN_EXCT_CLAMP_ONE: one peak-tested, clamped edge map
    return base + lists + cand + map;
}

static int exct_decode_impl(const float *t_heat, const float *l_heat, const float *b_heat,
                            const float *r_heat, const float *ct_heat, const float *t_regr,
                            const float *l_regr, const float *b_regr, const float *r_regr,
                            int B, int C, int H, int W, int K, float scores_thresh,
                            float center_thresh, int num_dets, int apply_sigmoid,
                            float *dets, void *workspace, size_t workspace_bytes,
                            void *stream, const int32_t *cls_map)
{
    if (!t_heat || !l_heat || !b_heat || !r_heat || !ct_heat || !dets || !workspace) return CN_ERR_NULL;
    if (num_dets <= 0 || K <= 0) return CN_ERR_SHAPE;
    if (K > 64 || num_dets > EXCT_MAX_DETS) return CN_ERR_UNSUPPORTED;  // K^4 must fit 32 bits / LDS sort
    if ((long)num_dets > (long)K * K * K * K) return CN_ERR_SHAPE;      // torch.topk: k out of range
    if (apply_sigmoid & ~CN_EXCT_CLAMP_ONE) return CN_ERR_UNSUPPORTED;  // the centre map is gathered, not scanned
    const bool clamp_one = (apply_sigmoid & CN_EXCT_CLAMP_ONE) != 0;
    const size_t base = cn_ctdet_decode_workspace_bytes(B, C, H, W, K);
    if (!base) return CN_ERR_UNSUPPORTED;
    if (workspace_bytes < cn_exct_decode_workspace_bytes(B, C, H, W, K)) return CN_ERR_WORKSPACE;
    const bool all_regr = t_regr && l_regr && b_regr && r_regr;  // decode.py:372-373
    const size_t slot = cn_align_up((size_t)B * K * 4, 256);
    char *p = (char *)workspace + base;
    ExctLists L;
    const float *heats[4] = {t_heat, l_heat, b_heat, r_heat};
    for (int e = 0; e < 4; ++e) {
        float *s = (float *)p; p += slot;
        int32_t *i = (int32_t *)p; p += slot;
        int32_t *c = (int32_t *)p; p += slot;
        int rc;
        if (clamp_one) {
            // peak test on the raw map, survivors clamped to 1 (decode.py:297-305), then the plain _topk
            // over every cell of the result (zeros of the suppressed cells take part, as in the reference)
            const size_t cells = (size_t)B * C * H * W;
            float *tmp = (float *)((char *)workspace + base + 4 * 3 * slot +
                                   cn_align_up((size_t)B * K * K * K * K * sizeof(float), 256));
            hipLaunchKernelGGL(exct_nms_clamp_kernel, dim3((unsigned)((cells + 255) / 256 < 8192 ? (cells + 255) / 256 : 8192)),
                               dim3(256), 0, (hipStream_t)stream, heats[e], tmp, H, W, cells);
            CN_CHECK_LAUNCH();
            rc = cn_topk_f32(tmp, B, C, H, W, K, CN_DECODE_NO_PEAK_TEST, s, i, c, workspace, base, stream);
        } else {
            rc = cn_topk_f32(heats[e], B, C, H, W, K, 0, s, i, c, workspace, base, stream);
        }
        if (rc != CN_OK) return rc;
        L.score[e] = s; L.ind[e] = i; L.cls[e] = c;
    }
    float *cand = (float *)p;
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)K * K * K * K;
    dim3 grid((unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096), B);
    hipLaunchKernelGGL(exct_score_kernel, grid, dim3(256), 0, st, L, ct_heat, cand, K, H, W, C,
                       scores_thresh, center_thresh);
    CN_CHECK_LAUNCH();
    ExctRegr R;
    R.r[0] = all_regr ? t_regr : nullptr; R.r[1] = all_regr ? l_regr : nullptr;
    R.r[2] = all_regr ? b_regr : nullptr; R.r[3] = all_regr ? r_regr : nullptr;
    const size_t lds = sizeof(SelShared) + EXCT_MAX_DETS * sizeof(u64);
    hipLaunchKernelGGL(exct_select_kernel, dim3(B), dim3(NTM), lds, st, cand, L, R, dets, K, H, W,
                       num_dets, cls_map);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

extern "C" int cn_exct_decode_f32(const float *t_heat, const float *l_heat, const float *b_heat,
                                  const float *r_heat, const float *ct_heat, const float *t_regr,
                                  const float *l_regr, const float *b_regr, const float *r_regr,
                                  int B, int C, int H, int W, int K, float scores_thresh,
                                  float center_thresh, int num_dets, int apply_sigmoid,
                                  float *dets, void *workspace, size_t workspace_bytes,
                                  void *stream)
{
    return exct_decode_impl(t_heat, l_heat, b_heat, r_heat, ct_heat, t_regr, l_regr, b_regr, r_regr, B, C, H, W, K,
                            scores_thresh, center_thresh, num_dets, apply_sigmoid, dets, workspace,
                            workspace_bytes, stream, nullptr);
}

// agnex_ct_decode (models/decode.py:121-271): the class-agnostic form -- ONE top / left / bottom / right map per
// image, the centre map over C classes.  The grouping is exct_decode's over single-channel maps (every point is
// "class 0": the class rule never fires and subtracts an exact 0) against the per-cell maximum of the centre map;
// a detection's class is that map's arg-max at the box centre.  Workspace: the single-class exct workspace + the
// (B, H, W) maximum and arg-max planes.
extern "C" size_t cn_agnex_ct_decode_workspace_bytes(int B, int C, int H, int W, int K)
{
    const size_t base = cn_exct_decode_workspace_bytes(B, 1, H, W, K);
    if (!base || C <= 0) return 0;
    return base + 2 * cn_align_up((size_t)B * H * W * 4, 256);
}

extern "C" int cn_agnex_ct_decode_f32(const float *t_heat, const float *l_heat, const float *b_heat,
                                      const float *r_heat, const float *ct_heat, const float *t_regr,
                                      const float *l_regr, const float *b_regr, const float *r_regr,
                                      int B, int C, int H, int W, int K, float scores_thresh,
                                      float center_thresh, int num_dets, int flags, float *dets,
                                      void *workspace, size_t workspace_bytes, void *stream)
{
    if (!t_heat || !l_heat || !b_heat || !r_heat || !ct_heat || !dets || !workspace) return CN_ERR_NULL;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return CN_ERR_SHAPE;
    const size_t base = cn_exct_decode_workspace_bytes(B, 1, H, W, K);
    if (!base) return CN_ERR_UNSUPPORTED;
    if (workspace_bytes < cn_agnex_ct_decode_workspace_bytes(B, C, H, W, K)) return CN_ERR_WORKSPACE;
    const size_t cells = (size_t)B * H * W;
    float *vmax = (float *)((char *)workspace + base);
    int32_t *imax = (int32_t *)((char *)workspace + base + cn_align_up(cells * 4, 256));
    const unsigned grid = (unsigned)((cells + 255) / 256 < 8192 ? (cells + 255) / 256 : 8192);
    hipLaunchKernelGGL(channel_max_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, ct_heat, vmax, imax, C,
                       H * W, cells);
    CN_CHECK_LAUNCH();
    return exct_decode_impl(t_heat, l_heat, b_heat, r_heat, vmax, t_regr, l_regr, b_regr, r_regr, B, 1, H, W, K,
                            scores_thresh, center_thresh, num_dets, flags, dets, workspace, base, stream, imax);
}
