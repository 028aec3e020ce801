// How far behind an MFMA may an LDS read overwrite that MFMA's A operand?  ("operand hazard")
//
// A victim wave issues   MFMA_0 (A = ra, B = rb)  ->  D further MFMAs on other registers  ->  a burst of six
// ds_read_b128, the first of them INTO ra (WHICH = 0) or rb (WHICH = 1)  ->  AFTER more MFMAs  ->  waits.
// ra / rb hold ones, the LDS line holds twos, the other MFMAs multiply by zero: without the hazard every
// accumulator element ends at 16 * trips; each time MFMA_0 picks up a register the read has already
// overwritten, the rows / columns fed by that lane gain 8.  Contention for the SIMD's matrix pipe
// (which is what lets issued MFMAs queue up) comes from a second wave per SIMD: another victim, or a wave that
// issues nothing but MFMAs.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_operand_hazard.hip -o tools/ubench/mfma_operand_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

// mode 0: every wave a victim; mode 1: waves 4 .. of the block issue MFMAs only
template <int D, int AFTER, int WHICH>
__global__ __launch_bounds__(1024) void k(float *out, int trips, int mode)
{
    __shared__ u4 twos[1024];
    twos[threadIdx.x] = u4{0x40004000u, 0x40004000u, 0x40004000u, 0x40004000u};   // 2.0 halves
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    f16v acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    unsigned one = 0x3c003c00u, zero = 0u;
    asm volatile("" : "+v"(one), "+v"(zero));
    const h8 rc = __builtin_bit_cast(h8, u4{one, one, one, one});
    const h8 rz = __builtin_bit_cast(h8, u4{zero, zero, zero, zero});
    if (mode == 1 && wave >= 4) {
        f16v a2 = acc0, a3 = acc0;
        for (int t = 0; t < trips * (1 + D + AFTER) / 2; ++t) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(rc, rz, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(rc, rz, acc1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(rc, rz, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(rc, rz, a3, 0, 0, 0);
        }
        float r = 0.f;
        for (int j = 0; j < 16; ++j) r += acc0[j] + acc1[j] + a2[j] + a3[j];
        if (r == 12345.678f) out[0] = r;
        return;
    }
    const unsigned addr = (unsigned)(size_t)&twos[threadIdx.x];
    for (int t = 0; t < trips; ++t) {
        unsigned o = one;
        asm volatile("" : "+v"(o));
        h8 ra = __builtin_bit_cast(h8, u4{o, o, o, o});
        asm volatile("" : "+v"(o));
        h8 rb = __builtin_bit_cast(h8, u4{o, o, o, o});
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ra, rb, acc0, 0, 0, 0);          // the last reader of ra, rb
#pragma unroll
        for (int d = 0; d < D; ++d) {       // (the kernels' pattern: two accumulator chains, alternating)
            if (d & 1) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(rc, rz, acc0, 0, 0, 0);
            else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(rc, rz, acc1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        u4 j0, j1, j2, j3, j4;
        if (WHICH == 0)
            asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %6\n\tds_read_b128 %4, %6\n\tds_read_b128 %5, %6"
                         : "+v"(ra), "=v"(j0), "=v"(j1), "=v"(j2), "=v"(j3), "=v"(j4) : "v"(addr) : "memory");
        else
            asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %6\n\tds_read_b128 %4, %6\n\tds_read_b128 %5, %6"
                         : "+v"(rb), "=v"(j0), "=v"(j1), "=v"(j2), "=v"(j3), "=v"(j4) : "v"(addr) : "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int d = 0; d < AFTER; ++d) {
            if (d & 1) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(rc, rz, acc0, 0, 0, 0);
            else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(rc, rz, acc1, 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra), "+v"(rb), "+v"(j0), "+v"(j1), "+v"(j2), "+v"(j3), "+v"(j4) :: "memory");
    }
    // deviation of the worst element of this lane, in events (8 per overwritten lane and MFMA)
    float worst = 0.f;
    for (int j = 0; j < 16; ++j) worst = fmaxf(worst, fabsf(acc0[j] - 16.f * trips) / 8.f);
    for (int j = 0; j < 16; ++j) worst = fmaxf(worst, fabsf(acc1[j]) / 8.f);
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = worst;
}

template <int D, int AFTER, int WHICH>
void run(float *d, int mode, int threads, int blocks, const char *what)
{
    const int trips = 20000;
    hipMemset(d, 0, sizeof(float) * 1024 * 1024);
    hipLaunchKernelGGL((k<D, AFTER, WHICH>), dim3(blocks), dim3(threads), 0, 0, d, trips, mode);
    hipDeviceSynchronize();
    std::vector<float> h((size_t)blocks * threads);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    long waves = 0, badw = 0;
    double worst = 0, sum = 0;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < threads / 64; ++w) {
            if (mode == 1 && w >= 4) continue;
            ++waves;
            float m = 0.f;
            for (int l = 0; l < 64; ++l) m = fmaxf(m, h[(size_t)b * threads + w * 64 + l]);
            badw += m > 0.f;
            sum += m;
            if (m > worst) worst = m;
        }
    printf("%c overwritten, D = %d, %d MFMAs behind the reads, %-34s: %5ld of %5ld victim waves hit; events per wave (worst lane): mean %8.1f  max %6.0f  of %d\n",
           WHICH ? 'B' : 'A', D, AFTER, what, badw, waves, sum / waves, worst, trips);
}

template <int D, int AFTER>
void sweep(float *d)
{
    run<D, AFTER, 0>(d, 0, 512, 256, "2 victims / SIMD");
    run<D, AFTER, 0>(d, 1, 512, 256, "victim + MFMA-only wave / SIMD");
    run<D, AFTER, 0>(d, 0, 768, 512, "6 victims / SIMD (2 blocks / CU)");
    run<D, AFTER, 1>(d, 0, 512, 256, "2 victims / SIMD");
    run<D, AFTER, 1>(d, 1, 512, 256, "victim + MFMA-only wave / SIMD");
    run<D, AFTER, 1>(d, 1, 768, 256, "victim + 2 MFMA-only waves / SIMD");
    run<D, AFTER, 1>(d, 0, 768, 512, "6 victims / SIMD (2 blocks / CU)");
}

int main()
{
    float *d;
    if (hipMalloc(&d, sizeof(float) * 1024 * 1024) != hipSuccess) return 1;
    sweep<0, 2>(d);
    sweep<1, 2>(d);
    sweep<2, 2>(d);
    sweep<2, 4>(d);
    sweep<3, 2>(d);
    sweep<4, 2>(d);
    sweep<6, 2>(d);
    return 0;
}
