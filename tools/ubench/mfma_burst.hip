// Bursts of MFMAs between VALU phases, as in the deformable kernels: per iteration every wave issues NV VALU
// instructions (v_pk_fma_f32, independent) and then NM MFMAs (32x32x16 f16) over NA accumulators.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_burst.hip -o tools/ubench/mfma_burst
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));
template <int NV, int NM, int NA, bool PRIO, int KIND = 1>
__global__ __launch_bounds__(256) void k(float *out, int trips, float seed)
{
    f16v acc[NA];
    for (int i = 0; i < NA; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    h8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (_Float16)(seed + i); y[i] = (_Float16)(seed * 0.5f + threadIdx.x); }
    f2 p[8], q[8];
    for (int i = 0; i < 8; ++i) { p[i] = f2{seed + i, seed - i}; q[i] = f2{seed * 0.5f + i, 1.f}; }
    const f2 sp = {seed * 1.0001f, seed * 0.999f};
    unsigned long long t_m = 0;
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[v & 7]) : "v"(q[v & 7]), "v"(sp));
            if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(p[v & 7][0]) : "v"(q[v & 7][0]), "v"(sp[0]));
            if (KIND == 2) asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "+v"(p[v & 7][1]) : "v"(q[v & 7][1]), "s"(seed), "v"(p[v & 7][0]));
            if (KIND == 3) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p[v & 7][1]) : "v"(q[v & 7][0]), "v"(q[v & 7][1]));
            if (KIND == 4) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(p[v & 7][0]) : "v"(q[v & 7][0]));
        }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long c0 = __builtin_readcyclecounter();
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int m = 0; m < NM; ++m) acc[m % NA] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc[m % NA], 0, 0, 0);
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        t_m += __builtin_readcyclecounter() - c0;
    }
    float r = 0.f;
    for (int i = 0; i < 8; ++i) r += p[i][0] + p[i][1];
    for (int i = 0; i < NA; ++i) for (int j = 0; j < 16; ++j) r += acc[i][j];
    if (r == 12345.678f) out[0] = r;
    if (threadIdx.x == 0 && blockIdx.x == 7) out[1] = (float)(t_m / (double)trips);
}
template <int NV, int NM, int NA, bool PRIO, int KIND = 1>
void run(float *d, int wps)
{
    const int trips = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV, NM, NA, PRIO, KIND>), dim3(256 * wps), dim3(256), 0, 0, d, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, NM, NA, PRIO, KIND>), dim3(256 * wps), dim3(256), 0, 0, d, trips, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    const char *names[] = {"v_fma_f32", "v_pk_fma_f32", "v_fma_mixlo_f16", "v_cvt_pk_f16_f32", "v_xor_b32"};
    printf("%-17s x %3d + %2d mfma (%d acc), %d waves/SIMD: %7.1f ns per wave-iteration per SIMD; one wave's MFMA block %4.0f cycles\n",
           names[KIND], NV, NM, NA, wps, ms * 1e6 / trips / wps, h[1]);
}
template <int KIND>
void sweep(float *d)
{
    run<0, 6, 2, false, KIND>(d, 4);
    run<28, 0, 2, false, KIND>(d, 4);
    run<28, 6, 2, false, KIND>(d, 4);
    run<56, 6, 2, false, KIND>(d, 4);
}
int main()
{
    float *d; hipMalloc(&d, 64);
    sweep<0>(d); sweep<1>(d); sweep<2>(d); sweep<3>(d); sweep<4>(d);
    return 0;
}
