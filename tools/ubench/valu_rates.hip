// VALU issue cost per instruction type on gfx950 (wave64), measured with N waves per SIMD:
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rates.hip -o tools/ubench/valu_rates && tools/ubench/valu_rates
// Every variant is a loop of 64 independent instructions of one kind (8 register sets), 2000 trips.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int trips, float seed)
{
    float a[8], b[8];
    f2 p[8], q[8];
    unsigned u[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; b[i] = seed * 0.5f + i; p[i] = f2{a[i], b[i]}; q[i] = f2{b[i], a[i]}; u[i] = threadIdx.x * 2654435761u + i; }
    const float s = seed * 1.0001f;
    const f2 sp = {s, s * 0.999f};
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#define X(i) \
            if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(s)); \
            if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(q[i]), "v"(sp)); \
            if (KIND == 2) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(a[i]), "v"(b[i])); \
            if (KIND == 3) asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "s"(s), "v"(a[i])); \
            if (KIND == 4) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b[i])); \
            if (KIND == 5) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(u[i]) : "v"(u[(i + 1) & 7])); \
            if (KIND == 6) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i])); \
            if (KIND == 7) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[i]) : "v"(q[i])); \
            if (KIND == 8) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(u[i]), "s"(s)); \
            if (KIND == 9) asm volatile("v_max3_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(s)); \
            if (KIND == 10) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(a[i]) : "v"(b[i])); \
            if (KIND == 11) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(u[i]), "v"(u[(i + 1) & 7]));
            REP8(X)
#undef X
        }
    }
    float acc = 0.f;
    for (int i = 0; i < 8; ++i) acc += a[i] + p[i][0] + p[i][1] + (float)u[i];
    if (acc == 12345.678f) out[0] = acc;
}
template <int KIND>
void run(const char *name, float *d)
{
    const int trips = 2000;
    for (int wps : {1, 2, 4}) {      // waves per SIMD: 256 CUs x 4 SIMDs x wps waves
        dim3 grid(256 * wps), blk(256);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, grid, blk, 0, 0, d, 10, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, grid, blk, 0, 0, d, trips, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double inst_per_simd = (double)trips * 64 * wps;
        printf("%-18s waves/SIMD %d: %8.3f ms  -> %6.2f ns per wave-instruction per SIMD (x clock GHz = cycles)\n", name, wps, ms, ms * 1e6 / inst_per_simd);
    }
}
// Do MFMA and VALU instructions of DIFFERENT waves on one SIMD overlap?  512 workgroups of four waves (two waves per
// SIMD).  mode 0: every wave issues `trips` x 16 MFMAs (32x32x16 f16); mode 1: every wave `trips` x 16 x NV v_fma_f32;
// mode 2: even workgroups the MFMAs, odd workgroups the VALU work; mode 3: every wave BOTH, interleaved (1 MFMA + NV VALU).
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int NV>
__global__ __launch_bounds__(256) void kmix(float *out, int trips, int mode, float seed)
{
    f16v acc0 = {}, acc1 = {};
    h8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (_Float16)(seed + i); y[i] = (_Float16)(seed * 0.5f + threadIdx.x); }
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + i + threadIdx.x;
    const float s = seed * 1.0001f;
    const bool do_m = mode == 0 || mode == 3 || (mode == 2 && !(blockIdx.x & 1));
    const bool do_v = mode == 1 || mode == 3 || (mode == 2 && (blockIdx.x & 1));
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (do_m) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc0, 0, 0, 0);
            }
            if (do_v) {
#pragma unroll
                for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[v & 7]) : "v"(a[(v + 3) & 7]), "v"(s));
            }
            if (do_m) {
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc1, 0, 0, 0);
            }
            if (do_v) {
#pragma unroll
                for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[v & 7]) : "v"(a[(v + 3) & 7]), "v"(s));
            }
        }
    }
    float r = 0.f;
    for (int i = 0; i < 8; ++i) r += a[i];
    for (int i = 0; i < 16; ++i) r += acc0[i] + acc1[i];
    if (r == 12345.678f) out[0] = r;
}
template <int NV>
void runmix(float *d)
{
    const int trips = 1000;
    for (int mode = 0; mode < 4; ++mode) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(kmix<NV>, dim3(512), dim3(256), 0, 0, d, 10, mode, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kmix<NV>, dim3(512), dim3(256), 0, 0, d, trips, mode, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("mix NV=%2d mode %d (%s): %8.3f ms  (per wave: %d MFMAs, %d VALU)\n", NV, mode,
               mode == 0 ? "all MFMA" : mode == 1 ? "all VALU" : mode == 2 ? "half the waves MFMA, half VALU" : "every wave both, interleaved",
               ms, trips * 16, trips * 16 * NV);
    }
}
int main()
{
    float *d; hipMalloc(&d, 64);
    run<0>("v_fma_f32", d); run<1>("v_pk_fma_f32", d); run<6>("v_mul_f32", d); run<7>("v_pk_mul_f32", d);
    run<2>("v_cvt_pk_f16_f32", d); run<3>("v_fma_mixlo_f16", d); run<8>("v_fma_mix_f32", d);
    run<4>("v_mov_b32", d); run<5>("v_xor_b32", d); run<9>("v_max3_f32", d); run<10>("v_cndmask_b32", d); run<11>("v_dot2_f32_f16", d);
    runmix<4>(d); runmix<8>(d); runmix<14>(d);
    return 0;
}
