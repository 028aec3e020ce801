/*
 * oracle/ref_cuda_host_shim.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Lets g++ compile the reference's own DCNv2 sampling kernels
 * (/root/reference/src/lib/models/networks/DCNv2/src/cuda/dcn_v2_im2col_cuda.cu)
 * as plain host C++, from where they lie, so that `oracle/_ref/libdcn_v2_ref.so` runs the
 * reference's arithmetic itself (see oracle/Makefile, target _ref).  Nothing here restates
 * the algorithm: the shim only supplies the CUDA vocabulary the file expects.
 *
 *   __global__ / __device__    -> nothing (ordinary functions)
 *   blockIdx/threadIdx = 0, blockDim/gridDim = 1
 *        -> CUDA_KERNEL_LOOP (dcn_v2_im2col_cuda.cu:6-9) becomes `for (i = 0; i < n; i += 1)`:
 *           one host "thread" walks the whole grid-stride loop.
 *   kernel<<<grid, block, 0, stream>>>(args)
 *        -> the Makefile recipe deletes the `<<<...>>>` launch configuration on the fly
 *           (the stream is piped into g++, no modified source is ever written), leaving a
 *           plain call `kernel(args)`.
 *   atomicAdd (col2im, backward only) -> sequential add.
 */
#ifndef CN_REF_CUDA_HOST_SHIM_H
#define CN_REF_CUDA_HOST_SHIM_H
#include <math.h>
#include <stdio.h>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline

struct cn_ref_dim3 { int x, y, z; };
static const cn_ref_dim3 blockIdx = {0, 0, 0};
static const cn_ref_dim3 threadIdx = {0, 0, 0};
static const cn_ref_dim3 blockDim = {1, 1, 1};
static const cn_ref_dim3 gridDim = {1, 1, 1};

typedef void *cudaStream_t;
typedef int cudaError_t;
static const cudaError_t cudaSuccess = 0;
static inline cudaError_t cudaGetLastError() { return cudaSuccess; }
static inline const char *cudaGetErrorString(cudaError_t) { return "host build"; }
static inline float atomicAdd(float *addr, float v) { float old = *addr; *addr = old + v; return old; }

#endif
