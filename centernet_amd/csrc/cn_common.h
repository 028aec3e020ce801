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
