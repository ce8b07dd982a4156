// Shared helpers for the gfx950 kernels of libwoft_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/woft_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline int woft_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? WOFT_OK : WOFT_ELAUNCH;
}

// (Measured and removed in round 3: hipExtLaunchKernelGGL(..., hipExtAnyOrderLaunch) -- an AQL packet without the barrier
//  bit, so that a launch that does not depend on its predecessor starts while that one still runs -- is accepted and has no
//  effect on gfx950 / ROCm 7.2, as hip_ext.h says for GFX9: frame times identical to three digits with and without it.)
template <typename K, typename... A>
static inline void woft_launch(int, K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t s, A... args) {
    hipLaunchKernelGGL(kernel, grid, block, lds, s, args...);
}

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
