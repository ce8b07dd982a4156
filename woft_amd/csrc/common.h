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

// Gate functions of the split-bf16 / fp16 precisions (round 3): hardware exp2 and reciprocal (v_exp_f32, v_rcp_f32: 1 ulp each)
// instead of libm's expf / tanhf / IEEE division -- ~12 instead of ~35-50 instructions per value.  The GRU's gates are 5 x 16
// sigmoids and 2 x 16 tanh per lane and half step; with the library functions they were 36 k of the 187 k cycles of a fused
// half step (tools/gru_probe.py).  Absolute error <= 3e-7 on values in (0, 1) / (-1, 1) (the operands' own products carry
// 2^-16 relative in these precisions); the exact-fp32 precision keeps the library functions.
__device__ __forceinline__ float exp_fast_(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float sigmoid_fast_(float x) { return __builtin_amdgcn_rcpf(1.0f + exp_fast_(-x)); }
__device__ __forceinline__ float tanh_fast_(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + exp_fast_(2.0f * x)); }
template <bool FAST> __device__ __forceinline__ float sigmoid_t(float x) { return FAST ? sigmoid_fast_(x) : sigmoidf_(x); }
template <bool FAST> __device__ __forceinline__ float tanh_t(float x) { return FAST ? tanh_fast_(x) : tanhf(x); }
