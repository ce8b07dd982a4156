// "MXP": the producer-side operand form of precision 4 (f16mx8) -- per pixel and 32-channel block 128 bytes in place of 32 floats:
//   bytes   0 ..  63  fp16(a[0..31])
//   bytes  64 ..  95  fp8 e4m3 of a[0..31] / 2^(s - 127)
//   bytes  96 .. 127  fp8 e4m3 of (a - fp16(a))[0..31] / 2^(s - 11 - 127)
// The block's E8M0 scale s is NOT stored: writer and reader derive it from the fp16 plane, s = (largest fp16 exponent field of the
// block) + 105, i.e. the largest |fp16(a)| scaled into [128, 256) (e4m3 holds 448); the remainder a - fp16(a) is at most 2^-11 of its
// element, so s - 11 serves it.  A lane holds four channels; the eight lanes of a block are an aligned group of 8 (the epilogue's
// lane & 7, the loaders' v): block maxima by three DPP steps (half-row mirror, two quad swaps).
// Included by conv_regb.hip (precision-4 parts: epilogue stores behind WOFT_EPI_MXP, copy-only loader) and elementwise.hip (woft_pack_split)
// only -- the epilogue the other conv kernels share stays byte-identical (round 4: growing it broke their code generation).
#pragma once
#include "halo_map.h"

namespace {

__device__ __forceinline__ uint32_t dpp_max8_u32(uint32_t x) {
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x141, 0xf, 0xf, false));   // row_half_mirror
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    return x;
}
struct MxpWords { uint32_t h0, h1, qa, ql; };
__device__ __forceinline__ MxpWords mxp_pack(const f32x4 y) {          // this lane's four channels of its block
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    const bf16x4 h = cvt16<16>(y);
    const u32x2_t hb = __builtin_bit_cast(u32x2_t, h);
    const uint32_t a0 = hb[0] & 0x7fff7fffu, a1 = hb[1] & 0x7fff7fffu;
    uint32_t m = max(max(a0 & 0xffffu, a0 >> 16), max(a1 & 0xffffu, a1 >> 16));
    m = dpp_max8_u32(m);
    const uint32_t sa = (m >> 10) + 105u;                               // >= 105: the remainder's scale sa - 11 is valid
    const float ia = __builtin_bit_cast(float, (254u - sa) << 23), il = __builtin_bit_cast(float, (265u - sa) << 23);
    const f32x4 la = y - __builtin_convertvector(__builtin_bit_cast(f16x4, h), f32x4);
    int qa = 0, ql = 0;
    qa = __builtin_amdgcn_cvt_pk_fp8_f32(y[0] * ia, y[1] * ia, qa, false);
    qa = __builtin_amdgcn_cvt_pk_fp8_f32(y[2] * ia, y[3] * ia, qa, true);
    ql = __builtin_amdgcn_cvt_pk_fp8_f32(la[0] * il, la[1] * il, ql, false);
    ql = __builtin_amdgcn_cvt_pk_fp8_f32(la[2] * il, la[3] * il, ql, true);
    return MxpWords{hb[0], hb[1], (uint32_t)qa, (uint32_t)ql};
}

}  // namespace
