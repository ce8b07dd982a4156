// Row <-> pixel assignment of the LDS-halo convolution kernels (conv.hip: conv_halo_bf16_kernel; conv_regb.hip:
// conv_regb_kernel) and the bf16 LDS row pitch they share.
#pragma once
#include "common.h"

namespace {

constexpr int LDB = 40;      // bf16 tiles: elements per row (80 B: 16-B aligned, conflict-free b128 reads)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// ---- fp16 operating point (precision 3; the reference's `mixed_precision`: autocast = fp16 on its CUDA devices,
// weighted_raft.py:204,215,233) -- the split-bf16 kernels instantiated with TERMS = 16 run ONE product per MFMA on fp16
// operands (v_mfma_f32_32x32x16_f16: bf16's rate, 3 more mantissa bits).  LDS tiles, weight planes and fragments are the
// same 16-bit containers; only the two type-specific operations differ: the fp32 -> 16-bit conversion of the activations
// and the MFMA opcode.  (fp16 saturates at 65504: the activations of this network -- normalised images, ReLU / tanh /
// sigmoid outputs, correlations / sqrt(D), flow in pixels -- stay orders of magnitude below; weights are converted on the
// host, which checks their range.)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
template <int TERMS>
__device__ __forceinline__ bf16x4 cvt16(const f32x4 v) {
    if constexpr (TERMS == 16) return __builtin_bit_cast(bf16x4, __builtin_convertvector(v, f16x4));
    else return __builtin_convertvector(v, bf16x4);
}
// bf16x4 -> f32x4, exact, from the two packed words (a shift / a mask per element).  `__builtin_convertvector(hi, f32x4)` is lowered
// element by element through a second v_cvt_pk_bf16_f32 of the fp32 source plus a shift: 8 instead of 4 vector instructions per
// float4 of every halo conversion -- and vector instructions ADD to the MFMAs' SIMD time.
__device__ __forceinline__ f32x4 widen_bf16x4(const bf16x4 h) {
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    const u32x2_t w = __builtin_bit_cast(u32x2_t, h);
    f32x4 r;
    r[0] = __builtin_bit_cast(float, w[0] << 16);
    r[1] = __builtin_bit_cast(float, w[0] & 0xffff0000u);
    r[2] = __builtin_bit_cast(float, w[1] << 16);
    r[3] = __builtin_bit_cast(float, w[1] & 0xffff0000u);
    return r;
}
template <int TERMS>
__device__ __forceinline__ f32x16 mma16(const bf16x8 a, const bf16x8 b, const f32x16 c) {
    if constexpr (TERMS == 16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// ---- which pixel of the TY x TX patch each MFMA tile row holds ----------------------------------
// ds_read_b128 serves a wave in four groups of 16 lanes -- {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same +32
// -- one LDS cycle per group when the 16 addresses fall into 16 different 16-byte slots of a 256-byte line.  The
// halo rows have an 80-byte pitch (5 slots), so two pixels collide iff their halo row indices are congruent
// mod 16; a tap only adds a constant to all of them.  The row <-> pixel assignment is ours to choose, so it is
// chosen per group:
//   TX == 16: a group = the 16 pixels of ONE patch row (halo rows r .. r+15: all residues)          [tile_row_perm]
//   9 x 9   : the 81 pixels are dealt by residue of (11 ty + tx) mod 16 -- no residue class has more than 6
//             members, there are 6 groups -- so each group gets at most one pixel per residue             [kPerm9]
// (rows without a pixel repeat another pixel of their group: same address = broadcast, and are never stored).
__device__ __forceinline__ constexpr int tile_row_perm(int l) {          // lane (0..31) -> 16 * group + slot
    return l < 4 ? l : l < 12 ? l + 12 : l < 16 ? l - 8 : l < 20 ? l + 8 : l < 28 ? l - 12 : l;
}
constexpr int lane_of_slot(int g, int t) {                               // inverse of tile_row_perm
    return g == 0 ? (t < 4 ? t : t < 8 ? t + 8 : t + 12) : (t < 8 ? t + 4 : t < 12 ? t + 8 : t + 16);
}
struct Perm9 {
    unsigned char v[96];                                                 // pixel index (0..80) | 0x80 if filler
};
constexpr Perm9 make_perm9() {
    Perm9 t{};
    for (int i = 0; i < 96; ++i) t.v[i] = 0xFF;
    int cnt[16] = {};
    for (int pix = 0; pix < 81; ++pix) {
        const int c = ((pix / 9) * 11 + pix % 9) % 16;
        const int k = cnt[c]++;                                          // group 0..5 = (tile k/2, half k%2)
        t.v[(k / 2) * 32 + lane_of_slot(k % 2, c)] = (unsigned char)pix;
    }
    for (int k = 0; k < 6; ++k) {
        int filler = 0;
        for (int c = 0; c < 16; ++c)
            if (t.v[(k / 2) * 32 + lane_of_slot(k % 2, c)] != 0xFF) { filler = t.v[(k / 2) * 32 + lane_of_slot(k % 2, c)]; break; }
        for (int c = 0; c < 16; ++c)
            if (t.v[(k / 2) * 32 + lane_of_slot(k % 2, c)] == 0xFF) t.v[(k / 2) * 32 + lane_of_slot(k % 2, c)] = (unsigned char)(filler | 0x80);
    }
    return t;
}
__device__ const Perm9 kPerm9 = make_perm9();
struct Perm9Mask {
    unsigned m[3];                                                       // bit r of m[i]: row r of tile i holds a pixel
};
constexpr Perm9Mask make_perm9_mask() {
    const Perm9 t = make_perm9();
    Perm9Mask k{};
    for (int i = 0; i < 96; ++i)
        if ((t.v[i] & 0x80) == 0) k.m[i / 32] |= 1u << (i % 32);
    return k;
}

// tile row (0 .. BM-1) -> pixel of the patch (ty * TX + tx) and whether the row holds a pixel at all
template <int TY, int TX>
__device__ __forceinline__ int halo_row_pixel(int row, bool& valid) {
    if constexpr (TY == 9 && TX == 9) {
        const int e = kPerm9.v[row];
        valid = (e & 0x80) == 0;
        return e & 0x7F;
    } else {
        static_assert(TX == 16, "patch width 16 or the 9x9 table");
        const int pl = (row & ~31) + tile_row_perm(row & 31);
        valid = pl < TY * TX;
        return valid ? pl : 0;
    }
}

template <int TY, int TX>
struct HaloRowMap {
    int img0, n_img, y0, x0, ho, wo;
    __device__ __forceinline__ int64_t operator()(int row) const {
        bool valid;
        const int pl = halo_row_pixel<TY, TX>(row, valid);
        if (!valid) return -1;
        const int y = y0 + pl / TX, x = x0 + pl % TX;
        return (y < ho && x < wo) ? ((int64_t)img0 * ho + y) * wo + x : -1;
    }
};

}  // namespace
