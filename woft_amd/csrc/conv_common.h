// Pieces shared by the fp32-MFMA and the split-bf16-MFMA implicit-GEMM kernels: the gathered,
// zero-filled A-operand loader (NHWC, two-source channel concat, "flat" tiny-Cin packing) and the
// fused epilogue.  See conv.hip for the GEMM formulation.
#pragma once
#include "common.h"

namespace woft {

constexpr int BK = 32;

template <int RA>
struct ARows {
    int iy0[RA], ix0[RA];
    int64_t img_base[RA];
    bool mvalid[RA];
};

template <int RA>
__device__ __forceinline__ void a_rows_init(const woft_conv_params& p, int64_t m0, int r0, int64_t M, ARows<RA>& a) {
    const int hw = p.ho * p.wo;
#pragma unroll
    for (int j = 0; j < RA; ++j) {
        int64_t m = m0 + r0 + 32 * j;
        a.mvalid[j] = m < M;
        if (!a.mvalid[j]) m = 0;
        const int img = (int)(m / hw);
        const int rem = (int)(m - (int64_t)img * hw);
        const int oy = rem / p.wo, ox = rem - oy * p.wo;
        a.iy0[j] = oy * p.stride - p.pad_y;
        a.ix0[j] = ox * p.stride - p.pad_x;
        a.img_base[j] = (int64_t)img * p.h * p.w;
    }
}

// K step ks -> (tap, channel chunk); loads this thread's RA float4s (rows r0 + 32 j, floats 4v..4v+3)
template <int RA>
__device__ __forceinline__ void a_load(const woft_conv_params& p, const ARows<RA>& a, int ks, int nchunk, int v,
                                       f32x4 (&ra)[RA]) {
    const int tap = ks / nchunk;
    const int c0 = (ks - tap * nchunk) * BK;
    const int ky = tap / p.taps_x, kx = tap - ky * p.taps_x;
    if (!p.flat) {
        const bool second = (p.in1 != nullptr) && (c0 >= p.c_split);
        const float* src = second ? p.in1 : p.in0;
        const int cs = second ? p.cs1 : p.cs0;
        const int cc = (second ? c0 - p.c_split : c0) + 4 * v;
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            const int iy = a.iy0[j] + ky, ix = a.ix0[j] + kx;
            const bool ok = a.mvalid[j] && iy >= 0 && iy < p.h && ix >= 0 && ix < p.w;
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (ok) val = *(const f32x4*)(src + (a.img_base[j] + (int64_t)iy * p.w + ix) * cs + cc);
            ra[j] = val;
        }
    } else {
        const int dpix = (4 * v) / p.cs0;
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            const int iy = a.iy0[j] + ky, ixp = a.ix0[j] + dpix;
            const bool ok = a.mvalid[j] && iy >= 0 && iy < p.h && ixp >= 0 && ixp < p.w;
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (ok) val = *(const f32x4*)(p.in0 + (a.img_base[j] + (int64_t)iy * p.w + a.ix0[j]) * p.cs0 + 4 * v);
            ra[j] = val;
        }
    }
}

// Fused epilogue on the 32x32 MFMA accumulators of one wave.
// C/D layout: column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
template <int BM, int BN>
__device__ __forceinline__ void conv_epilogue(const woft_conv_params& p, f32x16 (&acc)[BM / 64][BN / 64], int64_t m0,
                                              int n0, int wm, int wn, int r32, int hh, int64_t M) {
    constexpr int TM = BM / 64, TN = BN / 64;
    const bool do_stats = p.stat_sum != nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / 2) + j * 32 + r32;
        const bool nvalid = n < p.cout;
        const float bias = (p.bias != nullptr) ? p.bias[n] : 0.f;
        int64_t col = n;
        if (p.out_pitch != 0) col = (int64_t)(n / p.out_w) * p.out_pitch + (n % p.out_w);
        col += p.co_off;
        float ssum = 0.f, ssq = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                const bool ok = nvalid && (m < M);
                float y = p.alpha * acc[i][j][r] + bias;
                if (do_stats && ok) { ssum += y; ssq += y * y; }
                if (!ok) continue;
                switch (p.epi) {
                    case WOFT_EPI_LINEAR: break;
                    case WOFT_EPI_RELU: y = fmaxf(y, 0.f); break;
                    case WOFT_EPI_SIGMOID: y = sigmoidf_(y); break;
                    case WOFT_EPI_TANH: y = tanhf(y); break;
                    case WOFT_EPI_RELU_RES_RELU:
                        y = fmaxf(p.e0[m * p.lde0 + n] + fmaxf(y, 0.f), 0.f);
                        break;
                    case WOFT_EPI_GRU_ZR:
                        y = sigmoidf_(y);
                        if (n >= p.split) {
                            p.out1[m * p.ldo1 + (n - p.split)] = y * p.e0[m * p.lde0 + (n - p.split)];
                            continue;
                        }
                        break;
                    case WOFT_EPI_GRU_Q: {
                        const float z = p.e1[m * p.lde1 + n], hprev = p.e0[m * p.lde0 + n];
                        y = (1.f - z) * hprev + z * tanhf(y);
                    } break;
                    case WOFT_EPI_CTX: y = (n < p.split) ? tanhf(y) : fmaxf(y, 0.f); break;
                    default: break;
                }
                p.out[m * p.ldo + col] = y;
            }
        }
        if (do_stats) {
            ssum += __shfl_xor(ssum, 32);
            ssq += __shfl_xor(ssq, 32);
            if (hh == 0) {
                const int64_t row = (int64_t)blockIdx.x * 2 + wm;
                p.stat_sum[row * p.cout_pad + n] = ssum;
                p.stat_sq[row * p.cout_pad + n] = ssq;
            }
        }
    }
}

}  // namespace woft
