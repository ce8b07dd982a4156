// Implicit-GEMM convolution / NT-GEMM on the gfx950 matrix cores (MI355X).
//
//   C[m][n] = alpha * sum_k A[m][k] * B[n][k] + bias[n]      (then a fused epilogue)
//
//   m : output pixel (image, oy, ox) in NHWC order          (GEMM M, up to ~10^7)
//   n : output channel                                       (GEMM N)
//   k : (tap, input channel); 32 channels of one tap per K step
//
// A rows are gathered from one or two NHWC fp32 sources (channel concat without a copy), zero
// filled outside the image.  B is the pre-packed weight matrix [cout_pad][taps*cin_pad] (K
// contiguous) -- or, for the all-pairs correlation volume (corr.py:62-69), the second feature map.
//
// Block = 256 threads = 4 waves as 2(M) x 2(N); block tile BM x BN in {64,128}^2; each wave owns
// (BM/2) x (BN/2) as 32x32 MFMA tiles.  Two arithmetic modes:
//
//  * precision 0 -- v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulation (a k-ordered
//    fmaf chain), 157 TF peak.  Operand tiles in LDS as fp32, 36-float row pitch (conflict-free
//    ds_read_b128); each lane half consumes its own contiguous 16-wide k range of the step.
//  * precision 1/2 -- v_mfma_f32_32x32x16_bf16 on SPLIT operands: every fp32 operand x is split
//    on the fly into hi = bf16(x) and lo = bf16(x - hi) (weights are pre-split once), and
//      precision 1 ("bf16x3"):  acc += Ahi*Bhi + Ahi*Blo + Alo*Bhi   (~2^-16 relative products)
//      precision 2 ("bf16"):    acc += Ahi*Bhi
//    with fp32 accumulation; the MFMA rate is 16x the fp32 one, so bf16x3 has a 5.3x higher matrix
//    ceiling than precision 0 at near-fp32 accuracy.  LDS tiles are bf16 with an 80-byte row pitch
//    (conflict-free ds_read_b128 fragment reads).
//
// In both modes the global loads for step k+1 are issued into registers before the MFMAs of step k.
#include "conv_common.h"

namespace {

using woft::ARows;
using woft::BK;

constexpr int LDS_LD = 36;   // fp32 tiles: floats per row (144 B: 16-B aligned, conflict-free b128 reads)

template <int BM, int BN>
__global__ __launch_bounds__(256) void conv_mfma_f32_kernel(const woft_conv_params p) {
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int RA = BM / 32, RB = BN / 32;
    constexpr int SMEM_FLOATS = ((BM + BN) * LDS_LD > 4 * woft::STAGE_FLOATS) ? (BM + BN) * LDS_LD : 4 * woft::STAGE_FLOATS;
    __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
    float* As = smem;
    float* Bs = smem + BM * LDS_LD;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r32 = lane & 31, hh = lane >> 5;
    const int v = tid & 7, r0 = tid >> 3;

    const int64_t M = (int64_t)p.n_img * p.ho * p.wo;
    int m_tile, n_tile;
    woft::tile_of_block(blockIdx.x, (int)((M + BM - 1) / BM), p.cout_pad / BN, m_tile, n_tile);
    const int64_t m0 = (int64_t)m_tile * BM;
    const int n0 = n_tile * BN;
    const int nchunk = p.cin_pad / BK;
    const int nk = p.taps_y * p.taps_x * nchunk;
    const int64_t ktot = (int64_t)nk * BK;

    ARows<RA> arows;
    woft::a_rows_init<RA>(p, m0, r0, M, arows);
    const float* brow[RB];
#pragma unroll
    for (int j = 0; j < RB; ++j) brow[j] = p.wgt + (int64_t)(n0 + r0 + 32 * j) * ktot + 4 * v;

    f32x4 ra[RA], rb[RB];
    auto load_tiles = [&](int ks) {
        woft::a_load<RA>(p, arows, ks, nchunk, v, ra);
#pragma unroll
        for (int j = 0; j < RB; ++j) rb[j] = *(const f32x4*)(brow[j] + (int64_t)ks * BK);
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int j = 0; j < RA; ++j) *(f32x4*)(As + (r0 + 32 * j) * LDS_LD + 4 * v) = ra[j];
#pragma unroll
        for (int j = 0; j < RB; ++j) *(f32x4*)(Bs + (r0 + 32 * j) * LDS_LD + 4 * v) = rb[j];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const float* a_frag = As + (wm * (BM / 2) + r32) * LDS_LD + hh * 16;
    const float* b_frag = Bs + (wn * (BN / 2) + r32) * LDS_LD + hh * 16;

    load_tiles(0);
    store_tiles();
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        if (ks + 1 < nk) load_tiles(ks + 1);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x4 a[TM][2], b[TN][2];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                a[i][0] = *(const f32x4*)(a_frag + i * 32 * LDS_LD + half * 8);
                a[i][1] = *(const f32x4*)(a_frag + i * 32 * LDS_LD + half * 8 + 4);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                b[j][0] = *(const f32x4*)(b_frag + j * 32 * LDS_LD + half * 8);
                b[j][1] = *(const f32x4*)(b_frag + j * 32 * LDS_LD + half * 8 + 4);
            }
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s >> 2][s & 3], b[j][s >> 2][s & 3],
                                                                         acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (ks + 1 < nk) {
            store_tiles();
            __syncthreads();
        }
    }
    // the loop ends with a block barrier: operand tiles are dead, reuse the LDS for output staging
    woft::conv_epilogue<BM, BN>(p, acc, smem + wave * woft::STAGE_FLOATS, m0, n0, wm, wn, lane, M, m_tile);
}

// ---- split-bf16 kernel -------------------------------------------------------------------------
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int LDB = 40;      // bf16 tiles: elements per row (80 B: 16-B aligned, conflict-free b128 reads)

template <int BM, int BN, int TERMS, bool DEEP>
__global__ __launch_bounds__(256) void conv_mfma_bf16_kernel(const woft_conv_params p) {
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int RA = BM / 32;            // float4 rows per thread (A, fp32 source)
    constexpr int RB = BN / 64;            // 16-B rows per thread and plane (B, pre-split bf16)
    constexpr int NP = (TERMS == 3) ? 2 : 1;
    constexpr int BUF = (BM + BN) * LDB * NP;          // one LDS stage: A planes then B planes
    constexpr int NBUF = DEEP ? 2 : 1;
    constexpr int SMEM_ELEMS = (NBUF * BUF > 8 * woft::STAGE_FLOATS) ? NBUF * BUF : 8 * woft::STAGE_FLOATS;
    __shared__ __attribute__((aligned(16))) __bf16 smem[SMEM_ELEMS];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r32 = lane & 31, hh = lane >> 5;
    const int v = tid & 7, r0 = tid >> 3;              // A loader: float4 column, base row
    const int vb = tid & 3, rb0 = tid >> 2;            // B loader: 16-B column, base row

    const int64_t M = (int64_t)p.n_img * p.ho * p.wo;
    int m_tile, n_tile;
    woft::tile_of_block(blockIdx.x, (int)((M + BM - 1) / BM), p.cout_pad / BN, m_tile, n_tile);
    const int64_t m0 = (int64_t)m_tile * BM;
    const int n0 = n_tile * BN;
    const int nchunk = p.cin_pad / BK;
    const int nk = p.taps_y * p.taps_x * nchunk;
    const int64_t ktot = (int64_t)nk * BK;

    ARows<RA> arows;
    woft::a_rows_init<RA>(p, m0, r0, M, arows);
    const __bf16* bsrc[NP];
    bsrc[0] = (const __bf16*)p.wgt_hi;
    if (NP == 2) bsrc[NP - 1] = (const __bf16*)p.wgt_lo;

    // Two register sets: tiles are fetched TWO K steps ahead (the operands mostly come from beyond
    // the XCD's L2), two LDS stages: one block barrier per K step.
    f32x4 raA[RA], raB[RA];
    bf16x8 rbA[NP][RB], rbB[NP][RB];
    auto load_tiles = [&](int ks, f32x4 (&ra)[RA], bf16x8 (&rb)[NP][RB]) {
        woft::a_load<RA>(p, arows, ks, nchunk, v, ra);
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
            for (int j = 0; j < RB; ++j)
                rb[pl][j] = *(const bf16x8*)(bsrc[pl] + (int64_t)(n0 + rb0 + 64 * j) * ktot + (int64_t)ks * BK + 8 * vb);
    };
    auto store_tiles = [&](int buf, const f32x4 (&ra)[RA], const bf16x8 (&rb)[NP][RB]) {
        __bf16* As = smem + buf * BUF;
        __bf16* Bs = As + NP * BM * LDB;
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            const bf16x4 hi = __builtin_convertvector(ra[j], bf16x4);
            *(bf16x4*)(As + (r0 + 32 * j) * LDB + 4 * v) = hi;
            if (NP == 2) {
                const f32x4 rem = ra[j] - __builtin_convertvector(hi, f32x4);
                *(bf16x4*)(As + BM * LDB + (r0 + 32 * j) * LDB + 4 * v) = __builtin_convertvector(rem, bf16x4);
            }
        }
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
            for (int j = 0; j < RB; ++j) *(bf16x8*)(Bs + pl * BN * LDB + (rb0 + 64 * j) * LDB + 8 * vb) = rb[pl][j];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // lane (r32, hh) feeds k = s*16 + hh*8 + [0,8) of the step for both operands
    const int a_off = (wm * (BM / 2) + r32) * LDB + hh * 8;
    const int b_off = NP * BM * LDB + (wn * (BN / 2) + r32) * LDB + hh * 8;
    auto compute = [&](int buf) {
        const __bf16* a_frag = smem + buf * BUF + a_off;
        const __bf16* b_frag = smem + buf * BUF + b_off;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 a[NP][TM], b[NP][TN];
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[pl][i] = *(const bf16x8*)(a_frag + pl * BM * LDB + i * 32 * LDB + s * 16);
#pragma unroll
                for (int j = 0; j < TN; ++j) b[pl][j] = *(const bf16x8*)(b_frag + pl * BN * LDB + j * 32 * LDB + s * 16);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (NP == 2) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[NP - 1][i], b[0][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[NP - 1][j], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0][j], acc[i][j], 0, 0, 0);
                }
        }
    };

    if (!DEEP) {
        // one LDS stage, tiles fetched one K step ahead, two barriers per step; lowest register and LDS
        // footprint -> most resident waves
        load_tiles(0, raA, rbA);
        store_tiles(0, raA, rbA);
        __syncthreads();
        for (int ks = 0; ks < nk; ++ks) {
            if (ks + 1 < nk) load_tiles(ks + 1, raA, rbA);
            compute(0);
            __syncthreads();
            if (ks + 1 < nk) {
                store_tiles(0, raA, rbA);
                __syncthreads();
            }
        }
        woft::conv_epilogue<BM, BN>(p, acc, (float*)smem + wave * woft::STAGE_FLOATS, m0, n0, wm, wn, lane, M, m_tile);
        return;
    }
    load_tiles(0, raA, rbA);
    if (nk > 1) load_tiles(1, raB, rbB);
    store_tiles(0, raA, rbA);
    if (nk > 2) load_tiles(2, raA, rbA);
    __syncthreads();
    for (int ks = 0; ks < nk; ks += 2) {
        // even step: stage 0 holds tile ks; set B holds tile ks+1, set A tile ks+2
        if (ks + 1 < nk) store_tiles(1, raB, rbB);
        if (ks + 3 < nk) load_tiles(ks + 3, raB, rbB);
        compute(0);
        __syncthreads();
        if (ks + 1 >= nk) break;
        // odd step: stage 1 holds tile ks+1; set A holds tile ks+2, set B tile ks+3
        if (ks + 2 < nk) store_tiles(0, raA, rbA);
        if (ks + 4 < nk) load_tiles(ks + 4, raA, rbA);
        compute(1);
        __syncthreads();
    }
    // every path leaves the loop through a block barrier: the operand stages are dead, reuse them
    woft::conv_epilogue<BM, BN>(p, acc, (float*)smem + wave * woft::STAGE_FLOATS, m0, n0, wm, wn, lane, M, m_tile);
}

// fp32 matrix -> hi / lo bf16 planes (used for the dynamic B operand of the correlation GEMM)
__global__ void split_bf16_kernel(const float* __restrict__ x, int64_t n4, __bf16* __restrict__ hi,
                                  __bf16* __restrict__ lo) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const f32x4 vv = *(const f32x4*)(x + i * 4);
    const bf16x4 h = __builtin_convertvector(vv, bf16x4);
    *(bf16x4*)(hi + i * 4) = h;
    if (lo != nullptr) *(bf16x4*)(lo + i * 4) = __builtin_convertvector(vv - __builtin_convertvector(h, f32x4), bf16x4);
}

// developer tuning knobs (A/B experiments only; set once at start-up, never from the hot path):
//   [0] 0 = single-stage mainloop (default), 1 = two-stage / two-step-ahead mainloop
int g_tuning[4] = {0, 0, 0, 0};

template <int BM, int BN>
int launch_conv(const woft_conv_params& p, hipStream_t s) {
    const int64_t M = (int64_t)p.n_img * p.ho * p.wo;
    dim3 grid((unsigned)(ceil_div64(M, BM) * (p.cout_pad / BN)));       // 1-D: see woft::tile_of_block
    const bool deep = g_tuning[0] != 0;
    if (p.precision == 0)
        hipLaunchKernelGGL((conv_mfma_f32_kernel<BM, BN>), grid, dim3(256), 0, s, p);
    else if (p.precision == 1 && !deep)
        hipLaunchKernelGGL((conv_mfma_bf16_kernel<BM, BN, 3, false>), grid, dim3(256), 0, s, p);
    else if (p.precision == 1)
        hipLaunchKernelGGL((conv_mfma_bf16_kernel<BM, BN, 3, true>), grid, dim3(256), 0, s, p);
    else if (!deep)
        hipLaunchKernelGGL((conv_mfma_bf16_kernel<BM, BN, 1, false>), grid, dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL((conv_mfma_bf16_kernel<BM, BN, 1, true>), grid, dim3(256), 0, s, p);
    return woft_launch_status();
}

}  // namespace

extern "C" int woft_conv2d(const woft_conv_params* pp, void* stream) {
    if (pp == nullptr) return WOFT_EINVAL;
    const woft_conv_params& p = *pp;
    if (p.in0 == nullptr || p.out == nullptr) return WOFT_EINVAL;
    if (p.precision < 0 || p.precision > 2) return WOFT_EINVAL;
    if (p.precision == 0 && p.wgt == nullptr) return WOFT_EINVAL;
    if (p.precision >= 1 && p.wgt_hi == nullptr) return WOFT_EINVAL;
    if (p.precision == 1 && p.wgt_lo == nullptr) return WOFT_EINVAL;
    if (p.cin_pad <= 0 || p.cin_pad % BK != 0) return WOFT_EINVAL;
    if (p.in1 != nullptr && (p.c_split % BK != 0 || p.c_split <= 0 || p.c_split >= p.cin_pad)) return WOFT_EINVAL;
    if (p.cs0 % 4 != 0 || (p.in1 != nullptr && p.cs1 % 4 != 0)) return WOFT_EINVAL;
    if (p.flat && (p.taps_x != 1 || p.in1 != nullptr || (p.cs0 != 4 && p.cs0 != 8 && p.cs0 != 16 && p.cs0 != 32)))
        return WOFT_EINVAL;
    if (p.n_img <= 0 || p.h <= 0 || p.w <= 0 || p.ho <= 0 || p.wo <= 0 || p.taps_y <= 0 || p.taps_x <= 0 ||
        p.stride <= 0 || p.cout <= 0)
        return WOFT_EINVAL;
    if ((p.tile_m != 64 && p.tile_m != 128) || (p.tile_n != 64 && p.tile_n != 128)) return WOFT_EINVAL;
    if (p.cout_pad % p.tile_n != 0 || p.cout > p.cout_pad) return WOFT_EINVAL;
    if (p.epi < 0 || p.epi > WOFT_EPI_CTX) return WOFT_EINVAL;
    if ((p.epi == WOFT_EPI_RELU_RES_RELU || p.epi == WOFT_EPI_GRU_ZR || p.epi == WOFT_EPI_GRU_Q) && p.e0 == nullptr)
        return WOFT_EINVAL;
    if (p.epi == WOFT_EPI_GRU_Q && p.e1 == nullptr) return WOFT_EINVAL;
    if (p.epi == WOFT_EPI_GRU_ZR && p.out1 == nullptr) return WOFT_EINVAL;
    if (p.e0 != nullptr && p.lde0 % 4 != 0) return WOFT_EINVAL;
    if (p.e1 != nullptr && p.lde1 % 4 != 0) return WOFT_EINVAL;
    if (p.out1 != nullptr && (p.ldo1 % 4 != 0 || p.split % 4 != 0)) return WOFT_EINVAL;
    if ((p.stat_sum == nullptr) != (p.stat_sq == nullptr)) return WOFT_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (p.tile_m == 128 && p.tile_n == 128) return launch_conv<128, 128>(p, s);
    if (p.tile_m == 128 && p.tile_n == 64) return launch_conv<128, 64>(p, s);
    if (p.tile_m == 64 && p.tile_n == 128) return launch_conv<64, 128>(p, s);
    return launch_conv<64, 64>(p, s);
}

extern "C" int woft_set_tuning(int key, int value) {
    if (key < 0 || key >= 4) return WOFT_EINVAL;
    g_tuning[key] = value;
    return WOFT_OK;
}

extern "C" int woft_split_bf16(const float* x, int64_t n, void* hi, void* lo, void* stream) {
    if (!x || !hi || n <= 0 || n % 4 != 0) return WOFT_EINVAL;
    hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)ceil_div64(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, x,
                       n / 4, (__bf16*)hi, (__bf16*)lo);
    return woft_launch_status();
}
