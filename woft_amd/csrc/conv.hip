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

// developer tuning knobs (A/B experiments only; set once at start-up, never from the hot path):
//   [0] gather kernel: 0 = register-staged operands, one LDS stage; 1 = B by LDS-DMA, two stages, one barrier per step
//   [1] halo kernel:   0 = weight tile through registers, 1 = weight tile by LDS-DMA (global_load_lds; default)
int g_tuning[4] = {0, 1, 0, 0};

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
    // DEEP = false: register-staged operands, one LDS stage, two barriers per K step.
    // DEEP = true : B (pre-split bf16) is copied global -> LDS by global_load_lds into two stages of unpadded,
    //               source-swizzled 64-byte rows; A (fp32 -> hi/lo) goes through registers into two stages;
    //               one barrier per K step, no staging registers for B.
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int RA = BM / 32;            // float4 rows per thread (A, fp32 source)
    constexpr int RB = BN / 64;            // 16-B rows per thread and plane (B, pre-split bf16; register path)
    constexpr int NP = (TERMS == 3) ? 2 : 1;
    constexpr int LDBB = DEEP ? 32 : LDB;
    constexpr int A_ELEMS = NP * BM * LDB, B_ELEMS = NP * BN * LDBB;
    constexpr int NST = DEEP ? 2 : 1;
    constexpr int SMEM_ELEMS = (NST * (A_ELEMS + B_ELEMS) > 8 * woft::STAGE_FLOATS) ? NST * (A_ELEMS + B_ELEMS)
                                                                                     : 8 * woft::STAGE_FLOATS;
    __shared__ __attribute__((aligned(16))) __bf16 smem[SMEM_ELEMS];
    __bf16* Asm = smem;                                // [NST][NP][BM][LDB]
    __bf16* Bsm = smem + NST * A_ELEMS;                // [NST][NP][BN][LDBB]

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

    f32x4 ra[RA];
    bf16x8 rb[NP][RB];
    auto load_a = [&](int ks) { woft::a_load<RA>(p, arows, ks, nchunk, v, ra); };
    auto store_a = [&](int stage) {
        __bf16* As = Asm + stage * A_ELEMS;
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            const bf16x4 hi = __builtin_convertvector(ra[j], bf16x4);
            *(bf16x4*)(As + (r0 + 32 * j) * LDB + 4 * v) = hi;
            if (NP == 2) {
                const f32x4 rem = ra[j] - __builtin_convertvector(hi, f32x4);
                *(bf16x4*)(As + BM * LDB + (r0 + 32 * j) * LDB + 4 * v) = __builtin_convertvector(rem, bf16x4);
            }
        }
    };
    auto load_b = [&](int ks) {
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
            for (int j = 0; j < RB; ++j)
                rb[pl][j] = *(const bf16x8*)(bsrc[pl] + (int64_t)(n0 + rb0 + 64 * j) * ktot + (int64_t)ks * BK + 8 * vb);
    };
    auto store_b = [&]() {
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
            for (int j = 0; j < RB; ++j) *(bf16x8*)(Bsm + pl * BN * LDB + (rb0 + 64 * j) * LDB + 8 * vb) = rb[pl][j];
    };
    // LDS-DMA of the B tile: one wave instruction = 16 rows x 64 B; lane L -> (row L/4, physical chunk L%4),
    // logical chunk = physical ^ ((row >> 2) & 3)  (swizzle applied on the source side)
    constexpr int DMA_PER_PLANE = BN / 16, DMA_TOTAL = NP * DMA_PER_PLANE;
    auto dma_b = [&](int ks, int stage) {
#pragma unroll
        for (int t = 0; t < (DMA_TOTAL + 3) / 4; ++t) {
            const int q = wave + t * 4;
            if (q < DMA_TOTAL) {
                const int pl = q / DMA_PER_PLANE, cb = q - pl * DMA_PER_PLANE;
                const int row = cb * 16 + (lane >> 2);
                const int c = (lane & 3) ^ ((row >> 2) & 3);
                const __bf16* src = bsrc[pl] + (int64_t)(n0 + row) * ktot + (int64_t)ks * BK + c * 8;
                __bf16* dstl = Bsm + stage * B_ELEMS + pl * BN * LDBB + cb * 16 * LDBB;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)dstl, 16, 0, 0);
            }
        }
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
    const int sw = (r32 >> 2) & 3;
    auto compute = [&](int stage) {
        const __bf16* a_frag = Asm + stage * A_ELEMS + a_off;
        const __bf16* b_rows = Bsm + stage * B_ELEMS + (wn * (BN / 2) + r32) * LDBB;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int bcol = DEEP ? (((s * 2 + hh) ^ sw) * 8) : (hh * 8 + s * 16);
            bf16x8 a[NP][TM], b[NP][TN];
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[pl][i] = *(const bf16x8*)(a_frag + pl * BM * LDB + i * 32 * LDB + s * 16);
#pragma unroll
                for (int j = 0; j < TN; ++j) b[pl][j] = *(const bf16x8*)(b_rows + pl * BN * LDBB + j * 32 * LDBB + bcol);
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
        load_a(0);
        load_b(0);
        store_a(0);
        store_b();
        __syncthreads();
        for (int ks = 0; ks < nk; ++ks) {
            if (ks + 1 < nk) { load_a(ks + 1); load_b(ks + 1); }
            compute(0);
            __syncthreads();
            if (ks + 1 < nk) {
                store_a(0);
                store_b();
                __syncthreads();
            }
        }
    } else {
        load_a(0);
        dma_b(0, 0);
        store_a(0);
        __syncthreads();
        for (int ks = 0; ks < nk; ++ks) {
            const bool nxt = ks + 1 < nk;
            if (nxt) {
                dma_b(ks + 1, (ks + 1) & 1);         // lands in the idle stage while this step computes
                load_a(ks + 1);
            }
            compute(ks & 1);
            if (nxt) store_a((ks + 1) & 1);          // idle A stage: last read in step ks-1, a barrier ago
            __syncthreads();                         // (drains the DMA)
        }
    }
    // every path leaves the loop through a block barrier: the operand stages are dead, reuse them
    woft::conv_epilogue<BM, BN>(p, acc, (float*)smem + wave * woft::STAGE_FLOATS, m0, n0, wm, wn, lane, M, m_tile);
}

// ---- split-bf16 kernel with an LDS-resident input halo -----------------------------------------
// Stride-1 multi-tap convolutions (3x3, 1x5, 5x1): the workgroup's M tile is a TY x TX patch of output
// pixels of one image; for every 32-channel chunk the (TY+kh-1) x (TX+kw-1) input halo is converted
// and written to LDS ONCE and all kh*kw taps read their A fragments from it at shifted rows, so the
// A-side L2->LDS traffic, the fp32->bf16 splitting and the LDS writes drop by the number of taps;
// only the weight tile is re-staged per tap.  TY x TX = 8 x 16 for images, 9 x 9 (= the whole image)
// for the weight head's patches (weighted_raft.py:363-376).
// One workgroup = G tiles ("groups") of TY x TX output pixels (G > 1: consecutive images, i.e. weight-head
// patches) x BN output channels, NWAVES waves laid out WM (rows) x NWAVES/WM (columns).  The weight
// tile staged per tap is shared by all G*TY*TX rows: the more rows, the less weight traffic per MAC.
template <int TY, int TX, int G>
struct HaloRowMap {
    int img0, n_img, y0, x0, ho, wo;
    __device__ __forceinline__ int64_t operator()(int row) const {
        constexpr int NPIX = TY * TX, BMG = (NPIX + 31) / 32 * 32;
        const int g = row / BMG, pl = row - g * BMG;
        if (pl >= NPIX || img0 + g >= n_img) return -1;
        const int y = y0 + pl / TX, x = x0 + pl % TX;
        return (y < ho && x < wo) ? ((int64_t)(img0 + g) * ho + y) * wo + x : -1;
    }
};

template <int TY, int TX, int G, int BN, int TERMS, int NWAVES, int WM, bool DMA>
__global__ __launch_bounds__(NWAVES * 64) void conv_halo_bf16_kernel(const woft_conv_params p) {
    constexpr int NT = NWAVES * 64;
    constexpr int NPIX = TY * TX;
    constexpr int BMG = (NPIX + 31) / 32 * 32;          // rows per group (padded to MFMA tiles)
    constexpr int BM = G * BMG;
    constexpr int WN = NWAVES / WM;
    constexpr int WROWS = BM / WM, WCOLS = BN / WN;
    constexpr int TM = WROWS / 32, TN = WCOLS / 32;
    static_assert(BM % (32 * WM) == 0 && WCOLS % 32 == 0 && TM >= 1 && TN >= 1, "bad wave layout");
    constexpr int NP = (TERMS == 3) ? 2 : 1;
    constexpr int H33 = (TY + 2) * (TX + 2), H15 = TY * (TX + 4), H51 = (TY + 4) * TX;
    constexpr int HROWS = (H33 > H15 ? (H33 > H51 ? H33 : H51) : (H15 > H51 ? H15 : H51));   // per group
    constexpr int HTOT = G * HROWS;
    constexpr int LROWS = NT / 8;                        // halo rows covered per loader pass
    constexpr int RH = (HTOT + LROWS - 1) / LROWS;       // halo float4 rows per thread
    constexpr int BROWS = NT / 4;                        // weight rows covered per loader pass
    constexpr int RB = (BN + BROWS - 1) / BROWS;
    // DMA variant: the weight tile is copied global -> LDS by global_load_lds (no staging registers, no
    // ds_write pass) into one of TWO stages of unpadded 64-byte rows whose four 16-byte chunks are XOR
    // swizzled with (row >> 2) & 3 on the SOURCE side (the DMA image is lane-linear), which keeps the
    // ds_read_b128 fragment reads conflict free; a K step then needs a single block barrier.
    constexpr int LDBB = DMA ? 32 : LDB;                 // B row pitch in elements
    constexpr int A_ELEMS = NP * HTOT * LDB, B_ELEMS = NP * BN * LDBB;
    constexpr int NBST = DMA ? 2 : 1;
    constexpr int STAGE_ELEMS = 2 * NWAVES * woft::STAGE_FLOATS;
    constexpr int SMEM_ELEMS = (A_ELEMS + NBST * B_ELEMS > STAGE_ELEMS) ? A_ELEMS + NBST * B_ELEMS : STAGE_ELEMS;
    __shared__ __attribute__((aligned(16))) __bf16 smem[SMEM_ELEMS];
    __bf16* As = smem;
    __bf16* Bs = smem + A_ELEMS;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r32 = lane & 31, hh = lane >> 5;
    const int v = tid & 7, r0 = tid >> 3;
    const int vb = tid & 3, rb0 = tid >> 2;

    const int tyn = (p.ho + TY - 1) / TY, txn = (p.wo + TX - 1) / TX;
    const int img_groups = (p.n_img + G - 1) / G;
    int m_tile, n_tile;
    woft::tile_of_block(blockIdx.x, img_groups * tyn * txn, p.cout_pad / BN, m_tile, n_tile);
    const int ig = m_tile / (tyn * txn);
    const int trem = m_tile - ig * (tyn * txn);
    const int img0 = ig * G;
    const int y0 = (trem / txn) * TY, x0 = (trem % txn) * TX;
    const int n0 = n_tile * BN;
    const int taps = p.taps_y * p.taps_x;
    const int nchunk = p.cin_pad / BK;
    const int nk = taps * nchunk;
    const int64_t ktot = (int64_t)nk * BK;
    const int HX = TX + p.taps_x - 1;
    const int hrows = (TY + p.taps_y - 1) * HX;          // used halo rows per group (<= HROWS)

    // halo pixels owned by this thread (rows r0 + LROWS j): global pixel index or -1 (zero fill)
    int hpix[RH];                // (n_img * h * w < 2^31: validated by the launcher)
#pragma unroll
    for (int j = 0; j < RH; ++j) {
        const int ht = r0 + LROWS * j;
        const int g = ht / HROWS, h = ht - g * HROWS;
        const int hy = h / HX, hx = h - hy * HX;
        const int iy = y0 + hy - p.pad_y, ix = x0 + hx - p.pad_x;
        const bool ok = ht < HTOT && h < hrows && img0 + g < p.n_img && iy >= 0 && iy < p.h && ix >= 0 && ix < p.w;
        hpix[j] = ok ? ((img0 + g) * p.h + iy) * p.w + ix : -1;
    }
    const __bf16* bsrc[NP];
    bsrc[0] = (const __bf16*)p.wgt_hi;
    if (NP == 2) bsrc[NP - 1] = (const __bf16*)p.wgt_lo;

    f32x4 rh[RH];
    bf16x8 rb[NP][RB];
    auto load_halo = [&](int chunk) {
        const int c0 = chunk * BK;
        const bool second = (p.in1 != nullptr) && (c0 >= p.c_split);
        const float* src = second ? p.in1 : p.in0;
        const int cs = second ? p.cs1 : p.cs0;
        const int cc = (second ? c0 - p.c_split : c0) + 4 * v;
#pragma unroll
        for (int j = 0; j < RH; ++j) {
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (hpix[j] >= 0) val = *(const f32x4*)(src + (int64_t)hpix[j] * cs + cc);
            rh[j] = val;
        }
    };
    auto store_halo = [&]() {
#pragma unroll
        for (int j = 0; j < RH; ++j) {
            const int ht = r0 + LROWS * j;
            if (ht >= HTOT) continue;
            const bf16x4 hi = __builtin_convertvector(rh[j], bf16x4);
            *(bf16x4*)(As + ht * LDB + 4 * v) = hi;
            if (NP == 2) {
                const f32x4 rem = rh[j] - __builtin_convertvector(hi, f32x4);
                *(bf16x4*)(As + HTOT * LDB + ht * LDB + 4 * v) = __builtin_convertvector(rem, bf16x4);
            }
        }
    };
    auto load_b = [&](int ks) {
        const int chunk = ks / taps, tap = ks - chunk * taps;
        const int64_t koff = (int64_t)tap * p.cin_pad + chunk * BK + 8 * vb;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
            for (int j = 0; j < RB; ++j)
                if (rb0 + BROWS * j < BN)
                    rb[pl][j] = *(const bf16x8*)(bsrc[pl] + (int64_t)(n0 + rb0 + BROWS * j) * ktot + koff);
    };
    auto store_b = [&]() {
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
            for (int j = 0; j < RB; ++j)
                if (rb0 + BROWS * j < BN) *(bf16x8*)(Bs + pl * BN * LDB + (rb0 + BROWS * j) * LDB + 8 * vb) = rb[pl][j];
    };

    // DMA: one wave instruction moves 16 rows x 64 B (1 KiB); lane L -> (row L/4, physical chunk L%4)
    constexpr int DMA_PER_PLANE = BN / 16, DMA_TOTAL = NP * DMA_PER_PLANE;
    auto dma_b = [&](int ks, int stage) {
        const int chunk = ks / taps, tap = ks - chunk * taps;
        const int64_t koff = (int64_t)tap * p.cin_pad + chunk * BK;
#pragma unroll
        for (int t = 0; t < (DMA_TOTAL + NWAVES - 1) / NWAVES; ++t) {
            const int q = wave + t * NWAVES;               // wave-uniform instruction index
            if (q < DMA_TOTAL) {
                const int pl = q / DMA_PER_PLANE, cb = q - pl * DMA_PER_PLANE;
                const int row = cb * 16 + (lane >> 2);
                const int c = (lane & 3) ^ ((row >> 2) & 3);
                const __bf16* src = bsrc[pl] + (int64_t)(n0 + row) * ktot + koff + c * 8;
                __bf16* dstl = Bs + stage * B_ELEMS + pl * BN * LDBB + cb * 16 * LDBB;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)dstl, 16, 0, 0);
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // A fragment rows: output pixel (ty, tx) of group g reads halo row g*HROWS + (ty + ky) * HX + (tx + kx)
    int abase[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int ml = wm * WROWS + i * 32 + r32;
        const int g = ml / BMG, pl = ml - g * BMG;
        abase[i] = g * HROWS + ((pl < NPIX) ? (pl / TX) * HX + (pl % TX) : 0);
    }
    const __bf16* b_frag = Bs + (wn * WCOLS + r32) * LDB + hh * 8;

    if (DMA) {
        // physical 16-byte chunk of logical chunk (s*2 + hh) in this lane's B rows (rows = 32*j + r32 + const)
        const int sw = (r32 >> 2) & 3;
        const __bf16* bfr = Bs + (wn * WCOLS + r32) * LDBB;
        load_halo(0);
        dma_b(0, 0);
        store_halo();
        __syncthreads();
        for (int ks = 0; ks < nk; ++ks) {
            const int chunk = ks / taps, tap = ks - chunk * taps;
            const bool nxt = ks + 1 < nk;
            const bool new_chunk = nxt && (tap + 1 == taps);
            if (nxt) dma_b(ks + 1, (ks + 1) & 1);          // lands while this step computes
            if (new_chunk) load_halo(chunk + 1);
            const int ky = tap / p.taps_x, kx = tap - ky * p.taps_x;
            const int toff = (ky * HX + kx) * LDB + hh * 8;
            const __bf16* bst = bfr + (ks & 1) * B_ELEMS;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int pc = ((s * 2 + hh) ^ sw) * 8;
                bf16x8 b[NP][TN];
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                    for (int j = 0; j < TN; ++j) b[pl][j] = *(const bf16x8*)(bst + pl * BN * LDBB + j * 32 * LDBB + pc);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    bf16x8 a[NP];
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl) a[pl] = *(const bf16x8*)(As + pl * HTOT * LDB + abase[i] * LDB + toff + s * 16);
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if (NP == 2) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[NP - 1], b[0][j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[NP - 1][j], acc[i][j], 0, 0, 0);
                        }
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0][j], acc[i][j], 0, 0, 0);
                    }
                }
            }
            __syncthreads();                             // (drains the DMA: the next stage is complete)
            if (new_chunk) {
                store_halo();
                __syncthreads();
            }
        }
        const HaloRowMap<TY, TX, G> rowmap_d{img0, p.n_img, y0, x0, p.ho, p.wo};
        woft::conv_epilogue_t<TM, TN, WROWS, WCOLS>(p, acc, (float*)smem + wave * woft::STAGE_FLOATS, rowmap_d, n0, wm,
                                                    wn, lane, m_tile);
        return;
    }

    load_halo(0);
    load_b(0);
    store_halo();
    store_b();
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        const int chunk = ks / taps, tap = ks - chunk * taps;
        const bool nxt = ks + 1 < nk;
        const bool new_chunk = nxt && (tap + 1 == taps);
        if (nxt) load_b(ks + 1);
        if (new_chunk) load_halo(chunk + 1);
        const int ky = tap / p.taps_x, kx = tap - ky * p.taps_x;
        const int toff = (ky * HX + kx) * LDB + hh * 8;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 b[NP][TN];
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                for (int j = 0; j < TN; ++j) b[pl][j] = *(const bf16x8*)(b_frag + pl * BN * LDB + j * 32 * LDB + s * 16);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                bf16x8 a[NP];            // A fragments are loaded per row tile: keeps tall wave tiles in registers
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) a[pl] = *(const bf16x8*)(As + pl * HTOT * LDB + abase[i] * LDB + toff + s * 16);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (NP == 2) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[NP - 1], b[0][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[NP - 1][j], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0][j], acc[i][j], 0, 0, 0);
                }
            }
        }
        __syncthreads();
        if (nxt) {
            store_b();
            if (new_chunk) store_halo();
            __syncthreads();
        }
    }
    const HaloRowMap<TY, TX, G> rowmap{img0, p.n_img, y0, x0, p.ho, p.wo};
    woft::conv_epilogue_t<TM, TN, WROWS, WCOLS>(p, acc, (float*)smem + wave * woft::STAGE_FLOATS, rowmap, n0, wm, wn,
                                                lane, m_tile);
}

template <int TY, int TX, int G, int BN, int NWAVES, int WM>
int launch_halo(const woft_conv_params& p, hipStream_t s) {
    const int tyn = (p.ho + TY - 1) / TY, txn = (p.wo + TX - 1) / TX;
    const int64_t mt = (int64_t)((p.n_img + G - 1) / G) * tyn * txn;
    dim3 grid((unsigned)(mt * (p.cout_pad / BN)));
    const bool dma = g_tuning[1] != 0;
    if (p.precision == 1 && dma)
        hipLaunchKernelGGL((conv_halo_bf16_kernel<TY, TX, G, BN, 3, NWAVES, WM, true>), grid, dim3(NWAVES * 64), 0, s, p);
    else if (p.precision == 1)
        hipLaunchKernelGGL((conv_halo_bf16_kernel<TY, TX, G, BN, 3, NWAVES, WM, false>), grid, dim3(NWAVES * 64), 0, s, p);
    else if (dma)
        hipLaunchKernelGGL((conv_halo_bf16_kernel<TY, TX, G, BN, 1, NWAVES, WM, true>), grid, dim3(NWAVES * 64), 0, s, p);
    else
        hipLaunchKernelGGL((conv_halo_bf16_kernel<TY, TX, G, BN, 1, NWAVES, WM, false>), grid, dim3(NWAVES * 64), 0, s, p);
    return woft_launch_status();
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
    if (p.halo != 0) {
        // LDS-halo kernels: split-bf16 precisions, stride 1, multi-tap, non-flat, same-size output
        if (p.precision == 0 || p.flat || p.stride != 1 || p.taps_y * p.taps_x < 2) return WOFT_EINVAL;
        if (p.taps_y + p.taps_x > 6 || p.taps_y > 5 || p.taps_x > 5) return WOFT_EINVAL;      // 3x3, 1x5, 5x1 (and smaller)
        if (p.ho != p.h + 2 * p.pad_y - p.taps_y + 1 || p.wo != p.w + 2 * p.pad_x - p.taps_x + 1) return WOFT_EINVAL;
        if ((int64_t)p.n_img * p.h * p.w >= (1ll << 31)) return WOFT_EINVAL;
        if (p.halo == 1 && p.tile_n == 128) return launch_halo<8, 16, 1, 128, 4, 2>(p, s);
        if (p.halo == 1 && p.tile_n == 64) return launch_halo<8, 16, 1, 64, 4, 2>(p, s);
        if (p.halo == 2 && p.tile_n == 128 && p.ho == 9 && p.wo == 9) return launch_halo<9, 9, 1, 128, 4, 1>(p, s);
        if (p.halo == 4 && p.tile_n == 128) return launch_halo<4, 16, 1, 128, 4, 2>(p, s);
        return WOFT_EINVAL;
    }
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
