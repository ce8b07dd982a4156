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
#include <type_traits>

#include "conv_common.h"
#include "dma.h"
#include "halo_map.h"

// Build parts (woft_amd/build.py): this file is compiled once per precision code of woft_conv_params -- -DWOFT_ONLY_PREC=1 (bf16x3:
// the TERMS = 3 instances + the C ABI entry points), 2 (bf16: TERMS = 1 + the exact-fp32 kernel), 3 (fp16: TERMS = 16 + the
// correlation GEMM) -- a single unit took 9+ minutes on one core.  Part k exports woft_conv_dispatch_p<k>.
#if !defined(WOFT_ONLY_PREC)
#error "conv.hip is compiled in parts: -DWOFT_ONLY_PREC=1|2|3 (woft_amd/build.py)"
#endif
#define WOFT_CAT2_(a, b) a##b
#define WOFT_CAT2(a, b) WOFT_CAT2_(a, b)
#define WOFT_CAT4_(a, b, c, d) a##b##c##d
#define WOFT_CAT4(a, b, c, d) WOFT_CAT4_(a, b, c, d)
#if WOFT_ONLY_PREC == 1
constexpr int PART_TERMS = 3;
#elif WOFT_ONLY_PREC == 3
constexpr int PART_TERMS = 16;
#else
constexpr int PART_TERMS = 1;
#endif

// developer knob (woft_set_tuning; never set on the hot path): [2] = ablation bits of corr_gemm_bf16_kernel for
// tools/bench_cgemm.py (1 no stores, 2 no epilogue, 4 no operand DMA after the first step, 8 no MFMAs, 16 non-temporal stores)
#if WOFT_ONLY_PREC == 1
int g_tuning[4] = {0, 0, 0, 0};
#else
extern int g_tuning[4];
#endif

namespace {

using woft::ARows;
using woft::BK;

constexpr int LDS_LD = 36;   // fp32 tiles: floats per row (144 B: 16-B aligned, conflict-free b128 reads)

#if WOFT_ONLY_PREC == 2
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

#endif  // WOFT_ONLY_PREC == 2 (fp32 kernel)

// Per-tap ("gather") implicit GEMM on split-bf16 operands: any stride / tap shape / flat packing, 1x1 included.
// A rows (fp32, NHWC) are fetched one K step ahead into registers, split into bf16 hi / lo and written to the idle
// one of two LDS stages; the weight tile goes global -> LDS by LDS-DMA into the idle one of two stages (unpadded
// 64-byte rows, source-side XOR swizzle, as in conv_halo_bf16_kernel); one barrier per K step.
// Like the halo kernel the loop is written to issue little besides MFMAs: (tap, chunk) advance by counters instead
// of divisions, the per-row pixel offsets are recomputed once per TAP (32-bit), loads are unconditional (clamped
// offset + select), the wave id is scalar so the DMA bookkeeping stays on the SALU.
template <int BM, int BN, int TERMS>
__global__ __launch_bounds__(256) void conv_mfma_bf16_kernel(const woft_conv_params pa, const woft_conv_params pb, const int split) {
    // (workgroups [0, split): layer pa; the rest: layer pb of the same launch -- woft_conv2d_pair; split = gridDim.x otherwise)
    const bool second_layer = (int)blockIdx.x >= split;
    const woft_conv_params p = second_layer ? pb : pa;     // (a copy, not a reference: see conv_regb_kernel)
    const int bid = second_layer ? (int)blockIdx.x - split : (int)blockIdx.x;
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int RA = BM / 32;            // float4 rows per thread (A, fp32 source)
    constexpr int NP = (TERMS == 3) ? 2 : 1;
    constexpr int A_PLANE = BM * LDB, A_STAGE = NP * A_PLANE;
    constexpr int B_PLANE = BN * 32, B_STAGE = NP * B_PLANE;
    constexpr int SMEM_ELEMS = (2 * (A_STAGE + B_STAGE) > 8 * woft::STAGE_FLOATS) ? 2 * (A_STAGE + B_STAGE)
                                                                                   : 8 * woft::STAGE_FLOATS;
    __shared__ __attribute__((aligned(16))) __bf16 smem[SMEM_ELEMS];
    __bf16* Asm = smem;                                // [2][NP][BM][LDB]
    __bf16* Bsm = smem + 2 * A_STAGE;                  // [2][NP][BN][32]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r32 = lane & 31, hh = lane >> 5;
    const int v = tid & 7, r0 = tid >> 3;              // A loader: float4 column, base row

    const int64_t M = (int64_t)p.n_img * p.ho * p.wo;
    int m_tile, n_tile;
    woft::tile_of_block(bid, (int)((M + BM - 1) / BM), p.cout_pad / BN, m_tile, n_tile);
    const int64_t m0 = (int64_t)m_tile * BM;
    const int n0 = n_tile * BN;
    const int nchunk = p.cin_pad / BK;
    const int taps = p.taps_y * p.taps_x;
    const int nk = taps * nchunk;
    const int ktot = nk * BK;                           // (< 2^20: validated by the launcher)

    // rows of this thread: top-left input pixel of the receptive field (32-bit pixel index: validated)
    int iy0[RA], ix0[RA], ibase[RA];
    bool mvalid[RA];
    {
        const int hw = p.ho * p.wo;
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            int64_t m = m0 + r0 + 32 * j;
            mvalid[j] = m < M;
            if (!mvalid[j]) m = 0;
            const int img = (int)(m / hw);
            const int rem = (int)(m - (int64_t)img * hw);
            const int oy = rem / p.wo, ox = rem - oy * p.wo;
            iy0[j] = oy * p.stride - p.pad_y;
            ix0[j] = ox * p.stride - p.pad_x;
            ibase[j] = img * p.h * p.w;
        }
    }
    // prefetch position (tap (ky, kx), chunk) and, per tap, each row's element offset / validity
    int pf_chunk = 0, pf_ky = 0, pf_kx = 0;
    uint32_t poff[RA];                                  // pixel index of the tap (0 where there is none)
    bool pok[RA];
    const int flat_dpix = p.flat ? (4 * v) / p.cs0 : 0;
    auto tap_rows = [&]() {
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            const int iy = iy0[j] + pf_ky, ix = ix0[j] + pf_kx;
            const int ixp = ix + flat_dpix;             // flat: the float4 covers pixel ix + dpix of the row segment
            pok[j] = mvalid[j] && iy >= 0 && iy < p.h && ixp >= 0 && ixp < p.w;
            poff[j] = pok[j] ? (uint32_t)(ibase[j] + iy * p.w + ix) : 0u;
        }
    };
    // TWO register sets: the rows of step k+2 are requested while step k computes and step k+1's set is converted
    // into the idle LDS stage -- a short-K layer (1x1 convs: a dozen steps of 24 MFMAs) otherwise waits for the full
    // HBM latency of its activation rows in every step
    f32x4 ra0[RA], ra1[RA];
    bool rok0[RA], rok1[RA];
    auto load_a = [&](f32x4 (&ra)[RA], bool (&rok)[RA]) {   // the step at the prefetch position
        const int c0 = pf_chunk * BK;
        const bool second = (p.in1 != nullptr) && (c0 >= p.c_split);
        // (the chunk's channel offset goes into the 32-bit lane offset, the scalar base is in0 / in1 itself.  The former form --
        //  `p.in1 + (c0 - p.c_split)` as the base -- was MISCOMPILED inside the K loop: the 64-bit shift of the index took its high
        //  half from an unrelated live SGPR (s_lshl_b64 s[6:7], s[72:73], 2 with s73 never cleared: ISA of round 4), so every
        //  two-source layer on this kernel faulted -- unnoticed for three rounds because the GRU's two-source convs run on the
        //  pixel-tile kernels wherever the map is at least 8 x 16; tests/test_kernels_gpu.py::test_gru_convs_on_the_per_tap_kernel)
        const float* base = second ? p.in1 : p.in0;
        const uint32_t coff = (uint32_t)((second ? c0 - p.c_split : (p.flat ? 0 : c0)) + 4 * v);
        const int cs = second ? p.cs1 : p.cs0;
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            // (flat: poff may address up to dpix pixels left of a valid one; still inside the image row segment)
            ra[j] = *(const f32x4*)(base + (int64_t)(int32_t)(poff[j] * (uint32_t)cs + coff));
            rok[j] = pok[j];
        }
    };
    auto advance = [&]() {                              // next K step: chunk, then tap
        if (++pf_chunk == nchunk) {
            pf_chunk = 0;
            if (++pf_kx == p.taps_x) { pf_kx = 0; ++pf_ky; }
            tap_rows();
        }
    };
    auto store_a = [&](int stage, f32x4 (&ra)[RA], bool (&rok)[RA]) {
        __bf16* As = Asm + stage * A_STAGE;
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            const f32x4 val = rok[j] ? ra[j] : zero;
            const bf16x4 hi = cvt16<TERMS>(val);
            *(bf16x4*)(As + (r0 + 32 * j) * LDB + 4 * v) = hi;
            if (NP == 2) {
                const f32x4 rem = val - widen_bf16x4(hi);
                *(bf16x4*)(As + A_PLANE + (r0 + 32 * j) * LDB + 4 * v) = __builtin_convertvector(rem, bf16x4);
            }
        }
    };

    // weight DMA (see conv_halo_bf16_kernel)
    constexpr int DMA_PER_PLANE = BN / 16, DMA_TOTAL = NP * DMA_PER_PLANE, DMA_PER_WAVE = DMA_TOTAL / 4;
    static_assert(DMA_TOTAL % 4 == 0 || DMA_TOTAL < 4, "weight DMA instructions must divide over the waves");
    constexpr int DMA_LOOPS = DMA_PER_WAVE > 0 ? DMA_PER_WAVE : 1;
    const uint32_t dma_lane = (uint32_t)(((lane >> 2) * ktot + (((lane & 3) ^ ((lane >> 4) & 3)) * 8)) * 2);
    const char* wrow[DMA_LOOPS];
    uint32_t wdst[DMA_LOOPS];
    const uint32_t bs_addr = lds_addr_of(Bsm);
#pragma unroll
    for (int t = 0; t < DMA_LOOPS; ++t) {
        const int q = (wave + t * 4) % DMA_TOTAL;       // (DMA_TOTAL < 4: the spare waves repeat a piece, harmless)
        const int pl = q / DMA_PER_PLANE, cb = q - pl * DMA_PER_PLANE;
        wrow[t] = (const char*)((NP == 2 && pl == 1) ? p.wgt_lo : p.wgt_hi) + ((int64_t)(n0 + cb * 16) * ktot) * 2;
        wdst[t] = bs_addr + (uint32_t)(pl * B_PLANE + cb * 16 * 32) * 2;
    }
    auto dma_b = [&](int ks, int stage) {
#pragma unroll
        for (int t = 0; t < DMA_LOOPS; ++t)
            lds_dma16(wrow[t] + (int64_t)ks * (BK * 2), dma_lane, wdst[t] + (uint32_t)(stage * B_STAGE * 2));
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // lane (r32, hh) feeds k = s*16 + hh*8 + [0,8) of the step for both operands
    const __bf16* a_frag = Asm + (wm * (BM / 2) + r32) * LDB + hh * 8;
    const int sw = (r32 >> 2) & 3;
    int b_frag[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) b_frag[s2] = (wn * (BN / 2) + r32) * 32 + (((s2 * 2 + hh) ^ sw) * 8);
    auto compute = [&](int stage) {
        const __bf16* af = a_frag + stage * A_STAGE;
        const __bf16* bf = Bsm + stage * B_STAGE;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 a[NP][TM], b[NP][TN];
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[pl][i] = *(const bf16x8*)(af + pl * A_PLANE + i * 32 * LDB + s2 * 16);
#pragma unroll
                for (int j = 0; j < TN; ++j) b[pl][j] = *(const bf16x8*)(bf + b_frag[s2] + pl * B_PLANE + j * 32 * 32);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (NP == 2) {
                        acc[i][j] = mma16<TERMS>(a[NP - 1][i], b[0][j], acc[i][j]);
                        acc[i][j] = mma16<TERMS>(a[0][i], b[NP - 1][j], acc[i][j]);
                    }
                    acc[i][j] = mma16<TERMS>(a[0][i], b[0][j], acc[i][j]);
                }
        }
    };

    tap_rows();
    dma_b(0, 0);
    load_a(ra0, rok0);
    store_a(0, ra0, rok0);
    if (nk > 1) {                                        // step 1 -> set 0 (stays in flight across the barrier)
        advance();
        load_a(ra0, rok0);
        dma_wait<RA>();
    } else {
        dma_wait<0>();
    }
    __syncthreads();
    int stage = 0;
    // one K step: `cur` holds the rows of step ks + 1 (requested a step ago), `nxt` receives those of step ks + 2
    auto step = [&](int ks, f32x4 (&cur)[RA], bool (&cok)[RA], f32x4 (&nxt)[RA], bool (&nok)[RA]) {
        dma_b(ks + 1, stage ^ 1);                       // lands in the idle stage while this step computes
        const bool ahead = ks + 2 < nk;
        if (ahead) {
            advance();
            load_a(nxt, nok);
        }
        compute(stage);
        store_a(stage ^ 1, cur, cok);                 // idle A stage: last read in step ks-1, a barrier ago
        if (ahead) dma_wait<RA>();                      // (the RA loads of `nxt` were issued after the DMA)
        else dma_wait<0>();
        __syncthreads();
        stage ^= 1;
    };
    for (int ks = 0; ks + 1 < nk; ks += 2) {
        step(ks, ra0, rok0, ra1, rok1);
        if (ks + 2 < nk) step(ks + 1, ra1, rok1, ra0, rok0);
    }
    compute(stage);
    __syncthreads();
    // the operand stages are dead: reuse them
    woft::conv_epilogue<BM, BN>(p, acc, (float*)smem + wave * woft::STAGE_FLOATS, m0, n0, wm, wn, lane, M, m_tile);
}

// LDS-halo convolution, stride 1, KY x KX taps in {3x3, 1x5, 5x1}, split-bf16 MFMA.
//
// A workgroup (4 waves) owns a TY x TX output patch of one image and BN output channels.  Per 32-channel
// chunk it stages the (TY+KY-1) x (TX+KX-1) input halo ONCE into LDS (fp32 -> bf16 hi / lo planes, 80-byte
// pixel rows) and all KY*KX taps read their A fragments from it at compile-time offsets -- the per-tap
// gather kernel re-loads and re-converts the same pixels for every tap.  The weight tile of each (chunk, tap)
// K step is copied global -> LDS by global_load_lds one step ahead into one of two stages of unpadded 64-byte
// rows whose four 16-byte chunks are XOR-swizzled with (row >> 2) & 3 on the SOURCE side (the DMA image is
// lane-linear), which keeps the ds_read_b128 fragment reads conflict-free.
//
// The loop is built to ISSUE little besides MFMAs (the first version spent ~250 scalar/vector bookkeeping
// instructions per 24 MFMAs -- runtime tap divisions, 64-bit per-lane DMA addresses, exec-masked loads -- and
// was issue bound at a third of the matrix peak): taps are unrolled (A offsets are ds_read immediates), the
// wave id is made scalar so DMA bases and LDS destinations live in SGPRs, every lane offset is one 32-bit VGPR
// computed once, halo loads are unconditional (clamped address + select).
// STAGES = 2: weight tile of step k+1 lands while step k computes (one barrier per step).
// STAGES = 1: one weight stage, two barriers per step, registers capped for 4 workgroups per CU (so that the ~1000
//             workgroups of a 4x16-tiled layer at 1/8 of 1080p fit on the chip in one round).  Measured SLOWER than
//             two stages at three workgroups per CU (GRU q conv 72 vs 64 us) and not instantiated.
// NORM (compile time, so that the other layers do not pay for it): p.in_norm != 0, the producer's InstanceNorm (+ ReLU)
// is applied while the halo is staged.
// C0 (9 x 9 windows, 3 x 3 taps, 128 input channels): the INPUT of this layer is not read from memory but computed
// here, 32 channels at a time, as relu(conv 3x3 (5 -> 128) + bias) of the lookup window of one source pixel (the
// weight head's first layer, weighted_raft.py:336,363-376) -- see conv0_* below.
template <int TY, int TX, int KY, int KX, int BN, int TERMS, int WM, int STAGES, bool NORM, bool C0 = false>
__global__ __launch_bounds__(256, (STAGES == 1 ? 4 : (TY == 9 ? 3 : 1))) void conv_halo_bf16_kernel(const woft_conv_params p) {
    static_assert(!C0 || (TY == 9 && TX == 9 && KY == 3 && KX == 3 && !NORM && STAGES == 2), "C0: weight-head windows");
    constexpr int NWAVES = 4;
    constexpr int NPIX = TY * TX;
    constexpr int BM = (NPIX + 31) / 32 * 32;           // rows (padded to MFMA tiles)
    constexpr int WN = NWAVES / WM;
    constexpr int WROWS = BM / WM, WCOLS = BN / WN;
    constexpr int TM = WROWS / 32, TN = WCOLS / 32;
    static_assert(BM % (32 * WM) == 0 && WCOLS % 32 == 0 && TM >= 1 && TN >= 1, "bad wave layout");
    constexpr int NP = (TERMS == 3) ? 2 : 1;
    constexpr int TAPS = KY * KX;
    constexpr int HX = TX + KX - 1, HY = TY + KY - 1, HROWS = HX * HY;
    constexpr int RH = (HROWS + 31) / 32;               // halo float4 rows per thread (32 rows per pass)
    constexpr int A_PLANE = HROWS * LDB, A_ELEMS = NP * A_PLANE;
    constexpr int B_PLANE = BN * 32, B_STAGE = NP * B_PLANE;
    constexpr int STAGE_ELEMS = 2 * NWAVES * woft::STAGE_FLOATS;
    constexpr int SMEM_ELEMS = (A_ELEMS + STAGES * B_STAGE > STAGE_ELEMS) ? A_ELEMS + STAGES * B_STAGE : STAGE_ELEMS;
    __shared__ __attribute__((aligned(16))) __bf16 smem[SMEM_ELEMS];
    __bf16* As = smem;
    __bf16* Bs = smem + A_ELEMS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: DMA bookkeeping stays on the SALU
    const int wm = wave / WN, wn = wave % WN;
    const int r32 = lane & 31, hh = lane >> 5;
    const int v = tid & 7, r0 = tid >> 3;

    const int tyn = (p.ho + TY - 1) / TY, txn = (p.wo + TX - 1) / TX;
    int m_tile, n_tile;
    woft::tile_of_block(blockIdx.x, p.n_img * tyn * txn, p.cout_pad / BN, m_tile, n_tile);
    const int img0 = m_tile / (tyn * txn);
    if (TY == 9) {      // weight-head windows: a NEGATIVE entry of the window list = window not wanted in this launch (the
                        // list is rewritten on the device per frame: woft_wh_needed) -- the whole workgroup leaves
        if (p.wh0_index != nullptr && p.wh0_index[img0] < 0) return;
        if (p.out_index != nullptr && p.out_index[img0] < 0) return;
    }
    const int trem = m_tile - img0 * (tyn * txn);
    const int y0 = (trem / txn) * TY, x0 = (trem % txn) * TX;
    const int n0 = n_tile * BN;
    const int nchunk = p.cin_pad / BK;
    const int ktot = TAPS * p.cin_pad;                  // (< 2^20: validated by the launcher)

    // halo pixels owned by this thread (rows r0 + 32 j): element offset of the pixel's channel vector / cs,
    // clamped to pixel 0 where there is no pixel (then the loaded value is replaced by zero)
    int hpix[RH];
    bool hok[RH];
#pragma unroll
    for (int j = 0; j < RH; ++j) {
        const int ht = r0 + 32 * j;
        const int hy = ht / HX, hx = ht - hy * HX;
        const int iy = y0 + hy - p.pad_y, ix = x0 + hx - p.pad_x;
        hok[j] = ht < HROWS && iy >= 0 && iy < p.h && ix >= 0 && ix < p.w;
        hpix[j] = hok[j] ? (img0 * p.h + iy) * p.w + ix : 0;
    }
    f32x4 rh[RH], nmu, nrs;
    constexpr int NLOAD = C0 ? 3 * NP : RH + (NORM ? 2 : 0);   // vector loads per load_halo (counted by vmcnt below)
    // ---- C0: first layer of the weight head, fused ------------------------------------------------------------
    // a1^T[ch][pix] = sum_k W0[ch][k] x[k][pix], k = tap * 5 + ci (45 -> 48 = 3 MFMA K steps), as a TRANSPOSED
    // product (A = weights, B = im2col of the window), so that an accumulator lane holds 4 x 4 consecutive channels
    // of ONE pixel = 8-byte stores into the halo rows.  Wave i < 3 owns pixels 32 i .. 32 i + 31 (natural order);
    // its im2col fragments are built once from the window (LDS), the 32 x 48 weight slice of the next chunk is
    // prefetched during the first tap of a chunk (NLOAD loads by EVERY wave: the counted vmcnt waits are per wave).
    bf16x8 c0x[3][NP], c0w[3][NP];
    f32x16 c0acc;
    const int c0pix = wave * 32 + r32;                   // (wave 3: no pixel)
    const bool c0ok = C0 && wave < 3 && c0pix < 81;
    auto load_halo = [&](int chunk) {                    // NLOAD unconditional 16-byte loads
        if constexpr (C0) {
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
                    c0w[s3][pl] = *(const bf16x8*)((const __bf16*)p.wh0_w + (((chunk * 3 + s3) * NP + pl) * 64 + lane) * 8);
            return;
        }
        const int c0 = chunk * BK;
        const bool second = (p.in1 != nullptr) && (c0 >= p.c_split);
        const float* src = (second ? p.in1 + (c0 - p.c_split) : p.in0 + c0) + 4 * v;
        const int cs = second ? p.cs1 : p.cs0;
#pragma unroll
        for (int j = 0; j < RH; ++j) rh[j] = *(const f32x4*)(src + (uint32_t)(hpix[j] * cs));
        if (NORM) {
            nmu = *(const f32x4*)(p.in_mean + c0 + 4 * v);
            nrs = *(const f32x4*)(p.in_rstd + c0 + 4 * v);
        }
    };
    auto conv0_compute = [&]() {                         // c0acc = W0[chunk] x im2col  (waves 0-2)
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) asm volatile("" : "+v"(c0w[s3][pl]));
        if (wave < 3) {
#pragma unroll
            for (int r = 0; r < 16; ++r) c0acc[r] = 0.f;
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3) {
                if (NP == 2) {
                    c0acc = mma16<TERMS>(c0w[s3][NP - 1], c0x[s3][0], c0acc);
                    c0acc = mma16<TERMS>(c0w[s3][0], c0x[s3][NP - 1], c0acc);
                }
                c0acc = mma16<TERMS>(c0w[s3][0], c0x[s3][0], c0acc);
            }
        }
    };
    auto conv0_store = [&](int chunk) {                  // relu(c0acc + bias) -> bf16 hi / lo rows of the halo
        if (c0ok) {
            const int hrow = (c0pix / 9 + 1) * HX + c0pix % 9 + 1;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = 8 * g + 4 * hh;           // accumulator rows (r & 3) + 8 (r >> 2) + 4 hh = channels
                const f32x4 b4 = *(const f32x4*)(p.wh0_bias + chunk * BK + ch);
                f32x4 val;
#pragma unroll
                for (int e = 0; e < 4; ++e) val[e] = fmaxf(c0acc[4 * g + e] + b4[e], 0.f);
                const bf16x4 hi = __builtin_convertvector(val, bf16x4);
                *(bf16x4*)(As + hrow * LDB + ch) = hi;
                if (NP == 2) {
                    const f32x4 rem = val - widen_bf16x4(hi);
                    *(bf16x4*)(As + A_PLANE + hrow * LDB + ch) = __builtin_convertvector(rem, bf16x4);
                }
            }
        }
    };
    auto store_halo = [&]() {
        // (pins the use of the prefetched registers HERE: the conversions must not be scheduled up into the taps,
        //  where their wait would drain the weight DMA queue early)
#pragma unroll
        for (int j = 0; j < RH; ++j) asm volatile("" : "+v"(rh[j]));
        if (NORM) asm volatile("" : "+v"(nmu), "+v"(nrs));
#pragma unroll
        for (int j = 0; j < RH; ++j) {
            const int ht = r0 + 32 * j;
            if (RH * 32 > HROWS && ht >= HROWS) continue;
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            f32x4 x = rh[j];
            if (NORM) {                                  // InstanceNorm (+ ReLU) of the producer, applied on load
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float y = (x[e] - nmu[e]) * nrs[e];
                    if (p.in_norm == 2) y = fmaxf(y, 0.f);
                    x[e] = y;
                }
            }
            const f32x4 val = hok[j] ? x : zero;
            const bf16x4 hi = cvt16<TERMS>(val);
            *(bf16x4*)(As + ht * LDB + 4 * v) = hi;
            if (NP == 2) {
                const f32x4 rem = val - widen_bf16x4(hi);
                *(bf16x4*)(As + A_PLANE + ht * LDB + 4 * v) = __builtin_convertvector(rem, bf16x4);
            }
        }
    };

    // weight DMA: one wave instruction moves 16 rows x 64 B; lane L -> (row L/4, physical chunk L%4) which holds
    // logical chunk (L%4) ^ ((row >> 2) & 3) = (L%4) ^ ((L >> 4) & 3) for every 16-row group
    constexpr int DMA_PER_PLANE = BN / 16, DMA_TOTAL = NP * DMA_PER_PLANE, DMA_PER_WAVE = DMA_TOTAL / NWAVES;
    static_assert(DMA_TOTAL % NWAVES == 0, "weight DMA instructions must divide over the waves");
    const uint32_t dma_lane = (uint32_t)(((lane >> 2) * ktot + (((lane & 3) ^ ((lane >> 4) & 3)) * 8)) * 2);   // bytes
    const char* wrow[DMA_PER_WAVE];                      // scalar: plane base + first row of the 16-row group
    uint32_t wdst[DMA_PER_WAVE];                         // scalar: LDS byte address inside stage 0
    const uint32_t bs_addr = lds_addr_of(Bs);
#pragma unroll
    for (int t = 0; t < DMA_PER_WAVE; ++t) {
        const int q = wave + t * NWAVES;
        const int pl = q / DMA_PER_PLANE, cb = q - pl * DMA_PER_PLANE;
        wrow[t] = (const char*)((NP == 2 && pl == 1) ? p.wgt_lo : p.wgt_hi) + ((int64_t)(n0 + cb * 16) * ktot) * 2;
        wdst[t] = bs_addr + (uint32_t)(pl * B_PLANE + cb * 16 * 32) * 2;
    }
    auto dma_b = [&](int koff, int stage) {              // koff: first k of the step (elements), scalar
#pragma unroll
        for (int t = 0; t < DMA_PER_WAVE; ++t)
            lds_dma16(wrow[t] + (int64_t)koff * 2, dma_lane, wdst[t] + (uint32_t)(stage * B_STAGE * 2));
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addresses: one VGPR each, everything else is an immediate
    const __bf16* a_frag[TM];        // output pixel (ty, tx) reads halo row (ty + ky) * HX + (tx + kx)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        bool valid;
        const int pl = halo_row_pixel<TY, TX>(wm * WROWS + i * 32 + r32, valid);
        a_frag[i] = As + ((pl / TX) * HX + (pl % TX)) * LDB + hh * 8;
    }
    const int sw = (r32 >> 2) & 3;
    int b_frag[2];                   // element offset of logical chunk (s*2 + hh) in this lane's B rows
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) b_frag[s2] = (wn * WCOLS + r32) * 32 + (((s2 * 2 + hh) ^ sw) * 8);

    if constexpr (C0) {
        // window of this workgroup's source pixel -> LDS (inside weight stage 1, which the first DMA of the loop
        // fills only after the barrier below); halo buffer zeroed once: its border rows are the conv's zero padding
        float* win = (float*)(Bs + B_STAGE);
        const int src = p.wh0_index ? p.wh0_index[img0] : img0;
        const float* lk = p.wh0_lookup + (int64_t)src * p.wh0_ld;
        const float mv = p.wh0_mean[src];
        if (tid < 81) *(f32x4*)(win + 4 * tid) = *(const f32x4*)(lk + 4 * tid);
        for (int i = tid; i < A_ELEMS / 8; i += 256) *(f32x4*)(As + 8 * i) = f32x4{0.f, 0.f, 0.f, 0.f};
        load_halo(0);
        dma_b(0, 0);
        __syncthreads();
        const int py = c0pix / 9, px = c0pix - py * 9;
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) {
            float xv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {                // k = 16 s3 + 8 hh + e: both candidates are constants
                const int k0 = 16 * s3 + e, k1 = k0 + 8;
                const int t0 = k0 / 5, t1 = k1 / 5;
                const int dy = hh ? t1 / 3 - 1 : t0 / 3 - 1, dx = hh ? t1 % 3 - 1 : t0 % 3 - 1;
                const int ci = hh ? k1 % 5 : k0 % 5;
                const bool kok = hh ? (k1 < 45) : (k0 < 45);
                const int yy = py + dy, xx = px + dx;
                const bool ok = c0ok && kok && yy >= 0 && yy < 9 && xx >= 0 && xx < 9;
                const float wv = win[ok ? (yy * 9 + xx) * 4 + (ci & 3) : 0];
                xv[e] = ok ? (ci == 4 ? mv : wv) : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const __bf16 hi = (__bf16)xv[e];
                c0x[s3][0][e] = hi;
                if (NP == 2) c0x[s3][NP - 1][e] = (__bf16)(xv[e] - (float)hi);
            }
        }
        conv0_compute();
        conv0_store(0);
    } else {
        load_halo(0);
        if (STAGES == 2) dma_b(0, 0);
        store_halo();
    }
    dma_wait<0>();                                       // this wave's DMA has landed before the others read it
    __syncthreads();
    int stage = 0;
    // one 32-channel chunk = TAPS unrolled K steps; MORE (compile time: the last chunk is peeled, so the compiler
    // can pair each prefetch with its store) = another chunk follows
    auto run_chunk = [&](int chunk, auto more_tag) {
        constexpr bool more = decltype(more_tag)::value;
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            // next step's weights (and, during the first tap, the next chunk's halo: RH loads issued AFTER the DMA,
            // so that "at most RH loads in flight" = DMA complete while the halo loads cross the barrier)
            if (C0 && more && tap == TAPS - 1) conv0_compute();   // (its weights were requested eight taps ago)
            if (STAGES == 2) {
                if (tap + 1 < TAPS) dma_b((tap + 1) * p.cin_pad + chunk * BK, stage ^ 1);
                else if (more) dma_b((chunk + 1) * BK, stage ^ 1);
            } else {
                dma_b(tap * p.cin_pad + chunk * BK, 0);      // this step's weights; the stage is free (barrier below)
            }
            if (tap == 0 && more) load_halo(chunk + 1);
            if (STAGES == 1) {
                if (tap == 0 && more) dma_wait<NLOAD>();
                else dma_wait<0>();
                __syncthreads();
            }
            const int ky = tap / KX, kx = tap - ky * KX;
            const __bf16* bst = Bs + stage * B_STAGE;
            // software pipeline over the 2 * TM (k half, row tile) sub-steps: the fragments of sub-step n + 1 are
            // requested BEFORE the MFMAs of sub-step n, so that their LDS latency hides behind this wave's own MFMAs
            constexpr int NS = 2 * TM;
            bf16x8 bq[2][NP][TN], aq[2][NP];
            auto load_b = [&](int s2) {
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        bq[s2 & 1][pl][j] = *(const bf16x8*)(bst + b_frag[s2] + pl * B_PLANE + j * 32 * 32);
            };
            auto load_a = [&](int n) {
                const int s2 = n / TM, i = n % TM;
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
                    aq[n & 1][pl] = *(const bf16x8*)(a_frag[i] + pl * A_PLANE + (ky * HX + kx) * LDB + s2 * 16);
            };
            load_b(0);
            load_a(0);
#pragma unroll
            for (int n = 0; n < NS; ++n) {
                const int s2 = n / TM, i = n % TM;
                if (n + 1 < NS) {
                    if ((n + 1) % TM == 0) load_b(s2 + 1);
                    load_a(n + 1);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (NP == 2) {
                        acc[i][j] = mma16<TERMS>(aq[n & 1][NP - 1], bq[s2 & 1][0][j], acc[i][j]);
                        acc[i][j] = mma16<TERMS>(aq[n & 1][0], bq[s2 & 1][NP - 1][j], acc[i][j]);
                    }
                    acc[i][j] = mma16<TERMS>(aq[n & 1][0], bq[s2 & 1][0][j], acc[i][j]);
                }
            }
            if (STAGES == 2) {
                if (tap == 0 && more) dma_wait<NLOAD>();
                else dma_wait<0>();
            }
            __syncthreads();
            if (STAGES == 2) stage ^= 1;
        }
        if (more) {
            if constexpr (C0) conv0_store(chunk + 1);
            else store_halo();
            __syncthreads();
        }
    };
    for (int chunk = 0; chunk + 1 < nchunk; ++chunk) run_chunk(chunk, std::true_type{});
    run_chunk(nchunk - 1, std::false_type{});
    if constexpr (TY == 9 && TX == 9) {
        if (p.epi == WOFT_EPI_WH_MEAN) {
            // weight-head tail fused (weighted_raft.py:341,378-383): this workgroup holds relu(conv) of one whole
            // patch, all channels: out[image] = e1[0] + mean_pixels <e0, relu(y)> -- the 1.3 GB activation is never
            // written.  Accumulator layout: column = r32, row = (r & 3) + 8 (r >> 2) + 4 hh of tile i.
            constexpr Perm9Mask vm = make_perm9_mask();
            float part = 0.f;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * WCOLS + j * 32 + r32;
                const bool nok = n < p.cout;
                const float wv = nok ? p.e0[n] : 0.f;
                const float bv = (nok && p.bias != nullptr) ? p.bias[n] : 0.f;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
                        const bool valid = (vm.m[wm * TM + i] >> row) & 1u;
                        const float y = fmaxf(p.alpha * acc[i][j][r] + bv, 0.f);
                        part += valid ? wv * y : 0.f;
                    }
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
            float* red = (float*)smem;                   // (operand stages are dead after the loop's last barrier)
            if (lane == 0) red[wave] = part;
            __syncthreads();
            if (tid == 0)
                p.out[p.out_index ? p.out_index[img0] : img0] = p.e1[0] + (red[0] + red[1] + red[2] + red[3]) * (1.f / 81.f);
            return;
        }
    }
    const HaloRowMap<TY, TX> rowmap{img0, p.n_img, y0, x0, p.ho, p.wo};
    woft::conv_epilogue_t<TM, TN, WROWS, WCOLS>(p, acc, (float*)smem + wave * woft::STAGE_FLOATS, rowmap, n0, wm, wn,
                                                lane, m_tile);
}

template <int TY, int TX, int BN, int WM, int STAGES>
int launch_halo(const woft_conv_params& p, hipStream_t s) {
    const int tyn = (p.ho + TY - 1) / TY, txn = (p.wo + TX - 1) / TX;
    const int64_t mt = (int64_t)p.n_img * tyn * txn;
    dim3 grid((unsigned)(mt * (p.cout_pad / BN)));
#define HALO_LAUNCH(KY, KX, T, N) \
    woft_launch(0, conv_halo_bf16_kernel<TY, TX, KY, KX, BN, T, WM, STAGES, N>, grid, dim3(256), 0, s, p)
#define HALO_TAPS(T)                                                                            \
    if (p.wh0_lookup != nullptr) {        /* weight head, first two layers in one launch */      \
        if constexpr (TY == 9 && STAGES == 2)                                                    \
            hipLaunchKernelGGL((conv_halo_bf16_kernel<TY, TX, 3, 3, BN, T, WM, STAGES, false, true>), grid, dim3(256), 0, s, p); \
        else return WOFT_EINVAL;                                                                \
    } else if (p.in_norm != 0) {          /* encoder residual blocks: 3x3 only */                \
        if (p.taps_y == 3 && p.taps_x == 3 && TY != 9) HALO_LAUNCH(3, 3, T, (TY != 9));         \
        else return WOFT_EINVAL;                                                                \
    } else if (p.taps_y == 3 && p.taps_x == 3) HALO_LAUNCH(3, 3, T, false);                     \
    else if (p.taps_y == 1 && p.taps_x == 5) HALO_LAUNCH(1, 5, T, false);                       \
    else if (p.taps_y == 5 && p.taps_x == 1) HALO_LAUNCH(5, 1, T, false);                       \
    else return WOFT_EINVAL
    if (p.precision != WOFT_ONLY_PREC) return WOFT_EINVAL;
#if WOFT_ONLY_PREC == 3
    if constexpr (TY == 9) return WOFT_EINVAL;           /* (the weight head keeps the split-bf16 arithmetic) */
    else { HALO_TAPS(16); }
#else
    HALO_TAPS(PART_TERMS);
#endif
#undef HALO_TAPS
#undef HALO_LAUNCH
    return woft_launch_status();
}

// ---- all-pairs correlation GEMM on pre-split operands -------------------------------------------
// vol[p][q] = alpha * <f1[p], f2[q]> (corr.py:62-69) with BOTH feature maps already converted to bf16 once
// (the per-tap kernel above re-splits the fp32 A tile in each of the ~255 column-tile workgroups that share
// it).  Operand rows are sequences of 128-byte LINES, one line per K step:
//   TERMS 3: line = [hi of 32 k | lo of 32 k]   (woft_split_bf16_lines)      K step = 32
//   TERMS 1: line = 64 k of the bf16 plane       (woft_split_bf16, hi only)   K step = 64
// so that every global_load_lds wave instruction moves 8 rows x one full 128-B line (half-line pieces cost
// the texture-addresser twice the cycles).  LDS image of a tile: [128 rows][8 chunks of 16 B], chunk c of row r
// stored at c ^ ((r >> 1) & 7) -- applied on the global side, the DMA destination is lane-linear -- which
// makes the ds_read_b128 fragment reads conflict-free (rows of equal parity alias mod 256 B; each 16-lane
// service group holds 8 even and 8 odd rows whose (r >> 1) & 7 are all different).
// One stage only (32 KiB): four workgroups per CU overlap each other's load, MFMA and store-drain phases,
// which measured faster than two stages with two workgroups (tools/bench_cgemm.py).
#if WOFT_ONLY_PREC == 3
template <int TERMS, int WM, int WN, int STAGES, bool OUT16 = false>
__global__ __launch_bounds__(WM * WN * 64, 4)
void corr_gemm_bf16_kernel(const __bf16* __restrict__ a, const __bf16* __restrict__ b, int line_elems_per_row,
                           const woft_conv_params p, int abl) {
    // workgroup = WM x WN waves, each owning a 64 x 64 block of the (64 WM) x (64 WN) tile
    constexpr int NWV = WM * WN;
    constexpr int BM = WM * 64, BN = WN * 64, TM = 2, TN = 2;
    constexpr int STG = (BM + BN) * 64;                 // elements of one stage: BM + BN rows of 128 B
    constexpr int SMEM_ELEMS = (STAGES * STG > 2 * NWV * woft::STAGE_FLOATS) ? STAGES * STG : 2 * NWV * woft::STAGE_FLOATS;
    __shared__ __attribute__((aligned(16))) __bf16 smem[SMEM_ELEMS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int r32 = lane & 31, hh = lane >> 5;
    const int64_t M = (int64_t)p.n_img * p.ho * p.wo;
    int m_tile, n_tile;
    woft::tile_of_block(blockIdx.x, (int)((M + BM - 1) / BM), p.cout_pad / BN, m_tile, n_tile);
    const int64_t m0 = (int64_t)m_tile * BM;
    const int n0 = n_tile * BN;
    const int ld = line_elems_per_row;                  // elements per operand row (all its lines)
    const int nk = ld / 64;

    // DMA piece q = wave + NWV t of a step: rows 8 q .. 8 q + 7 of the stage image [A rows | B rows], lane ->
    // (row lane / 8, physical chunk lane % 8) holding logical chunk (lane % 8) ^ ((row >> 1) & 7);
    // (row >> 1) & 7 = (4 (wave & 1) + lane / 16) & 7 for every t (NWV is even), so ONE lane-offset register serves
    // all pieces of the wave and the bases are scalar
    constexpr int NQ = (BM + BN) / 8, QPW = NQ / NWV;
    static_assert(NQ % NWV == 0 && NWV % 2 == 0, "DMA pieces must divide over an even number of waves");
    const uint32_t dma_lane =
        (uint32_t)(((lane >> 3) * ld + (((lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7)) * 8)) * 2);
    const char* srow[QPW];
    uint32_t sdst[QPW];
    const uint32_t smem_addr = lds_addr_of(smem);
#pragma unroll
    for (int t = 0; t < QPW; ++t) {
        const int q = wave + t * NWV;
        srow[t] = (const char*)(q < BM / 8 ? a + (m0 + q * 8) * ld : b + (int64_t)(n0 + (q - BM / 8) * 8) * ld);
        sdst[t] = smem_addr + (uint32_t)q * 1024u;
    }
    auto dma = [&](int ks, int stage) {
#pragma unroll
        for (int t = 0; t < QPW; ++t) lds_dma16(srow[t] + (int64_t)ks * 128, dma_lane, sdst[t] + (uint32_t)(stage * STG * 2));
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int sw = (r32 >> 1) & 7;
    const __bf16* a_rows0 = smem + (wm * 64 + r32) * 64;
    const __bf16* b_rows0 = smem + BM * 64 + (wn * 64 + r32) * 64;
    if (STAGES == 2) {
        dma(0, 0);
        dma_wait<0>();
        __syncthreads();
    }
    for (int ks = 0; ks < nk; ++ks) {
        if (STAGES == 2) {
            if (ks + 1 < nk && !(abl & 4)) dma(ks + 1, (ks + 1) & 1);     // lands while this step computes
        } else {
            if (!(abl & 4) || ks == 0) dma(ks, 0);
            dma_wait<0>();                               // this wave's pieces have landed before the others read
            __syncthreads();
        }
        const __bf16* a_rows = a_rows0 + (STAGES == 2 ? (ks & 1) * STG : 0);
        const __bf16* b_rows = b_rows0 + (STAGES == 2 ? (ks & 1) * STG : 0);
        if (!(abl & 8)) {
            if (TERMS == 3) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int ch = ((s * 2 + hh) ^ sw) * 8, cl = ((4 + s * 2 + hh) ^ sw) * 8;
                    bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        ah[i] = *(const bf16x8*)(a_rows + i * 32 * 64 + ch);
                        al[i] = *(const bf16x8*)(a_rows + i * 32 * 64 + cl);
                    }
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        bh[j] = *(const bf16x8*)(b_rows + j * 32 * 64 + ch);
                        bl[j] = *(const bf16x8*)(b_rows + j * 32 * 64 + cl);
                    }
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {      // small terms first, as in conv_mfma_bf16_kernel
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                        }
                }
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int ch = ((s * 2 + hh) ^ sw) * 8;
                    bf16x8 ah[TM], bh[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) ah[i] = *(const bf16x8*)(a_rows + i * 32 * 64 + ch);
#pragma unroll
                    for (int j = 0; j < TN; ++j) bh[j] = *(const bf16x8*)(b_rows + j * 32 * 64 + ch);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
            }
        }
        if (STAGES == 2) dma_wait<0>();
        __syncthreads();
    }
    if (abl & 2) {          // ablation: no epilogue at all (keep the accumulators live)
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) t += acc[i][j][r];
        if (t == 12345.678f) p.out[0] = t;
        return;
    }
    // epilogue: alpha * acc, transposed 32x32 at a time through this wave's LDS stage into 16-byte row stores
    // (8 rows x 128 B per wave instruction).  cout % 4 == 0 and ldo % 4 == 0 (validated by the launcher).
    float* stage = (float*)smem + wave * woft::STAGE_FLOATS;
    constexpr int LD = woft::STAGE_LD;
    const int rr = lane >> 3, c4 = (lane & 7) * 4;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * hh) * LD + r32] = acc[i][j][r];
            __builtin_amdgcn_wave_barrier();
            const int n = n0 + wn * 64 + j * 32 + c4;
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int row = rr + 8 * pass;
                const int64_t m = m0 + wm * 64 + i * 32 + row;
                f32x4 v = *(const f32x4*)(stage + row * LD + c4);
                if (m >= M || n >= p.cout || (abl & 1)) continue;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= p.alpha;
                if (OUT16) *(bf16x4*)((__bf16*)p.out + m * p.ldo + n) = __builtin_convertvector(v, bf16x4);   // 8 lanes x 8 B per row
                else if (abl & 16) __builtin_nontemporal_store(v, (f32x4*)(p.out + m * p.ldo + n));
                else *(f32x4*)(p.out + m * p.ldo + n) = v;
            }
            __builtin_amdgcn_wave_barrier();
        }
}

#endif  // WOFT_ONLY_PREC == 3 (correlation GEMM kernel)

#if WOFT_ONLY_PREC == 1
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

#endif

template <int BM, int BN>
int launch_conv(const woft_conv_params& p, const woft_conv_params* second, hipStream_t s) {
    auto blocks = [](const woft_conv_params& q) { return ceil_div64((int64_t)q.n_img * q.ho * q.wo, BM) * (q.cout_pad / BN); };
    const woft_conv_params& pb = second ? *second : p;
    const int split = (int)blocks(p);
    dim3 grid((unsigned)(blocks(p) + (second ? blocks(pb) : 0)));       // 1-D: see woft::tile_of_block
    if (p.precision == 0) {
#if WOFT_ONLY_PREC == 2
        if (second) return WOFT_EINVAL;
        woft_launch(0, conv_mfma_f32_kernel<BM, BN>, grid, dim3(256), 0, s, p);
        return woft_launch_status();
#else
        return WOFT_EINVAL;
#endif
    }
    if (p.precision != WOFT_ONLY_PREC) return WOFT_EINVAL;
    // split-bf16 kernels use 32-bit element offsets
    for (const woft_conv_params* q : {&p, &pb}) {
        const int64_t cs_max = (q->in1 != nullptr && q->cs1 > q->cs0) ? q->cs1 : q->cs0;
        if ((int64_t)q->n_img * q->h * q->w * cs_max >= (1ll << 31)) return WOFT_EINVAL;
        if ((int64_t)q->taps_y * q->taps_x * q->cin_pad >= (1 << 20)) return WOFT_EINVAL;
    }
    woft_launch(0, conv_mfma_bf16_kernel<BM, BN, PART_TERMS>, grid, dim3(256), 0, s, p, pb, split);
    return woft_launch_status();
}

}  // namespace

// conv_regb.hip, the part of this precision
int WOFT_CAT2(woft_conv_regb_launch_p, WOFT_ONLY_PREC)(const woft_conv_params& p, const woft_conv_params* second, void* stream);
int woft_conv_stem_launch(const woft_conv_params& p, void* stream);                                     // conv_stem.hip
int woft_conv_1x1_launch(const woft_conv_params& p, const woft_conv_params* second, void* stream);      // conv_1x1.hip
int woft_conv_dispatch_p1(const woft_conv_params& p, const woft_conv_params* second, void* stream);
int woft_conv_dispatch_p2(const woft_conv_params& p, const woft_conv_params* second, void* stream);
int woft_conv_dispatch_p3(const woft_conv_params& p, const woft_conv_params* second, void* stream);

#if WOFT_ONLY_PREC == 1
static int conv_check(const woft_conv_params& p) {
    if (p.in0 == nullptr || p.out == nullptr) return WOFT_EINVAL;
    if (p.precision < 0 || p.precision > 4) return WOFT_EINVAL;
    if (p.precision == 4 && ((p.halo != 8 && p.halo != 12) || p.wgt_frag == nullptr || p.wgt_mx == nullptr ||
                             p.in_norm != 0 || p.taps_y * p.taps_x == 1))
        return WOFT_EINVAL;                   // f16mx8: the register-streamed kernel's multi-tap instances only (woft_conv_params.wgt_mx)
    if (p.precision == 0 && p.wgt == nullptr) return WOFT_EINVAL;
    if (p.precision >= 1 && p.wgt_hi == nullptr) return WOFT_EINVAL;
    if (p.precision == 1 && p.wgt_lo == nullptr) return WOFT_EINVAL;
    if (p.cin_pad <= 0 || p.cin_pad % BK != 0) return WOFT_EINVAL;
    if (p.in1 != nullptr && (p.c_split % BK != 0 || p.c_split <= 0 || p.c_split >= p.cin_pad)) return WOFT_EINVAL;
    if (p.cs0 % 4 != 0 || (p.in1 != nullptr && p.cs1 % 4 != 0)) return WOFT_EINVAL;
    if (p.flat && (p.taps_x != 1 || p.in1 != nullptr || (p.cs0 != 4 && p.cs0 != 8 && p.cs0 != 16 && p.cs0 != 32)))
        return WOFT_EINVAL;
    // a pixel's K range must lie inside its own row of the tensor (the last pixel would otherwise be read past the allocation)
    if (!p.flat && (p.cs0 < (p.in1 != nullptr ? p.c_split : p.cin_pad) || (p.in1 != nullptr && p.cs1 < p.cin_pad - p.c_split)))
        return WOFT_EINVAL;
    if (p.n_img <= 0 || p.h <= 0 || p.w <= 0 || p.ho <= 0 || p.wo <= 0 || p.taps_y <= 0 || p.taps_x <= 0 ||
        p.stride <= 0 || p.cout <= 0)
        return WOFT_EINVAL;
    if ((p.tile_m != 64 && p.tile_m != 128) || (p.tile_n != 64 && p.tile_n != 128 && !(p.halo == 16 && p.tile_n == 256)))
        return WOFT_EINVAL;
    if (p.cout_pad % p.tile_n != 0 || p.cout > p.cout_pad) return WOFT_EINVAL;
    if (p.epi < 0 || p.epi > WOFT_EPI_FLOWHEAD) return WOFT_EINVAL;
    if (p.epi == WOFT_EPI_FLOWHEAD && (p.halo != 8 || p.precision == 0 || p.stat_sum != nullptr || p.bias_map != nullptr ||
                                       p.e0 == nullptr))
        return WOFT_EINVAL;
    if (p.epi == WOFT_EPI_WH_MEAN && (p.halo != 2 || p.cout_pad != p.tile_n || p.e0 == nullptr || p.e1 == nullptr ||
                                      p.stat_sum != nullptr))
        return WOFT_EINVAL;
    if ((p.epi == WOFT_EPI_RELU_RES_RELU || p.epi == WOFT_EPI_GRU_ZR || p.epi == WOFT_EPI_GRU_Q) && p.e0 == nullptr)
        return WOFT_EINVAL;
    if (p.epi == WOFT_EPI_GRU_Q && p.e1 == nullptr) return WOFT_EINVAL;
    if (p.epi == WOFT_EPI_GRU_ZR && p.out1 == nullptr) return WOFT_EINVAL;
    if (p.e0 != nullptr && p.lde0 % 4 != 0) return WOFT_EINVAL;
    if (p.e1 != nullptr && p.lde1 % 4 != 0) return WOFT_EINVAL;
    if (p.out1 != nullptr && (p.ldo1 % 4 != 0 || p.split % 4 != 0)) return WOFT_EINVAL;
    if ((p.stat_sum == nullptr) != (p.stat_sq == nullptr)) return WOFT_EINVAL;
    // epilogue contract (conv_common.h): 16-byte aligned outputs, no column remap; a ragged last channel group only with
    // the element-wise kinds and without statistics / per-pixel bias
    if (p.epi != WOFT_EPI_WH_MEAN) {
        if (p.out_pitch != 0 || p.ldo % 4 != 0 || p.co_off % 4 != 0 || p.epi == WOFT_EPI_CTX) return WOFT_EINVAL;
        const bool simple = p.epi == WOFT_EPI_LINEAR || p.epi == WOFT_EPI_RELU || p.epi == WOFT_EPI_SIGMOID || p.epi == WOFT_EPI_TANH;
        if (p.cout % 4 != 0 && (!simple || p.stat_sum != nullptr || p.bias_map != nullptr)) return WOFT_EINVAL;
    }
    if (p.in_norm < 0 || p.in_norm > 2) return WOFT_EINVAL;
    if (p.bias_map != nullptr && (p.cout % 4 != 0 || p.ld_bias_map < p.cout || p.ld_bias_map % 4 != 0 ||
                                  p.epi == WOFT_EPI_CTX || p.epi == WOFT_EPI_WH_MEAN || p.out_pitch != 0))
        return WOFT_EINVAL;
    if (p.in_norm != 0 && (p.halo == 0 || p.in1 != nullptr || p.in_mean == nullptr || p.in_rstd == nullptr)) return WOFT_EINVAL;
    if (p.wh0_lookup != nullptr &&
        (p.halo != 2 || p.precision == 0 || p.taps_y != 3 || p.taps_x != 3 || p.cin_pad != 128 || p.in1 != nullptr ||
         p.in_norm != 0 || p.ho != 9 || p.wo != 9 || p.wh0_ld < 324 || p.wh0_ld % 4 != 0 || p.wh0_mean == nullptr ||
         p.wh0_w == nullptr || p.wh0_bias == nullptr))
        return WOFT_EINVAL;
    return WOFT_OK;
}
#endif  // WOFT_ONLY_PREC == 1 (argument checks)

// second != NULL: one launch for two layers that run on the same kernel instance (woft_conv2d_pair)
int WOFT_CAT2(woft_conv_dispatch_p, WOFT_ONLY_PREC)(const woft_conv_params& p, const woft_conv_params* second, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (second != nullptr && (second->precision != p.precision || second->halo != p.halo ||
                              (p.halo == 0 && second->tile_m != p.tile_m) ||
                              (second->tile_n != p.tile_n && p.halo != 16) || p.precision == 0 ||
                              (p.halo != 0 && p.halo != 8 && p.halo != 12 && p.halo != 16)))
        return WOFT_EINVAL;
    if (p.halo == 7)                      // the encoders' 7x7 / stride-2 first layer on its own kernel (conv_stem.hip)
        return second != nullptr ? WOFT_EINVAL : woft_conv_stem_launch(p, stream);
    if (p.halo == 16)                     // wide 1x1 layers on the streamed GEMM kernel (conv_1x1.hip)
        return p.precision == 0 ? WOFT_EINVAL : woft_conv_1x1_launch(p, second, stream);
    for (const woft_conv_params* q : {&p, second}) {
        if (q == nullptr || q->halo == 0) continue;
        // LDS-halo kernels: split-bf16 precisions, stride 1, 3x3 / 1x5 / 5x1 taps, non-flat, same-size output
        if (q->precision == 0 || q->flat || q->stride != 1) return WOFT_EINVAL;
        if (q->ho != q->h + 2 * q->pad_y - q->taps_y + 1 || q->wo != q->w + 2 * q->pad_x - q->taps_x + 1) return WOFT_EINVAL;
        const int64_t cs_max = (q->in1 != nullptr && q->cs1 > q->cs0) ? q->cs1 : q->cs0;
        if ((int64_t)q->n_img * q->h * q->w * cs_max >= (1ll << 31)) return WOFT_EINVAL;     // 32-bit element offsets
        if ((int64_t)q->taps_y * q->taps_x * q->cin_pad >= (1 << 20)) return WOFT_EINVAL;
    }
    if (p.halo != 0) {
        if (p.halo == 8 || p.halo == 12)
            return WOFT_CAT2(woft_conv_regb_launch_p, WOFT_ONLY_PREC)(p, second, stream);
        if (p.halo == 1 && p.tile_n == 128) return launch_halo<8, 16, 128, 2, 2>(p, s);
        if (p.halo == 1 && p.tile_n == 64) return launch_halo<8, 16, 64, 2, 2>(p, s);
        if (p.halo == 2 && p.tile_n == 128 && p.ho == 9 && p.wo == 9) return launch_halo<9, 9, 128, 1, 2>(p, s);
        if (p.halo == 4 && p.tile_n == 128) return launch_halo<4, 16, 128, 2, 2>(p, s);
        if (p.halo == 4 && p.tile_n == 64) return launch_halo<4, 16, 64, 2, 2>(p, s);
        return WOFT_EINVAL;
    }
    if (p.tile_m == 128 && p.tile_n == 128) return launch_conv<128, 128>(p, second, s);
    if (p.tile_m == 128 && p.tile_n == 64) return launch_conv<128, 64>(p, second, s);
    if (p.tile_m == 64 && p.tile_n == 128) return launch_conv<64, 128>(p, second, s);
    return launch_conv<64, 64>(p, second, s);
}

#if WOFT_ONLY_PREC == 1
int woft_conv_regb_launch_p4(const woft_conv_params& p, const woft_conv_params* second, void* stream);   // conv_regb.hip, part 4

static int conv_dispatch_f16mx8(const woft_conv_params& p, const woft_conv_params* second, void* stream) {
    for (const woft_conv_params* q : {&p, second}) {
        if (q == nullptr) continue;
        if (q->precision != 4 || q->halo != p.halo || q->tile_n != p.tile_n || q->flat || q->stride != 1) return WOFT_EINVAL;
        if (q->ho != q->h + 2 * q->pad_y - q->taps_y + 1 || q->wo != q->w + 2 * q->pad_x - q->taps_x + 1) return WOFT_EINVAL;
        const int64_t cs_max = (q->in1 != nullptr && q->cs1 > q->cs0) ? q->cs1 : q->cs0;
        if ((int64_t)q->n_img * q->h * q->w * cs_max >= (1ll << 31)) return WOFT_EINVAL;     // 32-bit element offsets
    }
    return woft_conv_regb_launch_p4(p, second, stream);
}

static int conv_dispatch(const woft_conv_params& p, const woft_conv_params* second, void* stream) {
    switch (p.precision) {               // (the exact-fp32 kernel lives in the bf16 part)
        case 4: return conv_dispatch_f16mx8(p, second, stream);
        case 1: return woft_conv_dispatch_p1(p, second, stream);
        case 3: return woft_conv_dispatch_p3(p, second, stream);
        default: return woft_conv_dispatch_p2(p, second, stream);
    }
}

extern "C" int woft_conv2d(const woft_conv_params* pp, void* stream) {
    if (pp == nullptr) return WOFT_EINVAL;
    const int rc = conv_check(*pp);
    return rc != WOFT_OK ? rc : conv_dispatch(*pp, nullptr, stream);
}

extern "C" int woft_conv2d_pair(const woft_conv_params* a, const woft_conv_params* b, void* stream) {
    if (a == nullptr || b == nullptr) return WOFT_EINVAL;
    int rc = conv_check(*a);
    if (rc == WOFT_OK) rc = conv_check(*b);
    if (rc != WOFT_OK) return rc;
    if (a->stat_sum != nullptr || b->stat_sum != nullptr) return WOFT_EINVAL;     // (statistics rows are indexed by the launch's tiles)
    return conv_dispatch(*a, b, stream);
}
#endif  // WOFT_ONLY_PREC == 1 (woft_conv2d, woft_conv2d_pair)

#if WOFT_ONLY_PREC == 3
extern "C" int woft_corr_gemm_bf16(const void* a, const void* b, int64_t m, int64_t n, int64_t rows_a, int64_t rows_b,
                                   int32_t k, float alpha, void* out, int64_t ldo, int32_t terms, int32_t out_bf16,
                                   void* stream) {
    if (!a || !b || !out || m <= 0 || n <= 0 || k <= 0 || ldo < n || n % 4 != 0 || ldo % 4 != 0) return WOFT_EINVAL;
    if (terms != 1 && terms != 3) return WOFT_EINVAL;
    if (k % (terms == 3 ? 32 : 64) != 0) return WOFT_EINVAL;
    if (rows_a % 128 != 0 || rows_b % 128 != 0 || rows_a < m || rows_b < n || m >= (1ll << 31)) return WOFT_EINVAL;
    woft_conv_params p = {};
    p.n_img = 1; p.ho = 1; p.wo = (int32_t)m;            // M = m rows
    p.alpha = alpha;
    p.cout = (int32_t)n; p.cout_pad = (int32_t)rows_b;
    p.out = (float*)out; p.ldo = ldo;                    // (bf16 storage: the kernel re-types the pointer)
    p.epi = WOFT_EPI_LINEAR;
    const int abl = g_tuning[2];                         // ablation bits (tools/bench_cgemm.py); 0 in production
    hipStream_t s = (hipStream_t)stream;
    // workgroup shape: 2 x 2 waves (128 x 128 tile), one stage.  256 x 128 / one stage and 256 x 256 / two stages
    // (half the operand traffic) were measured at the same 2.05-2.17 ms for level 0: see DESIGN.md section 4.
    if (rows_a % 128 != 0 || rows_b % 128 != 0) return WOFT_EINVAL;
    dim3 grid((unsigned)(ceil_div64(m, 128) * (rows_b / 128)));
    if (terms == 3 && !out_bf16)
        hipLaunchKernelGGL((corr_gemm_bf16_kernel<3, 2, 2, 1>), grid, dim3(256), 0, s, (const __bf16*)a, (const __bf16*)b,
                           2 * k, p, abl);
    else if (terms == 3)
        hipLaunchKernelGGL((corr_gemm_bf16_kernel<3, 2, 2, 1, true>), grid, dim3(256), 0, s, (const __bf16*)a,
                           (const __bf16*)b, 2 * k, p, abl);
    else if (!out_bf16)
        hipLaunchKernelGGL((corr_gemm_bf16_kernel<1, 2, 2, 1>), grid, dim3(256), 0, s, (const __bf16*)a, (const __bf16*)b,
                           k, p, abl);
    else
        hipLaunchKernelGGL((corr_gemm_bf16_kernel<1, 2, 2, 1, true>), grid, dim3(256), 0, s, (const __bf16*)a,
                           (const __bf16*)b, k, p, abl);
    return woft_launch_status();
}

#endif  // WOFT_ONLY_PREC == 3 (woft_corr_gemm_bf16)

#if WOFT_ONLY_PREC == 1
int g_regb_dyn_lds = 0;     // developer knob [3]: extra dynamic LDS bytes of conv_regb launches (occupancy experiments)

extern "C" int woft_set_tuning(int key, int value) {
    if (key < 0 || key >= 4) return WOFT_EINVAL;
    g_tuning[key] = value;
    if (key == 3) g_regb_dyn_lds = value;
    return WOFT_OK;
}

__global__ void split_bf16_lines_kernel(const float* __restrict__ x, int64_t n4, __bf16* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const f32x4 vv = *(const f32x4*)(x + i * 4);
    const bf16x4 h = __builtin_convertvector(vv, bf16x4);
    __bf16* line = out + (i >> 3) * 64 + (i & 7) * 4;       // 32 inputs -> one 128-byte line [32 hi | 32 lo]
    *(bf16x4*)line = h;
    *(bf16x4*)(line + 32) = __builtin_convertvector(vv - __builtin_convertvector(h, f32x4), bf16x4);
}

extern "C" int woft_split_bf16_lines(const float* x, int64_t n, void* out, void* stream) {
    if (!x || !out || n <= 0 || n % 32 != 0) return WOFT_EINVAL;
    hipLaunchKernelGGL(split_bf16_lines_kernel, dim3((unsigned)ceil_div64(n / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                       x, n / 4, (__bf16*)out);
    return woft_launch_status();
}

extern "C" int woft_split_bf16(const float* x, int64_t n, void* hi, void* lo, void* stream) {
    if (!x || !hi || n <= 0 || n % 4 != 0) return WOFT_EINVAL;
    hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)ceil_div64(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, x,
                       n / 4, (__bf16*)hi, (__bf16*)lo);
    return woft_launch_status();
}
#endif  // WOFT_ONLY_PREC == 1
