// LDS-halo convolution with the WEIGHT operand streamed global -> registers (no LDS stage, no LDS-DMA, no
// per-tap barrier).  Same GEMM formulation, arithmetic modes and epilogues as conv_halo_bf16_kernel (conv.hip):
// stride 1, taps 3x3 / 1x5 / 5x1, split-bf16 operands on v_mfma_f32_32x32x16_bf16, fp32 accumulation.
//
// Why: in conv_halo_bf16_kernel every (tap, 32-channel chunk) K step copies a weight tile global -> LDS with LDS-DMA
// and publishes it with a workgroup barrier; measured per step of a 128 x 128 tile (round-1 in-kernel timeline):
// 300 cycles of DMA issue + 930 of fragment reads / MFMAs + 260-350 waiting for the DMA + 230 in the barrier, i.e.
// the matrix pipe idles more than half of a step.  Here a wave owns a 32-column band of the output tile for ALL of
// the tile's rows, so nobody else needs its weights: they are pre-packed on the host in MFMA-fragment order
// ([32-column band][chunk][tap][plane hi, lo][k half][64 lanes][8 bf16] -- the wave's whole K loop is ONE contiguous
// stream, every fragment one fully coalesced 1-KiB global_load_dwordx4) and fetched straight into VGPRs a few steps
// ahead (a register ring).  What remains in LDS is the input halo (double buffered), what remains of the
// synchronisation is ONE barrier per 32-channel chunk (= per 5 or 9 K steps), and between barriers the four waves
// of a workgroup drift freely, so that one wave's memory waits sit beside another's MFMAs on the CU.
//
//   WM = 1: 4 waves x (128 rows x 32 columns): BN = 128, weights read once per workgroup.
//   WM = 2: 2 x 2 waves x (64 rows x 32 columns): BN = 64 (the two row halves fetch the same fragments; L1 serves
//           the second), for layers with too few 128-wide tiles.
#include <type_traits>

#include "conv_common.h"
#include "halo_map.h"

extern int g_regb_dyn_lds;          // conv.hip (woft_set_tuning key 3)

// Build parts (woft_amd/build.py): this file is compiled once per (precision code, input format) -- WOFT_ONLY_PREC in {1, 2, 3},
// WOFT_ONLY_PK in {0, 1} -- each part exporting woft_conv_regb_launch_p<prec>_<pk>; conv.hip's dispatcher of the same precision
// picks the part by the layer's in_fmt.
#if !defined(WOFT_ONLY_PREC) || !defined(WOFT_ONLY_PK)
#error "conv_regb.hip is compiled in parts: -DWOFT_ONLY_PREC=1|2|3 -DWOFT_ONLY_PK=0|1 (woft_amd/build.py)"
#endif
#define WOFT_CAT4_(a, b, c, d) a##b##c##d
#define WOFT_CAT4(a, b, c, d) WOFT_CAT4_(a, b, c, d)
#define WOFT_REGB_ENTRY WOFT_CAT4(woft_conv_regb_launch_p, WOFT_ONLY_PREC, _, WOFT_ONLY_PK)

namespace {

using woft::BK;

// NORM (compile time; encoder layers, round 3): p.in_norm != 0 -- the producer's InstanceNorm (+ ReLU) applied while the halo is
// converted, with conv_halo_bf16_kernel's expression (bit-identical); the per-channel statistics of the chunk travel with its halo rows.
// PK (compile time; round 4): both input sources are SPLIT-PACKED (woft_conv_params.in_fmt): a halo row's 16 bytes already
// are [hi[0..3] | lo[0..3]] of its four channels -- the loader copies them into the two LDS planes, no conversion.
template <int TY, int TX, int KY, int KX, int WM, int TERMS, int NBUF, int DIST, int AD, int CU = 1, int HD = 1, bool IL = true,
          bool NORM = false, bool PK = (WOFT_ONLY_PK != 0)>
__global__ __launch_bounds__(256, 2) void conv_regb_kernel(const woft_conv_params pa, const woft_conv_params pb, const int split) {
    // (two independent layers that run on the same instance of this kernel may share ONE launch -- woft_conv2d_pair: the
    //  workgroups [0, split) belong to the first layer, the rest to the second; split = gridDim.x for a single layer)
    const bool second_layer = (int)blockIdx.x >= split;
    const woft_conv_params p = second_layer ? pb : pa;     // (a copy: a REFERENCE selected between the two argument
                                                            //  structs does not compile -- 'illegal VGPR to SGPR copy')
    const int bid = second_layer ? (int)blockIdx.x - split : (int)blockIdx.x;
    constexpr int NWAVES = 4;
    constexpr int NPIX = TY * TX;
    constexpr int BM = (NPIX + 31) / 32 * 32;
    constexpr int WN = NWAVES / WM;
    constexpr int BN = 32 * WN;
    constexpr int WROWS = BM / WM;
    constexpr int TM = WROWS / 32;
    constexpr int NP = (TERMS == 3) ? 2 : 1;
    constexpr int TAPS = KY * KX;
    // CU: chunks per unrolled group (the ring slot of step s = chunk * TAPS + tap must be a compile-time constant:
    // (CU * TAPS) % NBUF == 0; multi-tap layers: CU = 1, TAPS % NBUF == 0; 1x1 layers: TAPS = 1, CU = NBUF).
    // HD: how many chunks ahead the input tile is requested (1x1: a chunk is a single K step, too short to cover HBM latency)
    static_assert((CU * TAPS) % NBUF == 0 && DIST >= 1 && DIST < NBUF && CU % HD == 0, "register ring: static slots");
    static_assert(BM % (32 * WM) == 0, "bad wave layout");
    constexpr int HX = TX + KX - 1, HY = TY + KY - 1, HROWS = HX * HY;
    constexpr int RH = (HROWS + 31) / 32;
    constexpr int A_PLANE = HROWS * LDB, A_ELEMS = NP * A_PLANE;          // one halo buffer (bf16 elements)
    constexpr int STAGE_ELEMS = 2 * NWAVES * TM * woft::STAGE_FLOATS;     // epilogue staging: all TM tiles of every wave
    constexpr int SMEM_ELEMS = (2 * A_ELEMS > STAGE_ELEMS) ? 2 * A_ELEMS : STAGE_ELEMS;
    constexpr int STEP_ELEMS = NP * 2 * 64 * 8;                           // fragment elements of one K step of a band
    __shared__ __attribute__((aligned(16))) __bf16 smem[SMEM_ELEMS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int r32 = lane & 31, hh = lane >> 5;
    const int v = tid & 7, r0 = tid >> 3;

    const int tyn = (p.ho + TY - 1) / TY, txn = (p.wo + TX - 1) / TX;
    int m_tile, n_tile;
    woft::tile_of_block(bid, p.n_img * tyn * txn, p.cout_pad / BN, m_tile, n_tile);
    const int img0 = m_tile / (tyn * txn);
    const int trem = m_tile - img0 * (tyn * txn);
    const int y0 = (trem / txn) * TY, x0 = (trem % txn) * TX;
    const int n0 = n_tile * BN;
    const int nchunk = p.cin_pad / BK;
    const int nsteps = nchunk * TAPS;

    int hpix[RH];
    bool hok[RH];
#pragma unroll
    for (int j = 0; j < RH; ++j) {
        const int ht = r0 + 32 * j;
        const int hy = ht / HX, hx = ht - hy * HX;
        const int iy = y0 + hy - p.pad_y, ix = x0 + hx - p.pad_x;
        hok[j] = ht < HROWS && iy >= 0 && iy < p.h && ix >= 0 && ix < p.w;
        // (PK: taps outside the image read the ZERO PIXEL ROW that follows a split-packed tensor -- pixel index n_img * h * w,
        //  see woft_conv_params.in_fmt -- so the loader needs no per-element select)
        hpix[j] = hok[j] ? (img0 * p.h + iy) * p.w + ix : (PK ? p.n_img * p.h * p.w : 0);
    }
    f32x4 rh[HD][RH];                                    // ring: the input tile of chunk c waits in rh[c % HD]
    f32x4 nmu[HD], nrs[HD];                              // NORM: mean / rstd of the chunk's channels 4 v .. 4 v + 3
    auto load_halo = [&](int chunk, auto slot_tag) {
        constexpr int hs = decltype(slot_tag)::value;
        const int c0 = chunk * BK;
        const bool second = (p.in1 != nullptr) && (c0 >= p.c_split);
        // (wave-uniform base + 32-bit lane offset: the scalar-base addressing form -- with the lane's 4 v folded into the base the
        //  six addresses of a chunk cost ~34 vector instructions of 64-bit arithmetic, and those add to the MFMAs' SIMD time)
        const float* src = second ? p.in1 + (c0 - p.c_split) : p.in0 + c0;
        const int cs = second ? p.cs1 : p.cs0;
#pragma unroll
        for (int j = 0; j < RH; ++j) rh[hs][j] = *(const f32x4*)(src + (uint32_t)(hpix[j] * cs + 4 * v));
        if constexpr (NORM) {
            nmu[hs] = *(const f32x4*)(p.in_mean + c0 + 4 * v);
            nrs[hs] = *(const f32x4*)(p.in_rstd + c0 + 4 * v);
        }
    };
    auto store_halo_row = [&](__bf16* As, auto slot_tag, auto j_tag) {      // one of this thread's RH halo rows -> LDS
        constexpr int hs = decltype(slot_tag)::value, j = decltype(j_tag)::value;
        const int ht = r0 + 32 * j;
        if (RH * 32 > HROWS && ht >= HROWS) return;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        f32x4 x = rh[hs][j];
        if constexpr (NORM) {                            // (zero padding applies to the NORMALISED map: select afterwards)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float y = (x[e] - nmu[hs][e]) * nrs[hs][e];
                if (p.in_norm == 2) y = fmaxf(y, 0.f);
                x[e] = y;
            }
        }
        if constexpr (PK) {                              // (the zero row supplied the padding: a plain copy)
            static_assert(!NORM, "split-packed inputs are final activations");
            *(bf16x4*)(As + ht * LDB + 4 * v) = packed_hi(x);
            if (NP == 2) *(bf16x4*)(As + A_PLANE + ht * LDB + 4 * v) = packed_lo(x);
            return;
        }
        const f32x4 val = hok[j] ? x : zero;
        const bf16x4 hi = cvt16<TERMS>(val);
        *(bf16x4*)(As + ht * LDB + 4 * v) = hi;
        if (NP == 2) {
            const f32x4 rem = val - widen_bf16x4(hi);
            *(bf16x4*)(As + A_PLANE + ht * LDB + 4 * v) = __builtin_convertvector(rem, bf16x4);
        }
    };
    auto store_halo = [&](__bf16* As, auto slot_tag) {
        [&]<int... J>(std::integer_sequence<int, J...>) {
            (store_halo_row(As, slot_tag, std::integral_constant<int, J>{}), ...);
        }(std::make_integer_sequence<int, RH>{});
    };

    // this wave's weight stream: band (n0 / 32 + wn), steps in (chunk, tap) order, STEP_ELEMS per step
    const __bf16* wstream = (const __bf16*)p.wgt_frag + (int64_t)(n0 / 32 + wn) * nsteps * STEP_ELEMS + lane * 8;
    bf16x8 bq[NBUF][NP][2];
    auto fetch_b = [&](int step, auto slot_tag) {
        constexpr int slot = decltype(slot_tag)::value;
        const int s = step < nsteps ? step : nsteps - 1;                // (past the end: a harmless repeat)
        const __bf16* src = wstream + (int64_t)s * STEP_ELEMS;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) bq[slot][pl][s2] = *(const bf16x8*)(src + (pl * 2 + s2) * 512);
    };

    f32x16 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    int a_off[TM];                   // element offset of this lane's A row inside a halo buffer (tap (0,0), k half hh)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        bool valid;
        const int pl = halo_row_pixel<TY, TX>(wm * WROWS + i * 32 + r32, valid);
        a_off[i] = ((pl / TX) * HX + (pl % TX)) * LDB + hh * 8;
    }

    // developer probe (tools/regb_probe.py): s_memtime stamps of wave 0 -> in_rstd (unused by this kernel otherwise)
    unsigned long long* stamps = (p.in_mean == (const float*)1 && wave == 0 && lane == 0)
                                     ? (unsigned long long*)p.in_rstd + (size_t)bid * 32 : nullptr;
    if (stamps) { stamps[0] = __builtin_amdgcn_s_memtime(); stamps[30] = __builtin_amdgcn_s_memrealtime(); }
    // ---- prologue: halo of chunk 0, the first DIST steps of the weight stream ---------------------------------
    [&]<int... C>(std::integer_sequence<int, C...>) {
        ((C < nchunk ? load_halo(C, std::integral_constant<int, C % HD>{}) : (void)0), ...);
    }(std::make_integer_sequence<int, HD>{});
    [&]<int... S>(std::integer_sequence<int, S...>) {
        (fetch_b(S, std::integral_constant<int, S % NBUF>{}), ...);
    }(std::make_integer_sequence<int, DIST>{});
    store_halo(smem, std::integral_constant<int, 0>{});
    __syncthreads();

    // One 32-channel chunk = TAPS K steps, fully unrolled into "pairs": (tap, k half, two row tiles) = 4 (2 in plain
    // bf16) A-fragment reads + 6 (2) MFMAs, issued term-major so that consecutive MFMAs never chain on one accumulator.
    // The fragments of pair q + AD are requested BEFORE the MFMAs of pair q (ring of AD + 1 register sets) and
    // sched_barriers keep it that way: left to itself the compiler sinks every ds_read next to its use and follows it
    // with s_waitcnt lgkmcnt -- each pair then waits out the full LDS latency (PMC of that version: matrix pipe 55 %
    // busy while LDS, L1 and L2 were all under 30 % busy).
    constexpr int PT = TM, NQ = TAPS * PT, AR = AD + 1;                  // pairs per tap / per chunk
    static_assert(TM % 2 == 0, "row tiles are processed in pairs");
    static_assert(!IL || NQ >= RH, "interleaved halo conversion: one thread-row per pair");
    auto run_chunk = [&](int chunk, auto more_tag, auto phase_tag) {
        constexpr bool more = decltype(more_tag)::value;
        constexpr int phase = decltype(phase_tag)::value;                // chunk % CU
        const __bf16* As = smem + (chunk & 1) * A_ELEMS;
        bf16x8 aq[AR][2][NP];
        auto load_a = [&](auto q_tag) {
            constexpr int q = decltype(q_tag)::value;
            constexpr int tap = q / PT, r = q % PT, s2 = r / (TM / 2), i0 = 2 * (r % (TM / 2));
            constexpr int ky = tap / KX, kx = tap - ky * KX;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
                    aq[q % AR][d][pl] = *(const bf16x8*)(As + a_off[i0 + d] + pl * A_PLANE + (ky * HX + kx) * LDB + s2 * 16);
        };
        [&]<int... Q>(std::integer_sequence<int, Q...>) { (load_a(std::integral_constant<int, Q>{}), ...); }
        (std::make_integer_sequence<int, (AD < NQ ? AD : NQ)>{});
        __builtin_amdgcn_sched_barrier(0);
        [&]<int... Q>(std::integer_sequence<int, Q...>) {
            ([&] {
                constexpr int q = Q;
                constexpr int tap = q / PT, r = q % PT, s2 = r / (TM / 2), i0 = 2 * (r % (TM / 2));
                constexpr int slot = (phase * TAPS + tap) % NBUF, as = q % AR;
                if constexpr (r == 0) {
                    // weights of step (chunk, tap) + DIST into the slot that step (chunk, tap) - (NBUF - DIST) vacated
                    fetch_b(chunk * TAPS + tap + DIST, std::integral_constant<int, (phase * TAPS + tap + DIST) % NBUF>{});
                    // input tile of chunk + HD into the ring slot chunk's own tile left at the end of the previous chunk
                    if (tap == 0 && chunk + HD < nchunk) load_halo(chunk + HD, std::integral_constant<int, phase % HD>{});
                }
                if constexpr (q + AD < NQ) load_a(std::integral_constant<int, q + AD>{});
                __builtin_amdgcn_sched_barrier(0);
                // IL: the next chunk's halo (requested at the first tap of this chunk) is converted and written to the other
                // buffer ONE thread-row per pair over the last RH pairs of the chunk, its ~20 vector / LDS instructions
                // placed in the issue gaps between this pair's MFMAs (sched_group_barrier: 1 MFMA, then up to 4 others)
                // instead of as one block after the last MFMA, where the matrix pipe idled for the whole conversion
                constexpr bool il_row = IL && more && q >= NQ - RH;
                if constexpr (il_row)
                    store_halo_row(smem + ((chunk + 1) & 1) * A_ELEMS, std::integral_constant<int, (phase + 1) % HD>{},
                                   std::integral_constant<int, q - (NQ - RH)>{});
                if (NP == 2) {
                    acc[i0] = mma16<TERMS>(aq[as][0][NP - 1], bq[slot][0][s2], acc[i0]);
                    acc[i0 + 1] = mma16<TERMS>(aq[as][1][NP - 1], bq[slot][0][s2], acc[i0 + 1]);
                    acc[i0] = mma16<TERMS>(aq[as][0][0], bq[slot][NP - 1][s2], acc[i0]);
                    acc[i0 + 1] = mma16<TERMS>(aq[as][1][0], bq[slot][NP - 1][s2], acc[i0 + 1]);
                }
                acc[i0] = mma16<TERMS>(aq[as][0][0], bq[slot][0][s2], acc[i0]);
                acc[i0 + 1] = mma16<TERMS>(aq[as][1][0], bq[slot][0][s2], acc[i0 + 1]);
                if constexpr (il_row) {
#pragma unroll
                    for (int g = 0; g < (NP == 2 ? 6 : 2); ++g) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);      // up to three VALU
                        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);      // up to one LDS write
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                // next chunk's halo -> the other buffer (free since the barrier that ended the previous chunk); late
                // in the chunk so that its loads had the whole chunk to land
                if constexpr (q == NQ - 1 && !IL) {
                    if (more) store_halo(smem + ((chunk + 1) & 1) * A_ELEMS, std::integral_constant<int, (phase + 1) % HD>{});
                }
            }(), ...);
        }(std::make_integer_sequence<int, NQ>{});
        if (more) __syncthreads();
        if (stamps && chunk < 12) stamps[1 + chunk] = __builtin_amdgcn_s_memtime();
    };
    auto run_phase = [&](int chunk, auto more_tag) {     // chunk % CU selects the unrolled body with the right ring slots
        if constexpr (CU == 1) {
            run_chunk(chunk, more_tag, std::integral_constant<int, 0>{});
        } else {
            const int ph = chunk % CU;
            [&]<int... P>(std::integer_sequence<int, P...>) {
                ((ph == P ? run_chunk(chunk, more_tag, std::integral_constant<int, P>{}) : (void)0), ...);
            }(std::make_integer_sequence<int, CU>{});
        }
    };
    for (int chunk = 0; chunk + 1 < nchunk; ++chunk) run_phase(chunk, std::true_type{});
    run_phase(nchunk - 1, std::false_type{});
    __syncthreads();                                     // halo buffers are dead: reuse them as epilogue staging

    const HaloRowMap<TY, TX> rowmap{img0, p.n_img, y0, x0, p.ho, p.wo};
    if (p.epi == WOFT_EPI_FLOWHEAD) {
        // Flow head, second conv folded into the first one's epilogue (update.py:10-17: conv2(relu(conv1(h))), 3 x 3, 2
        // output channels).  A 3 x 3 conv is linear in its input pixels: delta[q] = b2 + sum_taps <W2[tap], y[q + tap]>, so
        // this launch emits, per pixel p and tap, the 2 partial dot products s[p][tap][o] = <W2[o][:, tap], y[p]> over the
        // channels this workgroup holds (18 values per pixel) and woft_flow_head_gather adds the 9 neighbours' shares --
        // the 256-channel activation (33 MB at 1/8 of 1080p, the store tail of this launch and three reads of the next)
        // is never written.  The partial products run on the matrix cores: relu(acc + bias) is transposed through the
        // wave's LDS staging area into A fragments (lane = pixel row, k = its 32 channels), split into bf16 hi / lo as any
        // other activation, and multiplied with the pre-split W2 fragments of the band (p.e0, [band][k half][plane][64][8],
        // column j = tap * 2 + o, 18 of 32 used).  The 32-channel shares of the waves of one row group are totalled in a
        // fixed order through LDS; the column tiles' shares land in separate planes of p.out ([n_tile][pixel][ldo]).
        float* stage = (float*)smem + wave * TM * woft::STAGE_FLOATS;
        const int ncol = n0 + wn * 32;
        f32x4 bv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bv[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.bias != nullptr) bv[q] = *(const f32x4*)(p.bias + ncol + (q >> 1) * 16 + 8 * hh + (q & 1) * 4);
        }
        const __bf16* wf = (const __bf16*)p.e0 + (int64_t)(ncol / 32) * (2 * NP * 512) + lane * 8;
        bf16x8 w2[2][NP];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) w2[s2][pl] = *(const bf16x8*)(wf + (s2 * NP + pl) * 512);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                stage[i * woft::STAGE_FLOATS + ((r & 3) + 8 * (r >> 2) + 4 * hh) * woft::STAGE_LD + r32] = acc[i][r];
        __builtin_amdgcn_wave_barrier();
        f32x16 sacc[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float* row = stage + i * woft::STAGE_FLOATS + r32 * woft::STAGE_LD + 8 * hh;
            bf16x8 ah[2], al[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    f32x4 yv = *(const f32x4*)(row + 16 * s2 + 4 * q);
                    const f32x4 b4 = bv[2 * s2 + q];
#pragma unroll
                    for (int e = 0; e < 4; ++e) yv[e] = fmaxf(p.alpha * yv[e] + b4[e], 0.f);
                    const bf16x4 hi = cvt16<TERMS>(yv);
                    bf16x4 lo = hi;
                    if constexpr (NP == 2) lo = __builtin_convertvector(yv - widen_bf16x4(hi), bf16x4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { ah[s2][4 * q + e] = hi[e]; al[s2][4 * q + e] = lo[e]; }
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[i][r] = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                if (NP == 2) {
                    sacc[i] = mma16<TERMS>(al[s2], w2[s2][0], sacc[i]);
                    sacc[i] = mma16<TERMS>(ah[s2], w2[s2][NP - 1], sacc[i]);
                }
                sacc[i] = mma16<TERMS>(ah[s2], w2[s2][0], sacc[i]);
            }
        }
        __builtin_amdgcn_wave_barrier();
        constexpr int SLD = 20;                          // floats per pixel row of a share: 18 values + 2 (16-byte rows)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (r32 < SLD) stage[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh) * SLD + r32] = sacc[i][r];
        __syncthreads();
        const int64_t M = (int64_t)p.n_img * p.ho * p.wo;
        float* dst = p.out + (int64_t)n_tile * M * p.ldo;
        for (int idx = tid; idx < BM * (SLD / 4); idx += 256) {
            const int row = idx / (SLD / 4), c4 = (idx - row * (SLD / 4)) * 4;
            const int wmr = row / WROWS, lr = row - wmr * WROWS;
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w_ = 0; w_ < WN; ++w_)
                t += *(const f32x4*)((const float*)smem + (wmr * WN + w_) * TM * woft::STAGE_FLOATS + lr * SLD + c4);
            const int64_t m = rowmap(row);
            if (m >= 0) *(f32x4*)(dst + m * p.ldo + c4) = t;
        }
        return;
    }
    f32x16 acc2[TM][1];
#pragma unroll
    for (int i = 0; i < TM; ++i) acc2[i][0] = acc[i];
    if (stamps) stamps[14] = __builtin_amdgcn_s_memtime();
    woft::conv_epilogue_t<TM, 1, WROWS, 32, TM>(p, acc2, (float*)smem + wave * TM * woft::STAGE_FLOATS, rowmap, n0, wm, wn,
                                                lane, m_tile, stamps ? stamps + 16 : nullptr);
    if (stamps) { stamps[15] = __builtin_amdgcn_s_memtime(); stamps[31] = __builtin_amdgcn_s_memrealtime(); }
}

template <int WM, int TY = 8>
int launch_regb(const woft_conv_params& p, const woft_conv_params* second, hipStream_t s) {
    constexpr int TX = 16, BN = 128 / WM;
    auto blocks = [](const woft_conv_params& q) {
        const int tyn = (q.ho + TY - 1) / TY, txn = (q.wo + TX - 1) / TX;
        return (int64_t)q.n_img * tyn * txn * (q.cout_pad / BN);
    };
    const woft_conv_params& pb = second ? *second : p;
    const int split = (int)blocks(p);
    dim3 grid((unsigned)(blocks(p) + (second ? blocks(pb) : 0)));
#define REGB(KY, KX, T, NB, D) \
    woft_launch(0, conv_regb_kernel<TY, TX, KY, KX, WM, T, NB, D, 2>, grid, dim3(256), (size_t)g_regb_dyn_lds, s, p, pb, split)
#if WOFT_ONLY_PK == 0
#define REGB_NORM_INSTANCE(T) \
            if (p.taps_y == 3 && p.taps_x == 3 && second == nullptr)                                        \
                woft_launch(0, conv_regb_kernel<TY, TX, 3, 3, WM, T, 3, 2, 2, 1, 1, true, true>, grid, dim3(256), (size_t)g_regb_dyn_lds, s, p, pb, split); \
            else return WOFT_EINVAL;
#else
#define REGB_NORM_INSTANCE(T) return WOFT_EINVAL;
#endif
#define REGB_TAPS(T)                                                     \
    if (p.in_norm != 0) {                 /* encoder residual blocks: 3x3, 8x16 x 64 tiles, single layer */ \
        if constexpr (WM == 2 && TY == 8) {                                                                 \
            REGB_NORM_INSTANCE(T)                                                                           \
        } else return WOFT_EINVAL;                                                                          \
    } else if (p.taps_y == 3 && p.taps_x == 3) REGB(3, 3, T, 3, 2);      \
    else if (p.taps_y == 1 && p.taps_x == 5) REGB(1, 5, T, 5, 3);        \
    else if (p.taps_y == 5 && p.taps_x == 1) REGB(5, 1, T, 5, 3);        \
    else if (p.taps_y == 1 && p.taps_x == 1) {    /* 1x1: three chunks per unrolled group, input tile three chunks ahead; \
                                                     64-column tiles only (the 128-column layout does not fit 256 registers) */ \
        if constexpr (WM == 2 && TY == 8)                                                                                       \
            woft_launch(0, conv_regb_kernel<TY, TX, 1, 1, WM, T, 3, 2, 1, 3, 3, false>, grid, dim3(256), (size_t)g_regb_dyn_lds, s, p, pb, split); \
        else return WOFT_EINVAL;                                                                                                \
    } else return WOFT_EINVAL
    if (p.precision != WOFT_ONLY_PREC) return WOFT_EINVAL;
#if WOFT_ONLY_PREC == 1
    REGB_TAPS(3);
#elif WOFT_ONLY_PREC == 3
    REGB_TAPS(16);
#else
    REGB_TAPS(1);
#endif
#undef REGB_TAPS
#undef REGB
    return woft_launch_status();
}

}  // namespace

// Called by woft_conv2d for p.halo == 8 (8 x 16-pixel tiles) and 12 (4 x 16 pixels x 128 columns: four waves = four column
// bands of 64 rows -- the per-wave work of the 8 x 16 x 64 layout with the weights fetched once per workgroup), after its
// argument checks.
int WOFT_REGB_ENTRY(const woft_conv_params& p, const woft_conv_params* second, void* stream) {
    // input format of this part: both sources split-packed (WOFT_ONLY_PK = 1) or both fp32
    for (const woft_conv_params* q : {&p, second}) {
        if (q == nullptr) continue;
        const int want = WOFT_ONLY_PK ? ((q->in1 != nullptr) ? 3 : 1) : 0;
        if ((q->in_fmt & 3) != want || (WOFT_ONLY_PK && q->in_norm != 0)) return WOFT_EINVAL;
    }
    if (second != nullptr) {            // one launch for two layers: the same kernel instance, no probe
        const woft_conv_params& b = *second;
        if (b.halo != p.halo || b.tile_n != p.tile_n || b.taps_y != p.taps_y || b.taps_x != p.taps_x || b.precision != p.precision ||
            b.wgt_frag == nullptr || b.in_norm != 0 || b.in_mean != nullptr || p.in_mean != nullptr || b.wh0_lookup != nullptr ||
            b.epi == WOFT_EPI_WH_MEAN || b.cout_pad % b.tile_n != 0 || (p.taps_y * p.taps_x == 1))
            return WOFT_EINVAL;
        if (b.epi == WOFT_EPI_FLOWHEAD && (b.e0 == nullptr || b.ldo < 20 || b.ldo % 4 != 0 || b.co_off != 0 || b.cout % 32 != 0)) return WOFT_EINVAL;
    }
    if (p.halo == 12) {
        if (p.wgt_frag == nullptr || p.in_norm != 0 || (p.in_mean != nullptr && p.in_mean != (const float*)1) || p.wh0_lookup != nullptr ||
            p.epi == WOFT_EPI_WH_MEAN || p.epi == WOFT_EPI_FLOWHEAD || p.tile_n != 128 || p.cout_pad % 128 != 0 || p.taps_y * p.taps_x == 1) return WOFT_EINVAL;
        return launch_regb<1, 4>(p, second, (hipStream_t)stream);
    }
    if (p.wgt_frag == nullptr || p.wh0_lookup != nullptr || p.epi == WOFT_EPI_WH_MEAN) return WOFT_EINVAL;
    if (p.in_norm != 0) {               // InstanceNorm applied on load: the 3x3 / 64-column instance only
        if (p.in_mean == nullptr || p.in_mean == (const float*)1 || p.in_rstd == nullptr || p.in1 != nullptr || p.tile_n != 64 ||
            p.taps_y != 3 || p.taps_x != 3 || second != nullptr || p.epi == WOFT_EPI_FLOWHEAD)
            return WOFT_EINVAL;
    } else if (p.in_mean != nullptr && p.in_mean != (const float*)1) return WOFT_EINVAL;
    if (p.stat_sum != nullptr && p.tile_n != 64) return WOFT_EINVAL;    // (statistics rows: two row halves per tile = the 2 x 2 wave layout)
    if (p.epi == WOFT_EPI_FLOWHEAD && (p.e0 == nullptr || p.ldo < 20 || p.ldo % 4 != 0 || p.co_off != 0 || p.cout % 32 != 0)) return WOFT_EINVAL;
    if (p.tile_n == 128 && p.cout_pad % 128 == 0) return launch_regb<1>(p, second, (hipStream_t)stream);
    if (p.tile_n == 64 && p.cout_pad % 64 == 0) return launch_regb<2>(p, second, (hipStream_t)stream);
    return WOFT_EINVAL;
}
