// LDS-halo convolution with the WEIGHT operand streamed global -> registers (no LDS stage, no LDS-DMA, no per-tap
// barrier): the body of ONE output tile as a device function, called by conv_regb_kernel (conv_regb.hip: one tile per
// workgroup, one layer -- or two independent ones -- per launch).
// Same GEMM formulation, arithmetic modes and epilogues as conv_halo_bf16_kernel (conv.hip): stride 1, taps 3x3 / 1x5 /
// 5x1, split-bf16 operands on v_mfma_f32_32x32x16_bf16, fp32 accumulation.
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
//
// TERMS = 28 (precision code 4, "f16mx8"; round 4): an fp32-emulating product in TWO matrix-pipe passes instead of
// bf16x3's three --  a * w ~= fp16(a) * fp16(w) + mx8(a - fp16(a)) * mx8(w) + mx8(a) * mx8(w - fp16(w)):  main term on
// v_mfma_f32_32x32x16_f16, both cross terms on the block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3, one E8M0 scale per
// 32-element K block; twice the bf16 rate).  K = 64 of one scaled MFMA = TWO TAPS x this chunk's 32 channels: MX block 0 = tap t,
// block 1 = tap t + 1 (operand convention pinned by tools/micro/mx_layout_probe.hip: block b = bytes 16 b .. 16 b + 15 of both
// lane halves, lane half hh = channels 16 hh .. + 15, block b's scale = the scale operand of lane half b).  Per halo row and chunk
// the LDS holds the fp16 plane (as TERMS = 16) plus two fp8 planes -- mx8(a), mx8(a - fp16(a)): 32 data bytes + the block's scale
// byte, 48-byte pitch -- written by the loader; the weights' three forms come pre-packed (wgt_frag: fp16 fragments, wgt_mx: the
// fp8 fragments of w and of w - fp16(w) per tap pair with their scales).  Measured error 2.2-2.3 x bf16x3's (mx_split_probe).
//
// Measured and not kept (rounds 3-4; git history): split-packed / MXP activations written by the producers'
// epilogues (loader copies instead of converting: -80 % vector instructions, no launch faster), InstanceNorm applied while the
// halo is converted (encoder layers: +-0 against the LDS-halo kernel), 1x1 layers on this kernel (45 vs 37 us on convc1).
#pragma once
#include <type_traits>

#include "conv_common.h"
#include "halo_map.h"

namespace {

using woft::BK;

template <int TY, int TX, int KY, int KX, int WM, int TERMS>
struct RegbGeom {
    static constexpr int NWAVES = 4;
    static constexpr int NPIX = TY * TX;
    static constexpr int BM = (NPIX + 31) / 32 * 32;
    static constexpr int WN = NWAVES / WM;
    static constexpr int BN = 32 * WN;
    static constexpr int WROWS = BM / WM;
    static constexpr int TM = WROWS / 32;
    static constexpr int NP = (TERMS == 3) ? 2 : 1;
    static constexpr bool MX = (TERMS == 28);
    static constexpr int TAPS = KY * KX;
    static constexpr int HX = TX + KX - 1, HY = TY + KY - 1, HROWS = HX * HY;
    static constexpr int RH = (HROWS + 31) / 32;
    static constexpr int QPITCH = 48;                                     // MX: bytes per row of an fp8 plane (32 data + scale + pad)
    static constexpr int Q_PLANE = HROWS * QPITCH / 2;                    // ... in bf16-sized elements
    static constexpr int A_PLANE = HROWS * LDB;
    static constexpr int A_ELEMS = NP * A_PLANE + (MX ? 2 * Q_PLANE : 0); // one halo buffer (bf16 elements)
    static constexpr int STAGE_ELEMS = 2 * NWAVES * TM * woft::STAGE_FLOATS;   // epilogue staging: all TM tiles of every wave
    static constexpr int SMEM_ELEMS = (2 * A_ELEMS > STAGE_ELEMS) ? 2 * A_ELEMS : STAGE_ELEMS;
    static_assert(BM % (32 * WM) == 0, "bad wave layout");
};

// One output tile: pixel tile m_tile (image-major, then rows of TY x TX tiles) x column tile n_tile (BN columns) of layer p.
// smem: RegbGeom<...>::SMEM_ELEMS bf16 elements, 16-byte aligned; all 256 threads of the workgroup call together.
// The caller guarantees that the tile's inputs are visible to this CU and may reuse smem after the call returns (the body ends
// with the epilogue's last store ISSUED, not completed).
// ParamsT: woft_conv_params -- or the same struct behind a constant-address-space reference (a table in device memory: only
// address space 4 makes its fields SCALAR loads -- through a plain pointer the compiler reads them with vector loads, base
// addresses end up in VGPRs and the epilogue's "+s" operands fail with 'illegal VGPR to SGPR copy').
template <int TY, int TX, int KY, int KX, int WM, int TERMS, int NBUF, int DIST, int AD, bool IL = true, class ParamsT = woft_conv_params>
__device__ __forceinline__ void regb_tile(const ParamsT& p, const int m_tile, const int n_tile, __bf16* smem,
                                          unsigned long long* stamps) {
    using G = RegbGeom<TY, TX, KY, KX, WM, TERMS>;
    constexpr int WN = G::WN, BN = G::BN, WROWS = G::WROWS, TM = G::TM, NP = G::NP, TAPS = G::TAPS;
    constexpr bool MX = G::MX;
    constexpr int NPAIR = (TAPS + 1) / 2;                                 // MX: tap pairs per chunk (an odd last tap pairs with zero weights)
    // the ring slot of step s = chunk * TAPS + tap must be a compile-time constant: TAPS % NBUF == 0
    static_assert(TAPS % NBUF == 0 && DIST >= 1 && DIST < NBUF, "register ring: static slots");
    constexpr int HX = G::HX, HROWS = G::HROWS, RH = G::RH, QPITCH = G::QPITCH, Q_PLANE = G::Q_PLANE;
    constexpr int A_PLANE = G::A_PLANE, A_ELEMS = G::A_ELEMS;
    constexpr int STEP_ELEMS = NP * 2 * 64 * 8;                           // fragment elements of one K step of a band

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int r32 = lane & 31, hh = lane >> 5;
    const int v = tid & 7, r0 = tid >> 3;

    const int tyn = (p.ho + TY - 1) / TY, txn = (p.wo + TX - 1) / TX;
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
        hpix[j] = hok[j] ? (img0 * p.h + iy) * p.w + ix : 0;
    }
    f32x4 rh[RH];                                        // the input tile of the next chunk, in flight
    auto load_halo = [&](int chunk) {
        const int c0 = chunk * BK;
        const bool second = (p.in1 != nullptr) && (c0 >= p.c_split);
        // (wave-uniform base + 32-bit lane offset: the scalar-base addressing form -- with the lane's 4 v folded into the base the
        //  six addresses of a chunk cost ~34 vector instructions of 64-bit arithmetic)
        const float* src = second ? p.in1 + (c0 - p.c_split) : p.in0 + c0;
        const int cs = second ? p.cs1 : p.cs0;
#pragma unroll
        for (int j = 0; j < RH; ++j) rh[j] = *(const f32x4*)(src + (uint32_t)(hpix[j] * cs + 4 * v));
    };
    auto store_halo_row = [&](__bf16* As, auto j_tag) {      // one of this thread's RH halo rows -> LDS
        constexpr int j = decltype(j_tag)::value;
        const int ht = r0 + 32 * j;
        if (RH * 32 > HROWS && ht >= HROWS) return;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        const f32x4 val = hok[j] ? rh[j] : zero;
        if constexpr (MX) {
            // fp16 plane + the two block-scaled fp8 planes of this row's 32 channels (the row's 8 loader lanes = one MX block)
            const bf16x4 h16 = cvt16<16>(val);
            *(bf16x4*)(As + ht * LDB + 4 * v) = h16;
            const f32x4 la = val - __builtin_convertvector(__builtin_bit_cast(f16x4, h16), f32x4);
            float ma = fmaxf(fmaxf(fabsf(val[0]), fabsf(val[1])), fmaxf(fabsf(val[2]), fabsf(val[3])));
            // block maximum over the row's eight loader lanes (an aligned group of 8): three DPP steps on the vector ALU -- half-row
            // mirror (i <-> 7 - i), then the quad swaps xor 1 and xor 2.  (The first version used __shfl_xor = ds_bpermute: three
            // dependent LDS round trips per row, and every lgkmcnt wait also drained the fragment prefetches behind them.)
            auto dpp_max = [](float x, auto ctrl_tag) {
                constexpr int ctrl = decltype(ctrl_tag)::value;
                const int xi = __builtin_bit_cast(int, x);
                const int yi = __builtin_amdgcn_update_dpp(xi, xi, ctrl, 0xf, 0xf, false);
                return fmaxf(x, __builtin_bit_cast(float, yi));
            };
            ma = dpp_max(ma, std::integral_constant<int, 0x141>{});     // row_half_mirror
            ma = dpp_max(ma, std::integral_constant<int, 0xB1>{});      // quad_perm [1,0,3,2]
            ma = dpp_max(ma, std::integral_constant<int, 0x4E>{});      // quad_perm [2,3,0,1]
            // E8M0 scale 2^(s - 127) with the block maximum scaled into [128, 256) (e4m3's largest: 448): s = biased exponent - 7.
            // The remainder a - fp16(a) is at most 2^-11 of its element WHILE fp16(a) IS NORMAL, so its block needs no maximum of
            // its own: scale s - 11.  Below 2^-14 the fp16 grid is absolute (2^-24): a remainder can reach 2^-25 whatever the block's
            // maximum, so s is floored at 105 (remainder scale 94: 2^-25 / 2^(94 - 127) = 256 <= 448; the block's own values then
            // sit below 128 -- nothing overflows, v_cvt_pk_fp8_f32 does not saturate).  Round-4 advisor finding: the floor was 11 and
            // a block of tiny ReLU outputs (max |a| < 2^-15) produced fp8 NaNs.
            int sa = (int)((__builtin_bit_cast(uint32_t, ma) >> 23) & 0xffu) - 7;
            sa = sa < 105 ? 105 : sa;
            const int sl = sa - 11;
            const float ia = __builtin_bit_cast(float, (uint32_t)(254 - sa) << 23), il = __builtin_bit_cast(float, (uint32_t)(254 - sl) << 23);
            int qa = 0, ql = 0;
            qa = __builtin_amdgcn_cvt_pk_fp8_f32(val[0] * ia, val[1] * ia, qa, false);
            qa = __builtin_amdgcn_cvt_pk_fp8_f32(val[2] * ia, val[3] * ia, qa, true);
            ql = __builtin_amdgcn_cvt_pk_fp8_f32(la[0] * il, la[1] * il, ql, false);
            ql = __builtin_amdgcn_cvt_pk_fp8_f32(la[2] * il, la[3] * il, ql, true);
            char* qpa = (char*)(As + A_PLANE) + ht * QPITCH;
            char* qpl = qpa + 2 * Q_PLANE;
            *(int*)(qpa + 4 * v) = qa;
            *(int*)(qpl + 4 * v) = ql;
            if (v == 0) { qpa[32] = (char)sa; qpl[32] = (char)sl; }
            return;
        } else {
            const bf16x4 hi = cvt16<TERMS>(val);
            *(bf16x4*)(As + ht * LDB + 4 * v) = hi;
            if (NP == 2) {
                const f32x4 rem = val - widen_bf16x4(hi);
                *(bf16x4*)(As + A_PLANE + ht * LDB + 4 * v) = __builtin_convertvector(rem, bf16x4);
            }
        }
    };
    auto store_halo = [&](__bf16* As) {
        [&]<int... J>(std::integer_sequence<int, J...>) {
            (store_halo_row(As, std::integral_constant<int, J>{}), ...);
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
    int q_off[TM];                   // MX: byte offset of this lane's row inside an fp8 plane (tap (0, 0)): data of its channel half at + 16 hh, scale byte at + 32
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        bool valid;
        const int pl = halo_row_pixel<TY, TX>(wm * WROWS + i * 32 + r32, valid);
        a_off[i] = ((pl / TX) * HX + (pl % TX)) * LDB + hh * 8;
        q_off[i] = ((pl / TX) * HX + (pl % TX)) * QPITCH;
    }
    // MX: this wave's fp8 weight stream -- per (chunk, tap pair, term): 64 lanes x 32 data bytes, then 64 scale dwords
    typedef int i32x8 __attribute__((ext_vector_type(8)));
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    constexpr int MXQ = 64 * 32 + 64 * 4;                                  // bytes per (pair, term)
    const char* wmx = MX ? (const char*)p.wgt_mx + (int64_t)(n0 / 32 + wn) * nchunk * NPAIR * 2 * MXQ : nullptr;
    i32x8 wq[1][2];                                                        // [one register set][term: w (meets l_a), l_w (meets a)]
    int wsc[1][2];
    auto fetch_mx = [&](int pair_idx, auto slot_tag) {                     // pair_idx = chunk * NPAIR + pair (clamped at the end)
        constexpr int slot = decltype(slot_tag)::value;
        const int last = nchunk * NPAIR - 1;
        const char* src = wmx + (int64_t)(pair_idx < last ? pair_idx : last) * (2 * MXQ);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const i32x4 lo = *(const i32x4*)(src + t * MXQ + lane * 32), hi = *(const i32x4*)(src + t * MXQ + lane * 32 + 16);
            wq[slot][t] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            wsc[slot][t] = *(const int*)(src + t * MXQ + 64 * 32 + lane * 4);
        }
    };

    if (stamps) { stamps[0] = __builtin_amdgcn_s_memtime(); stamps[30] = __builtin_amdgcn_s_memrealtime(); }
    // ---- prologue: halo of chunk 0, the first DIST steps of the weight stream ---------------------------------
    load_halo(0);
    [&]<int... S>(std::integer_sequence<int, S...>) {
        (fetch_b(S, std::integral_constant<int, S % NBUF>{}), ...);
    }(std::make_integer_sequence<int, DIST>{});
    if constexpr (MX) fetch_mx(0, std::integral_constant<int, 0>{});
    store_halo(smem);
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
    auto run_chunk = [&](int chunk, auto more_tag) {
        constexpr bool more = decltype(more_tag)::value;
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
                constexpr int slot = tap % NBUF, as = q % AR;
                if constexpr (r == 0) {
                    // weights of step (chunk, tap) + DIST into the slot that step (chunk, tap) - (NBUF - DIST) vacated
                    fetch_b(chunk * TAPS + tap + DIST, std::integral_constant<int, (tap + DIST) % NBUF>{});
                    // input tile of the next chunk into the registers this chunk's own tile left at the end of the previous chunk
                    if (tap == 0 && chunk + 1 < nchunk) load_halo(chunk + 1);
                }
                if constexpr (q + AD < NQ) load_a(std::integral_constant<int, q + AD>{});
                __builtin_amdgcn_sched_barrier(0);
                // IL: the next chunk's halo (requested at the first tap of this chunk) is converted and written to the other
                // buffer ONE thread-row per pair over the last RH pairs of the chunk, its ~20 vector / LDS instructions
                // placed in the issue gaps between this pair's MFMAs (sched_group_barrier: 1 MFMA, then up to 4 others)
                // instead of as one block after the last MFMA, where the matrix pipe idled for the whole conversion
                constexpr bool il_row = IL && more && q >= NQ - RH;
                if constexpr (il_row)
                    store_halo_row(smem + ((chunk + 1) & 1) * A_ELEMS, std::integral_constant<int, q - (NQ - RH)>{});
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
                    if (more) store_halo(smem + ((chunk + 1) & 1) * A_ELEMS);
                }
            }(), ...);
        }(std::make_integer_sequence<int, NQ>{});
        if (more) __syncthreads();
        if (stamps && chunk < 12) stamps[1 + chunk] = __builtin_amdgcn_s_memtime();
    };
    // ---- f16mx8: one chunk = TAPS fp16 steps (2 MFMAs per row tile) + NPAIR tap pairs of two scaled fp8 MFMAs per row tile.
    //      Register diet (the first version spilled 170-900 bytes per lane): the main term's A fragments are two half sets --
    //      k half 1 of tap t is requested before the MFMAs of its k half 0, k half 0 of tap t + 1 before the MFMAs of k half 1 --;
    //      the fp8 fragments go row tile by row tile, one tile ahead; the fp8 weights of pair p + 1 are requested right after pair
    //      p's MFMAs consumed the single register set (two taps of lead).
#ifndef MX_FENCE
#define MX_FENCE __builtin_amdgcn_sched_barrier(0)
#endif
    auto run_chunk_mx = [&](int chunk, auto more_tag) {
        constexpr bool more = decltype(more_tag)::value;
        const __bf16* As = smem + (chunk & 1) * A_ELEMS;
        const char* Qa = (const char*)(As + A_PLANE);
        const char* Ql = Qa + 2 * Q_PLANE;
        // Two pipelines.  DEEP (two row tiles per wave: the 64-column and the 4 x 16-pixel layouts): ALL fragments of tap t + 1 -- and,
        // at a pair's last tap, the pair's fp8 fragments and scales -- are requested before the MFMAs of tap t (two register sets by
        // tap parity).  Four row tiles per wave (8 x 16 pixels x 128 columns) do not have the registers: the main term's fragments go
        // as two half sets (k half 1 of tap t before the MFMAs of its k half 0, k half 0 of tap t + 1 before those of k half 1), the
        // fp8 fragments row tile by row tile.
        constexpr bool DEEP = TM <= 2;
        constexpr bool AM2 = DEEP && TAPS <= 5;                            // (3x3: the two-set form spills -- half sets there too)
        constexpr int NAM = AM2 ? 2 : 1;
        bf16x8 am[NAM][2][TM];                                             // [tap parity][k half][row tile]
        auto load_am = [&](auto tap_tag, auto s2_tag) {
            constexpr int tap = decltype(tap_tag)::value, s2 = decltype(s2_tag)::value;
            constexpr int ky = tap / KX, kx = tap - ky * KX;
#pragma unroll
            for (int i = 0; i < TM; ++i) am[AM2 ? (tap & 1) : 0][s2][i] = *(const bf16x8*)(As + a_off[i] + (ky * HX + kx) * LDB + s2 * 16);
        };
        i32x8 qa[DEEP ? TM : 1], ql[DEEP ? TM : 1];
        int sqa[DEEP ? TM : 1], sql[DEEP ? TM : 1];
        auto load_q = [&](auto pr_tag, auto i_tag) {
            constexpr int pr = decltype(pr_tag)::value, i = decltype(i_tag)::value, sb = DEEP ? i : 0;
            constexpr int t0 = 2 * pr, t1 = (2 * pr + 1 < TAPS) ? 2 * pr + 1 : 2 * pr;                    // (odd tail: zero weights)
            constexpr int o0 = ((t0 / KX) * HX + (t0 % KX)) * QPITCH, o1 = ((t1 / KX) * HX + (t1 % KX)) * QPITCH;
            const char* ra = Qa + q_off[i] + 16 * hh;
            const char* rl = Ql + q_off[i] + 16 * hh;
            const i32x4 a0 = *(const i32x4*)(ra + o0), a1 = *(const i32x4*)(ra + o1);
            const i32x4 l0 = *(const i32x4*)(rl + o0), l1 = *(const i32x4*)(rl + o1);
            qa[sb] = i32x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            ql[sb] = i32x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
            const int so = q_off[i] + (hh ? o1 : o0) + 32;                 // lane half b supplies block b's scale
            sqa[sb] = *(const unsigned char*)(Qa + so);
            sql[sb] = *(const unsigned char*)(Ql + so);
        };
        load_am(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        if constexpr (AM2) load_am(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
        [&]<int... T>(std::integer_sequence<int, T...>) {
            ([&] {
                constexpr int tap = T, ab = AM2 ? (tap & 1) : 0;
                constexpr int slot = tap % NBUF;                           // (TAPS % NBUF == 0: static ring slots)
                constexpr bool pair_end = (tap & 1) == 1 || tap == TAPS - 1;
                constexpr int pr = tap / 2;
                fetch_b(chunk * TAPS + tap + DIST, std::integral_constant<int, (tap + DIST) % NBUF>{});
                if (tap == 0 && chunk + 1 < nchunk) load_halo(chunk + 1);
                if constexpr (AM2) {
                    if constexpr (tap + 1 < TAPS) {
                        load_am(std::integral_constant<int, tap + 1>{}, std::integral_constant<int, 0>{});
                        load_am(std::integral_constant<int, tap + 1>{}, std::integral_constant<int, 1>{});
                    }
                } else {
                    load_am(std::integral_constant<int, tap>{}, std::integral_constant<int, 1>{});
                }
                if constexpr (DEEP) {
                    // the pair's fp8 fragments a tap ahead (at the pair's first tap) where the registers allow it: the 1x5 / 5x1 instances
                    // (41.7 -> 37.7 us on the GRU's q conv); the 3x3 instance spills with the longer live range (85 -> 125 us)
                    constexpr bool QEARLY = TAPS <= 5 && TY == 4;     // (the 8 x 16-pixel 1x5 / 5x1 instances spill with it)
                    if constexpr (QEARLY ? (tap & 1) == 0 : pair_end)
                        [&]<int... I>(std::integer_sequence<int, I...>) {
                            (load_q(std::integral_constant<int, pr>{}, std::integral_constant<int, I>{}), ...);
                        }(std::make_integer_sequence<int, TM>{});
                }
                MX_FENCE;
                // DEEP: the next chunk's halo (requested at tap 0) is converted and written to the other buffer ONE thread-row per half
                // tap over the last RH half taps, its ~80 vector / cross-lane / LDS instructions in the issue shadow of this half
                // tap's MFMAs -- as one block after the last MFMA it cost ~3 k of a 9.4 k-cycle chunk (stamps, first version)
                constexpr int NHS = 2 * TAPS;                              // half-tap slots per chunk
                constexpr bool ILX = DEEP && NHS >= RH;
                constexpr int rowa = 2 * tap - (NHS - RH), rowb = rowa + 1;
                if constexpr (ILX && more && rowa >= 0 && rowa < RH)
                    store_halo_row(smem + ((chunk + 1) & 1) * A_ELEMS, std::integral_constant<int, (rowa >= 0 && rowa < RH) ? rowa : 0>{});
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i] = mma16<16>(am[ab][0][i], bq[slot][0][0], acc[i]);
                if constexpr (ILX && more && rowa >= 0 && rowa < RH) {
#pragma unroll
                    for (int g = 0; g < TM; ++g) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                        __builtin_amdgcn_sched_group_barrier(0x002, 24, 0);     // then a share of the row's vector work
                        __builtin_amdgcn_sched_group_barrier(0x080, 4, 0);      // and of its cross-lane / LDS instructions
                    }
                    MX_FENCE;
                }
                if constexpr (ILX && more && rowb >= 0 && rowb < RH)
                    store_halo_row(smem + ((chunk + 1) & 1) * A_ELEMS, std::integral_constant<int, (rowb >= 0 && rowb < RH) ? rowb : 0>{});
                if constexpr (!AM2) {
                    MX_FENCE;
                    if constexpr (tap + 1 < TAPS) load_am(std::integral_constant<int, tap + 1>{}, std::integral_constant<int, 0>{});
                    MX_FENCE;
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i] = mma16<16>(am[ab][1][i], bq[slot][0][1], acc[i]);
                if constexpr (pair_end) {
                    if constexpr (DEEP) {
                        // term-major: consecutive MFMAs never chain on one accumulator (a dependent scaled MFMA waits out the whole
                        // 16-pass latency of its predecessor: the tile-major first version gained nothing over bf16x3)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
                            acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ql[i], wq[0][0], acc[i], 0, 0, 0, sql[i], 0, wsc[0][0]);
#pragma unroll
                        for (int i = 0; i < TM; ++i)
                            acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(qa[i], wq[0][1], acc[i], 0, 0, 0, sqa[i], 0, wsc[0][1]);
                    } else {
                        [&]<int... I>(std::integer_sequence<int, I...>) {
                            ([&] {
                                constexpr int i = I;
                                load_q(std::integral_constant<int, pr>{}, std::integral_constant<int, i>{});
                                acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ql[0], wq[0][0], acc[i], 0, 0, 0, sql[0], 0, wsc[0][0]);
                                acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(qa[0], wq[0][1], acc[i], 0, 0, 0, sqa[0], 0, wsc[0][1]);
                            }(), ...);
                        }(std::make_integer_sequence<int, TM>{});
                    }
                    // the NEXT pair's fp8 weights into the register set these MFMAs just read
                    fetch_mx(chunk * NPAIR + pr + 1, std::integral_constant<int, 0>{});
                }
                MX_FENCE;
            }(), ...);
        }(std::make_integer_sequence<int, TAPS>{});
        // next chunk's halo -> the other buffer (its loads had the whole chunk to land)
        if (stamps && chunk < 12) stamps[16 + chunk] = __builtin_amdgcn_s_memtime();        // (probe: MFMAs of the chunk issued)
        if (more) {
            if constexpr (!(DEEP && 2 * TAPS >= RH)) store_halo(smem + ((chunk + 1) & 1) * A_ELEMS);
            __syncthreads();
        }
        if (stamps && chunk < 12) stamps[1 + chunk] = __builtin_amdgcn_s_memtime();
    };
    auto run_phase = [&](int chunk, auto more_tag) {
        if constexpr (MX) run_chunk_mx(chunk, more_tag);
        else run_chunk(chunk, more_tag);
    };
    for (int chunk = 0; chunk + 1 < nchunk; ++chunk) run_phase(chunk, std::true_type{});
    run_phase(nchunk - 1, std::false_type{});
    __syncthreads();                                     // halo buffers are dead: reuse them as epilogue staging

    const HaloRowMap<TY, TX> rowmap{img0, p.n_img, y0, x0, p.ho, p.wo};
    if (p.epi == WOFT_EPI_FLOWHEAD) {
        // (f16mx8: the epilogue's small second conv keeps the split-bf16 arithmetic; its W2 fragments are packed for it)
        constexpr int ET = MX ? 3 : TERMS, ENP = (ET == 3) ? 2 : 1;
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
        const __bf16* wf = (const __bf16*)p.e0 + (int64_t)(ncol / 32) * (2 * ENP * 512) + lane * 8;
        bf16x8 w2[2][ENP];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int pl = 0; pl < ENP; ++pl) w2[s2][pl] = *(const bf16x8*)(wf + (s2 * ENP + pl) * 512);
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
                    const bf16x4 hi = cvt16<ET>(yv);
                    bf16x4 lo = hi;
                    if constexpr (ENP == 2) lo = __builtin_convertvector(yv - widen_bf16x4(hi), bf16x4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { ah[s2][4 * q + e] = hi[e]; al[s2][4 * q + e] = lo[e]; }
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[i][r] = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                if (ENP == 2) {
                    sacc[i] = mma16<ET>(al[s2], w2[s2][0], sacc[i]);
                    sacc[i] = mma16<ET>(ah[s2], w2[s2][ENP - 1], sacc[i]);
                }
                sacc[i] = mma16<ET>(ah[s2], w2[s2][0], sacc[i]);
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
        const uint64_t dbase = (uint64_t)(uintptr_t)(p.out + (int64_t)n_tile * M * p.ldo);   // wave-uniform; said explicitly (an
        woft::GPtr dst{((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(dbase >> 32)) << 32) |   // "+s" asm
                       (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)dbase)};   // operand here: 'illegal VGPR to SGPR copy')
        for (int idx = tid; idx < G::BM * (SLD / 4); idx += 256) {
            const int row = idx / (SLD / 4), c4 = (idx - row * (SLD / 4)) * 4;
            const int wmr = row / WROWS, lr = row - wmr * WROWS;
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w_ = 0; w_ < WN; ++w_)
                t += *(const f32x4*)((const float*)smem + (wmr * WN + w_) * TM * woft::STAGE_FLOATS + lr * SLD + c4);
            const int64_t m = rowmap(row);
            if (m >= 0) dst.st4(m * p.ldo + c4, t);
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

}  // namespace
