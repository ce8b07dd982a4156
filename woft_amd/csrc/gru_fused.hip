// One SepConvGRU half step (update.py:45-60) in ONE launch:
//   z = sigmoid(conv_z([h | m]) + gz_z),  r = sigmoid(conv_r([h | m]) + gz_r),
//   q = tanh(conv_q([r h | m]) + gq),      h' = (1 - z) h + z q
// (m = motion features; the context features' share of the three convolutions is the per-pixel bias maps gz / gq, see
// RaftEngine).  As two launches (woft_conv2d with EPI_GRU_ZR, then EPI_GRU_Q) every half step writes z and r h (33 MB at 1/8
// of 1080p), reads them back, and pays a second kernel boundary; here a workgroup owns an 8 x 16-pixel tile for the whole half
// step:
//   phase 1   z on the tile's 128 pixels and r on the 160 / 192 pixels of the q conv's 1x5 / 5x1 halo (the r h halo is
//             RECOMPUTED: +8 / +17 % MFMAs), same GEMM formulation, weight stream and product order as conv_regb_kernel
//             (weights in MFMA-fragment order global -> registers; input halo fp32 -> bf16 hi / lo in LDS, double buffered);
//   epilogue 1  r h -> LDS as the first four 32-channel chunks of the q conv's input halo (all four resident: 102 / 123 KB,
//             which is why this kernel runs ONE workgroup per CU -- of 8 waves, two per SIMD), z -> registers;
//   phase 2   q on the 128 pixels: chunks 0-3 from the resident r h, chunks 4-7 (motion) streamed through one halo buffer;
//   epilogue 2  the blend, with z still in the registers of the wave that produced it.
// Wave (g, w) owns the 32-column band w of z, r AND q for the row tiles of group g (rows {0, 1, border} / {2, 3(, border)}), so
// z and q meet in the same lanes and nothing but r h moves between the phases.  (First version: 4 waves, one per SIMD, every
// wave all rows -- 106 / 113 us against 106 / 107 us for the two launches: a wave alone on its SIMD exposes every LDS / L2
// wait; two waves per SIMD interleave them.)  Every value is computed by the same operations in the same order as the two-launch path: bit-identical.
#include <type_traits>

#include "conv_common.h"
#include "halo_map.h"

namespace {

using woft::BK;

template <int KY, int KX, int TERMS>
__global__ __launch_bounds__(512, 2) void gru_halfstep_kernel(const woft_conv_params pz, const woft_conv_params pq) {
    constexpr int NWAVES = 8, NT = 512;
    constexpr int TY = 8, TX = 16;
    constexpr int PY = KY / 2, PX = KX / 2, TAPS = KY * KX;
    static_assert(TAPS == 5 && (KY == 1 || KX == 1), "1x5 / 5x1 half steps");
    constexpr int QY = TY + 2 * PY, QX = TX + 2 * PX, QROWS = QY * QX;     // the q conv's input halo: 8 x 20 / 12 x 16
    constexpr int IY = TY + 4 * PY, IX = TX + 4 * PX, IROWS = IY * IX;     // the z|r conv's input halo: 8 x 24 / 16 x 16
    constexpr int NRT = QROWS / 32;                                        // MFMA row tiles of r in all (5 / 6); of z: 4
    constexpr int NR = 3, NZ = 2;                                          // ... per wave: r {2g, 2g+1, 4+g}, z {2g, 2g+1}
    constexpr int NP = (TERMS == 3) ? 2 : 1;
    constexpr int RH1 = (IROWS + 63) / 64, RH2 = (QROWS + 63) / 64;        // float4 halo rows per thread (64 rows per pass)
    static_assert(QROWS % 32 == 0 && IROWS % 64 == 0, "halo sizes are whole row tiles");
    constexpr int A1_PLANE = IROWS * LDB, A1_ELEMS = NP * A1_PLANE;        // one phase-1 halo buffer (bf16 elements)
    constexpr int R_PLANE = QROWS * LDB, R_CHUNK = NP * R_PLANE;           // one 32-channel chunk of the q conv's halo
    constexpr int STAGE_ELEMS = 2 * NWAVES * woft::STAGE_FLOATS;           // epilogue staging: one tile per wave
    constexpr int TAIL_ELEMS = R_CHUNK > STAGE_ELEMS ? R_CHUNK : STAGE_ELEMS;
    constexpr int SMEM_ELEMS = 4 * R_CHUNK + TAIL_ELEMS;
    static_assert(2 * A1_ELEMS <= 4 * R_CHUNK, "the phase-1 halo buffers alias the r.h chunks");
    constexpr int STEP_ELEMS = NP * 2 * 64 * 8;                            // fragment elements of one K step of a band
    __shared__ __attribute__((aligned(16))) __bf16 smem[SMEM_ELEMS];
    __bf16* const rhbuf = smem;                                            // [4 chunks][NP][QROWS][LDB]
    __bf16* const mbuf = smem + 4 * R_CHUNK;                               // phase 2: motion halo of the current chunk
    float* const stage_all = (float*)(smem + 4 * R_CHUNK);                 // epilogues: [4 waves][STAGE_FLOATS]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave_id & 3, grp = wave_id >> 2;                      // column band, row group
    const bool has3 = NRT > 4 + grp;                                       // this group has a border tile (1x5: group 0 only)
    auto tile_of = [&](int li) { return li < 2 ? 2 * grp + li : 4 + grp; };
    const int r32 = lane & 31, hh = lane >> 5;
    const int v = tid & 7, r0 = tid >> 3;
    const int rr = lane >> 3, c4 = (lane & 7) * 4;                         // staged layout: row rr + 8 ps, 4 channels at c4

    const int txn = (pz.wo + TX - 1) / TX;
    const int y0 = ((int)blockIdx.x / txn) * TY, x0 = ((int)blockIdx.x % txn) * TX;
    const int H = pz.h, W = pz.w;
    constexpr int NCHUNK = 8;                                              // [h 4 | motion 4] and [r h 4 | motion 4]
    // developer probe (tools/gru_probe.py): s_memtime stamps of wave 0 / lane 0 -> pq.in_rstd (unused otherwise)
    unsigned long long* stamps = (pq.in_mean == (const float*)1 && tid == 0) ? (unsigned long long*)pq.in_rstd + (size_t)blockIdx.x * 16 : nullptr;
    int n_stamp = 0;
    auto stamp = [&]() { if (stamps) stamps[n_stamp++] = __builtin_amdgcn_s_memtime(); };
    stamp();

    // row r of the phase-1 GEMM -> output position (oy, ox) relative to the tile: rows 0..127 the tile's pixels (same
    // row <-> pixel permutation as the other pixel-tile kernels), rows 128.. the border pixels of the q conv's halo
    auto row_pos = [&](int row, int& oy, int& ox) {
        if (row < 128) {
            const int pl = (row & ~31) + tile_row_perm(row & 31);
            oy = pl / TX;
            ox = pl % TX;
        } else if (PX > 0) {                                               // 1x5: 8 rows x {-2, -1, 16, 17}
            const int b = row - 128, j = b & 3;
            oy = b >> 2;
            ox = j < 2 ? j - 2 : TX - 2 + j;
        } else {                                                           // 5x1: {-2, -1, 8, 9} x 16 columns
            const int b = row - 128;
            const int pl = (b & ~31) + tile_row_perm(b & 31), br = pl / TX;
            oy = br < 2 ? br - 2 : TY - 2 + br;
            ox = pl % TX;
        }
    };
    auto gpix = [&](int oy, int ox) -> int {                               // global pixel index or -1
        const int gy = y0 + oy, gx = x0 + ox;
        return (gy >= 0 && gy < H && gx >= 0 && gx < W) ? gy * W + gx : -1;
    };

    // ---- phase 1 operands -----------------------------------------------------------------------------------------------
    int hpix1[RH1];
    bool hok1[RH1];
#pragma unroll
    for (int j = 0; j < RH1; ++j) {
        const int ht = r0 + 64 * j;
        const int iy = y0 - 2 * PY + ht / IX, ix = x0 - 2 * PX + ht % IX;
        hok1[j] = iy >= 0 && iy < H && ix >= 0 && ix < W;
        hpix1[j] = hok1[j] ? iy * W + ix : 0;
    }
    f32x4 rh[RH1];                                                         // (RH1 >= RH2: reused by phase 2)
    auto load_halo1 = [&](int chunk) {
        const int c0 = chunk * BK;
        const bool second = c0 >= pz.c_split;
        const float* src = (second ? pz.in1 + (c0 - pz.c_split) : pz.in0 + c0) + 4 * v;
        const int cs = second ? pz.cs1 : pz.cs0;
#pragma unroll
        for (int j = 0; j < RH1; ++j) rh[j] = *(const f32x4*)(src + (uint32_t)(hpix1[j] * cs));
    };
    bool hok2[RH2];                                                        // (phase 2's halo rows: set below)
    auto store_row = [&](__bf16* As, int plane, auto j_tag, auto phase_tag) {
        constexpr int j = decltype(j_tag)::value;
        const int ht = r0 + 64 * j;
        if (decltype(phase_tag)::value == 2 && QROWS % 64 != 0 && ht >= QROWS) return;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        bool ok;
        if constexpr (decltype(phase_tag)::value == 1) ok = hok1[j]; else ok = hok2[j];
        const f32x4 val = ok ? rh[j] : zero;
        const bf16x4 hi = cvt16<TERMS>(val);
        *(bf16x4*)(As + ht * LDB + 4 * v) = hi;
        if (NP == 2) *(bf16x4*)(As + plane + ht * LDB + 4 * v) = __builtin_convertvector(val - widen_bf16x4(hi), bf16x4);
    };
    auto store_rows = [&](__bf16* As, int plane, auto n_tag, auto phase_tag) {
        [&]<int... J>(std::integer_sequence<int, J...>) {
            (store_row(As, plane, std::integral_constant<int, J>{}, phase_tag), ...);
        }(std::make_integer_sequence<int, decltype(n_tag)::value>{});
    };

    int a1_off[NR];                                                        // A rows of this lane inside a phase-1 halo buffer
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        int oy, ox;
        row_pos((has3 || i < 2 ? tile_of(i) : 0) * 32 + r32, oy, ox);
        a1_off[i] = ((oy + PY) * IX + (ox + PX)) * LDB + hh * 8;
    }
    const __bf16* wz = (const __bf16*)pz.wgt_frag + (int64_t)wave * (NCHUNK * TAPS) * STEP_ELEMS + lane * 8;
    const __bf16* wr = (const __bf16*)pz.wgt_frag + (int64_t)(4 + wave) * (NCHUNK * TAPS) * STEP_ELEMS + lane * 8;
    const __bf16* wq = (const __bf16*)pq.wgt_frag + (int64_t)wave * (NCHUNK * TAPS) * STEP_ELEMS + lane * 8;

    bf16x8 bq[2][2][NP][2];                                                // [slot][band z, r][plane][k half]
    auto fetch1 = [&](int step, auto slot_tag) {
        constexpr int slot = decltype(slot_tag)::value;
        const int s = step < NCHUNK * TAPS ? step : NCHUNK * TAPS - 1;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bq[slot][0][pl][s2] = *(const bf16x8*)(wz + (int64_t)s * STEP_ELEMS + (pl * 2 + s2) * 512);
                bq[slot][1][pl][s2] = *(const bf16x8*)(wr + (int64_t)s * STEP_ELEMS + (pl * 2 + s2) * 512);
            }
    };

    f32x16 accz[NZ], accr[NR];
#pragma unroll
    for (int i = 0; i < NZ; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) accz[i][r] = 0.f;
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) accr[i][r] = 0.f;

    load_halo1(0);
    fetch1(0, std::integral_constant<int, 0>{});
    store_rows(smem, A1_PLANE, std::integral_constant<int, RH1>{}, std::integral_constant<int, 1>{});
    __syncthreads();
    stamp();                                             // [1] prologue done

    // one 32-channel chunk = 5 taps x 2 k halves; PH = chunk parity (the weight slot of a tap alternates and 5 is odd)
    auto chunk1 = [&](int chunk, auto ph_tag, auto more_tag) {
        constexpr int PH = decltype(ph_tag)::value;
        constexpr bool more = decltype(more_tag)::value;
        const __bf16* As = smem + (chunk & 1) * A1_ELEMS;
        bf16x8 aq[2][NR][NP];
        auto load_a = [&](auto g_tag) {
            constexpr int g = decltype(g_tag)::value, tap = g >> 1, s2 = g & 1;
            constexpr int ky = tap / KX, kx = tap - ky * KX;
#pragma unroll
            for (int i = 0; i < NR; ++i)
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
                    aq[g & 1][i][pl] = *(const bf16x8*)(As + a1_off[i] + pl * A1_PLANE + (ky * IX + kx) * LDB + s2 * 16);
        };
        load_a(std::integral_constant<int, 0>{});
        [&]<int... G>(std::integer_sequence<int, G...>) {
            ([&] {
                constexpr int g = G, tap = g >> 1, s2 = g & 1, slot = (PH * TAPS + tap) & 1, as = g & 1;
                if constexpr (s2 == 0) {
                    fetch1(chunk * TAPS + tap + 1, std::integral_constant<int, slot ^ 1>{});
                    if (tap == 0 && more) load_halo1(chunk + 1);
                }
                if constexpr (g + 1 < 2 * TAPS) load_a(std::integral_constant<int, g + 1>{});
                // the next chunk's halo rows (requested at the first tap) are converted and stored one per group over the
                // last groups of the chunk: this wave's conversion sits beside the other wave's MFMAs on the SIMD
                if constexpr (more && g >= 2 * TAPS - RH1)
                    store_row(smem + ((chunk + 1) & 1) * A1_ELEMS, A1_PLANE, std::integral_constant<int, g - (2 * TAPS - RH1)>{},
                              std::integral_constant<int, 1>{});
                __builtin_amdgcn_sched_barrier(0);
                if (NP == 2) {
#pragma unroll
                    for (int i = 0; i < NZ; ++i) {
                        accz[i] = mma16<TERMS>(aq[as][i][NP - 1], bq[slot][0][0][s2], accz[i]);
                        accr[i] = mma16<TERMS>(aq[as][i][NP - 1], bq[slot][1][0][s2], accr[i]);
                    }
#pragma unroll
                    for (int i = 0; i < NZ; ++i) {
                        accz[i] = mma16<TERMS>(aq[as][i][0], bq[slot][0][NP - 1][s2], accz[i]);
                        accr[i] = mma16<TERMS>(aq[as][i][0], bq[slot][1][NP - 1][s2], accr[i]);
                    }
                }
#pragma unroll
                for (int i = 0; i < NZ; ++i) {
                    accz[i] = mma16<TERMS>(aq[as][i][0], bq[slot][0][0][s2], accz[i]);
                    accr[i] = mma16<TERMS>(aq[as][i][0], bq[slot][1][0][s2], accr[i]);
                }
                if (has3) {                              // the group's border tile: r only (wave-uniform branch)
                    if (NP == 2) {
                        accr[2] = mma16<TERMS>(aq[as][2][NP - 1], bq[slot][1][0][s2], accr[2]);
                        accr[2] = mma16<TERMS>(aq[as][2][0], bq[slot][1][NP - 1][s2], accr[2]);
                    }
                    accr[2] = mma16<TERMS>(aq[as][2][0], bq[slot][1][0][s2], accr[2]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }(), ...);
        }(std::make_integer_sequence<int, 2 * TAPS>{});
        if (more) __syncthreads();
    };
    for (int chunk = 0; chunk + 2 < NCHUNK; chunk += 2) {
        chunk1(chunk, std::integral_constant<int, 0>{}, std::true_type{});
        chunk1(chunk + 1, std::integral_constant<int, 1>{}, std::true_type{});
    }
    chunk1(NCHUNK - 2, std::integral_constant<int, 0>{}, std::true_type{});
    chunk1(NCHUNK - 1, std::integral_constant<int, 1>{}, std::false_type{});
    __syncthreads();                                     // the halo buffers are dead: r.h chunks are written over them
    stamp();                                             // [2] phase 1 main loop done

    // ---- epilogue 1: r.h -> LDS (spatial order of the q conv's halo), z -> registers ----------------------------------------
    float* stage = stage_all + wave_id * woft::STAGE_FLOATS;
    const int ncol = wave * 32 + c4;                     // this lane's 4 channels inside the 128
    const float* hsrc = pz.e0;                           // previous state h [pixel][lde0]
    const float* gz = pz.bias_map;                       // [pixel][ld]: z bias 0..127 | r bias 128..255
    f32x4 zreg[NZ][4];
    auto stage_tile = [&](const f32x16& a) {
#pragma unroll
        for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * hh) * woft::STAGE_LD + r32] = a[r];
        __builtin_amdgcn_wave_barrier();
    };
    // operands of ALL this wave's tiles first (one memory round trip instead of one per tile): r bias + h of the r tiles,
    // z bias of the z tiles
    f32x4 rb4[NR][4], rh4[NR][4], zb4[NZ][4];
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            int oy, ox;
            row_pos((has3 || i < 2 ? tile_of(i) : 0) * 32 + rr + 8 * ps, oy, ox);
            const int m = gpix(oy, ox);
            const int64_t mm = m >= 0 ? m : 0;
            rb4[i][ps] = *(const f32x4*)(gz + mm * pz.ld_bias_map + 128 + ncol);
            rh4[i][ps] = *(const f32x4*)(hsrc + mm * pz.lde0 + ncol);
            if (i < NZ) zb4[i][ps] = *(const f32x4*)(gz + mm * pz.ld_bias_map + ncol);
        }
    [&]<int... I>(std::integer_sequence<int, I...>) {
        ([&] {
            constexpr int i = I;
            if (i == 2 && !has3) return;
            stage_tile(accr[i]);
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const f32x4 vs = *(const f32x4*)(stage + (rr + 8 * ps) * woft::STAGE_LD + c4);
                int oy, ox;
                row_pos(tile_of(i) * 32 + rr + 8 * ps, oy, ox);
                const bool ok = gpix(oy, ox) >= 0;
                f32x4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = sigmoid_fast_(pz.alpha * vs[e] + rb4[i][ps][e]);
                    y[e] = ok ? t * rh4[i][ps][e] : 0.f;
                }
                __bf16* dst = rhbuf + wave * R_CHUNK + ((oy + PY) * QX + (ox + PX)) * LDB + c4;
                const bf16x4 hi = cvt16<TERMS>(y);
                *(bf16x4*)dst = hi;
                if (NP == 2) *(bf16x4*)(dst + R_PLANE) = __builtin_convertvector(y - widen_bf16x4(hi), bf16x4);
            }
            __builtin_amdgcn_wave_barrier();
        }(), ...);
    }(std::make_integer_sequence<int, NR>{});
    [&]<int... I>(std::integer_sequence<int, I...>) {
        ([&] {
            constexpr int i = I;
            stage_tile(accz[i]);
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const f32x4 vs = *(const f32x4*)(stage + (rr + 8 * ps) * woft::STAGE_LD + c4);
#pragma unroll
                for (int e = 0; e < 4; ++e) zreg[i][ps][e] = sigmoid_fast_(pz.alpha * vs[e] + zb4[i][ps][e]);
            }
            __builtin_amdgcn_wave_barrier();
        }(), ...);
    }(std::make_integer_sequence<int, NZ>{});

    // ---- phase 2: q = conv([r.h | motion]) on the tile's 128 pixels ------------------------------------------------------
    int hpix2[RH2];
#pragma unroll
    for (int j = 0; j < RH2; ++j) {
        const int ht = r0 + 64 * j;
        const int iy = y0 - PY + ht / QX, ix = x0 - PX + ht % QX;
        hok2[j] = ht < QROWS && iy >= 0 && iy < H && ix >= 0 && ix < W;
        hpix2[j] = hok2[j] ? iy * W + ix : 0;
    }
    auto load_halo2 = [&](int chunk) {                   // chunks 4..7: motion channels 32 (chunk - 4) .. of pq.in1
        const float* src = pq.in1 + (chunk * BK - pq.c_split) + 4 * v;
#pragma unroll
        for (int j = 0; j < RH2; ++j) rh[j] = *(const f32x4*)(src + (uint32_t)(hpix2[j] * pq.cs1));
    };
    int a2_off[NZ];
#pragma unroll
    for (int i = 0; i < NZ; ++i) {
        int oy, ox;
        row_pos(tile_of(i) * 32 + r32, oy, ox);
        a2_off[i] = (oy * QX + ox) * LDB + hh * 8;
    }
    bf16x8 bq2[2][NP][2];
    auto fetch2 = [&](int step, auto slot_tag) {
        constexpr int slot = decltype(slot_tag)::value;
        const int s = step < NCHUNK * TAPS ? step : NCHUNK * TAPS - 1;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) bq2[slot][pl][s2] = *(const bf16x8*)(wq + (int64_t)s * STEP_ELEMS + (pl * 2 + s2) * 512);
    };
    f32x16 accq[NZ];
#pragma unroll
    for (int i = 0; i < NZ; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) accq[i][r] = 0.f;
    fetch2(0, std::integral_constant<int, 0>{});
    stamp();                                             // [3] epilogue 1 done (this wave)
    __syncthreads();                                     // every wave's r.h chunk is in place (and the staging area is free)
    stamp();                                             // [4] ... all waves

    auto chunk2 = [&](int chunk, auto ph_tag) {
        constexpr int PH = decltype(ph_tag)::value;
        const __bf16* As = chunk < 4 ? rhbuf + chunk * R_CHUNK : mbuf;
        bf16x8 aq[2][NZ][NP];
        auto load_a = [&](auto g_tag) {
            constexpr int g = decltype(g_tag)::value, tap = g >> 1, s2 = g & 1;
            constexpr int ky = tap / KX, kx = tap - ky * KX;
#pragma unroll
            for (int i = 0; i < NZ; ++i)
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
                    aq[g & 1][i][pl] = *(const bf16x8*)(As + a2_off[i] + pl * R_PLANE + (ky * QX + kx) * LDB + s2 * 16);
        };
        load_a(std::integral_constant<int, 0>{});
        [&]<int... G>(std::integer_sequence<int, G...>) {
            ([&] {
                constexpr int g = G, tap = g >> 1, s2 = g & 1, slot = (PH * TAPS + tap) & 1, as = g & 1;
                if constexpr (s2 == 0) {
                    fetch2(chunk * TAPS + tap + 1, std::integral_constant<int, slot ^ 1>{});
                    if (tap == 0 && chunk >= 3 && chunk + 1 < NCHUNK) load_halo2(chunk + 1);   // next motion chunk -> registers
                }
                if constexpr (g + 1 < 2 * TAPS) load_a(std::integral_constant<int, g + 1>{});
                __builtin_amdgcn_sched_barrier(0);
                if (NP == 2) {
#pragma unroll
                    for (int i = 0; i < NZ; ++i) accq[i] = mma16<TERMS>(aq[as][i][NP - 1], bq2[slot][0][s2], accq[i]);
#pragma unroll
                    for (int i = 0; i < NZ; ++i) accq[i] = mma16<TERMS>(aq[as][i][0], bq2[slot][NP - 1][s2], accq[i]);
                }
#pragma unroll
                for (int i = 0; i < NZ; ++i) accq[i] = mma16<TERMS>(aq[as][i][0], bq2[slot][0][s2], accq[i]);
                __builtin_amdgcn_sched_barrier(0);
            }(), ...);
        }(std::make_integer_sequence<int, 2 * TAPS>{});
        if (chunk >= 3 && chunk + 1 < NCHUNK) {          // the motion buffer: free once every wave is done with this chunk
            __syncthreads();
            store_rows(mbuf, R_PLANE, std::integral_constant<int, RH2>{}, std::integral_constant<int, 2>{});
            __syncthreads();
        }
    };
    for (int chunk = 0; chunk < NCHUNK; chunk += 2) {
        chunk2(chunk, std::integral_constant<int, 0>{});
        chunk2(chunk + 1, std::integral_constant<int, 1>{});
    }
    __syncthreads();                                     // motion buffer dead: it is the staging area again
    stamp();                                             // [5] phase 2 main loop done

    // ---- epilogue 2: h' = (1 - z) h + z tanh(q + gq) ----------------------------------------------------------------------
    const float* gq = pq.bias_map;
    f32x4 qb4[NZ][4], qh4[NZ][4];
    int mrow[NZ][4];
#pragma unroll
    for (int i = 0; i < NZ; ++i)
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            int oy, ox;
            row_pos(tile_of(i) * 32 + rr + 8 * ps, oy, ox);
            mrow[i][ps] = gpix(oy, ox);
            const int64_t mm = mrow[i][ps] >= 0 ? mrow[i][ps] : 0;
            qb4[i][ps] = *(const f32x4*)(gq + mm * pq.ld_bias_map + ncol);
            qh4[i][ps] = *(const f32x4*)(pq.e0 + mm * pq.lde0 + ncol);
        }
    [&]<int... I>(std::integer_sequence<int, I...>) {
        ([&] {
            constexpr int i = I;
            stage_tile(accq[i]);
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const f32x4 vs = *(const f32x4*)(stage + (rr + 8 * ps) * woft::STAGE_LD + c4);
                f32x4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float zz = zreg[i][ps][e];
                    y[e] = (1.f - zz) * qh4[i][ps][e] + zz * tanh_fast_(pq.alpha * vs[e] + qb4[i][ps][e]);
                }
                if (mrow[i][ps] >= 0) *(f32x4*)(pq.out + (int64_t)mrow[i][ps] * pq.ldo + pq.co_off + ncol) = y;
            }
            __builtin_amdgcn_wave_barrier();
        }(), ...);
    }(std::make_integer_sequence<int, NZ>{});
    stamp();                                             // [6] epilogue 2 done
    if (stamps) stamps[15] = __builtin_amdgcn_s_memrealtime();
}

}  // namespace

// zr: the z|r conv as woft_conv2d would take it (EPI_GRU_ZR: in0 = h, in1 = motion, c_split 128, cin_pad 256, cout 256 with
// z in columns 0..127, wgt_frag, bias_map, e0 = h); q: the q conv (EPI_GRU_Q: in1 = motion, wgt_frag, bias_map, e0 = h, out =
// the new state; its in0 -- r.h -- is not read).  One image; 1x5 or 5x1; split-bf16 / fp16 precisions.
extern "C" int woft_gru_halfstep(const woft_conv_params* zr, const woft_conv_params* q, void* stream) {
    if (zr == nullptr || q == nullptr) return WOFT_EINVAL;
    const woft_conv_params& a = *zr;
    const woft_conv_params& b = *q;
    if (a.precision < 1 || a.precision > 3 || b.precision != a.precision) return WOFT_EINVAL;
    if (!((a.taps_y == 1 && a.taps_x == 5) || (a.taps_y == 5 && a.taps_x == 1)) || b.taps_y != a.taps_y || b.taps_x != a.taps_x)
        return WOFT_EINVAL;
    if (a.n_img != 1 || b.n_img != 1 || a.h != b.h || a.w != b.w || a.ho != a.h || a.wo != a.w || a.stride != 1 || b.stride != 1)
        return WOFT_EINVAL;
    if (a.cin_pad != 256 || b.cin_pad != 256 || a.c_split != 128 || b.c_split != 128 || a.cout != 256 || b.cout != 128) return WOFT_EINVAL;
    if (a.in0 == nullptr || a.in1 == nullptr || b.in1 == nullptr || a.in1 != b.in1 || a.cs1 != b.cs1 || a.cs0 % 4 != 0 || a.cs1 % 4 != 0)
        return WOFT_EINVAL;
    if (a.wgt_frag == nullptr || b.wgt_frag == nullptr || a.bias_map == nullptr || b.bias_map == nullptr || a.ld_bias_map < 256 ||
        b.ld_bias_map < 128 || a.ld_bias_map % 4 != 0 || b.ld_bias_map % 4 != 0)
        return WOFT_EINVAL;
    if (a.e0 == nullptr || b.e0 != a.e0 || a.lde0 % 4 != 0 || b.lde0 != a.lde0 || a.e0 != a.in0 || b.out == nullptr || b.ldo % 4 != 0 ||
        b.co_off % 4 != 0 || a.flat || b.flat || a.in_norm || b.in_norm || a.stat_sum != nullptr || b.stat_sum != nullptr)
        return WOFT_EINVAL;
    if (a.in_fmt != 0 || b.in_fmt != 0 || a.out_fmt != 0 || b.out_fmt != 0 || b.out1 != nullptr) return WOFT_EINVAL;   // (fp32 activations only)
    // The new state must not overwrite the old one: other workgroups still read h on their tiles' 1x5 / 5x1 halos (and z|r's e0)
    // while this one stores -- an in-place call would race silently
    {
        const int64_t px = (int64_t)a.h * a.w;
        const char* h0 = (const char*)a.in0;
        const char* h1 = h0 + (size_t)(px * a.cs0) * sizeof(float);
        const char* o0 = (const char*)(b.out + b.co_off);
        const char* o1 = o0 + (size_t)((px - 1) * b.ldo + b.cout) * sizeof(float);
        if (o0 < h1 && h0 < o1) return WOFT_EINVAL;
    }
    const int64_t cs_max = a.cs1 > a.cs0 ? a.cs1 : a.cs0;
    if ((int64_t)a.h * a.w * cs_max >= (1ll << 31)) return WOFT_EINVAL;                     // 32-bit element offsets
    dim3 grid((unsigned)(((a.h + 7) / 8) * ((a.w + 15) / 16)));
    hipStream_t s = (hipStream_t)stream;
#define GRU(KY, KX, T) hipLaunchKernelGGL((gru_halfstep_kernel<KY, KX, T>), grid, dim3(512), 0, s, a, b)
#define GRU_T(KY, KX) \
    if (a.precision == 1) GRU(KY, KX, 3); else if (a.precision == 3) GRU(KY, KX, 16); else GRU(KY, KX, 1)
    if (a.taps_x == 5) { GRU_T(1, 5); } else { GRU_T(5, 1); }
#undef GRU_T
#undef GRU
    return woft_launch_status();
}
