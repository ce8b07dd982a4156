// Pieces shared by the fp32-MFMA and the split-bf16-MFMA implicit-GEMM kernels: the gathered,
// zero-filled A-operand loader (NHWC, two-source channel concat, "flat" tiny-Cin packing) and the
// fused epilogue.  See conv.hip for the GEMM formulation.
#pragma once
#include "common.h"
#include "halo_map.h"
#include <utility>
#include <type_traits>

namespace woft {

constexpr int BK = 32;

// Workgroup id -> (M tile, N tile).  The dispatcher places consecutive workgroups on consecutive
// XCDs (private 4 MiB L2 each), so (1) ids are first re-dealt so that each XCD owns a contiguous
// range (bijective for any grid size), then (2) walked N-fastest inside column panels of at most 8 N
// tiles: the 64 workgroups resident on an XCD cover ~8 x 8 tiles and share their A and B panels
// through that XCD's L2 (the 1080p volume GEMM re-read fmap1 254 times from beyond L2 otherwise).
// Placement only affects speed, never results.
__device__ __forceinline__ void tile_of_block(int bid, int mt, int nt, int& m_tile, int& n_tile) {
    const int nwg = mt * nt;
    const int q = nwg / 8, rr = nwg % 8;
    const int xcd = bid % 8, idx = bid / 8;
    const int id = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
    constexpr int PANEL = 8;
    const int per_panel = mt * PANEL;
    const int panel = id / per_panel;
    const int n_first = panel * PANEL;
    const int width = (nt - n_first < PANEL) ? (nt - n_first) : PANEL;
    const int r = id - panel * per_panel;
    m_tile = r / width;
    n_tile = n_first + r % width;
}

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

// Fused epilogue.  Each 32x32 accumulator tile (C/D layout: column = lane & 31,
// row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) is transposed through a per-wave LDS staging
// tile (32 x 36 floats) so that every lane then owns 4 consecutive output channels of one pixel:
// bias / activation / GRU / residual operands and the result move as 16-byte accesses and one wave
// instruction stores 8 rows x 128 B of full cache lines (4 store instructions per tile instead of 16
// dword ones -- the epilogue is store-issue bound otherwise, e.g. 4.2 GB of the 1080p volume).
constexpr int STAGE_LD = 36;
constexpr int STAGE_FLOATS = 32 * STAGE_LD;     // per wave

// RowMap: local row of the workgroup tile -> global output pixel index m, or -1 (no such pixel)
struct LinearRows {
    int64_t m0, M;
    __device__ __forceinline__ int64_t operator()(int row) const { const int64_t m = m0 + row; return m < M ? m : -1; }
};

// The epilogue's parameters are read ONCE into registers and laundered through an empty asm so that they stay there.
// (The first version read them through `p.` inside the (tile, pass) loops; the compiler re-materialised them from the
// kernel-argument segment each time -- s_load_dword + s_waitcnt lgkmcnt(0), ~330 scalar-cache round trips per wave.
// In-kernel stamps (tools/regb_probe.py) showed the ReLU epilogue of a 128 x 128 tile taking 29-42 k cycles: 15-20 us of
// a 68 us GRU-gate launch, as much as the whole main loop of the plain-bf16 mode.)
template <class T>
__device__ __forceinline__ T keep_sgpr(T x) {
    asm volatile("" : "+s"(x));
    return x;
}
// (a pointer that went through the asm is no longer known to point to global memory and would be accessed with the slower
//  flat_* instructions: the laundered value is carried as an integer and re-typed as an address-space-1 pointer per access)
typedef __attribute__((address_space(1))) f32x4 g_f32x4;
typedef __attribute__((address_space(1))) float g_f32;
struct GPtr {
    uint64_t a;
    __device__ __forceinline__ bool null() const { return a == 0; }
    __device__ __forceinline__ f32x4 ld4(int64_t off) const { return *(const g_f32x4*)(a + 4 * (uint64_t)off); }
    __device__ __forceinline__ void st4(int64_t off, f32x4 v) const { *(g_f32x4*)(a + 4 * (uint64_t)off) = v; }
    __device__ __forceinline__ void st1(int64_t off, float v) const { *(g_f32*)(a + 4 * (uint64_t)off) = v; }
};
__device__ __forceinline__ GPtr keep_gptr(const void* p) { return GPtr{keep_sgpr((uint64_t)(uintptr_t)p)}; }

// Launch-time contract (woft_conv2d validates it): ldo, co_off multiples of 4, no column remap; a ragged last channel
// group (cout % 4 != 0) only with the element-wise kinds LINEAR / RELU / SIGMOID / TANH and without statistics.
//
// Code size matters here: the epilogue runs ONCE per workgroup, so every instruction of it is an instruction-cache miss
// (stamps: the fully unrolled version -- TN x TM x 4 row groups x a 7-way switch, ~1300 instructions on the taken path --
// took 15 k cycles even with its stores disabled).  The row-group body is therefore a REAL loop (`#pragma unroll 1`) over
// the ST accumulator tiles that were transposed into the wave's LDS staging area just before: the loop body is fetched
// once and then runs from the instruction cache.  ST = tiles staged at once (the caller provides ST * STAGE_FLOATS floats
// per wave); stage ALL of a wave's tiles when the LDS allows, one at a time otherwise.
// The InstanceNorm statistics variant (encoder layers) keeps per-column-tile accumulators and the unrolled form.
struct EpiRegs {
    GPtr out, out1, bias, bias_map, e0, e1;
    int64_t ldo;
    int ldo1, lde0, lde1, ld_bias_map, co_off, cout, split, epi;
    float alpha;
    bool no_store;
    bool fast;             // gate functions on the hardware exp2 / rcp (split-bf16 / fp16 precisions), see common.h
};

// One row group (8 rows x 32 columns of a transposed tile) in two steps, so that a caller can issue the operand loads of
// SEVERAL groups before the first store: on gfx9 stores count in vmcnt like loads and may complete out of order with them,
// so waiting for any load that was issued after a store also waits for that store -- a loop of (load operands, wait,
// store) serialises on the full store latency every turn (measured: ~1 k cycles per row group, 17 k per epilogue).
struct EpiOps {
    f32x4 b4, o0, o1;
};
template <int EPI>
__device__ __forceinline__ EpiOps epi_load(const EpiRegs& a, const int64_t m, const int n, const bool nok, const f32x4 bias4) {
    EpiOps o;
    o.b4 = bias4;
    o.o0 = o.o1 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!nok) return o;
    if (!a.bias_map.null()) o.b4 = a.bias_map.ld4(m * a.ld_bias_map + n);
    if (EPI == WOFT_EPI_RELU_RES_RELU || EPI == WOFT_EPI_GRU_Q) o.o0 = a.e0.ld4(m * a.lde0 + n);
    if (EPI == WOFT_EPI_GRU_ZR && n >= a.split) o.o0 = a.e0.ld4(m * a.lde0 + (n - a.split));
    if (EPI == WOFT_EPI_GRU_Q) o.o1 = a.e1.ld4(m * a.lde1 + n);
    return o;
}
// v = this lane's 4 consecutive channels of pixel m; returns y = alpha * v + bias (the value the statistics are taken of)
template <int EPI, bool FAST>
__device__ __forceinline__ f32x4 epi_finish(const EpiRegs& a, const f32x4 v, const EpiOps& o, const int64_t m, const int n,
                                            const bool nok, const int nrag) {
    f32x4 y, ypre;
#pragma unroll
    for (int e = 0; e < 4; ++e) ypre[e] = y[e] = a.alpha * v[e] + o.b4[e];
    bool stored = a.no_store;
    switch (EPI) {                                     // (compile time: the taken path is straight-line and small)
        case WOFT_EPI_RELU:
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = fmaxf(y[e], 0.f);
            break;
        case WOFT_EPI_SIGMOID:
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = sigmoid_t<FAST>(y[e]);
            break;
        case WOFT_EPI_TANH:
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = tanh_t<FAST>(y[e]);
            break;
        case WOFT_EPI_RELU_RES_RELU:
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = fmaxf(o.o0[e] + fmaxf(y[e], 0.f), 0.f);
            break;
        case WOFT_EPI_GRU_ZR:
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = sigmoid_t<FAST>(y[e]);
            if (n >= a.split) {                        // split % 4 == 0 (validated): whole vector is r
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] *= o.o0[e];
                if (!a.no_store) a.out1.st4(m * a.ldo1 + (n - a.split), y);
                stored = true;
            }
            break;
        case WOFT_EPI_GRU_Q:
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = (1.f - o.o1[e]) * o.o0[e] + o.o1[e] * tanh_t<FAST>(y[e]);
            break;
        default: break;
    }
    if (stored) return ypre;
    if (nok) {
        a.out.st4(m * a.ldo + a.co_off + n, y);
    } else {                                           // ragged group (element-wise kinds only)
#pragma unroll
        for (int e = 0; e < 3; ++e)
            if (e < nrag) a.out.st1(m * a.ldo + a.co_off + n + e, y[e]);
    }
    return ypre;
}

// The row groups of `nst` staged tiles (tile t = t_first + u; column tile j = t / TM, row tile i = t % TM) for ONE epilogue
// kind.  A real loop over batches of LP tiles: the body is fetched once and then runs from the instruction cache.  (With a
// run-time `switch (epi)` inside eight unrolled row groups, every group jumped over the unused kinds' sigmoid / tanh
// expansions -- two or three instruction-cache misses per group: 660 cycles per group of pure ALU work, stamped.)
// Inside a batch ALL operand loads are issued before the first store (see epi_load).
template <int EPI, bool FAST, int TM, int WROWS, int LP, typename RowMap>
__device__ __forceinline__ void epi_tiles(const EpiRegs& a, const float* stage, const RowMap& rowmap, const int t_first,
                                          const int nst, const int ncol0, const int wm, const int rr, const int c4,
                                          const bool stats, float (&ssum)[4], float (&ssq)[4]) {
#pragma unroll 1
    for (int u0 = 0; u0 < nst; u0 += LP) {
        int64_t mm[LP][4];
        EpiOps ops[LP][4];
        int nn[LP];
        bool nk[LP];
#pragma unroll
        for (int d = 0; d < LP; ++d) {
            const int t = t_first + u0 + d;
            const int j = t / TM, i = t - j * TM;
            const int n = ncol0 + j * 32;
            nn[d] = n;
            nk[d] = n + 3 < a.cout;
            const bool any = nk[d] || n < a.cout;
            f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
            if (!a.bias.null() && any) bias4 = a.bias.ld4(n);
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                mm[d][ps] = any ? rowmap(wm * WROWS + i * 32 + rr + 8 * ps) : -1;
                ops[d][ps] = epi_load<EPI>(a, mm[d][ps] < 0 ? 0 : mm[d][ps], n, nk[d], bias4);
            }
        }
#pragma unroll
        for (int d = 0; d < LP; ++d) {
            const int nrag = (!nk[d] && nn[d] < a.cout) ? a.cout - nn[d] : 0;
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const f32x4 v = *(const f32x4*)(stage + (u0 + d) * STAGE_FLOATS + (rr + 8 * ps) * STAGE_LD + c4);
                if (mm[d][ps] < 0) continue;
                const f32x4 y = epi_finish<EPI, FAST>(a, v, ops[d][ps], mm[d][ps], nn[d], nk[d], nrag);
                if (stats) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { ssum[e] += y[e]; ssq[e] += y[e] * y[e]; }
                }
            }
        }
    }
}

template <int TM, int WROWS, int LP, typename RowMap>
__device__ __forceinline__ void epi_dispatch(const EpiRegs& a, const float* stage, const RowMap& rowmap, int t_first, int nst,
                                             int ncol0, int wm, int rr, int c4, bool stats, float (&ssum)[4], float (&ssq)[4]) {
#define WOFT_EPI_CASE(E, F) \
    case E: epi_tiles<E, F, TM, WROWS, LP>(a, stage, rowmap, t_first, nst, ncol0, wm, rr, c4, stats, ssum, ssq); break
    if (a.fast) {            // (the kinds with a sigmoid / tanh exist in two versions; the others do not depend on it)
        switch (a.epi) {
            WOFT_EPI_CASE(WOFT_EPI_SIGMOID, true);
            WOFT_EPI_CASE(WOFT_EPI_TANH, true);
            WOFT_EPI_CASE(WOFT_EPI_GRU_ZR, true);
            WOFT_EPI_CASE(WOFT_EPI_GRU_Q, true);
            default: break;
        }
    }
    switch (a.epi) {
        WOFT_EPI_CASE(WOFT_EPI_LINEAR, false);
        WOFT_EPI_CASE(WOFT_EPI_RELU, false);
        WOFT_EPI_CASE(WOFT_EPI_RELU_RES_RELU, false);
        default: break;
    }
    if (!a.fast) {
        switch (a.epi) {
            WOFT_EPI_CASE(WOFT_EPI_SIGMOID, false);
            WOFT_EPI_CASE(WOFT_EPI_TANH, false);
            WOFT_EPI_CASE(WOFT_EPI_GRU_ZR, false);
            WOFT_EPI_CASE(WOFT_EPI_GRU_Q, false);
            default: break;
        }
    }
#undef WOFT_EPI_CASE
}

// ST = accumulator tiles transposed into the wave's LDS staging area at once (the caller provides ST * STAGE_FLOATS floats
// per wave): all of a wave's tiles when the LDS allows, one at a time otherwise.
template <int TM, int TN, int WROWS, int WCOLS, int ST = 1, typename RowMap, class PT = woft_conv_params>
__device__ __forceinline__ void conv_epilogue_t(const PT& p, f32x16 (&acc)[TM][TN], float* stage,
                                                const RowMap& rowmap, int n0, int wm, int wn, int lane, int m_tile,
                                                unsigned long long* dbg = nullptr) {
    constexpr int NT = TM * TN;
    static_assert(NT % ST == 0, "staged tiles must divide the wave's tiles");
    int dbg_i = 0;                                             // developer probe: s_memtime stamps of the phases below
    auto stamp = [&]() { if (dbg) dbg[dbg_i++] = __builtin_amdgcn_s_memtime(); };
    stamp();
    EpiRegs a;
    a.out = keep_gptr(p.out); a.out1 = keep_gptr(p.out1); a.bias = keep_gptr(p.bias); a.bias_map = keep_gptr(p.bias_map);
    a.e0 = keep_gptr(p.e0); a.e1 = keep_gptr(p.e1);
    a.ldo = keep_sgpr(p.ldo); a.ldo1 = keep_sgpr(p.ldo1); a.lde0 = keep_sgpr(p.lde0); a.lde1 = keep_sgpr(p.lde1);
    a.ld_bias_map = keep_sgpr(p.ld_bias_map); a.co_off = keep_sgpr(p.co_off); a.cout = keep_sgpr(p.cout);
    a.split = keep_sgpr(p.split); a.epi = keep_sgpr(p.epi); a.alpha = keep_sgpr(p.alpha);
    a.no_store = p.out_w == -12345;                            // (micro-benchmark ablation, tools/bench_conv.py)
    a.fast = p.precision != 0 && p.out_w != -12346;            // (-12346: developer ablation, library gate functions)
    const GPtr stat_sum = keep_gptr(p.stat_sum), stat_sq = keep_gptr(p.stat_sq);
    const int cout_pad = keep_sgpr(p.cout_pad);
    const bool do_stats = !stat_sum.null();

    const int r32 = lane & 31, hh = lane >> 5;
    const int rr = lane >> 3, c4 = (lane & 7) * 4;
    const int ncol0 = n0 + wn * WCOLS + c4;
    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
    stamp();
    // (compile-time recursion, not `#pragma unroll`: if the optimizer declines to unroll a loop around the seven inlined
    //  kind instantiations, acc[i][j] becomes a run-time index and the accumulators move to scratch memory -- the 9x9
    //  weight-head kernel ran 4x slower that way)
    auto stage_tile = [&](auto t_tag, auto u_tag) {
        constexpr int t = decltype(t_tag)::value, u = decltype(u_tag)::value;
        constexpr int j = t / TM, i = t % TM;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            stage[u * STAGE_FLOATS + ((r & 3) + 8 * (r >> 2) + 4 * hh) * STAGE_LD + r32] = acc[i][j][r];
    };
    if (!do_stats) {
        // tiles in (column tile j, row tile i) order, t = j * TM + i; ST at a time through the staging area
        constexpr int LP = (ST % 2 == 0) ? 2 : 1;
        [&]<int... B>(std::integer_sequence<int, B...>) {
            ([&] {
                constexpr int t0 = B * ST;
                [&]<int... U>(std::integer_sequence<int, U...>) {
                    (stage_tile(std::integral_constant<int, t0 + U>{}, std::integral_constant<int, U>{}), ...);
                }(std::make_integer_sequence<int, ST>{});
                __builtin_amdgcn_wave_barrier();
                stamp();
                epi_dispatch<TM, WROWS, LP>(a, stage, rowmap, t0, ST, ncol0, wm, rr, c4, false, ssum, ssq);
                stamp();
                __builtin_amdgcn_wave_barrier();
            }(), ...);
        }(std::make_integer_sequence<int, NT / ST>{});
        return;
    }
    // ---- with InstanceNorm partial statistics (encoder layers; cout % 4 == 0): per column tile j, the statistics of
    //      y = alpha * acc + bias over this wave's rows
    [&]<int... J>(std::integer_sequence<int, J...>) {
    ([&] {
        constexpr int j = J;
#pragma unroll
        for (int e = 0; e < 4; ++e) ssum[e] = ssq[e] = 0.f;
        [&]<int... I>(std::integer_sequence<int, I...>) {
            ([&] {
                stage_tile(std::integral_constant<int, j * TM + I>{}, std::integral_constant<int, 0>{});
                __builtin_amdgcn_wave_barrier();
                epi_dispatch<TM, WROWS, 1>(a, stage, rowmap, j * TM + I, 1, ncol0, wm, rr, c4, true, ssum, ssq);
                __builtin_amdgcn_wave_barrier();
            }(), ...);
        }(std::make_integer_sequence<int, TM>{});
        const int n = ncol0 + j * 32;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int o = 8; o < 64; o <<= 1) {
                ssum[e] += __shfl_xor(ssum[e], o);
                ssq[e] += __shfl_xor(ssq[e], o);
            }
        }
        if (rr == 0) {
            const int64_t row = (int64_t)m_tile * 2 + wm;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                stat_sum.st1(row * cout_pad + n + e, ssum[e]);
                stat_sq.st1(row * cout_pad + n + e, ssq[e]);
            }
        }
    }(), ...);
    }(std::make_integer_sequence<int, TN>{});
}

template <int BM, int BN>
__device__ __forceinline__ void conv_epilogue(const woft_conv_params& p, f32x16 (&acc)[BM / 64][BN / 64],
                                              float* stage, int64_t m0, int n0, int wm, int wn, int lane, int64_t M,
                                              int m_tile) {
    conv_epilogue_t<BM / 64, BN / 64, BM / 2, BN / 2>(p, acc, stage, LinearRows{m0, M}, n0, wm, wn, lane, m_tile);
}

}  // namespace woft
