// Volume-free correlation lookup (the reference's alt_cuda_corr idea, corr.py:72-100 + the CUDA
// extension it binds): the (2r+1)^2 x L bilinear samples of corr.py:29-59 computed straight from the feature maps,
//   corr_l(p, q) = alpha * <fmap1[p], pool_l(fmap2)[q]>         (pooling commutes with the dot product)
// without ever materialising the P x P volume (5.6 GB at 1080p, 89 GB at 4K).
//
// A workgroup (2 x 2 waves, two workgroups per CU) owns an 8 x 8 block of source pixels whose split features stay in
// registers as MFMA A fragments (32 rows per wave).  Per pyramid level it takes the bounding box of the 64 lookup
// windows (smooth flow: (8 / 2^l + 2r + 2)^2 target pixels) and computes the 64 x |box| block of correlations as a
// split-bf16 MFMA GEMM, 64 box positions (a "chunk") at a time: the target rows stream global -> LDS by LDS-DMA through
// a 6-stage ring of K steps (one workgroup barrier per two K steps), same operand format and product order as
// corr_gemm_bf16_kernel -- every correlation value is bit-identical to the one the volume GEMM would have stored.
// After a chunk every lane drops its 16 correlations into the (2r+2)^2 windows (LDS) of the pixels whose window
// contains that box position; when the box is done the samples are interpolated from the pixel's own window with the
// arithmetic of corr_lookup_kernel, so the output equals the volume path's bit for bit.  A window of an outlier pixel
// only enlarges its block's box (more chunks): always correct, fast when the flow is locally smooth.
//
// Sampling rule, channel order and zero padding: as corr_lookup_kernel (lookup.hip).
//
// Round 6 (profiles/r06_lookup_*; every step bit-identical, A/B'd in one GPU call): 75-82 -> 65-72 us at 1/8 of 1080p.
//   * wave 0 (lane = source pixel) computes window origins, interpolation weights, drop-test operands and the bounding
//     boxes (DPP reductions) of ALL levels in the prologue, while the other waves fetch their A fragments (was: per level,
//     a barrier, then 24 cross-lane shuffles in every wave);
//   * the stream of level l + 1 is primed BEFORE the samples of level l are interpolated and written;
//   * windows are cleared only when the box was clipped at a map border (cells outside the map are the only ones never
//     written) and stored x-major; a sampling thread takes a window COLUMN: the two cell columns it needs are one run of
//     2 (2r+2) floats, a horizontal interpolation is computed once for the two outputs that use it, 2r + 1 consecutive
//     outputs per thread (was: a sample at a time, four LDS reads each);
//   * window drop: one packed 16-bit test per value (v_pk_sub_u16, v_pk_min_u16, compare), branch-free stores (positions
//     outside a pixel's window go to a per-lane dummy cell), operands fetched four pixels at a time;
//   * a wave's two 1-KiB DMA pieces of a K step share one M0 set-up and leave behind the first MFMAs of a K step instead
//     of right after the barrier; ablation bits and stamps are template parameters (no run-time branches in the K loops).
// Measured and NOT kept in round 6: the same lookup as four AUTONOMOUS waves (a wave owns all 64 rows = 256 registers of A
// fragments -> one wave per SIMD, private LDS rings, no barrier in the stream): bit-identical, 90-101 us -- with one wave
// per SIMD nothing hides a dependent instruction's latency, a DMA issue or another wave's phase; the two co-resident
// workgroups of THIS kernel are what overlaps one's drop / sampling / DMA issue with the other's MFMAs
// (profiles/r06_lookup_autonomous_waves.txt, tools/micro/dma_issue_probe.hip; the code is in the git history).
// Where the time is now (profiles/r06_lookup_otf_timeline.txt): a workgroup lives ~99 k cycles: set-up 5 k, its 15 chunks 73 k
// (4.9 k each: per two K steps ~50 barrier + ~850 fragment reads / 12 MFMAs / DMA issue with the other workgroup's wave sharing
// the SIMD; ~1.2 k window drop), sampling + priming of the next level 16 k.  Both the matrix pipe (3.07 k per chunk and SIMD for
// two workgroups) and the L2 -> LDS path (64 KB per chunk and workgroup at the ~40-50 B/clk a CU's LDS-DMA sustains: ~3.2 k)
// would allow ~3.2 k per chunk.
// Measured and NOT kept (rounds 2-5; git history): DMA pieces one per k sub-step (88.3 vs 82.0 us); the window drop of chunk
// c - 1 under the MFMAs of chunk c (+-0 in round 3; again in round 6 on this kernel, in 20 slices behind the sub-steps' MFMAs:
// 74.6 vs 72.0 us -- anything between two MFMAs on the one accumulator delays the dependent MFMA); a fully unrolled sampling loop (spills); 16 x 8-pixel blocks (-1.7 % frames/s);
// box indexing by float reciprocal (slower).
#include <type_traits>
#include <utility>

#include "common.h"
#include "dma.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <bool MAX>
__device__ __forceinline__ int wave_reduce_minmax(int v) {       // all 64 lanes active; result in every lane's return value (uniform)
#define WOFT_DPP_STEP(ctrl, rmask)                                                        \
    {                                                                                     \
        const int o = __builtin_amdgcn_update_dpp(v, v, ctrl, rmask, 0xf, false);         \
        v = MAX ? (o > v ? o : v) : (o < v ? o : v);                                      \
    }
    WOFT_DPP_STEP(0xB1, 0xf)      // quad_perm [1, 0, 3, 2]
    WOFT_DPP_STEP(0x4E, 0xf)      // quad_perm [2, 3, 0, 1]
    WOFT_DPP_STEP(0x141, 0xf)     // row_half_mirror
    WOFT_DPP_STEP(0x140, 0xf)     // row_mirror: every lane of a 16-lane row holds the row's result
    WOFT_DPP_STEP(0x142, 0xa)     // row_bcast15 into rows 1 and 3
    WOFT_DPP_STEP(0x143, 0xc)     // row_bcast31 into rows 2 and 3: lane 63 holds the result
#undef WOFT_DPP_STEP
    return __builtin_amdgcn_readlane(v, 63);
}

// a wave's two 1-KiB pieces of one K step (rows 8 wave .. and 8 (wave + 4) .. of the 64-row stage) with ONE M0 set-up: the immediate
// offset moves the LDS destination AND the memory address, so the first piece's lane offset carries + 4096
__device__ __forceinline__ void lds_dma16x2(const void* gbase, uint32_t o0_plus_4096, uint32_t o1, uint32_t lds_addr_second) {
    const uint64_t gb = (uint64_t)(uintptr_t)gbase;
    const uint32_t g_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(gb >> 32));
    const uint32_t g_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)gb);
    const uint64_t gu = ((uint64_t)g_hi << 32) | (uint64_t)g_lo;
    uint32_t saved_m0;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %2, %4 offset:-4096\n\tglobal_load_lds_dwordx4 %3, %4\n\ts_mov_b32 m0, %0"
                 : "=&s"(saved_m0)
                 : "s"(lds_addr_second), "v"(o0_plus_4096), "v"(o1), "s"(gu)
                 : "memory", "vcc");
}

template <int TERMS, int R, int K, int ABL>
__global__ __launch_bounds__(256, 2) void corr_lookup_otf_kernel(const woft_lookup_otf_params p) {
    constexpr int NPX = 64, NT = 256;
    constexpr int NW = 2 * R + 1, N2 = NW * NW;
    constexpr int NPL = (TERMS == 1) ? 1 : 2;
    constexpr int LD = (TERMS == 1) ? K : 2 * K;
    constexpr int NK = LD / 64;
    constexpr int NSUB = (TERMS == 1) ? 4 : 2;
    constexpr int GS = 2, NGRP = 3, NST = GS * NGRP, DEPTH = NGRP - 1;
    static_assert(NK % GS == 0, "K steps per chunk must be a multiple of the steps per barrier");
    __shared__ __attribute__((aligned(16))) __bf16 stage[NST * 64 * 64];    // NST stages of 64 B rows, 128 B each
    constexpr int WS = NW + 1, WLD = WS * WS + 2;       // a window: WS x WS cells, x-major (cell (cx, cy) at cx * WS + cy); even stride
    __shared__ __attribute__((aligned(16))) float Wn[NPX * WLD];
    constexpr int MAXL = 4;
    __shared__ __attribute__((aligned(16))) int s_org[MAXL][NPX];    // window origin per level, clamped to 16 bits and packed (x | y << 16)
    __shared__ __attribute__((aligned(16))) int s_pm[MAXL][NPX];     // byte offset in Wn of the window's cell (0, 0) minus 4 (ox * WS + oy)
    __shared__ float s_fx[MAXL][NPX], s_fy[MAXL][NPX];
    __shared__ int s_box[MAXL][4];                       // unclipped bounding box of the valid window origins: min x, max x, min y, max y
    __shared__ float s_trash[64];
    __shared__ uint32_t s_probe[24];                     // developer probe (ABL & 64): phases of one chunk (level 0, second chunk), thread 0
    int n_probe = 0;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r32 = lane & 31, hh = lane >> 5;
    constexpr int ld = LD, nk = NK;
    const int tiles_x = (p.wf + 7) / 8, ntiles = tiles_x * ((p.hf + 7) / 8);
    int tile;
    {
        const int q = ntiles / 8, rr = ntiles % 8, xcd = blockIdx.x % 8, idx = blockIdx.x / 8;
        tile = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
    }
    const int px0 = (tile % tiles_x) * 8, py0 = (tile / tiles_x) * 8;
    if (p.need != nullptr) {      // nobody wants this block's samples (the weight head on a subset of the source pixels)
        int any = 0;
        if (tid < NPX) {
            const int y = py0 + (tid >> 3), x = px0 + (tid & 7);
            if (y < p.hf && x < p.wf) any = p.need[y * p.wf + x];
        }
        if (!__syncthreads_or(any)) return;
    }
    // lookup centre of source pixel `tid` (wave 0), read once: requested BEFORE the A fragments (loads return in order)
    if (tid < NPX) {
        const int y = py0 + (tid >> 3), x = px0 + (tid & 7);
        float cx = 0.f, cy = 0.f;
        if (y < p.hf && x < p.wf) {
            const int64_t i = (int64_t)y * p.wf + x;
            cx = p.coords[i * 2];
            cy = p.coords[i * 2 + 1];
            if (p.fh_part != nullptr) {
                // the previous iteration's flow-head gather for this pixel (woft_flow_head_gather's operations, in its
                // order: planes first, then the 9 taps), then coords1 += delta and the flow operands of this iteration
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                float dx = p.fh_bias ? p.fh_bias[0] : 0.f, dy = p.fh_bias ? p.fh_bias[1] : 0.f;
                const int64_t plane = (int64_t)p.hf * p.wf * p.fh_ld;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int yy = y + ky - 1, xx = x + kx - 1;
                        f32x2 v = {0.f, 0.f};
                        if (yy >= 0 && yy < p.hf && xx >= 0 && xx < p.wf) {
                            const float* src = p.fh_part + ((int64_t)yy * p.wf + xx) * p.fh_ld + (ky * 3 + kx) * 2;
                            v = *(const f32x2*)src;
                            for (int t = 1; t < p.fh_planes; ++t) v += *(const f32x2*)(src + t * plane);
                        }
                        dx += v[0];
                        dy += v[1];
                    }
                p.fh_delta[i * p.fh_ld_delta] = dx;
                p.fh_delta[i * p.fh_ld_delta + 1] = dy;
                cx += dx;
                cy += dy;
                ((float*)p.coords)[i * 2] = cx;
                ((float*)p.coords)[i * 2 + 1] = cy;
                const float fx = cx - (float)x, fy = cy - (float)y;
                if (p.fh_flow4 != nullptr) *(f32x4*)(p.fh_flow4 + i * 4) = f32x4{fx, fy, 0.f, 0.f};
                if (p.fh_flow_cat != nullptr) {
                    p.fh_flow_cat[i * p.fh_ld_cat] = fx;
                    p.fh_flow_cat[i * p.fh_ld_cat + 1] = fy;
                }
            }
        }
        // window origins, interpolation weights, drop-test operands and bounding boxes of ALL levels, while the other waves fetch
        // their A fragments (the DPP reductions cost this wave ~1 k cycles per level)
        const bool cvalid = y < p.hf && x < p.wf;
#pragma unroll
        for (int l = 0; l < MAXL; ++l) {
            if (l < p.levels) {
                int wx0 = 0, wy0 = 0;
                float fx = 0.f, fy = 0.f;
                if (cvalid) {
                    const float sc = 1.0f / (float)(1 << l);
                    const float xs = cx * sc, ys = cy * sc;
                    float flx = floorf(xs), fly = floorf(ys);
                    fx = xs - flx;
                    fy = ys - fly;
                    flx = fminf(fmaxf(flx, -1.0e6f), 1.0e6f);
                    fly = fminf(fmaxf(fly, -1.0e6f), 1.0e6f);
                    wx0 = (int)flx - R;
                    wy0 = (int)fly - R;
                }
                s_fx[l][tid] = fx; s_fy[l][tid] = fy;
                // drop test operand: a position (tx, ty) of the map (0 <= tx, ty < 2^14) lies in the window iff both 16-bit halves of
                // (tx | ty << 16) - org are below WS; origins beyond +-2^14 are clamped (their windows contain no map position)
                const int ox = wx0 < -16384 ? -16384 : (wx0 > 16384 ? 16384 : wx0);
                const int oy = wy0 < -16384 ? -16384 : (wy0 > 16384 ? 16384 : wy0);
                s_org[l][tid] = cvalid ? ((ox & 0xffff) | (oy << 16)) : 0x40004000;      // (pixels outside the grid: never inside)
                s_pm[l][tid] = 4 * (tid * WLD - (ox * WS + oy));     // (BYTE offset: the drop adds its position's once per chunk)
                const int b0 = wave_reduce_minmax<false>(cvalid ? wx0 : 0x3fffffff), b1 = wave_reduce_minmax<true>(cvalid ? wx0 : -0x3fffffff);
                const int b2 = wave_reduce_minmax<false>(cvalid ? wy0 : 0x3fffffff), b3 = wave_reduce_minmax<true>(cvalid ? wy0 : -0x3fffffff);
                if (tid == 0) { s_box[l][0] = b0; s_box[l][1] = b1; s_box[l][2] = b2; s_box[l][3] = b3; }
            }
        }
    }
    // The block's source features stay in REGISTERS for the whole kernel, as the MFMA A fragments of this wave's 32 rows
    bf16x8 afr[NK][NSUB][NPL];
    {
        const int m = wm * 32 + r32;
        int y = py0 + (m >> 3), x = px0 + (m & 7);
        y = y < p.hf ? y : p.hf - 1;                    // rows outside the grid repeat a valid pixel (never written)
        x = x < p.wf ? x : p.wf - 1;
        const char* row = (const char*)p.f1 + (int64_t)(y * p.wf + x) * (ld * 2);
#pragma unroll
        for (int ks = 0; ks < NK; ++ks)
#pragma unroll
            for (int s2 = 0; s2 < NSUB; ++s2) {
                if (TERMS == 0) {
                    afr[ks][s2][0] = *(const bf16x8*)(row + ks * 128 + hh * 64 + s2 * 32);
                    afr[ks][s2][NPL - 1] = *(const bf16x8*)(row + ks * 128 + hh * 64 + s2 * 32 + 16);
                } else {
                    afr[ks][s2][0] = *(const bf16x8*)(row + ks * 128 + (s2 * 2 + hh) * 16);
                    if (TERMS == 3) afr[ks][s2][NPL - 1] = *(const bf16x8*)(row + ks * 128 + 64 + (s2 * 2 + hh) * 16);
                }
            }
    }
    // B stream: the 8 DMA pieces of a step (piece q = rows 8 q .. 8 q + 7 of the 64 box positions) are issued by waves q = wave, wave + 4
    const int swz = (4 * (wave & 1) + (lane >> 4)) & 7;
    const uint32_t chunk_off = (uint32_t)(((lane & 7) ^ swz) * 16);
    const uint32_t st_addr = lds_addr_of(stage);
    const int sw = (r32 >> 1) & 7;
    const __bf16* b_rows = stage + (wn * 32 + r32) * 64;

    uint32_t* stamps = ((ABL & 16) && tid == 0 && p.ldo >= 4 * N2 + 24)
                           ? (uint32_t*)(p.out + ((int64_t)py0 * p.wf + px0) * p.ldo + 4 * N2) : nullptr;
    int n_stamp = 0;
    auto stamp = [&]() __attribute__((always_inline)) { if ((ABL & 16) && stamps && n_stamp < 24) stamps[n_stamp++] = (uint32_t)__builtin_amdgcn_s_memtime(); };
    stamp();

    __syncthreads();                                     // origins / boxes of all levels are in LDS
    stamp();
    // (level-indexed kernel arguments through select chains: a run-time index into the argument struct would make the compiler
    //  keep a copy of it in scratch memory)
    auto sel4 = [](int l, int a0, int a1, int a2, int a3) __attribute__((always_inline)) { return l == 0 ? a0 : (l == 1 ? a1 : (l == 2 ? a2 : a3)); };
    // per-level stream state: box, rows, chunk count; `issue` requests this wave's two pieces of the next step
    int W = 0, bx0 = 0, by0 = 0, bw = 1, N = 0, S = 0, G = 0;
    bool clipped = false;
    const char* f2 = nullptr;
    uint32_t b_off[2] = {0u, 0u};
    int is_c = 0, is_k = 0;                              // (chunk, k step) of the next step to request
    auto issue = [&](int s_idx) __attribute__((always_inline)) {
        if (is_k == 0) {                                 // new chunk: rows of the box positions c0 .. c0 + 63
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                int pos = is_c * 64 + (wave + 4 * tt) * 8 + (lane >> 3);
                pos = pos < N ? pos : N - 1;            // (columns past the box repeat its last position; never read)
                const int by = pos / bw, bx = pos - by * bw;
                b_off[tt] = (uint32_t)((by0 + by) * W + bx0 + bx) * (uint32_t)(ld * 2) + chunk_off + (tt == 0 ? 4096u : 0u);
            }
        }
        const uint32_t st = st_addr + (uint32_t)(s_idx % NST) * 8192u;
        if (!(ABL & 1) || s_idx < NST) lds_dma16x2(f2 + is_k * 128, b_off[0], b_off[1], st + (uint32_t)(wave + 4) * 1024u);
        if (++is_k == nk) { is_k = 0; ++is_c; }
    };
    // the stream of level l is primed (its first DEPTH groups requested) BEFORE the samples of level l - 1 are interpolated and written:
    // the first rows arrive while the workgroup is busy with that
    auto start_level = [&](int l) __attribute__((always_inline)) {
        W = sel4(l, p.w[0], p.w[1], p.w[2], p.w[3]);
        const int H = sel4(l, p.h[0], p.h[1], p.h[2], p.h[3]);
        bx0 = __builtin_amdgcn_readfirstlane(s_box[l][0]);
        by0 = __builtin_amdgcn_readfirstlane(s_box[l][2]);
        int bx1 = __builtin_amdgcn_readfirstlane(s_box[l][1]), by1 = __builtin_amdgcn_readfirstlane(s_box[l][3]);
        clipped = bx0 < 0 || by0 < 0 || bx1 + NW > W - 1 || by1 + NW > H - 1;
        bx0 = bx0 > 0 ? bx0 : 0;
        by0 = by0 > 0 ? by0 : 0;
        bx1 = (bx1 + NW < W - 1) ? bx1 + NW : W - 1;         // windows span wx0 .. wx0 + 2R + 1
        by1 = (by1 + NW < H - 1) ? by1 + NW : H - 1;
        bw = bx1 - bx0 + 1;
        const int bh = by1 - by0 + 1;
        N = (bw > 0 && bh > 0) ? bw * bh : 0;
        bw = bw > 0 ? bw : 1;
        f2 = (const char*)(l == 0 ? p.f2[0] : (l == 1 ? p.f2[1] : (l == 2 ? p.f2[2] : p.f2[3])));
        S = ((N + 63) / 64) * nk;
        G = S / GS;
        is_c = 0; is_k = 0;
        for (int g = 0; g < DEPTH && g < G; ++g)
#pragma unroll
            for (int e = 0; e < GS; ++e) issue(g * GS + e);
    };
    start_level(0);
    for (int l = 0; l < p.levels; ++l) {
        if (clipped) {                                   // cells outside the map are never written and must read as zero
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            for (int i = tid; i < NPX * WLD / 4; i += NT) ((f32x4*)Wn)[i] = z4;
            static_assert((NPX * WLD) % 4 == 0, "window array is zeroed 16 bytes at a time");
        }
        // (visible to all waves after the first step barrier below; S == 0: the barrier before the interpolation)
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        int c0 = 0;
        stamp();
        for (int s0 = 0; s0 < S; s0 += NK) {             // one 64-column chunk per iteration, K steps unrolled
            [&]<int... KGS>(std::integer_sequence<int, KGS...>) {
            ([&] {
                constexpr int kg = KGS;
                const int g = s0 / GS + kg;
                const int rem = G - 1 - g;
                const bool pr = (ABL & 64) && l == 0 && s0 == NK && tid == 0;
                auto mark = [&]() __attribute__((always_inline)) { if ((ABL & 64) && pr && n_probe < 24) s_probe[n_probe++] = (uint32_t)__builtin_amdgcn_s_memtime(); };
                mark();
                if (rem >= DEPTH - 1) dma_wait<(DEPTH - 1) * GS * 2>();
                else dma_wait<0>();
                mark();
                __syncthreads();                         // ... for every wave; and group g - 1 is fully consumed
                mark();
                const bool feed = g + DEPTH < G;
                mark();
                constexpr int NU = GS * NSUB;
                bf16x8 bq[2][NPL];
                auto load_b = [&](auto u_tag) __attribute__((always_inline)) {
                    constexpr int u = decltype(u_tag)::value;
                    constexpr int e = u / NSUB, s2 = u % NSUB;
                    const __bf16* br = b_rows + ((s0 + kg * GS + e) % NST) * 4096;
                    if (TERMS == 0) {
                        bq[u & 1][0] = *(const bf16x8*)(br + ((hh * 4 + s2 * 2) ^ sw) * 8);
                        bq[u & 1][NPL - 1] = *(const bf16x8*)(br + ((hh * 4 + s2 * 2 + 1) ^ sw) * 8);
                    } else {
                        bq[u & 1][0] = *(const bf16x8*)(br + ((s2 * 2 + hh) ^ sw) * 8);
                        if (TERMS == 3) bq[u & 1][NPL - 1] = *(const bf16x8*)(br + ((4 + s2 * 2 + hh) ^ sw) * 8);
                    }
                };
                if (!(ABL & 2)) {
                    load_b(std::integral_constant<int, 0>{});
                    [&]<int... U>(std::integer_sequence<int, U...>) {
                        ([&] {
                            constexpr int u = U, e = u / NSUB, s2 = u % NSUB;
                            constexpr int ks = kg * GS + e;
                            if constexpr (u + 1 < NU) load_b(std::integral_constant<int, u + 1>{});
                            __builtin_amdgcn_sched_barrier(0);
                            const bf16x8 bh = bq[u & 1][0];
                            if (TERMS == 0) {
                                const f32x4 a0 = __builtin_bit_cast(f32x4, afr[ks][s2][0]), a1 = __builtin_bit_cast(f32x4, afr[ks][s2][NPL - 1]);
                                const f32x4 b0 = __builtin_bit_cast(f32x4, bh), b1 = __builtin_bit_cast(f32x4, bq[u & 1][NPL - 1]);
#pragma unroll
                                for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b0[s], acc, 0, 0, 0);
#pragma unroll
                                for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b1[s], acc, 0, 0, 0);
                            } else if (TERMS == 3) {
                                const bf16x8 bl = bq[u & 1][NPL - 1];
                                const bf16x8 ah = afr[ks][s2][0], al = afr[ks][s2][NPL - 1];
                                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);   // (order of corr_gemm_bf16_kernel)
                                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
                            } else {
                                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ks][s2][0], bh, acc, 0, 0, 0);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            // this wave's two pieces of step (g + DEPTH) GS + e go out behind the first sub-step's MFMAs of K step e: one
                            // pair per K step instead of all four pieces right after the barrier, where the 16 pieces of the four waves
                            // queued at the CU's one texture addresser (~400 cycles with nobody computing; A/B: -1.5 us)
                            if constexpr (s2 == 0) { if (feed) issue((g + DEPTH) * GS + e); }
                        }(), ...);
                    }(std::make_integer_sequence<int, NU>{});
                }
                mark();
            }(), ...);
            }(std::make_integer_sequence<int, NK / GS>{});
            // ---- chunk complete: every lane drops its 16 correlations (one box position, 16 source pixels) into the windows that
            //      contain that position.  Branch-free; operands of four pixels per fetch, all fetched before the first store ----
            if (!(ABL & 4)) {
                typedef int i32x4 __attribute__((ext_vector_type(4)));
                typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
                const int pos = c0 + wn * 32 + r32;
                const int by = pos / bw, bx = pos - by * bw;
                const int tx = bx0 + bx, ty = by0 + by;  // target pixel of this column
                // (a column past the box end takes a position no window contains: 32767 - origin >= 16383 in both halves)
                const int d_pos = pos < N ? ((tx & 0xffff) | (ty << 16)) : 0x7fff7fff;
                const uint32_t d_q4 = lds_addr_of(Wn) + 4u * (uint32_t)(tx * WS + ty);
                const uint32_t trash = lds_addr_of(s_trash) + 4u * (uint32_t)lane;
                i32x4 org[4], pm[4];
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    org[g4] = *(const i32x4*)&s_org[l][wm * 32 + 4 * hh + 8 * g4];
                    pm[g4] = *(const i32x4*)&s_pm[l][wm * 32 + 4 * hh + 8 * g4];
                }
                const u16x2 lim = {(unsigned short)(WS - 1), (unsigned short)(WS - 1)};
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const u16x2 dd = __builtin_bit_cast(u16x2, d_pos) - __builtin_bit_cast(u16x2, (int)org[r >> 2][r & 3]);
                    const u16x2 mn = __builtin_elementwise_min(dd, lim);
                    const bool inside = __builtin_bit_cast(uint32_t, mn) == __builtin_bit_cast(uint32_t, dd);
                    const uint32_t cell = inside ? (uint32_t)pm[r >> 2][r & 3] + d_q4 : trash;
                    *(__attribute__((address_space(3))) float*)(uintptr_t)cell = acc[r] * p.alpha;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            if ((ABL & 64) && l == 0 && s0 == NK && tid == 0 && n_probe < 24) s_probe[n_probe++] = (uint32_t)__builtin_amdgcn_s_memtime();
            c0 += 64;
        }
        stamp();
        __syncthreads();                                 // every wave's drops are in the windows; the ring is free
        if (l + 1 < p.levels) start_level(l + 1);
        // ---- bilinear samples from the pixel's own window: the arithmetic of corr_lookup_kernel, one window column per item ----
        if (!(ABL & 8)) {
            for (int it = tid; it < NPX * NW; it += NT) {
                const int pix = it / NW, i = it - pix * NW;
                const int gy = py0 + (pix >> 3), gx = px0 + (pix & 7);
                if (gy >= p.hf || gx >= p.wf) continue;
                const float fx = s_fx[l][pix], fy = s_fy[l][pix];
                const float* cq = Wn + pix * WLD + i * WS;
                float h[WS];
#pragma unroll
                for (int j = 0; j < WS; ++j) h[j] = cq[j] * (1.f - fx) + cq[WS + j] * fx;
                float* o = p.out + ((int64_t)gy * p.wf + gx) * p.ldo + l * N2 + i * NW;
#pragma unroll
                for (int j = 0; j < NW; ++j) o[j] = h[j] * (1.f - fy) + h[j + 1] * fy;
            }
        }
        stamp();
        __syncthreads();
    }
    if ((ABL & 64) && tid == 0 && p.ldo >= 4 * N2 + 24) {
        uint32_t* o = (uint32_t*)(p.out + ((int64_t)py0 * p.wf + px0) * p.ldo + 4 * N2);
        for (int i = 0; i < 24; ++i) o[i] = i < n_probe ? s_probe[i] : 0u;
    }
}

}  // namespace

extern "C" int woft_corr_lookup_otf(const woft_lookup_otf_params* pp, void* stream) {
    if (!pp) return WOFT_EINVAL;
    const woft_lookup_otf_params& p = *pp;
    if (p.levels < 1 || p.levels > 4 || !p.f1 || !p.coords || !p.out || p.hf <= 0 || p.wf <= 0) return WOFT_EINVAL;
    if (p.terms != 0 && p.terms != 1 && p.terms != 3) return WOFT_EINVAL;
    if (p.k <= 0 || p.k % (p.terms == 1 ? 64 : 32) != 0) return WOFT_EINVAL;
    const int64_t row_bytes = (int64_t)p.k * (p.terms == 1 ? 2 : 4);
    if ((int64_t)p.hf * p.wf * row_bytes >= (1ll << 32)) return WOFT_EINVAL;        // 32-bit lane offsets
    for (int l = 0; l < p.levels; ++l)
        if (!p.f2[l] || p.h[l] <= 0 || p.w[l] <= 0 || (int64_t)p.h[l] * p.w[l] * row_bytes >= (1ll << 32)) return WOFT_EINVAL;
    const int nout = p.levels * (2 * p.radius + 1) * (2 * p.radius + 1);
    if (p.ldo < nout) return WOFT_EINVAL;
    if (p.fh_part != nullptr && (p.fh_delta == nullptr || p.fh_planes < 1 || p.fh_ld < 20 || p.fh_ld % 4 != 0 || p.fh_ld_delta < 2 ||
                                 (p.fh_flow_cat != nullptr && p.fh_ld_cat < 2) || p.need != nullptr))
        return WOFT_EINVAL;
    for (int l = 0; l < p.levels; ++l) {
        if (p.h[l] > 16384 || p.w[l] > 16384) return WOFT_EINVAL;                       // (packed 16-bit map positions in the drop test)
        if ((int64_t)p.h[l] * p.w[l] * row_bytes >= (1ll << 32) - 4096) return WOFT_EINVAL;   // (lane offsets carry + 4 KiB: lds_dma16x2)
    }
    hipStream_t s = (hipStream_t)stream;
    // 8 x 8 source pixels per workgroup.  (16 x 8: 40 % less target-row traffic, but one 8-wave workgroup per CU and 20 % more steps
    // per workgroup: measured 110 vs 98 us at 1080p in round 1, -1.7 % frames/s in round 2 -- the per-workgroup chain of K steps binds.)
    dim3 grid((unsigned)(((p.wf + 7) / 8) * ((p.hf + 7) / 8)));
#define OTF(T, RR, KK, AB) woft_launch(0, corr_lookup_otf_kernel<T, RR, KK, AB>, grid, dim3(256), 0, s, p)
    if (p.ablate != 0) {          // developer instances (tools/bench_lookup_otf.py): full model, split-bf16 only
        if (!(p.k == 256 && p.radius == 4 && p.terms == 3)) return WOFT_EINVAL;
        switch (p.ablate) {
            case 1: OTF(3, 4, 256, 1); break;       // no target-row stream after the first ring fill
            case 2: OTF(3, 4, 256, 2); break;       // no fragment reads / MFMAs
            case 3: OTF(3, 4, 256, 3); break;
            case 8: OTF(3, 4, 256, 8); break;       // no interpolation / output
            case 15: OTF(3, 4, 256, 15); break;     // ... nor window drops: the skeleton
            case 16: OTF(3, 4, 256, 16); break;     // phase stamps of thread 0 in the padding columns of the block's first output row
            case 64: OTF(3, 4, 256, 64); break;     // stamps inside one chunk (level 0, second chunk)
            default: return WOFT_EINVAL;
        }
    } else if (p.k == 256 && p.radius == 4) {                                                         /* full model  */
        if (p.terms == 3) OTF(3, 4, 256, 0); else if (p.terms == 0) OTF(0, 4, 256, 0); else OTF(1, 4, 256, 0);
    } else if (p.k == 128 && p.radius == 3) {                                                         /* small model */
        if (p.terms == 3) OTF(3, 3, 128, 0); else if (p.terms == 0) OTF(0, 3, 128, 0); else OTF(1, 3, 128, 0);
    }
    else return WOFT_EINVAL;
#undef OTF
    return woft_launch_status();
}
