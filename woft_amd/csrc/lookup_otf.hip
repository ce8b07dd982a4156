// Volume-free correlation lookup (the reference's alt_cuda_corr idea, corr.py:72-100 + the CUDA
// extension it binds): the (2r+1)^2 x L bilinear samples of corr.py:29-59 computed straight from the feature maps,
//   corr_l(p, q) = alpha * <fmap1[p], pool_l(fmap2)[q]>         (pooling commutes with the dot product)
// without ever materialising the P x P volume (5.6 GB at 1080p, 89 GB at 4K).
//
// A workgroup owns an 8 x 8 block of source pixels.  Per pyramid level it finds the bounding box of the 64 lookup
// windows (for a smooth flow field: (8 / 2^l + 2r + 1)^2 target pixels), and computes the 64 x |box| block of
// correlations as a split-bf16 MFMA GEMM -- 64 box positions at a time, both operand tiles copied global -> LDS by
// LDS-DMA exactly as in corr_gemm_bf16_kernel (same operand format, same product order: every correlation value is
// bit-identical to the one the volume GEMM would have stored).  After each 64-column chunk every lane drops its 16
// correlations into the (2r+2)^2 windows (LDS) of the pixels whose window contains that box position; when the box
// is done the samples are interpolated from the pixel's own window with the arithmetic of corr_lookup_kernel, so the
// output equals the volume path's bit for bit.  A window of an outlier pixel only enlarges its block's box (more
// chunks): always correct, fast when the flow is locally smooth.
//
// Sampling rule, channel order and zero padding: as corr_lookup_kernel (lookup.hip).
#include <type_traits>
#include <utility>

#include "common.h"
#include "dma.h"

// Measured and NOT kept (rounds 2-3, tools/bench_lookup_otf.py; the code is in the git history): DMA pieces issued one per k
// sub-step instead of four in a row after the barrier (88.3 vs 82.0 us); the window drop of chunk c - 1 spread under the MFMAs of
// chunk c (81.3 vs 81.9 us, +-0 in a frame); a fully unrolled sampling loop (+-0, spills at 256 registers).  Round 5: the three
// `pos / bw` per chunk and lane as a float multiply with the box width's reciprocal + one correction step (84.3-85.0 vs 82.0-82.8 us:
// the divisor is wave-uniform, so the compiler already hoists its reciprocal out of the chunk loop and a use costs a v_mul_hi_u32).
#define OTF_SAMPLE_UNROLL 3      // unroll factor of the sampling loop

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// TW: width of the block of source pixels (TW x 8; 8 -> 4 waves, 16 -> 8 waves, each owning 32 rows x 32 columns)
template <int TERMS, int R, int K, int TW>
__global__ __launch_bounds__(TW * 32, 2) void corr_lookup_otf_kernel(const woft_lookup_otf_params p) {
    constexpr int NPX = TW * 8;                         // source pixels (GEMM rows) per workgroup
    constexpr int NT = NPX * 4;                         // threads: four per source pixel
    constexpr int NWV = NT / 64;                        // waves: NPX / 32 row groups x 2 column halves
    constexpr int TSH = (TW == 16) ? 4 : 3;             // log2(TW)
    constexpr int NW = 2 * R + 1, N2 = NW * NW;
    // TERMS = 0 (exact fp32, round 3): the operand rows are the fp32 feature rows themselves -- a 128-byte line holds 32 k,
    // a sub-step is 8 x v_mfma_f32_32x32x2_f32 on the k pairs (16 hh + 8 s2 + s, hh = 0 / 1), lines and sub-steps in
    // conv_mfma_f32_kernel's order: every correlation value is bit-identical to the fp32 volume's.  Same ring, same boxes.
    constexpr int NPL = (TERMS == 1) ? 1 : 2;           // 16-byte fragments per lane, line and sub-step (hi, lo | two fp32 quads)
    constexpr int LD = (TERMS == 1) ? K : 2 * K;        // bf16-sized elements per operand row (fp32: K floats = 2 K of them)
    constexpr int NK = LD / 64;                         // K steps (one 128-byte line each)
    constexpr int NSUB = (TERMS == 1) ? 4 : 2;          // MFMA k sub-steps per line
    // LDS ring of B rows: NST stages of one K step each, consumed GS steps per workgroup barrier, DEPTH groups in flight
    // beyond the one being computed.  Where the time goes (round-2 ablation, tools/bench_lookup_otf.py OTF_ABL bits +
    // s_memtime stamps, 1080p, 89 us): without the target-row stream -6 us, without MFMAs -21 us, without both
    // and without window drops / output still 51 us -- the skeleton: per level ~9 k cycles of coordinates, box, window
    // zeroing and sampling, per 64-column chunk ~4.6 k cycles of which ~2.5 k were the window drop fetching each pixel's
    // window origin from LDS one read at a time (fixed below), and LDS-latency waits before every MFMA triple (fixed
    // below: B fragments one sub-step ahead).  89.2 -> 85.0 us, +1.4 % frames/s.
    constexpr int GS = 2, NGRP = 3, NST = GS * NGRP, DEPTH = NGRP - 1;
    static_assert(NK % GS == 0, "K steps per chunk must be a multiple of the steps per barrier");
    __shared__ __attribute__((aligned(16))) __bf16 stage[NST * 64 * 64];    // NST stages of 64 B rows, 128 B each
    constexpr int WS = NW + 1, WLD = WS * WS + 1;       // (2r+2)^2 window of a pixel (+1: spreads the LDS banks)
    __shared__ __attribute__((aligned(16))) float Wn[NPX * WLD];                     // the windows of the source pixels at the current level
    __shared__ int2 s_w0[NPX];                          // window origin (x, y) of every source pixel at the current level
    __shared__ float s_fx[NPX], s_fy[NPX];
    __shared__ float2 s_cc[NPX];                         // lookup centre of every source pixel (level 0 units)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r32 = lane & 31, hh = lane >> 5;
    constexpr int ld = LD, nk = NK;
    // workgroup -> 8 x 8 tile: consecutive workgroup ids land on consecutive XCDs (private L2 each), so the ids are
    // re-dealt to give every XCD a contiguous band of tiles -- neighbouring tiles' boxes overlap ~5x and then hit in L2
    const int tiles_x = (p.wf + TW - 1) / TW, ntiles = tiles_x * ((p.hf + 7) / 8);
    int tile;
    {
        const int q = ntiles / 8, rr = ntiles % 8, xcd = blockIdx.x % 8, idx = blockIdx.x / 8;
        tile = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
    }
    const int px0 = (tile % tiles_x) * TW, py0 = (tile / tiles_x) * 8;
    if (p.need != nullptr) {      // nobody wants this block's samples (the weight head on a subset of the source pixels)
        int any = 0;
        for (int e = threadIdx.x; e < NPX; e += NT) {
            const int y = py0 + (e >> TSH), x = px0 + (e & (TW - 1));
            if (y < p.hf && x < p.wf) any |= p.need[y * p.wf + x];
        }
        if (!__syncthreads_or(any)) return;
    }

    // lookup centre of source pixel `tid` (threads < NPX), read once: it is the same at every level.  Requested BEFORE the
    // A fragments (loads return in order: waiting for it must not wait for the 64 KB of source features behind it) and kept
    // in LDS, not in registers: across the level loop the compiler spilled it to scratch -- four scratch reloads per level,
    // each followed by s_waitcnt vmcnt(0) (round-3 reading of the ISA: ~2 k cycles of every level's set-up)
    if (tid < NPX) {
        const int y = py0 + (tid >> TSH), x = px0 + (tid & (TW - 1));
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
        s_cc[tid] = make_float2(cx, cy);
    }
    // The block's source features stay in REGISTERS for the whole kernel, as the MFMA A fragments of this wave's
    // 32 rows (lane (r32, hh): row r32, k = 8 (2 s + hh) .. + 7 of every line; hi and lo halves of the line) -- the
    // first version re-fetched the A tile with every 64-column chunk and was bound by that L2 -> LDS traffic.
    bf16x8 afr[NK][NSUB][NPL];
    {
        const int m = wm * 32 + r32;
        int y = py0 + (m >> TSH), x = px0 + (m & (TW - 1));
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
    // B stream: the 8 DMA pieces of a step (piece q = rows 8 q .. 8 q + 7 of the 64 box positions) are issued by
    // waves q = wave + NWV t; lane -> (row lane / 8, physical chunk lane % 8) holding logical chunk
    // (lane % 8) ^ ((row >> 1) & 7), and (row >> 1) & 7 = (4 (q & 1) + lane / 16) & 7 with q & 1 = wave & 1
    constexpr int QPW = 8 / NWV;                        // pieces per wave and step (2 or 1)
    const int swz = (4 * (wave & 1) + (lane >> 4)) & 7;
    const uint32_t chunk_off = (uint32_t)(((lane & 7) ^ swz) * 16);
    const uint32_t st_addr = lds_addr_of(stage);
    const int sw = (r32 >> 1) & 7;
    const __bf16* b_rows = stage + (wn * 32 + r32) * 64;

    // developer probe (ablate & 16): s_memtime stamps of thread 0 -> the padding columns of the tile's first output row
    uint32_t* stamps = ((p.ablate & 16) && tid == 0 && p.ldo >= 4 * N2 + 24)
                           ? (uint32_t*)(p.out + ((int64_t)py0 * p.wf + px0) * p.ldo + 4 * N2) : nullptr;
    int n_stamp = 0;
    auto stamp = [&]() { if (stamps && n_stamp < 24) stamps[n_stamp++] = (uint32_t)__builtin_amdgcn_s_memtime(); };
    stamp();
    const int mypix = tid >> 2, part = tid & 3;         // sampling: 4 threads per source pixel
    const int gy = py0 + (mypix >> TSH), gx = px0 + (mypix & (TW - 1));
    const bool pvalid = gy < p.hf && gx < p.wf;
    constexpr int NS = (N2 + 3) / 4;                    // samples per thread

    for (int l = 0; l < p.levels; ++l) {
        const int W = p.w[l], H = p.h[l];
        if (tid < NPX) {
            int wx0 = 0x3fffffff, wy0 = 0x3fffffff;     // (outside the grid: excluded from the box)
            float fx = 0.f, fy = 0.f;
            // (the index is laundered so that the four LDS addresses below are recomputed here -- one shift each, the array
            //  bases are instruction immediates: hoisted out of the level loop at 256 registers they were SPILLED, and every
            //  level began with four scratch reloads, each behind an s_waitcnt vmcnt(0))
            int t = tid;
            asm volatile("" : "+v"(t));
            const float2 cc = s_cc[t];                   // (written by this same thread)
            const float ccx = cc.x, ccy = cc.y;
            const bool cvalid = py0 + (t >> TSH) < p.hf && px0 + (t & (TW - 1)) < p.wf;
            if (cvalid) {
                const float sc = 1.0f / (float)(1 << l);
                const float xs = ccx * sc, ys = ccy * sc;
                float flx = floorf(xs), fly = floorf(ys);
                fx = xs - flx;
                fy = ys - fly;
                flx = fminf(fmaxf(flx, -1.0e6f), 1.0e6f);
                fly = fminf(fmaxf(fly, -1.0e6f), 1.0e6f);
                wx0 = (int)flx - R;
                wy0 = (int)fly - R;
            }
            s_w0[t] = make_int2(wx0, wy0); s_fx[t] = fx; s_fy[t] = fy;
        }
        stamp();
        __syncthreads();
        stamp();
        // bounding box of the valid windows: butterfly over the 64 lanes of every wave (all waves hold the result)
        int bx0, bx1, by0, by1;
        {
            bx0 = 0x3fffffff; bx1 = -0x3fffffff; by0 = 0x3fffffff; by1 = -0x3fffffff;
#pragma unroll
            for (int e = lane; e < NPX; e += 64) {
                const int vx = s_w0[e].x, vy = s_w0[e].y;
                if (vx != 0x3fffffff) {
                    bx0 = vx < bx0 ? vx : bx0; bx1 = vx > bx1 ? vx : bx1;
                    by0 = vy < by0 ? vy : by0; by1 = vy > by1 ? vy : by1;
                }
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                const int a0 = __shfl_xor(bx0, d, 64), a1 = __shfl_xor(bx1, d, 64);
                const int c0 = __shfl_xor(by0, d, 64), c1 = __shfl_xor(by1, d, 64);
                bx0 = a0 < bx0 ? a0 : bx0; bx1 = a1 > bx1 ? a1 : bx1;
                by0 = c0 < by0 ? c0 : by0; by1 = c1 > by1 ? c1 : by1;
            }
            bx0 = __builtin_amdgcn_readfirstlane(bx0); bx1 = __builtin_amdgcn_readfirstlane(bx1);
            by0 = __builtin_amdgcn_readfirstlane(by0); by1 = __builtin_amdgcn_readfirstlane(by1);
        }
        bx0 = bx0 > 0 ? bx0 : 0;
        by0 = by0 > 0 ? by0 : 0;
        bx1 = (bx1 + NW < W - 1) ? bx1 + NW : W - 1;     // windows span wx0 .. wx0 + 2R + 1
        by1 = (by1 + NW < H - 1) ? by1 + NW : H - 1;
        const int bw = bx1 - bx0 + 1, bh = by1 - by0 + 1;
        const int N = (bw > 0 && bh > 0) ? bw * bh : 0;

        {                                                           // cells outside the map stay zero
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            for (int i = tid; i < NPX * WLD / 4; i += NT) ((f32x4*)Wn)[i] = z4;
            static_assert((NPX * WLD) % 4 == 0, "window array is zeroed 16 bytes at a time");
        }
        // (visible to all waves after the first step barrier below; S == 0: the barrier before the interpolation)

        const char* f2 = (const char*)p.f2[l];
        // ---- K steps of all 64-column chunks of the box as ONE stream through an LDS ring: the B rows of step
        //      of group g + DEPTH are requested while group g computes (one workgroup barrier per GS steps) ----
        const int nchunk = (N + 63) / 64;
        const int S = nchunk * nk;
        uint32_t b_off[2] = {0u, 0u};
        int is_c = 0, is_k = 0;                          // (chunk, k step) of the next step to request
        // one DMA piece (t of QPW) of step s_idx; the pieces of a step are issued in order, t = 0 first
        auto issue_piece = [&](int s_idx, int t) {
            if (t == 0 && is_k == 0) {                   // new chunk: rows of the box positions c0 .. c0 + 63
#pragma unroll
                for (int tt = 0; tt < QPW; ++tt) {
                    int pos = is_c * 64 + (wave + NWV * tt) * 8 + (lane >> 3);
                    pos = pos < N ? pos : N - 1;        // (columns past the box repeat its last position; never read)
                    const int by = pos / bw, bx = pos - by * bw;
                    b_off[tt] = (uint32_t)((by0 + by) * W + bx0 + bx) * (uint32_t)(ld * 2) + chunk_off;
                }
            }
            const uint32_t st = st_addr + (uint32_t)(s_idx % NST) * 8192u;
            if (!(p.ablate & 1) || s_idx < NST)
                lds_dma16(f2 + is_k * 128, b_off[t], st + (uint32_t)(wave + NWV * t) * 1024u);
            if (t == QPW - 1 && ++is_k == nk) { is_k = 0; ++is_c; }
        };
        auto issue = [&](int s_idx) {
#pragma unroll
            for (int t = 0; t < QPW; ++t) issue_piece(s_idx, t);
        };
        const int G = S / GS;                            // groups of GS steps (S = nchunk * NK, NK % GS == 0)
        for (int g = 0; g < DEPTH && g < G; ++g)
#pragma unroll
            for (int e = 0; e < GS; ++e) issue(g * GS + e);
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
                // this group's pieces have landed; those of the following (up to DEPTH - 1) groups may still fly
                const int rem = G - 1 - g;
                if (rem >= DEPTH - 1) dma_wait<(DEPTH - 1) * GS * QPW>();
                else dma_wait<0>();
                __syncthreads();                         // ... for every wave; and group g - 1 is fully consumed
                // the GS * QPW pieces of group g + DEPTH go into the stages of group g - 1, all of them right after the barrier
                // (one per k sub-step between the fragment reads and the MFMAs measured 6 us slower: see above)
                const bool feed = g + DEPTH < G;
                if (feed) {
#pragma unroll
                    for (int e = 0; e < GS; ++e) issue((g + DEPTH) * GS + e);
                }
                // the GS steps of the group as one list of k sub-steps; the B fragments of sub-step u + 1 are requested
                // before the MFMAs of sub-step u (two register sets) -- left alone the compiler reads, waits out the LDS
                // latency and only then issues the three MFMAs, every sub-step
                constexpr int NU = GS * NSUB;
                bf16x8 bq[2][NPL];
                auto load_b = [&](auto u_tag) {
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
                if (!(p.ablate & 2)) {
                    load_b(std::integral_constant<int, 0>{});
                    [&]<int... U>(std::integer_sequence<int, U...>) {
                        ([&] {
                            constexpr int u = U, e = u / NSUB, s2 = u % NSUB;
                            const int ks = kg * GS + e;
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
                        }(), ...);
                    }(std::make_integer_sequence<int, NU>{});
                }
            }(), ...);
            }(std::make_integer_sequence<int, NK / GS>{});
            // ---- chunk complete: every lane drops its 16 correlations (one box position, 16 source pixels) into
            //      the windows that contain that position (zero outside the map = never written) ----
            {
                const int pos = c0 + wn * 32 + r32;
                const int by = pos / bw, bx = pos - by * bw;
                const int tx = bx0 + bx, ty = by0 + by;  // target pixel of this column
                const bool col_ok = pos < N && !(p.ablate & 4);
                // window origins of the 16 source pixels this lane's accumulator holds: ALL read before the first window
                // cell is written (a read after a write to the same address space is not moved above it: the reads then
                // ran one at a time, each waiting out the LDS latency -- ~2.5 k cycles of every 7.6 k-cycle chunk)
                int2 w0[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) w0[r] = s_w0[wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    const int cx = tx - w0[r].x, cy = ty - w0[r].y;
                    if (col_ok && cx >= 0 && cx < WS && cy >= 0 && cy < WS) Wn[m * WLD + cy * WS + cx] = acc[r] * p.alpha;
                    acc[r] = 0.f;
                }
            }
            c0 += 64;
        }
        stamp();
        __syncthreads();
        // ---- bilinear samples from the pixel's own window: the arithmetic of corr_lookup_kernel ----
        // (4 threads per pixel, a sample at a time.  Batching the window reads, or a wave per pixel with lane = sample and
        // contiguous stores, measured the same or slower: 4.6 k / 8 k vs 3.6 k cycles per level)
        if (pvalid && !(p.ablate & 8)) {
            const float fx = s_fx[mypix], fy = s_fy[mypix];
            const float* wq = Wn + mypix * WLD;
            float* o = p.out + ((int64_t)gy * p.wf + gx) * p.ldo + l * N2;
#pragma unroll OTF_SAMPLE_UNROLL                        // (fully unrolled, its 21 x 2 hoisted offsets cost spills at 256 registers)
            for (int k = 0; k < NS; ++k) {
                const int s = part + 4 * k;
                if (s >= N2) break;
                const int i = (NW == 9) ? (s * 57) >> 9 : (s * 37) >> 8;   // = s / NW for s < NW^2 (NW = 9 / 7)
                const int j = s - i * NW;                           // i: x offset, j: y offset (x-major window)
                const float* q = wq + j * WS + i;
                const float top = q[0] * (1.f - fx) + q[1] * fx;
                const float bot = q[WS] * (1.f - fx) + q[WS + 1] * fx;
                o[s] = top * (1.f - fy) + bot * fy;
            }
        }
        stamp();
        __syncthreads();
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
    hipStream_t s = (hipStream_t)stream;
    // 8 x 8 source pixels per workgroup.  (TW = 16: 40 % less target-row traffic, but one 8-wave workgroup per CU and
    // 20 % more steps per workgroup: measured 110 vs 98 us at 1080p in round 1, -1.7 % frames/s in round 2 with two steps
    // per barrier -- the per-workgroup chain of K steps binds.)
    dim3 grid((unsigned)(((p.wf + 7) / 8) * ((p.hf + 7) / 8)));
#define OTF(T, RR, KK) woft_launch(0, corr_lookup_otf_kernel<T, RR, KK, 8>, grid, dim3(256), 0, s, p)
    if (p.k == 256 && p.radius == 4) {                                                                /* full model  */
        if (p.terms == 3) OTF(3, 4, 256); else if (p.terms == 0) OTF(0, 4, 256); else OTF(1, 4, 256);
    } else if (p.k == 128 && p.radius == 3) {                                                         /* small model */
        if (p.terms == 3) OTF(3, 3, 128); else if (p.terms == 0) OTF(0, 3, 128); else OTF(1, 3, 128);
    }
    else return WOFT_EINVAL;
#undef OTF
    return woft_launch_status();
}
