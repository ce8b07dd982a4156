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


// =====================================================================================================================
// Round 6: the same lookup as FOUR AUTONOMOUS WAVES (corr_lookup_otf_w_kernel).
//
// Why.  Ablations of the kernel above (profiles/r05_lookup_otf_timeline.txt): of its 77 us at 1080p only 16 are fragment
// reads + MFMAs and 11 window drops + sampling; 44 us are the SKELETON -- sixty K-step groups per workgroup, each a
// dma_wait + workgroup barrier + DMA re-issue, four per-level set-ups with their barriers, and an A-fragment prologue
// whose 64 row-strided loads per wave are bound by the texture addresser (32 cache lines per instruction).  The barriers
// exist because a 64-column chunk of target rows is shared by the two row halves (wm = 0 / 1) of the 2 x 2 wave grid.
//
// Here a wave owns ALL 64 source pixels of the block -- both 32-row MFMA tiles, 256 registers of A fragments: one wave
// per SIMD, 512 registers, one workgroup per CU -- so the target rows of a 32-column tile are read by exactly one wave.
// Each wave streams its own tiles through its own LDS ring, ordered by nothing but its own vmcnt: NO barrier in the
// stream (tools/micro/dma_issue_probe.hip: four 1-KiB pieces per 12 MFMAs cost a lone wave 437 instead of 386 cycles --
// when at least two 4-KiB steps are in flight; an L2-resident piece takes ~1 000 cycles to land under load).  The tiles
// of all levels form one list dealt round-robin to the four waves (27 tiles -> 7 per wave; 32 instead of 64 columns of
// padding granularity).  The windows of TWO levels are resident (2 x 26 KB), so the list is worked in two phases --
// levels 0-1, then 2-3 -- with the sampling of a phase between them; the ring takes the rest of the LDS (NSTG stages of
// 4 KB per wave).  The source rows reach the registers through LDS (64 full-line LDS-DMA pieces into the not yet used
// window area, then 64 conflict-free ds_read_b128 per wave) instead of 128 row-strided global loads per wave.  Windows
// are stored x-major, so that a sampling thread reads the two window columns of an output column as one run of 2 WS
// floats, interpolates horizontally once per cell pair, and writes 2R + 1 consecutive outputs.  Same products, same
// order, same window arithmetic: bit-identical to the kernel above and to the volume lookup.
//   PIPE = 0: the window drop of a tile after its last MFMA; 2: the drop of tile i - 1 in slices between the MFMAs of tile i.
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

// the four 1-KiB pieces of one 4-KiB ring stage with ONE M0 set-up: the instruction's immediate offset moves the LDS destination AND the
// memory address by 1024 q, so the scalar base is passed 3072 bytes low and lane offset q carries + (3 - q) * 1024
__device__ __forceinline__ void lds_dma16x4(const void* gbase_minus_3072, uint32_t o0, uint32_t o1, uint32_t o2, uint32_t o3, uint32_t lds_addr) {
    const uint64_t gb = (uint64_t)(uintptr_t)gbase_minus_3072;
    const uint32_t g_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(gb >> 32));
    const uint32_t g_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)gb);
    const uint64_t gu = ((uint64_t)g_hi << 32) | (uint64_t)g_lo;
    uint32_t saved_m0;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %2, %6\n\tglobal_load_lds_dwordx4 %3, %6 offset:1024\n\t"
                 "global_load_lds_dwordx4 %4, %6 offset:2048\n\tglobal_load_lds_dwordx4 %5, %6 offset:3072\n\ts_mov_b32 m0, %0"
                 : "=&s"(saved_m0)
                 : "s"(lds_addr), "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(gu)
                 : "memory", "vcc");
}

template <int TERMS, int R, int K, int NSTG, int PIPE, int ABL>
__global__ __launch_bounds__(256, 1) void corr_lookup_otf_w_kernel(const woft_lookup_otf_params p) {
    constexpr int NPX = 64, NT = 256, MAXL = 4, NWB = 2;
    constexpr int NW = 2 * R + 1, N2 = NW * NW;
    constexpr int NPL = (TERMS == 1) ? 1 : 2;
    constexpr int LD = (TERMS == 1) ? K : 2 * K;         // bf16-sized elements per operand row
    constexpr int NK = LD / 64;                          // K steps (one 128-byte line each)
    constexpr int NSUB = (TERMS == 1) ? 4 : 2;           // MFMA k sub-steps per line
    constexpr int WS = NW + 1, WLD = WS * WS + 2;        // a window: WS x WS cells, x-major (cell (cx, cy) at cx * WS + cy); even stride
    constexpr int STG = 2048;                            // bf16 elements per ring stage: 32 rows x 128 B
    constexpr int WN_FLOATS = NWB * NPX * WLD;
    constexpr int A_FLOATS = NPX * NK * 32;              // the staged source rows: 64 rows x NK lines of 128 B
    // the source rows are staged where the windows and the first ring stages will be: one array [windows | ring]
    constexpr int RING_FLOATS = 4 * NSTG * STG / 2;
    static_assert(A_FLOATS <= WN_FLOATS + RING_FLOATS, "the source rows are staged in the window + ring area");
    static_assert(WN_FLOATS % 4 == 0, "ring 16-byte aligned behind the windows");
    __shared__ __attribute__((aligned(16))) float lds_main[WN_FLOATS + RING_FLOATS];
    float* const Wn = lds_main;
    __bf16* const ring = (__bf16*)(lds_main + WN_FLOATS);
    __shared__ __attribute__((aligned(16))) int s_org[MAXL][NPX];   // window origin clamped to 16 bits and packed (x | y << 16): drop test
    __shared__ __attribute__((aligned(16))) int s_pm[MAXL][NPX];    // float index of window cell (0, 0) minus (ox * WS + oy): drop address
    __shared__ float s_fx[MAXL][NPX], s_fy[MAXL][NPX];
    __shared__ int s_box[MAXL][4];                       // unclipped bounding box of the window origins: min x, max x, min y, max y
    __shared__ float s_trash[64];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31, hh = lane >> 5;
    const int levels = p.levels;
    const int tiles_x = (p.wf + 7) / 8, ntiles = tiles_x * ((p.hf + 7) / 8);
    int tile;
    {
        const int q = ntiles / 8, rr = ntiles % 8, xcd = blockIdx.x % 8, idx = blockIdx.x / 8;
        tile = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
    }
    const int px0 = (tile % tiles_x) * 8, py0 = (tile / tiles_x) * 8;
    if (p.need != nullptr) {
        int any = 0;
        if (tid < NPX) {
            const int y = py0 + (tid >> 3), x = px0 + (tid & 7);
            if (y < p.hf && x < p.wf) any = p.need[y * p.wf + x];
        }
        if (!__syncthreads_or(any)) return;
    }
    uint32_t* stamps = ((ABL & 16) && lane == 0 && p.ldo >= 4 * N2 + 24 && py0 + wave < p.hf)
                           ? (uint32_t*)(p.out + ((int64_t)(py0 + wave) * p.wf + px0) * p.ldo + 4 * N2) : nullptr;
    int n_stamp = 0;
    auto stamp = [&]() __attribute__((always_inline)) { if ((ABL & 16) && stamps && n_stamp < 24) stamps[n_stamp++] = (uint32_t)__builtin_amdgcn_s_memtime(); };
    stamp();

    // ---- source rows -> LDS, full 128-byte lines by LDS-DMA, XOR-swizzled like the target stages ----
    const uint32_t main_addr = lds_addr_of(lds_main);
    {
        const int swz = (4 * (wave & 1) + (lane >> 4)) & 7;      // (row >> 1) & 7 of row 8 q + lane / 8 for q = wave (mod 2)
        const uint32_t chunk_sw = (uint32_t)(((lane & 7) ^ swz) * 16);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int q = wave + 4 * t;                  // rows 8 q .. 8 q + 7 of the block
            const int m = q * 8 + (lane >> 3);
            int y = py0 + (m >> 3), x = px0 + (m & 7);
            y = y < p.hf ? y : p.hf - 1;                 // rows outside the grid repeat a valid pixel (never written)
            x = x < p.wf ? x : p.wf - 1;
            const uint32_t off = (uint32_t)(y * p.wf + x) * (uint32_t)(LD * 2) + chunk_sw;
#pragma unroll
            for (int ks = 0; ks < NK; ++ks)
                lds_dma16((const char*)p.f1 + ks * 128, off, main_addr + (uint32_t)(ks * (NPX * 128) + q * 1024));
        }
    }
    // ---- wave 0: lookup centres (+ the folded flow-head gather of the previous iteration), window origins and boxes of all levels ----
    if (tid < NPX) {
        const int y = py0 + (tid >> 3), x = px0 + (tid & 7);
        float cx = 0.f, cy = 0.f;
        const bool cvalid = y < p.hf && x < p.wf;
        if (cvalid) {
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
#pragma unroll
        for (int l = 0; l < MAXL; ++l) {
            if (l < levels) {
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
                s_fx[l][tid] = fx;
                s_fy[l][tid] = fy;
                // drop test operand: a position (tx, ty) of the map (0 <= tx, ty < 2^14) lies in the window iff both 16-bit halves of
                // (tx | ty << 16) - org are below WS; origins beyond +-2^14 are clamped (their windows contain no map position)
                const int ox = wx0 < -16384 ? -16384 : (wx0 > 16384 ? 16384 : wx0);
                const int oy = wy0 < -16384 ? -16384 : (wy0 > 16384 ? 16384 : wy0);
                s_org[l][tid] = cvalid ? ((ox & 0xffff) | (oy << 16)) : 0x40004000;      // (pixels outside the grid: never inside)
                s_pm[l][tid] = ((l & (NWB - 1)) * NPX + tid) * WLD - (ox * WS + oy);
                // bounding box of the valid pixels' origins (pixels outside the grid: neutral)
                const int b0 = wave_reduce_minmax<false>(cvalid ? wx0 : 0x3fffffff), b1 = wave_reduce_minmax<true>(cvalid ? wx0 : -0x3fffffff);
                const int b2 = wave_reduce_minmax<false>(cvalid ? wy0 : 0x3fffffff), b3 = wave_reduce_minmax<true>(cvalid ? wy0 : -0x3fffffff);
                if (tid == 0) { s_box[l][0] = b0; s_box[l][1] = b1; s_box[l][2] = b2; s_box[l][3] = b3; }
            }
        }
    }
    dma_wait<0>();
    __syncthreads();
    stamp();
    // ---- A fragments of both 32-row tiles: LDS -> registers (lane (r32, hh): row r32 of the tile, its 16-byte share of every line) ----
    bf16x8 afr[2][NK][NSUB][NPL];
    const int sw = (r32 >> 1) & 7;
    {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                const __bf16* ar = (const __bf16*)lds_main + ks * (NPX * 64) + (t * 32 + r32) * 64;
#pragma unroll
                for (int s2 = 0; s2 < NSUB; ++s2) {
                    if (TERMS == 0) {
                        afr[t][ks][s2][0] = *(const bf16x8*)(ar + ((hh * 4 + s2 * 2) ^ sw) * 8);
                        afr[t][ks][s2][NPL - 1] = *(const bf16x8*)(ar + ((hh * 4 + s2 * 2 + 1) ^ sw) * 8);
                    } else {
                        afr[t][ks][s2][0] = *(const bf16x8*)(ar + ((s2 * 2 + hh) ^ sw) * 8);
                        if (TERMS == 3) afr[t][ks][s2][NPL - 1] = *(const bf16x8*)(ar + ((4 + s2 * 2 + hh) ^ sw) * 8);
                    }
                }
            }
    }
    // ---- boxes of all levels (wave 0 left them in LDS), tile list ----
    int bx0_[MAXL], by0_[MAXL], bw_[MAXL], N_[MAXL], T_[MAXL + 1];
    bool clip_[MAXL];
    T_[0] = 0;
#pragma unroll
    for (int l = 0; l < MAXL; ++l) {
        bx0_[l] = 0; by0_[l] = 0; bw_[l] = 1; N_[l] = 0; clip_[l] = false;
        if (l < levels) {
            int bx0 = __builtin_amdgcn_readfirstlane(s_box[l][0]), bx1 = __builtin_amdgcn_readfirstlane(s_box[l][1]);
            int by0 = __builtin_amdgcn_readfirstlane(s_box[l][2]), by1 = __builtin_amdgcn_readfirstlane(s_box[l][3]);
            const int W = p.w[l], H = p.h[l];
            clip_[l] = bx0 < 0 || by0 < 0 || bx1 + NW > W - 1 || by1 + NW > H - 1;
            bx0 = bx0 > 0 ? bx0 : 0;
            by0 = by0 > 0 ? by0 : 0;
            bx1 = (bx1 + NW < W - 1) ? bx1 + NW : W - 1;       // windows span wx0 .. wx0 + 2R + 1
            by1 = (by1 + NW < H - 1) ? by1 + NW : H - 1;
            const int bw = bx1 - bx0 + 1, bh = by1 - by0 + 1;
            bx0_[l] = bx0; by0_[l] = by0;
            if (bw > 0 && bh > 0) { bw_[l] = bw; N_[l] = bw * bh; }
        }
        T_[l + 1] = T_[l] + (N_[l] + 31) / 32;
    }
    const int ntl = T_[MAXL];                            // tiles of the block; this wave's: wave, wave + 4, ...
    const int my_tiles = ntl > wave ? (ntl - wave + 3) / 4 : 0;
    const int tiles_ph0 = T_[2] > wave ? (T_[2] - wave + 3) / 4 : 0;     // ... of which the first belong to levels 0 - 1
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();                                     // every wave has its A fragments: windows and ring may be used
    stamp();
    // cells that correspond to positions outside the map are never written and must read as zero: only levels whose box was
    // clipped at a border have any
    auto clear_windows = [&](int ph) __attribute__((always_inline)) {                   // -> whether anything was cleared (uniform over the workgroup)
        const bool c0 = ph == 0 ? clip_[0] : clip_[2], c1 = ph == 0 ? clip_[1] : clip_[3];
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        static_assert((NPX * WLD) % 4 == 0, "a level's windows are cleared 16 bytes at a time");
        if (c0) for (int i = tid; i < NPX * WLD / 4; i += NT) ((f32x4*)Wn)[i] = z4;
        if (c1) for (int i = tid; i < NPX * WLD / 4; i += NT) ((f32x4*)(Wn + NPX * WLD))[i] = z4;
        return c0 || c1;
    };
    const bool cleared0 = clear_windows(0);
    // ---- this wave's stream of target rows: producer state = (tile, k step) of the next step to request ----
    const uint32_t ring_addr = lds_addr_of(ring) + (uint32_t)wave * (NSTG * STG * 2);
    const __bf16* b_rows = ring + wave * (NSTG * STG) + r32 * 64;
    const uint32_t chunk_e = (uint32_t)(((lane & 7) ^ ((lane >> 4) & 7)) * 16);          // pieces 0, 2: (row >> 1) & 7 = lane / 16
    const uint32_t chunk_o = (uint32_t)(((lane & 7) ^ ((4 + (lane >> 4)) & 7)) * 16);    // pieces 1, 3
    // (level-indexed quantities through select chains: a run-time index into an array of SGPR values -- or into the kernel
    //  argument struct -- would make the compiler keep a copy in scratch memory)
    auto sel4 = [](int l, int a0, int a1, int a2, int a3) __attribute__((always_inline)) { return l == 0 ? a0 : (l == 1 ? a1 : (l == 2 ? a2 : a3)); };
    auto tile_level = [&](int g) __attribute__((always_inline)) { return (g >= T_[1] ? 1 : 0) + (g >= T_[2] ? 1 : 0) + (g >= T_[3] ? 1 : 0); };
    const int last_g = wave + 4 * (my_tiles - 1);
    uint32_t b_off[4] = {0u, 0u, 0u, 0u};
    const char* pf2 = (const char*)p.f2[0];
    int pr_i = 0, pr_k = 0, pr_s = 0;                    // producer: tile (of this wave), k step, step number
    auto new_tile = [&]() __attribute__((always_inline)) {                              // rows of the box positions c0 .. c0 + 31 of the producer's tile
        int g = wave + 4 * pr_i;
        g = g < last_g ? g : last_g;                     // (requests past the end of the stream repeat the last tile: harmless)
        const int l = tile_level(g);
        const int c0 = (g - sel4(l, T_[0], T_[1], T_[2], T_[3])) * 32;
        const int bw = sel4(l, bw_[0], bw_[1], bw_[2], bw_[3]), N = sel4(l, N_[0], N_[1], N_[2], N_[3]);
        const int W = sel4(l, p.w[0], p.w[1], p.w[2], p.w[3]);
        const int bx0 = sel4(l, bx0_[0], bx0_[1], bx0_[2], bx0_[3]), by0 = sel4(l, by0_[0], by0_[1], by0_[2], by0_[3]);
        pf2 = (const char*)(l == 0 ? p.f2[0] : (l == 1 ? p.f2[1] : (l == 2 ? p.f2[2] : p.f2[3])));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int pos = c0 + q * 8 + (lane >> 3);
            pos = pos < N ? pos : N - 1;                 // (columns past the box repeat its last position; never used)
            const int by = pos / bw, bx = pos - by * bw;
            // (+ (3 - q) KiB: the scalar base is passed 3 KiB low and piece q's immediate offset adds q KiB -- lds_dma16x4)
            b_off[q] = (uint32_t)((by0 + by) * W + bx0 + bx) * (uint32_t)(LD * 2) + ((q & 1) ? chunk_o : chunk_e) + (uint32_t)(3 - q) * 1024u;
        }
    };
    auto issue_pieces = [&]() __attribute__((always_inline)) {                          // the four 1-KiB pieces of the producer's step, then advance it
        const uint32_t st = ring_addr + (uint32_t)(pr_s % NSTG) * (STG * 2);
        if (!(ABL & 1) || pr_s < NSTG) lds_dma16x4(pf2 + pr_k * 128 - 3072, b_off[0], b_off[1], b_off[2], b_off[3], st);
        ++pr_s;
        if (++pr_k == NK) { pr_k = 0; ++pr_i; }
    };
    if (my_tiles > 0) {                                  // NSTG - 1 steps ahead, always exactly that many (the waits below count on it)
        for (int e = 0; e < NSTG - 1; ++e) {
            if (pr_k == 0) new_tile();
            issue_pieces();
        }
    }
    if (cleared0) __syncthreads();                       // the cleared windows are visible before the first drop
    stamp();

    f32x16 acc[2], prev[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[t][r] = 0.f; prev[t][r] = 0.f; }
    // drop operands of a finished tile: its level, this lane's map position (packed halves / in window-cell units), column valid
    int d_l = 0, d_pos = 0, d_q = 0;
    bool d_ok = false;
    // The window drop of a tile in 40 SLICES: 8 batches (row tile t = batch / 4, group g4 = batch % 4: accumulator elements 4 g4 .. 4 g4 + 3
    // = source pixels m0 .. m0 + 3) of one operand fetch + four values.  Branch-free (a conditional store is control flow and would cut
    // the MFMA stream into scheduling regions): positions outside a pixel's window go to a per-lane dummy cell.  Inside <=> both
    // 16-bit halves of (pos - org) are below WS <=> min(half, WS - 1) == half for both (v_pk_sub_u16, v_pk_min_u16).
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    i32x4 org_r = {0, 0, 0, 0}, pm_r = {0, 0, 0, 0};
    auto drop_slice = [&](const f32x16 (&a)[2], auto k_tag) __attribute__((always_inline)) {
        constexpr int k = decltype(k_tag)::value, batch = k / 5, part = k % 5, t = batch >> 2, g4 = batch & 3;
        if constexpr (part == 0) {
            const int m0 = t * 32 + 4 * hh + 8 * g4;
            org_r = *(const i32x4*)&s_org[d_l][m0];
            pm_r = *(const i32x4*)&s_pm[d_l][m0];
        } else {
            constexpr int e = part - 1;
            const u16x2 lim = {(unsigned short)(WS - 1), (unsigned short)(WS - 1)};
            const u16x2 dd = __builtin_bit_cast(u16x2, d_pos) - __builtin_bit_cast(u16x2, (int)org_r[e]);
            const u16x2 mn = __builtin_elementwise_min(dd, lim);
            const bool inside = d_ok && __builtin_bit_cast(uint32_t, mn) == __builtin_bit_cast(uint32_t, dd);
            float* cell = inside ? (Wn + (pm_r[e] + d_q)) : (s_trash + lane);
            *cell = a[t][4 * g4 + e] * p.alpha;
        }
    };
    auto drop_all = [&](const f32x16 (&a)[2]) __attribute__((always_inline)) {
        [&]<int... KK>(std::integer_sequence<int, KK...>) { (drop_slice(a, std::integral_constant<int, KK>{}), ...); }(std::make_integer_sequence<int, 40>{});
    };
    auto finish_tile = [&](int g) __attribute__((always_inline)) {                      // this lane's column of tile g -> drop operands
        const int l = tile_level(g);
        const int c0 = (g - sel4(l, T_[0], T_[1], T_[2], T_[3])) * 32;
        const int bw = sel4(l, bw_[0], bw_[1], bw_[2], bw_[3]);
        const int pos = c0 + r32;
        const int by = pos / bw, bx = pos - by * bw;
        const int tx = sel4(l, bx0_[0], bx0_[1], bx0_[2], bx0_[3]) + bx, ty = sel4(l, by0_[0], by0_[1], by0_[2], by0_[3]) + by;
        d_l = l;
        d_pos = (tx & 0xffff) | (ty << 16);
        d_q = tx * WS + ty;
        d_ok = pos < sel4(l, N_[0], N_[1], N_[2], N_[3]);
    };
    // ---- bilinear samples of the two resident levels (2 ph, 2 ph + 1), the arithmetic of corr_lookup_kernel.  An item = one window
    //      column i of one pixel and level: the cells of columns i and i + 1 are one run of 2 WS floats (x-major windows); the horizontal
    //      interpolation of a cell pair is shared by the two outputs that use it; outputs i * NW .. i * NW + NW - 1 are consecutive ----
    auto sample_phase = [&](int ph) __attribute__((always_inline)) {
        if (ABL & 8) return;
        const int nl = (2 * ph + 1 < levels) ? 2 : ((2 * ph < levels) ? 1 : 0);
        const int items = nl * NPX * NW;
        for (int it = tid; it < items; it += NT) {
            const int lb = it / (NPX * NW), rem = it - lb * (NPX * NW);      // level within the phase = window buffer
            const int pix = rem / NW, i = rem - pix * NW;
            const int gy = py0 + (pix >> 3), gx = px0 + (pix & 7);
            if (gy >= p.hf || gx >= p.wf) continue;
            const int l = 2 * ph + lb;
            const float fx = s_fx[l][pix], fy = s_fy[l][pix];
            const float* cq = Wn + (lb * NPX + pix) * WLD + i * WS;
            float h[WS];
#pragma unroll
            for (int j = 0; j < WS; ++j) h[j] = cq[j] * (1.f - fx) + cq[WS + j] * fx;
            float* o = p.out + ((int64_t)gy * p.wf + gx) * p.ldo + l * N2 + i * NW;
#pragma unroll
            for (int j = 0; j < NW; ++j) o[j] = h[j] * (1.f - fy) + h[j + 1] * fy;
        }
    };
    // The stream as one list of k sub-steps (NSUB per K step, NK K steps per tile, tile after tile).  The B fragments of sub-step
    // u + 1 are requested BEFORE the MFMAs of sub-step u (two register sets) and the order is pinned: left alone the compiler sinks
    // every ds_read next to its use and waits out the LDS latency in front of each MFMA group.  Per K step: its first sub-step
    // requests step + NSTG - 1 into the stage that step - 1 has finished reading; its last one waits until step + 1 has landed (at
    // most the 4 (NSTG - 2) pieces of the younger steps still in flight) before it reads that step's first fragments.
    static_assert(NSTG >= 3, "ring: the stage being read, the next one landed, one in flight");
    static_assert((NK * NSUB) % 2 == 0, "fragment register sets alternate per sub-step across tiles");
    bf16x8 bq[2][NPL];
    auto load_b = [&](int st, auto s2_tag, auto set_tag) __attribute__((always_inline)) {
        constexpr int s2 = decltype(s2_tag)::value, set = decltype(set_tag)::value;
        const __bf16* br = b_rows + (st % NSTG) * STG;
        if (TERMS == 0) {
            bq[set][0] = *(const bf16x8*)(br + ((hh * 4 + s2 * 2) ^ sw) * 8);
            bq[set][NPL - 1] = *(const bf16x8*)(br + ((hh * 4 + s2 * 2 + 1) ^ sw) * 8);
        } else {
            bq[set][0] = *(const bf16x8*)(br + ((s2 * 2 + hh) ^ sw) * 8);
            if (TERMS == 3) bq[set][NPL - 1] = *(const bf16x8*)(br + ((4 + s2 * 2 + hh) ^ sw) * 8);
        }
    };
    // between the two phases: every wave's drops of levels 0 - 1 are in the windows -> sample them -> the windows are free for levels 2 - 3
    auto phase_boundary = [&]() __attribute__((always_inline)) {
        if (PIPE && !(ABL & 4)) { drop_all(prev); d_ok = false; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
        sample_phase(0);
        __syncthreads();
        if (clear_windows(1)) __syncthreads();
    };
    int step = 0;
    if (my_tiles > 0) {
        dma_wait<4 * (NSTG - 2)>();                      // step 0 has landed
        load_b(0, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    }
    stamp();
    for (int i = 0; i < my_tiles; ++i) {
        if (i == tiles_ph0) { phase_boundary(); stamp(); }
        [&]<int... US>(std::integer_sequence<int, US...>) {
        ([&] {
            constexpr int u = US, ks = u / NSUB, s2 = u % NSUB;
            if constexpr (s2 == 0) {
                if constexpr ((ks + NSTG - 1) % NK == 0) new_tile();      // (S is a multiple of NK: the producer's k step is known here)
                issue_pieces();
            }
            if constexpr (s2 == NSUB - 1) dma_wait<4 * (NSTG - 2)>();
            __builtin_amdgcn_sched_barrier(0);
            load_b(step + (s2 == NSUB - 1 ? 1 : 0), std::integral_constant<int, (s2 + 1) % NSUB>{}, std::integral_constant<int, (u + 1) & 1>{});
            // the sub-step's MFMAs one by one.  TERMS = 3: per accumulator the order of corr_gemm_bf16_kernel (lo x hi, hi x lo,
            // hi x hi), the two row tiles alternating; TERMS = 0: per row tile the 4 + 4 k pairs in conv_mfma_f32_kernel's order.
            // PIPE = 2: the slices of the PREVIOUS tile's window drop are dealt over the MFMAs of this tile and pinned behind them, so
            // that the vector / LDS work issues while the matrix pipe runs.
            constexpr int MPS = (TERMS == 3) ? 6 : (TERMS == 1 ? 2 : 16), MT = NK * NSUB * MPS;
            [&]<int... JJ>(std::integer_sequence<int, JJ...>) {
            ([&] {
                constexpr int j = JJ;
                if (!(ABL & 2)) {
                    if constexpr (TERMS == 3) {
                        constexpr int t = j & 1, term = j >> 1;          // 0: lo x hi, 1: hi x lo, 2: hi x hi
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[t][ks][s2][term == 0 ? NPL - 1 : 0],
                                                                         bq[u & 1][term == 1 ? NPL - 1 : 0], acc[t], 0, 0, 0);
                    } else if constexpr (TERMS == 1) {
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[j][ks][s2][0], bq[u & 1][0], acc[j], 0, 0, 0);
                    } else {
                        constexpr int t = j >> 3, half = (j >> 2) & 1, e = j & 3;
                        const f32x4 av = __builtin_bit_cast(f32x4, afr[t][ks][s2][half ? NPL - 1 : 0]);
                        const f32x4 bv = __builtin_bit_cast(f32x4, bq[u & 1][half ? NPL - 1 : 0]);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[e], acc[t], 0, 0, 0);
                    }
                }
                if constexpr (PIPE == 2 && !(ABL & 4)) {
                    constexpr int gm = u * MPS + j, K0 = (40 * gm) / MT, K1 = (40 * (gm + 1)) / MT;
                    [&]<int... KK>(std::integer_sequence<int, KK...>) {
                        (drop_slice(prev, std::integral_constant<int, K0 + KK>{}), ...);
                    }(std::make_integer_sequence<int, K1 - K0>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
            }(), ...);
            }(std::make_integer_sequence<int, MPS>{});
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (s2 == NSUB - 1) ++step;
        }(), ...);
        }(std::make_integer_sequence<int, NK * NSUB>{});
        // tile complete
        if (PIPE) {
#pragma unroll
            for (int t = 0; t < 2; ++t) prev[t] = acc[t];
            finish_tile(wave + 4 * i);
        } else if (!(ABL & 4)) {
            finish_tile(wave + 4 * i);
            drop_all(acc);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    }
    if (my_tiles <= tiles_ph0) { phase_boundary(); stamp(); }        // (this wave has no tile in levels 2 - 3: the boundary still is everybody's)
    if (PIPE && !(ABL & 4) && my_tiles > 0) drop_all(prev);
    dma_wait<0>();
    stamp();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    stamp();
    sample_phase(1);
    stamp();
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
    for (int l = 0; l < p.levels; ++l)
        if (p.h[l] > 16384 || p.w[l] > 16384) return WOFT_EINVAL;                       // (packed 16-bit map positions in the drop test)
    const int variant = (p.ablate >> 8) & 0xff, abl = p.ablate & 0xff;
    if (variant != 0) {                                 // round 6: four autonomous waves per 8 x 8 block (1: drop after the tile, 2: piped)
        for (int l = 0; l < p.levels; ++l)                                              // (lane offsets carry up to + 3 KiB: lds_dma16x4)
            if ((int64_t)p.h[l] * p.w[l] * row_bytes >= (1ll << 32) - 4096) return WOFT_EINVAL;
        dim3 gridw((unsigned)(((p.wf + 7) / 8) * ((p.hf + 7) / 8)));
#define OTFW(T, RR, KK, NS, PP, AB) woft_launch(0, corr_lookup_otf_w_kernel<T, RR, KK, NS, PP, AB>, gridw, dim3(256), 0, s, p)
        if (p.k == 256 && p.radius == 4 && p.terms == 3) {
            // variants (developer): 1 / 3 = drop after the tile / in slices between the next tile's MFMAs, 6-stage rings; 4, 5 = as 3 with 4 / 5 stages
            if (variant == 1) {
                switch (abl) {
                    case 0: OTFW(3, 4, 256, 6, 0, 0); break;
                    case 16: OTFW(3, 4, 256, 6, 0, 16); break;
                    case 2: OTFW(3, 4, 256, 6, 0, 2); break;
                    case 6: OTFW(3, 4, 256, 6, 0, 6); break;
                    case 14: OTFW(3, 4, 256, 6, 0, 14); break;
                    case 15: OTFW(3, 4, 256, 6, 0, 15); break;
                    default: return WOFT_EINVAL;
                }
            } else if (variant == 3) {
                switch (abl) {
                    case 0: OTFW(3, 4, 256, 6, 2, 0); break;
                    case 16: OTFW(3, 4, 256, 6, 2, 16); break;
                    case 8: OTFW(3, 4, 256, 6, 2, 8); break;
                    default: return WOFT_EINVAL;
                }
            } else if (variant == 4 && abl == 0) { OTFW(3, 4, 256, 4, 2, 0);
            } else if (variant == 5 && abl == 0) { OTFW(3, 4, 256, 3, 2, 0);
            } else return WOFT_EINVAL;
        } else return WOFT_EINVAL;
#undef OTFW
        return woft_launch_status();
    }
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
