// conv1x1_kernel (woft_conv_params.halo == 16): 1x1 / stride-1 convolution = a plain GEMM out[m][n] = sum_k x[m][k] w[n][k] over
// the pixels m, for the wide 1x1 layers of the hot path -- the motion encoder's convc1 (324 lookup samples -> 256, update.py:89),
// the encoders' closing 1x1 (128 -> 256, extractor.py:166).
//
// Why a kernel of its own: the per-tap gather kernel (conv.hip: conv_mfma_bf16_kernel) runs these layers on 64 x 64 tiles -- every
// fp32 activation row is fetched and split into its bf16 terms by each of the cout / 64 column-tile workgroups, one K step ahead,
// with a workgroup barrier per step and both operand tiles through LDS (15-17 % MFMA-busy, profiles/r05_pmc_mfma_util_conv.csv).
// Here a workgroup owns 64 pixels x ALL 128 TN output columns (TN = 2: 256): the activations are read and converted ONCE per
// layer; each of the four waves owns a 32 TN-column band for all 64 rows, so its weights are nobody else's -- streamed global ->
// registers in MFMA-fragment order (the register-streamed conv kernel's packing, conv_regb_body.h) NB half-chunks ahead, no LDS,
// no barrier; the activation tile of a 32-channel chunk goes through a double-buffered LDS tile (one barrier per chunk) and is
// requested DEPTH chunks ahead in registers: a 1x1 layer has only 24 MFMAs per wave and chunk to hide a load behind, and its
// input is usually the previous launch's output on its way in from beyond L2.
//
// FLAT (round 5): the same tile loop for a "flat"-packed tiny-Cin layer (woft_conv_params.flat: the K chunk of tap row ky is the NHWC
// row itself, 32 floats from pixel ox - pad_x on; the motion encoder's convf1, 7x7 on the 2-channel flow, update.py:91) -- K chunks =
// tap rows, validity per pixel of the row -- so that convc1 and convf1, which are independent, share ONE launch (woft_conv2d_pair)
// as they did on the gather kernel.  Measured at 1/8 of 1080p (bf16x3): convc1 alone 29.7 us (gather kernel 37), the pair 39.8-42.4
// (45.5); ablations of convc1 alone (tools/regb_probe.py c1, ABL bits): no output stores -7.6 us (507 workgroups = ONE round in
// lock-step: all of them store their 33 MB at the same time), no weight stream -3.9, no activation loads -2.6, none of the three
// 17.6 us -- 11 chunks at ~1.1 us of barrier + fragment-read latency + 24 MFMAs each.  Tried and not kept: the flat tile and the
// wide tile of the same pixels in ONE workgroup (the short tile's stores under the long tile's K loop): 41.6 vs 39.8 us as two
// groups of workgroups; the shorter layer's workgroups first or last: the same; two chunks per workgroup barrier (four LDS
// sub-buffers): 32.7 vs 31.9 us, 18.0 vs 17.6 without global traffic -- the barrier is not what a chunk costs.
//
// Arithmetic: operands, products and their order as in the other split-bf16 kernels (TERMS 3: lo*hi, hi*lo, hi*hi per k16 step,
// k ascending; fp32 accumulation) -- the result equals the gather kernel's bit for bit (tests/test_kernels_gpu.py).
#include "conv_common.h"

namespace {

using woft::BK;

template <int TERMS, int TN>
struct Geom1x1 {
    static constexpr int NP = (TERMS == 3) ? 2 : 1;
    static constexpr int BM = 64, TM = 2;
    static constexpr int WCOLS = 32 * TN, BN = 4 * WCOLS;
    static constexpr int A_PLANE = BM * LDB, A_ELEMS = NP * A_PLANE;      // one activation buffer (bf16 elements)
    static constexpr int ST = 2;                                          // accumulator tiles staged at once in the epilogue
    static constexpr int STAGE_ELEMS = 2 * 4 * ST * woft::STAGE_FLOATS;
    static constexpr int SMEM_ELEMS = (2 * A_ELEMS > STAGE_ELEMS) ? 2 * A_ELEMS : STAGE_ELEMS;
};

// One 64-pixel x (128 TN)-column output tile.  DEPTH: chunks of activations in flight in registers (the chunk loop is unrolled
// DEPTH times: static register-ring slots); NB: half-chunk (k16) weight steps in the register ring, NB - 1 of them in flight.
template <int TERMS, int TN, bool FLAT, int DEPTH, int NB>
__device__ __forceinline__ void tile_1x1(const woft_conv_params& p, const int m_tile, const int n_tile, __bf16* smem) {
    using G = Geom1x1<TERMS, TN>;
    constexpr int NP = G::NP, TM = G::TM, A_PLANE = G::A_PLANE, A_ELEMS = G::A_ELEMS, WCOLS = G::WCOLS, BN = G::BN;
    static_assert(DEPTH % 2 == 0 && (2 * DEPTH) % NB == 0 && NB >= 2, "static ring slots");
    constexpr int STEP_ELEMS = NP * 2 * 64 * 8;                           // fragment elements of one 32-channel chunk of a band

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31, hh = lane >> 5;
    const int row = tid >> 2, v = tid & 3;                                // loader: pixel row of the tile, 8-channel group of the chunk

    const int64_t M = (int64_t)p.n_img * p.ho * p.wo;
    const int64_t m0 = (int64_t)m_tile * G::BM;
    const int n0 = n_tile * BN;
    const int nchunk = FLAT ? p.taps_y : p.cin_pad / BK;                   // K chunks of 32 (FLAT: one per tap row)

    // developer ablations (tools/regb_probe.py, timing only): out_w = -12350 - bits; 1: no activation loads behind the prologue,
    // 2: no weight loads behind the prologue, 4: no output stores
    const int abl = (p.out_w <= -12350 && p.out_w > -12358) ? -12350 - p.out_w : 0;
    const bool rok = m0 + row < M;
    const uint32_t pix = rok ? (uint32_t)(m0 + row) : 0u;
    // FLAT: output pixel (oy, ox) of this thread's row; its two float4s of a chunk are the input pixels (floats) 8 v .. 8 v + 7 of the
    // 32-float run that starts at input pixel ox - pad_x of input row oy + chunk - pad_y
    int oy = 0, img_row0 = 0;
    uint32_t run0 = 0;                                                    // element offset of the run inside its input row
    bool okx[2] = {true, true};
    if constexpr (FLAT) {
        const int hw = p.ho * p.wo;
        const int img = (int)(pix / (uint32_t)hw), rem = (int)(pix - (uint32_t)img * (uint32_t)hw);
        oy = rem / p.wo;
        const int ox = rem - oy * p.wo;
        img_row0 = img * p.h;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int ixp = ox - p.pad_x + (8 * v + 4 * e) / p.cs0;
            okx[e] = rok && ixp >= 0 && ixp < p.w;
        }
        run0 = (uint32_t)((ox - p.pad_x) * p.cs0 + 8 * v);                 // (may wrap below zero: used only where okx)
    }
    f32x4 rh[DEPTH][2];                                                   // activation chunks in flight
    auto load_a = [&](int chunk, auto slot_tag) {
        constexpr int slot = decltype(slot_tag)::value;
        if ((abl & 1) && chunk >= DEPTH) return;
        const int ck = chunk < nchunk ? chunk : nchunk - 1;                // (past the end: a harmless repeat, never converted)
        if constexpr (FLAT) {
            int iy = oy + ck - p.pad_y;
            iy = iy < 0 ? 0 : (iy >= p.h ? p.h - 1 : iy);                  // (outside: any valid row -- replaced by zeros in store_a)
            const uint32_t rowoff = (uint32_t)(img_row0 + iy) * (uint32_t)(p.w * p.cs0);
#pragma unroll
            for (int e = 0; e < 2; ++e) rh[slot][e] = *(const f32x4*)(p.in0 + (okx[e] ? rowoff + run0 + 4 * e : 0u));
        } else {
            const int c0 = ck * BK;
            // two sources: the base pointer is one of the two kernel arguments as it stands and the channel offset goes into the
            // 32-bit lane offset (conv.hip's form).  `in1 + (c0 - c_split)` as a 64-bit base formed here is the expression conv.hip
            // documents as miscompiled on this toolchain (stale high half of the 64-bit shift): not used
            const bool second = (p.in1 != nullptr) && (c0 >= p.c_split);
            const float* base = second ? p.in1 : p.in0;
            const uint32_t cs = (uint32_t)(second ? p.cs1 : p.cs0);
            const uint32_t coff = (uint32_t)(second ? c0 - p.c_split : c0);
            rh[slot][0] = *(const f32x4*)(base + (pix * cs + coff + 8 * v));
            rh[slot][1] = *(const f32x4*)(base + (pix * cs + coff + 8 * v + 4));
        }
    };
    auto store_a = [&](int chunk, __bf16* As, auto slot_tag) {              // fp32 -> bf16 terms -> LDS
        constexpr int slot = decltype(slot_tag)::value;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        bool oky = true;
        if constexpr (FLAT) {
            const int iy = oy + chunk - p.pad_y;
            oky = iy >= 0 && iy < p.h;
        }
        bf16x4 hi[2], lo[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const f32x4 val = (FLAT ? (okx[e] && oky) : rok) ? rh[slot][e] : zero;
            hi[e] = cvt16<TERMS>(val);
            if (NP == 2) lo[e] = __builtin_convertvector(val - widen_bf16x4(hi[e]), bf16x4);
        }
        *(bf16x8*)(As + row * LDB + 8 * v) = __builtin_shufflevector(hi[0], hi[1], 0, 1, 2, 3, 4, 5, 6, 7);
        if (NP == 2) *(bf16x8*)(As + A_PLANE + row * LDB + 8 * v) = __builtin_shufflevector(lo[0], lo[1], 0, 1, 2, 3, 4, 5, 6, 7);
    };

    // this wave's weight streams: bands n0 / 32 + wave TN + j, chunk-major, STEP_ELEMS per chunk: [plane][k half][64 lanes][8]
    const __bf16* wstream[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) wstream[j] = (const __bf16*)p.wgt_frag + (int64_t)(n0 / 32 + wave * TN + j) * nchunk * STEP_ELEMS + lane * 8;
    bf16x8 bq[NB][TN][NP];
    auto fetch_b = [&](int step, auto slot_tag) {                          // step = 2 chunk + k half
        constexpr int slot = decltype(slot_tag)::value;
        if ((abl & 2) && step >= NB - 1) return;
        const int s = step < 2 * nchunk ? step : 2 * nchunk - 1;          // (past the end: a harmless repeat)
        const int off = (s >> 1) * STEP_ELEMS + (s & 1) * 512;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) bq[slot][j][pl] = *(const bf16x8*)(wstream[j] + off + pl * 1024);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int a_off = r32 * LDB + hh * 8;                                  // this lane's row of row tile 0, k half 0

    // ---- prologue: the first DEPTH activation chunks and NB - 1 weight steps requested; chunk 0 converted ------------
    [&]<int... D>(std::integer_sequence<int, D...>) {
        (load_a(D, std::integral_constant<int, D>{}), ...);
    }(std::make_integer_sequence<int, DEPTH>{});
    [&]<int... S>(std::integer_sequence<int, S...>) {
        (fetch_b(S, std::integral_constant<int, S>{}), ...);
    }(std::make_integer_sequence<int, NB - 1>{});
    store_a(0, smem, std::integral_constant<int, 0>{});
    __syncthreads();

    // chunk c (ring slot D = c % DEPTH, LDS buffer c & 1): the weights of k16 step 2 c + s2 + NB - 1 are requested, then the 12 (4)
    // MFMAs of step 2 c + s2 -- term-major: consecutive MFMAs never chain on one accumulator; under the second half's MFMAs chunk
    // c + 1 (requested DEPTH - 1 chunks ago) is converted into the other LDS buffer and chunk c + DEPTH requested into its registers
    auto run_chunk = [&](int c, auto d_tag) {
        constexpr int D = decltype(d_tag)::value;
        const __bf16* As = smem + (D & 1) * A_ELEMS;
        bf16x8 aq[2][TM][NP];
        auto read_a = [&](auto s2_tag) {
            constexpr int s2 = decltype(s2_tag)::value;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) aq[s2][i][pl] = *(const bf16x8*)(As + a_off + i * 32 * LDB + pl * A_PLANE + s2 * 16);
        };
        read_a(std::integral_constant<int, 0>{});
        read_a(std::integral_constant<int, 1>{});
        [&]<int... S2>(std::integer_sequence<int, S2...>) {
            ([&] {
                constexpr int s2 = S2, slot = (2 * D + s2) % NB;
                fetch_b(2 * c + s2 + NB - 1, std::integral_constant<int, (2 * D + s2 + NB - 1) % NB>{});
                __builtin_amdgcn_sched_barrier(0);
                if (NP == 2) {
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int i = 0; i < TM; ++i) acc[i][j] = mma16<TERMS>(aq[s2][i][NP - 1], bq[slot][j][0], acc[i][j]);
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int i = 0; i < TM; ++i) acc[i][j] = mma16<TERMS>(aq[s2][i][0], bq[slot][j][NP - 1], acc[i][j]);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int i = 0; i < TM; ++i) acc[i][j] = mma16<TERMS>(aq[s2][i][0], bq[slot][j][0], acc[i][j]);
                if constexpr (s2 == 0) {
                    // (between the halves, so that the conversion's vector / LDS instructions issue in the shadow of the MFMAs
                    //  around them; the registers of slot D are free: chunk c went to LDS one chunk ago
                    //  -- both unconditional, so that they stay in the MFMAs' scheduling region: behind the last chunk the buffer written
                    //  here is never read)
                    store_a(c + 1, smem + ((D + 1) & 1) * A_ELEMS, std::integral_constant<int, (D + 1) % DEPTH>{});
                    load_a(c + DEPTH, std::integral_constant<int, D>{});
                }
                __builtin_amdgcn_sched_barrier(0);
            }(), ...);
        }(std::make_integer_sequence<int, 2>{});
        __syncthreads();
    };
    for (int c0 = 0; c0 < nchunk; c0 += DEPTH) {
        [&]<int... D>(std::integer_sequence<int, D...>) {
            ((c0 + D < nchunk ? run_chunk(c0 + D, std::integral_constant<int, D>{}) : (void)0), ...);
        }(std::make_integer_sequence<int, DEPTH>{});
    }
    if (abl & 4) {
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
    woft::conv_epilogue_t<TM, TN, G::BM, WCOLS, G::ST>(p, acc, (float*)smem + wave * G::ST * woft::STAGE_FLOATS,
                                                        woft::LinearRows{m0, M}, n0, 0, wave, lane, m_tile);
}

// the two tile forms: 1x1 layers 256 columns per workgroup, flat layers 128

template <int TERMS>
__global__ __launch_bounds__(256, 2) void conv1x1_kernel(const woft_conv_params pa, const woft_conv_params pb, const int split) {
    // (two independent layers may share ONE launch -- woft_conv2d_pair: the workgroups [0, split) belong to the first layer, the rest
    //  to the second; split = gridDim.x for a single layer)
    __shared__ __attribute__((aligned(16))) __bf16 smem[Geom1x1<TERMS, 2>::SMEM_ELEMS];
    static_assert(Geom1x1<TERMS, 2>::SMEM_ELEMS >= Geom1x1<TERMS, 1>::SMEM_ELEMS, "one LDS array for every mode");
    const bool second_layer = (int)blockIdx.x >= split;
    const woft_conv_params p = second_layer ? pb : pa;     // (a copy: see conv_regb_kernel)
    const int bid = second_layer ? (int)blockIdx.x - split : (int)blockIdx.x;
    const int64_t M = (int64_t)p.n_img * p.ho * p.wo;
    int m_tile, n_tile;
    woft::tile_of_block(bid, (int)((M + 63) / 64), p.cout_pad / (p.flat ? 128 : 256), m_tile, n_tile);
    if (!p.flat) tile_1x1<TERMS, 2, false, 4, 4>(p, m_tile, n_tile, smem);
    else tile_1x1<TERMS, 1, true, 4, 4>(p, m_tile, n_tile, smem);
}

}  // namespace

// Called by woft_conv2d / woft_conv2d_pair for halo == 16, after their argument checks.  tile_n = 256 (the layer's columns in whole
// 256-wide tiles); flat layers: 128.
int woft_conv_1x1_launch(const woft_conv_params& a, const woft_conv_params* second, void* stream) {
    for (const woft_conv_params* q : {&a, second}) {
        if (q == nullptr) continue;
        const woft_conv_params& p = *q;
        if (p.precision < 1 || p.precision > 3 || p.precision != a.precision) return WOFT_EINVAL;
        if (p.stride != 1 || p.ho != p.h || p.wo != p.w || p.taps_x != 1) return WOFT_EINVAL;
        if (p.flat) {           // (woft_conv2d has checked cs0 in {4, 8, 16, 32} and in1 == NULL)
            if (p.cin_pad != 32 || p.tile_n != 128 || 2 * p.pad_y + 1 != p.taps_y) return WOFT_EINVAL;
        } else if (p.taps_y != 1 || p.pad_y != 0 || p.pad_x != 0 || p.tile_n != 256) {
            return WOFT_EINVAL;
        }
        if (p.wgt_frag == nullptr || p.stat_sum != nullptr || p.in_norm != 0 || p.wh0_lookup != nullptr) return WOFT_EINVAL;
        if (p.epi == WOFT_EPI_FLOWHEAD || p.epi == WOFT_EPI_WH_MEAN || p.epi == WOFT_EPI_CTX) return WOFT_EINVAL;
        if (p.cout_pad % p.tile_n != 0) return WOFT_EINVAL;
        const int64_t cs_max = (p.in1 != nullptr && p.cs1 > p.cs0) ? p.cs1 : p.cs0;
        if ((int64_t)p.n_img * p.h * p.w * cs_max >= (1ll << 31)) return WOFT_EINVAL;       // 32-bit element offsets
    }
    auto blocks = [](const woft_conv_params& q) {
        return ceil_div64((int64_t)q.n_img * q.ho * q.wo, 64) * (q.cout_pad / (q.flat ? 128 : 256));
    };
    const woft_conv_params& pb = second ? *second : a;
    const int split = (int)blocks(a);
    dim3 grid((unsigned)(blocks(a) + (second ? blocks(pb) : 0)));
    hipStream_t s = (hipStream_t)stream;
    switch (a.precision) {
        case 1: woft_launch(0, conv1x1_kernel<3>, grid, dim3(256), 0, s, a, pb, split); break;
        case 2: woft_launch(0, conv1x1_kernel<1>, grid, dim3(256), 0, s, a, pb, split); break;
        default: woft_launch(0, conv1x1_kernel<16>, grid, dim3(256), 0, s, a, pb, split); break;
    }
    return woft_launch_status();
}
