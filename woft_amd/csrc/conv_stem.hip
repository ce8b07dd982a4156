// The encoders' first layer (extractor.py:127-129,168: conv 7x7, stride 2, 3 -> 64 channels on the full-resolution image) as its
// own kernel -- on the per-tap gather kernel (conv_mfma_bf16_kernel, "flat" packing) it was 115-121 us of a 1080p frame for
// 29 GF of matrix work and 166 MB of traffic: seven K steps of one barrier each per 64-pixel tile, the activation rows gathered
// and converted once per tap row.
//
// Here a workgroup computes 8 x 16 OUTPUT pixels x 64 channels, STRIP tiles in a row:
//   * the (2*8+5) x (2*16+6)-pixel input patch (NHWC4 fp32) is read ONCE, converted to bf16 hi / lo planes in LDS; the
//     next tile's patch is requested (registers) before this tile's MFMAs;
//   * "flat" packing as the gather kernel's: the K chunk of tap row ky is 8 pixels x 4 channels of the image row itself,
//     k = 4 dx + c (the eighth pixel meets zero weights), so an A fragment is a 16-byte LDS read at column 2 x + 4 s2 + 2 hh
//     of patch row 2 y + ky -- the 16 pixels of a tile row read 16 consecutive slots (conflict-free);
//   * the whole weight matrix of the wave's 32 columns (7 x 32 x 32, hi + lo) lives in REGISTERS for all tiles;
//   * products in the gather kernel's order (tap rows, then the two 16-wide k sub-steps; lo*hi, hi*lo, hi*hi), the
//     accumulator rows follow the 8 x 16 tile map of the LDS-halo kernels (halo_map.h) and the shared epilogue
//     (InstanceNorm partial statistics, BatchNorm-folded bias + ReLU): every output value is bit-identical to the
//     gather kernel's.
#include "common.h"
#include "halo_map.h"
#include "conv_common.h"

namespace {

constexpr int S_TY = 8, S_TX = 16, S_BN = 64, S_STRIP = 4;

template <int TERMS>
__global__ __launch_bounds__(256, 2) void conv_stem_kernel(const woft_conv_params p) {
    constexpr int NP = (TERMS == 3) ? 2 : 1;
    constexpr int PY = 2 * S_TY + 5, PX = 2 * S_TX + 6;            // patch: 21 rows x 38 pixels
    constexpr int PITCH = PX * 4 + 8;                              // bf16 elements per patch row (320 B)
    constexpr int A_PLANE = PY * PITCH;
    constexpr int NPATCH = PY * PX;                                // float4 loads per patch
    constexpr int RL = (NPATCH + 255) / 256;                       // ... per thread
    constexpr int TM = 2;                                          // 64 rows per wave (WM = 2), 32 columns (WN = 2)
    __shared__ __attribute__((aligned(16))) __bf16 As[NP * A_PLANE];
    __shared__ __attribute__((aligned(16))) float stage[4 * woft::STAGE_FLOATS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r32 = lane & 31, hh = lane >> 5;

    const int tyn = (p.ho + S_TY - 1) / S_TY, txn = (p.wo + S_TX - 1) / S_TX;
    const int mt = p.n_img * tyn * txn;
    const int nt = p.cout_pad / S_BN;
    const int strip = (int)blockIdx.x / nt, n_tile = (int)blockIdx.x - strip * nt;
    const int n0 = n_tile * S_BN;
    const int ktot = 7 * 32;

    // weights of this wave's 32 columns: fragment (ky, s2) = k 32 ky + 16 s2 + 8 hh .. + 7 of column n0 + 32 wn + r32
    bf16x8 bw[7][2][NP];
    {
        const int64_t roff = (int64_t)(n0 + wn * 32 + r32) * ktot + hh * 8;
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bw[ky][s2][0] = *(const bf16x8*)((const __bf16*)p.wgt_hi + roff + ky * 32 + s2 * 16);
                if (NP == 2) bw[ky][s2][NP - 1] = *(const bf16x8*)((const __bf16*)p.wgt_lo + roff + ky * 32 + s2 * 16);
            }
    }
    // A fragment base of row tile i: the pixel this lane's tile row holds (rows without a pixel repeat pixel 0)
    int a_off[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        bool valid;
        const int pl = halo_row_pixel<S_TY, S_TX>(wm * 64 + i * 32 + r32, valid);
        a_off[i] = (2 * (pl / S_TX)) * PITCH + (2 * (pl % S_TX) + 2 * hh) * 4;
    }

    f32x4 rp[RL];
    auto load_patch = [&](int m_tile) {                            // RL unconditional 16-byte loads (clamped + select)
        const int img = m_tile / (tyn * txn), trem = m_tile - img * (tyn * txn);
        const int y0 = (trem / txn) * S_TY, x0 = (trem % txn) * S_TX;
#pragma unroll
        for (int j = 0; j < RL; ++j) {
            const int idx = tid + 256 * j;
            const int pr = idx / PX, pc = idx - pr * PX;
            const int iy = 2 * y0 - p.pad_y + pr, ix = 2 * x0 - p.pad_x + pc;
            const bool ok = idx < NPATCH && iy >= 0 && iy < p.h && ix >= 0 && ix < p.w;
            const int64_t pix = ok ? ((int64_t)img * p.h + iy) * p.w + ix : 0;
            const f32x4 v = *(const f32x4*)(p.in0 + pix * 4);
            rp[j] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int j = 0; j < RL; ++j) {
            const int idx = tid + 256 * j;
            if (idx < NPATCH) {
                const int pr = idx / PX, pc = idx - pr * PX;
                const bf16x4 hi = cvt16<TERMS>(rp[j]);
                *(bf16x4*)(As + pr * PITCH + pc * 4) = hi;
                if (NP == 2) {
                    const f32x4 rem = rp[j] - widen_bf16x4(hi);
                    *(bf16x4*)(As + A_PLANE + pr * PITCH + pc * 4) = __builtin_convertvector(rem, bf16x4);
                }
            }
        }
    };

    const int t_first = strip * S_STRIP;
    const int t_end = (t_first + S_STRIP < mt) ? t_first + S_STRIP : mt;
    load_patch(t_first);
#pragma unroll 1
    for (int m_tile = t_first; m_tile < t_end; ++m_tile) {
        store_patch();
        __syncthreads();
        if (m_tile + 1 < t_end) load_patch(m_tile + 1);            // in flight across this tile's MFMAs and epilogue
        f32x16 acc[TM][1];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 a[NP][TM];
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        a[pl][i] = *(const bf16x8*)(As + pl * A_PLANE + a_off[i] + ky * PITCH + s2 * 16);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    if (NP == 2) {
                        acc[i][0] = mma16<TERMS>(a[NP - 1][i], bw[ky][s2][0], acc[i][0]);
                        acc[i][0] = mma16<TERMS>(a[0][i], bw[ky][s2][NP - 1], acc[i][0]);
                    }
                    acc[i][0] = mma16<TERMS>(a[0][i], bw[ky][s2][0], acc[i][0]);
                }
            }
        __syncthreads();                                           // the patch planes are free for the next tile
        const int img = m_tile / (tyn * txn), trem = m_tile - img * (tyn * txn);
        const HaloRowMap<S_TY, S_TX> rowmap{img, p.n_img, (trem / txn) * S_TY, (trem % txn) * S_TX, p.ho, p.wo};
        woft::conv_epilogue_t<TM, 1, 64, 32>(p, acc, stage + wave * woft::STAGE_FLOATS, rowmap, n0, wm, wn, lane, m_tile);
    }
}

}  // namespace

// conv.hip: woft_conv2d routes p.halo == 7 here (validated by conv_check first)
int woft_conv_stem_launch(const woft_conv_params& p, void* stream) {
    if (!p.flat || p.cs0 != 4 || p.stride != 2 || p.taps_y != 7 || p.taps_x != 1 || p.pad_y != 3 || p.pad_x != 3 ||
        p.cin_pad != 32 || p.in1 != nullptr || p.in_norm != 0 || p.precision == 0 || p.cout_pad % S_BN != 0 ||
        p.bias_map != nullptr || p.wh0_lookup != nullptr)
        return WOFT_EINVAL;
    if (p.ho != (p.h + 2 * p.pad_y - 7) / 2 + 1 || p.wo != (p.w + 2 * p.pad_x - 7) / 2 + 1) return WOFT_EINVAL;
    if ((int64_t)p.n_img * p.h * p.w >= (1ll << 31)) return WOFT_EINVAL;
    const int64_t mt = (int64_t)p.n_img * ((p.ho + S_TY - 1) / S_TY) * ((p.wo + S_TX - 1) / S_TX);
    const int64_t strips = (mt + S_STRIP - 1) / S_STRIP;
    if (strips * (p.cout_pad / S_BN) >= (1ll << 31)) return WOFT_EINVAL;
    dim3 grid((unsigned)(strips * (p.cout_pad / S_BN)));
    hipStream_t s = (hipStream_t)stream;
    if (p.precision == 1) woft_launch(0, conv_stem_kernel<3>, grid, dim3(256), 0, s, p);
    else if (p.precision == 3) woft_launch(0, conv_stem_kernel<16>, grid, dim3(256), 0, s, p);
    else woft_launch(0, conv_stem_kernel<1>, grid, dim3(256), 0, s, p);
    return woft_launch_status();
}
