// Implicit-GEMM convolution / NT-GEMM on f32-input MFMA for gfx950 (MI355X).
//
//   C[m][n] = alpha * sum_k A[m][k] * B[n][k] + bias[n]      (then a fused epilogue)
//
//   m : output pixel (image, oy, ox) in NHWC order          (GEMM M, up to ~10^7)
//   n : output channel                                       (GEMM N)
//   k : (tap, input channel); 32 channels of one tap per K step
//
// A rows are gathered from one or two NHWC sources (channel concat without a copy), zero filled
// outside the image.  B is the pre-packed weight matrix [cout_pad][taps*cin_pad] (K contiguous) --
// or, for the all-pairs correlation volume (corr.py:62-69), the second feature map itself.
//
// Block = 256 threads = 4 waves as 2(M) x 2(N); block tile BM x BN in {64,128}^2; each wave owns
// (BM/2) x (BN/2) as 32x32 tiles of v_mfma_f32_32x32x2_f32 (exact fp32: a k-ordered fmaf chain).
// Both operand tiles are staged through LDS with a 36-float row pitch (conflict-free
// ds_read_b128 fragment reads); global loads for step k+1 are issued into registers before the
// MFMAs of step k (register double buffering).  Each lane half (lane>>5) consumes its own
// contiguous 16-wide k range of the step, so fragments are read as 4 x ds_read_b128.
#include "common.h"

namespace {

constexpr int BK = 32;
constexpr int LDS_LD = 36;   // floats; 144 B rows: 16-B aligned, conflict-free for b128 reads

template <int BM, int BN>
__global__ __launch_bounds__(256) void conv_mfma_f32_kernel(const woft_conv_params p) {
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int RA = BM / 32, RB = BN / 32;
    __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDS_LD];
    float* As = smem;
    float* Bs = smem + BM * LDS_LD;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r32 = lane & 31, hh = lane >> 5;
    const int v = tid & 7, r0 = tid >> 3;

    const int64_t M = (int64_t)p.n_img * p.ho * p.wo;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int nchunk = p.cin_pad / BK;
    const int nk = p.taps_y * p.taps_x * nchunk;
    const int64_t ktot = (int64_t)nk * BK;

    // ---- per-thread A row bookkeeping -------------------------------------------------------
    int iy0[RA], ix0[RA];
    int64_t img_base[RA];
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
            img_base[j] = (int64_t)img * p.h * p.w;
        }
    }
    const float* brow[RB];
#pragma unroll
    for (int j = 0; j < RB; ++j) brow[j] = p.wgt + (int64_t)(n0 + r0 + 32 * j) * ktot + 4 * v;

    f32x4 ra[RA], rb[RB];
    auto load_tiles = [&](int ks) {
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
                const int iy = iy0[j] + ky, ix = ix0[j] + kx;
                const bool ok = mvalid[j] && iy >= 0 && iy < p.h && ix >= 0 && ix < p.w;
                f32x4 val = {0.f, 0.f, 0.f, 0.f};
                if (ok) val = *(const f32x4*)(src + (img_base[j] + (int64_t)iy * p.w + ix) * cs + cc);
                ra[j] = val;
            }
        } else {
            const int dpix = (4 * v) / p.cs0;
#pragma unroll
            for (int j = 0; j < RA; ++j) {
                const int iy = iy0[j] + ky, ixp = ix0[j] + dpix;
                const bool ok = mvalid[j] && iy >= 0 && iy < p.h && ixp >= 0 && ixp < p.w;
                f32x4 val = {0.f, 0.f, 0.f, 0.f};
                if (ok) val = *(const f32x4*)(p.in0 + (img_base[j] + (int64_t)iy * p.w + ix0[j]) * p.cs0 + 4 * v);
                ra[j] = val;
            }
        }
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

    // ---- epilogue -----------------------------------------------------------------------------
    // C/D layout of the 32x32 MFMA: column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
    const bool do_stats = p.stat_sum != nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / 2) + j * 32 + r32;
        const bool nvalid = n < p.cout;
        const float bias = (p.bias != nullptr) ? p.bias[n] : 0.f;
        int64_t col = n;
        if (p.out_pitch != 0) col = (int64_t)(n / p.out_w) * p.out_pitch + (n % p.out_w);
        col += p.co_off;
        float ssum = 0.f, ssq = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                const bool ok = nvalid && (m < M);
                float y = p.alpha * acc[i][j][r] + bias;
                if (do_stats && ok) { ssum += y; ssq += y * y; }
                if (!ok) continue;
                switch (p.epi) {
                    case WOFT_EPI_LINEAR: break;
                    case WOFT_EPI_RELU: y = fmaxf(y, 0.f); break;
                    case WOFT_EPI_SIGMOID: y = sigmoidf_(y); break;
                    case WOFT_EPI_TANH: y = tanhf(y); break;
                    case WOFT_EPI_RELU_RES_RELU:
                        y = fmaxf(p.e0[m * p.lde0 + n] + fmaxf(y, 0.f), 0.f);
                        break;
                    case WOFT_EPI_GRU_ZR:
                        y = sigmoidf_(y);
                        if (n >= p.split) {
                            p.out1[m * p.ldo1 + (n - p.split)] = y * p.e0[m * p.lde0 + (n - p.split)];
                            continue;
                        }
                        break;
                    case WOFT_EPI_GRU_Q: {
                        const float z = p.e1[m * p.lde1 + n], hprev = p.e0[m * p.lde0 + n];
                        y = (1.f - z) * hprev + z * tanhf(y);
                    } break;
                    case WOFT_EPI_CTX: y = (n < p.split) ? tanhf(y) : fmaxf(y, 0.f); break;
                    default: break;
                }
                p.out[m * p.ldo + col] = y;
            }
        }
        if (do_stats) {
            ssum += __shfl_xor(ssum, 32);
            ssq += __shfl_xor(ssq, 32);
            if (hh == 0) {
                const int64_t row = (int64_t)blockIdx.x * 2 + wm;
                p.stat_sum[row * p.cout_pad + n] = ssum;
                p.stat_sq[row * p.cout_pad + n] = ssq;
            }
        }
    }
}

template <int BM, int BN>
int launch_conv(const woft_conv_params& p, hipStream_t s) {
    const int64_t M = (int64_t)p.n_img * p.ho * p.wo;
    dim3 grid((unsigned)ceil_div64(M, BM), (unsigned)(p.cout_pad / BN));
    hipLaunchKernelGGL((conv_mfma_f32_kernel<BM, BN>), grid, dim3(256), 0, s, p);
    return woft_launch_status();
}

}  // namespace

extern "C" int woft_conv2d(const woft_conv_params* pp, void* stream) {
    if (pp == nullptr) return WOFT_EINVAL;
    const woft_conv_params& p = *pp;
    if (p.in0 == nullptr || p.wgt == nullptr || p.out == nullptr) return WOFT_EINVAL;
    if (p.cin_pad <= 0 || p.cin_pad % BK != 0) return WOFT_EINVAL;
    if (p.in1 != nullptr && (p.c_split % BK != 0 || p.c_split <= 0 || p.c_split >= p.cin_pad)) return WOFT_EINVAL;
    if (p.cs0 % 4 != 0 || (p.in1 != nullptr && p.cs1 % 4 != 0)) return WOFT_EINVAL;
    if (p.flat && (p.taps_x != 1 || p.in1 != nullptr || (p.cs0 != 4 && p.cs0 != 8 && p.cs0 != 16 && p.cs0 != 32)))
        return WOFT_EINVAL;
    if (p.n_img <= 0 || p.h <= 0 || p.w <= 0 || p.ho <= 0 || p.wo <= 0 || p.taps_y <= 0 || p.taps_x <= 0 ||
        p.stride <= 0 || p.cout <= 0)
        return WOFT_EINVAL;
    if ((p.tile_m != 64 && p.tile_m != 128) || (p.tile_n != 64 && p.tile_n != 128)) return WOFT_EINVAL;
    if (p.cout_pad % p.tile_n != 0 || p.cout > p.cout_pad) return WOFT_EINVAL;
    if (p.epi < 0 || p.epi > WOFT_EPI_CTX) return WOFT_EINVAL;
    if ((p.epi == WOFT_EPI_RELU_RES_RELU || p.epi == WOFT_EPI_GRU_ZR || p.epi == WOFT_EPI_GRU_Q) && p.e0 == nullptr)
        return WOFT_EINVAL;
    if (p.epi == WOFT_EPI_GRU_Q && p.e1 == nullptr) return WOFT_EINVAL;
    if (p.epi == WOFT_EPI_GRU_ZR && p.out1 == nullptr) return WOFT_EINVAL;
    if ((p.stat_sum == nullptr) != (p.stat_sq == nullptr)) return WOFT_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (p.tile_m == 128 && p.tile_n == 128) return launch_conv<128, 128>(p, s);
    if (p.tile_m == 128 && p.tile_n == 64) return launch_conv<128, 64>(p, s);
    if (p.tile_m == 64 && p.tile_n == 128) return launch_conv<64, 128>(p, s);
    return launch_conv<64, 64>(p, s);
}
