// 3x3 stride-1 "same" convolution with one or two output channels, exact fp32 on the vector ALUs.
//
// The flow head's second conv (update.py:10-17: 256 -> 2; small model 128 -> 2) has GEMM N = 2: on the matrix
// cores it pays for a 64-column tile (97 % padding) and, worse, runs one long serial K loop per workgroup.  Here
// a wave owns a run of 16 output pixels of one image row; lane L holds CPL = cin / 64 consecutive input channels
// and the 9 x COUT x CPL weights of those channels in registers.  The three input columns of the 3x3 window
// slide along the run (each new column = 3 loads of CPL floats per lane, one full channel vector per wave
// instruction), so an input pixel is read three times in total instead of nine.  Each output is the sum of the 64
// per-lane partial sums (totalled once per run, through LDS).
#include "common.h"

namespace {

template <int CPL>
struct Vec;
template <>
struct Vec<4> { typedef float T __attribute__((ext_vector_type(4))); };
template <>
struct Vec<2> { typedef float T __attribute__((ext_vector_type(2))); };

constexpr int RUN = 16;

template <int COUT, int CPL>
__global__ __launch_bounds__(256) void conv3x3_narrow_kernel(const float* __restrict__ in, int cs, int n_img, int h, int w,
                                                             const float* __restrict__ wgt, int ktot, int cin_pad,
                                                             const float* __restrict__ bias, float* __restrict__ out,
                                                             int64_t ldo, int co_off, float* __restrict__ coords1,
                                                             float* __restrict__ flow4, float* __restrict__ flow_cat,
                                                             int ld_cat) {
    typedef typename Vec<CPL>::T vec;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int runs_per_row = (w + RUN - 1) / RUN;
    const int64_t run = (int64_t)blockIdx.x * 4 + wave;
    if (run >= (int64_t)n_img * h * runs_per_row) return;
    const int xr = (int)(run % runs_per_row);
    const int y = (int)((run / runs_per_row) % h);
    const int img = (int)(run / ((int64_t)runs_per_row * h));
    const int x0 = xr * RUN;

    // weights of this lane's channels: wv[o][tap]  (packed weight row o: k = tap * cin_pad + c)
    vec wv[COUT][9];
#pragma unroll
    for (int o = 0; o < COUT; ++o)
#pragma unroll
        for (int t = 0; t < 9; ++t) wv[o][t] = *(const vec*)(wgt + (int64_t)o * ktot + t * cin_pad + CPL * lane);

    const float* base = in + ((int64_t)img * h * w) * cs + CPL * lane;
    auto load_col = [&](int x, vec (&col)[3]) {          // rows y-1, y, y+1 of column x (zero outside the image)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = y + ky - 1;
            const bool ok = x >= 0 && x < w && yy >= 0 && yy < h;
            const vec val = *(const vec*)(base + (ok ? ((int64_t)yy * w + x) * cs : 0));
            col[ky] = ok ? val : (vec)(0.f);
        }
    };

    // four column buffers used round-robin: step k reads columns (k, k+1, k+2) mod 4 = x-1, x, x+1 and prefetches
    // x+2 into (k+3) mod 4, so the names come back after four pixels and the run loop stays rolled (few registers:
    // all waves of a 1/8-resolution 1080p frame are resident at once)
    vec cb[4][3];
    load_col(x0 - 1, cb[0]);
    load_col(x0, cb[1]);
    load_col(x0 + 1, cb[2]);
    const float b0 = bias ? bias[0] : 0.f, b1 = (bias && COUT > 1) ? bias[COUT - 1] : 0.f;
    float* orow = out + (((int64_t)img * h + y) * w) * ldo + co_off;
    // per-lane partial sums of the whole run first (LDS, [value][lane]), ONE reduction at the end: a butterfly per pixel was
    // a chain of six dependent cross-lane moves per output (192 per run, most of the kernel's 23 us).  Then lane L totals
    // one half (32 lanes' partials) of value L >> 1 in lane order and adds its partner's half.
    constexpr int NV = RUN * COUT, PLD = 65;             // (65: the 32 values a wave reads in one go fall into different banks)
    __shared__ float part[4][NV * PLD];
    float* mypart = part[wave];
#pragma unroll 1
    for (int i0 = 0; i0 < RUN; i0 += 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int x = x0 + i0 + k;
            load_col(x + 2, cb[(k + 3) & 3]);            // in flight while this pixel is accumulated
            const vec(&l)[3] = cb[k & 3];
            const vec(&m)[3] = cb[(k + 1) & 3];
            const vec(&r)[3] = cb[(k + 2) & 3];
#pragma unroll
            for (int o = 0; o < COUT; ++o) {
                float t = 0.f;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int e = 0; e < CPL; ++e) {
                        t = fmaf(wv[o][ky * 3 + 0][e], l[ky][e], t);
                        t = fmaf(wv[o][ky * 3 + 1][e], m[ky][e], t);
                        t = fmaf(wv[o][ky * 3 + 2][e], r[ky][e], t);
                    }
                mypart[((i0 + k) * COUT + o) * PLD + lane] = t;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // (the partials are private to this wave)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float v[1];
    {
        constexpr int LPV = 64 / NV;                     // lanes per value: 2 (NV = 32) or 4 (NV = 16)
        const int val = lane / LPV, sub = lane % LPV;
        const float* src = mypart + val * PLD + sub * (64 / LPV);
        float t = 0.f;
#pragma unroll
        for (int e = 0; e < 64 / LPV; ++e) t += src[e];
#pragma unroll
        for (int d = 1; d < LPV; d <<= 1) t += __shfl_xor(t, d, 64);
        v[0] = t;
    }
    constexpr int SH = (NV == 32) ? 1 : (NV == 16) ? 2 : 3;             // lanes per value = 1 << SH
    static_assert(NV == 32 || NV == 16 || NV == 8, "run x output channels");
    const int idx = lane >> SH, pix = idx / COUT, o = idx - pix * COUT;
    const int x = x0 + pix;
    const float tot = v[0] + (o == 0 ? b0 : b1);
    float other = 0.f;
    if (COUT == 2) other = __shfl_xor(tot, 1 << SH, 64);                // the pixel's other channel
    if ((lane & ((1 << SH) - 1)) == 0 && x < w) {
        orow[(int64_t)x * ldo + (o == 0 ? 0 : COUT - 1)] = tot;
        if (COUT == 2 && o == 0 && coords1 != nullptr) {   // coords1 += delta (weighted_raft.py:237), same fp32 adds as
            const int64_t i = (int64_t)y * w + x;          // woft_coords_update on the stored delta (n_img == 1)
            const float cx = coords1[i * 2] + tot, cy = coords1[i * 2 + 1] + other;
            coords1[i * 2] = cx;
            coords1[i * 2 + 1] = cy;
            const float fx = cx - (float)x, fy = cy - (float)y;
            if (flow4 != nullptr) *(f32x4*)(flow4 + i * 4) = f32x4{fx, fy, 0.f, 0.f};
            if (flow_cat != nullptr) {
                flow_cat[i * ld_cat] = fx;
                flow_cat[i * ld_cat + 1] = fy;
            }
        }
    }
}

// woft_flow_head_gather: a workgroup owns a run of GT pixels of one image row.  The partial products of the run's 3 x (GT + 2)
// neighbourhood are summed over the column-tile planes while they are copied to LDS with fully coalesced 16-byte loads (a
// pixel's 18 values are 80 contiguous bytes: read per pixel from 9 neighbours they were 9 x planes scattered 8-byte loads per
// thread, 10.5 us for 5 MB), then every thread adds its 9 taps from LDS in a fixed order.
constexpr int GT = 128;
__global__ __launch_bounds__(GT) void flow_head_gather_kernel(const float* __restrict__ part, int n_planes, int ld, int h, int w,
                                                              const float* __restrict__ bias2, float* __restrict__ delta,
                                                              int64_t ld_delta, float* __restrict__ coords1,
                                                              float* __restrict__ flow4, float* __restrict__ flow_cat,
                                                              int ld_cat) {
    constexpr int LD = 20;                               // floats kept per pixel (18 + 2)
    __shared__ __attribute__((aligned(16))) float nb[3][(GT + 2) * LD];
    const int runs = (w + GT - 1) / GT;
    const int y = blockIdx.x / runs, x0 = (blockIdx.x - y * runs) * GT;
    const int64_t plane = (int64_t)h * w * ld;
    for (int i = threadIdx.x; i < 3 * (GT + 2) * (LD / 4); i += GT) {
        const int r = i / ((GT + 2) * (LD / 4)), rem = i - r * ((GT + 2) * (LD / 4));
        const int px = rem / (LD / 4), c4 = (rem - px * (LD / 4)) * 4;
        const int yy = y + r - 1, xx = x0 + px - 1;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (yy >= 0 && yy < h && xx >= 0 && xx < w) {    // (outside the image: zero padding, update.py:14)
            const float* src = part + ((int64_t)yy * w + xx) * ld + c4;
            v = *(const f32x4*)src;
            for (int t = 1; t < n_planes; ++t) v += *(const f32x4*)(src + t * plane);
        }
        *(f32x4*)(&nb[r][px * LD + c4]) = v;
    }
    __syncthreads();
    const int x = x0 + threadIdx.x;
    if (x >= w) return;
    float dx = bias2 ? bias2[0] : 0.f, dy = bias2 ? bias2[1] : 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const float* v = &nb[ky][(threadIdx.x + kx) * LD + (ky * 3 + kx) * 2];
            dx += v[0];
            dy += v[1];
        }
    const int64_t i = (int64_t)y * w + x;
    delta[i * ld_delta] = dx;
    delta[i * ld_delta + 1] = dy;
    const float cx = coords1[i * 2] + dx, cy = coords1[i * 2 + 1] + dy;
    coords1[i * 2] = cx;
    coords1[i * 2 + 1] = cy;
    const float fx = cx - (float)x, fy = cy - (float)y;
    if (flow4 != nullptr) *(f32x4*)(flow4 + i * 4) = f32x4{fx, fy, 0.f, 0.f};
    if (flow_cat != nullptr) {
        flow_cat[i * ld_cat] = fx;
        flow_cat[i * ld_cat + 1] = fy;
    }
}

}  // namespace

extern "C" int woft_flow_head_gather(const float* part, int32_t n_planes, int32_t ld, int32_t h, int32_t w, const float* bias2,
                                     float* delta, int64_t ld_delta, float* coords1, float* flow4, float* flow_cat,
                                     int32_t ld_cat, void* stream) {
    if (!part || !delta || !coords1 || n_planes < 1 || ld < 20 || ld % 4 != 0 || h <= 0 || w <= 0 || ld_delta < 2 ||
        (flow_cat != nullptr && ld_cat < 2) || (int64_t)h * w >= (1ll << 31))
        return WOFT_EINVAL;
    hipLaunchKernelGGL(flow_head_gather_kernel, dim3((unsigned)((int64_t)h * ((w + GT - 1) / GT))), dim3(GT), 0, (hipStream_t)stream,
                       part, n_planes, ld, h, w, bias2, delta, ld_delta, coords1, flow4, flow_cat, ld_cat);
    return woft_launch_status();
}

static int narrow_launch(const float* in, int32_t cs, int32_t n_img, int32_t h, int32_t w, int32_t cin_pad,
                         const float* wgt, const float* bias, int32_t cout, float* out, int64_t ldo, int32_t co_off,
                         float* coords1, float* flow4, float* flow_cat, int32_t ld_cat, void* stream) {
    if (!in || !wgt || !out || n_img <= 0 || h <= 0 || w <= 0) return WOFT_EINVAL;
    if ((cin_pad != 256 && cin_pad != 128) || cs < cin_pad || cs % 4 != 0) return WOFT_EINVAL;
    if (cout < 1 || cout > 2 || co_off < 0 || ldo < co_off + cout) return WOFT_EINVAL;
    const int64_t runs = (int64_t)n_img * h * ((w + RUN - 1) / RUN);
    dim3 grid((unsigned)ceil_div64(runs, 4));
    const int ktot = 9 * cin_pad;
    hipStream_t s = (hipStream_t)stream;
#define NARROW(CO, CPL) \
    hipLaunchKernelGGL((conv3x3_narrow_kernel<CO, CPL>), grid, dim3(256), 0, s, in, cs, n_img, h, w, wgt, ktot, cin_pad, \
                       bias, out, ldo, co_off, coords1, flow4, flow_cat, ld_cat)
    if (cin_pad == 256) { if (cout == 2) NARROW(2, 4); else NARROW(1, 4); }
    else { if (cout == 2) NARROW(2, 2); else NARROW(1, 2); }
#undef NARROW
    return woft_launch_status();
}

extern "C" int woft_conv3x3_narrow(const float* in, int32_t cs, int32_t n_img, int32_t h, int32_t w, int32_t cin_pad,
                                   const float* wgt, const float* bias, int32_t cout, float* out, int64_t ldo,
                                   int32_t co_off, void* stream) {
    return narrow_launch(in, cs, n_img, h, w, cin_pad, wgt, bias, cout, out, ldo, co_off, nullptr, nullptr, nullptr, 0,
                         stream);
}

extern "C" int woft_flow_head_update(const float* in, int32_t cs, int32_t h, int32_t w, int32_t cin_pad,
                                     const float* wgt, const float* bias, float* delta, int64_t ld_delta,
                                     float* coords1, float* flow4, float* flow_cat, int32_t ld_cat, void* stream) {
    if (!coords1 || (flow_cat != nullptr && ld_cat < 2)) return WOFT_EINVAL;
    return narrow_launch(in, cs, 1, h, w, cin_pad, wgt, bias, 2, delta, ld_delta, 0, coords1, flow4, flow_cat, ld_cat,
                         stream);
}
