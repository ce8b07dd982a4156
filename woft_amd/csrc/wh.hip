// Weight-head glue kernels (weighted_raft.py:258-279, 347-384): the mean-response channel in its
// algebraic form, the packing of the (scrambled) lookup channels into 9x9 patches, and the final
// 1x1 conv + patch mean.  The three 3x3 convolutions in between run on the MFMA conv kernel.
#include "common.h"

namespace {

// stage 1: partial[b][c] = sum over the pixel strip of block b;  stage 2: total[c] (double)
__global__ void colsum_stage1(const float* __restrict__ f, int64_t n_pix, int c, double* __restrict__ partial,
                              int n_part) {
    const int64_t per = (n_pix + n_part - 1) / n_part;
    const int64_t beg = (int64_t)blockIdx.x * per;
    const int64_t end = beg + per < n_pix ? beg + per : n_pix;
    // (eight independent chains: the single-accumulator loop was one dependent load latency per pixel)
    for (int ch = threadIdx.x; ch < c; ch += blockDim.x) {
        double s[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        int64_t q = beg;
        for (; q + 8 <= end; q += 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) s[k] += (double)f[(q + k) * c + ch];
        }
        for (; q < end; ++q) s[0] += (double)f[q * c + ch];
        partial[(int64_t)blockIdx.x * c + ch] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    }
}
__global__ void colsum_stage2(const double* __restrict__ partial, int n_part, int c, double* __restrict__ total) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    double s[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    int b = 0;
    for (; b + 8 <= n_part; b += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] += partial[(int64_t)(b + k) * c + ch];
    }
    for (; b < n_part; ++b) s[0] += partial[(int64_t)b * c + ch];
    total[ch] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}

// mean[p] = alpha * <f1[p], total>   -- one wavefront per pixel, shuffle reduction
__global__ __launch_bounds__(256) void wh_mean_kernel(const float* __restrict__ f1, int c,
                                                      const double* __restrict__ total, float alpha,
                                                      int64_t n_pix, float* __restrict__ mean) {
    const int lane = threadIdx.x & 63;
    const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= n_pix) return;
    double s = 0.0;
    for (int ch = lane; ch < c; ch += 64) s += (double)f1[p * c + ch] * total[ch];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) mean[p] = (float)(s * (double)alpha);
}

// x8[p][t][0..3] = lookup[p][t*4 .. t*4+3]  (the reference reads the level-major lookup channels
// as (hp wp level)), x8[p][t][4] = mean[p], x8[p][t][5..7] = 0
__global__ void wh_pack_kernel(const float* __restrict__ lookup, int ld, const float* __restrict__ mean,
                               int64_t n_pix, int nwin2, float* __restrict__ x8) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pix * nwin2) return;
    const int64_t p = i / nwin2;
    const int t = (int)(i - p * nwin2);
    const f32x4 a = *(const f32x4*)(lookup + p * ld + t * 4);
    const f32x4 b = {mean[p], 0.f, 0.f, 0.f};
    *(f32x4*)(x8 + i * 8) = a;
    *(f32x4*)(x8 + i * 8 + 4) = b;
}

// out[p] = bias + (1/nwin2) * sum_t <w, act[p][t][:]>  -- one wavefront per pixel
__global__ __launch_bounds__(256) void wh_reduce_kernel(const float* __restrict__ act, int c, int nwin2,
                                                        const float* __restrict__ w, float bias, int64_t n_pix,
                                                        float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= n_pix) return;
    const int c4 = c / 4, n4 = nwin2 * c4;
    const float* a = act + p * (int64_t)nwin2 * c;
    float s = 0.f;
    for (int i = lane; i < n4; i += 64) {
        const f32x4 v = *(const f32x4*)(a + (int64_t)i * 4);
        const f32x4 ww = *(const f32x4*)(w + (i % c4) * 4);
        s += v[0] * ww[0] + v[1] * ww[1] + v[2] * ww[2] + v[3] * ww[3];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[p] = bias + s / (float)nwin2;
}

// First conv of the weight head (weighted_raft.py:336: 5 -> 128 channels, 3x3, ReLU) on the NW x NW lookup window
// of every source pixel, exact fp32 on the vector ALUs, reading the lookup buffer directly (no packed copy):
//   x[t][0..3] = lookup[p][4 t .. 4 t + 3]   (the reference reads the level-major channels as (hp wp level),
//   x[t][4]    = mean[p]                       weighted_raft.py:267-272, 363-376)
// GEMM K is only 45, the output is 1.3 GB at 1080p: the matrix-core kernel is launch/epilogue bound on it.  Here a
// lane owns TWO output channels (their 90 weights live in registers, packed FMAs) and a wave walks over the window; the window
// values are wave-uniform, so they are fetched by SCALAR loads and enter the FMAs as scalar operands -- no LDS, no
// vector loads.  Each wave instruction stores the 512 contiguous bytes of one pixel's channel vector.
template <int NW>
__global__ __launch_bounds__(256) void wh_conv0_kernel(const float* __restrict__ lookup, int ld,
                                                       const float* __restrict__ mean, int n_pix,
                                                       const float* __restrict__ wt, const float* __restrict__ b0,
                                                       float* __restrict__ out, const int* __restrict__ index) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int co = 2 * lane;                             // this lane's two output channels (packed FMAs)
    f32x2 w[3][3][5];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int ci = 0; ci < 5; ++ci) w[ky][kx][ci] = *(const f32x2*)(wt + (ky * 32 + kx * 8 + ci) * 128 + co);
    const f32x2 bias = *(const f32x2*)(b0 + co);
    for (int p = blockIdx.x * 4 + wave; p < n_pix; p += gridDim.x * 4) {
        const int src = index ? index[p] : p;           // window p of the output = source pixel index[p]
        if (src < 0) continue;                          // (dynamic window list: -1 = window not wanted in this launch)
        const float* __restrict__ lk = lookup + (int64_t)src * ld;
        const float mv = mean[src];
        float* __restrict__ o = out + (int64_t)p * (NW * NW * 128) + co;
#pragma unroll 1
        for (int y = 0; y < NW; ++y) {
#pragma unroll
            for (int x = 0; x < NW; ++x) {
                f32x2 acc = bias;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int yy = y + ky - 1;
                    if (yy < 0 || yy >= NW) continue;                  // (wave-uniform)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int xx = x + kx - 1;
                        if (xx < 0 || xx >= NW) continue;              // (compile time)
                        const float* xv = lk + (yy * NW + xx) * 4;
#pragma unroll
                        for (int ci = 0; ci < 4; ++ci) {
                            const f32x2 xs = {xv[ci], xv[ci]};
                            acc = __builtin_elementwise_fma(w[ky][kx][ci], xs, acc);
                        }
                        const f32x2 ms = {mv, mv};
                        acc = __builtin_elementwise_fma(w[ky][kx][4], ms, acc);
                    }
                }
                const f32x2 r = {fmaxf(acc[0], 0.f), fmaxf(acc[1], 0.f)};
                *(f32x2*)(o + (y * NW + x) * 128) = r;
            }
        }
    }
}

}  // namespace

// ---- the windows a set of selected correspondences needs (tracker: the fit reads the weights of its <= 500 Sobol-sampled
//      correspondences only, and which ones they are is decided by the flow alone) --------------------------------------
// The x8 convex upsampling of the weight map at full-resolution pixel (x, y) reads the 3x3 neighbourhood of its 1/8-res
// cell ((y + top) >> 3, (x + left) >> 3) (weighted_raft.py:92-103: unfold with padding 1).
__global__ void wh_mark_kernel(const float* __restrict__ pts, const int* __restrict__ count, int n_max, int top, int left,
                               int hf, int wf, int* __restrict__ bitmap) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = count ? min(count[0], n_max) : n_max;
    if (i >= n) return;
    const int cx = ((int)pts[2 * i] + left) >> 3, cy = ((int)pts[2 * i + 1] + top) >> 3;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int yy = cy + dy, xx = cx + dx;
            if (yy >= 0 && yy < hf && xx >= 0 && xx < wf) bitmap[yy * wf + xx] = 1;       // (every writer writes 1)
        }
}

__global__ void wh_dyn_index_kernel(const int* __restrict__ index, int n_win, const int* __restrict__ bitmap,
                                    int* __restrict__ dyn, int* __restrict__ n_needed) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_win) return;
    const int src = index[j];
    const bool need = bitmap[src] != 0;
    dyn[j] = need ? src : -1;
    if (need && n_needed != nullptr) atomicAdd(n_needed, 1);
}

extern "C" int woft_wh_needed(const float* pts, const int32_t* count, int32_t n_max, int32_t top, int32_t left, int32_t hf,
                              int32_t wf, const int32_t* index, int32_t n_win, int32_t* bitmap, int32_t* dyn_index,
                              int32_t* n_needed, void* stream) {
    if (!pts || !index || !bitmap || !dyn_index || n_max <= 0 || n_win <= 0 || hf <= 0 || wf <= 0) return WOFT_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    (void)hipMemsetAsync(bitmap, 0, (size_t)hf * wf * sizeof(int32_t), s);
    if (n_needed) (void)hipMemsetAsync(n_needed, 0, sizeof(int32_t), s);
    hipLaunchKernelGGL(wh_mark_kernel, dim3((n_max + 255) / 256), dim3(256), 0, s, pts, count, n_max, top, left, hf, wf,
                       bitmap);
    hipLaunchKernelGGL(wh_dyn_index_kernel, dim3((n_win + 255) / 256), dim3(256), 0, s, index, n_win, bitmap, dyn_index,
                       n_needed);
    return woft_launch_status();
}

extern "C" int woft_wh_conv0(const float* lookup, int32_t ld_lookup, const float* mean, int64_t n_pix, int32_t nwin,
                             const float* wt, const float* bias, float* out, const int32_t* index, void* stream) {
    if (!lookup || !mean || !wt || !bias || !out || n_pix <= 0 || n_pix >= (1ll << 31)) return WOFT_EINVAL;
    if (ld_lookup < nwin * nwin * 4 || (nwin != 9 && nwin != 7)) return WOFT_EINVAL;
    const int64_t quads = (n_pix + 3) / 4;                // one wave per window, four per workgroup
    dim3 grid((unsigned)(quads < 256 * 8 ? quads : 256 * 8));
    hipStream_t s = (hipStream_t)stream;
    if (nwin == 9)
        hipLaunchKernelGGL(wh_conv0_kernel<9>, grid, dim3(256), 0, s, lookup, ld_lookup, mean, (int)n_pix, wt, bias, out, index);
    else
        hipLaunchKernelGGL(wh_conv0_kernel<7>, grid, dim3(256), 0, s, lookup, ld_lookup, mean, (int)n_pix, wt, bias, out, index);
    return woft_launch_status();
}

extern "C" int woft_colsum(const float* f, int64_t n_pix, int32_t c, double* ws, int32_t n_part, double* total,
                           void* stream) {
    if (!f || !ws || !total || n_pix <= 0 || c <= 0 || n_part <= 0) return WOFT_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(colsum_stage1, dim3(n_part), dim3(256), 0, s, f, n_pix, c, ws, n_part);
    hipLaunchKernelGGL(colsum_stage2, dim3((c + 255) / 256), dim3(256), 0, s, ws, n_part, c, total);
    return woft_launch_status();
}

extern "C" int woft_wh_pack(const float* lookup, int32_t ld_lookup, const float* f1, int32_t c, const double* f2_total,
                            float alpha, int64_t n_pix, int32_t nwin, float* mean, float* x8, void* stream) {
    if (!lookup || !f1 || !f2_total || !mean || n_pix <= 0 || nwin <= 0 || c <= 0) return WOFT_EINVAL;
    if (ld_lookup < nwin * nwin * 4 || ld_lookup % 4 != 0) return WOFT_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(wh_mean_kernel, dim3((unsigned)ceil_div64(n_pix, 4)), dim3(256), 0, s, f1, c, f2_total, alpha,
                       n_pix, mean);
    if (x8 == nullptr) return woft_launch_status();     // mean only (woft_wh_conv0 reads the lookup buffer itself)
    const int64_t n = n_pix * nwin * nwin;
    hipLaunchKernelGGL(wh_pack_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, s, lookup, ld_lookup, mean,
                       n_pix, nwin * nwin, x8);
    return woft_launch_status();
}

extern "C" int woft_wh_reduce(const float* act, int32_t c, int32_t nwin2, const float* w, float bias, int64_t n_pix,
                              float* out, void* stream) {
    if (!act || !w || !out || c <= 0 || c % 4 != 0 || nwin2 <= 0 || n_pix <= 0) return WOFT_EINVAL;
    hipLaunchKernelGGL(wh_reduce_kernel, dim3((unsigned)ceil_div64(n_pix, 4)), dim3(256), 0, (hipStream_t)stream, act,
                       c, nwin2, w, bias, n_pix, out);
    return woft_launch_status();
}
