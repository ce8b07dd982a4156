// Multi-scale correlation lookup (corr.py:29-59, utils/utils.py:59-73) -- the HBM-roofline kernel.
//
// Volume layout (chosen by this library: it owns the producer GEMM and this consumer): level l of
// the pyramid is [P][Ht_l][Wt_l][4][4] fp32 -- the (y2, x2) plane of every source pixel p is cut
// into 4x4 tiles of 64 contiguous bytes (one HBM/L2 sector), zero filled beyond the map.  A
// (2r+2)^2 window then touches 3-4 tiles per axis (10.6 sectors on average for r = 4) instead of
// 2r+2 row segments of 40 B (15.6 sectors), and every access is an aligned 16-byte load.
//
// One wavefront per source pixel p.  For every level the 64 lanes cover the 4x4 block of tiles that
// contains the window -- lane = (tile row a, tile column b, row r inside the tile) -- with ONE
// float4 load each, stage the 16x16 patch in LDS, then every lane produces its share of the
// L*(2r+1)^2 bilinear samples from 4 LDS reads and the wave writes the pixel's output channels as
// one contiguous run (NHWC).  All loads of a lane are issued before any is consumed.
//
// Algorithmic bytes per pixel per call: L * ((2r+2)^2 * 4 + (2r+1)^2 * 4) = 2896 B for r=4, L=4.
//
// Sampling rule (what grid_sample(align_corners=True, padding zeros) computes after the reference's
// normalise / un-normalise round trip, evaluated directly in pixel coordinates):
//   xs = x / 2^l + (i - r), ys = y / 2^l + (j - r);  x0 = floor(xs), fx = xs - x0  (same for y)
//   out[l*(2r+1)^2 + i*(2r+1) + j] = sum over the 4 neighbours, zero outside [0,W) x [0,H).
#include "common.h"

namespace {

constexpr int WAVES_PER_BLOCK = 4;

template <int R>
__global__ __launch_bounds__(256) void corr_lookup_kernel(const woft_lookup_params p) {
    constexpr int NW = 2 * R + 1;        // output window side
    constexpr int N2 = NW * NW;
    constexpr int NOUT = (4 * N2 + 63) / 64;
    __shared__ __attribute__((aligned(16))) float patch[WAVES_PER_BLOCK][4][256];   // 16x16 per level
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t pix = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wave;
    const bool active = pix < p.n_pix;
    const int nout = p.levels * N2;

    float cx = 0.f, cy = 0.f;
    if (active) {
        cx = p.coords[pix * 2 + 0];
        cy = p.coords[pix * 2 + 1];
    }
    const int a = lane >> 4, b = (lane >> 2) & 3, r = lane & 3;   // tile row / tile col / row in tile

    // ---- gather: one 16-B load per lane and level ------------------------------------------------
    f32x4 v[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        const float sc = 1.0f / (float)(1 << l);    // exact: x / 2^l
        float flx = floorf(cx * sc), fly = floorf(cy * sc);
        flx = fminf(fmaxf(flx, -1.0e6f), 1.0e6f);   // keeps the int conversion defined for wild coords
        fly = fminf(fmaxf(fly, -1.0e6f), 1.0e6f);
        const int wx0 = (int)flx - R, wy0 = (int)fly - R;          // window origin (integer pixel)
        const int tx = (wx0 >> 2) + b, ty = (wy0 >> 2) + a;         // floor division by 4
        f32x4 val = {0.f, 0.f, 0.f, 0.f};
        if (active && l < p.levels && tx >= 0 && tx < p.wt[l] && ty >= 0 && ty < p.ht[l] &&
            4 * b < (wx0 & 3) + 2 * R + 2 && 4 * a < (wy0 & 3) + 2 * R + 2)
            val = *(const f32x4*)(p.vol[l] + pix * p.plane[l] + ((int64_t)ty * p.wt[l] + tx) * 16 + r * 4);
        v[l] = val;
    }
#pragma unroll
    for (int l = 0; l < 4; ++l) *(f32x4*)(&patch[wave][l][(4 * a + r) * 16 + 4 * b]) = v[l];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // the patch is private to this wave: no block barrier
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- interpolate + write (dword stores, 256 B per wave instruction) ---------------------------
    if (!active) return;
    float* o = p.out + pix * p.ldo;
#pragma unroll
    for (int k = 0; k < NOUT; ++k) {
        const int c = lane + 64 * k;
        if (c >= nout) break;
        const int l = c / N2;
        const int rem = c - l * N2;
        const int i = rem / NW, j = rem - i * NW;      // i: x offset, j: y offset (x-major window)
        const float sc = 1.0f / (float)(1 << l);
        const float xs = cx * sc, ys = cy * sc;
        float flx = floorf(xs), fly = floorf(ys);
        const float wx = xs - flx, wy = ys - fly;
        flx = fminf(fmaxf(flx, -1.0e6f), 1.0e6f);
        fly = fminf(fmaxf(fly, -1.0e6f), 1.0e6f);
        const int ox = ((int)flx - R) & 3, oy = ((int)fly - R) & 3;   // window origin inside the patch
        const float* q = &patch[wave][l][(oy + j) * 16 + ox + i];
        const float top = q[0] * (1.f - wx) + q[1] * wx;
        const float bot = q[16] * (1.f - wx) + q[17] * wx;
        o[c] = top * (1.f - wy) + bot * wy;
    }
}

// rows of an NHWC map in 4x4-tile order: out[((ty*Wt + tx)*16 + dy*4 + dx)][c] = in[4ty+dy][4tx+dx][c], 0 outside
__global__ void tile_rows_kernel(const float* __restrict__ in, int h, int w, int c4, float* __restrict__ out,
                                 int wt, int64_t n_rows) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * c4) return;
    const int64_t row = i / c4;
    const int cc = (int)(i - row * c4);
    const int t = (int)(row >> 4), e = (int)(row & 15);
    const int y = 4 * (t / wt) + (e >> 2), x = 4 * (t % wt) + (e & 3);
    f32x4 val = {0.f, 0.f, 0.f, 0.f};
    if (y < h && x < w) val = *(const f32x4*)(in + ((int64_t)y * w + x) * c4 * 4 + cc * 4);
    *(f32x4*)(out + i * 4) = val;
}

__global__ void coords_update_kernel(float* __restrict__ coords1, const float* __restrict__ delta, int ld_delta,
                                     int wf, int64_t n_pix, float* __restrict__ flow4,
                                     float* __restrict__ flow_cat, int ld_cat) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pix) return;
    float x = coords1[i * 2], y = coords1[i * 2 + 1];
    if (delta != nullptr) {
        x += delta[i * ld_delta];
        y += delta[i * ld_delta + 1];
    } else {                       // initialise to the identity grid (utils/utils.py:76-79)
        x = (float)(i % wf);
        y = (float)(i / wf);
    }
    coords1[i * 2] = x;
    coords1[i * 2 + 1] = y;
    const float fx = x - (float)(i % wf), fy = y - (float)(i / wf);
    if (flow4 != nullptr) {
        f32x4 f = {fx, fy, 0.f, 0.f};
        *(f32x4*)(flow4 + i * 4) = f;
    }
    if (flow_cat != nullptr) {
        flow_cat[i * ld_cat] = fx;
        flow_cat[i * ld_cat + 1] = fy;
    }
}

}  // namespace

extern "C" int woft_corr_lookup(const woft_lookup_params* pp, void* stream) {
    if (!pp) return WOFT_EINVAL;
    const woft_lookup_params& p = *pp;
    if (p.levels < 1 || p.levels > 4 || !p.coords || !p.out || p.n_pix <= 0) return WOFT_EINVAL;
    for (int l = 0; l < p.levels; ++l)
        if (!p.vol[l] || p.ht[l] <= 0 || p.wt[l] <= 0 || p.plane[l] < (int64_t)p.ht[l] * p.wt[l] * 16) return WOFT_EINVAL;
    const int nout = p.levels * (2 * p.radius + 1) * (2 * p.radius + 1);
    if (p.ldo < nout) return WOFT_EINVAL;
    dim3 grid((unsigned)ceil_div64(p.n_pix, WAVES_PER_BLOCK));
    if (p.radius == 4)
        hipLaunchKernelGGL(corr_lookup_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else if (p.radius == 3)
        hipLaunchKernelGGL(corr_lookup_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else
        return WOFT_EINVAL;
    return woft_launch_status();
}

extern "C" int woft_tile_rows(const float* in, int32_t h, int32_t w, int32_t c, float* out, void* stream) {
    if (!in || !out || h <= 0 || w <= 0 || c <= 0 || c % 4 != 0) return WOFT_EINVAL;
    const int wt = (w + 3) / 4, ht = (h + 3) / 4;
    const int64_t n_rows = (int64_t)ht * wt * 16;
    hipLaunchKernelGGL(tile_rows_kernel, dim3((unsigned)ceil_div64(n_rows * (c / 4), 256)), dim3(256), 0,
                       (hipStream_t)stream, in, h, w, c / 4, out, wt, n_rows);
    return woft_launch_status();
}

extern "C" int woft_coords_update(float* coords1, const float* delta, int32_t ld_delta, int32_t wf, int64_t n_pix,
                                  float* flow4, float* flow_cat, int32_t ld_cat, void* stream) {
    if (!coords1 || !delta || wf <= 0 || n_pix <= 0 || ld_delta < 2) return WOFT_EINVAL;
    hipLaunchKernelGGL(coords_update_kernel, dim3((unsigned)ceil_div64(n_pix, 256)), dim3(256), 0,
                       (hipStream_t)stream, coords1, delta, ld_delta, wf, n_pix, flow4, flow_cat, ld_cat);
    return woft_launch_status();
}

extern "C" int woft_coords_init(float* coords1, int32_t hf, int32_t wf, float* flow4, float* flow_cat,
                                int32_t ld_cat, void* stream) {
    if (!coords1 || hf <= 0 || wf <= 0) return WOFT_EINVAL;
    const int64_t n = (int64_t)hf * wf;
    hipLaunchKernelGGL(coords_update_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       coords1, (const float*)nullptr, 0, wf, n, flow4, flow_cat, ld_cat);
    return woft_launch_status();
}
