// Multi-scale correlation lookup (corr.py:29-59, utils/utils.py:59-73) -- the HBM-roofline kernel.
//
// One wavefront per source pixel p.  For every pyramid level l the wave gathers the
// (2r+2) x (2r+2) integer neighbourhood of (x, y) / 2^l from vol_l[p] into LDS (adjacent lanes read
// adjacent columns of a window row, so a row is one coalesced segment), then every lane produces
// its share of the L*(2r+1)^2 bilinear samples from 4 LDS reads and the wave writes the pixel's
// output channels as one contiguous run (NHWC).
//
// Algorithmic bytes per pixel per call: L * ((2r+2)^2 * 4 + (2r+1)^2 * 4) = 2896 B for r=4, L=4.
//
// Sampling rule (what grid_sample(align_corners=True, padding zeros) computes after the reference's
// normalise / un-normalise round trip, evaluated directly in pixel coordinates):
//   xs = x / 2^l + (i - r), ys = y / 2^l + (j - r);  x0 = floor(xs), fx = xs - x0  (same for y)
//   out[l*(2r+1)^2 + i*(2r+1) + j] = sum over the 4 neighbours, zero outside [0,W) x [0,H).
#include "common.h"

namespace {

constexpr int MAX_WIN = 10;              // 2r+2 for r = 4
constexpr int WAVES_PER_BLOCK = 4;

template <int R>
__global__ __launch_bounds__(256) void corr_lookup_kernel(const woft_lookup_params p) {
    constexpr int WIN = 2 * R + 2;       // integer patch side
    constexpr int NW = 2 * R + 1;        // output window side
    constexpr int W2 = WIN * WIN, N2 = NW * NW;
    constexpr int NLOAD = (4 * W2 + 63) / 64, NOUT = (4 * N2 + 63) / 64;
    __shared__ float patch[WAVES_PER_BLOCK][4 * W2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t pix = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wave;
    const bool active = pix < p.n_pix;
    const int ntap = p.levels * W2, nout = p.levels * N2;

    float cx = 0.f, cy = 0.f;
    if (active) {
        cx = p.coords[pix * 2 + 0];
        cy = p.coords[pix * 2 + 1];
    }

    // ---- gather: all of a lane's taps are issued before any is consumed -----------------------
    float v[NLOAD];
#pragma unroll
    for (int k = 0; k < NLOAD; ++k) {
        const int t = lane + 64 * k;
        const int l = t / W2;                       // level of this tap (lane dependent)
        const int rem = t - l * W2;
        const int ry = rem / WIN, rx = rem - ry * WIN;
        const float sc = 1.0f / (float)(1 << l);    // exact: x / 2^l
        float flx = floorf(cx * sc), fly = floorf(cy * sc);
        flx = fminf(fmaxf(flx, -1.0e6f), 1.0e6f);   // keeps the int conversion defined for wild coords
        fly = fminf(fmaxf(fly, -1.0e6f), 1.0e6f);
        const int gx = (int)flx - R + rx, gy = (int)fly - R + ry;
        float val = 0.f;
        if (active && t < ntap) {
            const int ll = l < 4 ? l : 3;
            if (gx >= 0 && gx < p.wl[ll] && gy >= 0 && gy < p.hl[ll])
                val = p.vol[ll][pix * p.plane[ll] + (int64_t)gy * p.pitch[ll] + gx];
        }
        v[k] = val;
    }
#pragma unroll
    for (int k = 0; k < NLOAD; ++k) {
        const int t = lane + 64 * k;
        if (t < 4 * W2) patch[wave][t] = v[k];
    }
    __syncthreads();

    // ---- interpolate + write (dword stores, 256 B per wave instruction; 16-B-per-lane stores were
    //      measured slower: 53 vs 49 us at 1080p) -------------------------------------------------
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
        const float wx = xs - floorf(xs), wy = ys - floorf(ys);
        const float* q = &patch[wave][l * W2 + j * WIN + i];
        const float top = q[0] * (1.f - wx) + q[1] * wx;
        const float bot = q[WIN] * (1.f - wx) + q[WIN + 1] * wx;
        o[c] = top * (1.f - wy) + bot * wy;
    }
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
        if (!p.vol[l] || p.hl[l] <= 0 || p.wl[l] <= 0 || p.pitch[l] < p.wl[l]) return WOFT_EINVAL;
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
