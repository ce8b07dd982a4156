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
    __shared__ float patch[WAVES_PER_BLOCK][4][WIN * WIN];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t pix = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wave;
    const bool active = pix < p.n_pix;
    const int L = p.levels;

    float cx = 0.f, cy = 0.f;
    if (active) {
        cx = p.coords[pix * 2 + 0];
        cy = p.coords[pix * 2 + 1];
    }
    // per-level integer origin and fractional parts (division by 2^l is exact in fp32)
    int x0[4], y0[4];
    float fx[4], fy[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        const float sc = 1.0f / (float)(1 << l);
        const float xs = cx * sc, ys = cy * sc;
        float flx = floorf(xs), fly = floorf(ys);
        fx[l] = xs - flx;
        fy[l] = ys - fly;
        flx = fminf(fmaxf(flx, -1.0e6f), 1.0e6f);   // keeps the int conversion defined for wild coords
        fly = fminf(fmaxf(fly, -1.0e6f), 1.0e6f);
        x0[l] = (int)flx - R;
        y0[l] = (int)fly - R;
    }

    // ---- gather: L * WIN*WIN taps, lane-contiguous along window rows -------------------------
    const int ntap = L * WIN * WIN;
    for (int t = lane; t < ntap; t += 64) {
        const int l = t / (WIN * WIN);
        const int rem = t - l * (WIN * WIN);
        const int ry = rem / WIN, rx = rem - ry * WIN;
        int ox = x0[0], oy = y0[0];
        if (l == 1) { ox = x0[1]; oy = y0[1]; }
        if (l == 2) { ox = x0[2]; oy = y0[2]; }
        if (l == 3) { ox = x0[3]; oy = y0[3]; }
        const int gx = ox + rx, gy = oy + ry;
        float val = 0.f;
        if (active && gx >= 0 && gx < p.wl[l] && gy >= 0 && gy < p.hl[l])
            val = p.vol[l][pix * p.plane[l] + (int64_t)gy * p.pitch[l] + gx];
        patch[wave][l][rem] = val;
    }
    __syncthreads();

    // ---- interpolate + write --------------------------------------------------------------------
    if (!active) return;
    const int nout = L * NW * NW;
    float* o = p.out + pix * p.ldo;
    for (int c = lane; c < nout; c += 64) {
        const int l = c / (NW * NW);
        const int rem = c - l * (NW * NW);
        const int i = rem / NW, j = rem - i * NW;      // i: x offset, j: y offset (x-major window)
        float wx = fx[0], wy = fy[0];
        if (l == 1) { wx = fx[1]; wy = fy[1]; }
        if (l == 2) { wx = fx[2]; wy = fy[2]; }
        if (l == 3) { wx = fx[3]; wy = fy[3]; }
        const float* q = &patch[wave][l][j * WIN + i];
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
