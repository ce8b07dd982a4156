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
// B16: the same layout with bf16 elements (32-byte tiles, one 8-byte load per lane and level; written by the
// correlation GEMM with out_bf16) -- L * ((2r+2)^2 * 2 + (2r+1)^2 * 4) = 2096 B; interpolation and output stay fp32.
//
// Sampling rule (what grid_sample(align_corners=True, padding zeros) computes after the reference's
// normalise / un-normalise round trip, evaluated directly in pixel coordinates):
//   xs = x / 2^l + (i - r), ys = y / 2^l + (j - r);  x0 = floor(xs), fx = xs - x0  (same for y)
//   out[l*(2r+1)^2 + i*(2r+1) + j] = sum over the 4 neighbours, zero outside [0,W) x [0,H).
#include "common.h"

namespace {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int WAVES_PER_BLOCK = 4;
constexpr int PIX_PER_WAVE = 1;     // pixels a wave walks (measured: 4 with the next gather in flight = the same 30 us)

// One pixel's gathered tiles and interpolation constants (registers)
template <int NL, int NV>
struct LookupSet {
    f32x4 v[4][NL][NV];
    float fwx[4], fwy[4];
    int org[4];
};

template <int R, bool B16, int TW>
__global__ __launch_bounds__(256) void corr_lookup_kernel(const woft_lookup_params p) {
    constexpr int NW = 2 * R + 1;        // output window side
    constexpr int N2 = NW * NW;
    // tiles of 4 rows x TW columns; the window (2R+2 <= 10 wide) touches up to NTX tile columns and 4 tile rows:
    // a 16 x PW patch per level.  One load moves SEG elements of a tile row (16 B; 8 B for 4-wide bf16 tiles).
    constexpr int NTX = (TW == 4) ? 4 : 3, PW = NTX * TW;
    constexpr int SEG = B16 ? TW : 4, SPT = TW / SEG;            // segment length, segments per tile row
    constexpr int NLOAD = 16 * NTX * SPT, NL = (NLOAD + 63) / 64;  // loads per level, per lane
    constexpr int NV = (B16 && TW == 8) ? 2 : 1;
    constexpr int TSH = (TW == 4) ? 2 : 3;
    static_assert(2 * R + 2 + TW - 1 <= PW && 2 * R + 2 + 3 <= 16, "window does not fit the patch");
    __shared__ __attribute__((aligned(16))) float patch[WAVES_PER_BLOCK][4][16 * PW];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // a wave walks PIX_PER_WAVE consecutive pixels (wave-uniform index: scalar loads and address parts), the gather of
    // pixel n + 1 in flight while pixel n is interpolated and written.  With the default of one pixel per wave the loop
    // below is a single gather + emit; four pixels per wave changed nothing (the kernel is paced by the memory system:
    // DESIGN.md section 5), so the variant with the fewest registers stays.
    const int64_t pix0 = ((int64_t)blockIdx.x * WAVES_PER_BLOCK + wave) * PIX_PER_WAVE;

    int dst[NL];             // where this lane's loads land in the patch (level independent)
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        const int s = lane + 64 * k;
        const int t = s / (4 * SPT), w = s % (4 * SPT);
        const int a = t / NTX, b = t % NTX, r = w / SPT, h = w % SPT;
        dst[k] = (s < NLOAD) ? (4 * a + r) * PW + b * TW + h * SEG : -1;
    }
    const int wi = lane / NW, wj = lane - wi * NW;       // window offsets of sample `lane`: i = x offset, j = y offset
    const int woff = wj * PW + wi;

    using Set = LookupSet<NL, NV>;
    // ---- gather: tile-major lane order (the lanes of one tile read its 4 * TW elements as one contiguous run) ------
    auto gather = [&](int64_t pix, Set& g) {
        const bool active = pix < p.n_pix;
        float cx = 0.f, cy = 0.f;
        if (active) {
            cx = p.coords[pix * 2 + 0];
            cy = p.coords[pix * 2 + 1];
        }
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const float sc = 1.0f / (float)(1 << l);    // exact: x / 2^l
            float flx = floorf(cx * sc), fly = floorf(cy * sc);
            g.fwx[l] = cx * sc - flx;
            g.fwy[l] = cy * sc - fly;
            flx = fminf(fmaxf(flx, -1.0e6f), 1.0e6f);   // keeps the int conversion defined for wild coords
            fly = fminf(fmaxf(fly, -1.0e6f), 1.0e6f);
            const int wx0 = (int)flx - R, wy0 = (int)fly - R;          // window origin (integer pixel)
            g.org[l] = (wy0 & 3) * PW + (wx0 & (TW - 1));              // ... inside the patch
#pragma unroll
            for (int k = 0; k < NL; ++k) {
                const int s = lane + 64 * k;
                const int t = s / (4 * SPT), w = s % (4 * SPT);
                const int a = t / NTX, b = t % NTX, r = w / SPT, h = w % SPT;
                const int tx = (wx0 >> TSH) + b, ty = (wy0 >> 2) + a;   // floor divisions
                f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
                if (active && !(p.ablate & 1) && s < NLOAD && l < p.levels && tx >= 0 && tx < p.wt[l] && ty >= 0 &&
                    ty < p.ht[l] && TW * b < (wx0 & (TW - 1)) + 2 * R + 2 && 4 * a < (wy0 & 3) + 2 * R + 2) {
                    const int64_t at = pix * p.plane[l] + ((int64_t)ty * p.wt[l] + tx) * (4 * TW) + r * TW + h * SEG;
                    if (B16 && TW == 8) {
                        const bf16x8 q = *(const bf16x8*)((const __bf16*)p.vol[l] + at);
                        lo = __builtin_convertvector(__builtin_shufflevector(q, q, 0, 1, 2, 3), f32x4);
                        hi = __builtin_convertvector(__builtin_shufflevector(q, q, 4, 5, 6, 7), f32x4);
                    } else if (B16) {
                        lo = __builtin_convertvector(*(const bf16x4*)((const __bf16*)p.vol[l] + at), f32x4);
                    } else {
                        lo = *(const f32x4*)((const float*)p.vol[l] + at);
                    }
                }
                g.v[l][k][0] = lo;
                if (NV == 2) g.v[l][k][1] = hi;
            }
        }
    };
    // ---- patch -> LDS, interpolate, write.  Everything that depends on the level is wave-uniform and sits in the set;
    // what depends on the lane -- the sample's (i, j) inside the window -- does not depend on the level.  One pass per level
    // for the first 64 samples of its window, then the remaining N2 - 64 of all levels together.
    auto emit = [&](int64_t pix, const Set& g) {
#pragma unroll
        for (int l = 0; l < 4; ++l)
#pragma unroll
            for (int k = 0; k < NL; ++k) {
                if (dst[k] < 0) continue;
                *(f32x4*)(&patch[wave][l][dst[k]]) = g.v[l][k][0];
                if (NV == 2) *(f32x4*)(&patch[wave][l][dst[k] + 4]) = g.v[l][k][1];
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // the patch is private to this wave: no block barrier
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (pix < p.n_pix && !(p.ablate & 2)) {
            float* o = p.out + pix * p.ldo;
            auto sample = [&](const float* q, float wx, float wy) {
                const float top = q[0] * (1.f - wx) + q[1] * wx;
                const float bot = q[PW] * (1.f - wx) + q[PW + 1] * wx;
                return top * (1.f - wy) + bot * wy;
            };
#pragma unroll
            for (int l = 0; l < 4; ++l)
                if (l < p.levels && lane < N2) o[l * N2 + lane] = sample(&patch[wave][l][g.org[l] + woff], g.fwx[l], g.fwy[l]);
            if (N2 > 64) {
                constexpr int TAIL = (N2 > 64) ? N2 - 64 : 1;
#pragma unroll
                for (int k = 0; k < (4 * TAIL + 63) / 64; ++k) {
                    const int u = lane + 64 * k;
                    const int l = u / TAIL, rem = 64 + u - l * TAIL;
                    if (l >= p.levels) break;
                    const int i = rem / NW, j = rem - i * NW;
                    const float wx = l == 0 ? g.fwx[0] : l == 1 ? g.fwx[1] : l == 2 ? g.fwx[2] : g.fwx[3];
                    const float wy = l == 0 ? g.fwy[0] : l == 1 ? g.fwy[1] : l == 2 ? g.fwy[2] : g.fwy[3];
                    const int og = l == 0 ? g.org[0] : l == 1 ? g.org[1] : l == 2 ? g.org[2] : g.org[3];
                    o[l * N2 + rem] = sample(&patch[wave][0][0] + l * (16 * PW) + og + j * PW + i, wx, wy);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // the next pixel's patch overwrites this one
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    Set ga, gb;
    gather(pix0, ga);
#pragma unroll
    for (int n = 0; n < PIX_PER_WAVE; n += 2) {
        if (n + 1 < PIX_PER_WAVE) gather(pix0 + n + 1, gb);
        emit(pix0 + n, ga);
        if (n + 2 < PIX_PER_WAVE) gather(pix0 + n + 2, ga);
        if (n + 1 < PIX_PER_WAVE) emit(pix0 + n + 1, gb);
    }
}

// rows of an NHWC map in (4 x tw)-tile order: out[((ty*Wt + tx)*4*tw + dy*tw + dx)][c] = in[4ty+dy][tw*tx+dx][c], 0 outside
__global__ void tile_rows_kernel(const float* __restrict__ in, int h, int w, int c4, float* __restrict__ out,
                                 int wt, int tw, int64_t n_rows) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * c4) return;
    const int64_t row = i / c4;
    const int cc = (int)(i - row * c4);
    const int t = (int)(row / (4 * tw)), e = (int)(row % (4 * tw));
    const int y = 4 * (t / wt) + e / tw, x = tw * (t % wt) + e % tw;
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
    constexpr int tw = 4;             // 4x4-element tiles (4x8 tiles were measured slower: 31.2 vs 29.8 us)
    for (int l = 0; l < p.levels; ++l)
        if (!p.vol[l] || p.ht[l] <= 0 || p.wt[l] <= 0 || p.plane[l] < (int64_t)p.ht[l] * p.wt[l] * 4 * tw) return WOFT_EINVAL;
    const int nout = p.levels * (2 * p.radius + 1) * (2 * p.radius + 1);
    if (p.ldo < nout) return WOFT_EINVAL;
    dim3 grid((unsigned)ceil_div64(p.n_pix, WAVES_PER_BLOCK * PIX_PER_WAVE));
#define LOOKUP(R_, B_, T_) hipLaunchKernelGGL((corr_lookup_kernel<R_, B_, T_>), grid, dim3(256), 0, (hipStream_t)stream, p)
#define LOOKUP_R(R_)                                                   \
    if (!p.vol_bf16) LOOKUP(R_, false, 4);                             \
    else LOOKUP(R_, true, 4)
    if (p.radius == 4) { LOOKUP_R(4); }
    else if (p.radius == 3) { LOOKUP_R(3); }
    else return WOFT_EINVAL;
#undef LOOKUP_R
#undef LOOKUP
    return woft_launch_status();
}

extern "C" int woft_tile_rows(const float* in, int32_t h, int32_t w, int32_t c, float* out, void* stream) {
    constexpr int tile_w = 4;
    if (!in || !out || h <= 0 || w <= 0 || c <= 0 || c % 4 != 0) return WOFT_EINVAL;
    const int wt = (w + tile_w - 1) / tile_w, ht = (h + 3) / 4;
    const int64_t n_rows = (int64_t)ht * wt * 4 * tile_w;
    hipLaunchKernelGGL(tile_rows_kernel, dim3((unsigned)ceil_div64(n_rows * (c / 4), 256)), dim3(256), 0,
                       (hipStream_t)stream, in, h, w, c / 4, out, wt, tile_w, n_rows);
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
