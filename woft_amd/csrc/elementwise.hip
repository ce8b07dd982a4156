// Streaming (HBM-bound) helpers: input normalisation, InstanceNorm finalize/apply, 2x2 pooling.
#include <string.h>

#include "common.h"
#include "halo_map.h"

namespace {

// ---- uint8 BGR HWC -> normalised RGB NHWC4 with replicate padding ------------------------------
__global__ void preprocess_kernel(const uint8_t* __restrict__ img, int h, int w, float* __restrict__ out,
                                  int hp, int wp, int pad_top, int pad_left) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)hp * wp) return;
    const int y = (int)(i / wp), x = (int)(i - (int64_t)y * wp);
    const int sy = min(max(y - pad_top, 0), h - 1), sx = min(max(x - pad_left, 0), w - 1);
    const uint8_t* px = img + ((int64_t)sy * w + sx) * 3;
    f32x4 v;
    // reference order of operations: 2 * (x / 255.0) - 1.0 on the RGB-flipped image
    v[0] = 2.f * ((float)px[2] / 255.0f) - 1.0f;
    v[1] = 2.f * ((float)px[1] / 255.0f) - 1.0f;
    v[2] = 2.f * ((float)px[0] / 255.0f) - 1.0f;
    v[3] = 0.f;
    *(f32x4*)(out + i * 4) = v;
}

// ---- InstanceNorm statistics: reduce the conv epilogue's per-wave-row partial sums -----------
// s1 / s2: [n_part][ld] fp32 partial sums / sums of squares per channel.  Workgroup g of G takes the rows g, g + G, ...
// with 16-byte loads along the channels (a wave reads whole rows: the first version gave every channel its own workgroup
// striding down a column -- 64 cache lines per load instruction, 12-46 us per call, 15 calls per encoder pass), sums them in
// fp64 and leaves [2][channels] doubles in `ws`; the LAST workgroup to finish (agent-scope release -> ticket -> acquire)
// totals the G partial rows in a fixed order and writes mean / rstd.  G = 1 (ws NULL or few rows): no ticket.
constexpr int FIN_T = 256, FIN_MAXG = 64, FIN_MAXC = 256;
__global__ __launch_bounds__(FIN_T) void inorm_finalize_kernel(const float* __restrict__ s1, const float* __restrict__ s2,
                                                               int n_part, int ld, int channels, int channels_pad,
                                                               int64_t count, float eps, float* __restrict__ mean,
                                                               float* __restrict__ rstd, double* ws) {
    __shared__ double red[2][FIN_T * 4];
    __shared__ int s_last;
    const int c4n = (channels + 3) / 4;                  // 16-byte columns that hold real channels
    const int rpp = FIN_T / c4n;                         // rows per pass of the workgroup
    const int tr = threadIdx.x / c4n, tc = threadIdx.x - tr * c4n;
    const int G = gridDim.x, g = blockIdx.x;
    double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
    if (tr < rpp)
        for (int i = g * rpp + tr; i < n_part; i += G * rpp) {
            const f32x4 u = *(const f32x4*)(s1 + (int64_t)i * ld + tc * 4), v = *(const f32x4*)(s2 + (int64_t)i * ld + tc * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                a[k] += (double)u[k];
                b[k] += (double)v[k];
            }
        }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        red[0][threadIdx.x * 4 + k] = a[k];
        red[1][threadIdx.x * 4 + k] = b[k];
    }
    __syncthreads();
    const int c = threadIdx.x;                           // channel of this thread from here on (channels_pad <= FIN_T)
    double ta = 0.0, tb = 0.0;
    if (c < channels) {
        const int q = c >> 2, k = c & 3;
        for (int r = 0; r < rpp; ++r) {
            ta += red[0][(r * c4n + q) * 4 + k];
            tb += red[1][(r * c4n + q) * 4 + k];
        }
    }
    if (G > 1) {
        if (c < channels) {
            ws[(int64_t)g * 2 * FIN_MAXC + c] = ta;
            ws[(int64_t)g * 2 * FIN_MAXC + FIN_MAXC + c] = tb;
        }
        int* ticket = (int*)(ws + (int64_t)FIN_MAXG * 2 * FIN_MAXC);
        // Hand-off to the last workgroup, the guide's counter form (Guideline 16): every wave's stores acknowledged -> barrier ->
        // ONE lane: agent-scope release (writes back this XCD's L2) -> wait -> relaxed ticket; the workgroup that draws the last
        // ticket: ONE agent-scope acquire (invalidates this CU's L1) -> barrier -> plain loads.  (Round 6: was __threadfence() by all
        // 256 threads on both sides -- 2-4 x one lane's cost; this kernel runs 16 times per encoder pass, 1.75 % of a frame.)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (restated where the compiler cannot drop it: guide, pitfall 12)
            s_last = (__hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == G - 1);
            if (s_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (!s_last) return;
        if (threadIdx.x == 0) *ticket = 0;               // ready for the next call on this stream
        // total of the G partial rows, fixed order: slice sl of the workgroup takes the rows sl, sl + nsl, ... of channel cc
        // (loads of a thread are independent: issued in batches), then the slices in order
        int cp = 1;
        while (cp < channels) cp <<= 1;                  // power of two >= channels (<= FIN_T)
        const int nsl = FIN_T / cp, sl = threadIdx.x / cp, cc = threadIdx.x & (cp - 1);
        double pa = 0.0, pb = 0.0;
        if (cc < channels) {
#pragma unroll 8
            for (int j = sl; j < G; j += nsl) {
                pa += ws[(int64_t)j * 2 * FIN_MAXC + cc];
                pb += ws[(int64_t)j * 2 * FIN_MAXC + FIN_MAXC + cc];
            }
        }
        __syncthreads();                                 // (red is free: every thread passed the barriers above)
        red[0][threadIdx.x] = pa;
        red[1][threadIdx.x] = pb;
        __syncthreads();
        ta = tb = 0.0;
        if (c < channels)
            for (int r = 0; r < nsl; ++r) {
                ta += red[0][r * cp + c];
                tb += red[1][r * cp + c];
            }
    }
    if (c < channels) {
        const double mu = ta / (double)count;
        double var = tb / (double)count - mu * mu;       // biased variance, as nn.InstanceNorm2d
        if (var < 0.0) var = 0.0;
        mean[c] = (float)mu;
        rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    } else if (c < channels_pad) {                       // padding channels of the activation buffer: keep them exactly zero
        mean[c] = 0.f;
        rstd[c] = 0.f;
    }
}

// res_mode (mode 2 only): 0 = res is an activation; 1 / 2 = res is a RAW conv output normalised here with its own statistics
// (2: followed by ReLU) -- the shortcut of a residual block whose normalisation was deferred to this kernel
__global__ void inorm_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                   const float* __restrict__ rstd, const float* __restrict__ res,
                                   const float* __restrict__ res_mean, const float* __restrict__ res_rstd, int res_mode,
                                   float* __restrict__ out, int64_t n4, int channels, int mode) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const int c = (int)((i * 4) % channels);
        f32x4 v = *(const f32x4*)(x + i * 4);
        const f32x4 mu = *(const f32x4*)(mean + c), rs = *(const f32x4*)(rstd + c);
        f32x4 r = {0.f, 0.f, 0.f, 0.f};
        if (mode == 2) {
            r = *(const f32x4*)(res + i * 4);
            if (res_mode != 0) {
                const f32x4 rmu = *(const f32x4*)(res_mean + c), rrs = *(const f32x4*)(res_rstd + c);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    r[k] = (r[k] - rmu[k]) * rrs[k];
                    if (res_mode == 2) r[k] = fmaxf(r[k], 0.f);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float y = (v[k] - mu[k]) * rs[k];
            if (mode >= 1) y = fmaxf(y, 0.f);
            if (mode == 2) y = fmaxf(r[k] + y, 0.f);
            v[k] = y;
        }
        *(f32x4*)(out + i * 4) = v;
    }
}

// ---- 2x2 stride-2 average pool of an NHWC map (floor output size) ---------------------------
__global__ void avgpool2_kernel(const float* __restrict__ in, int h, int w, int c, float* __restrict__ out) {
    const int ho = h / 2, wo = w / 2, c4 = c / 4;
    const int64_t n = (int64_t)ho * wo * c4;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int cc = (int)(i % c4);
    const int64_t pix = i / c4;
    const int oy = (int)(pix / wo), ox = (int)(pix - (int64_t)oy * wo);
    const float* b = in + ((int64_t)(2 * oy) * w + 2 * ox) * c + cc * 4;
    const f32x4 p00 = *(const f32x4*)b, p01 = *(const f32x4*)(b + c);
    const f32x4 p10 = *(const f32x4*)(b + (int64_t)w * c), p11 = *(const f32x4*)(b + (int64_t)w * c + c);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = (p00[k] + p01[k] + p10[k] + p11[k]) * 0.25f;
    *(f32x4*)(out + pix * c + cc * 4) = o;
}

// ---- target feature pyramid in ONE launch (corr = "otf": pooled maps + their split operands) -------------------------
// avgpool2_kernel x (levels - 1) and split_bf16(_lines) x levels were 7 launches of 5-12 us each per frame for 45 MB of
// traffic.  A workgroup takes an 8 x 8 block of level-0 pixels: thread (sub-block s of 4 x 4 pixels, channel quad q) reads
// its 16 float4s, forms the level-1 (2 x 2) and level-2 (1) averages in registers with avgpool2_kernel's own expression
// ((p00 + p01 + p10 + p11) * 0.25 of the level below: bit-identical), level 3 from the four sub-blocks through LDS, and
// writes every level's fp32 map (levels >= 1) and split operand (TERMS 3: [hi | lo] lines of 32 channels; 1: the bf16 plane).
typedef __bf16 pbf16x4 __attribute__((ext_vector_type(4)));
struct PyrArgs {
    const float* in;
    float* pooled[3];
    __bf16* split[4];
    int h, w, c, levels;
};
template <int TERMS>
__device__ __forceinline__ void pyr_split(__bf16* out, int64_t pix, int c, int q, const f32x4 v) {
    const pbf16x4 hi = __builtin_convertvector(v, pbf16x4);
    if (TERMS == 3) {
        __bf16* line = out + pix * (2 * (int64_t)c) + (q >> 3) * 64 + (q & 7) * 4;
        *(pbf16x4*)line = hi;
        *(pbf16x4*)(line + 32) = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), pbf16x4);
    } else {
        *(pbf16x4*)(out + pix * (int64_t)c + q * 4) = hi;
    }
}
template <int TERMS>
__global__ __launch_bounds__(256) void feature_pyramid_kernel(const PyrArgs a) {
    __shared__ f32x4 l2s[4][64];
    const int bxn = (a.w + 7) / 8;
    const int by = (int)blockIdx.x / bxn, bx = (int)blockIdx.x - by * bxn;
    const int sub = threadIdx.x >> 6, q0 = threadIdx.x & 63;
    const int c4 = a.c / 4;
    const int y0 = by * 8 + (sub >> 1) * 4, x0 = bx * 8 + (sub & 1) * 4;
    const int h1 = a.h / 2, w1 = a.w / 2, h2 = h1 / 2, w2 = w1 / 2, h3 = h2 / 2, w3 = w2 / 2;
    for (int qb = 0; qb < c4; qb += 64) {                  // (uniform trip count: the barriers below are reached by all)
        const int q = qb + q0;
        const bool qok = q < c4;
        f32x4 p[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int y = y0 + i, x = x0 + j;
                const bool ok = qok && y < a.h && x < a.w;
                p[i][j] = ok ? *(const f32x4*)(a.in + ((int64_t)y * a.w + x) * a.c + q * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
                if (ok) pyr_split<TERMS>(a.split[0], (int64_t)y * a.w + x, a.c, q, p[i][j]);
            }
        f32x4 o1[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    o1[i][j][k] = (p[2 * i][2 * j][k] + p[2 * i][2 * j + 1][k] + p[2 * i + 1][2 * j][k] + p[2 * i + 1][2 * j + 1][k]) * 0.25f;
                const int y = y0 / 2 + i, x = x0 / 2 + j;
                if (a.levels > 1 && qok && y < h1 && x < w1) {
                    const int64_t pix = (int64_t)y * w1 + x;
                    *(f32x4*)(a.pooled[0] + pix * a.c + q * 4) = o1[i][j];
                    pyr_split<TERMS>(a.split[1], pix, a.c, q, o1[i][j]);
                }
            }
        f32x4 o2;
#pragma unroll
        for (int k = 0; k < 4; ++k) o2[k] = (o1[0][0][k] + o1[0][1][k] + o1[1][0][k] + o1[1][1][k]) * 0.25f;
        {
            const int y = y0 / 4, x = x0 / 4;
            if (a.levels > 2 && qok && y < h2 && x < w2) {
                const int64_t pix = (int64_t)y * w2 + x;
                *(f32x4*)(a.pooled[1] + pix * a.c + q * 4) = o2;
                pyr_split<TERMS>(a.split[2], pix, a.c, q, o2);
            }
        }
        __syncthreads();                                   // (l2s of the previous channel pass is consumed)
        l2s[sub][q0] = o2;
        __syncthreads();
        if (sub == 0 && a.levels > 3 && qok && by < h3 && bx < w3) {
            f32x4 o3;
#pragma unroll
            for (int k = 0; k < 4; ++k) o3[k] = (l2s[0][q0][k] + l2s[1][q0][k] + l2s[2][q0][k] + l2s[3][q0][k]) * 0.25f;
            const int64_t pix = (int64_t)by * w3 + bx;
            *(f32x4*)(a.pooled[2] + pix * a.c + q * 4) = o3;
            pyr_split<TERMS>(a.split[3], pix, a.c, q, o3);
        }
    }
}

}  // namespace

extern "C" int woft_abi_version(void) { return 10000 * 0 + 100 * 3 + 0; }

extern "C" int woft_preprocess_bgr_u8(const uint8_t* img, int32_t h, int32_t w, float* out, int32_t hp, int32_t wp,
                                      int32_t pad_top, int32_t pad_left, void* stream) {
    if (!img || !out || h <= 0 || w <= 0 || hp < h || wp < w || pad_top < 0 || pad_left < 0) return WOFT_EINVAL;
    const int64_t n = (int64_t)hp * wp;
    hipLaunchKernelGGL(preprocess_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, (hipStream_t)stream, img, h,
                       w, out, hp, wp, pad_top, pad_left);
    return woft_launch_status();
}

extern "C" int64_t woft_inorm_ws_bytes(void) { return (int64_t)FIN_MAXG * 2 * FIN_MAXC * 8 + 64; }

extern "C" int woft_inorm_finalize(const float* stat_sum, const float* stat_sq, int32_t n_part, int32_t ld,
                                   int32_t channels, int32_t channels_pad, int64_t count, float eps, float* mean,
                                   float* rstd, void* ws, void* stream) {
    if (!stat_sum || !stat_sq || !mean || !rstd || n_part <= 0 || channels <= 0 || count <= 0) return WOFT_EINVAL;
    if (channels_pad < channels || channels_pad > FIN_MAXC || ld % 4 != 0 || ld < channels) return WOFT_EINVAL;
    const int rpp = FIN_T / ((channels + 3) / 4);
    int G = (n_part + 4 * rpp - 1) / (4 * rpp);          // >= 4 passes per workgroup
    G = (ws == nullptr || G < 2) ? 1 : (G > FIN_MAXG ? FIN_MAXG : G);
    hipLaunchKernelGGL(inorm_finalize_kernel, dim3(G), dim3(FIN_T), 0, (hipStream_t)stream, stat_sum, stat_sq, n_part, ld,
                       channels, channels_pad, count, eps, mean, rstd, (double*)ws);
    return woft_launch_status();
}

extern "C" int woft_inorm_apply(const float* x, const float* mean, const float* rstd, const float* res,
                                const float* res_mean, const float* res_rstd, int32_t res_mode, float* out,
                                int64_t n_pix, int32_t channels, int32_t mode, void* stream) {
    if (!x || !mean || !rstd || !out || n_pix <= 0 || channels <= 0 || channels % 4 != 0) return WOFT_EINVAL;
    if (mode < 0 || mode > 2 || (mode == 2 && !res)) return WOFT_EINVAL;
    if (res_mode < 0 || res_mode > 2 || (res_mode != 0 && (mode != 2 || !res_mean || !res_rstd))) return WOFT_EINVAL;
    const int64_t n4 = n_pix * channels / 4;
    const int64_t blocks = ceil_div64(n4, 256);
    hipLaunchKernelGGL(inorm_apply_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0,
                       (hipStream_t)stream, x, mean, rstd, res, res_mean, res_rstd, res_mode, out, n4, channels, mode);
    return woft_launch_status();
}

extern "C" int woft_avgpool2_nhwc(const float* in, int32_t h, int32_t w, int32_t c, float* out, void* stream) {
    if (!in || !out || h < 2 || w < 2 || c <= 0 || c % 4 != 0) return WOFT_EINVAL;
    const int64_t n = (int64_t)(h / 2) * (w / 2) * (c / 4);
    hipLaunchKernelGGL(avgpool2_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, (hipStream_t)stream, in, h,
                       w, c, out);
    return woft_launch_status();
}

// sizeof() of the ABI structs, so the Python ctypes mirror can verify its layout at load time.
extern "C" int woft_upload_u8(const void* src, void* pinned, void* dev, int64_t bytes, int32_t n_chunks, void* stream) {
    if (!src || !pinned || !dev || bytes <= 0 || n_chunks < 1 || n_chunks > 64) return WOFT_EINVAL;
    const int64_t per = ((bytes + n_chunks - 1) / n_chunks + 4095) / 4096 * 4096;          // whole pages per piece
    for (int64_t off = 0; off < bytes; off += per) {
        const int64_t n = bytes - off < per ? bytes - off : per;
        memcpy((char*)pinned + off, (const char*)src + off, (size_t)n);
        if (hipMemcpyAsync((char*)dev + off, (const char*)pinned + off, (size_t)n, hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess)
            return WOFT_ELAUNCH;
    }
    return WOFT_OK;
}

extern "C" int woft_sizeof(int which) {
    if (which == 0) return (int)sizeof(woft_conv_params);
    if (which == 1) return (int)sizeof(woft_lookup_params);
    if (which == 2) return (int)sizeof(woft_lookup_otf_params);
    return -1;
}

extern "C" int woft_feature_pyramid(const float* in, int32_t h, int32_t w, int32_t c, int32_t levels, float* const* pooled,
                                    void* const* split, int32_t terms, void* stream) {
    if (!in || !pooled || !split || h <= 0 || w <= 0 || c <= 0 || c % 32 != 0 || levels < 1 || levels > 4 ||
        (terms != 1 && terms != 3))
        return WOFT_EINVAL;
    PyrArgs a;
    a.in = in; a.h = h; a.w = w; a.c = c; a.levels = levels;
    for (int l = 0; l < 4; ++l) {
        a.split[l] = l < levels ? (__bf16*)split[l] : nullptr;
        if (l < levels && a.split[l] == nullptr) return WOFT_EINVAL;
        if (l >= 1) {
            a.pooled[l - 1] = l < levels ? pooled[l - 1] : nullptr;
            if (l < levels && a.pooled[l - 1] == nullptr) return WOFT_EINVAL;
        }
    }
    const dim3 grid((unsigned)(((h + 7) / 8) * ((w + 7) / 8)));
    if (terms == 3) hipLaunchKernelGGL(feature_pyramid_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(feature_pyramid_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, a);
    return woft_launch_status();
}
