// Streaming (HBM-bound) helpers: input normalisation, InstanceNorm finalize/apply, 2x2 pooling.
#include "common.h"

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
__global__ void inorm_finalize_kernel(const float* __restrict__ s1, const float* __restrict__ s2, int n_part, int ld,
                                      int channels, int64_t count, float eps, float* __restrict__ mean,
                                      float* __restrict__ rstd) {
    const int c = blockIdx.x;
    if (c >= channels) {            // padding channels of the activation buffer: keep them exactly zero
        if (threadIdx.x == 0) { mean[c] = 0.f; rstd[c] = 0.f; }
        return;
    }
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < n_part; i += blockDim.x) {
        a += (double)s1[(int64_t)i * ld + c];
        b += (double)s2[(int64_t)i * ld + c];
    }
    __shared__ double sa[256], sb[256];
    sa[threadIdx.x] = a;
    sb[threadIdx.x] = b;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            sa[threadIdx.x] += sa[threadIdx.x + s];
            sb[threadIdx.x] += sb[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double mu = sa[0] / (double)count;
        double var = sb[0] / (double)count - mu * mu;   // biased variance, as nn.InstanceNorm2d
        if (var < 0.0) var = 0.0;
        mean[c] = (float)mu;
        rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

__global__ void inorm_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                   const float* __restrict__ rstd, const float* __restrict__ res,
                                   float* __restrict__ out, int64_t n4, int channels, int mode) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const int c = (int)((i * 4) % channels);
        f32x4 v = *(const f32x4*)(x + i * 4);
        const f32x4 mu = *(const f32x4*)(mean + c), rs = *(const f32x4*)(rstd + c);
        f32x4 r = {0.f, 0.f, 0.f, 0.f};
        if (mode == 2) r = *(const f32x4*)(res + i * 4);
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

}  // namespace

extern "C" int woft_abi_version(void) { return 10000 * 0 + 100 * 1 + 0; }

extern "C" int woft_preprocess_bgr_u8(const uint8_t* img, int32_t h, int32_t w, float* out, int32_t hp, int32_t wp,
                                      int32_t pad_top, int32_t pad_left, void* stream) {
    if (!img || !out || h <= 0 || w <= 0 || hp < h || wp < w || pad_top < 0 || pad_left < 0) return WOFT_EINVAL;
    const int64_t n = (int64_t)hp * wp;
    hipLaunchKernelGGL(preprocess_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, (hipStream_t)stream, img, h,
                       w, out, hp, wp, pad_top, pad_left);
    return woft_launch_status();
}

extern "C" int woft_inorm_finalize(const float* stat_sum, const float* stat_sq, int32_t n_part, int32_t ld,
                                   int32_t channels, int32_t channels_pad, int64_t count, float eps, float* mean,
                                   float* rstd, void* stream) {
    if (!stat_sum || !stat_sq || !mean || !rstd || n_part <= 0 || channels <= 0 || count <= 0) return WOFT_EINVAL;
    if (channels_pad < channels) return WOFT_EINVAL;
    hipLaunchKernelGGL(inorm_finalize_kernel, dim3(channels_pad), dim3(256), 0, (hipStream_t)stream, stat_sum, stat_sq,
                       n_part, ld, channels, count, eps, mean, rstd);
    return woft_launch_status();
}

extern "C" int woft_inorm_apply(const float* x, const float* mean, const float* rstd, const float* res, float* out,
                                int64_t n_pix, int32_t channels, int32_t mode, void* stream) {
    if (!x || !mean || !rstd || !out || n_pix <= 0 || channels <= 0 || channels % 4 != 0) return WOFT_EINVAL;
    if (mode < 0 || mode > 2 || (mode == 2 && !res)) return WOFT_EINVAL;
    const int64_t n4 = n_pix * channels / 4;
    const int64_t blocks = ceil_div64(n4, 256);
    hipLaunchKernelGGL(inorm_apply_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0,
                       (hipStream_t)stream, x, mean, rstd, res, out, n4, channels, mode);
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
extern "C" int woft_sizeof(int which) {
    if (which == 0) return (int)sizeof(woft_conv_params);
    if (which == 1) return (int)sizeof(woft_lookup_params);
    if (which == 2) return (int)sizeof(woft_lookup_otf_params);
    return -1;
}
