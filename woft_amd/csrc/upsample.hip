// Final-resolution outputs: convex (learned) upsampling or bilinear x8 of the flow and of the weight
// logits, fused with the operator-boundary epilogue (un-pad crop, sigmoid, correspondences).
//   weighted_raft.py:92-103, 285-288 ; utils/utils.py:82-84 ; optical_flow/raft.py:148-159,185-199.
#include "common.h"

namespace {

// One wavefront per 1/8-res pixel (hc, wc); lane = fine position i*8 + j inside its 8x8 cell.
//   out[c, 8hc+i, 8wc+j] = sum_k softmax_k(mask[k*64 + i*8 + j]) * 8 * v[c, hc+ky-1, wc+kx-1],  k = ky*3+kx
__global__ __launch_bounds__(256) void convex_upsample_kernel(
    const float* __restrict__ coords1, const float* __restrict__ wlow, const float* __restrict__ mask, int ld_mask,
    int hf, int wf, int crop_top, int crop_left, int h, int w, float* __restrict__ flow_up, float* __restrict__ dst,
    float* __restrict__ wout, int do_sigmoid) {
    const int lane = threadIdx.x & 63;
    const int64_t pix = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pix >= (int64_t)hf * wf) return;
    const int hc = (int)(pix / wf), wc = (int)(pix - (int64_t)hc * wf);
    const float* m = mask + pix * ld_mask + lane;
    float e[9];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        e[k] = m[k * 64];
        mx = fmaxf(mx, e[k]);
    }
    float den = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        e[k] = expf(e[k] - mx);
        den += e[k];
    }
    float ax = 0.f, ay = 0.f, aw = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int ny = hc + k / 3 - 1, nx = wc + k % 3 - 1;
        float vx = 0.f, vy = 0.f, vw = 0.f;
        if (ny >= 0 && ny < hf && nx >= 0 && nx < wf) {
            const int64_t q = (int64_t)ny * wf + nx;
            vx = 8.f * (coords1[q * 2] - (float)nx);
            vy = 8.f * (coords1[q * 2 + 1] - (float)ny);
            if (wlow != nullptr) vw = 8.f * wlow[q];
        }
        const float s = e[k] / den;
        ax += s * vx;
        ay += s * vy;
        aw += s * vw;
    }
    const int y = 8 * hc + (lane >> 3) - crop_top, x = 8 * wc + (lane & 7) - crop_left;
    if (y < 0 || y >= h || x < 0 || x >= w) return;
    const int64_t o = (int64_t)y * w + x, hw = (int64_t)h * w;
    if (flow_up != nullptr) {
        flow_up[o] = ax;
        flow_up[hw + o] = ay;
    }
    if (dst != nullptr) {
        dst[o] = (float)x + ax;
        dst[hw + o] = (float)y + ay;
    }
    if (wout != nullptr && wlow != nullptr) {
        float v = aw / 8.f;
        if (do_sigmoid) v = sigmoidf_(v);
        wout[o] = v;
    }
}

// The weight channel of convex_upsample_kernel at a list of full-resolution pixels only: wsel[i] = wout[pts[i]], the same
// operations in the same order (one thread per pixel: the fit reads the weights of its <= 500 drawn correspondences).
__global__ void convex_weights_at_kernel(const float* __restrict__ pts, const int* __restrict__ count, int n_max,
                                         const float* __restrict__ wlow, const float* __restrict__ mask, int ld_mask, int hf,
                                         int wf, int crop_top, int crop_left, int do_sigmoid, float* __restrict__ wsel) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = count ? min(count[0], n_max) : n_max;
    if (i >= n) return;
    const int X = (int)pts[2 * i] + crop_left, Y = (int)pts[2 * i + 1] + crop_top;
    const int hc = Y >> 3, wc = X >> 3, lane = (Y & 7) * 8 + (X & 7);
    const float* m = mask + ((int64_t)hc * wf + wc) * ld_mask + lane;
    float e[9];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        e[k] = m[k * 64];
        mx = fmaxf(mx, e[k]);
    }
    float den = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        e[k] = expf(e[k] - mx);
        den += e[k];
    }
    float aw = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int ny = hc + k / 3 - 1, nx = wc + k % 3 - 1;
        float vw = 0.f;
        if (ny >= 0 && ny < hf && nx >= 0 && nx < wf) vw = 8.f * wlow[(int64_t)ny * wf + nx];
        const float s = e[k] / den;
        aw += s * vw;
    }
    float v = aw / 8.f;
    if (do_sigmoid) v = sigmoidf_(v);
    wsel[i] = v;
}

// 8 * bilinear(align_corners=True) upsampling (small model, no mask head).
__global__ void upflow8_kernel(const float* __restrict__ coords1, const float* __restrict__ wlow, int hf, int wf,
                               int crop_top, int crop_left, int h, int w, float* __restrict__ flow_up,
                               float* __restrict__ dst, float* __restrict__ wout, int do_sigmoid) {
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= (int64_t)h * w) return;
    const int y = (int)(o / w), x = (int)(o - (int64_t)y * w);
    const int Y = y + crop_top, X = x + crop_left;
    const int H8 = 8 * hf, W8 = 8 * wf;
    const float ry = (H8 > 1) ? (float)(hf - 1) / (float)(H8 - 1) : 0.f;
    const float rx = (W8 > 1) ? (float)(wf - 1) / (float)(W8 - 1) : 0.f;
    const float sy = ry * (float)Y, sx = rx * (float)X;
    const int y0 = (int)sy, x0 = (int)sx;
    const int yp = (y0 < hf - 1) ? 1 : 0, xp = (x0 < wf - 1) ? 1 : 0;
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const int64_t q00 = (int64_t)y0 * wf + x0, q01 = q00 + xp, q10 = q00 + (int64_t)yp * wf, q11 = q10 + xp;
    auto fl = [&](int64_t q, int ch) { return coords1[q * 2 + ch] - (float)(ch == 0 ? (q % wf) : (q / wf)); };
    const float fx = hy * (hx * fl(q00, 0) + lx * fl(q01, 0)) + ly * (hx * fl(q10, 0) + lx * fl(q11, 0));
    const float fy = hy * (hx * fl(q00, 1) + lx * fl(q01, 1)) + ly * (hx * fl(q10, 1) + lx * fl(q11, 1));
    const int64_t hw = (int64_t)h * w;
    if (flow_up != nullptr) {
        flow_up[o] = 8.f * fx;
        flow_up[hw + o] = 8.f * fy;
    }
    if (dst != nullptr) {
        dst[o] = (float)x + 8.f * fx;
        dst[hw + o] = (float)y + 8.f * fy;
    }
    if (wout != nullptr && wlow != nullptr) {
        float v = hy * (hx * wlow[q00] + lx * wlow[q01]) + ly * (hx * wlow[q10] + lx * wlow[q11]);
        v = (8.f * v) / 8.f;
        if (do_sigmoid) v = sigmoidf_(v);
        wout[o] = v;
    }
}

struct H9 { double v[9]; };

// dst(x, y) = src(Hinv (x, y)); bilinear with zero border (or nearest).  `valid` = warp(ones) > 0.
__global__ void warp_kernel(const uint8_t* __restrict__ img, int h, int w, int c, H9 hi, uint8_t* __restrict__ out,
                            uint8_t* __restrict__ valid, int nearest) {
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= (int64_t)h * w) return;
    const int y = (int)(o / w), x = (int)(o - (int64_t)y * w);
    const double d = hi.v[6] * x + hi.v[7] * y + hi.v[8];
    const double sx = (hi.v[0] * x + hi.v[1] * y + hi.v[2]) / d;
    const double sy = (hi.v[3] * x + hi.v[4] * y + hi.v[5]) / d;
    if (nearest) {
        const double rx = rint(sx), ry = rint(sy);
        const bool ok = rx >= 0 && rx < w && ry >= 0 && ry < h;
        for (int k = 0; k < c; ++k) out[o * c + k] = ok ? img[((int64_t)ry * w + (int64_t)rx) * c + k] : 0;
        if (valid) valid[o] = ok ? 1 : 0;
        return;
    }
    double fx0 = floor(sx), fy0 = floor(sy);
    const float fx = (float)(sx - fx0), fy = (float)(sy - fy0);
    fx0 = fmin(fmax(fx0, -4.0), (double)w + 4.0);
    fy0 = fmin(fmax(fy0, -4.0), (double)h + 4.0);
    const int x0 = (int)fx0, y0 = (int)fy0;
    const bool okx0 = x0 >= 0 && x0 < w, okx1 = x0 + 1 >= 0 && x0 + 1 < w;
    const bool oky0 = y0 >= 0 && y0 < h, oky1 = y0 + 1 >= 0 && y0 + 1 < h;
    const float m00 = (okx0 && oky0) ? 1.f : 0.f, m01 = (okx1 && oky0) ? 1.f : 0.f;
    const float m10 = (okx0 && oky1) ? 1.f : 0.f, m11 = (okx1 && oky1) ? 1.f : 0.f;
    const int cx0 = min(max(x0, 0), w - 1), cx1 = min(max(x0 + 1, 0), w - 1);
    const int cy0 = min(max(y0, 0), h - 1), cy1 = min(max(y0 + 1, 0), h - 1);
    if (out != nullptr) {
        for (int k = 0; k < c; ++k) {
            const float t00 = m00 * (float)img[((int64_t)cy0 * w + cx0) * c + k];
            const float t01 = m01 * (float)img[((int64_t)cy0 * w + cx1) * c + k];
            const float t10 = m10 * (float)img[((int64_t)cy1 * w + cx0) * c + k];
            const float t11 = m11 * (float)img[((int64_t)cy1 * w + cx1) * c + k];
            const float top = t00 * (1.f - fx) + t01 * fx, bot = t10 * (1.f - fx) + t11 * fx;
            const float v = top * (1.f - fy) + bot * fy;
            out[o * c + k] = (uint8_t)fminf(fmaxf(rintf(v), 0.f), 255.f);
        }
    }
    if (valid != nullptr) {
        const float top = m00 * (1.f - fx) + m01 * fx, bot = m10 * (1.f - fx) + m11 * fx;
        valid[o] = (top * (1.f - fy) + bot * fy) > 0.f ? 1 : 0;
    }
}

// cv2.resize(..., fx, fy, INTER_LINEAR) geometry: src = (dst + 0.5) * scale - 0.5, edge clamped;
// float interpolation (OpenCV's is fixed point: expect +-1 grey level).
__global__ void resize_linear_kernel(const uint8_t* __restrict__ img, int h, int w, int c, uint8_t* __restrict__ out,
                                     int ho, int wo, float sy, float sx) {
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= (int64_t)ho * wo) return;
    const int y = (int)(o / wo), x = (int)(o - (int64_t)y * wo);
    float fy = ((float)y + 0.5f) * sy - 0.5f, fx = ((float)x + 0.5f) * sx - 0.5f;
    int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
    fy -= (float)y0;
    fx -= (float)x0;
    if (y0 < 0) { y0 = 0; fy = 0.f; }
    if (x0 < 0) { x0 = 0; fx = 0.f; }
    if (y0 >= h - 1) { y0 = h - 1; fy = 0.f; }
    if (x0 >= w - 1) { x0 = w - 1; fx = 0.f; }
    const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
    for (int k = 0; k < c; ++k) {
        const float t00 = img[((int64_t)y0 * w + x0) * c + k], t01 = img[((int64_t)y0 * w + x1) * c + k];
        const float t10 = img[((int64_t)y1 * w + x0) * c + k], t11 = img[((int64_t)y1 * w + x1) * c + k];
        const float top = t00 * (1.f - fx) + t01 * fx, bot = t10 * (1.f - fx) + t11 * fx;
        out[o * c + k] = (uint8_t)fminf(fmaxf(rintf(top * (1.f - fy) + bot * fy), 0.f), 255.f);
    }
}


// Pre-computed flow (utils/caching.py:53-59 -> raft.py:93-106,159-195) to the provider's outputs:
// dst[0][i] = x + flow[0][i], dst[1][i] = y + flow[1][i], wout[i] = weights[i] or sigmoid(weights[i])
__global__ void flow_to_tc_kernel(const float* __restrict__ flow, const float* __restrict__ wts, int h, int w,
                                  float* __restrict__ dst, float* __restrict__ wout, int do_sigmoid) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = (int64_t)h * w;
    if (i >= n) return;
    if (dst != nullptr) {
        dst[i] = (float)(i % w) + flow[i];
        dst[n + i] = (float)(i / w) + flow[n + i];
    }
    if (wts != nullptr && wout != nullptr) wout[i] = do_sigmoid ? sigmoidf_(wts[i]) : wts[i];
}

}  // namespace

extern "C" int woft_resize_linear_u8(const uint8_t* img, int32_t h, int32_t w, int32_t c, uint8_t* out, int32_t ho,
                                     int32_t wo, float scale_y, float scale_x, void* stream) {
    if (!img || !out || h <= 0 || w <= 0 || c <= 0 || c > 4 || ho <= 0 || wo <= 0) return WOFT_EINVAL;
    const int64_t n = (int64_t)ho * wo;
    hipLaunchKernelGGL(resize_linear_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, (hipStream_t)stream, img,
                       h, w, c, out, ho, wo, scale_y, scale_x);
    return woft_launch_status();
}

extern "C" int woft_convex_weights_at(const float* pts, const int32_t* count, int32_t n_max, const float* wlow,
                                      const float* mask, int32_t ld_mask, int32_t hf, int32_t wf, int32_t crop_top,
                                      int32_t crop_left, int32_t do_sigmoid, float* wsel, void* stream) {
    if (!pts || !wlow || !mask || !wsel || n_max <= 0 || hf <= 0 || wf <= 0 || ld_mask < 576) return WOFT_EINVAL;
    hipLaunchKernelGGL(convex_weights_at_kernel, dim3((n_max + 255) / 256), dim3(256), 0, (hipStream_t)stream, pts, count,
                       n_max, wlow, mask, ld_mask, hf, wf, crop_top, crop_left, do_sigmoid, wsel);
    return woft_launch_status();
}

extern "C" int woft_convex_upsample(const float* coords1, const float* wlow, const float* mask, int32_t ld_mask,
                                    int32_t hf, int32_t wf, int32_t crop_top, int32_t crop_left, int32_t h, int32_t w,
                                    float* flow_up, float* dst, float* wout, int32_t do_sigmoid, void* stream) {
    if (!coords1 || !mask || hf <= 0 || wf <= 0 || h <= 0 || w <= 0 || ld_mask < 576) return WOFT_EINVAL;
    if (crop_top < 0 || crop_left < 0 || crop_top + h > 8 * hf || crop_left + w > 8 * wf) return WOFT_EINVAL;
    const int64_t n = (int64_t)hf * wf;
    hipLaunchKernelGGL(convex_upsample_kernel, dim3((unsigned)ceil_div64(n, 4)), dim3(256), 0, (hipStream_t)stream,
                       coords1, wlow, mask, ld_mask, hf, wf, crop_top, crop_left, h, w, flow_up, dst, wout, do_sigmoid);
    return woft_launch_status();
}

extern "C" int woft_upflow8(const float* coords1, const float* wlow, int32_t hf, int32_t wf, int32_t crop_top,
                            int32_t crop_left, int32_t h, int32_t w, float* flow_up, float* dst, float* wout,
                            int32_t do_sigmoid, void* stream) {
    if (!coords1 || hf <= 0 || wf <= 0 || h <= 0 || w <= 0) return WOFT_EINVAL;
    if (crop_top < 0 || crop_left < 0 || crop_top + h > 8 * hf || crop_left + w > 8 * wf) return WOFT_EINVAL;
    const int64_t n = (int64_t)h * w;
    hipLaunchKernelGGL(upflow8_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, (hipStream_t)stream, coords1,
                       wlow, hf, wf, crop_top, crop_left, h, w, flow_up, dst, wout, do_sigmoid);
    return woft_launch_status();
}

extern "C" int woft_warp_perspective_u8(const uint8_t* img, int32_t h, int32_t w, int32_t c, const double* hinv,
                                        uint8_t* out, uint8_t* valid, int32_t nearest, void* stream) {
    if (!img || !hinv || (!out && !valid) || h <= 0 || w <= 0 || c <= 0 || c > 4) return WOFT_EINVAL;
    if (nearest && !out) return WOFT_EINVAL;
    H9 hi;
    for (int i = 0; i < 9; ++i) hi.v[i] = hinv[i];
    const int64_t n = (int64_t)h * w;
    hipLaunchKernelGGL(warp_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, (hipStream_t)stream, img, h, w,
                       c, hi, out, valid, nearest);
    return woft_launch_status();
}

extern "C" int woft_flow_to_tc(const float* flow, const float* weights, int32_t h, int32_t w, float* dst, float* wout,
                               int32_t do_sigmoid, void* stream) {
    if (!flow || h <= 0 || w <= 0) return WOFT_EINVAL;
    const int64_t n = (int64_t)h * w;
    hipLaunchKernelGGL(flow_to_tc_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, (hipStream_t)stream, flow,
                       weights, h, w, dst, wout, do_sigmoid);
    return woft_launch_status();
}
