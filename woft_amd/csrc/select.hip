// Correspondence masking + order-preserving compaction + Sobol subsampling on the device
// (tracker/YAOF_tracker_single_control.py:287-327 `_mask_coords` / `_mask_coords_flow`;
//  configs/YAOFT_single_control_repRAFT_sub500_noreliableinl_wLSq.py:31-53 `subsampler`).
//
// Reference semantics reproduced exactly:
//   keep[i] = tmask[y_i][x_i]                                   (source pixel i = y_i * gw + x_i inside the mask)
//             and, when check_dst:  not (dx < 0 or dy < 0 or rint(dx) >= W or rint(dy) >= H)
//                                   and (pwmask == NULL or pwmask[rint(dy)][rint(dx)])
//   N = number kept; if n_draw == 0 or n_draw >= N: every kept correspondence is selected, else the
//   selected ranks are the DISTINCT values of (int32) rint(float32(N) * u_k), k < n_draw, in increasing
//   order (the reference sets a boolean mask: original order, duplicates collapse).
// Three launches, no host round trip: (a) flags + per-1024 counts, (b) one workgroup: scan of the
// counts, N, sorted unique rank list, (c) ranks -> output slots.  Outputs are in the H-fit kernel's
// format: pa[k] = (dst_x, dst_y), pb[k] = (src_x, src_y), w[k].
#include "common.h"

namespace {

constexpr int CHUNK = 1024;          // pixels per workgroup in (a) and (c): 256 threads x 4 consecutive
constexpr int MAX_DRAW = 1024;

struct Ws {                          // int32 workspace layout
    int* hdr;                        // [0] N kept, [1] M selected, [2] all-mode
    int* counts;                     // [nb]
    int* offsets;                    // [nb]
    int* sel;                        // [MAX_DRAW] sorted unique ranks
    uint8_t* flags;                  // [n]
};
__host__ __device__ inline Ws ws_layout(int* ws, int nb) {
    Ws o;
    o.hdr = ws;
    o.counts = ws + 4;
    o.offsets = ws + 4 + nb;
    o.sel = ws + 4 + 2 * nb;
    o.flags = (uint8_t*)(ws + 4 + 2 * nb + MAX_DRAW);
    return o;
}

__device__ __forceinline__ int block_exclusive_scan(int v, int* total) {
    // 256 threads; returns exclusive prefix of v, *total = block sum
    __shared__ int wsum[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < wave; ++k) base += wsum[k];
    *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    return base + inc - v;
}

// The correspondences live on the flow grid (gh x gw: the frame, or the frame cropped to a multiple of 8 with
// padding_mode 'crop', optical_flow/raft.py:235-247); the masks have the frame's size (mh x mw).
struct Geo {
    int gh, gw, mh, mw;
};

__device__ __forceinline__ bool keep_flag(const float* __restrict__ dst, const uint8_t* __restrict__ tmask,
                                          const uint8_t* __restrict__ pwmask, const Geo g, int check_dst, int64_t i,
                                          int64_t n) {
    const int64_t mi = (g.gw == g.mw) ? i : (i / g.gw) * g.mw + (i % g.gw);
    bool keep = tmask[mi] != 0;
    if (keep && check_dst) {
        const float dx = dst[i], dy = dst[n + i];
        // NaN compares false below; treat it as out of bounds explicitly
        const bool oob = !(dx >= 0.f) || !(dy >= 0.f) || rintf(dx) >= (float)g.mw || rintf(dy) >= (float)g.mh;
        keep = !oob;
        if (keep && pwmask != nullptr) keep = pwmask[(int64_t)rintf(dy) * g.mw + (int64_t)rintf(dx)] != 0;
    }
    return keep;
}

// flags only (the tracker's generic path: the host compacts with the boolean mask, as the reference does)
__global__ __launch_bounds__(256) void flags_kernel(const float* __restrict__ dst, const uint8_t* __restrict__ tmask,
                                                    const uint8_t* __restrict__ pwmask, Geo g, int check_dst,
                                                    uint8_t* __restrict__ flags) {
    const int64_t n = (int64_t)g.gh * g.gw;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) flags[i] = keep_flag(dst, tmask, pwmask, g, check_dst, i, n) ? 1 : 0;
}

__global__ __launch_bounds__(256) void select_flags_kernel(const float* __restrict__ dst, const uint8_t* __restrict__ tmask,
                                                           const uint8_t* __restrict__ pwmask, Geo g,
                                                           int check_dst, int* ws, int nb) {
    const Ws s = ws_layout(ws, nb);
    const int64_t n = (int64_t)g.gh * g.gw;
    const int64_t i0 = (int64_t)blockIdx.x * CHUNK + threadIdx.x * 4;
    int cnt = 0;
    uint8_t f[4] = {0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int64_t i = i0 + e;
        if (i >= n) continue;
        f[e] = keep_flag(dst, tmask, pwmask, g, check_dst, i, n) ? 1 : 0;
        cnt += f[e];
    }
    if (i0 < n) {
        if (i0 + 3 < n) *(uchar4*)(s.flags + i0) = make_uchar4(f[0], f[1], f[2], f[3]);
        else for (int e = 0; e < 4 && i0 + e < n; ++e) s.flags[i0 + e] = f[e];
    }
    int total;
    block_exclusive_scan(cnt, &total);
    if (threadIdx.x == 0) s.counts[blockIdx.x] = total;
}

__global__ __launch_bounds__(1024) void select_plan_kernel(int* ws, int nb, const float* __restrict__ sobol_u, int n_draw,
                                                           int cap, int* __restrict__ count) {
    const Ws s = ws_layout(ws, nb);
    __shared__ int part[1024];
    __shared__ int keys[MAX_DRAW];
    __shared__ int carry_s;
    const int t = threadIdx.x;
    // ---- exclusive scan of the per-chunk counts (chunks of 1024 with a running carry) ----------
    if (t == 0) carry_s = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nb; b0 += 1024) {
        const int b = b0 + t;
        const int v = (b < nb) ? s.counts[b] : 0;
        part[t] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const int add = (t >= o) ? part[t - o] : 0;
            __syncthreads();
            part[t] += add;
            __syncthreads();
        }
        const int carry = carry_s;
        if (b < nb) s.offsets[b] = carry + part[t] - v;
        __syncthreads();
        if (t == 1023) carry_s = carry + part[1023];
        __syncthreads();
    }
    const int N = carry_s;
    const bool all = (n_draw == 0) || (n_draw >= N);
    // ---- Sobol ranks: rint(float(N) * u_k) in float32, sorted, duplicates removed ---------------
    int key = 0x7fffffff;
    if (!all && t < n_draw) {
        const int r = (int)rintf(__fmul_rn((float)N, sobol_u[t]));
        if (r >= 0 && r < N) key = r;
    }
    keys[t] = key;
    __syncthreads();
    for (int k = 2; k <= MAX_DRAW; k <<= 1)            // bitonic sort, ascending
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int ixj = t ^ j;
            if (ixj > t) {
                const int a = keys[t], b = keys[ixj];
                const bool up = (t & k) == 0;
                if ((a > b) == up) { keys[t] = b; keys[ixj] = a; }
            }
            __syncthreads();
        }
    const int mine = keys[t];
    const int uniq = (!all && mine != 0x7fffffff && (t == 0 || keys[t - 1] != mine)) ? 1 : 0;
    part[t] = uniq;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int add = (t >= o) ? part[t - o] : 0;
        __syncthreads();
        part[t] += add;
        __syncthreads();
    }
    if (uniq) s.sel[part[t] - 1] = mine;
    if (t == 0) {
        const int M = all ? N : part[1023];
        s.hdr[0] = N;
        s.hdr[1] = M;
        s.hdr[2] = all ? 1 : 0;
        count[0] = M < cap ? M : cap;
        count[1] = N;
    }
}

__global__ __launch_bounds__(256) void select_gather_kernel(const float* __restrict__ dst, const float* __restrict__ wgt,
                                                            int h, int w, const int* ws, int nb,
                                                            float* __restrict__ pa, float* __restrict__ pb,
                                                            float* __restrict__ wo, int cap) {
    const Ws s = ws_layout(const_cast<int*>(ws), nb);
    const int64_t n = (int64_t)h * w;
    const int64_t i0 = (int64_t)blockIdx.x * CHUNK + threadIdx.x * 4;
    int f[4] = {0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (i0 + e < n) f[e] = s.flags[i0 + e];
    const int cnt = f[0] + f[1] + f[2] + f[3];
    int total;
    int rank = s.offsets[blockIdx.x] + block_exclusive_scan(cnt, &total);
    const bool all = s.hdr[2] != 0;
    const int M = s.hdr[1];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (!f[e]) continue;
        const int r = rank++;
        int slot = r;
        if (!all) {                                  // binary search r in the sorted unique rank list
            int lo = 0, hi = M - 1;
            slot = -1;
            while (lo <= hi) {
                const int mid = (lo + hi) >> 1;
                const int v = s.sel[mid];
                if (v == r) { slot = mid; break; }
                if (v < r) lo = mid + 1; else hi = mid - 1;
            }
        }
        if (slot < 0 || slot >= cap) continue;
        const int64_t i = i0 + e;
        pa[2 * slot] = dst[i];
        pa[2 * slot + 1] = dst[n + i];
        pb[2 * slot] = (float)(i % w);
        pb[2 * slot + 1] = (float)(i / w);
        if (wo != nullptr) wo[slot] = (wgt != nullptr) ? wgt[i] : 1.f;
    }
}

}  // namespace

extern "C" int64_t woft_tc_select_ws_bytes(int64_t n) {
    const int64_t nb = (n + CHUNK - 1) / CHUNK;
    return (4 + 2 * nb + MAX_DRAW) * 4 + ((n + 15) / 16) * 16;
}

extern "C" int woft_tc_flags(const float* dst, const uint8_t* tmask, const uint8_t* pwmask, int32_t gh, int32_t gw,
                             int32_t mh, int32_t mw, int32_t check_dst, uint8_t* flags, void* stream) {
    if (!tmask || !flags || gh <= 0 || gw <= 0 || gh > mh || gw > mw || (check_dst && !dst)) return WOFT_EINVAL;
    const int64_t n = (int64_t)gh * gw;
    const Geo g = {gh, gw, mh, mw};
    hipLaunchKernelGGL(flags_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dst, tmask,
                       pwmask, g, check_dst, flags);
    return woft_launch_status();
}

extern "C" int woft_tc_select(const float* dst, const float* w, const uint8_t* tmask, const uint8_t* pwmask, int32_t gh,
                              int32_t gw, int32_t mh, int32_t mw, int32_t check_dst, const float* sobol_u,
                              int32_t n_draw, void* ws, float* pa, float* pb, float* wout, int32_t cap, int32_t* count,
                              void* stream) {
    if (!dst || !tmask || !ws || !pa || !pb || !count || gh <= 0 || gw <= 0 || gh > mh || gw > mw || cap <= 0)
        return WOFT_EINVAL;
    if (n_draw < 0 || n_draw > MAX_DRAW || (n_draw > 0 && !sobol_u)) return WOFT_EINVAL;
    const int64_t n = (int64_t)gh * gw;
    const int nb = (int)((n + CHUNK - 1) / CHUNK);
    const Geo g = {gh, gw, mh, mw};
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(select_flags_kernel, dim3(nb), dim3(256), 0, s, dst, tmask, pwmask, g, check_dst, (int*)ws, nb);
    hipLaunchKernelGGL(select_plan_kernel, dim3(1), dim3(1024), 0, s, (int*)ws, nb, sobol_u, n_draw, cap, count);
    hipLaunchKernelGGL(select_gather_kernel, dim3(nb), dim3(256), 0, s, dst, w, gh, gw, (const int*)ws, nb, pa, pb,
                       wout, cap);
    return woft_launch_status();
}
