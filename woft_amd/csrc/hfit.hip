// Weighted / iteratively re-weighted least-squares homography (utils/least_squares_H.py:142-210,
// 268-346) and the re-detection inlier test (least_squares_H.py:474-489, configs/*_wLSq.py:14-21).
//
// The reference solves the (2N x 8) inhomogeneous DLT system with a QR factorisation in fp32.  Here
// the rows are built in fp32 exactly as the reference builds them (Hartley normalisation, plain-w
// row weighting, per-row IRLS re-weighting) and the 9x9 Gram matrix [A b]^T [A b] is accumulated in
// fp64 (streaming, 20 B per correspondence per pass) and solved by an fp64 Cholesky factorisation:
// the least-squares solution is the same, the conditioning loss of the normal equations is absorbed
// by the wider type.  The Gram matrix is a (16 x 2N) x (2N x 16) GEMM (9 live columns): it runs on the
// fp64 matrix cores, v_mfma_f64_16x16x4_f64 -- one MFMA consumes the 4 system rows of 2
// correspondences, lane (c = lane & 15, k = lane >> 4) supplying element c of row k as BOTH operands
// (A[i][k] = B[k][i] = R_k[i]).  One workgroup does the whole fit in a single launch (N <= 500 after
// the Sobol subsampler of the default configs).  Configs without a subsampler fit up to H*W = 2 M (1080p) / 8.3 M
// (4K) correspondences: with a workspace the same arithmetic runs as a streaming multi-workgroup pipeline
// (`hfit_sum` -> `hfit_dist` -> per solve `hfit_gram` + `hfit_solve`; 20 B per correspondence per pass, one
// correspondence per lane with fp64 accumulators on the vector ALUs -- see hfit_gram_kernel --, partial Gram matrices
// reduced in a fixed order -> deterministic).  The same pipeline, one solve per call, serves
// ARBITRARY re-weighting callables (least_squares_H.py:280,337): `woft_hfit_step` takes the per-row re-weights the
// host computed from the residuals of the previous call and returns the residuals A x - b of its own solution.
#include "common.h"
#include <algorithm>

namespace {

constexpr int HT = 1024;           // threads of the fit workgroup
constexpr int NG = 45;             // upper triangle of the 9x9 Gram matrix

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Sum `cnt` per-thread doubles over the workgroup; result valid for every thread in out[0..cnt).
template <int CNT>
__device__ void block_sum(double (&vals)[CNT], double* red /* [16][CNT] */, double* out /* [CNT] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < CNT; ++i) {
        const double s = wave_sum(vals[i]);
        if (lane == 0) red[wave * CNT + i] = s;
    }
    __syncthreads();
    if ((int)threadIdx.x < CNT) {
        double s = 0.0;
        for (int wv = 0; wv < HT / 64; ++wv) s += red[wv * CNT + threadIdx.x];
        out[threadIdx.x] = s;
    }
    __syncthreads();
}

// rows of the DLT system for one correspondence, in fp32 as the reference builds them
__device__ __forceinline__ void build_rows(float x1, float y1, float x2, float y2, float wv, float (&rx)[9],
                                           float (&ry)[9]) {
    rx[0] = 0.f; rx[1] = 0.f; rx[2] = 0.f; rx[3] = -x1; rx[4] = -y1; rx[5] = -1.f; rx[6] = y2 * x1; rx[7] = y2 * y1;
    rx[8] = -y2;
    ry[0] = x1; ry[1] = y1; ry[2] = 1.f; ry[3] = 0.f; ry[4] = 0.f; ry[5] = 0.f; ry[6] = -x2 * x1; ry[7] = -x2 * y1;
    ry[8] = x2;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        rx[k] *= wv;
        ry[k] *= wv;
    }
}

__device__ __forceinline__ float reweight_fn(float r, int mode, float k) {
    const float a = fabsf(r);
    if (mode == 2 && a < k) return 1.f;
    return 1.f / (a + 1e-8f);
}

__global__ __launch_bounds__(HT) void hfit_kernel(const float* __restrict__ pa, const float* __restrict__ pb,
                                                  const float* __restrict__ w, int n_max,
                                                  const int* __restrict__ count, int reweight, float huber_k,
                                                  int n_solves, float* __restrict__ Hout, int* __restrict__ status) {
    __shared__ double red[(HT / 64) * 81];
    __shared__ double tot[NG];
    __shared__ double gram[81];
    __shared__ float sol_s[8];
    __shared__ int fail_s;
    int n = n_max;
    if (count != nullptr) n = min(count[0], n_max);
    if (n < 4) {
        if (threadIdx.x == 0) {
            status[0] = 1;
            for (int i = 0; i < 9; ++i) Hout[i] = nanf("");
        }
        return;
    }
    if (threadIdx.x == 0) fail_s = 0;

    // ---- Hartley normalisation (kornia normalize_points semantics, see oracle/hfit_ref.py) ------
    double s4[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < n; i += HT) {
        s4[0] += (double)pa[2 * i];
        s4[1] += (double)pa[2 * i + 1];
        s4[2] += (double)pb[2 * i];
        s4[3] += (double)pb[2 * i + 1];
    }
    block_sum<4>(s4, red, tot);
    const float m1x = (float)(tot[0] / n), m1y = (float)(tot[1] / n);
    const float m2x = (float)(tot[2] / n), m2y = (float)(tot[3] / n);
    __syncthreads();
    double d2[2] = {0.0, 0.0};
    for (int i = threadIdx.x; i < n; i += HT) {
        const float ax = pa[2 * i] - m1x, ay = pa[2 * i + 1] - m1y;
        const float bx = pb[2 * i] - m2x, by = pb[2 * i + 1] - m2y;
        d2[0] += (double)sqrtf(ax * ax + ay * ay);
        d2[1] += (double)sqrtf(bx * bx + by * by);
    }
    block_sum<2>(d2, red, tot);
    const float s1 = sqrtf(2.0f) / ((float)(tot[0] / n) + 1e-8f);
    const float s2 = sqrtf(2.0f) / ((float)(tot[1] / n) + 1e-8f);
    const float t1x = -s1 * m1x, t1y = -s1 * m1y, t2x = -s2 * m2x, t2y = -s2 * m2y;
    __syncthreads();

    // ---- (re-)weighted normal equations, n_solves solves -----------------------------------------
    typedef double f64x4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ec = lane & 15, ek = lane >> 4;            // element column, row-in-quad (point = ek >> 1, x/y row = ek & 1)
    for (int it = 0; it < n_solves; ++it) {
        float sol[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) sol[k] = (it > 0) ? sol_s[k] : 0.f;
        f64x4 g4 = {0.0, 0.0, 0.0, 0.0};
        for (int base = wave * 2; base < n; base += (HT / 64) * 2) {
            const int i = base + (ek >> 1);
            double val = 0.0;
            if (i < n && ec < 9) {
                const float x1 = s1 * pa[2 * i] + t1x, y1 = s1 * pa[2 * i + 1] + t1y;
                const float x2 = s2 * pb[2 * i] + t2x, y2 = s2 * pb[2 * i + 1] + t2y;
                const float wv = (w != nullptr) ? w[i] : 1.f;
                float rx[9], ry[9];
                build_rows(x1, y1, x2, y2, wv, rx, ry);
                float q = 1.f;
                if (it > 0 && reweight != 0) {
                    float resx = -rx[8], resy = -ry[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        resx += rx[k] * sol[k];
                        resy += ry[k] * sol[k];
                    }
                    q = sqrtf(reweight_fn((ek & 1) ? resy : resx, reweight, huber_k));
                }
                float e = 0.f;
#pragma unroll
                for (int k = 0; k < 9; ++k)
                    if (k == ec) e = (ek & 1) ? ry[k] : rx[k];
                val = (double)(e * q);
            }
            g4 = __builtin_amdgcn_mfma_f64_16x16x4f64(val, val, g4, 0, 0, 0);
        }
        // D layout of the f64 MFMA: column = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = ek + 4 * r;
            if (row < 9 && ec < 9) red[wave * 81 + row * 9 + ec] = g4[r];
        }
        __syncthreads();
        if (threadIdx.x < 81) {
            double sacc = 0.0;
            for (int wv = 0; wv < HT / 64; ++wv) sacc += red[wv * 81 + threadIdx.x];
            gram[threadIdx.x] = sacc;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            // Cholesky of G[0:8,0:8] = L L^T, solve L L^T x = G[0:8,8]
            double L[8][8], rhs[8];
            double G[9][9];
            for (int a = 0; a < 9; ++a)
                for (int b = 0; b < 9; ++b) G[a][b] = gram[a * 9 + b];
            bool ok = true;
            for (int i = 0; i < 8 && ok; ++i) {
                for (int j = 0; j <= i; ++j) {
                    double s = G[i][j];
                    for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
                    if (i == j) {
                        if (!(s > 0.0)) { ok = false; break; }
                        L[i][i] = sqrt(s);
                    } else {
                        L[i][j] = s / L[j][j];
                    }
                }
            }
            if (ok) {
                for (int i = 0; i < 8; ++i) {
                    double s = G[i][8];
                    for (int k = 0; k < i; ++k) s -= L[i][k] * rhs[k];
                    rhs[i] = s / L[i][i];
                }
                for (int i = 7; i >= 0; --i) {
                    double s = rhs[i];
                    for (int k = i + 1; k < 8; ++k) s -= L[k][i] * rhs[k];
                    rhs[i] = s / L[i][i];
                }
                for (int i = 0; i < 8; ++i) sol_s[i] = (float)rhs[i];
            } else {
                fail_s = 1;
            }
        }
        __syncthreads();
        if (fail_s) break;
    }

    if (threadIdx.x == 0) {
        if (fail_s) {
            status[0] = 2;
            for (int i = 0; i < 9; ++i) Hout[i] = nanf("");
            return;
        }
        // H = T2^-1 * Hn * T1, then H / (h33 + 1e-8)   (least_squares_H.py:204-209)
        const double hn[9] = {sol_s[0], sol_s[1], sol_s[2], sol_s[3], sol_s[4], sol_s[5], sol_s[6], sol_s[7], 1.0};
        const double T1[9] = {s1, 0, t1x, 0, s1, t1y, 0, 0, 1};
        const double i2 = 1.0 / (double)s2;
        const double T2i[9] = {i2, 0, -(double)t2x * i2, 0, i2, -(double)t2y * i2, 0, 0, 1};
        double tmp[9], hh[9];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                double s = 0;
                for (int k = 0; k < 3; ++k) s += hn[r * 3 + k] * T1[k * 3 + c];
                tmp[r * 3 + c] = s;
            }
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                double s = 0;
                for (int k = 0; k < 3; ++k) s += T2i[r * 3 + k] * tmp[k * 3 + c];
                hh[r * 3 + c] = s;
            }
        const double den = hh[8] + 1e-8;
        for (int i = 0; i < 9; ++i) Hout[i] = (float)(hh[i] / den);
        status[0] = 0;
    }
}

// frac[0] = mean_i( || proj(H, A_i) - B_i || <= thr )
__global__ __launch_bounds__(HT) void inlier_frac_kernel(const float* __restrict__ pa, const float* __restrict__ pb,
                                                         int n_max, const int* __restrict__ count,
                                                         const float* __restrict__ H, float thr,
                                                         float* __restrict__ frac) {
    __shared__ double red[HT / 64];
    __shared__ double tot[1];
    int n = n_max;
    if (count != nullptr) n = min(count[0], n_max);
    float h[9];
    for (int i = 0; i < 9; ++i) h[i] = H[i];
    double c[1] = {0.0};
    for (int i = threadIdx.x; i < n; i += HT) {
        const float x = pa[2 * i], y = pa[2 * i + 1];
        const float px = h[0] * x + h[1] * y + h[2], py = h[3] * x + h[4] * y + h[5], pz = h[6] * x + h[7] * y + h[8];
        const float sc = (fabsf(pz) > 1e-8f) ? 1.f / (pz + 1e-8f) : 1.f;
        const float dx = sc * px - pb[2 * i], dy = sc * py - pb[2 * i + 1];
        c[0] += (sqrtf(dx * dx + dy * dy) <= thr) ? 1.0 : 0.0;
    }
    block_sum<1>(c, red, tot);
    if (threadIdx.x == 0) frac[0] = (n > 0) ? (float)(tot[0] / n) : 0.f;
}

// ------------------------------------------------------------------------------------------------------------------
// Streaming multi-workgroup pipeline (large N, external re-weights).  Workspace (doubles first):
//   psum[G][4] | pdist[G][2] | pgram[G][81] | then floats: sol[8] | norm[8] (s1, t1x, t1y, s2, t2x, t2y, n, -)
// ------------------------------------------------------------------------------------------------------------------
constexpr int MT = 256;            // threads per workgroup of the streaming kernels
constexpr int MAXG = 1024;         // workgroups (4 per CU)

struct MWs {
    double* psum;
    double* pdist;
    double* pgram;
    float* sol;
    float* norm;
};
__host__ __device__ inline MWs mws_layout(void* ws) {
    MWs o;
    o.psum = (double*)ws;
    o.pdist = o.psum + MAXG * 4;
    o.pgram = o.pdist + MAXG * 2;
    o.sol = (float*)(o.pgram + MAXG * 81);
    o.norm = o.sol + 8;
    return o;
}
inline int mws_groups(int n_max) { return (int)std::min<int64_t>(MAXG, std::max<int64_t>(1, ((int64_t)n_max + 2047) / 2048)); }

template <int CNT>
__device__ void block_sum_m(double (&vals)[CNT], double* red /* [MT/64][CNT] */, double* out /* global [CNT] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < CNT; ++i) {
        const double s = wave_sum(vals[i]);
        if (lane == 0) red[wave * CNT + i] = s;
    }
    __syncthreads();
    if ((int)threadIdx.x < CNT) {
        double s = 0.0;
        for (int wv = 0; wv < MT / 64; ++wv) s += red[wv * CNT + threadIdx.x];
        out[threadIdx.x] = s;
    }
}

__device__ __forceinline__ int fit_count(const int* count, int n_max) { return count ? min(count[0], n_max) : n_max; }

__global__ __launch_bounds__(MT) void hfit_sum_kernel(const float* __restrict__ pa, const float* __restrict__ pb, int n_max,
                                                      const int* __restrict__ count, void* ws) {
    __shared__ double red[(MT / 64) * 4];
    const MWs s = mws_layout(ws);
    const int n = fit_count(count, n_max);
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t i = (int64_t)blockIdx.x * MT + threadIdx.x; i < n; i += (int64_t)gridDim.x * MT) {
        const float2 a = ((const float2*)pa)[i], b = ((const float2*)pb)[i];
        v[0] += (double)a.x; v[1] += (double)a.y; v[2] += (double)b.x; v[3] += (double)b.y;
    }
    block_sum_m<4>(v, red, s.psum + blockIdx.x * 4);
}

// total of per-workgroup partials part[G][CNT] -> out[CNT] (LDS), by the whole workgroup in a fixed order: lane t sums
// the groups t, t + MT, ... , then the usual wave / workgroup tree.  Every workgroup repeats it (G x CNT doubles out of
// L2) and gets the same bits.  (A serial loop in CNT lanes is G dependent L2 round trips: it WAS the streaming fit's
// whole cost, ~250 us per kernel at G = 1024.)
template <int CNT>
__device__ __forceinline__ void partial_total(const double* __restrict__ part, int G, double* out /* LDS [CNT] */) {
    __shared__ double tred[(MT / 64) * CNT];
    double acc[CNT];
#pragma unroll
    for (int c = 0; c < CNT; ++c) acc[c] = 0.0;
    for (int g = threadIdx.x; g < G; g += MT)
#pragma unroll
        for (int c = 0; c < CNT; ++c) acc[c] += part[(int64_t)g * CNT + c];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < CNT; ++c) {
        const double t = wave_sum(acc[c]);
        if (lane == 0) tred[wave * CNT + c] = t;
    }
    __syncthreads();
    if ((int)threadIdx.x < CNT) {
        double t = 0.0;
        for (int wv = 0; wv < MT / 64; ++wv) t += tred[wv * CNT + threadIdx.x];
        out[threadIdx.x] = t;
    }
    __syncthreads();
}

// means from the partial sums
__device__ __forceinline__ void fit_means(const MWs& s, int n, double* sh /* [4] */) {
    partial_total<4>(s.psum, (int)gridDim.x, sh);
    if (threadIdx.x < 4) sh[threadIdx.x] /= n;
    __syncthreads();
}

__global__ __launch_bounds__(MT) void hfit_dist_kernel(const float* __restrict__ pa, const float* __restrict__ pb, int n_max,
                                                       const int* __restrict__ count, void* ws) {
    __shared__ double red[(MT / 64) * 2];
    __shared__ double mean[4];
    const MWs s = mws_layout(ws);
    const int n = fit_count(count, n_max);
    if (n < 1) return;
    fit_means(s, n, mean);
    const float m1x = (float)mean[0], m1y = (float)mean[1], m2x = (float)mean[2], m2y = (float)mean[3];
    double v[2] = {0.0, 0.0};
    for (int64_t i = (int64_t)blockIdx.x * MT + threadIdx.x; i < n; i += (int64_t)gridDim.x * MT) {
        const float2 a = ((const float2*)pa)[i], b = ((const float2*)pb)[i];
        const float ax = a.x - m1x, ay = a.y - m1y, bx = b.x - m2x, by = b.y - m2y;
        v[0] += (double)sqrtf(ax * ax + ay * ay);
        v[1] += (double)sqrtf(bx * bx + by * by);
    }
    block_sum_m<2>(v, red, s.pdist + blockIdx.x * 2);
}

// normalisation parameters from the partials -> shared nf[6] (same formulas as the single-workgroup kernel)
__device__ __forceinline__ void fit_norm(const MWs& s, int n, double* mean /* [4] */, double* dist /* [2] */, float* nf) {
    fit_means(s, n, mean);
    partial_total<2>(s.pdist, (int)gridDim.x, dist);
    if (threadIdx.x == 0) {
        dist[0] /= n;
        dist[1] /= n;
        const float m1x = (float)mean[0], m1y = (float)mean[1], m2x = (float)mean[2], m2y = (float)mean[3];
        const float s1 = sqrtf(2.0f) / ((float)dist[0] + 1e-8f), s2 = sqrtf(2.0f) / ((float)dist[1] + 1e-8f);
        nf[0] = s1; nf[1] = -s1 * m1x; nf[2] = -s1 * m1y; nf[3] = s2; nf[4] = -s2 * m2x; nf[5] = -s2 * m2y;
    }
    __syncthreads();
}

// partial Gram matrix of this workgroup's share of the rows.  Row re-weights: `rew` (2 per correspondence, external)
// or, when use_sol, sqrt(reweight_fn(residual of `sol`)) of the built-in losses.
//
// Streaming form: ONE correspondence per lane per turn, the 36 structurally non-zero entries of the symmetric 9x9 matrix as
// fp64 accumulators on the vector ALUs (rows x: columns 3..8, rows y: columns 0..2, 6..8 -> 2 x 21 fp64 FMAs per
// correspondence).  The fp64 matrix-core form of the one-workgroup kernel spends a whole wave instruction on TWO
// correspondences (K = 4 system rows per v_mfma_f64_16x16x4, 81 of 256 outputs live) with every lane rebuilding both rows;
// this is HBM-bound streaming work, not a GEMM (profiles/r02_hfit_fullframe.txt: 3.2 TB/s algorithmic at N = 8.3 M).
// Same fp32 rows, same (double)(row * q) operands, fp64 products and sums: only the summation order differs.
__global__ __launch_bounds__(MT) void hfit_gram_kernel(const float* __restrict__ pa, const float* __restrict__ pb,
                                                       const float* __restrict__ w, int n_max,
                                                       const int* __restrict__ count, const float* __restrict__ rew,
                                                       int reweight, float huber_k, int use_sol, void* ws) {
    __shared__ double mean[4], dist[2];
    __shared__ float nf[6];
    __shared__ double red[(MT / 64) * 81];
    const MWs s = mws_layout(ws);
    const int n = fit_count(count, n_max);
    if (n < 4) return;
    fit_norm(s, n, mean, dist, nf);
    const float s1 = nf[0], t1x = nf[1], t1y = nf[2], s2 = nf[3], t2x = nf[4], t2y = nf[5];
    if (blockIdx.x == 0 && threadIdx.x < 6) s.norm[threadIdx.x] = nf[threadIdx.x];
    float sol[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) sol[k] = use_sol ? s.sol[k] : 0.f;
    // columns touched by an x row / a y row, and the accumulators of their pairwise products (upper triangle)
    constexpr int CX[6] = {3, 4, 5, 6, 7, 8}, CY[6] = {0, 1, 2, 6, 7, 8};
    double gx[21], gy[21];
#pragma unroll
    for (int k = 0; k < 21; ++k) gx[k] = gy[k] = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * MT + threadIdx.x; i < n; i += (int64_t)gridDim.x * MT) {
        const float2 a = ((const float2*)pa)[i], b = ((const float2*)pb)[i];
        const float x1 = s1 * a.x + t1x, y1 = s1 * a.y + t1y, x2 = s2 * b.x + t2x, y2 = s2 * b.y + t2y;
        const float wv = (w != nullptr) ? w[i] : 1.f;
        float rx[9], ry[9];
        build_rows(x1, y1, x2, y2, wv, rx, ry);
        float qx = 1.f, qy = 1.f;
        if (rew != nullptr) {
            const float2 q2 = ((const float2*)rew)[i];
            qx = q2.x;
            qy = q2.y;
        } else if (use_sol && reweight != 0) {
            float resx = -rx[8], resy = -ry[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                resx += rx[k] * sol[k];
                resy += ry[k] * sol[k];
            }
            qx = sqrtf(reweight_fn(resx, reweight, huber_k));
            qy = sqrtf(reweight_fn(resy, reweight, huber_k));
        }
        double ex[6], ey[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            ex[k] = (double)(rx[CX[k]] * qx);
            ey[k] = (double)(ry[CY[k]] * qy);
        }
        int t = 0;
#pragma unroll
        for (int p0 = 0; p0 < 6; ++p0)
#pragma unroll
            for (int p1 = p0; p1 < 6; ++p1, ++t) {
                gx[t] += ex[p0] * ex[p1];
                gy[t] += ey[p0] * ey[p1];
            }
    }
    // workgroup reduction -> the full symmetric 9x9 partial matrix (zeros where no row has both columns)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = threadIdx.x; k < (MT / 64) * 81; k += MT) red[k] = 0.0;
    __syncthreads();
    {
        int t = 0;
#pragma unroll
        for (int p0 = 0; p0 < 6; ++p0)
#pragma unroll
            for (int p1 = p0; p1 < 6; ++p1, ++t) {
                const double sx = wave_sum(gx[t]), sy = wave_sum(gy[t]);
                if (lane == 0) {
                    red[wave * 81 + CX[p0] * 9 + CX[p1]] += sx;
                    red[wave * 81 + CY[p0] * 9 + CY[p1]] += sy;     // (x and y rows share the columns 6..8: same cell, two adds)
                }
            }
    }
    __syncthreads();
    if (threadIdx.x < 81) {
        const int r = threadIdx.x / 9, c = threadIdx.x % 9;
        const int idx = (r <= c) ? r * 9 + c : c * 9 + r;              // mirror the upper triangle
        double sacc = 0.0;
        for (int wv = 0; wv < MT / 64; ++wv) sacc += red[wv * 81 + idx];
        s.pgram[(int64_t)blockIdx.x * 81 + threadIdx.x] = sacc;
    }
}

__device__ bool cholesky_solve8(const double* gram /* [81] */, float* sol_out) {
    double L[8][8], rhs[8];
    for (int i = 0; i < 8; ++i) {
        for (int j = 0; j <= i; ++j) {
            double s = gram[i * 9 + j];
            for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
            if (i == j) {
                if (!(s > 0.0)) return false;
                L[i][i] = sqrt(s);
            } else {
                L[i][j] = s / L[j][j];
            }
        }
    }
    for (int i = 0; i < 8; ++i) {
        double s = gram[i * 9 + 8];
        for (int k = 0; k < i; ++k) s -= L[i][k] * rhs[k];
        rhs[i] = s / L[i][i];
    }
    for (int i = 7; i >= 0; --i) {
        double s = rhs[i];
        for (int k = i + 1; k < 8; ++k) s -= L[k][i] * rhs[k];
        rhs[i] = s / L[i][i];
    }
    for (int i = 0; i < 8; ++i) sol_out[i] = (float)rhs[i];
    return true;
}

__device__ void denormalise(const float* sol, float s1, float t1x, float t1y, float s2, float t2x, float t2y, float* Hout) {
    // H = T2^-1 * Hn * T1, then H / (h33 + 1e-8)   (least_squares_H.py:204-209, 340-345)
    const double hn[9] = {sol[0], sol[1], sol[2], sol[3], sol[4], sol[5], sol[6], sol[7], 1.0};
    const double T1[9] = {s1, 0, t1x, 0, s1, t1y, 0, 0, 1};
    const double i2 = 1.0 / (double)s2;
    const double T2i[9] = {i2, 0, -(double)t2x * i2, 0, i2, -(double)t2y * i2, 0, 0, 1};
    double tmp[9], hh[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += hn[r * 3 + k] * T1[k * 3 + c];
            tmp[r * 3 + c] = s;
        }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += T2i[r * 3 + k] * tmp[k * 3 + c];
            hh[r * 3 + c] = s;
        }
    const double den = hh[8] + 1e-8;
    for (int i = 0; i < 9; ++i) Hout[i] = (float)(hh[i] / den);
}

// one workgroup: reduce the partial Gram matrices (fixed order), solve, publish sol (and H / status)
constexpr int ST = 1024;           // threads of the solve kernel: 16 waves share the partial matrices
__global__ __launch_bounds__(ST) void hfit_solve_kernel(int n_max, const int* __restrict__ count, int groups, void* ws,
                                                         float* __restrict__ Hout, int* __restrict__ status) {
    __shared__ double gram[81];
    const MWs s = mws_layout(ws);
    const int n = fit_count(count, n_max);
    if (n < 4) {
        if (threadIdx.x == 0) {
            status[0] = 1;
            for (int i = 0; i < 9; ++i) Hout[i] = nanf("");
        }
        return;
    }
    {   // wave v sums the groups v, v + 16, ... (lanes = matrix entries, 81 = 64 + 17), then the waves in order
        __shared__ double part[(ST / 64) * 81];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        double t0 = 0.0, t1 = 0.0;
#pragma unroll 4
        for (int g = wave; g < groups; g += ST / 64) {
            t0 += s.pgram[(int64_t)g * 81 + lane];
            if (lane < 17) t1 += s.pgram[(int64_t)g * 81 + 64 + lane];
        }
        part[wave * 81 + lane] = t0;
        if (lane < 17) part[wave * 81 + 64 + lane] = t1;
        __syncthreads();
        if (threadIdx.x < 81) {
            double t = 0.0;
            for (int wv = 0; wv < ST / 64; ++wv) t += part[wv * 81 + threadIdx.x];
            gram[threadIdx.x] = t;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (status[0] == 2) return;                       // an earlier solve of this fit was singular
        float sol[8];
        if (!cholesky_solve8(gram, sol)) {
            status[0] = 2;
            for (int i = 0; i < 9; ++i) Hout[i] = nanf("");
            return;
        }
        for (int i = 0; i < 8; ++i) s.sol[i] = sol[i];
        denormalise(sol, s.norm[0], s.norm[1], s.norm[2], s.norm[3], s.norm[4], s.norm[5], Hout);
        status[0] = 0;
    }
}

// res[2i], res[2i+1] = rows of (A x - b) for the weighted (not re-weighted) system, least_squares_H.py:334
__global__ __launch_bounds__(MT) void hfit_resid_kernel(const float* __restrict__ pa, const float* __restrict__ pb,
                                                        const float* __restrict__ w, int n_max,
                                                        const int* __restrict__ count, const void* ws,
                                                        float* __restrict__ res) {
    const MWs s = mws_layout(const_cast<void*>(ws));
    const int n = fit_count(count, n_max);
    const int64_t i = (int64_t)blockIdx.x * MT + threadIdx.x;
    if (i >= n) return;
    const float s1 = s.norm[0], t1x = s.norm[1], t1y = s.norm[2], s2 = s.norm[3], t2x = s.norm[4], t2y = s.norm[5];
    const float2 a = ((const float2*)pa)[i], b = ((const float2*)pb)[i];
    float rx[9], ry[9];
    build_rows(s1 * a.x + t1x, s1 * a.y + t1y, s2 * b.x + t2x, s2 * b.y + t2y, (w != nullptr) ? w[i] : 1.f, rx, ry);
    float resx = -rx[8], resy = -ry[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        resx += rx[k] * s.sol[k];
        resy += ry[k] * s.sol[k];
    }
    ((float2*)res)[i] = make_float2(resx, resy);
}

}  // namespace

extern "C" int64_t woft_hfit_ws_bytes(void) { return (int64_t)MAXG * (4 + 2 + 81) * 8 + 16 * 4; }

extern "C" int woft_hfit(const float* pa, const float* pb, const float* w, int32_t n_max, const int32_t* count,
                         int32_t reweight, float huber_k, int32_t n_irls, void* ws, float* Hout, int32_t* status,
                         void* stream) {
    if (!pa || !pb || !Hout || !status || n_max < 0 || reweight < 0 || reweight > 2 || n_irls < 0) return WOFT_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (ws == nullptr || n_max <= WOFT_HFIT_SINGLE_MAX) {
        hipLaunchKernelGGL(hfit_kernel, dim3(1), dim3(HT), 0, st, pa, pb, w, n_max, count, reweight, huber_k, n_irls + 1,
                           Hout, status);
        return woft_launch_status();
    }
    const int G = mws_groups(n_max);
    (void)hipMemsetAsync(status, 0, sizeof(int32_t), st);
    hipLaunchKernelGGL(hfit_sum_kernel, dim3(G), dim3(MT), 0, st, pa, pb, n_max, count, ws);
    hipLaunchKernelGGL(hfit_dist_kernel, dim3(G), dim3(MT), 0, st, pa, pb, n_max, count, ws);
    const int solves = (reweight != 0) ? n_irls + 1 : 1;     // without a loss every pass re-solves the same system
    for (int it = 0; it < solves; ++it) {
        hipLaunchKernelGGL(hfit_gram_kernel, dim3(G), dim3(MT), 0, st, pa, pb, w, n_max, count, (const float*)nullptr,
                           reweight, huber_k, it > 0 ? 1 : 0, ws);
        hipLaunchKernelGGL(hfit_solve_kernel, dim3(1), dim3(ST), 0, st, n_max, count, G, ws, Hout, status);
    }
    return woft_launch_status();
}

extern "C" int woft_hfit_step(const float* pa, const float* pb, const float* w, int32_t n, const float* rew, int32_t first,
                              void* ws, float* res, float* Hout, int32_t* status, void* stream) {
    if (!pa || !pb || !ws || !Hout || !status || n < 0) return WOFT_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int G = mws_groups(n);
    if (first) {
        (void)hipMemsetAsync(status, 0, sizeof(int32_t), st);
        hipLaunchKernelGGL(hfit_sum_kernel, dim3(G), dim3(MT), 0, st, pa, pb, n, (const int*)nullptr, ws);
        hipLaunchKernelGGL(hfit_dist_kernel, dim3(G), dim3(MT), 0, st, pa, pb, n, (const int*)nullptr, ws);
    }
    hipLaunchKernelGGL(hfit_gram_kernel, dim3(G), dim3(MT), 0, st, pa, pb, w, n, (const int*)nullptr, rew, 0, 0.f, 0, ws);
    hipLaunchKernelGGL(hfit_solve_kernel, dim3(1), dim3(ST), 0, st, n, (const int*)nullptr, G, ws, Hout, status);
    if (res != nullptr && n >= 4)
        hipLaunchKernelGGL(hfit_resid_kernel, dim3((n + MT - 1) / MT), dim3(MT), 0, st, pa, pb, w, n, (const int*)nullptr,
                           (const void*)ws, res);
    return woft_launch_status();
}

extern "C" int woft_inlier_frac(const float* pa, const float* pb, int32_t n_max, const int32_t* count, const float* H,
                                float thr, float* frac, void* stream) {
    if (!pa || !pb || !H || !frac || n_max < 0) return WOFT_EINVAL;
    hipLaunchKernelGGL(inlier_frac_kernel, dim3(1), dim3(HT), 0, (hipStream_t)stream, pa, pb, n_max, count, H, thr,
                       frac);
    return woft_launch_status();
}
