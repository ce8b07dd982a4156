// Micro-benchmark: how fast can 256-thread workgroups write a [P][N] fp32 matrix in GEMM-epilogue-shaped pieces?
// (hipcc --offload-arch=gfx950 -O3 -I woft_amd/csrc tools/micro/wr_pattern.hip -o gpurun_out/wr_pattern)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "conv_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: 128x128 tile, wave w owns the 64x64 quadrant, 8 rows x 128 B per store instruction (the conv epilogue)
// MODE 1: 128x128 tile, 2 rows x 512 B per store instruction (wave w owns rows w*32 .. w*32+31)
// MODE 2: 64x256 tile, 1 row x 1 KB per store instruction
// MODE 3: linear fill (grid-stride)
template <int MODE, bool NT, bool REMAP>
__global__ __launch_bounds__(256) void wr_kernel(float* out, int64_t P, int64_t N, int64_t ld, int m_tiles, int n_tiles) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
    auto st = [&](float* p) {
        if (NT) __builtin_nontemporal_store(v, (f32x4*)p);
        else *(f32x4*)p = v;
    };
    if (MODE == 3) {
        const int64_t total = P * ld / 4;
        for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < total; i += (int64_t)gridDim.x * 256) st(out + i * 4);
        return;
    }
    int mt, nt;
    if (REMAP) woft::tile_of_block(blockIdx.x, m_tiles, n_tiles, mt, nt);
    else { mt = blockIdx.x / n_tiles; nt = blockIdx.x % n_tiles; }
    if (MODE == 0) {
        const int64_t m0 = (int64_t)mt * 128 + (wave >> 1) * 64, n0 = (int64_t)nt * 128 + (wave & 1) * 64;
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
                for (int r = 0; r < 4; ++r) {
                    const int64_t m = m0 + i * 32 + r * 8 + (lane >> 3), n = n0 + j * 32 + (lane & 7) * 4;
                    if (m < P && n < N) st(out + m * ld + n);
                }
    } else if (MODE == 1) {
        const int64_t m0 = (int64_t)mt * 128 + wave * 32, n0 = (int64_t)nt * 128;
        for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + r * 2 + (lane >> 5), n = n0 + (lane & 31) * 4;
            if (m < P && n < N) st(out + m * ld + n);
        }
    } else {
        const int64_t m0 = (int64_t)mt * 64 + wave * 16, n0 = (int64_t)nt * 256;
        for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + r, n = n0 + lane * 4;
            if (m < P && n < N) st(out + m * ld + n);
        }
    }
}

template <int MODE, bool NT, bool REMAP>
void run(const char* name, float* out, int64_t P, int64_t N) {
    const int bm = MODE == 2 ? 64 : 128, bn = MODE == 2 ? 256 : 128;
    const int m_tiles = (int)((P + bm - 1) / bm), n_tiles = (int)((N + bn - 1) / bn);
    const unsigned grid = MODE == 3 ? 256 * 16 : (unsigned)(m_tiles * n_tiles);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9;
    for (int it = 0; it < 5; ++it) {
        hipEventRecord(a);
        hipLaunchKernelGGL((wr_kernel<MODE, NT, REMAP>), dim3(grid), dim3(256), 0, 0, out, P, N, N, m_tiles, n_tiles);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (it && ms < best) best = ms;
    }
    printf("%-44s %8.1f us  %6.2f TB/s\n", name, best * 1e3, (double)P * N * 4 / best / 1e9);
}

int main() {
    const int64_t P = 32400, N = 32640;
    float* out;
    hipMalloc(&out, P * N * 4);
    run<3, false, false>("linear fill", out, P, N);
    run<3, true, false>("linear fill nt", out, P, N);
    run<0, false, true>("128x128, 8 rows x 128 B, xcd remap", out, P, N);
    run<0, true, true>("128x128, 8 rows x 128 B, xcd remap, nt", out, P, N);
    run<0, false, false>("128x128, 8 rows x 128 B, row-major tiles", out, P, N);
    run<1, false, true>("128x128, 2 rows x 512 B, xcd remap", out, P, N);
    run<1, true, true>("128x128, 2 rows x 512 B, xcd remap, nt", out, P, N);
    run<1, false, false>("128x128, 2 rows x 512 B, row-major tiles", out, P, N);
    run<2, false, true>("64x256, 1 row x 1 KB, xcd remap", out, P, N);
    run<2, true, true>("64x256, 1 row x 1 KB, xcd remap, nt", out, P, N);
    run<2, false, false>("64x256, 1 row x 1 KB, row-major tiles", out, P, N);
    return 0;
}
