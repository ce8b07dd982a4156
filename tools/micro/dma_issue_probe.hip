// What does one LDS-DMA piece (global_load_lds_dwordx4, 1 KiB per wave instruction) cost the wave that issues it, alone and beside
// MFMAs -- and does the M0 handling matter?  One workgroup per CU (4 or 8 waves), every wave streams 4-KiB steps (four pieces) from an
// L2-resident buffer through its own 3-stage LDS ring, NIT steps, and runs NMFMA independent-accumulator MFMAs per step.
//   mode 0: no loads at all (the MFMAs alone)
//   mode 1: four pieces per step, M0 saved / set / restored around every piece (lds_dma16 of csrc/dma.h), all four before the MFMAs
//   mode 2: M0 set ONCE per step, the four pieces through the instruction's immediate offset (0 / 1024 / 2048 / 3072 -- it moves the
//           LDS destination AND the memory address: the lane offsets of piece q are biased by -1024 q)
//   mode 3: as 2, one piece after every NMFMA / 4 MFMAs instead of four in a row
//   mode 4: four plain global_load_dwordx4 into registers instead (coalesced 1 KiB each), consumed by a dummy add
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/micro/dma_issue_probe.bin tools/micro/dma_issue_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint32_t lds_addr_of(const void* p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p; }
__device__ __forceinline__ void dma_m0_each(const void* gbase, uint32_t lane_off, uint32_t lds_addr) {
    const uint64_t gb = (uint64_t)(uintptr_t)gbase;
    const uint64_t gu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(gb >> 32)) << 32) |
                        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)gb);
    uint32_t saved;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(saved) : "s"(lds_addr), "v"(lane_off), "s"(gu) : "memory", "vcc");
}
template <int OFF>
__device__ __forceinline__ void dma_off(uint64_t gu, uint32_t lane_off) {       // M0 already holds the stage address
    asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" ::"v"(lane_off), "s"(gu), "n"(OFF) : "memory");
}
__device__ __forceinline__ uint32_t set_m0(uint32_t v) {
    uint32_t saved;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0" : "=&s"(saved) : "s"(v) : "memory");
    return saved;
}
__device__ __forceinline__ void restore_m0(uint32_t v) { asm volatile("s_mov_b32 m0, %0" ::"s"(v) : "memory"); }

template <int MODE, int NMFMA, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64, NWAVES / 4) void probe(const char* __restrict__ src, uint32_t src_bytes, int nit, uint32_t* out_cycles,
                                                                  float* sink) {
    __shared__ __attribute__((aligned(16))) char ring[NWAVES * 3 * 4096];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t ring_addr = lds_addr_of(ring) + wave * 3 * 4096;
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(lane + i); b[i] = (__bf16)(float)(lane - i); }
    // every wave walks its own 256-KiB window of the buffer (L2 / MALL resident after the first pass)
    const uint32_t win = ((blockIdx.x * NWAVES + wave) * 262144u) % (src_bytes - 262144u);
    const uint64_t gb = (uint64_t)(uintptr_t)(src + win);
    const uint64_t gu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(gb >> 32)) << 32) |
                        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)gb);
    f32x4 keep = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < nit; ++it) {
        const uint32_t step_off = (uint32_t)((it * 4096) & 262143);
        const uint32_t lane_off = step_off + lane * 16;
        const uint32_t st = ring_addr + (uint32_t)(it % 3) * 4096u;
        if (MODE == 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) dma_m0_each((const void*)gu, lane_off + q * 1024, st + q * 1024);
        } else if (MODE == 2) {
            const uint32_t saved = set_m0(st);
            dma_off<0>(gu, lane_off); dma_off<1024>(gu, lane_off); dma_off<2048>(gu, lane_off); dma_off<3072>(gu, lane_off);
            restore_m0(saved);
        } else if (MODE == 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) keep += *(const f32x4*)(src + win + lane_off + q * 1024);
        }
        if (MODE == 1 || MODE == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // two steps stay in flight
        uint32_t saved3 = 0;
        if (MODE == 3) saved3 = set_m0(st);
#pragma unroll
        for (int m = 0; m < NMFMA; ++m) {
            acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
            if (MODE == 3 && NMFMA >= 4) {
                if (m == 0) dma_off<0>(gu, lane_off);
                if (m == NMFMA / 4) dma_off<1024>(gu, lane_off);
                if (m == 2 * (NMFMA / 4)) dma_off<2048>(gu, lane_off);
                if (m == 3 * (NMFMA / 4)) dma_off<3072>(gu, lane_off);
            }
        }
        if (MODE == 3) { restore_m0(saved3); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = keep[0] + keep[1] + keep[2] + keep[3];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    if (s == 12345.678f) sink[0] = s + ((const float*)ring)[lane];
    if (lane == 0) out_cycles[blockIdx.x * NWAVES + wave] = (uint32_t)(t1 - t0);
}

template <int MODE, int NMFMA, int NWAVES>
static void run(const char* name, const char* src, uint32_t bytes, uint32_t* dcy, float* sink) {
    const int nit = 512, grid = 256;
    std::vector<uint32_t> cy(grid * NWAVES);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((probe<MODE, NMFMA, NWAVES>), dim3(grid), dim3(NWAVES * 64), 0, 0, src, bytes, nit, dcy, sink);
        hipDeviceSynchronize();
    }
    hipMemcpy(cy.data(), dcy, cy.size() * 4, hipMemcpyDeviceToHost);
    std::sort(cy.begin(), cy.end());
    printf("%-58s %d waves/CU, %2d MFMAs/step: median %7.1f cycles per 4-KiB step (p10 %7.1f, p90 %7.1f); MFMA issue alone would be %d\n", name, NWAVES, NMFMA,
           cy[cy.size() / 2] / (double)nit, cy[cy.size() / 10] / (double)nit, cy[cy.size() * 9 / 10] / (double)nit, NMFMA * 32 * (NWAVES / 4));
}

int main() {
    const uint32_t bytes = 64u << 20;
    char* src; uint32_t* dcy; float* sink;
    hipMalloc(&src, bytes); hipMemset(src, 1, bytes); hipMalloc(&dcy, 256 * 8 * 4); hipMalloc(&sink, 64);
#define ALL(NM, NW)                                                                                      \
    run<0, NM, NW>("mode 0: MFMAs only", src, bytes, dcy, sink);                                         \
    run<1, NM, NW>("mode 1: 4 pieces, M0 saved/set/restored per piece", src, bytes, dcy, sink);          \
    run<2, NM, NW>("mode 2: 4 pieces, M0 once + immediate offsets", src, bytes, dcy, sink);              \
    run<3, NM, NW>("mode 3: as 2, pieces spread between the MFMAs", src, bytes, dcy, sink);              \
    run<4, NM, NW>("mode 4: 4 plain global_load_dwordx4 (coalesced)", src, bytes, dcy, sink);
    ALL(0, 4) ALL(12, 4) ALL(12, 8) ALL(6, 8) ALL(24, 4)
    return 0;
}
