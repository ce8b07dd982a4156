// LDS-DMA (global_load_lds) helpers shared by the MFMA kernels.
#pragma once
#include "common.h"

namespace {

// One LDS-DMA wave instruction: lane L copies 16 bytes from (gbase + lane_off) to LDS byte address lds_addr + 16 L.
// gbase and lds_addr are wave-uniform (SGPRs).  Written as inline assembly so that (1) the address is the
// scalar-base + 32-bit-lane-offset form and (2) the compiler does not count it: it would otherwise drain EVERY
// outstanding global load (vmcnt(0)) at the next barrier, including prefetches that are meant to stay in flight.
// The caller orders it explicitly with dma_wait<N>() before the barrier that publishes the data.
__device__ __forceinline__ void lds_dma16(const void* gbase, uint32_t lane_off, uint32_t lds_addr) {
    // (the base is wave-uniform by construction; saying so explicitly keeps the "s" operand an SGPR pair even when the
    //  register allocator is short of SGPRs -- it otherwise handed the asm a VGPR pair: "invalid operand")
    const uint64_t gb = (uint64_t)(uintptr_t)gbase;
    const uint32_t g_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(gb >> 32));
    const uint32_t g_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)gb);      // (the builtin returns int:
    const uint64_t gu = ((uint64_t)g_hi << 32) | (uint64_t)g_lo;                             //  no sign extension here)
    uint32_t saved_m0;                                   // m0 is a reserved register: hand it back as found
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(saved_m0)
                 : "s"(lds_addr), "v"(lane_off), "s"(gu)
                 : "memory", "vcc");                     // (vcc: keeps the 64-bit base out of a register pair the
                                                         //  instruction's saddr field cannot encode)
}
template <int N>
__device__ __forceinline__ void dma_wait() {             // at most N vector-memory loads still in flight
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

}  // namespace
