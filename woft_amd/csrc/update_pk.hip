// Persistent update-block kernel (woft_update_pk, include/woft_hip.h): the register-streamed conv layers of one refinement
// iteration -- convf2, convc2, convm, z|r and q of both SepConvGRU half steps, the flow head's conv (+ the mask head's first conv
// in the last iteration) -- as ONE launch of 2 x (number of CUs) resident workgroups that pull (layer, tile) work items from a
// device-side queue.  Reference being replaced: update.py:89-97, 45-60, 10-17, 118-125 (BasicUpdateBlock.forward, update.py:127-136).
//
// Why (rounds 3-5): with two workgroups per CU the conv main loops keep the matrix pipe ~90 % busy, but a launch
// pays ~20-25 us besides -- the write-back of its dirty output at the kernel boundary, the ramp, first loads at HBM latency, the
// epilogue and the skew between its first and last workgroup -- and seven such launches were ~40 % of a 0.54 ms iteration; all
// workgroups of a launch also reach their prologues and epilogues together, so nobody's MFMAs cover them.  Here
//   * a work item = one output tile, computed by regb_tile (conv_regb_body.h) -- the very code of conv_regb_kernel: same products,
//     same order, bit-identical results;
//   * items are handed out in dependency order (layer-major) from ONE atomic counter; the next item is requested while the
//     current one computes (the dequeue latency is never exposed);
//   * a tile of layer k + 1 may start as soon as the tiles of layer k under its input halo are complete: per-pixel-tile
//     completion counters, incremented once per column tile.  Producer: write-through (sc1) output stores (WOFT_STORE_WT) ->
//     every wave drains its stores -> workgroup barrier -> one relaxed agent-scope counter increment.  Consumer: one wave polls
//     its (at most 36) counters relaxed, with s_sleep, then ONE agent-scope acquire, then a workgroup barrier -- the
//     placement-independent hand-off of the CDNA guide (Guideline 16, form R1): nothing depends on which XCD runs what;
//   * the queue order is a topological order, so a workgroup only ever waits for items that were pulled before its own, by
//     workgroups that are running: no deadlock whatever the residency; every spin is bounded anyway (give-up code in state[2]);
//   * the last workgroup to leave zeroes the queue state again: no memset launch between iterations, hipGraph-replayable.
#define WOFT_STORE_WT 1
#include "conv_regb_body.h"

// (Instantiated for precision 1, bf16x3 -- the shipped arithmetic -- only: the kernel measured SLOWER than the per-layer launches
//   and stays opt-in as the instrument of that measurement; one translation unit, not four.)

namespace {

constexpr int MAIN_TERMS = 3;

constexpr int ST_HEAD = 0, ST_EXITED = 1, ST_ERR = 2, ST_CNT0 = 16;     // words of the state buffer
constexpr unsigned SPIN_LIMIT = 400000u;                                  // polls (~1 us each) before a workgroup gives up

struct PkHdr {
    int32_t item0[WOFT_PK_MAX_LAYERS + 1];       // first item of every layer; unused entries = n_items
};

constexpr int cmax(int a, int b) { return a > b ? a : b; }
template <int T>
constexpr int smem_of() {
    int m = 0;
    m = cmax(m, RegbGeom<8, 16, 3, 3, 2, T>::SMEM_ELEMS); m = cmax(m, RegbGeom<8, 16, 3, 3, 1, T>::SMEM_ELEMS);
    m = cmax(m, RegbGeom<8, 16, 1, 5, 1, T>::SMEM_ELEMS); m = cmax(m, RegbGeom<4, 16, 1, 5, 1, T>::SMEM_ELEMS);
    m = cmax(m, RegbGeom<8, 16, 5, 1, 1, T>::SMEM_ELEMS); m = cmax(m, RegbGeom<4, 16, 5, 1, 1, T>::SMEM_ELEMS);
    return m;
}
constexpr int PK_SMEM = smem_of<MAIN_TERMS>();

typedef __attribute__((address_space(1))) uint32_t gu32;
__device__ __forceinline__ uint32_t ld_relaxed(const uint32_t* p) {
    return __hip_atomic_load((const gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256, 2) void update_pk_kernel(const woft_pk_layer* __restrict__ tab, const PkHdr hdr, const int n_layers,
                                                           const int n_items, uint32_t* state, const int n_cnt, const int options) {
    __shared__ __attribute__((aligned(16))) __bf16 smem[PK_SMEM + 16];     // (+ two control words behind the tiles: ONE __shared__ object)
    int* s_ctl = (int*)(smem + PK_SMEM);                                   // [0] the workgroup's current item, [1] give-up flag
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the layer table through the CONSTANT address space: its fields are scalar loads (s_load) like kernel arguments
    typedef const __attribute__((address_space(4))) woft_pk_layer CLayer;
    const CLayer* ctab = (const CLayer*)(uintptr_t)tab;

    // developer timeline (tools/pk_timeline.py): state[4..5] = device address of [n_items][4] 64-bit words, or 0 -- per item the
    // constant-rate clock (s_memrealtime, 100 MHz) at its start, after the dependency wait, after the tile, and the XCC id
    unsigned long long* tl = nullptr;
    {
        const unsigned long long a = ((unsigned long long)state[5] << 32) | state[4];
        tl = (unsigned long long*)(uintptr_t)a;
    }

    // options bit 2 (experiment): every second workgroup starts ~20 us late, so that the two workgroups of a CU are half a tile out
    // of phase (one's prologue / epilogue stores beside the other's main loop) instead of in lock-step from the launch on
    if ((options & 4) && (blockIdx.x & 8)) {           // (consecutive ids sit on consecutive XCDs: bit 3 alternates within a CU pair)
        for (int i = 0; i < 4; ++i) __builtin_amdgcn_s_sleep(127);
    }

    int my_next = 0;                                   // (thread 0) the item after the current one, requested a whole item ahead
    if (tid == 0) {
        my_next = (int)__hip_atomic_fetch_add((gu32*)(state + ST_HEAD), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_ctl[0] = my_next;
        s_ctl[1] = 0;
    }
    __syncthreads();
    for (;;) {
        const int item = __builtin_amdgcn_readfirstlane(s_ctl[0]);
        if (item >= n_items) break;
        if (tid == 0)
            my_next = (int)__hip_atomic_fetch_add((gu32*)(state + ST_HEAD), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int l = 0;
#pragma unroll
        for (int k = 1; k < WOFT_PK_MAX_LAYERS; ++k) l += (item >= hdr.item0[k]) ? 1 : 0;
        const CLayer& L = ctab[l];
        const int idx = item - L.item0;
        const int n_nt = L.n_nt;
        const int m_tile = idx / n_nt, n_tile = idx - m_tile * n_nt;
        if (tl != nullptr && tid == 0) {
            tl[4 * item] = __builtin_amdgcn_s_memrealtime();
            tl[4 * item + 3] = (unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));   // HW_REG_XCC_ID, bits 0-3
        }

        // ---- wait for the producers' tiles under this tile's input halo ----
        if (L.n_dep > 0) {
            if (wave == 0) {
                const int ty = L.ty, n_tx = L.n_tx;
                const int m_ty = m_tile / n_tx, m_tx = m_tile - m_ty * n_tx;
                const int y0 = m_ty * ty, x0 = m_tx * 16;
                const int ho = L.conv.ho, wo = L.conv.wo;
                const uint32_t* addr = state;
                uint32_t expect = 0;
                bool need = false;
                int base = 0;
#pragma unroll
                for (int d = 0; d < WOFT_PK_MAX_DEPS; ++d) {
                    if (d < L.n_dep) {
                        const CLayer& P = ctab[L.dep[d]];
                        const int hy = L.dep_hy[d], hx = L.dep_hx[d];
                        const int ya = (y0 - hy > 0 ? y0 - hy : 0) / P.ty, yb = (y0 + ty - 1 + hy < ho - 1 ? y0 + ty - 1 + hy : ho - 1) / P.ty;
                        const int xa = (x0 - hx > 0 ? x0 - hx : 0) >> 4, xb = (x0 + 15 + hx < wo - 1 ? x0 + 15 + hx : wo - 1) >> 4;
                        const int ncx = xb - xa + 1, cnt = (yb - ya + 1) * ncx;
                        const int k = lane - base;
                        if (k >= 0 && k < cnt) {
                            const int r = k / ncx, c = k - r * ncx;
                            addr = state + ST_CNT0 + P.cnt_off + (ya + r) * P.n_tx + (xa + c);
                            expect = (uint32_t)P.n_nt;
                            need = true;
                        }
                        base += cnt;
                    }
                }
                unsigned spins = 0;
                int fail = 0;
                for (;;) {
                    const uint32_t v = need ? ld_relaxed(addr) : expect;
                    if (__all(v >= expect)) break;
                    if (++spins > SPIN_LIMIT || ld_relaxed(state + ST_ERR) != 0u) { fail = 1; break; }
                    __builtin_amdgcn_s_sleep(8);
                }
                if (fail) {
                    if (lane == 0) {
                        __hip_atomic_store((gu32*)(state + ST_ERR), (uint32_t)(item + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        s_ctl[1] = 1;
                    }
                } else if (!(options & 1)) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // ONE invalidate after the match; plain loads from here on
                }
            }
            __syncthreads();
            if (s_ctl[1] != 0) break;
        }

        if (tl != nullptr && tid == 0) tl[4 * item + 1] = __builtin_amdgcn_s_memrealtime();
        // ---- the tile ----
        const auto& p = L.conv;
        switch (L.kind) {
#define PK_CASE(K, TY, KY, KX, WM, T, NB, D) \
    case K: regb_tile<TY, 16, KY, KX, WM, T, NB, D, 2>(p, m_tile, n_tile, smem, nullptr); break
            PK_CASE(0, 8, 3, 3, 2, MAIN_TERMS, 3, 2);
            PK_CASE(1, 8, 3, 3, 1, MAIN_TERMS, 3, 2);
            PK_CASE(4, 8, 1, 5, 1, MAIN_TERMS, 5, 3);
            PK_CASE(5, 4, 1, 5, 1, MAIN_TERMS, 5, 3);
            PK_CASE(7, 8, 5, 1, 1, MAIN_TERMS, 5, 3);
            PK_CASE(8, 4, 5, 1, 1, MAIN_TERMS, 5, 3);
#undef PK_CASE
            default: break;
        }

        // ---- publish: every wave's write-through stores have left, then ONE counter increment ----
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                               // (also: the staging area in smem is free for the next tile's halo)
        if (tid == 0) {
            __hip_atomic_fetch_add((gu32*)(state + ST_CNT0 + L.cnt_off + m_tile), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_ctl[0] = my_next;
            if (tl != nullptr) tl[4 * item + 2] = __builtin_amdgcn_s_memrealtime();
        }
        __syncthreads();
    }

    // ---- leave; the last workgroup out restores the zeroed state for the next launch ----
    __syncthreads();
    if (tid == 0) {
        const unsigned e = __hip_atomic_fetch_add((gu32*)(state + ST_EXITED), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_ctl[1] = (e == gridDim.x - 1) ? 2 : 0;
    }
    __syncthreads();
    if (s_ctl[1] == 2) {                               // (every other workgroup has done its last access to the state)
        for (int i = tid; i < n_cnt; i += 256) state[ST_CNT0 + i] = 0u;
        if (tid == 0) { state[ST_HEAD] = 0u; state[ST_EXITED] = 0u; }
    }
}

int kind_of(const woft_conv_params& c) {
    int taps;
    if (c.taps_y == 3 && c.taps_x == 3) taps = 0;
    else if (c.taps_y == 1 && c.taps_x == 5) taps = 1;
    else if (c.taps_y == 5 && c.taps_x == 1) taps = 2;
    else return -1;
    int shape;
    if (c.halo == 8 && c.tile_n == 64) shape = 0;
    else if (c.halo == 8 && c.tile_n == 128) shape = 1;
    else if (c.halo == 12 && c.tile_n == 128) shape = 2;
    else return -1;
    // the tile forms the layer chooser (woft_amd/ops.py) gives the update block: 3x3 on 8x16 x 64 | 128, 1x5 / 5x1 on 8x16 x 128 and
    // 4x16 x 128 -- the other three combinations (3x3 on 4x16 tiles, 1x5 / 5x1 on 64-column tiles) are not instantiated (each
    // instance is a minute of this unit's compile time, the longest of the build)
    const int kind = taps * 3 + shape;
    return (kind == 2 || kind == 3 || kind == 6) ? -1 : kind;
}

}  // namespace

static int pk_launch(const woft_pk_layer* table_dev, const woft_pk_layer* th, int32_t n, uint32_t* state, int32_t options, int n_cu,
                     void* stream) {
    PkHdr hdr;
    int n_items = 0, n_cnt = 0;
    for (int l = 0; l < n; ++l) {
        hdr.item0[l] = th[l].item0;
        n_items = th[l].item0 + th[l].n_ty * th[l].n_tx * th[l].n_nt;
        n_cnt = th[l].cnt_off + th[l].n_ty * th[l].n_tx;
    }
    for (int l = n; l <= WOFT_PK_MAX_LAYERS; ++l) hdr.item0[l] = n_items;
    int grid = 2 * n_cu;
    if (grid > n_items) grid = n_items;
    hipLaunchKernelGGL(update_pk_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, table_dev, hdr, (int)n, n_items, state,
                       n_cnt, (int)options);
    return woft_launch_status();
}

// every layer in precision 1 (bf16x3)
static bool pk_family_ok(const woft_pk_layer* t, int32_t n) {
    for (int l = 0; l < n; ++l)
        if (t[l].conv.precision != 1) return false;
    return true;
}

extern "C" int woft_update_pk_prepare(woft_pk_layer* t, int32_t n) {
    if (t == nullptr || n < 1 || n > WOFT_PK_MAX_LAYERS) return WOFT_EINVAL;
    if (!pk_family_ok(t, n)) return WOFT_EINVAL;
    int item = 0, cnt = 0;
    for (int l = 0; l < n; ++l) {
        woft_pk_layer& L = t[l];
        const woft_conv_params& c = L.conv;
        if (c.in0 == nullptr || c.out == nullptr || c.wgt_frag == nullptr || c.n_img != 1 || c.stride != 1 || c.flat != 0) return WOFT_EINVAL;
        if (c.in_norm != 0 || c.stat_sum != nullptr || c.in_mean != nullptr || c.wh0_lookup != nullptr || c.out_pitch != 0) return WOFT_EINVAL;
        if (c.ho != c.h || c.wo != c.w || c.pad_y != c.taps_y / 2 || c.pad_x != c.taps_x / 2) return WOFT_EINVAL;
        if (c.cin_pad <= 0 || c.cin_pad % 32 != 0 || (c.in1 != nullptr && (c.c_split <= 0 || c.c_split % 32 != 0 || c.c_split >= c.cin_pad))) return WOFT_EINVAL;
        const int kind = kind_of(c);
        if (kind < 0 || c.cout_pad % c.tile_n != 0 || c.cout > c.cout_pad) return WOFT_EINVAL;
        switch (c.epi) {
            case WOFT_EPI_LINEAR: case WOFT_EPI_RELU: case WOFT_EPI_SIGMOID: case WOFT_EPI_TANH: break;
            case WOFT_EPI_GRU_ZR: if (c.e0 == nullptr || c.out1 == nullptr || c.cout % 4 != 0) return WOFT_EINVAL; break;
            case WOFT_EPI_GRU_Q: if (c.e0 == nullptr || c.e1 == nullptr || c.cout % 4 != 0) return WOFT_EINVAL; break;
            case WOFT_EPI_FLOWHEAD:
                if (c.halo != 8 || c.e0 == nullptr || c.ldo < 20 || c.ldo % 4 != 0 || c.co_off != 0 || c.cout % 32 != 0 || c.bias_map != nullptr) return WOFT_EINVAL;
                break;
            default: return WOFT_EINVAL;
        }
        if (c.ldo % 4 != 0 || c.co_off % 4 != 0) return WOFT_EINVAL;
        // write-through stores address their tensors through 32-bit byte offsets
        const int64_t px = (int64_t)c.ho * c.wo;
        if (px * c.ldo * 4 * (c.epi == WOFT_EPI_FLOWHEAD ? c.cout_pad / c.tile_n : 1) >= (1ll << 31) || px * (c.ldo1 > 0 ? c.ldo1 : 1) * 4 >= (1ll << 31))
            return WOFT_EINVAL;
        const int64_t cs_max = (c.in1 != nullptr && c.cs1 > c.cs0) ? c.cs1 : c.cs0;
        if (px * cs_max >= (1ll << 31)) return WOFT_EINVAL;                         // (32-bit element offsets of the halo loader)
        if (L.n_dep < 0 || L.n_dep > WOFT_PK_MAX_DEPS) return WOFT_EINVAL;
        L.kind = kind;
        L.ty = c.halo == 12 ? 4 : 8;
        L.n_ty = (c.ho + L.ty - 1) / L.ty;
        L.n_tx = (c.wo + 15) / 16;
        L.n_nt = c.cout_pad / c.tile_n;
        L.item0 = item;
        L.cnt_off = cnt;
        item += L.n_ty * L.n_tx * L.n_nt;
        cnt += L.n_ty * L.n_tx;
        int lanes = 0;
        for (int d = 0; d < L.n_dep; ++d) {
            if (L.dep[d] < 0 || L.dep[d] >= l || L.dep_hy[d] < 0 || L.dep_hx[d] < 0 || L.dep_hy[d] > 8 || L.dep_hx[d] > 16) return WOFT_EINVAL;
            const woft_pk_layer& P = t[L.dep[d]];
            if (P.conv.ho != c.ho || P.conv.wo != c.wo) return WOFT_EINVAL;
            lanes += ((L.ty + 2 * L.dep_hy[d] + P.ty - 2) / P.ty + 1) * ((16 + 2 * L.dep_hx[d] + 14) / 16 + 1);
        }
        if (lanes > 64) return WOFT_EINVAL;            // (one polling lane per producer tile)
    }
    return WOFT_OK;
}

extern "C" int64_t woft_update_pk_state_bytes(const woft_pk_layer* t, int32_t n) {
    if (t == nullptr || n < 1 || n > WOFT_PK_MAX_LAYERS) return -1;
    return 4ll * (ST_CNT0 + t[n - 1].cnt_off + t[n - 1].n_ty * t[n - 1].n_tx);
}

extern "C" int woft_update_pk(const woft_pk_layer* table_dev, const woft_pk_layer* th, int32_t n, uint32_t* state, int32_t options,
                              void* stream) {
    if (table_dev == nullptr || th == nullptr || state == nullptr || n < 1 || n > WOFT_PK_MAX_LAYERS) return WOFT_EINVAL;
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return WOFT_ELAUNCH;
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    if (!pk_family_ok(th, n)) return WOFT_EINVAL;
    return pk_launch(table_dev, th, n, state, options, n_cu, stream);
}
