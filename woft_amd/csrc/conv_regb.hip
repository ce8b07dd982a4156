// conv_regb_kernel: one output tile of the register-streamed convolution (conv_regb_body.h) per workgroup; one layer per launch,
// or two independent layers that select the same instance (woft_conv2d_pair).
#include "conv_regb_body.h"

extern int g_regb_dyn_lds;          // conv.hip (woft_set_tuning key 3)

// Build parts (woft_amd/build.py): this file is compiled once per precision code -- WOFT_ONLY_PREC in {1, 2, 3, 4} -- each part
// exporting woft_conv_regb_launch_p<prec>; conv.hip's dispatcher of the same precision calls it.
#if !defined(WOFT_ONLY_PREC)
#error "conv_regb.hip is compiled in parts: -DWOFT_ONLY_PREC=1|2|3|4 (woft_amd/build.py)"
#endif
#define WOFT_CAT2_(a, b) a##b
#define WOFT_CAT2(a, b) WOFT_CAT2_(a, b)
#define WOFT_REGB_ENTRY WOFT_CAT2(woft_conv_regb_launch_p, WOFT_ONLY_PREC)

namespace {

template <int TY, int TX, int KY, int KX, int WM, int TERMS, int NBUF, int DIST, int AD>
__global__ __launch_bounds__(256, 2) void conv_regb_kernel(const woft_conv_params pa, const woft_conv_params pb, const int split) {
    // (two independent layers that run on the same instance of this kernel may share ONE launch -- woft_conv2d_pair: the
    //  workgroups [0, split) belong to the first layer, the rest to the second; split = gridDim.x for a single layer)
    const bool second_layer = (int)blockIdx.x >= split;
    const woft_conv_params p = second_layer ? pb : pa;     // (a copy: a REFERENCE selected between the two argument
                                                            //  structs does not compile -- 'illegal VGPR to SGPR copy')
    const int bid = second_layer ? (int)blockIdx.x - split : (int)blockIdx.x;
    using G = RegbGeom<TY, TX, KY, KX, WM, TERMS>;
    __shared__ __attribute__((aligned(16))) __bf16 smem[G::SMEM_ELEMS];
    const int tyn = (p.ho + TY - 1) / TY, txn = (p.wo + TX - 1) / TX;
    int m_tile, n_tile;
    woft::tile_of_block(bid, p.n_img * tyn * txn, p.cout_pad / G::BN, m_tile, n_tile);
    // developer probe (tools/regb_probe.py): s_memtime stamps of wave 0 -> in_rstd (unused by this kernel otherwise)
    unsigned long long* stamps = (p.in_mean == (const float*)1 && threadIdx.x == 0)
                                     ? (unsigned long long*)p.in_rstd + (size_t)bid * 32 : nullptr;
    regb_tile<TY, TX, KY, KX, WM, TERMS, NBUF, DIST, AD>(p, m_tile, n_tile, smem, stamps);
}

template <int WM, int TY = 8>
int launch_regb(const woft_conv_params& p, const woft_conv_params* second, hipStream_t s) {
    constexpr int TX = 16, BN = 128 / WM;
    auto blocks = [](const woft_conv_params& q) {
        const int tyn = (q.ho + TY - 1) / TY, txn = (q.wo + TX - 1) / TX;
        return (int64_t)q.n_img * tyn * txn * (q.cout_pad / BN);
    };
    const woft_conv_params& pb = second ? *second : p;
    const int split = (int)blocks(p);
    dim3 grid((unsigned)(blocks(p) + (second ? blocks(pb) : 0)));
#define REGB(KY, KX, T, NB, D) \
    woft_launch(0, conv_regb_kernel<TY, TX, KY, KX, WM, T, NB, D, 2>, grid, dim3(256), (size_t)g_regb_dyn_lds, s, p, pb, split)
#define REGB_TAPS(T)                                                     \
    if (p.taps_y == 3 && p.taps_x == 3) REGB(3, 3, T, 3, 2);             \
    else if (p.taps_y == 1 && p.taps_x == 5) REGB(1, 5, T, 5, 3);        \
    else if (p.taps_y == 5 && p.taps_x == 1) REGB(5, 1, T, 5, 3);        \
    else return WOFT_EINVAL
    if (p.precision != WOFT_ONLY_PREC) return WOFT_EINVAL;
#if WOFT_ONLY_PREC == 1
    REGB_TAPS(3);
#elif WOFT_ONLY_PREC == 3
    REGB_TAPS(16);
#elif WOFT_ONLY_PREC == 4
    if (p.wgt_mx == nullptr || pb.wgt_mx == nullptr) return WOFT_EINVAL;
    if (p.taps_y == 3 && p.taps_x == 3) REGB(3, 3, 28, 3, 2);
    else if (p.taps_y == 1 && p.taps_x == 5) REGB(1, 5, 28, 5, 3);
    else if (p.taps_y == 5 && p.taps_x == 1) REGB(5, 1, 28, 5, 3);
    else return WOFT_EINVAL;
#else
    REGB_TAPS(1);
#endif
#undef REGB_TAPS
#undef REGB
    return woft_launch_status();
}

}  // namespace

// Called by woft_conv2d for p.halo == 8 (8 x 16-pixel tiles) and 12 (4 x 16 pixels x 128 columns: four waves = four column
// bands of 64 rows -- the per-wave work of the 8 x 16 x 64 layout with the weights fetched once per workgroup), after its
// argument checks.
int WOFT_REGB_ENTRY(const woft_conv_params& p, const woft_conv_params* second, void* stream) {
    if (second != nullptr) {            // one launch for two layers: the same kernel instance, no probe
        const woft_conv_params& b = *second;
        if (b.halo != p.halo || b.tile_n != p.tile_n || b.taps_y != p.taps_y || b.taps_x != p.taps_x || b.precision != p.precision ||
            b.wgt_frag == nullptr || b.in_norm != 0 || b.in_mean != nullptr || p.in_mean != nullptr || b.wh0_lookup != nullptr ||
            b.epi == WOFT_EPI_WH_MEAN || b.cout_pad % b.tile_n != 0)
            return WOFT_EINVAL;
        if (b.epi == WOFT_EPI_FLOWHEAD && (b.e0 == nullptr || b.ldo < 20 || b.ldo % 4 != 0 || b.co_off != 0 || b.cout % 32 != 0)) return WOFT_EINVAL;
    }
    if (p.halo == 12) {
        if (p.wgt_frag == nullptr || p.in_norm != 0 || (p.in_mean != nullptr && p.in_mean != (const float*)1) || p.wh0_lookup != nullptr ||
            p.epi == WOFT_EPI_WH_MEAN || p.epi == WOFT_EPI_FLOWHEAD || p.tile_n != 128 || p.cout_pad % 128 != 0) return WOFT_EINVAL;
        return launch_regb<1, 4>(p, second, (hipStream_t)stream);
    }
    if (p.wgt_frag == nullptr || p.wh0_lookup != nullptr || p.epi == WOFT_EPI_WH_MEAN) return WOFT_EINVAL;
    if (p.in_norm != 0 || p.stat_sum != nullptr) return WOFT_EINVAL;    // (InstanceNorm plumbing: the LDS-halo kernel, conv.hip)
    if (p.in_mean != nullptr && p.in_mean != (const float*)1) return WOFT_EINVAL;
    if (p.epi == WOFT_EPI_FLOWHEAD && (p.e0 == nullptr || p.ldo < 20 || p.ldo % 4 != 0 || p.co_off != 0 || p.cout % 32 != 0)) return WOFT_EINVAL;
    if (p.tile_n == 128 && p.cout_pad % 128 == 0) return launch_regb<1>(p, second, (hipStream_t)stream);
    if (p.tile_n == 64 && p.cout_pad % 64 == 0) return launch_regb<2>(p, second, (hipStream_t)stream);
    return WOFT_EINVAL;
}
