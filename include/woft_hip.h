/* woft_hip.h -- C ABI of libwoft_hip.so: the MI355X (gfx950) kernels of the WOFT hot path.
 *
 * Conventions:
 *   - every entry point is extern "C", takes plain device pointers / sizes, enqueues on the
 *     hipStream_t passed as `stream` (void*), never allocates, never synchronises;
 *   - the caller owns every buffer (PyTorch caching allocator in the Python host);
 *   - return 0 on success, WOFT_EINVAL (-1) for a rejected argument, WOFT_ELAUNCH (-2) when
 *     hipGetLastError() reports a launch failure.  No C++ exceptions cross the ABI.
 *   - activations are NHWC fp32 with an explicit channel stride ("cs", floats per pixel).
 *
 * Each entry point cites the reference code it replaces (paths relative to
 * /root/reference/pytracking/).
 */
#ifndef WOFT_HIP_H
#define WOFT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WOFT_OK 0
#define WOFT_EINVAL (-1)
#define WOFT_ELAUNCH (-2)

/* ABI / build identification: returns 10000*major + 100*minor + patch. */
int woft_abi_version(void);
/* sizeof(woft_conv_params) (which = 0) / woft_lookup_params (1) / woft_lookup_otf_params (2): layout check for FFI mirrors. */
int woft_sizeof(int which);
/* developer knob for the micro-benchmarks in tools/ (ablation bits of the correlation GEMM, key 2); 0 in production. */
int woft_set_tuning(int key, int value);

/* ---------------------------------------------------------------------------------------
 * Implicit-GEMM convolution on f32-input MFMA (v_mfma_f32_32x32x2_f32), NHWC.
 * Replaces every nn.Conv2d on the path: external/RAFT/raft_core/extractor.py:10-45,127-147,
 * update.py:9-10,19-21,36-42,82-86,118-125, weighted_raft.py:336-341 -- and, used as an
 * NT GEMM, the all-pairs correlation torch.matmul of corr.py:62-69.
 * ------------------------------------------------------------------------------------- */
enum woft_epilogue {
    WOFT_EPI_LINEAR = 0,        /* y = alpha*acc + bias                                      */
    WOFT_EPI_RELU = 1,          /* relu(y)                                                   */
    WOFT_EPI_SIGMOID = 2,
    WOFT_EPI_TANH = 3,
    WOFT_EPI_RELU_RES_RELU = 4, /* relu(e0[m][n] + relu(y))   residual block, extractor.py:48-56 */
    WOFT_EPI_GRU_ZR = 5,        /* n <  split: out[m][n]   = sigmoid(y)            (z)
                                   n >= split: out1[m][n-split] = sigmoid(y) * e0[m][n-split] (r*h)
                                   update.py:47-50 */
    WOFT_EPI_GRU_Q = 6,         /* out = (1-z)*h + z*tanh(y), h = e0, z = e1   update.py:50-51 */
    WOFT_EPI_CTX = 7,           /* n < split: tanh(y) else relu(y)   weighted_raft.py:217-219   */
    WOFT_EPI_WH_MEAN = 8,       /* halo 2 (9x9 patches), one N tile: out[image] = e1[0] + mean over the 81 pixels of
                                   <e0[0..cout), relu(y)> -- the weight head's last ReLU, 1x1 conv and patch mean
                                   (weighted_raft.py:341,378-383) without writing the activation               */
    WOFT_EPI_FLOWHEAD = 9       /* halo 8 only.  FlowHead (update.py:10-17) conv2(relu(conv1(x))) with conv2 (3x3 -> 2
                                   channels) folded into conv1's epilogue: instead of relu(y) the launch writes, per
                                   pixel m, the 18 partial products s[tap * 2 + o] = <W2[o][:, tap], relu(y[m])> over
                                   the channels of column tile t to out[(t * M + m) * ldo + 0..17] (ldo >= 20; t <
                                   cout_pad / tile_n planes; M = n_img * ho * wo).  e0 = W2 as bf16 MFMA B fragments,
                                   [cout / 32 bands][2 k halves][planes hi(, lo)][64 lanes][8], lane = 32 * (k half
                                   of the half) + column j, columns j >= 18 zero.  woft_flow_head_gather finishes
                                   the 3x3 sum.                                                                */
};

typedef struct woft_conv_params {
    const float* in0;      /* source of input channels [0, c_split)                            */
    const float* in1;      /* source of input channels [c_split, cin_pad) or NULL              */
    int32_t cs0, cs1;      /* floats per pixel of in0 / in1                                    */
    int32_t c_split;       /* multiple of 32; == cin_pad when in1 is NULL                      */
    int32_t n_img, h, w;   /* input batch and spatial size                                     */
    int32_t ho, wo;        /* output spatial size                                              */
    int32_t taps_y, taps_x;/* kernel taps (flat mode: taps_x must be 1)                        */
    int32_t stride, pad_y, pad_x;
    int32_t cin_pad;       /* GEMM-K per tap, multiple of 32                                   */
    int32_t flat;          /* 1: the 32-float K chunk of a tap runs along x over 32/cs0 pixels */
    const float* wgt;      /* [cout_pad][taps_y*taps_x*cin_pad] fp32, K contiguous (precision 0) */
    const void* wgt_hi;    /* same shape, bf16: hi = bf16(w)          (precision 1, 2); fp16(w) (precision 3) */
    const void* wgt_lo;    /* same shape, bf16: lo = bf16(w - hi)     (precision 1)            */
    int32_t precision;     /* (4: see wgt_mx)  0: fp32 MFMA; 1: split-bf16 x3 (fp32-emulating); 2: bf16; 3: fp16 operands, fp32
                              accumulation (the reference's mixed_precision autocast, weighted_raft.py:204,215,233;
                              not for the 9 x 9 weight-head windows, which the reference keeps in fp32)   */
    const float* bias;     /* [cout_pad] or NULL                                               */
    float alpha;           /* scale applied to the accumulator                                 */
    int32_t cout;          /* valid output channels                                            */
    int32_t cout_pad;      /* rows of wgt, multiple of the N tile (64 or 128)                  */
    float* out;            /* out[m*ldo + co_off + col(n)]                                     */
    int64_t ldo;
    int32_t co_off;
    int32_t out_w, out_pitch; /* if out_pitch != 0: col(n) = (n / out_w)*out_pitch + n % out_w  */
    int32_t epi;           /* enum woft_epilogue                                               */
    int32_t split;         /* channel split of GRU_ZR / CTX                                    */
    const float* e0;       /* epilogue operand (residual / h), row stride lde0                 */
    const float* e1;       /* epilogue operand (z), row stride lde1                            */
    int32_t lde0, lde1;
    float* out1;           /* second output of GRU_ZR, row stride ldo1                         */
    int32_t ldo1;
    float* stat_sum;       /* optional [2*ceil(M/BM)][cout_pad] per-wave-row partial sums of y */
    float* stat_sq;        /*          ... and of y*y  (InstanceNorm statistics)               */
    int32_t tile_m, tile_n;/* block tile: 128 or 64 each (halo 16: tile_n 256; flat: 128)      */
    int32_t halo;          /* 0: gather A per tap.  Split-bf16 precisions, stride 1, 3x3/1x5/5x1 only:
                              input halo of the output tile resident in LDS for all taps; tile_m ignored:
                              1 = 8x16 px, 4 = 4x16 px, 2 = one 9x9 image per workgroup (weight-head patches;
                              ho = wo = 9).  (Larger tiles / several patches per workgroup were measured
                              1.5-3x slower: one workgroup per CU cannot hide its own latencies.)
                              7 = the encoders' first layer (extractor.py:127-129: 7x7, stride 2, pad 3) on its own kernel:
                              flat packing of an NHWC4 image (cs0 = 4, taps_y = 7, taps_x = 1, cin_pad = 32), split-bf16
                              precisions, cout_pad % 64 == 0, 8x16 output pixels x 64 channels per tile, statistics rows
                              indexed by those tiles; results bit-identical to halo = 0
                              16 = wide 1x1 / stride-1 layers (update.py:89 convc1, extractor.py:166 conv2) on the streamed
                              GEMM kernel (conv_1x1.hip): 64 pixels x tile_n = 256 columns per workgroup, the activations
                              read and converted once per layer, weights from wgt_frag (taps = 1); also stride-1 flat
                              layers (flat != 0, cin_pad = 32, taps_x = 1, pad_y = taps_y / 2; update.py:91 convf1) with
                              tile_n = 128, K chunks = tap rows; split-bf16 precisions, cout_pad % tile_n == 0, no
                              statistics / in_norm; bit-identical to halo = 0.  woft_conv2d_pair takes two such layers */
    int32_t in_norm;       /* halo != 0 only: 0 = use in0 as is; 1 / 2 = in0 holds a RAW conv output whose
                              InstanceNorm is applied while loading (extractor.py:44-47: x = (x - in_mean[c]) *
                              in_rstd[c], 2: followed by ReLU), zero padding applied after it; in1 must be NULL */
    const float* in_mean;  /* [cin] per-channel statistics of in0 (woft_inorm_finalize)          */
    const float* in_rstd;
    const float* bias_map; /* optional per-pixel bias [M][ld_bias_map] used instead of bias[] (cout % 4 == 0): the
                              contribution of input channels that do not change between launches, computed once
                              (the GRU's context features `inp`, update.py:45-60: conv(W,[h,inp,motion]) =
                              conv(W_h,m, [h,motion]) + conv(W_inp, inp))                              */
    int32_t ld_bias_map;
    const int32_t* out_index; /* WOFT_EPI_WH_MEAN only, optional: image i writes out[out_index[i]] (the weight head
                              evaluated on a subset of the source pixels)                        */
    /* Weight head only (halo == 2, 9 x 9 windows, 3 x 3 taps, cin_pad == 128, split-bf16 precisions): when
       wh0_lookup != NULL the head's FIRST conv (5 -> 128 channels, 3 x 3, zero padding, ReLU;
       weighted_raft.py:336,363-376) is evaluated inside this launch, 32 channels at a time, from the lookup
       window of the source pixel -- in0 is ignored, the 1.3 GB activation between the two layers never exists.
       Input channels of window position t: lookup[s][4 t .. 4 t + 3] and mean[s] (woft_wh_conv0's layout). */
    const float* wh0_lookup;   /* [P][wh0_ld] fp32, wh0_ld >= 324, % 4 == 0                               */
    int32_t wh0_ld;
    const float* wh0_mean;     /* [P]                                                                     */
    const void* wh0_w;         /* bf16 MFMA fragments [4 chunks][3 K steps][1 or 2 planes hi, lo][64 lanes][8]:
                                  lane L, element e = W0[32 chunk + L % 32][k = 16 step + 8 (L / 32) + e],
                                  k = (3 ky + kx) * 5 + ci, zero for k >= 45                                */
    const float* wh0_bias;     /* [128]                                                                   */
    const int32_t* wh0_index;  /* optional: window i of this launch is source pixel wh0_index[i]          */
    /* halo == 8 (8x16-pixel tiles, split-bf16 precisions, stride 1, 3x3 / 1x5 / 5x1, no in_norm): the weight operand is
       streamed global -> registers instead of through LDS; it is read from wgt_frag, the same weights in MFMA-fragment
       order: [cout_pad / 32 bands][cin_pad / 32 chunks][taps][planes: hi (, lo)][2 k halves][64 lanes][8] bf16, lane L
       element e of k half s = W[32 band + L % 32][tap][32 chunk + 8 (2 s + L / 32) + e].  tile_n 128: one wave per
       32-column band and all 128 rows; tile_n 64: 2 x 2 waves.  halo == 12: the same kernel on 4x16-pixel tiles x 128
       columns (tile_n 128, cout_pad % 128 == 0, multi-tap layers): one wave per band and all 64 rows.            */
    const void* wgt_frag;
    /* precision 4 ("f16mx8", round 4; halo 8 / 12 with 3x3 / 1x5 / 5x1 taps only): an fp32-emulating product in two matrix-pipe
       passes -- fp16(a) * fp16(w) on v_mfma_f32_32x32x16_f16 + the two cross terms (a - fp16(a)) * w and a * (w - fp16(w)) on the
       block-scaled fp8 form v_mfma_scale_f32_32x32x64_f8f6f4, one scaled MFMA per PAIR of taps (K = 2 taps x 32 channels; an odd
       last tap pairs with zero weights).  wgt_frag: the fp16 fragments (the precision-3 format).  wgt_mx: per 32-column band, 32-
       channel chunk, tap pair and term (0: w, multiplied with the activations' remainder; 1: w - fp16(w), multiplied with the
       activations): 64 lanes x 32 bytes fp8 e4m3 (lane L: column 32 band + L % 32; bytes 0-15 = channels 16 (L / 32) .. + 15 of
       the chunk at the pair's first tap, bytes 16-31 = at its second tap), then 64 x int32 whose low byte is the E8M0 scale of the
       (column, tap, chunk) block the lane half supplies (L / 32 = 0: first tap, 1: second tap; value = 2^(scale - 127)).
       Measured: 2.2-2.3 x the error of precision 1 at 1.64 x its matrix-pipe rate (tools/micro/mx_split_probe.hip). */
    const void* wgt_mx;
} woft_conv_params;

int woft_conv2d(const woft_conv_params* p, void* stream);
/* Two INDEPENDENT layers in one launch: both must select the same kernel instance -- same precision (split-bf16 only),
 * halo mode (0 = per-tap kernel, any tap shapes; 8 / 12 = register-streamed kernel, equal tap shape; 16 = streamed GEMM
 * kernel: a 1x1 and a flat layer, or two of a kind, each with its own tile_n), tile_m / tile_n; no InstanceNorm
 * statistics.  The first layer's workgroups are dispatched first.  Results are those of two woft_conv2d calls; what is
 * saved is one kernel boundary and the partly empty last round of workgroups of each launch (the motion encoder's
 * correlation and flow branches, update.py:91-95, are independent until `conv` joins them).  WOFT_EINVAL when the layers
 * do not share a kernel: the caller launches them one after the other. */
int woft_conv2d_pair(const woft_conv_params* a, const woft_conv_params* b, void* stream);
/* Host frame -> device (the plugin API hands track() a numpy frame per call: TRK:57-62, WOFT_demo.py:61-78).  `src` is
 * pageable host memory, `pinned` a page-locked staging buffer and `dev` the device buffer, `bytes` each.  The frame is moved in
 * n_chunks pieces: piece k is copied src -> pinned by the calling thread and its asynchronous H2D copy is enqueued on `stream`
 * at once, so the DMA of piece k runs under the host memcpy of piece k + 1 (one memcpy of the whole frame followed by one H2D
 * copy serialises the two: 0.35 ms per 1080p frame; 4 pieces: 0.18 ms; more pieces lose to the ~20 us per hipMemcpyAsync call).
 * Returns after the last piece is ENQUEUED: `pinned` may be rewritten once the stream has passed this point (the caller records an
 * event).  (Measured alternative, what the Python host now does by default: the runtime's own pageable copy, 0.13 ms.) */
int woft_upload_u8(const void* src, void* pinned, void* dev, int64_t bytes, int32_t n_chunks, void* stream);
/* fp32 array (n % 4 == 0) -> bf16 planes hi = bf16(x), lo = bf16(x - hi) (lo may be NULL): the
 * split form of a dynamic B operand (fmap2 in the correlation GEMM). */
int woft_split_bf16(const float* x, int64_t n, void* hi, void* lo, void* stream);
/* 3x3, stride 1, zero padding 1, cout in {1, 2}, cin_pad in {128, 256} (the flow head's second conv,
 * update.py:10-17: hidden -> 2), exact fp32 FMAs on the vector ALUs instead of a 97 % padded matrix-core tile.
 * in: NHWC fp32 with channel stride cs; wgt: packed fp32 rows [cout][9 * cin_pad] (k = tap * cin_pad + c, as
 * woft_conv2d's fp32 weights); out[pixel * ldo + co_off + o] = bias[o] + sum. */
int woft_conv3x3_narrow(const float* in, int32_t cs, int32_t n_img, int32_t h, int32_t w, int32_t cin_pad,
                        const float* wgt, const float* bias, int32_t cout, float* out, int64_t ldo, int32_t co_off,
                        void* stream);
/* The flow head's second conv and the coordinate update in one launch (update.py:10-17, weighted_raft.py:236-237:
 * delta = conv3x3(in) (2 channels, as woft_conv3x3_narrow, also stored to delta[pixel * ld_delta + 0..1]);
 * coords1 += delta; flow = coords1 - grid written as woft_coords_update writes it (flow4, flow_cat optional).
 * One image of h x w pixels; the same fp32 operations as the two separate calls. */
int woft_flow_head_update(const float* in, int32_t cs, int32_t h, int32_t w, int32_t cin_pad, const float* wgt,
                          const float* bias, float* delta, int64_t ld_delta, float* coords1, float* flow4,
                          float* flow_cat, int32_t ld_cat, void* stream);
/* Second half of the flow head when its first conv ran with WOFT_EPI_FLOWHEAD: delta[q][o] = bias2[o] + sum over the
 * 3x3 taps (ky, kx) and the n_planes column-tile planes of part[(plane * h * w + q + (ky-1) * w + (kx-1)) * ld +
 * (ky * 3 + kx) * 2 + o] (ld >= 20, % 4 == 0; pixels outside the image contribute nothing: zero padding, update.py:14; planes
 * first, then taps, in a fixed order);
 * then, exactly as woft_flow_head_update: delta stored, coords1 += delta, flow = coords1 - grid written to flow4 /
 * flow_cat (optional).  One image of h x w pixels. */
int woft_flow_head_gather(const float* part, int32_t n_planes, int32_t ld, int32_t h, int32_t w, const float* bias2,
                          float* delta, int64_t ld_delta, float* coords1, float* flow4, float* flow_cat, int32_t ld_cat,
                          void* stream);
/* x (n floats, n % 32 == 0) -> n/32 lines of 128 bytes, line = [bf16 hi of 32 values | bf16 lo of the same 32],
 * hi = bf16(x), lo = bf16(x - hi): the operand format of woft_corr_gemm_bf16 with terms = 3. */
int woft_split_bf16_lines(const float* x, int64_t n, void* out, void* stream);
/* All-pairs correlation (corr.py:62-69) on pre-split operands: out[i][j] = alpha * <A[i], B[j]>, i < m, j < n.
 * terms = 3 (hi*hi + hi*lo + lo*hi, fp32-emulating): a / b = woft_split_bf16_lines of fmap1 [rows_a][k] /
 * of woft_tile_rows(fmap2) [rows_b][k]; terms = 1: a / b = their plain bf16 planes (woft_split_bf16 hi), k % 64 == 0.
 * rows_a, rows_b: allocated rows, multiples of 128 (rows beyond m / n are read, their products never stored).
 * out: fp32 [m][ldo], or -- out_bf16 != 0 -- bf16 [m][ldo] (fp32 accumulators rounded to nearest even once, at the
 * store): the bf16-storage volume of the plain-bf16 operating point (2096 B per pixel and lookup). */
int woft_corr_gemm_bf16(const void* a, const void* b, int64_t m, int64_t n, int64_t rows_a, int64_t rows_b, int32_t k,
                        float alpha, void* out, int64_t ldo, int32_t terms, int32_t out_bf16, void* stream);

/* InstanceNorm (extractor.py:28-32,129-130; nn.InstanceNorm2d eps=1e-5, biased variance):
 * finalize per-channel statistics from the conv epilogue's partial sums ... */
int woft_inorm_finalize(const float* stat_sum, const float* stat_sq, int32_t n_part, int32_t ld,
                        int32_t channels, int32_t channels_pad, int64_t count, float eps,
                        float* mean, float* rstd, void* ws, void* stream);   /* mean/rstd[channels..channels_pad) := 0 */
/* ws: NULL (one workgroup does it all), or woft_inorm_ws_bytes() bytes of device scratch, ZEROED ONCE by the caller and
 * then owned by the calls on one stream (it holds the cross-workgroup partial sums and their ticket counter, which every
 * call leaves at zero); channels_pad <= 256, ld % 4 == 0. */
int64_t woft_inorm_ws_bytes(void);
/* ... and apply them.  mode 0: (x-mean)*rstd ; 1: relu(.) ; 2: relu(res' + relu(.)), where res' = res (res_mode 0), or
 * -- res being the RAW conv output of the block's shortcut, normalised here with its own statistics -- (res-res_mean)*res_rstd
 * (res_mode 1; the 1x1 downsample branch, extractor.py:40-45) or relu of that (res_mode 2; the block input). */
int woft_inorm_apply(const float* x, const float* mean, const float* rstd, const float* res,
                     const float* res_mean, const float* res_rstd, int32_t res_mode,
                     float* out, int64_t n_pix, int32_t channels, int32_t mode, void* stream);

/* uint8 BGR HWC image -> normalised RGB NHWC4 fp32 (2*x/255-1, 4th channel 0).
 * optical_flow/raft.py:113-120 + weighted_raft.py:194-195.  replicate-pads to (hp, wp) with the
 * top/left offsets (pad_top, pad_left)  (utils/utils.py:7-19 InputPadder). */
int woft_preprocess_bgr_u8(const uint8_t* img, int32_t h, int32_t w, float* out,
                           int32_t hp, int32_t wp, int32_t pad_top, int32_t pad_left, void* stream);

/* 2x2 stride-2 average pool of an NHWC feature map, floor sizes (corr.py:25-27 applied to
 * fmap2 instead of the volume: identical by linearity, as corr.py:77-81 does). */
int woft_avgpool2_nhwc(const float* in, int32_t h, int32_t w, int32_t c, float* out, void* stream);

/* The target feature pyramid of the volume-free correlation (corr.py:25-27,77-81) in ONE launch: from the level-0 map
 * in [h*w][c] fp32, the (levels - 1) pooled maps pooled[l-1] = woft_avgpool2_nhwc applied l times (floor sizes, fp32,
 * bit-identical to the chained calls) and every level's split operand split[l] -- terms = 3: woft_split_bf16_lines of
 * level l, terms = 1: woft_split_bf16's hi plane.  c % 32 == 0, 1 <= levels <= 4. */
int woft_feature_pyramid(const float* in, int32_t h, int32_t w, int32_t c, int32_t levels, float* const* pooled,
                         void* const* split, int32_t terms, void* stream);

/* Correlation lookup, corr.py:29-59 + utils/utils.py:59-73.
 * vol[l]: level l of the pyramid as [P][ht_l][wt_l][4][4] fp32: the target plane of each source pixel
 * in 4x4 tiles (64 contiguous bytes), ht = ceil(H_l/4), wt = ceil(W_l/4), zeros beyond the map.  The
 * producer is the correlation GEMM run against woft_tile_rows(fmap2_l).  coords: [P][2] (x,y) at level 0.
 * out: [P][ldo] with channel l*(2r+1)^2 + i*(2r+1) + j  <-  sample at (x/2^l + i - r, y/2^l + j - r). */
typedef struct woft_lookup_params {
    const void* vol[4];     /* fp32, or bf16 when vol_bf16 (woft_corr_gemm_bf16 with out_bf16) */
    int32_t ht[4], wt[4];   /* tiles per column / row at level l */
    int64_t plane[4];       /* elements per source pixel at level l (>= ht*wt*16) */
    int32_t levels, radius;
    const float* coords;
    int64_t n_pix;
    float* out;
    int32_t ldo;
    int32_t vol_bf16;       /* 0: fp32 volume (2896 B per pixel and call, r = 4), 1: bf16 storage (2096 B), fp32 out */
    int32_t ablate;         /* developer knob of tools/bench_lookup.py (1: no volume reads, 2: no output); 0 in production */
} woft_lookup_params;
int woft_corr_lookup(const woft_lookup_params* p, void* stream);
/* Volume-free correlation lookup (the reference's alternate_corr path: corr.py:72-100 and its alt_cuda_corr
 * extension): the same samples as woft_corr_lookup computed directly from the feature maps,
 * corr_l(p, q) = alpha * <f1[p], f2_l[q]>, f2_l = fmap2 average-pooled l times -- no P x P volume in memory.
 * f1 / f2[l]: row-major split operands in the format of woft_corr_gemm_bf16 (terms = 3: woft_split_bf16_lines,
 * terms = 1: the bf16 plane; terms = 0, exact fp32: the fp32 feature rows themselves, k % 32 == 0, products on
 * v_mfma_f32_32x32x2_f32 in the order of woft_conv2d's fp32 kernel), one row of k features per pixel; every correlation
 * value is the one that GEMM would have produced (same products, same order).  Cost grows with the spread of the flow inside each 8 x 8
 * block of source pixels (bounding box of their windows); results do not depend on it. */
typedef struct woft_lookup_otf_params {
    const void* f1;         /* [hf*wf][k] source features (split)                                  */
    const void* f2[4];      /* [h[l]*w[l]][k] target features of level l (split)                    */
    int32_t h[4], w[4];
    int32_t levels, radius, terms;
    int32_t hf, wf, k;
    float alpha;            /* 1 / sqrt(k)  (corr.py:68)                                            */
    const float* coords;    /* [hf*wf][2]                                                           */
    float* out;             /* [hf*wf][ldo], channel order of woft_corr_lookup                      */
    const int32_t* need;    /* optional [hf*wf]: source pixels whose samples are wanted (non-zero); an 8 x 8 block without
                               any is skipped (its output rows are left untouched).  NULL: every pixel                 */
    int32_t ldo;
    int32_t ablate;         /* developer knob of tools/bench_lookup_otf.py (1: no target-row stream after the first steps,
                               2: no MFMAs, 4: no window scatter, 8: no interpolation / output); 0 in production */
    /* Optional (fh_part != NULL): the previous refinement iteration's woft_flow_head_gather, folded into this launch.
       Before a workgroup reads the lookup centres of its 8 x 8 source pixels it finishes the flow head for exactly those
       pixels -- delta = fh_bias + 3x3 sum of the partial products (same operations and order as woft_flow_head_gather) --
       and writes coords += delta (coords is then read-write), fh_delta, fh_flow4 and fh_flow_cat as that call does.  No
       other workgroup reads these pixels' coordinates, and every later launch on the stream sees the updated flow.       */
    const float* fh_part;   /* [fh_planes][hf*wf][fh_ld] (fh_ld >= 20, % 4 == 0)                     */
    const float* fh_bias;   /* [2] or NULL                                                          */
    float* fh_delta;        /* [hf*wf][fh_ld_delta]                                                 */
    float* fh_flow4;        /* [hf*wf][4] or NULL                                                   */
    float* fh_flow_cat;     /* [hf*wf][fh_ld_cat] (2 values written) or NULL                        */
    int32_t fh_planes, fh_ld, fh_ld_delta, fh_ld_cat;
} woft_lookup_otf_params;
int woft_corr_lookup_otf(const woft_lookup_otf_params* p, void* stream);
/* NHWC map [h][w][c] -> its rows in 4x4-tile order [(ceil(h/4)*ceil(w/4)*16)][c], zero rows outside
 * the map: the B operand of the correlation GEMM that yields the tiled volume layout above. */
int woft_tile_rows(const float* in, int32_t h, int32_t w, int32_t c, float* out, void* stream);

/* coords1 += delta; flow = coords1 - coords0 (weighted_raft.py:232,237).
 * delta: [P][ld_delta] (first two channels); flow4: [P][4] = (fx, fy, 0, 0);
 * flow_cat: optional, writes (fx, fy) at flow_cat[p*ld_cat + 0..1]. */
int woft_coords_update(float* coords1, const float* delta, int32_t ld_delta, int32_t wf, int64_t n_pix,
                       float* flow4, float* flow_cat, int32_t ld_cat, void* stream);
int woft_coords_init(float* coords1, int32_t hf, int32_t wf, float* flow4, float* flow_cat,
                     int32_t ld_cat, void* stream);

/* Weight-head glue (weighted_raft.py:258-279, 347-384).
 * woft_colsum: total[c] = sum_q f[q][c] in fp64 (ws: [n_part][c] doubles).
 * woft_wh_pack: mean[p] = alpha * <f1[p], total>  (= mean_q vol[p,q], weighted_raft.py:358-361, in its
 *   algebraic form) and the head's input patches (weighted_raft.py:267-272, 363-376):
 *   x8[p][hp][wp][0..3] = lookup[p][(hp*nwin + wp)*4 + 0..3], x8[..][4] = mean[p], x8[..][5..7] = 0.
 * woft_wh_reduce: final 1x1 conv + mean over the patch (weighted_raft.py:341,378-383):
 *   out[p] = bias + mean_t <w, act[p][t][:]>
 * woft_wh_pack with x8 == NULL computes mean[] only.
 * woft_wh_conv0: the head's first conv + ReLU (weighted_raft.py:336; 5 -> 128 channels, 3x3, zero padding) straight
 *   from the lookup buffer and mean[] (same input definition as x8), exact fp32:
 *   out[p][hp][wp][0..127]; wt: fp32 [96][128], wt[(ky*32 + kx*8 + ci)*128 + co]; nwin in {7, 9}.
 *   index (optional): window p of the output is source pixel index[p] (n_pix = number of windows): the head
 *   evaluated on a subset of the source pixels, e.g. the tracker's template-mask region. */
int woft_colsum(const float* f, int64_t n_pix, int32_t c, double* ws, int32_t n_part, double* total, void* stream);
int woft_wh_conv0(const float* lookup, int32_t ld_lookup, const float* mean, int64_t n_pix, int32_t nwin,
                  const float* wt, const float* bias, float* out, const int32_t* index, void* stream);
int woft_wh_pack(const float* lookup, int32_t ld_lookup, const float* f1, int32_t c, const double* f2_total,
                 float alpha, int64_t n_pix, int32_t nwin, float* mean, float* x8, void* stream);
int woft_wh_reduce(const float* act, int32_t c, int32_t nwin2, const float* w, float bias,
                   int64_t n_pix, float* out, void* stream);
/* The windows of a window list that a set of full-resolution pixels needs: dyn_index[j] = index[j] if the 1/8-res pixel
 * index[j] lies in the 3x3 neighbourhood of the cell of one of the n = min(count[0], n_max) points pts[i] = (x, y) (image
 * coordinates; cell = ((y + top) >> 3, (x + left) >> 3): the support of the x8 convex upsampling, weighted_raft.py:92-103),
 * else -1.  A negative entry of woft_conv_params.wh0_index / out_index makes the weight-head launches skip that window.
 * bitmap: hf*wf int32 of scratch; n_needed (optional): receives the number of kept windows.  The tracker's fit reads the
 * weights of its Sobol-sampled correspondences only (TRK:287-312 + the subsampler), and the flow alone decides which. */
int woft_wh_needed(const float* pts, const int32_t* count, int32_t n_max, int32_t top, int32_t left, int32_t hf, int32_t wf,
                   const int32_t* index, int32_t n_win, int32_t* bitmap, int32_t* dyn_index, int32_t* n_needed, void* stream);

/* Convex upsampling of flow and weight logits + TC epilogue
 * (weighted_raft.py:92-103,285-288; optical_flow/raft.py:148-159,185-199).
 * coords1: [P][2]; wlow: [P] or NULL; mask: [P][ld_mask] channel k*64 + i*8 + j.
 * Crops to the unpadded window (crop_top, crop_left, h, w) and writes
 *   flow_up [2][h][w]            (may be NULL)
 *   dst     [2][h*w] = grid + flow (may be NULL)
 *   wout    [h*w]   = weights_up (logit, or sigmoid when do_sigmoid)  (NULL if wlow NULL) */
int woft_convex_upsample(const float* coords1, const float* wlow, const float* mask, int32_t ld_mask,
                         int32_t hf, int32_t wf, int32_t crop_top, int32_t crop_left, int32_t h, int32_t w,
                         float* flow_up, float* dst, float* wout, int32_t do_sigmoid, void* stream);
/* The weight output of woft_convex_upsample at a list of pixels only: wsel[i] = wout at pixel pts[i] = (x, y) of the
 * un-padded image, i < min(count[0], n_max) (count: device, may be NULL); same operations, same order. */
int woft_convex_weights_at(const float* pts, const int32_t* count, int32_t n_max, const float* wlow, const float* mask,
                           int32_t ld_mask, int32_t hf, int32_t wf, int32_t crop_top, int32_t crop_left,
                           int32_t do_sigmoid, float* wsel, void* stream);
/* Pre-computed flow read from the reference's cache files (utils/caching.py:53-59; raft.py:93-106) to the same
 * outputs: dst[2][h*w] = pixel grid + flow[2][h*w] (may be NULL), wout[h*w] = weights or sigmoid(weights)
 * (weights / wout may be NULL). */
int woft_flow_to_tc(const float* flow, const float* weights, int32_t h, int32_t w, float* dst, float* wout,
                    int32_t do_sigmoid, void* stream);
/* bilinear x8 upsampling, align_corners=True, times 8 (utils/utils.py:82-84), same outputs. */
int woft_upflow8(const float* coords1, const float* wlow, int32_t hf, int32_t wf,
                 int32_t crop_top, int32_t crop_left, int32_t h, int32_t w,
                 float* flow_up, float* dst, float* wout, int32_t do_sigmoid, void* stream);

/* Perspective warp, dst(x) = bilinear src(Hinv x), zeros outside
 * (tracker/YAOF_tracker_single_control.py:89-95).  hinv: 9 doubles on the HOST (by value copy).
 * img: HWC uint8 (c channels).  valid (may be NULL): uint8 1 where warp(ones) > 0. */
int woft_warp_perspective_u8(const uint8_t* img, int32_t h, int32_t w, int32_t c, const double* hinv,
                             uint8_t* out, uint8_t* valid, int32_t nearest, void* stream);

/* cv2.resize(img, None, fx, fy) with INTER_LINEAR geometry (tracker/YAOF_tracker_single_control.py:27-30,60-61,
 * `downscale_inputs`): src = (dst + 0.5) * scale - 0.5, edge clamped; scale = 1 / fx. */
int woft_resize_linear_u8(const uint8_t* img, int32_t h, int32_t w, int32_t c, uint8_t* out, int32_t ho, int32_t wo,
                          float scale_y, float scale_x, void* stream);

/* Correspondence masking + order-preserving compaction + Sobol subsampling on the device
 * (tracker/YAOF_tracker_single_control.py:287-327; configs/..._wLSq.py:31-53), outputs in woft_hfit's format.
 * The correspondences live on the flow grid gh x gw (the frame; with padding_mode 'crop' the frame cropped to a
 * multiple of 8, optical_flow/raft.py:235-247), the masks have the frame's size mh x mw (gh <= mh, gw <= mw).
 * dst: [2][gh*gw] (x plane, y plane) target coords of source pixel i = y*gw + x; w: [gh*gw] or NULL; tmask, pwmask:
 * uint8 [mh][mw]; check_dst != 0 adds the bounds test of dst against (mw, mh) and, if pwmask != NULL,
 * pwmask[rint(dy)][rint(dx)].
 * sobol_u: [n_draw] float32 1-D Sobol points (n_draw <= 1024; 0 = keep all).  ws: woft_tc_select_ws_bytes(gh*gw) bytes.
 * pa[k] = (dst_x, dst_y), pb[k] = (src_x, src_y), wout[k]; count[0] = selected (<= cap), count[1] = kept by the masks. */
int64_t woft_tc_select_ws_bytes(int64_t n);
int woft_tc_select(const float* dst, const float* w, const uint8_t* tmask, const uint8_t* pwmask, int32_t gh,
                   int32_t gw, int32_t mh, int32_t mw, int32_t check_dst, const float* sobol_u, int32_t n_draw,
                   void* ws, float* pa, float* pb, float* wout, int32_t cap, int32_t* count, void* stream);
/* The keep rule of woft_tc_select alone (`_mask_coords` / `_mask_coords_flow`, YAOF_tracker_single_control.py:287-327):
 * flags[i] = 1 where correspondence i survives, uint8 [gh*gw].  For callers that compact with their own code (the
 * tracker's generic path hands the compacted tensors to the config's subsampler / estimator callables). */
int woft_tc_flags(const float* dst, const uint8_t* tmask, const uint8_t* pwmask, int32_t gh, int32_t gw, int32_t mh,
                  int32_t mw, int32_t check_dst, uint8_t* flags, void* stream);

/* Weighted / iteratively re-weighted least-squares homography, utils/least_squares_H.py:142-210
 * (n_irls = 0) and :280-346 (n_irls = 5 -> 6 solves); reweight: 0 none, 1 L1 (:268-269),
 * 2 Huber(k) (:272-277).  pa, pb: [n][2] points (A -> B), w: [n] or NULL; n = min(count[0], n_max)
 * when count != NULL (device), else n_max.  Hout: 9 floats (row major, device).
 * status[0] (device) = 0 ok, 1 fewer than 4 points, 2 singular system.
 * ws: NULL, or woft_hfit_ws_bytes() bytes of device scratch: with it, fits of more than WOFT_HFIT_SINGLE_MAX
 * correspondences (configs without a subsampler: up to H*W) run as a streaming multi-workgroup pipeline instead of
 * in one workgroup; same arithmetic (fp32 rows, fp64 Gram matrix, fp64 Cholesky; only the summation order differs). */
#define WOFT_HFIT_SINGLE_MAX 2048
int64_t woft_hfit_ws_bytes(void);
int woft_hfit(const float* pa, const float* pb, const float* w, int32_t n_max, const int32_t* count,
              int32_t reweight, float huber_k, int32_t n_irls, void* ws, float* Hout, int32_t* status, void* stream);
/* ONE re-weighted solve of the same system, for arbitrary `reweighting_fn` callables (least_squares_H.py:280,323-337):
 * rew: [2n] per-row re-weights sqrt(reweighting_fn(residual)) of the previous step or NULL (= ones, first step);
 * first != 0 (re)computes the normalisation into ws; res (may be NULL): [2n] residuals A x - b of THIS step's solution
 * on the weighted, not re-weighted, system (:334) -- the host applies the user's callable to them and calls again;
 * Hout / status as woft_hfit (H of this step's solution, de-normalised). */
int woft_hfit_step(const float* pa, const float* pb, const float* w, int32_t n, const float* rew, int32_t first,
                   void* ws, float* res, float* Hout, int32_t* status, void* stream);
/* torch_proj_errors + inlier fraction (least_squares_H.py:474-489; configs/..._wLSq.py:14-21):
 * frac[0] = mean(|proj(H, A) - B| <= thr). */
int woft_inlier_frac(const float* pa, const float* pb, int32_t n_max, const int32_t* count, const float* H,
                     float thr, float* frac, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WOFT_HIP_H */
