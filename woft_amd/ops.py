"""Thin Python layer over the C ABI: tensor bookkeeping, weight packing, one function per kernel.

All tensors are torch CUDA (=HIP) tensors; the library enqueues on torch's current stream.
Activations are NHWC fp32: a tensor of shape (n_pix, channel_stride) (pixels of all images
flattened) -- `Act` carries the geometry.
"""
import ctypes as C
import math
import os
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from ._lib import ConvParams, LookupOtfParams, LookupParams, check, ptr, stream_ptr

DEV = "cuda"


def _round_up(x, m):
    return (x + m - 1) // m * m


@dataclass
class Act:
    """NHWC activation: t is (n_img*h*w, cs) fp32 contiguous; `c` valid channels."""
    t: torch.Tensor
    n: int
    h: int
    w: int
    c: int

    @property
    def cs(self):
        return self.t.shape[1]

    @property
    def n_pix(self):
        return self.n * self.h * self.w

    def nchw(self):
        """(n, c, h, w) copy -- test / boundary helper."""
        return self.t[:, :self.c].reshape(self.n, self.h, self.w, self.c).permute(0, 3, 1, 2).contiguous()


def _rows(n_pix, cs, zero):
    return (torch.zeros if zero else torch.empty)(n_pix, cs, dtype=torch.float32, device=DEV)


def act_from_nchw(x, cs=None):
    n, c, h, w = x.shape
    cs = cs or _round_up(c, 4)
    t = _rows(n * h * w, cs, True)
    t[:, :c] = x.to(DEV).permute(0, 2, 3, 1).reshape(n * h * w, c)
    return Act(t, n, h, w, c)


def new_act(n, h, w, c, cs=None, zero=False):
    cs = cs or _round_up(c, 4)
    return Act(_rows(n * h * w, cs, zero), n, h, w, c)


# ------------------------------------------------------------------------------------------
# weight packing
# ------------------------------------------------------------------------------------------
@dataclass
class PackedConv:
    wgt: torch.Tensor        # (cout_pad, taps*cin_pad) fp32 on device
    bias: torch.Tensor       # (cout_pad,)
    cout: int
    cout_pad: int
    cin_pad: int
    taps_y: int
    taps_x: int
    pad_y: int
    pad_x: int
    stride: int
    flat: int
    flat_cs: int = 0
    kh: int = 0              # real kernel size (flat packing folds kw into the K chunk)
    kw: int = 0
    wgt_hi: torch.Tensor = None   # bf16 split of wgt: hi = bf16(w), lo = bf16(w - hi)
    wgt_lo: torch.Tensor = None
    _frag: dict = None            # planes -> weights in MFMA-fragment order (woft_conv_params.wgt_frag), built on demand
    _f16: torch.Tensor = None     # fp16(w) in a bf16-typed container (precision "fp16"), built on demand

    def wgt_f16(self):
        if self._f16 is None:
            w = self.wgt.detach().cpu()
            if float(w.abs().max()) >= 65504.0:
                raise ValueError("precision 'fp16': a weight exceeds the fp16 range")
            self._f16 = w.to(torch.float16).view(torch.bfloat16).to(self.wgt.device)
        return self._f16

    def frag(self, planes, f16=False):
        """[cout_pad/32 bands][chunks][taps][planes][2 k halves][64 lanes][8] bf16 (see woft_conv_params.wgt_frag);
        f16: one plane of fp16 values in the same 16-bit containers (precision "fp16")."""
        if self._frag is None:
            self._frag = {}
        key = "f16" if f16 else planes
        if key not in self._frag:
            taps, nchunk = self.taps_y * self.taps_x, self.cin_pad // 32
            w = self.wgt.detach().cpu().reshape(self.cout_pad // 32, 32, taps, nchunk, 2, 2, 8)
            w = w.permute(0, 3, 2, 4, 5, 1, 6).contiguous()        # band, chunk, tap, k half, lane half, row, e
            if f16:
                pl = [w.to(torch.float16).view(torch.bfloat16)]
            else:
                hi = w.to(torch.bfloat16)
                pl = [hi] + ([(w - hi.float()).to(torch.bfloat16)] if planes == 2 else [])
            self._frag[key] = torch.stack(pl, dim=3).contiguous().to(self.wgt.device)
        return self._frag[key]

    def frag_mx(self):
        """fp8 fragments of the weights for precision "f16mx8" (woft_conv_params.wgt_mx): per 32-column band, 32-channel chunk, tap
        pair and term (0: w, 1: w - fp16(w)) -- 64 lanes x 32 bytes e4m3 (lane L: column L % 32; bytes 0-15 = channels 16 (L // 32)
        .. + 15 at the pair's first tap, bytes 16-31 at its second tap; an odd last tap pairs with zeros), then 64 int32 scales: lane
        half 0 carries the E8M0 scale of the (column, first tap, chunk) block, lane half 1 that of the second tap's block.  A block's
        scale puts its largest magnitude into [128, 256) (e4m3 holds up to 448)."""
        if self._frag is None:
            self._frag = {}
        if "mx" not in self._frag:
            taps, nchunk = self.taps_y * self.taps_x, self.cin_pad // 32
            npair = (taps + 1) // 2
            w = self.wgt.detach().cpu().reshape(self.cout_pad, taps, nchunk, 32)
            if float(w.abs().max()) >= 65504.0:
                raise ValueError("precision 'f16mx8': a weight exceeds the fp16 range")
            lw = w - w.to(torch.float16).float()
            bands = self.cout_pad // 32
            out = torch.zeros(bands, nchunk, npair, 2, 64 * 32 + 64 * 4, dtype=torch.uint8)
            for t, x in enumerate((w, lw)):
                amax = x.abs().amax(dim=3)                                            # [cout_pad][taps][nchunk]
                e = torch.floor(torch.log2(torch.clamp(amax, min=1e-38)))
                sc = torch.where(amax > 0, e - 7 + 127, torch.full_like(e, 127.0)).clamp(0, 254)   # E8M0 byte
                q = (x / torch.exp2(sc - 127)[..., None]).to(torch.float8_e4m3fn).view(torch.uint8)    # [cout_pad][taps][nchunk][32]
                if taps % 2:                                                          # zero weights (and unit scale) behind an odd last tap
                    q = torch.cat([q, torch.zeros_like(q[:, :1])], 1)
                    sc = torch.cat([sc, torch.full_like(sc[:, :1], 127.0)], 1)
                q = q.reshape(bands, 32, npair, 2, nchunk, 2, 16)                     # band, col, pair, tap-in-pair, chunk, hh, 16 channels
                data = q.permute(0, 4, 2, 5, 1, 3, 6).reshape(bands, nchunk, npair, 64 * 32)    # lane = 32 hh + col; [tap0 16 B | tap1 16 B]
                scl = sc.reshape(bands, 32, npair, 2, nchunk).permute(0, 4, 2, 3, 1).reshape(bands, nchunk, npair, 64)   # lane = 32 tap-in-pair + col
                scl4 = torch.zeros(bands, nchunk, npair, 64, 4, dtype=torch.uint8)
                scl4[..., 0] = scl.to(torch.uint8)
                out[:, :, :, t, :64 * 32] = data
                out[:, :, :, t, 64 * 32:] = scl4.reshape(bands, nchunk, npair, 256)
            self._frag["mx"] = out.contiguous().to(self.wgt.device)
        return self._frag["mx"]

    def __post_init__(self):
        if self.wgt is not None and self.wgt_hi is None and self.wgt.dtype == torch.float32:
            w = self.wgt.detach().cpu()
            hi = w.to(torch.bfloat16)
            self.wgt_hi = hi.to(self.wgt.device)
            self.wgt_lo = (w - hi.float()).to(torch.bfloat16).to(self.wgt.device)

    def out_hw(self, h, w):
        kh, kw = (self.kh or self.taps_y), (self.kw or self.taps_x)
        return (h + 2 * self.pad_y - kh) // self.stride + 1, (w + 2 * self.pad_x - kw) // self.stride + 1


def pack_conv(weight, bias, stride=1, padding=None, cin_layout=None, flat_cs=0, scale=1.0):
    """OIHW conv weight -> [cout_pad][taps][cin_pad] (K contiguous).

    cin_layout: list of (src_begin, src_end, dst_begin) channel ranges mapping the reference's
                input-channel order onto the (padded) channel order of our input buffers;
                default = identity.
    flat_cs:    >0 -> "flat" packing for tiny Cin: the K chunk of tap ky is kw pixels x flat_cs
                channels laid out as the NHWC row itself (k = kx*flat_cs + c), padded to 32.
    """
    weight = weight.detach().float().cpu() * scale
    cout, cin, kh, kw = weight.shape
    if padding is None:
        padding = (kh // 2, kw // 2)
    if isinstance(padding, int):
        padding = (padding, padding)
    cout_pad = _round_up(cout, 64)
    if cout_pad > 64:
        cout_pad = _round_up(cout, 128)
    b = torch.zeros(cout_pad)
    if bias is not None:
        b[:cout] = bias.detach().float().cpu() * scale
    if flat_cs:
        assert kw * flat_cs <= 32 and cin <= flat_cs
        wp = torch.zeros(cout_pad, kh, 32)
        for kx in range(kw):
            wp[:cout, :, kx * flat_cs:kx * flat_cs + cin] = weight[:, :, :, kx].permute(0, 2, 1)
        return PackedConv(wp.reshape(cout_pad, kh * 32).contiguous().to(DEV), b.to(DEV), cout, cout_pad, 32,
                          kh, 1, padding[0], padding[1], stride, 1, flat_cs, kh, kw)
    if cin_layout is None:
        cin_layout = [(0, cin, 0)]
    cin_pad = _round_up(max(d + (e - s) for s, e, d in cin_layout), 32)
    wp = torch.zeros(cout_pad, kh * kw, cin_pad)
    wt = weight.permute(0, 2, 3, 1).reshape(cout, kh * kw, cin)
    for s, e, d in cin_layout:
        wp[:cout, :, d:d + (e - s)] = wt[:, :, s:e]
    return PackedConv(wp.reshape(cout_pad, kh * kw * cin_pad).contiguous().to(DEV), b.to(DEV), cout, cout_pad,
                      cin_pad, kh, kw, padding[0], padding[1], stride, 0, 0, kh, kw)


def fold_bn(weight, bias, bn_w, bn_b, mean, var, eps=1e-5):
    """Eval-mode BatchNorm folded into the preceding conv (extractor.py:22-26)."""
    s = (bn_w.double() / torch.sqrt(var.double() + eps))
    w = (weight.double() * s[:, None, None, None]).float()
    b = ((bias.double() - mean.double()) * s + bn_b.double()).float()
    return w, b


# "f16mx8" (round 4): fp16 main term + two block-scaled fp8 cross terms -- an fp32-emulating product in two matrix-pipe passes instead
# of bf16x3's three (woft_conv_params.wgt_mx); layers whose kernel has no such instance run in bf16x3
PRECISION = {"fp32": 0, "bf16x3": 1, "bf16": 2, "fp16": 3, "f16mx8": 4}
# which layers precision "f16mx8" actually runs in two passes: "auto" (default) = where it measured faster than bf16x3; "all" = every
# multi-tap layer of the register-streamed kernel (the kernel tests use it)
MX_LAYERS = os.environ.get("WOFT_MX_LAYERS", "auto")
SLOW_GATES = os.environ.get("WOFT_SLOW_GATES", "0") != "0"     # developer A/B: libm sigmoid / tanh in the conv epilogues
USE_HALO = os.environ.get("WOFT_HALO", "1") != "0"
USE_REGB = os.environ.get("WOFT_REGB", "1") != "0"
REGB_TY4 = os.environ.get("WOFT_REGB_TY4", "1") != "0"
USE_STEM = os.environ.get("WOFT_STEM", "1") != "0"        # 7x7 / stride-2 first layer on conv_stem.hip (0: gather kernel)
USE_1X1 = os.environ.get("WOFT_1X1", "1") != "0"          # wide 1x1 layers on conv_1x1.hip (0: gather kernel)
HALO_TILES = {1: (8, 16, 1), 2: (9, 9, 1), 4: (4, 16, 1), 7: (8, 16, 1), 8: (8, 16, 1), 12: (4, 16, 1)}     # (TY, TX, images per workgroup)
HALO_MIN_BLOCKS = int(os.environ.get("WOFT_HALO_MIN_BLOCKS", "400"))
WH_HALO = int(os.environ.get("WOFT_WH_HALO", "2"))


TILE_MIN_BLOCKS = int(os.environ.get("WOFT_TILE_MIN_BLOCKS", "400"))


def pick_tiles(m, cout_pad):
    """Block tile: 128-wide in N when the padded cout allows; 128 rows in M when that still yields
    at least TILE_MIN_BLOCKS workgroups (256 CUs), else 64."""
    tn = 128 if cout_pad % 128 == 0 else 64
    blocks128 = math.ceil(m / 128) * (cout_pad // tn)
    tm = 128 if blocks128 >= TILE_MIN_BLOCKS else 64
    return tm, tn


# ------------------------------------------------------------------------------------------
# kernels
# ------------------------------------------------------------------------------------------
def conv_params(x, pc, out, co_off=0, epi=_lib.EPI_LINEAR, x2=None, c_split=0, e0=None, e1=None, out1=None,
                split=0, alpha=1.0, stats=None, ho=None, wo=None, tiles=None, cout=None, precision=0, halo=None,
                in_norm=0, in_stats=None, bias_map=None, x2_off=0, wh0=None):
    """Build (and keep alive) a woft_conv_params for `out[:, co_off:co_off+cout] = epi(conv(x))`.
    in_norm = 1 / 2 with in_stats = (mean, rstd): x is a RAW conv output, InstanceNorm (2: + ReLU) applied while
    loading -- LDS-halo kernel only (check p.halo on the result; the caller falls back to woft_inorm_apply).
    wh0 = (lookup Act, mean, fragments (pack_wh0_frags), bias, index | None): the weight head's first conv is
    evaluated inside this launch from the lookup windows and x is not read (whole-window kernel only: p.halo == 2)."""
    if ho is None or wo is None:
        ho, wo = pc.out_hw(x.h, x.w)
    p = ConvParams()
    p.in0, p.cs0 = ptr(x.t), x.cs
    if x2 is not None:                  # (x2_off: first channel of x2 used as the second source)
        p.in1, p.cs1, p.c_split = x2.t.data_ptr() + 4 * x2_off, x2.cs, c_split
    else:
        p.in1, p.cs1, p.c_split = None, 0, pc.cin_pad
    p.n_img, p.h, p.w, p.ho, p.wo = x.n, x.h, x.w, ho, wo
    p.taps_y, p.taps_x, p.stride, p.pad_y, p.pad_x = pc.taps_y, pc.taps_x, pc.stride, pc.pad_y, pc.pad_x
    p.cin_pad, p.flat = pc.cin_pad, pc.flat
    p.wgt, p.bias, p.alpha = ptr(pc.wgt), ptr(pc.bias), alpha
    p.wgt_hi, p.wgt_lo, p.precision = ptr(pc.wgt_hi), ptr(pc.wgt_lo), PRECISION.get(precision, precision)
    if p.precision == 3:                # fp16 operands: the weight plane holds fp16 values
        p.wgt_hi, p.wgt_lo = ptr(pc.wgt_f16()), None
    p.cout, p.cout_pad = (cout or pc.cout), pc.cout_pad
    p.out, p.ldo, p.co_off = ptr(out.t), out.cs, co_off
    p.out_w, p.out_pitch = (-12346 if SLOW_GATES else 0), 0
    p.epi, p.split = epi, split
    p.e0, p.lde0 = (ptr(e0.t), e0.cs) if e0 is not None else (None, 0)
    p.e1, p.lde1 = (ptr(e1.t), e1.cs) if e1 is not None else (None, 0)
    p.out1, p.ldo1 = (ptr(out1.t), out1.cs) if out1 is not None else (None, 0)
    m = x.n * ho * wo
    tm, tn = tiles or pick_tiles(m, pc.cout_pad)
    if tiles is None and p.precision == 0 and tm == 128 and math.ceil(m / 128) * (pc.cout_pad // tn) < 2048:
        tm = 64                         # fp32 kernel: 64-row tiles up to ~2000 workgroups (3-15 % per layer, gather_sweep.py fp32)
    if tiles is None and p.precision != 0 and (pc.flat or pc.taps_y * pc.taps_x == 1):
        tm, tn = 64, 64                 # short-K gather layers (1x1, flat 7x7): 64 x 64 tiles measured 5-35 % faster
                                        # than 128-wide ones at every resolution of a 1080p frame (tools/gather_sweep.py)
    if tiles is None and tn == 128 and stats is None and _round_up(p.cout, 64) < pc.cout_pad:
        tn = 64                         # the last 64 columns of the 128-padded weight matrix are padding (cout 192, 576)
    if tn == 64 and stats is None:
        p.cout_pad = _round_up(p.cout, 64)              # column tiles actually launched (convc2: 3 instead of 4 x 64)
    p.tile_m, p.tile_n = tm, tn
    # LDS-halo kernel for the split-bf16 precisions on stride-1 multi-tap convs (see conv.hip)
    auto_halo = halo is None
    if halo is None:
        halo = 0
        if USE_HALO and p.precision != 0 and not pc.flat and pc.stride == 1 and pc.taps_y * pc.taps_x > 1 \
                and (ho, wo) == (x.h, x.w) and (pc.taps_y, pc.taps_x) in ((3, 3), (1, 5), (5, 1)):
            nt = pc.cout_pad // tn
            if (x.h, x.w) == (9, 9) and tn == 128:
                halo = WH_HALO if x.n >= 4 * 256 else 2
            elif x.h >= 8 and x.w >= 16:
                # Tile choice, measured on MI355X with tools/tile_sweep.py (1080p layer shapes).  Many independent
                # workgroups beat larger tiles (16x16 / multi-patch workgroups run at 1 block per CU and lose to
                # 8x16 by 1.5-3x): take the 8x16 pixel tile only while it still yields ~2 workgroups per CU, else 4x16.
                b816 = x.n * math.ceil(ho / 8) * math.ceil(wo / 16)
                auto = tiles is None
                if auto and p.precision in (PRECISION["bf16"], PRECISION["fp16"]):
                    # plain bf16 (one LDS plane, fewer registers): 64-channel column tiles win throughout --
                    # 8x16 x 64 for the 256-wide layers (~1000 workgroups), 4x16 x 64 for the narrower ones
                    tn = 64
                    if stats is None:
                        p.cout_pad = _round_up(p.cout, 64)
                    halo = 1 if 4 * b816 * (p.cout_pad // 64) >= 7 * HALO_MIN_BLOCKS else 4   # (>= 700 workgroups)
                else:
                    # bf16x3: 64-channel column tiles (three workgroups per CU) unless their grid lands just over
                    # one round of the 768 resident slots while the 128-wide tiles (two per CU) still fit in one
                    # round -- the 256-wide layers at 1/8 of 1080p; 4x16 pixel tiles when 8x16 gives too few workgroups
                    if auto:
                        w64 = b816 * math.ceil(p.cout / 64)
                        wide_ok = pc.cout_pad % 128 == 0 and b816 * (pc.cout_pad // 128) <= 512
                        if tn == 128 and not (wide_ok and 768 < w64 < 1536):
                            tn = 64
                        if tn == 64 and stats is None:   # (with statistics the rows keep the padded width)
                            p.cout_pad = _round_up(p.cout, 64)
                    halo = 1 if b816 * (p.cout_pad // tn) >= HALO_MIN_BLOCKS else 4
                p.tile_n = tn
    # the encoders' first layer (7x7, stride 2, 3 -> 64 on the NHWC4 image, flat packing): its own kernel (conv_stem.hip, halo 7)
    # -- bit-identical to the gather kernel, 8x16-pixel tiles (the statistics rows follow them)
    if USE_STEM and auto_halo and halo == 0 and tiles is None and p.precision != 0 and pc.flat and x.cs == 4 and x2 is None \
            and (pc.taps_y, pc.taps_x, pc.stride, pc.pad_y, pc.pad_x, pc.cin_pad) == (7, 1, 2, 3, 3, 32) \
            and pc.cout_pad % 64 == 0 and not in_norm and bias_map is None and wh0 is None \
            and (ho, wo) == ((x.h - 1) // 2 + 1, (x.w - 1) // 2 + 1):
        halo = 7
        p.tile_n = tn = 64
    # stride-1 multi-tap layers without InstanceNorm plumbing: the kernel that streams the weights global -> registers
    # (conv_regb.hip, halo 8) -- bit-identical, 1-8 % faster per layer on the update block's shapes (tools/regb_check.py)
    if USE_REGB and auto_halo and halo in (1, 4) and tiles is None and stats is None and not in_norm and p.precision != 0:
        p.tile_n = tn = (p.tile_n if halo == 1 else 64)
        halo = 8
    # the GRU q convs (1x5 / 5x1, 128 columns) on 4x16-pixel x 128-column tiles instead of 8x16 x 64: same workgroup count and
    # per-wave work (64 rows x 32 columns), but the four waves are four column bands -- the weight fragments are fetched once
    # per workgroup instead of by both row halves.  Alone -6...8 % per layer at 1/8 of 1080p; inside a frame +-0 at 1080p
    # and 4K, +1.5 % frames/s at 720p.  (convm the same alone, nothing in a frame: left on 8x16 x 64; a layer that would pad
    # to 128 columns -- convc2, 192 -> 256 -- loses 19 %.)  WOFT_REGB_TY4=0: off
    if REGB_TY4 and auto_halo and halo == 8 and p.tile_n == 64 and pc.cout_pad % 128 == 0 and _round_up(p.cout, 64) == pc.cout_pad \
            and (pc.taps_y, pc.taps_x) in ((1, 5), (5, 1)):
        halo, p.tile_n = 12, 128
    # wide 1x1 / stride-1 layers (convc1: 324 -> 256; the encoders' closing 128 -> 256): the streamed GEMM kernel (conv_1x1.hip, halo
    # 16) -- 64 pixels x all 256 columns per workgroup, activations read and converted once per layer; bit-identical to the gather kernel
    if USE_1X1 and auto_halo and halo == 0 and tiles is None and stats is None and not in_norm and wh0 is None and p.precision != 0 \
            and not pc.flat and (pc.taps_y, pc.taps_x, pc.stride, pc.pad_y, pc.pad_x) == (1, 1, 1, 0, 0) and (ho, wo) == (x.h, x.w) \
            and pc.cout_pad % 256 == 0 and _round_up(p.cout, 256) == pc.cout_pad:
        # (layers that only fill 128-column tiles gain nothing: mask head conv2 256 -> 576 61.5 vs 59.9 us, 128 -> 128 13.8 vs 14.0)
        halo = 16
        p.tile_m, p.tile_n, p.cout_pad = 64, 256, pc.cout_pad
        tm = 64
    # ... and the flat-packed 7x7 conv on the flow (convf1, update.py:91) on the same kernel (128 columns per workgroup, K chunks = tap
    # rows), so that it keeps sharing convc1's launch (pair_ok)
    if USE_1X1 and auto_halo and halo == 0 and tiles is None and stats is None and not in_norm and wh0 is None and p.precision != 0 \
            and pc.flat and x2 is None and (pc.taps_x, pc.stride, pc.cin_pad) == (1, 1, 32) and 2 * pc.pad_y + 1 == pc.taps_y \
            and (ho, wo) == (x.h, x.w) and pc.cout_pad % 128 == 0 and _round_up(p.cout, 128) == pc.cout_pad:
        halo = 16
        p.tile_m, p.tile_n, p.cout_pad = 64, 128, pc.cout_pad
        tm = 64
    p.halo = halo
    p.wgt_frag = None
    p.wgt_mx = None
    if p.precision == 4 and not (halo in (8, 12) and pc.taps_y * pc.taps_x > 1 and not in_norm):
        p.precision = 1                 # f16mx8 exists on the register-streamed kernel's multi-tap instances: elsewhere bf16x3
    if p.precision == 4 and MX_LAYERS == "auto":
        # measured per layer at 1080p (profiles/r04_layer_times_f16mx8*.txt against ..._bf16x3.txt): the 3x3 layers are 8-10 % faster in
        # f16mx8 than in bf16x3 (motion encoder, flow head, context encoder); the GRU's 1x5 / 5x1 z|r layers are 4-7 % slower (four row
        # tiles per wave do not fit the 256 registers: half-size workgroups), the q layers equal, and the 128-column instance with a
        # full-width store epilogue (mask head conv: spills) 75 vs 55 us -> those stay bf16x3
        tn_mx = p.tile_n if p.tile_n in (64, 128) else (128 if pc.cout_pad % 128 == 0 else 64)
        if (pc.taps_y, pc.taps_x) != (3, 3) or (tn_mx == 128 and pc.cout_pad % 128 == 0 and epi != _lib.EPI_FLOWHEAD):
            p.precision = 1
    if p.precision == 4 and halo == 8 and tiles is None and p.tile_n == 128:
        # two row tiles per wave have the registers for the deep fragment pipeline; four (8 x 16 pixels x 128 columns) spill:
        # 1x5 / 5x1 layers take the 4 x 16-pixel x 128-column layout (WOFT_MX_ZR = 12; 64: 64-column tiles; 128: keep)
        # (3x3 layers keep their 128-column choice: that instance spills 40 bytes and still beats 64 columns, 52.7 vs 58.8 us on fh1)
        if (pc.taps_y, pc.taps_x) in ((1, 5), (5, 1)) and pc.cout_pad % 128 == 0 and _round_up(p.cout, 128) == pc.cout_pad \
                and os.environ.get("WOFT_MX_ZR", "12") == "12":
            halo = p.halo = 12
        elif (pc.taps_y, pc.taps_x) != (3, 3) and os.environ.get("WOFT_MX_ZR", "12") == "64":
            p.tile_n = 64
    if halo in (8, 12):                 # weights streamed to registers in MFMA-fragment order (conv_regb.hip)
        assert p.precision != 0 and not pc.flat and pc.stride == 1 and not in_norm
        assert (pc.taps_y, pc.taps_x) in ((3, 3), (1, 5), (5, 1)) and (ho, wo) == (x.h, x.w)
        frag = pc.frag(2 if p.precision == 1 else 1, f16=p.precision in (3, 4))
        p.wgt_frag = ptr(frag)
        if p.precision == 4:
            p.wgt_mx = ptr(pc.frag_mx())
        if tiles is None and p.tile_n not in (64, 128):
            p.tile_n = 128 if pc.cout_pad % 128 == 0 else 64
        if halo == 12:
            assert pc.cout_pad % 128 == 0 and pc.taps_y * pc.taps_x > 1
            p.tile_n = 128
        if p.tile_n == 128 and pc.cout_pad % 128 != 0:
            p.tile_n = 64
        p.cout_pad = pc.cout_pad if stats is not None else _round_up(p.cout, p.tile_n)
    if halo == 16:
        p.wgt_frag = ptr(pc.frag(2 if p.precision == 1 else 1, f16=p.precision == 3))
    p.bias_map, p.ld_bias_map = (ptr(bias_map.t), bias_map.cs) if bias_map is not None else (None, 0)
    p.in_norm, p.in_mean, p.in_rstd = 0, None, None
    if in_norm and halo in (1, 4) and (pc.taps_y, pc.taps_x) == (3, 3):   # (instantiated for the 3x3 pixel tiles)
        p.in_norm, p.in_mean, p.in_rstd = int(in_norm), ptr(in_stats[0]), ptr(in_stats[1])
    p._m_tiles = math.ceil(m / tm)
    if halo in HALO_TILES:
        ty, tx, g = HALO_TILES[halo]
        p._m_tiles = math.ceil(x.n / g) * math.ceil(ho / ty) * math.ceil(wo / tx)
    if stats is not None:
        rows = 2 * p._m_tiles
        assert stats[0].numel() >= rows * pc.cout_pad
        p.stat_sum, p.stat_sq = ptr(stats[0]), ptr(stats[1])
    else:
        p.stat_sum, p.stat_sq = None, None
    if wh0 is not None:
        lk, mean, frag, b0, index = wh0
        assert halo == 2 and p.precision != 0 and pc.cin_pad == 128, "fused first layer: 9x9 whole-window kernel only"
        p.wh0_lookup, p.wh0_ld, p.wh0_mean = ptr(lk.t), lk.cs, ptr(mean)
        p.wh0_w, p.wh0_bias, p.wh0_index = ptr(frag), ptr(b0), (ptr(index) if index is not None else None)
    p._keep = (x, x2, pc, out, e0, e1, out1, stats, in_stats, bias_map, wh0, p.wgt_frag and pc._frag)
    p._m = m
    return p


def pack_wh0_frags(w0, planes):
    """First conv of the weight head (128, 5, 3, 3) (weighted_raft.py:336) -> bf16 MFMA A fragments
    [4 chunks][3 K steps][planes][64 lanes][8] (woft_conv_params.wh0_w): k = (3 ky + kx) * 5 + ci, 45 -> 48."""
    k = torch.zeros(128, 48)
    k[:, :45] = w0.detach().float().cpu().permute(0, 2, 3, 1).reshape(128, 45)
    k = k.reshape(4, 32, 3, 2, 8).permute(0, 2, 3, 1, 4).reshape(4, 3, 64, 8)      # lane = 32 * (k half) + row
    hi = k.to(torch.bfloat16)
    pl = [hi] + ([(k - hi.float()).to(torch.bfloat16)] if planes == 2 else [])
    return torch.stack(pl, dim=2).contiguous().to(DEV)


def pack_flowhead_frags(w2, planes, f16=False):
    """FlowHead.conv2 weight (2, C, 3, 3) (update.py:11) -> bf16 MFMA B fragments [C/32 bands][2 k halves][planes][64 lanes][8]
    for WOFT_EPI_FLOWHEAD: lane = 32 * hh + j, column j = (3 ky + kx) * 2 + o (18 of 32 used), element e = channel
    32 band + 16 (k half) + 8 hh + e."""
    w2 = w2.detach().float().cpu()
    c = w2.shape[1]
    assert w2.shape[0] == 2 and tuple(w2.shape[2:]) == (3, 3) and c % 32 == 0
    cols = torch.zeros(32, c)
    cols[:18] = w2.permute(2, 3, 0, 1).reshape(18, c)                 # row j = (ky * 3 + kx) * 2 + o
    f = cols.reshape(32, c // 32, 2, 2, 8).permute(1, 2, 3, 0, 4).reshape(c // 32, 2, 64, 8)   # band, k half, (hh, j), e
    if f16:                             # precision "fp16": one plane of fp16 values in the 16-bit containers
        return f.to(torch.float16).view(torch.bfloat16).reshape(c // 32, 2, 1, 64, 8).contiguous().to(DEV)
    hi = f.to(torch.bfloat16)
    pl = [hi] + ([(f - hi.float()).to(torch.bfloat16)] if planes == 2 else [])
    return torch.stack(pl, dim=2).contiguous().to(DEV)


def flowhead_params(x, pc, part, frags, **kw):
    """conv_params for FlowHead.conv1 with the second conv folded into its epilogue (WOFT_EPI_FLOWHEAD), or None when the
    layer does not run on the register-streamed kernel (halo 8) here.  part: (planes * n_pix, >= 20) fp32."""
    p = conv_params(x, pc, Act(part, 1, x.h, x.w, 18), epi=_lib.EPI_FLOWHEAD, **kw)
    if p.halo != 8 or pc.cout % 32 != 0:
        return None
    n_planes = p.cout_pad // p.tile_n
    assert part.shape[0] >= n_planes * x.n_pix and part.shape[1] >= 20 and x.n == 1
    p.e0, p.lde0 = ptr(frags), 0
    p._keep = p._keep + (part, frags)
    p._n_planes = n_planes
    return p


def flow_head_gather(part, n_planes, h, w, bias2, delta, coords, flow4=None, flow_cat=None, ld_cat=0):
    check(_lib.load().woft_flow_head_gather(ptr(part), n_planes, part.shape[1], h, w, ptr(bias2), ptr(delta.t), delta.cs,
                                            ptr(coords), ptr(flow4), ptr(flow_cat), ld_cat, stream_ptr()),
          "woft_flow_head_gather")


def pair_ok(a, b):
    """True when woft_conv2d_pair takes the two layers in one launch (they select the same kernel instance)."""
    if a.precision == 0 or a.precision != b.precision or a.halo != b.halo:
        return False
    if a.halo == 16:                                          # (conv_1x1.hip: a kernel whose workgroups pick their layer's tile form)
        return not (a.stat_sum or b.stat_sum)
    if a.tile_n != b.tile_n:
        return False
    if a.halo == 0 and a.tile_m != b.tile_m:                  # (the pixel-tile kernels ignore tile_m)
        return False
    if a.stat_sum or b.stat_sum or a.in_norm or b.in_norm:
        return False
    if a.halo == 0:
        return True
    return a.halo in (8, 12) and (a.taps_y, a.taps_x) == (b.taps_y, b.taps_x) and a.taps_y * a.taps_x > 1


def run_conv_pair(a, b):
    check(_lib.load().woft_conv2d_pair(C.byref(a), C.byref(b), stream_ptr()), "woft_conv2d_pair")


def run_conv(p):
    check(_lib.load().woft_conv2d(C.byref(p), stream_ptr()), "woft_conv2d")


def conv2d(x, pc, c_out_stride=None, **kw):
    """Convenience: allocate the output and run.  Returns Act."""
    ho, wo = pc.out_hw(x.h, x.w)
    out = kw.pop("out", None) or new_act(x.n, ho, wo, pc.cout, cs=c_out_stride or _round_up(pc.cout, 4), zero=True)
    p = conv_params(x, pc, out, ho=ho, wo=wo, **kw)
    run_conv(p)
    return out


def inorm_ws(device="cuda"):
    """Scratch of woft_inorm_finalize (cross-workgroup partial sums + ticket), zeroed once; one per stream of calls."""
    return torch.zeros(int(_lib.load().woft_inorm_ws_bytes()), dtype=torch.uint8, device=device)


def inorm_finalize(stats, n_part, ld, channels, count, mean, rstd, eps=1e-5, channels_pad=None, ws=None):
    check(_lib.load().woft_inorm_finalize(ptr(stats[0]), ptr(stats[1]), n_part, ld, channels, channels_pad or channels,
                                          count, eps, ptr(mean), ptr(rstd), ptr(ws), stream_ptr()), "woft_inorm_finalize")


def inorm_apply(x, mean, rstd, out, mode, res=None, res_stats=None, res_mode=0):
    """res_mode 1 / 2: `res` is a raw conv output, normalised (2: + ReLU) with res_stats = (mean, rstd) inside the kernel."""
    assert out.cs == x.cs and (res is None or res.cs == x.cs)
    rm, rr = res_stats if res_stats is not None else (None, None)
    check(_lib.load().woft_inorm_apply(ptr(x.t), ptr(mean), ptr(rstd), ptr(res.t) if res is not None else None,
                                       ptr(rm), ptr(rr), res_mode, ptr(out.t), x.n_pix, x.cs, mode, stream_ptr()),
          "woft_inorm_apply")


class PyramidArgs:
    """Pointer tables of woft_feature_pyramid, built once (kept alive with the tensors they point to)."""
    def __init__(self, maps, splits, terms):
        import ctypes
        self.maps, self.splits, self.terms = maps, splits, terms
        self.levels = len(maps)
        assert 1 <= self.levels <= 4 and len(splits) == self.levels and all(m.cs == maps[0].cs == m.c for m in maps)
        for l in range(1, self.levels):
            assert (maps[l].h, maps[l].w) == (maps[l - 1].h // 2, maps[l - 1].w // 2)
        self.pooled = (ctypes.c_void_p * 3)(*[maps[l].t.data_ptr() if l < self.levels else None for l in range(1, 4)])
        self.split = (ctypes.c_void_p * 4)(*[splits[l].data_ptr() if l < self.levels else None for l in range(4)])


def feature_pyramid(a):
    """maps[0] -> pooled maps[1:] (avgpool2 chained) and every level's split correlation operand, one launch."""
    m = a.maps[0]
    check(_lib.load().woft_feature_pyramid(ptr(m.t), m.h, m.w, m.cs, a.levels, a.pooled, a.split, a.terms, stream_ptr()),
          "woft_feature_pyramid")


def preprocess(img_u8, out, hp, wp, pad_top, pad_left):
    h, w = img_u8.shape[:2]
    check(_lib.load().woft_preprocess_bgr_u8(ptr(img_u8), h, w, ptr(out.t), hp, wp, pad_top, pad_left, stream_ptr()),
          "woft_preprocess_bgr_u8")


def avgpool2(x, out):
    check(_lib.load().woft_avgpool2_nhwc(ptr(x.t), x.h, x.w, x.cs, ptr(out.t), stream_ptr()), "woft_avgpool2_nhwc")


def split_bf16(x, hi, lo=None):
    check(_lib.load().woft_split_bf16(ptr(x), x.numel(), ptr(hi), ptr(lo), stream_ptr()), "woft_split_bf16")


def conv3x3_narrow(x, pc, out, co_off=0):
    """3x3 / stride 1 / pad 1 conv with cout <= 2 in exact fp32 (woft_conv3x3_narrow); x, out: Act."""
    assert (pc.taps_y, pc.taps_x, pc.stride, pc.pad_y, pc.pad_x, pc.flat) == (3, 3, 1, 1, 1, 0) and pc.cout <= 2
    assert out.n == x.n and out.h == x.h and out.w == x.w
    check(_lib.load().woft_conv3x3_narrow(ptr(x.t), x.cs, x.n, x.h, x.w, pc.cin_pad, ptr(pc.wgt), ptr(pc.bias), pc.cout,
                                          ptr(out.t), out.cs, co_off, stream_ptr()), "woft_conv3x3_narrow")


def flow_head_update(x, pc, delta, coords, flow4=None, flow_cat=None, ld_cat=0):
    """conv3x3_narrow (2 channels -> delta) + coords_update in one launch (woft_flow_head_update)."""
    assert narrow_ok(x, pc) and pc.cout == 2 and x.n == 1 and coords.shape[0] == x.h * x.w
    check(_lib.load().woft_flow_head_update(ptr(x.t), x.cs, x.h, x.w, pc.cin_pad, ptr(pc.wgt), ptr(pc.bias), ptr(delta.t),
                                            delta.cs, ptr(coords), ptr(flow4), ptr(flow_cat), ld_cat, stream_ptr()),
          "woft_flow_head_update")


def narrow_ok(x, pc):
    """True when woft_conv3x3_narrow applies to this layer."""
    return ((pc.taps_y, pc.taps_x, pc.stride, pc.pad_y, pc.pad_x, pc.flat) == (3, 3, 1, 1, 1, 0) and pc.cout <= 2
            and pc.cin_pad in (128, 256) and x.cs >= pc.cin_pad)


def make_lookup_otf_params(f1s, f2s, dims, hf, wf, k, coords, out, radius, terms):
    """Volume-free lookup (woft_corr_lookup_otf): f1s / f2s[l] split feature rows (row-major), dims[l] = (h, w)."""
    p = LookupOtfParams()
    p.f1 = ptr(f1s)
    for l, (t, (h, w)) in enumerate(zip(f2s, dims)):
        p.f2[l], p.h[l], p.w[l] = ptr(t), h, w
    p.levels, p.radius, p.terms = len(f2s), radius, terms
    p.hf, p.wf, p.k = hf, wf, k
    p.alpha = 1.0 / math.sqrt(float(k))
    p.coords, p.out, p.ldo = ptr(coords), ptr(out), out.shape[1]
    p._keep = (f1s, f2s, coords, out)
    return p


def run_lookup_otf(p):
    check(_lib.load().woft_corr_lookup_otf(C.byref(p), stream_ptr()), "woft_corr_lookup_otf")


def split_bf16_lines(x, out):
    """x fp32 [rows][k] -> out bf16 [rows][2k]: per 32 values one 128-byte line [hi | lo] (woft_split_bf16_lines)."""
    check(_lib.load().woft_split_bf16_lines(ptr(x), x.numel(), ptr(out), stream_ptr()), "woft_split_bf16_lines")


def corr_gemm_bf16(a, b, m, n, alpha, out, terms):
    """out[:m, :n] = alpha * A B^T on pre-split bf16 operands (rows padded to 128); see woft_corr_gemm_bf16.
    out: float32, or bfloat16 for the bf16-storage volume."""
    k = a.shape[1] // 2 if terms == 3 else a.shape[1]
    assert out.dtype in (torch.float32, torch.bfloat16)
    check(_lib.load().woft_corr_gemm_bf16(ptr(a), ptr(b), m, n, a.shape[0], b.shape[0], k, float(alpha), ptr(out),
                                          out.shape[1], terms, int(out.dtype == torch.bfloat16), stream_ptr()),
          "woft_corr_gemm_bf16")


def corr_volume(f1, f2_rows, n_q, out, alpha, precision=0, f2_hi=None, f2_lo=None):
    """out[p][q] = alpha * <f1[p], f2_rows[q]>, q < n_q (rows of f2 already in the order the volume wants).
    f1: Act (P, C); f2_rows: tensor (rows_pad, C) zero padded to a multiple of 128 rows;
    f2_hi / f2_lo: its bf16 split (ops.split_bf16) for the bf16 precisions."""
    p = ConvParams()
    p.in0, p.cs0, p.in1, p.cs1, p.c_split = ptr(f1.t), f1.cs, None, 0, f1.cs
    p.n_img, p.h, p.w, p.ho, p.wo = 1, f1.h, f1.w, f1.h, f1.w
    p.taps_y = p.taps_x = p.stride = 1
    p.pad_y = p.pad_x = 0
    p.cin_pad, p.flat = f1.cs, 0
    p.wgt, p.bias, p.alpha = ptr(f2_rows), None, alpha
    p.wgt_hi, p.wgt_lo, p.precision = ptr(f2_hi), ptr(f2_lo), PRECISION.get(precision, precision)
    p.cout, p.cout_pad = n_q, f2_rows.shape[0]
    p.out, p.ldo, p.co_off = ptr(out), out.shape[1], 0
    p.out_w, p.out_pitch = 0, 0
    p.epi = _lib.EPI_LINEAR
    p.tile_m, p.tile_n = (128, 128) if f2_rows.shape[0] % 128 == 0 else (128, 64)
    p._keep = (f1, f2_rows, out, f2_hi, f2_lo)
    return p


def tiled_dims(h, w, tw=4):
    """(tile rows, tile cols, elements per plane) of an h x w map in the volume layout of 4 x 4 tiles."""
    ht, wt = (h + 3) // 4, (w + tw - 1) // tw
    return ht, wt, ht * wt * 4 * tw


def tile_rows(x, out):
    """x: Act (1, h, w, c) -> out rows in 4x4-tile order (first ht*wt*16 rows of `out`)."""
    check(_lib.load().woft_tile_rows(ptr(x.t), x.h, x.w, x.cs, ptr(out), stream_ptr()), "woft_tile_rows")


def tile_planes(planes, tw=4):
    """(P, h, w) tensor of per-source-pixel planes -> (P, ht*wt*4*tw) in the tiled layout (test helper)."""
    P, h, w = planes.shape
    ht, wt, n = tiled_dims(h, w, tw)
    pad = torch.zeros(P, ht * 4, wt * tw, dtype=planes.dtype, device=planes.device)
    pad[:, :h, :w] = planes
    return pad.reshape(P, ht, 4, wt, tw).permute(0, 1, 3, 2, 4).reshape(P, n).contiguous()


def untile_planes(vol, h, w, tw=4):
    """inverse of tile_planes: (P, >= ht*wt*4*tw) -> (P, h, w)."""
    ht, wt, n = tiled_dims(h, w, tw)
    P = vol.shape[0]
    return vol[:, :n].reshape(P, ht, wt, 4, tw).permute(0, 1, 3, 2, 4).reshape(P, ht * 4, wt * tw)[:, :h, :w]


def make_lookup_params(vols, dims, coords, out, radius):
    """vols[l]: (P, plane_l) tiled volumes (4 x 4 tiles), all float32 or all bfloat16; dims[l] = (H_l, W_l)."""
    p = LookupParams()
    assert len({v.dtype for v in vols}) == 1 and vols[0].dtype in (torch.float32, torch.bfloat16)
    p.vol_bf16 = int(vols[0].dtype == torch.bfloat16)
    for l, v in enumerate(vols):
        p.vol[l] = ptr(v)
        p.ht[l], p.wt[l], n = tiled_dims(*dims[l])
        assert v.shape[1] >= n
        p.plane[l] = v.shape[1]
    p.levels, p.radius = len(vols), radius
    p.coords, p.n_pix, p.out, p.ldo = ptr(coords), coords.shape[0], ptr(out), out.shape[1]
    p._keep = (vols, coords, out)
    return p


def run_lookup(p):
    check(_lib.load().woft_corr_lookup(C.byref(p), stream_ptr()), "woft_corr_lookup")


def coords_init(coords, hf, wf, flow4=None, flow_cat=None, ld_cat=0):
    check(_lib.load().woft_coords_init(ptr(coords), hf, wf, ptr(flow4), ptr(flow_cat), ld_cat, stream_ptr()),
          "woft_coords_init")


def coords_update(coords, delta, ld_delta, wf, flow4=None, flow_cat=None, ld_cat=0):
    check(_lib.load().woft_coords_update(ptr(coords), ptr(delta), ld_delta, wf, coords.shape[0], ptr(flow4),
                                         ptr(flow_cat), ld_cat, stream_ptr()), "woft_coords_update")


def wh_needed(pts, count, n_max, top, left, hf, wf, index, bitmap, dyn_index, n_needed=None):
    """dyn_index[j] = index[j] where the weight-head window of 1/8-res pixel index[j] is needed by one of the (count) points
    pts (n_max, 2) = (x, y) image coordinates, else -1 (woft_wh_needed)."""
    check(_lib.load().woft_wh_needed(ptr(pts), ptr(count), n_max, top, left, hf, wf, ptr(index), index.numel(), ptr(bitmap),
                                     ptr(dyn_index), ptr(n_needed), stream_ptr()), "woft_wh_needed")


def convex_upsample(coords, wlow, mask, hf, wf, crop, h, w, flow_up=None, dst=None, wout=None, do_sigmoid=False):
    check(_lib.load().woft_convex_upsample(ptr(coords), ptr(wlow), ptr(mask), mask.shape[1], hf, wf, crop[0], crop[1],
                                           h, w, ptr(flow_up), ptr(dst), ptr(wout), int(do_sigmoid), stream_ptr()),
          "woft_convex_upsample")


def convex_weights_at(pts, count, n_max, wlow, mask, hf, wf, crop, wsel, do_sigmoid=False):
    """wsel[i] = the weight convex_upsample would write at pixel pts[i] = (x, y), i < count (woft_convex_weights_at)."""
    check(_lib.load().woft_convex_weights_at(ptr(pts), ptr(count), n_max, ptr(wlow), ptr(mask), mask.shape[1], hf, wf,
                                             crop[0], crop[1], int(do_sigmoid), ptr(wsel), stream_ptr()),
          "woft_convex_weights_at")


def upflow8(coords, wlow, hf, wf, crop, h, w, flow_up=None, dst=None, wout=None, do_sigmoid=False):
    check(_lib.load().woft_upflow8(ptr(coords), ptr(wlow), hf, wf, crop[0], crop[1], h, w, ptr(flow_up), ptr(dst),
                                   ptr(wout), int(do_sigmoid), stream_ptr()), "woft_upflow8")


def warp_perspective_u8(img, Hmat, out=None, valid=None, nearest=False):
    """img: (H,W,C) or (H,W) uint8 CUDA tensor; Hmat: 3x3 numpy (src -> dst).  dst(x) = src(H^-1 x)."""
    h, w = img.shape[:2]
    c = 1 if img.dim() == 2 else img.shape[2]
    arr = (C.c_double * 9)(*np.linalg.inv(np.asarray(Hmat, dtype=np.float64)).ravel().tolist())
    check(_lib.load().woft_warp_perspective_u8(ptr(img), h, w, c, arr, ptr(out), ptr(valid), int(nearest),
                                               stream_ptr()), "woft_warp_perspective_u8")


def resize_by_factor_u8(img, factor):
    """cv2.resize(img, None, fx=1/factor, fy=1/factor) (INTER_LINEAR geometry) on the device."""
    h, w = img.shape[:2]
    c = 1 if img.dim() == 2 else img.shape[2]
    ho, wo = int(round(h / factor)), int(round(w / factor))
    out = torch.empty((ho, wo) if img.dim() == 2 else (ho, wo, c), dtype=torch.uint8, device=img.device)
    check(_lib.load().woft_resize_linear_u8(ptr(img), h, w, c, ptr(out), ho, wo, float(factor), float(factor),
                                            stream_ptr()), "woft_resize_linear_u8")
    return out


def tc_select_ws(n):
    return torch.empty(int(_lib.load().woft_tc_select_ws_bytes(n)), dtype=torch.uint8, device=DEV)


def tc_select(dst, w, tmask_u8, pwmask_u8, h, wimg, check_dst, sobol_u, ws, pa, pb, wout, count, grid=None):
    """Mask + compact + Sobol-subsample correspondences on the device (see csrc/select.hip).  (h, wimg): size of the
    masks (the frame); grid = (gh, gw): size of the flow grid when it differs (padding_mode 'crop')."""
    n_draw = 0 if sobol_u is None else sobol_u.numel()
    gh, gw = grid or (h, wimg)
    assert dst.numel() == 2 * gh * gw and tmask_u8.numel() == h * wimg and (w is None or w.numel() == gh * gw)
    check(_lib.load().woft_tc_select(ptr(dst), ptr(w), ptr(tmask_u8), ptr(pwmask_u8), gh, gw, h, wimg, int(check_dst),
                                     ptr(sobol_u), n_draw, ptr(ws), ptr(pa), ptr(pb), ptr(wout), pa.shape[0],
                                     ptr(count), stream_ptr()), "woft_tc_select")


def tc_flags(dst, tmask_u8, pwmask_u8, h, wimg, check_dst, grid=None, out=None):
    """The keep rule of tc_select alone -> bool tensor (gh*gw,) (woft_tc_flags)."""
    gh, gw = grid or (h, wimg)
    assert tmask_u8.numel() == h * wimg and (dst is None or dst.numel() == 2 * gh * gw)
    flags = out if out is not None else torch.empty(gh * gw, dtype=torch.uint8, device=tmask_u8.device)
    check(_lib.load().woft_tc_flags(ptr(dst), ptr(tmask_u8), ptr(pwmask_u8), gh, gw, h, wimg, int(check_dst),
                                    ptr(flags), stream_ptr()), "woft_tc_flags")
    return flags.view(torch.bool)


_HFIT_WS = {}


def hfit_ws(device=None):
    """Scratch of the streaming fit (woft_hfit_ws_bytes), one per device and stream, reused (stream-ordered)."""
    dev = torch.device(device or DEV)
    # (one per device AND stream: its use is ordered by the stream it was first used on -- two trackers on two streams of one
    #  process must not share it)
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device(), stream_ptr())
    if key not in _HFIT_WS:
        _HFIT_WS[key] = torch.empty(int(_lib.load().woft_hfit_ws_bytes()), dtype=torch.uint8, device=dev)
    return _HFIT_WS[key]


HFIT_SINGLE_MAX = 2048


def hfit(pa, pb, w, Hout, status, count=None, reweight=0, huber_k=1.0, n_irls=0, ws=None):
    n = pa.shape[0]
    if ws is None and n > HFIT_SINGLE_MAX:
        ws = hfit_ws(pa.device)
    check(_lib.load().woft_hfit(ptr(pa), ptr(pb), ptr(w), n, ptr(count), reweight, float(huber_k), n_irls,
                                ptr(ws), ptr(Hout), ptr(status), stream_ptr()), "woft_hfit")


def hfit_step(pa, pb, w, rew, first, res, Hout, status, ws=None):
    """One re-weighted solve with externally supplied row re-weights; residuals of its solution -> res."""
    ws = ws if ws is not None else hfit_ws(pa.device)
    check(_lib.load().woft_hfit_step(ptr(pa), ptr(pb), ptr(w), pa.shape[0], ptr(rew), int(bool(first)), ptr(ws),
                                     ptr(res), ptr(Hout), ptr(status), stream_ptr()), "woft_hfit_step")


def inlier_frac(pa, pb, Hm, frac, thr=5.0, count=None):
    check(_lib.load().woft_inlier_frac(ptr(pa), ptr(pb), pa.shape[0], ptr(count), ptr(Hm), float(thr), ptr(frac),
                                       stream_ptr()), "woft_inlier_frac")
