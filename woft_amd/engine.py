"""RAFT / WeightedRAFT inference engine on the HIP kernels (full and small models).

Host-side orchestration only: packs a reference-format state-dict once, plans every buffer and
every kernel-argument struct once per input resolution, then a frame is a fixed sequence of C-ABI
calls on torch's current stream -- no allocation, no host synchronisation inside.

Reference being replaced (paths under /root/reference/pytracking/external/RAFT/raft_core/):
  WeightedRAFT.forward  weighted_raft.py:179-315      RAFT.forward  raft.py:169-262
  BasicEncoder / SmallEncoder  extractor.py:118-267   CorrBlock     corr.py:11-69
  BasicUpdateBlock / SmallUpdateBlock  update.py:99-136   WeightHead  weighted_raft.py:318-384

Results-identical restructurings: BatchNorm(eval) folded into the cnet convs; the
mask head evaluated only after the last iteration (test_mode consumes only that one,
weighted_raft.py:240-255); pyramid levels built as correlations against 2x2-pooled fmap2
(linearity; the reference's own AlternateCorrBlock does the same, corr.py:77-81); the weight
head's mean-response channel in algebraic form; template-side tensors (fmap1, net, inp) cached
when the caller pins the source image.
"""
import math
import os

import torch

from . import _lib, ops
from .ops import Act, new_act

EPI = _lib
# InstanceNorm encoders: conv1's output and the downsample branch stay raw for their consumers to normalise (0: materialised)
DEFER_NORM = os.environ.get("WOFT_DEFER_NORM", "1") != "0"
# flow head: the 3x3 -> 2-channel conv folded into the first conv's epilogue + a per-pixel gather (0: two convs, the
# second on the vector ALUs in exact fp32 -- woft_flow_head_update)
FUSE_FLOWHEAD = os.environ.get("WOFT_FUSE_FLOWHEAD", "1") != "0"
# motion encoder: the correlation branch (convc1 -> convc2) and the flow branch (convf1 -> convf2) are independent until
# `conv` joins them (update.py:89-97): first layers in one launch, second layers in one launch (woft_conv2d_pair)
PAIR_BRANCHES = os.environ.get("WOFT_PAIR", "1") != "0"
PYRAMID_ONE_LAUNCH = os.environ.get("WOFT_PYRAMID", "1") != "0"    # target pyramid (pool + split of all levels) in one launch
# the flow-head gather of iteration k runs inside the lookup launch of iteration k + 1 (volume-free lookup; the last
# iteration's as its own launch): one launch fewer per iteration, same operations in the same order (0: always its own launch)
FOLD_GATHER = os.environ.get("WOFT_FOLD_GATHER", "1") != "0"


def _ru(x, m):
    return (x + m - 1) // m * m


class _Spec:
    """Architecture constants of the two model sizes (weighted_raft.py:34-72, raft.py:33-65)."""

    def __init__(self, small):
        self.small = small
        if small:
            self.fdim, self.hdim, self.cdim, self.radius = 128, 96, 64, 3
            self.fnorm, self.cnorm = "instance", "none"
        else:
            self.fdim, self.hdim, self.cdim, self.radius = 256, 128, 128, 4
            self.fnorm, self.cnorm = "instance", "batch"
        self.levels = 4
        self.nwin = 2 * self.radius + 1
        self.corr_c = self.levels * self.nwin ** 2            # 324 / 196
        self.corr_cs = _ru(self.corr_c, 32)                   # 352 / 224
        self.xdim = _ru(self.cdim + (82 if small else 128), 32)   # GRU input buffer [inp | motion | flow]
        self.flow_off = self.cdim + (80 if small else 126)    # channel of the flow inside that buffer


class _Enc:
    """Packed weights of one BasicEncoder / SmallEncoder (extractor.py:118-267)."""

    def __init__(self, sd, p, norm, spec, out_split=None):
        self.norm = norm

        def get(name, stride=1, flat_cs=0, bn=None):
            w, b = sd[name + ".weight"], sd[name + ".bias"]
            if norm == "batch" and bn is not None:
                w, b = ops.fold_bn(w, b, sd[bn + ".weight"], sd[bn + ".bias"], sd[bn + ".running_mean"],
                                   sd[bn + ".running_var"])
            return ops.pack_conv(w, b, stride=stride, flat_cs=flat_cs)

        self.conv1 = get(p + ".conv1", 2, 4, p + ".norm1")
        self.blocks = []
        for li, stride in ((1, 1), (2, 2), (3, 2)):
            for bi in range(2):
                q = f"{p}.layer{li}.{bi}"
                s = stride if bi == 0 else 1
                if spec.small:      # BottleneckBlock, extractor.py:60-116: 1x1 -> 3x3(stride) -> 1x1
                    convs = [get(q + ".conv1", 1, bn=q + ".norm1"), get(q + ".conv2", s, bn=q + ".norm2"),
                             get(q + ".conv3", 1, bn=q + ".norm3")]
                else:               # ResidualBlock, extractor.py:6-56: 3x3(stride) -> 3x3
                    convs = [get(q + ".conv1", s, bn=q + ".norm1"), get(q + ".conv2", 1, bn=q + ".norm2")]
                blk = dict(stride=s, convs=convs)
                if s != 1:
                    blk["down"] = get(q + ".downsample.0", s, bn=q + ".downsample.1")
                self.blocks.append(blk)
        w2, b2 = sd[p + ".conv2.weight"], sd[p + ".conv2.bias"]
        if out_split:               # cnet: GRU state (tanh) and context (relu), weighted_raft.py:217-219
            self.conv2_net = ops.pack_conv(w2[:out_split], b2[:out_split])
            self.conv2_inp = ops.pack_conv(w2[out_split:], b2[out_split:])
        else:
            self.conv2 = ops.pack_conv(w2, b2)


class RaftEngine:
    def __init__(self, state_dict, small=False, weighted=True, precision="fp32", corr="volume", volume_storage=None):
        """precision: "fp32" (exact fp32 MFMA), "bf16x3" (split-bf16, fp32-emulating), "bf16", or "fp16" = the reference's
        `mixed_precision` scoping (weighted_raft.py:204-219,233-234,258-290: autocast around fnet, cnet and the update block
        only): those convolutions on fp16 operands with fp32 accumulation, the correlation, the weight head and both
        upsamplings in fp32-class arithmetic (bf16x3 here).
        corr: "volume" (all-pairs volume + pyramid in HBM, corr.py:13-69) or "otf" (volume-free lookup from the
        feature maps, the reference's alternate_corr idea, corr.py:72-100; every precision, exact fp32 included: bit-identical
        to the volume path of the same precision).
        volume_storage: element type of the volume in HBM, "fp32" or "bf16" (fp32 accumulators rounded once at the GEMM's
        store; lookup interpolation and output stay fp32).  Default: "bf16" in the plain-bf16 precision -- its operating
        point, half the store stream and the lookup's reads, SURVEY 8d -- else "fp32"."""
        if precision not in ops.PRECISION:
            raise ValueError(f"precision must be one of {sorted(ops.PRECISION)}")
        if corr not in ("volume", "otf"):
            raise ValueError("corr must be 'volume' or 'otf'")
        self.precision = precision
        # arithmetic of the correlation (volume GEMM / volume-free lookup) and of the weight head's convolutions
        # ("f16mx8": two matrix-pipe passes per product on the register-streamed conv kernel -- the update block; every other
        #  kernel, the correlation and the weight head run in bf16x3)
        self.prec_corr = self.prec_wh = "bf16x3" if precision in ("fp16", "f16mx8") else precision
        self.corr = corr
        self.volume_storage = volume_storage or ("bf16" if precision == "bf16" else "fp32")
        if self.volume_storage not in ("fp32", "bf16") or (self.volume_storage == "bf16" and precision == "fp32"):
            raise ValueError("volume_storage: 'fp32', or 'bf16' with the split-bf16 precisions (their GEMM packs it)")
        _lib.load()
        if not torch.cuda.is_available():
            raise _lib.WoftHipError("woft_amd needs a HIP device: there is no CPU fallback")
        sd = {k: v.detach().float().cpu() for k, v in state_dict.items()}
        self.weighted = weighted
        self.spec = sp = _Spec(small)
        self.small = small
        self.radius, self.levels = sp.radius, sp.levels
        self.fnet = _Enc(sd, "fnet", sp.fnorm, sp)
        self.cnet = _Enc(sd, "cnet", sp.cnorm, sp, out_split=sp.hdim)
        u = "update_block."
        g = lambda n, **kw: ops.pack_conv(sd[u + n + ".weight"], sd[u + n + ".bias"], **kw)
        cat = lambda a, b, s: torch.cat([sd[u + a + s], sd[u + b + s]], 0)
        self.convc1 = g("encoder.convc1")
        self.convf1 = g("encoder.convf1", flat_cs=4)
        self.convf2 = g("encoder.convf2")
        self.convm = g("encoder.conv")
        self.fh1 = g("flow_head.conv1")
        self.fh2 = g("flow_head.conv2")
        # second conv of the flow head as MFMA fragments: folded into the first conv's epilogue (WOFT_EPI_FLOWHEAD)
        self.fh2_frag = (ops.pack_flowhead_frags(sd[u + "flow_head.conv2.weight"], 2 if precision in ("bf16x3", "f16mx8") else 1,
                                                 f16=precision == "fp16")
                         if precision != "fp32" and FUSE_FLOWHEAD else None)
        if small:
            self.zr = [ops.pack_conv(cat("gru.convz", "gru.convr", ".weight"), cat("gru.convz", "gru.convr", ".bias"))]
            self.q = [g("gru.convq")]
        else:
            self.convc2 = g("encoder.convc2")
            self.zr = [ops.pack_conv(cat("gru.convz1", "gru.convr1", ".weight"),
                                     cat("gru.convz1", "gru.convr1", ".bias"), padding=(0, 2)),
                       ops.pack_conv(cat("gru.convz2", "gru.convr2", ".weight"),
                                     cat("gru.convz2", "gru.convr2", ".bias"), padding=(2, 0))]
            self.q = [g("gru.convq1", padding=(0, 2)), g("gru.convq2", padding=(2, 0))]
            self.mk1 = g("mask.0")
            self.mk2 = g("mask.2", scale=0.25)         # ".25 * self.mask(net)"  update.py:135
        # The GRU convs see [h | inp | motion] (update.py:22-31, 45-60, 127-131) and `inp` (the context features) is the
        # same in every iteration: conv(W, [h, inp, motion]) = conv(W_[h,motion], [h, motion]) + conv(W_inp, inp).
        # The second term (+ bias) is computed once per source image (gate_inp) and enters the iterations as a
        # per-pixel bias; the per-iteration convs (gate_dyn) run without the context channels.
        hd_, cd_, mot = sp.hdim, sp.cdim, (82 if small else 128)
        dyn = [(0, hd_, 0), (hd_ + cd_, hd_ + cd_ + mot, hd_)]              # h -> 0.., motion(+flow) -> hd..
        ctx = [(hd_, hd_ + cd_, 0)]
        if small:
            zr_w = [(cat("gru.convz", "gru.convr", ".weight"), cat("gru.convz", "gru.convr", ".bias"), None)]
            q_w = [(sd[u + "gru.convq.weight"], sd[u + "gru.convq.bias"], None)]
        else:
            zr_w = [(cat("gru.convz1", "gru.convr1", ".weight"), cat("gru.convz1", "gru.convr1", ".bias"), (0, 2)),
                    (cat("gru.convz2", "gru.convr2", ".weight"), cat("gru.convz2", "gru.convr2", ".bias"), (2, 0))]
            q_w = [(sd[u + "gru.convq1.weight"], sd[u + "gru.convq1.bias"], (0, 2)),
                   (sd[u + "gru.convq2.weight"], sd[u + "gru.convq2.bias"], (2, 0))]
        self.zr_dyn = [ops.pack_conv(w_, None, padding=pd, cin_layout=dyn) for w_, _, pd in zr_w]
        self.zr_inp = [ops.pack_conv(w_, b_, padding=pd, cin_layout=ctx) for w_, b_, pd in zr_w]
        self.q_dyn = [ops.pack_conv(w_, None, padding=pd, cin_layout=dyn) for w_, _, pd in q_w]
        self.q_inp = [ops.pack_conv(w_, b_, padding=pd, cin_layout=ctx) for w_, b_, pd in q_w]
        if weighted:
            w = "weight_head.net."
            # class_params.weight_head_structure (weighted_raft.py:318-345) = the state-dict's layers net.0, net.2, ..., a ReLU after
            # each, then the closing 1x1 conv.  The shipped configs' [(128, 3)] * 3 (optical_flow/configs/v2_SNOB_large_g05_RAFT.py:16)
            # has its own kernels (first layer fused into the second's launch, mean fused into the third's epilogue, evaluated on
            # a subset of the windows); any other structure runs layer by layer on every window (wh_std = False).
            idx = sorted(int(k.split(".")[2]) for k in sd if k.startswith(w) and k.endswith(".weight"))
            shapes = [tuple(sd[f"{w}{i}.weight"].shape) for i in idx]
            if len(idx) < 2 or shapes[-1][0] != 1 or shapes[-1][2:] != (1, 1) or shapes[0][1] != sp.levels + 1 \
                    or any(a[0] != b[1] for a, b in zip(shapes, shapes[1:])) or any(sh[2] != sh[3] or sh[2] % 2 == 0 for sh in shapes):
                raise ValueError(f"weight head layers {shapes}: not a WeightHead (weighted_raft.py:318-345)")
            self.wh_std = shapes == [(128, 5, 3, 3), (128, 128, 3, 3), (128, 128, 3, 3), (1, 128, 1, 1)]
            last = idx[-1]
            if self.wh_std:
                self.wh0 = ops.pack_conv(sd[w + "0.weight"], sd[w + "0.bias"], flat_cs=8)
                self.wh2 = ops.pack_conv(sd[w + "2.weight"], sd[w + "2.bias"])
                self.wh4 = ops.pack_conv(sd[w + "4.weight"], sd[w + "4.bias"])
                # first conv as MFMA fragments for the fused two-layer launch (split-bf16 precisions, 9x9 windows)
                self.wh0_frag = (ops.pack_wh0_frags(sd[w + "0.weight"], 2 if self.prec_wh == "bf16x3" else 1)
                                 if precision != "fp32" and self.spec.nwin == 9 else None)
            else:
                # generic head: the first layer reads the 5-channel patches -- "flat" packing (8 floats per window position)
                # while kernel * 8 <= 32, else 32-channel rows; the other layers are ordinary convs on (windows, n, n, C)
                k0 = shapes[0][2]
                self.wh_flat0 = k0 * 8 <= 32
                self.wh_layers = [ops.pack_conv(sd[f"{w}{idx[0]}.weight"], sd[f"{w}{idx[0]}.bias"], flat_cs=8 if self.wh_flat0 else 0)]
                self.wh_layers += [ops.pack_conv(sd[f"{w}{i}.weight"], sd[f"{w}{i}.bias"]) for i in idx[1:-1]]
                self.wh0_frag = None
            self.wh6_c = shapes[-1][1]
            self.wh6_w = torch.zeros(_ru(self.wh6_c, 4))
            self.wh6_w[:self.wh6_c] = sd[f"{w}{last}.weight"].reshape(-1)
            self.wh6_w = self.wh6_w.contiguous().cuda()
            self.wh6_b = float(sd[f"{w}{last}.bias"].item())
        self._plans = {}

    def plan(self, hp, wp, slot=0):
        """Buffers + launch programs for one padded input size.  slot: an independent second set for the same size (the
        flow provider keeps a pinned source image -- the tracker's template -- resident in slot 0 and runs flows from other
        sources, the tracker's frame t-1 -> t flow of a lost frame, in slot 1: 5 GB more at 1080p, of 288)."""
        key = (hp, wp) if slot == 0 else (hp, wp, slot)
        if key not in self._plans:
            self._plans[key] = _Plan(self, hp, wp)
        return self._plans[key]


class _Plan:
    """Buffers + launch programs for one padded input size (hp, wp), both multiples of 8."""

    def __init__(self, eng, hp, wp):
        assert hp % 8 == 0 and wp % 8 == 0
        self.eng, self.hp, self.wp = eng, hp, wp
        self.prec = eng.precision
        self.source_tag = None
        self.lookup_events = None
        self.wh_events = None      # bench hook: list collecting (start, end) HIP events per lookup launch
        self.conv_events = None    # bench hook: {tag: [(start, end)]} for the tagged conv launches of the iteration program
        sp = eng.spec
        hf, wf = hp // 8, wp // 8
        self.hf, self.wf, self.P = hf, wf, hf * wf
        P = self.P
        dev = "cuda"
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        cp = self._cp

        self.img = [new_act(1, hp, wp, 3, cs=4), new_act(1, hp, wp, 3, cs=4)]
        # feature maps: f1 (source) and f2 rows (target, zero padded to the GEMM N tile)
        # source features: rows padded to the 128-row GEMM tile (pad rows stay zero), + their bf16 hi/lo planes
        self.f1rows = z(_ru(P, 128), sp.fdim)
        self.f1 = Act(self.f1rows[:P], 1, hf, wf, sp.fdim)
        self.prec_corr, self.prec_wh = eng.prec_corr, eng.prec_wh
        x3 = self.prec_corr == "bf16x3"
        bf = lambda rows: (torch.zeros(rows.shape[0], rows.shape[1] * (2 if x3 else 1), dtype=torch.bfloat16, device=dev)
                           if self.prec != "fp32" else None)       # GEMM operand: [hi|lo] lines / bf16 plane
        self.f1s = bf(self.f1rows)
        if eng.corr == "otf" and self.prec == "fp32":
            self.f1s = self.f1rows      # exact fp32 (terms = 0): the volume-free lookup reads the fp32 feature rows themselves
        # target feature pyramid: linear NHWC maps (f2act) and their rows in 4x4-tile order (f2rows, the B
        # operand of the correlation GEMM, zero padded to the N tile) -> volumes in the tiled layout
        # (corr = "otf": no volume; the lookup reads the row-major split maps f2s directly)
        self.otf = eng.corr == "otf"
        self.dims, self.f2rows, self.f2act, self.vol = [], [], [], []
        self.f2s = []
        h, w = hf, wf
        for _ in range(sp.levels):
            self.dims.append((h, w))
            self.f2act.append(new_act(1, h, w, sp.fdim, zero=True))
            if self.otf:
                self.f2s.append(bf(self.f2act[-1].t) if self.prec != "fp32" else self.f2act[-1].t)
            else:
                n = ops.tiled_dims(h, w)[2]
                rows = z(_ru(n, 128), sp.fdim)
                self.f2rows.append(rows)
                self.f2s.append(bf(rows))
                vdt = torch.bfloat16 if eng.volume_storage == "bf16" else torch.float32
                self.vol.append(torch.zeros(P, n, dtype=vdt, device=dev))
            h, w = h // 2, w // 2
        # context: GRU state and the GRU input buffer [inp | motion | flow | pad]
        self.net0 = new_act(1, hf, wf, sp.hdim, zero=True)
        self.xbuf = new_act(1, hf, wf, sp.xdim, zero=True)
        self._enc_scratch = {}
        n_stat = 2 * math.ceil((hp // 2) * (wp // 2) / 64) * 128
        self.stats = (z(n_stat), z(n_stat))
        self.fin_ws = ops.inorm_ws(dev)
        self._norm_slots = {}
        self.prog_f_src = self._encoder_program(eng.fnet, self.img[0], [(eng.fnet.conv2, self.f1, 0, EPI.EPI_LINEAR)])
        self.prog_f_dst = self._encoder_program(eng.fnet, self.img[1],
                                                [(eng.fnet.conv2, self.f2act[0], 0, EPI.EPI_LINEAR)])
        self.prog_c_src = self._encoder_program(eng.cnet, self.img[0],
                                                [(eng.cnet.conv2_net, self.net0, 0, EPI.EPI_TANH),
                                                 (eng.cnet.conv2_inp, self.xbuf, 0, EPI.EPI_RELU)])
        self.prog_volume = self._volume_program()

        # update block
        self.coords = z(P, 2)
        self.corr = new_act(1, hf, wf, sp.corr_c, cs=sp.corr_cs, zero=True)
        self.c1 = new_act(1, hf, wf, 96 if sp.small else 256, zero=True)
        self.cf = new_act(1, hf, wf, 128 if sp.small else 256, zero=True)     # [cor | flo]
        self.fl1 = new_act(1, hf, wf, 64 if sp.small else 128, zero=True)
        self.flow4 = new_act(1, hf, wf, 2, cs=4, zero=True)
        # per-pixel gate biases conv(W_inp, inp) + b: z|r and q of every GRU (half) step
        self.inp_c = new_act(1, hf, wf, sp.cdim, zero=True)
        self.gate_bias = [(new_act(1, hf, wf, 2 * sp.hdim, zero=True), new_act(1, hf, wf, sp.hdim, zero=True))
                          for _ in eng.zr_inp]
        self.prog_gate_bias = []
        for k in range(len(eng.zr_inp)):
            self.prog_gate_bias += [("conv", cp(self.inp_c, eng.zr_inp[k], self.gate_bias[k][0])),
                                    ("conv", cp(self.inp_c, eng.q_inp[k], self.gate_bias[k][1]))]
        self.zbuf = new_act(1, hf, wf, sp.hdim, zero=True)
        self.rh = new_act(1, hf, wf, sp.hdim, zero=True)
        self.hA = new_act(1, hf, wf, sp.hdim, zero=True)
        self.hB = new_act(1, hf, wf, sp.hdim, zero=True)
        self.fh = new_act(1, hf, wf, 128 if sp.small else 256, zero=True)
        self.fh_part = None        # flow head folded into one conv launch: per-pixel partial products of its second conv
        self.delta = new_act(1, hf, wf, 2, cs=4, zero=True)
        if self.otf:
            self.lookup = ops.make_lookup_otf_params(self.f1s, self.f2s, self.dims, hf, wf, sp.fdim, self.coords,
                                                     self.corr.t, sp.radius, 3 if x3 else (0 if self.prec == "fp32" else 1))
        else:
            self.lookup = ops.make_lookup_params(self.vol, self.dims, self.coords, self.corr.t, sp.radius)
        self.prog_iter_first = self._iter_program(first=True)
        self.prog_iter = self._iter_program(first=False)
        self.prog_mask = []
        if not sp.small:
            self.mk = new_act(1, hf, wf, 256, zero=True)
            self.mask = new_act(1, hf, wf, 576, zero=True)
            self.prog_mask = [cp(self.hB, eng.mk1, self.mk, epi=EPI.EPI_RELU),
                              cp(self.mk, eng.mk2, self.mask)]
            # last iteration: the flow head's conv and the mask head's first conv both read the final GRU state and are
            # independent (update.py:132-135) -> one launch when they select the same kernel instance (woft_conv2d_pair)
            self.prog_iter_last = None
            k = next((i for i, ent in enumerate(self.prog_iter) if len(ent) > 2 and ent[2] == "fh1"), None)
            if PAIR_BRANCHES and k is not None and ops.pair_ok(self.prog_iter[k][1], self.prog_mask[0]):
                self.prog_iter_last = (self.prog_iter[:k] + [("conv2", (self.prog_iter[k][1], self.prog_mask[0]), "fh1+mk1")]
                                       + self.prog_iter[k + 1:])
        self._fold = self._fold_gather_programs()
        if eng.weighted:
            n = sp.nwin
            self.x8 = new_act(P, n, n, 5, cs=8, zero=True)
            self.wmean = z(P)
            self.wlow = z(P)
            self.cs_ws = torch.zeros(256, sp.fdim, dtype=torch.float64, device=dev)
            self.cs_tot = torch.zeros(sp.fdim, dtype=torch.float64, device=dev)
            self.wh6_b = torch.tensor([eng.wh6_b], dtype=torch.float32, device=dev)
            self.wh_region = None                        # None = every source pixel
            self._wh_regions = {}
            self._wh_dyn = {}                            # per region: (dynamic window list, its programs, scratch)
            if eng.wh_std:
                self.a2 = new_act(P, n, n, 128)
                # first conv (5 -> 128): scalar-operand VALU kernel straight from the lookup buffer for the 7x7 / 9x9
                # windows (woft_wh_conv0), the generic conv on the packed x8 patches otherwise
                self.wh0_direct = n in (7, 9)
                self.wh0_fused = (eng.wh0_frag is not None and os.environ.get("WOFT_WH0_FUSED", "1") != "0"
                                  and cp(self.a2, eng.wh2, self.a2, epi=EPI.EPI_RELU, precision=self.prec_wh).halo == 2)
                # (with the first layer AND the tail fused into the two 128->128 launches only ONE activation exists)
                self.a1 = self.a2 if self.wh0_fused else new_act(P, n, n, 128)
                self.wh0_t = eng.wh0.wgt[:128].t().contiguous()          # [ky*32 + kx*8 + ci][co]
                self.prog_wh, self.wh_fused = self._wh_program(P, None)
                # the head restricted to a subset of the source pixels (set_weight_region): programs per region
            else:
                # any other weight_head_structure: layer by layer on every window, the closing 1x1 conv + window mean by
                # woft_wh_reduce (no window subsets, no fused layers: a correct path, not a tuned one)
                self.wh0_direct = self.wh0_fused = self.wh_fused = False
                cpw = lambda *a, **kw: self._cp(*a, precision=self.prec_wh, **kw)
                x = self.x8
                if not eng.wh_flat0:                     # first kernel wider than 3: 32-channel rows instead of the flat 8
                    self.x32 = new_act(P, n, n, 5, cs=32, zero=True)
                    x = self.x32
                # one activation per layer, zeroed once: a layer writes its cout channels only, so the pad channels that the next
                # layer's 32-channel K chunks (and woft_wh_reduce) read against zero weights stay exact zeros for ever -- a buffer
                # shared between layers of different widths would show a narrower layer what a wider one left behind (0 * inf = NaN)
                self.prog_wh = []
                for pc in eng.wh_layers:
                    out = new_act(P, n, n, pc.cout, cs=_ru(_ru(pc.cout, 4), 32), zero=True)
                    self.prog_wh.append(cpw(x, pc, out, epi=EPI.EPI_RELU))
                    x = out
                self.wh_last = x

    def _fold_gather_programs(self):
        """Variants of the iteration programs in which the flow-head gather that ends iteration k is done by the lookup
        launch that starts iteration k + 1 (woft_lookup_otf_params.fh_*): {id(program): variant, "gather": the last one}."""
        progs = [self.prog_iter_first, self.prog_iter] + ([self.prog_iter_last] if getattr(self, "prog_iter_last", None) else [])
        if not (FOLD_GATHER and self.otf and all(p and p[-1][0] == "fh_gather" and p[0][0] == "lookup" for p in progs)):
            return None
        n_planes, bias2 = self.prog_iter[-1][1]
        lk = type(self.lookup).from_buffer_copy(self.lookup)
        off = self.eng.spec.flow_off
        flow_cat = self.xbuf.t[:, off:]
        lk.fh_part, lk.fh_bias, lk.fh_delta = _lib.ptr(self.fh_part), _lib.ptr(bias2), _lib.ptr(self.delta.t)
        lk.fh_flow4, lk.fh_flow_cat = _lib.ptr(self.flow4.t), flow_cat.data_ptr()
        lk.fh_planes, lk.fh_ld, lk.fh_ld_delta, lk.fh_ld_cat = n_planes, self.fh_part.shape[1], self.delta.cs, self.xbuf.cs
        lk._keep = (self.lookup._keep, bias2, self.fh_part)
        out = {"gather": [self.prog_iter[-1]], "lookup": lk}
        for p in progs:
            head = p[0] if p is self.prog_iter_first else ("lookup", lk)
            out[id(p)] = [head] + p[1:-1]
        return out

    def _wh_program(self, n_win, index):
        """Launch list of the head's 128->128 layers on n_win windows (all source pixels, or those listed in the
        int32 tensor `index`) -> (program, fused): fused = the last layer runs on the whole-window kernel with
        ReLU + 1x1 conv + window mean in its epilogue."""
        eng, n = self.eng, self.eng.spec.nwin
        cp = lambda *a, **kw: self._cp(*a, precision=self.prec_wh, **kw)          # (fp32-class also in the fp16 mode)
        a1 = Act(self.a1.t[:n_win * n * n], n_win, n, n, 128)
        a2 = Act(self.a2.t[:n_win * n * n], n_win, n, n, 128)
        prog = ([] if self.wh0_direct else [cp(self.x8, eng.wh0, a1, epi=EPI.EPI_RELU)]) + [
            cp(a1, eng.wh2, a2, epi=EPI.EPI_RELU), cp(a2, eng.wh4, a1, epi=EPI.EPI_RELU)]
        if self.wh0_fused:          # layers 1 + 2 in one launch: the first activation (1.3 GB at 1080p) never exists
            assert prog[0].halo == 2
            prog[0] = cp(a1, eng.wh2, a2, epi=EPI.EPI_RELU,
                         wh0=(self.corr, self.wmean, eng.wh0_frag, eng.wh0.bias, index))
        last = prog[-1]
        fused = last.halo == 2
        if fused:
            last.epi = EPI.EPI_WH_MEAN
            last.e0, last.e1 = _lib.ptr(eng.wh6_w), _lib.ptr(self.wh6_b)
            last.out, last.ldo, last.co_off = _lib.ptr(self.wlow), 1, 0
            last.out_index = _lib.ptr(index) if index is not None else None
        return prog, fused

    def set_weight_region(self, index):
        """Evaluate the weight head only on the source pixels listed in `index` (int32 device tensor of 1/8-res
        pixel ids, or None for all): the other entries of the low-res weight map are zero.  Per-pixel results
        are unchanged (the head has no cross-pixel terms, weighted_raft.py:363-383); callers pass the pixels whose
        weights they consume (the tracker: its template mask, TRK:287-312, dilated by the upsampling support)."""
        if index is None or not (self.eng.weighted and self.wh0_direct and self.wh_fused):
            self.wh_region = None
            return
        key = index.data_ptr()
        if key not in self._wh_regions:
            prog, fused = self._wh_program(int(index.numel()), index)
            assert fused
            self._wh_regions[key] = (index, prog)
        self.wh_region = self._wh_regions[key]

    def _cp(self, *a, **kw):
        kw.setdefault("precision", self.prec)
        return ops.conv_params(*a, **kw)

    # ---- encoders ------------------------------------------------------------------------
    def _scratch(self, name, n, h, w, c):
        cs = _ru(c, 32)                       # every encoder activation is a conv input: whole K chunks
        key = (name, h, w, cs)
        if key not in self._enc_scratch:
            self._enc_scratch[key] = new_act(n, h, w, c, cs=cs, zero=True)
        return self._enc_scratch[key]

    def _encoder_program(self, e, img, outputs):
        """extractor.py:168-192 / 244-267.  InstanceNorm: conv (+ partial statistics) -> finalize ->
        normalise/relu(/residual) kernels.  BatchNorm(eval, folded) / no norm: everything in conv epilogues."""
        prog = []
        inorm = e.norm == "instance"
        tag = "i" if inorm else "c"

        def slot(name):
            """(mean, rstd) buffers of one normalised layer: a deferred normalisation may be consumed several launches later."""
            key = f"{tag}_{name}"
            if key not in self._norm_slots:
                self._norm_slots[key] = (torch.zeros(256, device="cuda"), torch.zeros(256, device="cuda"))
            return self._norm_slots[key]

        def materialise(x):
            """A pending activation ("raw", act, mode, stats[, cache]) -> its normalised tensor (apply kernel; once)."""
            if not isinstance(x, list):
                return x
            if x[4] is None:
                _, raw_x, mode, ms, _ = x
                x[4] = self._scratch(f"{tag}_mat_{len(prog)}", 1, raw_x.h, raw_x.w, raw_x.c)
                prog.append(("apply", (raw_x, x[4], mode - 1, None, ms, None, 0)))   # apply modes: 0 norm, 1 norm + relu
            return x[4]

        def layer(x, pc, name, relu, res=None, defer=False):
            """-> relu?(norm(conv(x)))  or, with res,  relu(res + relu(norm(conv(x)))).
            x (and res) may be a PENDING activation ["raw", act, mode, stats, materialised]: a raw conv output whose
            InstanceNorm (+ ReLU) is deferred to its consumers -- fused into this conv's LDS-halo loader when this conv runs
            on that kernel, into the residual operand of the block's last normalisation kernel, materialised by the apply
            kernel otherwise.  defer=True returns such a pending activation."""
            xin, in_norm, in_stats = x, 0, None
            if isinstance(x, list):
                _, raw_x, mode, ms, mat = x
                probe = self._cp(raw_x, pc, raw_x, in_norm=mode, in_stats=ms) if mat is None else None
                if probe is not None and probe.in_norm:
                    xin, in_norm, in_stats = raw_x, mode, ms
                else:
                    xin = materialise(x)
            ho, wo = pc.out_hw(xin.h, xin.w)
            out = self._scratch(f"{tag}_{name}", 1, ho, wo, pc.cout)
            kw = dict(in_norm=in_norm, in_stats=in_stats) if in_norm else {}
            if not inorm:
                epi = EPI.EPI_RELU_RES_RELU if res is not None else (EPI.EPI_RELU if relu else EPI.EPI_LINEAR)
                prog.append(("conv", self._cp(xin, pc, out, epi=epi, e0=materialise(res) if res is not None else None, **kw)))
                return out
            raw = self._scratch(f"{tag}_raw_{name}", 1, ho, wo, pc.cout)
            p = self._cp(xin, pc, raw, stats=self.stats, **kw)
            rows = 2 * p._m_tiles
            ms = slot(name)
            prog.append(("conv", p))
            prog.append(("fin", (rows, pc.cout_pad, pc.cout, raw.cs, p._m, ms)))
            if defer and res is None:
                return ["raw", raw, 2 if relu else 1, ms, None]
            if res is None:
                prog.append(("apply", (raw, out, 1 if relu else 0, None, ms, None, 0)))
            elif isinstance(res, list) and res[4] is None:      # shortcut still raw: normalised inside this kernel
                prog.append(("apply", (raw, out, 2, res[1], ms, res[3], res[2])))
            else:
                prog.append(("apply", (raw, out, 2, materialise(res), ms, None, 0)))
            return out

        # conv1's output and the 1x1 downsample branch stay raw where every consumer can normalise on the fly (the LDS-halo
        # conv of the first block, the residual operand of a block's closing kernel): three apply passes fewer per encoder
        x = layer(img, e.conv1, "c1", True, defer=inorm and DEFER_NORM)
        for i, blk in enumerate(e.blocks):
            res = x if blk["stride"] == 1 else layer(x, blk["down"], f"b{i}_d", False, defer=inorm and DEFER_NORM)
            y = x
            for k, pc in enumerate(blk["convs"][:-1]):
                # the block-internal activations have exactly one consumer (the next conv of the block): their
                # normalisation is deferred to it (the block output is materialised)
                y = layer(y, pc, f"b{i}_{k}", True, defer=inorm)
            x = layer(y, blk["convs"][-1], f"b{i}_o", True, res=res)
        x = materialise(x)
        for pc, out, co_off, epi in outputs:
            prog.append(("conv", self._cp(x, pc, out, co_off=co_off, epi=epi)))
        return prog

    def _volume_program(self):
        sp = self.eng.spec
        prog = []
        alpha = 1.0 / math.sqrt(float(sp.fdim))
        x3 = self.prec_corr == "bf16x3"
        if self.otf and self.prec != "fp32" and PYRAMID_ONE_LAUNCH and sp.fdim % 32 == 0 and self.f2act[0].cs == sp.fdim and sp.levels <= 4:
            # pooled maps and split operands of all levels in one launch (was 2 * levels - 1 launches)
            return [("pyramid", ops.PyramidArgs(self.f2act[:sp.levels], self.f2s[:sp.levels], 3 if x3 else 1))]
        for l in range(sp.levels):
            if l > 0:
                prog.append(("pool", (self.f2act[l - 1], self.f2act[l])))
            if self.otf:        # only the operands: pooled maps, split once (exact fp32: the maps themselves)
                if self.prec != "fp32":
                    prog.append(("split", (self.f2act[l].t, self.f2s[l])))
                continue
            prog.append(("tile", (self.f2act[l], self.f2rows[l])))
            if self.prec == "fp32":
                prog.append(("conv", ops.corr_volume(self.f1, self.f2rows[l], self.vol[l].shape[1], self.vol[l], alpha)))
            else:               # both operands pre-split once, GEMM fed by LDS-DMA (woft_corr_gemm_bf16)
                prog.append(("split", (self.f2rows[l], self.f2s[l])))
                prog.append(("cgemm", (self.f1s, self.f2s[l], self.P, self.vol[l].shape[1], alpha, self.vol[l],
                                       3 if x3 else 1)))
        return prog

    # ---- one refinement iteration (update.py:106-112,127-136; weighted_raft.py:228-237) ----
    def _iter_program(self, first):
        e, cp, sp = self.eng, self._cp, self.eng.spec
        hd = sp.hdim
        h_in = self.net0 if first else self.hB
        prog = [("lookup", self.lookup)]
        if sp.small:            # SmallMotionEncoder update.py:71-77: cor(96) | flo(32) -> 80, cat flow
            prog += [("conv", cp(self.corr, e.convc1, self.cf, co_off=0, epi=EPI.EPI_RELU)),
                     ("conv", cp(self.flow4, e.convf1, self.fl1, epi=EPI.EPI_RELU)),
                     ("conv", cp(self.fl1, e.convf2, self.cf, co_off=96, epi=EPI.EPI_RELU)),
                     ("conv", cp(self.cf, e.convm, self.xbuf, co_off=sp.cdim, epi=EPI.EPI_RELU))]
        else:                   # BasicMotionEncoder update.py:89-97: cor(192) | flo(64) -> 126, cat flow
            flo = [("conv", cp(self.flow4, e.convf1, self.fl1, epi=EPI.EPI_RELU), "convf1"),
                   ("conv", cp(self.fl1, e.convf2, self.cf, co_off=192, epi=EPI.EPI_RELU), "convf2")]
            cor = [("conv", cp(self.corr, e.convc1, self.c1, epi=EPI.EPI_RELU), "convc1"),
                   ("conv", cp(self.c1, e.convc2, self.cf, co_off=0, epi=EPI.EPI_RELU), "convc2")]
            if PAIR_BRANCHES:
                for c_, f_ in zip(cor, flo):             # (larger layer first: its workgroups are dispatched first)
                    if ops.pair_ok(c_[1], f_[1]):
                        prog.append(("conv2", (c_[1], f_[1]), c_[2] + "+" + f_[2]))
                    else:
                        prog += [c_, f_]
            else:
                prog += cor + flo
            prog.append(("conv", cp(self.cf, e.convm, self.xbuf, co_off=sp.cdim, epi=EPI.EPI_RELU), "convm"))
        # GRU half steps: z|r conv (sigmoid, r*h fused), q conv (tanh + state blend fused)
        states = [h_in, self.hA, self.hB] if len(e.zr) == 2 else [h_in, self.hB]
        if len(e.zr) == 1 and not first:
            states = [self.hB, self.hA]          # single-step GRU: ping-pong hB -> hA, copied back below
        for k, (zr, q) in enumerate(zip(e.zr, e.q)):
            hi, ho = states[k], states[k + 1]
            if self.gate_bias is not None:      # [h | motion] only; the inp term is the per-pixel bias (see RaftEngine)
                gz, gq = self.gate_bias[k]
                pzr = cp(hi, e.zr_dyn[k], self.zbuf, x2=self.xbuf, x2_off=sp.cdim, c_split=hd,
                         epi=EPI.EPI_GRU_ZR, split=hd, e0=hi, out1=self.rh, bias_map=gz)
                pq = cp(self.rh, e.q_dyn[k], ho, x2=self.xbuf, x2_off=sp.cdim, c_split=hd,
                        epi=EPI.EPI_GRU_Q, e0=hi, e1=self.zbuf, bias_map=gq)
                prog += [("conv", pzr, f"gru_zr{k}"), ("conv", pq, f"gru_q{k}")]
                continue
            prog += [("conv", cp(hi, zr, self.zbuf, x2=self.xbuf, c_split=hd, epi=EPI.EPI_GRU_ZR, split=hd, e0=hi,
                                 out1=self.rh)),
                     ("conv", cp(self.rh, q, ho, x2=self.xbuf, c_split=hd, epi=EPI.EPI_GRU_Q, e0=hi, e1=self.zbuf))]
        if len(e.zr) == 1 and not first:
            prog.append(("copy", (self.hA.t, self.hB.t)))
        fused = None
        if e.fh2_frag is not None and e.fh2.cout == 2:
            if self.fh_part is None:
                self.fh_part = torch.zeros(4 * self.P, 20, dtype=torch.float32, device="cuda")
            fused = ops.flowhead_params(self.hB, e.fh1, self.fh_part, e.fh2_frag, precision=self.prec)
        if fused is not None:                            # conv1 with conv2's partial products in its epilogue + the gather
            prog += [("conv", fused, "fh1"), ("fh_gather", (fused._n_planes, e.fh2.bias[:2].contiguous()))]
            return prog
        prog.append(("conv", cp(self.hB, e.fh1, self.fh, epi=EPI.EPI_RELU), "fh1"))
        if ops.narrow_ok(self.fh, e.fh2):                # second conv + coords1 += delta in one launch
            prog.append(("fh_update", (self.fh, e.fh2, self.delta)))
        else:
            prog += [("conv", cp(self.fh, e.fh2, self.delta)), ("coords", None)]
        return prog

    # ---- execution ------------------------------------------------------------------------
    def run(self, prog):
        for ent in prog:
            kind, a = ent[0], ent[1]
            if kind == "conv2":
                ev = self.conv_events
                if ev is not None and len(ent) > 2 and ent[2] in ev:
                    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    ops.run_conv_pair(*a)
                    t.record()
                    ev[ent[2]].append((s, t))
                else:
                    ops.run_conv_pair(*a)
            elif kind == "conv":
                ev = self.conv_events
                if ev is not None and len(ent) > 2 and ent[2] in ev:
                    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    ops.run_conv(a)
                    t.record()
                    ev[ent[2]].append((s, t))
                else:
                    ops.run_conv(a)
            elif kind == "fin":
                rows, ld, c, c_pad, count, ms = a
                ops.inorm_finalize(self.stats, rows, ld, c, count, ms[0], ms[1], channels_pad=c_pad, ws=self.fin_ws)
            elif kind == "apply":
                raw, out, mode, res, ms, res_ms, res_mode = a
                ops.inorm_apply(raw, ms[0], ms[1], out, mode, res=res, res_stats=res_ms, res_mode=res_mode)
            elif kind == "pyramid":
                ops.feature_pyramid(a)
            elif kind == "pool":
                ops.avgpool2(a[0], a[1])
            elif kind == "split":
                self._split(a[0], a[1])
            elif kind == "tile":
                ops.tile_rows(a[0], a[1])
            elif kind == "cgemm":
                ops.corr_gemm_bf16(*a)
            elif kind == "narrow":
                ops.conv3x3_narrow(*a)
            elif kind == "lookup":
                self._lookup(a)
            elif kind == "copy":
                a[1].copy_(a[0])
            elif kind == "fh_gather":
                off = self.eng.spec.flow_off
                ops.flow_head_gather(self.fh_part, a[0], self.hf, self.wf, a[1], self.delta, self.coords, self.flow4.t,
                                     self.xbuf.t[:, off:], self.xbuf.cs)
            elif kind == "fh_update":
                off = self.eng.spec.flow_off
                ops.flow_head_update(a[0], a[1], a[2], self.coords, self.flow4.t, self.xbuf.t[:, off:], self.xbuf.cs)
            elif kind == "coords":
                off = self.eng.spec.flow_off
                ops.coords_update(self.coords, self.delta.t, self.delta.cs, self.wf, self.flow4.t,
                                  self.xbuf.t[:, off:], self.xbuf.cs)
            else:
                raise ValueError(kind)

    def _lookup(self, params):
        run = ops.run_lookup_otf if self.otf else ops.run_lookup
        if self.lookup_events is None:
            run(params)
            return
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        run(params)
        e.record()
        self.lookup_events.append((s, e))

    def _split(self, rows, out):
        """fp32 rows -> the correlation GEMM's bf16 operand: [hi | lo] lines (bf16x3) or the bf16 plane (bf16)."""
        if self.prec_corr == "bf16x3":
            ops.split_bf16_lines(rows, out)
        else:
            ops.split_bf16(rows, out, None)

    def load_image(self, slot, img_u8, pad_top, pad_left):
        ops.preprocess(img_u8, self.img[slot], self.hp, self.wp, pad_top, pad_left)

    def encode_source(self, reuse_target=False):
        """fmap1, net, inp of the source image in img[0] (cacheable across frames).
        reuse_target: the source image IS the target image of this plan's previous flow() (consecutive lost frames: frame t was the
        target of the t-1 -> t flow and is the source of the t -> t+1 flow, TRK:181-184) -- its feature map is this plan's level-0
        target map: copied (fp32 map + correlation operand, two device copies) instead of a second fnet pass over the same image by
        the same launch program (bit-identical).  Volume-free correlation only (the volume mode keeps the target operand in
        4x4-tile order); -> whether the features were reused."""
        reused = bool(reuse_target) and self.otf and getattr(self, "target_valid", False)
        if reused:
            self.f1rows[:self.P].copy_(self.f2act[0].t.view(self.P, -1))
            if self.prec != "fp32":
                self.f1s[:self.P].copy_(self.f2s[0])
        else:
            self.run(self.prog_f_src)
        self.run(self.prog_c_src)
        if self.prec != "fp32" and not reused:
            self._split(self.f1rows, self.f1s)
        if self.gate_bias is not None:
            self.inp_c.t.copy_(self.xbuf.t[:, :self.eng.spec.cdim])
            self.run(self.prog_gate_bias)
        return reused

    def flow(self, iters, crop, h, w, flow_up=None, dst=None, wout=None, do_sigmoid=False, trace=None, defer_wh=False):
        """Target features -> volume -> `iters` refinements -> full-resolution outputs.
        defer_wh (full weighted model with a weight region set): stop before the weight head -- flow_up / dst are final,
        wout is NOT written -- and let finish_weights() evaluate the head where the caller then says it reads the weights."""
        e, sp = self.eng, self.eng.spec
        if iters < 1:
            raise ValueError("iters must be >= 1")
        self.run(self.prog_f_dst)
        self.run(self.prog_volume)
        self.target_valid = True                             # (level-0 target map + operand now belong to the image in img[1])
        off = sp.flow_off
        ops.coords_init(self.coords, self.hf, self.wf, self.flow4.t, self.xbuf.t[:, off:], self.xbuf.cs)
        last = getattr(self, "prog_iter_last", None) if iters > 1 else None
        fold = self._fold if trace is None else None        # (a trace reads the coordinates after every iteration)
        for it in range(iters):
            prog = self.prog_iter_first if it == 0 else (last if (last is not None and it == iters - 1) else self.prog_iter)
            if fold is not None:
                prog = fold[id(prog)]
            self.run(prog)
            if trace is not None:
                trace(self, it)
        if fold is not None:
            self.run(fold["gather"])
        for p in (self.prog_mask[1:] if last is not None else self.prog_mask):
            ops.run_conv(p)
        wlow = None
        if defer_wh:
            assert e.weighted and not sp.small and self.wh_region is not None
            ops.convex_upsample(self.coords, None, self.mask.t, self.hf, self.wf, crop, h, w, flow_up=flow_up, dst=dst,
                                wout=None, do_sigmoid=do_sigmoid)
            return
        if e.weighted:
            self._weight_head(self.wh_region[1] if self.wh_region is not None else self.prog_wh,
                              self.wh_region[0] if self.wh_region is not None else None)
            wlow = self.wlow
        wout = wout if e.weighted else None
        if sp.small:                                                 # no mask head: bilinear x8 (utils.py:82-84)
            ops.upflow8(self.coords, wlow, self.hf, self.wf, crop, h, w, flow_up=flow_up, dst=dst, wout=wout,
                        do_sigmoid=do_sigmoid)
        else:
            ops.convex_upsample(self.coords, wlow, self.mask.t, self.hf, self.wf, crop, h, w, flow_up=flow_up,
                                dst=dst, wout=wout, do_sigmoid=do_sigmoid)

    def finish_weights(self, pts, count, n_max, pad, crop, h, w, flow_up=None, dst=None, wout=None, do_sigmoid=False,
                       w_points=None):
        """After flow(defer_wh=True): the weight head on the windows of the current region that the (count) full-resolution
        pixels pts (n_max, 2) need (woft_wh_needed: the 3x3 upsampling support of their 1/8-res cells), then the upsampling
        with the weights.  wout is exact at those pixels (the head has no cross-pixel terms); elsewhere it is unspecified.
        w_points (n_max floats): receive the weights of the named pixels only, in their order, instead of the full map.
        pad = (top, left) of the padded image the 1/8-res grid belongs to."""
        index, _ = self.wh_region[:2]
        key = index.data_ptr()
        if key not in self._wh_dyn:
            dyn = torch.full_like(index, -1)
            prog, fused = self._wh_program(int(index.numel()), dyn)
            assert fused
            self._wh_dyn[key] = (dyn, prog, torch.zeros(self.P, dtype=torch.int32, device=index.device),
                                 torch.zeros(1, dtype=torch.int32, device=index.device))
        dyn, prog, bitmap, n_needed = self._wh_dyn[key]
        ops.wh_needed(pts, count, n_max, pad[0], pad[1], self.hf, self.wf, index, bitmap, dyn, n_needed)
        self._weight_head(prog, dyn, n_needed, need=bitmap)
        if w_points is not None:     # the weights of the named pixels only, in their order (no second full-resolution pass)
            ops.convex_weights_at(pts, count, n_max, self.wlow, self.mask.t, self.hf, self.wf, crop, w_points,
                                  do_sigmoid=do_sigmoid)
            return
        ops.convex_upsample(self.coords, self.wlow, self.mask.t, self.hf, self.wf, crop, h, w, flow_up=flow_up, dst=dst,
                            wout=wout, do_sigmoid=do_sigmoid)

    def _wh6_padded(self, cs):
        """The closing 1x1 conv's weights padded with zeros to the activation's channel stride (woft_wh_reduce walks whole rows)."""
        if getattr(self, "_wh6_pad", None) is None or self._wh6_pad.numel() != cs:
            self._wh6_pad = torch.zeros(cs, dtype=torch.float32, device="cuda")
            self._wh6_pad[:self.eng.wh6_c] = self.eng.wh6_w[:self.eng.wh6_c]
        return self._wh6_pad

    def _weight_head(self, prog_wh, index, n_needed=None, need=None):
        """Final lookup + the weight head (weighted_raft.py:266-272, 347-384) on all source pixels (index None) or on the
        windows listed in `index` -> self.wlow."""
        e, sp = self.eng, self.eng.spec
        if need is not None and self.otf:                        # final lookup, weighted_raft.py:266 -- only the 8x8
            self.lookup.need = _lib.ptr(need)                    # blocks that hold a wanted window (volume-free lookup)
            self._lookup(self.lookup)
            self.lookup.need = None
        else:
            self._lookup(self.lookup)
        lib = _lib.load()
        n = sp.nwin
        _lib.check(lib.woft_colsum(_lib.ptr(self.f2act[0].t), self.P, sp.fdim, _lib.ptr(self.cs_ws), 256,
                                   _lib.ptr(self.cs_tot), _lib.stream_ptr()), "woft_colsum")
        _lib.check(lib.woft_wh_pack(_lib.ptr(self.corr.t), self.corr.cs, _lib.ptr(self.f1.t), sp.fdim,
                                    _lib.ptr(self.cs_tot), 1.0 / (math.sqrt(float(sp.fdim)) * self.P), self.P, n,
                                    _lib.ptr(self.wmean), None if self.wh0_direct else _lib.ptr(self.x8.t),
                                    _lib.stream_ptr()), "woft_wh_pack")
        if not e.wh_std and not e.wh_flat0:
            self.x32.t[:, :8].copy_(self.x8.t)                   # (generic head, first kernel wider than 3)
        n_win = int(index.numel()) if index is not None else self.P
        if index is not None:
            self.wlow.zero_()                                    # pixels outside the region
        if self.wh0_direct and not self.wh0_fused:
            _lib.check(lib.woft_wh_conv0(_lib.ptr(self.corr.t), self.corr.cs, _lib.ptr(self.wmean), n_win, n,
                                         _lib.ptr(self.wh0_t), _lib.ptr(e.wh0.bias), _lib.ptr(self.a1.t),
                                         _lib.ptr(index) if index is not None else None,
                                         _lib.stream_ptr()), "woft_wh_conv0")
        for k, p in enumerate(prog_wh):
            if k == 0 and self.wh_events is not None:            # bench.py: HIP events around the first 128->128 layer
                s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                ops.run_conv(p)
                t.record()
                # (dynamic window list: the number of windows that really ran is on the device)
                self.wh_events.append((s, t, n_win if n_needed is None else n_needed.clone()))
            else:
                ops.run_conv(p)
        if not self.wh_fused:
            last = self.a1 if e.wh_std else self.wh_last
            _lib.check(lib.woft_wh_reduce(_lib.ptr(last.t), last.cs, n * n, _lib.ptr(self._wh6_padded(last.cs)), e.wh6_b, self.P,
                                          _lib.ptr(self.wlow), _lib.stream_ptr()), "woft_wh_reduce")
