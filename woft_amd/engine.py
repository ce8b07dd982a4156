"""RAFT / WeightedRAFT inference engine on the HIP kernels (full model).

Host-side orchestration only: packs a reference-format state-dict once, plans every buffer and
every kernel-argument struct once per input resolution, then a frame is a fixed sequence of C-ABI
calls on torch's current stream -- no allocation, no host synchronisation inside.

Reference being replaced (paths under /root/reference/pytracking/external/RAFT/raft_core/):
  WeightedRAFT.forward  weighted_raft.py:179-315      RAFT.forward  raft.py:169-262
  BasicEncoder          extractor.py:118-192          CorrBlock     corr.py:11-69
  BasicUpdateBlock      update.py:114-136             WeightHead    weighted_raft.py:318-384

Results-identical restructurings: BatchNorm(eval) folded into the cnet convs; the
mask head evaluated only after the last iteration (test_mode consumes only that one,
weighted_raft.py:240-255); pyramid levels built as correlations against 2x2-pooled fmap2
(linearity; the reference's own AlternateCorrBlock does the same, corr.py:77-81); the weight
head's mean-response channel in algebraic form; template-side tensors (fmap1, net, inp) cached
when the caller pins the source image.
"""
import math

import torch

from . import _lib, ops
from .ops import Act, new_act

EPI = _lib


def _ru(x, m):
    return (x + m - 1) // m * m


class _Enc:
    """Packed weights of one BasicEncoder (extractor.py:118-165)."""

    def __init__(self, sd, p, norm):
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
                blk = dict(stride=s, conv1=get(q + ".conv1", s, bn=q + ".norm1"),
                           conv2=get(q + ".conv2", 1, bn=q + ".norm2"))
                if s != 1:
                    blk["down"] = get(q + ".downsample.0", s, bn=q + ".downsample.1")
                self.blocks.append(blk)
        w2, b2 = sd[p + ".conv2.weight"], sd[p + ".conv2.bias"]
        if norm == "batch":            # cnet: split into the GRU state (tanh) and the context (relu)
            self.conv2_net = ops.pack_conv(w2[:128], b2[:128])
            self.conv2_inp = ops.pack_conv(w2[128:], b2[128:])
        else:
            self.conv2 = ops.pack_conv(w2, b2)


class RaftEngine:
    def __init__(self, state_dict, small=False, weighted=True, precision="fp32"):
        """precision: "fp32" (exact fp32 MFMA), "bf16x3" (split-bf16, fp32-emulating) or "bf16"."""
        if precision not in ops.PRECISION:
            raise ValueError(f"precision must be one of {sorted(ops.PRECISION)}")
        self.precision = precision
        if small:
            raise NotImplementedError("the small model runs on woft_amd.engine_small")
        _lib.load()
        if not torch.cuda.is_available():
            raise _lib.WoftHipError("woft_amd needs a HIP device: there is no CPU fallback")
        sd = {k: v.detach().float().cpu() for k, v in state_dict.items()}
        self.weighted = weighted
        self.radius, self.levels = 4, 4
        self.fnet = _Enc(sd, "fnet", "instance")
        self.cnet = _Enc(sd, "cnet", "batch")
        u = "update_block."
        g = lambda n, **kw: ops.pack_conv(sd[u + n + ".weight"], sd[u + n + ".bias"], **kw)
        self.convc1 = g("encoder.convc1")
        self.convc2 = g("encoder.convc2")
        self.convf1 = g("encoder.convf1", flat_cs=4)
        self.convf2 = g("encoder.convf2")
        self.convm = g("encoder.conv")
        cat = lambda a, b, s: torch.cat([sd[u + a + s], sd[u + b + s]], 0)
        self.zr1 = ops.pack_conv(cat("gru.convz1", "gru.convr1", ".weight"), cat("gru.convz1", "gru.convr1", ".bias"),
                                 padding=(0, 2))
        self.q1 = g("gru.convq1", padding=(0, 2))
        self.zr2 = ops.pack_conv(cat("gru.convz2", "gru.convr2", ".weight"), cat("gru.convz2", "gru.convr2", ".bias"),
                                 padding=(2, 0))
        self.q2 = g("gru.convq2", padding=(2, 0))
        self.fh1 = g("flow_head.conv1")
        self.fh2 = g("flow_head.conv2")
        self.mk1 = g("mask.0")
        self.mk2 = g("mask.2", scale=0.25)             # ".25 * self.mask(net)"  update.py:135
        if weighted:
            w = "weight_head.net."
            self.wh0 = ops.pack_conv(sd[w + "0.weight"], sd[w + "0.bias"], flat_cs=8)
            self.wh2 = ops.pack_conv(sd[w + "2.weight"], sd[w + "2.bias"])
            self.wh4 = ops.pack_conv(sd[w + "4.weight"], sd[w + "4.bias"])
            self.wh6_w = sd[w + "6.weight"].reshape(-1).contiguous().cuda()
            self.wh6_b = float(sd[w + "6.bias"].item())
        self._plans = {}

    # ------------------------------------------------------------------------------------
    def plan(self, hp, wp):
        key = (hp, wp)
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
        self.lookup_events = None      # bench hook: list collecting (start, end) HIP events per lookup launch
        hf, wf = hp // 8, wp // 8
        self.hf, self.wf, self.P = hf, wf, hf * wf
        P = self.P
        dev = "cuda"
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)

        self.img = [new_act(1, hp, wp, 3, cs=4), new_act(1, hp, wp, 3, cs=4)]
        # feature maps: f1 (source) and f2 rows (target, zero padded to the GEMM N tile)
        self.f1 = new_act(1, hf, wf, 256, zero=True)
        self.dims, self.pitch, self.f2rows, self.f2act, self.vol = [], [], [], [], []
        self.f2hi, self.f2lo = [], []
        h, w = hf, wf
        for _ in range(eng.levels):
            self.dims.append((h, w))
            self.pitch.append(_ru(w, 4))
            rows = z(_ru(h * w, 128), 256)
            self.f2rows.append(rows)
            self.f2hi.append(torch.zeros_like(rows, dtype=torch.bfloat16))
            self.f2lo.append(torch.zeros_like(rows, dtype=torch.bfloat16))
            self.f2act.append(Act(rows[:h * w], 1, h, w, 256))
            self.vol.append(z(P, h * _ru(w, 4)))
            h, w = h // 2, w // 2
        # context
        self.net0 = new_act(1, hf, wf, 128, zero=True)
        self.xbuf = new_act(1, hf, wf, 256, zero=True)        # [inp 128 | motion 126 | flow 2]
        self._enc_scratch = {}
        self.stats = (z(2 * math.ceil((hp // 2) * (wp // 2) / 64) * 128), z(2 * math.ceil((hp // 2) * (wp // 2) / 64) * 128))
        self.mean, self.rstd = z(256), z(256)
        self.prog_f_src = self._fnet_program(self.img[0], self.f1)
        self.prog_f_dst = self._fnet_program(self.img[1], self.f2act[0])
        self.prog_c_src = self._cnet_program(self.img[0])
        self.prog_volume = self._volume_program()

        # update block
        self.coords = z(P, 2)
        self.corr = new_act(1, hf, wf, 324, cs=352, zero=True)
        self.c1 = new_act(1, hf, wf, 256, zero=True)
        self.cf = new_act(1, hf, wf, 256, zero=True)
        self.fl1 = new_act(1, hf, wf, 128, zero=True)
        self.flow4 = new_act(1, hf, wf, 2, cs=4, zero=True)
        self.zbuf = new_act(1, hf, wf, 128, zero=True)
        self.rh = new_act(1, hf, wf, 128, zero=True)
        self.hA = new_act(1, hf, wf, 128, zero=True)
        self.hB = new_act(1, hf, wf, 128, zero=True)
        self.fh = new_act(1, hf, wf, 256, zero=True)
        self.delta = new_act(1, hf, wf, 2, cs=4, zero=True)
        self.mk = new_act(1, hf, wf, 256, zero=True)
        self.mask = new_act(1, hf, wf, 576, zero=True)
        self.lookup = ops.make_lookup_params(self.vol, self.dims, self.pitch, self.coords, self.corr.t, eng.radius)
        self.prog_iter_first = self._iter_program(first=True)
        self.prog_iter = self._iter_program(first=False)
        cp = self._cp
        self.prog_mask = [cp(self.hB, eng.mk1, self.mk, epi=EPI.EPI_RELU), cp(self.mk, eng.mk2, self.mask)]
        if eng.weighted:
            self.x8 = new_act(P, 9, 9, 5, cs=8, zero=True)
            self.a1 = new_act(P, 9, 9, 128)
            self.a2 = new_act(P, 9, 9, 128)
            self.wmean = z(P)
            self.wlow = z(P)
            self.cs_ws = torch.zeros(256, 256, dtype=torch.float64, device=dev)
            self.cs_tot = torch.zeros(256, dtype=torch.float64, device=dev)
            self.prog_wh = [cp(self.x8, eng.wh0, self.a1, epi=EPI.EPI_RELU),
                            cp(self.a1, eng.wh2, self.a2, epi=EPI.EPI_RELU),
                            cp(self.a2, eng.wh4, self.a1, epi=EPI.EPI_RELU)]

    def _cp(self, *a, **kw):
        kw.setdefault("precision", self.prec)
        return ops.conv_params(*a, **kw)

    # ---- encoders ------------------------------------------------------------------------
    def _scratch(self, name, n, h, w, c):
        key = (name, h, w, c)
        if key not in self._enc_scratch:
            self._enc_scratch[key] = new_act(n, h, w, c)
        return self._enc_scratch[key]

    def _fnet_program(self, img, fmap_out):
        """InstanceNorm encoder: conv (+ partial statistics) -> finalize -> normalise/relu(/residual)."""
        e = self.eng.fnet
        prog = []

        def conv_norm(x, pc, name, mode, res=None):
            ho, wo = pc.out_hw(x.h, x.w)
            raw = self._scratch("raw_" + name, 1, ho, wo, pc.cout)
            p = self._cp(x, pc, raw, stats=self.stats)
            rows = 2 * math.ceil(p._m / p.tile_m)
            out = self._scratch("act_" + name, 1, ho, wo, pc.cout)
            prog.append(("conv", p))
            prog.append(("fin", (rows, pc.cout_pad, pc.cout, p._m)))
            prog.append(("apply", (raw, out, mode, res)))
            return out

        x = conv_norm(img, e.conv1, "c1", 1)
        for i, blk in enumerate(e.blocks):
            y = conv_norm(x, blk["conv1"], f"b{i}a", 1)
            res = x
            if blk["stride"] != 1:
                res = conv_norm(x, blk["down"], f"b{i}d", 0)
            x = conv_norm(y, blk["conv2"], f"b{i}b", 2, res=res)
        prog.append(("conv", self._cp(x, e.conv2, fmap_out)))
        return prog

    def _cnet_program(self, img):
        """BatchNorm(eval)-folded encoder: every norm/relu/residual lives in a conv epilogue."""
        e = self.eng.cnet
        prog = []

        def conv(x, pc, name, **kw):
            ho, wo = pc.out_hw(x.h, x.w)
            out = self._scratch("c_" + name, 1, ho, wo, pc.cout)
            prog.append(("conv", self._cp(x, pc, out, **kw)))
            return out

        x = conv(img, e.conv1, "c1", epi=EPI.EPI_RELU)
        for i, blk in enumerate(e.blocks):
            y = conv(x, blk["conv1"], f"b{i}a", epi=EPI.EPI_RELU)
            res = x if blk["stride"] == 1 else conv(x, blk["down"], f"b{i}d")
            x = conv(y, blk["conv2"], f"b{i}b", epi=EPI.EPI_RELU_RES_RELU, e0=res)
        prog.append(("conv", self._cp(x, e.conv2_net, self.net0, epi=EPI.EPI_TANH)))
        prog.append(("conv", self._cp(x, e.conv2_inp, self.xbuf, co_off=0, epi=EPI.EPI_RELU)))
        return prog

    def _volume_program(self):
        prog = []
        for l in range(self.eng.levels):
            if l > 0:
                prog.append(("pool", (self.f2act[l - 1], self.f2act[l])))
            h, w = self.dims[l]
            if self.prec != "fp32":
                prog.append(("split", (self.f2rows[l], self.f2hi[l], self.f2lo[l] if self.prec == "bf16x3" else None)))
            prog.append(("conv", ops.corr_volume(self.f1, self.f2rows[l], h * w, self.vol[l], w, self.pitch[l],
                                                 1.0 / math.sqrt(256.0), precision=self.prec,
                                                 f2_hi=self.f2hi[l], f2_lo=self.f2lo[l])))
        return prog

    # ---- one refinement iteration (update.py:127-136, weighted_raft.py:228-237) -----------
    def _iter_program(self, first):
        e, cp = self.eng, self._cp
        h_in = self.net0 if first else self.hB
        return [
            ("lookup", self.lookup),
            ("conv", cp(self.corr, e.convc1, self.c1, epi=EPI.EPI_RELU)),
            ("conv", cp(self.c1, e.convc2, self.cf, co_off=0, epi=EPI.EPI_RELU)),
            ("conv", cp(self.flow4, e.convf1, self.fl1, epi=EPI.EPI_RELU)),
            ("conv", cp(self.fl1, e.convf2, self.cf, co_off=192, epi=EPI.EPI_RELU)),
            ("conv", cp(self.cf, e.convm, self.xbuf, co_off=128, epi=EPI.EPI_RELU)),
            ("conv", cp(h_in, e.zr1, self.zbuf, x2=self.xbuf, c_split=128, epi=EPI.EPI_GRU_ZR, split=128, e0=h_in,
                        out1=self.rh)),
            ("conv", cp(self.rh, e.q1, self.hA, x2=self.xbuf, c_split=128, epi=EPI.EPI_GRU_Q, e0=h_in, e1=self.zbuf)),
            ("conv", cp(self.hA, e.zr2, self.zbuf, x2=self.xbuf, c_split=128, epi=EPI.EPI_GRU_ZR, split=128,
                        e0=self.hA, out1=self.rh)),
            ("conv", cp(self.rh, e.q2, self.hB, x2=self.xbuf, c_split=128, epi=EPI.EPI_GRU_Q, e0=self.hA,
                        e1=self.zbuf)),
            ("conv", cp(self.hB, e.fh1, self.fh, epi=EPI.EPI_RELU)),
            ("conv", cp(self.fh, e.fh2, self.delta)),
            ("coords", None),
        ]

    # ---- execution ------------------------------------------------------------------------
    def run(self, prog):
        for kind, a in prog:
            if kind == "conv":
                ops.run_conv(a)
            elif kind == "fin":
                rows, ld, c, count = a
                ops.inorm_finalize(self.stats, rows, ld, c, count, self.mean, self.rstd)
            elif kind == "apply":
                raw, out, mode, res = a
                ops.inorm_apply(raw, self.mean, self.rstd, out, mode, res=res)
            elif kind == "pool":
                ops.avgpool2(a[0], a[1])
            elif kind == "split":
                ops.split_bf16(a[0], a[1], a[2])
            elif kind == "lookup":
                self._lookup(a)
            elif kind == "coords":
                ops.coords_update(self.coords, self.delta.t, self.delta.cs, self.wf, self.flow4.t,
                                  self.xbuf.t[:, 254:], self.xbuf.cs)
            else:
                raise ValueError(kind)

    def _lookup(self, params):
        if self.lookup_events is None:
            ops.run_lookup(params)
            return
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        ops.run_lookup(params)
        e.record()
        self.lookup_events.append((s, e))

    def load_image(self, slot, img_u8, pad_top, pad_left):
        ops.preprocess(img_u8, self.img[slot], self.hp, self.wp, pad_top, pad_left)

    def encode_source(self):
        """fmap1, net, inp of the source image in img[0] (cacheable across frames)."""
        self.run(self.prog_f_src)
        self.run(self.prog_c_src)

    def flow(self, iters, crop, h, w, flow_up=None, dst=None, wout=None, do_sigmoid=False, trace=None):
        """Target features -> volume -> `iters` refinements -> full-resolution outputs."""
        e = self.eng
        if iters < 1:
            raise ValueError("iters must be >= 1")
        self.run(self.prog_f_dst)
        self.run(self.prog_volume)
        ops.coords_init(self.coords, self.hf, self.wf, self.flow4.t, self.xbuf.t[:, 254:], self.xbuf.cs)
        for it in range(iters):
            self.run(self.prog_iter_first if it == 0 else self.prog_iter)
            if trace is not None:
                trace(self, it)
        for p in self.prog_mask:
            ops.run_conv(p)
        wlow = None
        if e.weighted:
            self._lookup(self.lookup)                                # final lookup, weighted_raft.py:266
            lib = _lib.load()
            _lib.check(lib.woft_colsum(_lib.ptr(self.f2act[0].t), self.P, 256, _lib.ptr(self.cs_ws), 256,
                                       _lib.ptr(self.cs_tot), _lib.stream_ptr()), "woft_colsum")
            _lib.check(lib.woft_wh_pack(_lib.ptr(self.corr.t), self.corr.cs, _lib.ptr(self.f1.t), 256,
                                        _lib.ptr(self.cs_tot), 1.0 / (16.0 * self.P), self.P, 9, _lib.ptr(self.wmean),
                                        _lib.ptr(self.x8.t), _lib.stream_ptr()), "woft_wh_pack")
            for p in self.prog_wh:
                ops.run_conv(p)
            _lib.check(lib.woft_wh_reduce(_lib.ptr(self.a1.t), 128, 81, _lib.ptr(e.wh6_w), e.wh6_b, self.P,
                                          _lib.ptr(self.wlow), _lib.stream_ptr()), "woft_wh_reduce")
            wlow = self.wlow
        ops.convex_upsample(self.coords, wlow, self.mask.t, self.hf, self.wf, crop, h, w, flow_up=flow_up, dst=dst,
                            wout=wout if e.weighted else None, do_sigmoid=do_sigmoid)
