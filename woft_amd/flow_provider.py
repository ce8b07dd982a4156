"""Flow-provider operator: the MI355X counterpart of the reference's RAFTWrapper
(/root/reference/pytracking/optical_flow/raft.py:29-218), same constructor config keys, same
`compute_flow` signature, return types and error behaviour.  `pytracking.optical_flow.raft`
(the compatibility shim) re-exports it under the reference's import path.
"""
import logging
import os
from pathlib import Path
from timeit import default_timer as timer

import numpy as np
import torch

from . import _lib, ops
from .engine import RaftEngine

logger = logging.getLogger(__name__)


def _pad_geometry(h, w, mode):
    """-> (hp, wp, pad_top, pad_left, out_h, out_w, load_h, load_w)"""
    if mode == "nopad":                                   # raft.py:221-226
        assert h % 8 == 0
        assert w % 8 == 0
        return h, w, 0, 0, h, w, h, w
    if mode == "crop":                                    # raft.py:235-247 (outputs keep the cropped size)
        ch, cw = (h // 8) * 8, (w // 8) * 8
        return ch, cw, 0, 0, ch, cw, ch, cw
    if mode == "RAFT":                                    # raft_core/utils/utils.py:7-26, 'sintel' mode
        ph = (((h // 8) + 1) * 8 - h) % 8
        pw = (((w // 8) + 1) * 8 - w) % 8
        return h + ph, w + pw, ph // 2, pw // 2, h, w, h, w
    if mode == "Michal":
        # the reference's MichalPadder.unpad(None) raises AttributeError for raft_type 'orig' / 'weighted'
        # (raft.py:148-150,264-265): the mode is unreachable in every shipped configuration
        raise NotImplementedError("padding_mode 'Michal' (raft.py:250-271) fails in the reference itself for "
                                  "raft_type 'orig'/'weighted'; it is not provided on the HIP path")
    raise ValueError(f"invalid padding_mode '{mode}'")


class RAFTWrapper:
    def __init__(self, config):
        self.C = config
        cp = config.class_params
        if cp.mask_estimation:
            raise NotImplementedError("mask_estimation (MaskHead) is unset in every shipped config")
        if self.C.raft_type not in ("orig", "weighted"):
            raise ValueError(f"Unknown RAFT type {self.C.raft_type}")
        logger.info(f"Loading weights from: {self.C.model}")

        def read(src):
            sd = src if isinstance(src, dict) else torch.load(src, map_location="cpu")
            return {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
        state_dict = read(self.C.model)
        if self.C.backbone_model:
            # raft.py:58-62 drops every fnet / cnet / update_block tensor of `model` ("will overwrite backbone later ...
            # weights pre-trained on standard RAFT"); the overwrite itself is not in the reference's wrapper (the backbone
            # stays at its random initialisation there).  Here the announced merge is carried out: the backbone tensors
            # come from the `backbone_model` checkpoint (a path or a state-dict), everything else (the weight head) from `model`.
            in_backbone = lambda k: ("fnet" in k) or ("cnet" in k) or ("update_block" in k)
            state_dict = {k: v for k, v in state_dict.items() if not in_backbone(k)}
            state_dict.update({k: v for k, v in read(self.C.backbone_model).items() if in_backbone(k)})
        weighted = self.C.raft_type == "weighted"
        small = bool(cp.small)
        # arithmetic of the convolutions / correlation products: "bf16x3" (split-bf16 operands, fp32 accumulation: fp32-EMULATING,
        # ~2^-16 relative per product; flow EPE <= 1e-4 px against the fp32 reference at 1080p, budget 1e-3 px), "fp32" (exact fp32
        # MFMA products, 4x slower), "bf16", "f16mx8", or "fp16".  `mixed_precision=True` selects "fp16" with the reference's scoping
        # (autocast = fp16 around fnet, cnet and the update block only, weighted_raft.py:204-219,233-234; correlation, weight head
        # and upsampling stay fp32-class).
        # A flow config WITHOUT the key (an unmodified reference config) gets "bf16x3" since round 5, not exact fp32: the reference
        # pins torch 1.8.1 (README.org:33), whose defaults on the GPUs of its time are torch.backends.cudnn.allow_tf32 = True AND
        # torch.backends.cuda.matmul.allow_tf32 = True (the PyTorch default from 1.7 to 1.11; the reference never touches either
        # flag) -- its own convolutions and its correlation matmul multiply 10-bit-mantissa TF32 operands on an A100.  bf16x3's
        # 16-bit products are 64x finer than that; exact fp32 stays one key away (precision = 'fp32' / WOFT_PRECISION=fp32) and is
        # timed in every bench line.
        for src, val in (("env WOFT_PRECISION", os.environ.get("WOFT_PRECISION")),
                         ("flow config key 'precision'", self.C.precision),
                         ("class_params.mixed_precision", "fp16" if cp.mixed_precision else None),
                         ("built-in default (no key in the flow config): fp32-emulating bf16x3 -- finer than the TF32 convolutions "
                          "the reference's torch 1.8.1 runs by default on Ampere; 'fp32' = exact products", "bf16x3")):
            if val:
                self.precision, self.precision_source = str(val), src
                break
        # What arithmetic the caller got, said out loud: SURVEY 8a reads "IEEE fp32 unless stated", so anything else is STATED here --
        # at WARNING level when nobody asked for it (the built-in default of a config without the key), at INFO level otherwise.
        note = (f"RAFT arithmetic: precision = '{self.precision}' (from: {self.precision_source}). "
                + ("This is the reference's arithmetic (exact IEEE fp32 products)." if self.precision == "fp32" else
                   "NOT the reference's IEEE-fp32 products: set precision = 'fp32' in the flow config (or WOFT_PRECISION=fp32) "
                   "for the reference's arithmetic (about 4x slower)."))
        logger.log(logging.WARNING if self.precision_source.startswith("built-in default") else logging.INFO, note)
        self.precision_note = note
        # correlation: "volume" (all-pairs volume + pyramid in HBM, corr.py:13-69) or "otf" (volume-free lookup, what
        # the reference's `alternate_corr` switch selects, corr.py:72-100).  Bit-identical results in every precision
        # (exact fp32 included: the lookup's fp32-MFMA instantiation), "otf" is faster and needs no P x P buffer: the default.
        self.corr = os.environ.get("WOFT_CORR") or getattr(self.C, "corr", None) or "otf"
        # flow config key `volume_storage` ("fp32" | "bf16"): element type of the correlation volume (corr = "volume")
        self.engine = RaftEngine(state_dict, small=small, weighted=weighted, precision=self.precision, corr=self.corr,
                                 volume_storage=getattr(self.C, "volume_storage", None))
        # opt-in (flow config key `graph`, env WOFT_GRAPH=1): the ~330 launches of a flow -- a static list per
        # resolution, fixed buffers, no allocation -- are captured once into a hipGraph and replayed per frame
        self.use_graph = (os.environ.get("WOFT_GRAPH") or str(int(bool(getattr(self.C, "graph", False))))) == "1"
        self._pinned = None
        self._pinned_key = None
        self._wmask, self._wregion, self._all_pixels = None, {}, {}
        self.weights_deferred, self._deferred = False, None
        self.defer_min_ratio = 6           # defer_weights: region windows per named pixel from which deferring pays
        self._out = {}
        self._cache_errors = set()
        self._last_dst = {}                # per buffer set: (id of the last dst_img object, padding geometry)
        self.source_features_reused = False

    def _run_flow(self, plan, iters, crop, oh, ow, o, weighted, do_sigmoid, defer_wh=False, want_flow=True):
        """plan.flow() eagerly, or -- use_graph -- as ONE hipGraph launch (captured at the second call with the same
        arguments; the per-launch event hooks of bench.py force the eager path)."""
        def eager():
            # (mode "TC" hands out dst = grid + flow only: the (2, H, W) flow map is then not written at all)
            plan.flow(iters, crop, oh, ow, flow_up=o["flow"] if want_flow else None, dst=o["dst"], wout=o["w"] if weighted else None,
                      do_sigmoid=do_sigmoid, defer_wh=defer_wh)
        if not self.use_graph or plan.lookup_events is not None or plan.wh_events is not None or plan.conv_events is not None:
            return eager()
        graphs = plan.__dict__.setdefault("_graphs", {})
        region = plan.wh_region
        key = (iters, crop, oh, ow, weighted, do_sigmoid, want_flow, bool(defer_wh), o["flow"].data_ptr(),
               region[0].data_ptr() if region is not None else 0)
        g = graphs.get(key)
        if g is None:
            eager()                                        # this call's results; also the warm-up the capture needs
            if key in graphs:                              # (second sighting: capture for the calls to come)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    eager()
            graphs[key] = g
        else:
            g.replay()

    def finish_weights(self, pts, count, n_max, out=None):
        """After compute_flow(..., defer_weights=True) with `weights_deferred` set: evaluate the weight head for the
        (count, a device int32; at most n_max) source pixels pts (n_max, 2) = (x, y) and return the (1, H*W) weight tensor
        (borrowed), exact at those pixels and unspecified elsewhere -- or, with out (n_max floats), just the weights of
        the named pixels, in their order.  The weights of a source pixel do not depend on the
        other pixels (weighted_raft.py:363-383), and the caller knows which correspondences it keeps before it needs
        their weights (TRK:287-312 + subsampler: decided by the flow alone)."""
        plan, crop, oh, ow, o, do_sigmoid = self._deferred
        self._deferred = None
        plan.finish_weights(pts, count, n_max, crop, crop, oh, ow, flow_up=o["flow"], dst=o["dst"], wout=o["w"],
                            do_sigmoid=do_sigmoid, w_points=out)
        return o["w"] if out is None else out

    # ---- template caching (results-identical: InstanceNorm is per sample, extractor.py:171-190) ----
    def pin_source(self, src_img):
        """Declare `src_img` (an ndarray the caller will not mutate) as a recurring source image:
        its feature/context tensors are computed once and reused by compute_flow(src_img, ...)."""
        self._pinned = src_img
        self._pinned_key = None
        self._wmask, self._wregion = None, {}

    def pin_weight_region(self, mask):
        """Declare that, for flows FROM the pinned source image, the caller consumes the flow weights only at the
        pixels where `mask` (bool ndarray, source-image size; None = everywhere) is set: the weight head is then
        evaluated on those 1/8-resolution pixels only (+ the 3x3 support of the x8 upsampling), the other weights
        are unspecified.  The weights at the masked pixels are unchanged (the head is per source pixel,
        weighted_raft.py:363-383).  Flows from other sources always get the full weight map."""
        self._wmask = None if mask is None else np.ascontiguousarray(np.asarray(mask) > 0)
        self._wregion = {}

    def _weight_region(self, key, hp, wp, top, left, oh, ow):
        if self._wmask is None:
            return None
        if key not in self._wregion:
            m = self._wmask[:oh, :ow]
            coarse = np.zeros((hp // 8, wp // 8), bool)
            ys, xs = np.nonzero(m)
            coarse[(ys + top) >> 3, (xs + left) >> 3] = True
            pad = np.pad(coarse, 1)
            dil = np.zeros_like(coarse)
            for dy in range(3):                            # convex / bilinear x8 upsampling reads the 3x3 neighbours
                for dx in range(3):
                    dil |= pad[dy:dy + coarse.shape[0], dx:dx + coarse.shape[1]]
            idx = np.flatnonzero(dil).astype(np.int32)
            self._wregion[key] = torch.from_numpy(idx).cuda() if idx.size and idx.size < dil.size else None
        return self._wregion[key]

    def postprocess_weights(self, flat_weights, fn):
        s = self.last_flow_shape
        weights = fn(flat_weights.reshape(s["batch"], 1, s["H"], s["W"]))
        return weights.reshape(s["batch"], s["H"] * s["W"])

    def _outputs(self, h, w):
        key = (h, w)
        if key not in self._out:
            z = lambda *s: torch.empty(*s, dtype=torch.float32, device="cuda")
            self._out[key] = dict(flow=z(2, h, w), dst=z(2, h * w), w=z(1, h * w),
                                  src=torch.stack([torch.arange(h * w, device="cuda") % w,
                                                   torch.div(torch.arange(h * w, device="cuda"), w, rounding_mode="floor")]))
        return self._out[key]

    def _cached_flow(self, src_img, identifier, mode, numpy_out, do_sigmoid, borrow=False):
        """utils/caching.py:53-59 wire format: <flow_cache_dir>/<dataset>/<sequence>/<i>-<i+1>.npz holding
        'half_flow' (2,H,W) and 'half_weights' (1,H,W) (any float dtype, cast to fp32); then the same
        post-processing as a computed flow (raft.py:152-195)."""
        dataset_name, seq_name, frame_i = identifier
        path = Path(self.C.flow_cache_dir) / dataset_name / seq_name / f"{frame_i}-{frame_i + 1}.npz"
        data = np.load(path, allow_pickle=False)           # plain float arrays: nothing to unpickle
        flow = np.ascontiguousarray(data["half_flow"].astype(np.float32))
        wts = data["half_weights"].astype(np.float32)
        wts = np.ascontiguousarray(wts) if wts.size > 1 else None
        h, w = flow.shape[1:]
        post = self.C.weights_postprocessing_fn if wts is not None else None
        o = self._outputs(h, w)
        o["flow"].copy_(torch.from_numpy(flow), non_blocking=True)
        wl = None
        if wts is not None:
            wl = torch.from_numpy(wts).cuda(non_blocking=True)
        lib = _lib.load()
        _lib.check(lib.woft_flow_to_tc(_lib.ptr(o["flow"]), _lib.ptr(wl), h, w, _lib.ptr(o["dst"]),
                                       _lib.ptr(o["w"]) if wl is not None else None, int(bool(do_sigmoid) and not post),
                                       _lib.stream_ptr()), "woft_flow_to_tc")
        if post:
            self._postprocess_logits(o, h, w, do_sigmoid)
        return self._deliver(o, o["w"] if wl is not None else None, mode, h, w, numpy_out, borrow)

    def _postprocess_logits(self, o, h, w, do_sigmoid):
        """raft.py:152-159: `weights_postprocessing_fn` maps the (1, 1, H, W) weight LOGITS, then the sigmoid (if asked)."""
        wts = self.C.weights_postprocessing_fn(o["w"].reshape(1, 1, h, w))
        wts = torch.as_tensor(wts, device=o["w"].device, dtype=torch.float32).reshape(1, h * w)
        o["w"].copy_(torch.sigmoid(wts) if do_sigmoid else wts)

    def _deliver(self, o, weights, mode, oh, ow, numpy_out, borrow):
        """The caller's view of the outputs.  The kernels write into per-resolution buffers that the NEXT call at the
        same size overwrites; like the reference, compute_flow hands out tensors of its own (device copies, a few
        microseconds) unless the caller passes borrow=True and consumes the results before its next call (the
        tracker does)."""
        own = (lambda t: t) if borrow else (lambda t: t.clone())
        host = lambda t: t.cpu().numpy()
        if mode == "flow":
            wts = weights.reshape(1, oh, ow) if weights is not None else None
            if numpy_out:
                return host(o["flow"]), (host(wts) if wts is not None else None)
            return own(o["flow"]), (own(wts) if wts is not None else None)
        self.last_flow_shape = {"batch": 1, "delta": 2, "H": oh, "W": ow}
        if numpy_out:
            return host(o["src"]), host(o["dst"]), (host(weights) if weights is not None else None)
        # (the int64 source grid is a constant of the resolution -- 33 MB at 1080p: the provider's own tensor under borrow=True
        #  (the tracker only indexes it); any other caller gets a tensor it may edit in place, as the reference's callers may --
        #  round-4 advisor finding: a shared grid that a caller offsets or sorts would corrupt every later call at this size)
        return own(o["src"]), own(o["dst"]), (own(weights) if weights is not None else None)

    def compute_flow(self, src_img, dst_img, mode="TC", vis=False, src_img_identifier=None,
                     numpy_out=False, do_sigmoid=False, borrow=False, defer_weights=False, weight_region=False,
                     src_is_previous_dst=False):
        """src_img / dst_img: (H, W, 3) uint8 BGR (numpy, or CUDA tensors already on the device).
        mode 'TC' -> (src_coords (2,HW) int64, dst_coords (2,HW) f32, weights (1,HW) f32 | None)
        mode 'flow' -> (flow (2,H,W), weights (1,H,W) | None).
        borrow (extension, default off): return the provider's own output buffers, valid until the next call.
        weight_region (extension, default off): the caller reads the weights only inside the region declared with
        pin_weight_region() (flows from the pinned source); without it every call returns the full weight map, as the
        reference does.
        defer_weights (extension, default off: 0; else the number of source pixels the caller will name; honoured for flows
        from the pinned source whose weight region is large against it, else ignored: check `weights_deferred` after the
        call): return the flow / correspondences with weights = None and evaluate the weight head later, in
        finish_weights(), only where the caller then says it reads the weights."""
        assert mode in ["flow", "TC"]
        assert src_img.shape == dst_img.shape
        if src_img_identifier is not None:                 # pre-computed flow (raft.py:92-109)
            try:
                return self._cached_flow(src_img, src_img_identifier, mode, numpy_out, do_sigmoid, borrow)
            except Exception as ex:   # no such file / array, object placeholders, truncated archives, wrong shapes / dtypes:
                # the reference falls back on ANY exception (raft.py:108-109): compute the flow
                key = (type(ex), str(ex))                  # (the reference logs each distinct error once)
                if key not in self._cache_errors:
                    self._cache_errors.add(key)
                    logger.warning(f"no cached flow: {ex}")
        H, W = src_img.shape[:2]
        hp, wp, top, left, oh, ow, lh, lw = _pad_geometry(H, W, self.C.padding_mode)
        # a flow from another source leaves the pinned source's tensors (fmap1, net, inp, gate biases) where they are:
        # it runs in a second buffer set (the reference's lost branch, TRK:181-184, alternates template and frame t-1)
        pinned_here = src_img is self._pinned
        plan = self.engine.plan(hp, wp, 0 if (pinned_here or self._pinned is None) else 1)
        start_time = timer()

        def up(a):
            if isinstance(a, torch.Tensor):
                t = a if a.is_cuda else a.cuda(non_blocking=True)
            else:
                t = torch.from_numpy(np.ascontiguousarray(a)).cuda(non_blocking=True)
            if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
                raise TypeError(f"compute_flow takes (H, W, 3) uint8 BGR images, got {tuple(t.shape)} {t.dtype}")
            if (lh, lw) != (H, W):
                t = t[:lh, :lw].contiguous()
            return t.contiguous()

        key = (hp, wp, top, left)
        if pinned_here and self._pinned_key == key and plan.source_tag is self:
            pass                                           # fmap1 / net / inp still resident in the plan
        else:
            s = up(src_img)
            plan.load_image(0, s, top, left)
            # src_is_previous_dst (extension, the shim's tracker on consecutive lost frames): the caller states that src_img is the
            # image it passed as dst_img in its previous call for this buffer set; honoured only when that is the very same object
            # and geometry -- then the source features are the previous call's target features (engine.encode_source)
            reuse = bool(src_is_previous_dst) and self._last_dst.get(id(plan)) == (id(src_img), key)
            self.source_features_reused = bool(plan.encode_source(reuse_target=reuse))
            plan.source_tag = self if pinned_here else None
            if pinned_here:
                self._pinned_key = key
        post = self.C.weights_postprocessing_fn or None     # (a map -> map callable may read any pixel: full map)
        region = None
        if weight_region and not post:
            if pinned_here:
                region = self._weight_region(key, hp, wp, top, left, oh, ow)
            if region is None and defer_weights:
                # no pinned region (another source, or no mask declared), but the caller will name the pixels whose weights
                # it reads (defer_weights): the window list is every source pixel, thinned on the device in finish_weights()
                if plan.P not in self._all_pixels:
                    self._all_pixels[plan.P] = torch.arange(plan.P, dtype=torch.int32, device="cuda")
                region = self._all_pixels[plan.P]
        plan.set_weight_region(region)
        d = up(dst_img)
        plan.load_image(1, d, top, left)
        self._last_dst[id(plan)] = (id(dst_img), key)         # (what this buffer set's target features will belong to after this call)
        o = self._outputs(oh, ow)
        weighted = self.C.raft_type == "weighted"
        # (defer_weights = the number of pixels the caller will name: worth it only if their 3x3 supports cannot cover most
        # of the region anyway -- measured with 500 pixels: +0.7 % at 720p (3 900 windows), +6 % at 1080p, +11 % at 4K)
        self.weights_deferred = bool(defer_weights and weighted and not self.engine.small and plan.wh_region is not None
                                     and mode == "TC" and not numpy_out
                                     and int(plan.wh_region[0].numel()) > self.defer_min_ratio * int(defer_weights))
        self._deferred = (plan, (top, left), oh, ow, o, bool(do_sigmoid)) if self.weights_deferred else None
        self._run_flow(plan, int(self.C.iters), (top, left), oh, ow, o, weighted, bool(do_sigmoid) and not post,
                       defer_wh=self.weights_deferred, want_flow=(mode == "flow"))
        if self.weights_deferred:
            return self._deliver(o, None, mode, oh, ow, numpy_out, borrow)
        logger.debug(f"flow enqueue time [s]: {float(timer() - start_time)}")
        weights = o["w"] if weighted else None
        if post and weights is not None:
            self._postprocess_logits(o, oh, ow, do_sigmoid)    # the logits, then the sigmoid (raft.py:152-159)
        return self._deliver(o, weights, mode, oh, ow, numpy_out, borrow)
