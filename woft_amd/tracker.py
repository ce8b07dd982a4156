"""WOFT tracker on the HIP flow / fit operators: the counterpart of the reference's
YAOFTrackerSingleControl (/root/reference/pytracking/tracker/YAOF_tracker_single_control.py,
TRK below) with the same constructor / init / track / set_fast_meta surface, the same config keys
and the same `meta` fields, so reference-style config files and WOFT_demo.py drive it unchanged.

Organisation (this file's own, not the reference's): a frame is `_global_stage` (template ->
pre-warped frame) and, when the re-detection test rejects its homography, `_local_stage`
(frame t-1 -> frame t); both hand a dense correspondence field plus the two masks that prune it
to ONE solver, `_solve`, which has two interchangeable back ends with identical results (tested
bit for bit):

  * device back end  -- configs built from woft_amd.presets (tagged callables): masking,
    order-preserving compaction, Sobol selection, H fit and inlier test are HIP kernels
    (csrc/select.hip, csrc/hfit.hip) with ONE device->host read per flow;
  * callable back end -- any other reference-format config: the keep rule runs as the same
    `select` kernel (woft_tc_flags), the surviving correspondences are handed to the config's own
    subsampler / estimator / re-detection callables exactly as TRK:141-162,196-199 hands them.

Frames live on the GPU (the two cv2.warpPerspective calls of TRK:89-95 are one HIP kernel) and the
template's feature / context tensors are computed once (the flow provider pins the template).
"""
import logging
import os
import time
from inspect import signature
from types import SimpleNamespace

import numpy as np
import torch

from . import ops
from .homography import compose_H

logger = logging.getLogger(__name__)
_EYE = np.eye(3)


def _count_components(mask_bool):
    from scipy import ndimage
    _, n = ndimage.label(mask_bool, structure=np.ones((3, 3), dtype=bool))   # 8-connected, as findContours
    return n


def make_forward_compatible(subsampler_fn):
    """Subsamplers written for (coords_a, coords_b, weights) are lifted to the 4-argument form that also carries the
    post-hoc weights (TRK:344-362); a lifted one cannot forward post-hoc weights and says so."""
    if len(signature(subsampler_fn).parameters) != 3:
        return subsampler_fn

    def lifted(coords_a, coords_b, weights, post_weights):
        if post_weights is not None:
            raise NotImplementedError("a 3-argument subsampler cannot carry post-hoc processed weights; "
                                      "give it a 4th parameter")
        return (*subsampler_fn(coords_a, coords_b, weights), None)
    if hasattr(subsampler_fn, "woft_spec"):
        lifted.woft_spec = subsampler_fn.woft_spec
    return lifted


def _device_u8(img, copy=False):
    """(H, W[, C]) uint8 image -> packed CUDA tensor (the kernels take raw pointers to packed uint8 rows)."""
    if isinstance(img, torch.Tensor):
        t = img if img.is_cuda else img.cuda()
    else:
        a = np.asarray(img)
        if a.dtype != np.uint8:
            raise TypeError(f"frames and masks must be uint8, got {a.dtype}")
        return torch.from_numpy(np.ascontiguousarray(a)).cuda()
    if t.dtype != torch.uint8:
        raise TypeError(f"frames and masks must be uint8, got {t.dtype}")
    if not t.is_contiguous():
        return t.contiguous()
    return t.clone() if copy and t is img else t


class _FrameUploader:
    """Host frames (numpy, as WOFT_demo.py:61-78 / TRK:113-120 hand them to track()) -> one of two alternating device buffers,
    so that the frame that becomes `prev_img` stays valid while the next one arrives (the local stage reads frame t-1,
    TRK:181-184).  Measured on the GPU box (tools/micro/upload_probe.py, 1080p BGR = 6.2 MB): the runtime's own pageable copy
    0.131 ms -- it pipelines its staging internally and sits at the PCIe floor (6.2 MB / ~55 GB/s = 0.11 ms) --; one memcpy into a
    pinned buffer + one asynchronous H2D copy 0.35 ms (round 4's path: the two do not overlap); the same in 4 pipelined pieces
    (woft_upload_u8) 0.18 ms, in 8 pieces 0.38 ms (a hipMemcpyAsync call costs ~20 us).  So the frame is copied DIRECTLY
    (`WOFT_UPLOAD=staged`: the pinned path in 4 pieces).  Nothing of a frame's GPU work can start before its pixels are there, and
    the caller hands over frame t only after frame t-1's result: the 0.13 ms are on the critical path (1.5 % of a 1080p frame).
    LIFETIME of what track() keeps: `tracker.prev_img` (and anything else holding the returned device frame) is valid until
    the SECOND next host-frame track() call, which reuses its buffer; a caller that wants a frame for longer clones it.  A change
    of frame shape allocates new buffers (the old prev_img stays valid, as its own tensor)."""
    STAGED = os.environ.get("WOFT_UPLOAD", "direct") == "staged"

    def __init__(self):
        self.key, self.stage, self.dev, self.done, self.i = None, None, None, None, 0

    def __call__(self, a):
        a = np.asarray(a)
        if a.dtype != np.uint8:
            raise TypeError(f"frames and masks must be uint8, got {a.dtype}")
        key = a.shape
        if key != self.key:
            self.dev = [torch.empty(key, dtype=torch.uint8, device="cuda") for _ in range(2)]
            if self.STAGED:
                self.stage = [torch.empty(key, dtype=torch.uint8).pin_memory() for _ in range(2)]
                self.done = [torch.cuda.Event(), torch.cuda.Event()]
            self.key, self.i = key, 0
        i = self.i = self.i ^ 1
        if not self.STAGED:
            self.dev[i].copy_(torch.from_numpy(np.ascontiguousarray(a)))       # (returns when the host pages have been read)
            return self.dev[i]
        self.done[i].synchronize()                   # (the copy that last read this staging buffer: two frames ago)
        if a.flags.c_contiguous and a.nbytes >= (1 << 20):
            from . import _lib
            _lib.check(_lib.load().woft_upload_u8(a.ctypes.data, self.stage[i].data_ptr(), self.dev[i].data_ptr(), a.nbytes, 4,
                                                  _lib.stream_ptr()), "woft_upload_u8")
        else:                                        # (non-contiguous views, e.g. a BGR<->RGB flipped array; small frames)
            np.copyto(self.stage[i].numpy(), a)
            self.dev[i].copy_(self.stage[i], non_blocking=True)
        self.done[i].record()
        return self.dev[i]


class _Fit(SimpleNamespace):
    """Result of one solve: H (3x3 float64, cur -> src) and, for the global stage, the re-detection verdict."""


class YAOFTrackerSingleControl:
    DEVICE = "cuda"                  # (a host-logic test may build the tracker around a stub flow provider on "cpu")

    def __init__(self, config):
        self.C = config
        if self.C.subsampler_fn:
            self.C.subsampler_fn = make_forward_compatible(self.C.subsampler_fn)
        self.flower = config.flow_config.of_class(config.flow_config)
        self.device = self.DEVICE
        self._fused = self._fused_specs()
        self._sparse_weights = False
        self._replay = None
        self._announced = False
        self._local_chain = None          # (target frame object, frame number) of the last frame t-1 -> t flow
        self._n_tracked = 0
        self._upload = _FrameUploader()
        self.host_wait_s = 0.0       # seconds this tracker's thread spent blocked on the per-flow result read (device back end)

    # ---- which solver back end ------------------------------------------------------------------
    def _fused_specs(self):
        """Parameters of the device back end -- dict(reweight, huber_k, n_irls, thr, min_frac, n_draw, sobol_u) -- when the
        config's estimator / subsampler / re-detection callables are the weighted LSq / IRLS estimators, the Sobol-n draw and
        the inlier-fraction test: tagged by woft_amd.presets, or found to BEHAVE as those (woft_amd.probe: reference-format
        configs define them inline, configs/..._wLSq.py:14-53); else None (callable back end: the config's own functions run)."""
        C = self.C
        self.solver_decision = "callable back end"
        v = C.device_solver
        if os.environ.get("WOFT_FUSED", "1") == "0" or (not isinstance(v, type(C)) and v is not None and not v):
            self.solver_decision += " (device back end switched off)"
            return None
        if C.post_hoc_weights_postprocessing_fn or C.flow_numpy_out:
            return None
        if not hasattr(self.flower, "pin_source") or self.flower.C.raft_type != "weighted":
            return None
        from .probe import solver_spec
        spec, how = solver_spec(C.H_estimator, C.subsampler_fn or None, C.redet_success_fn, device=self.device)
        self.solver_decision = ("device back end" if spec is not None else "callable back end") + f" ({how})"
        # A decision taken by PROBING replaces the config's own callables with the device solver on the strength of what they did
        # on synthetic inputs (woft_amd.probe): said at WARNING level, with the way out.  Tagged presets / no replacement: INFO.
        probed = spec is not None and "probed" in how
        logger.log(logging.WARNING if probed else logging.INFO,
                   f"tracker solver: {self.solver_decision}"
                   + (" -- the config's estimator / subsampler / re-detection callables were recognised by behavioural probing and "
                      "are replaced by the HIP solver (same results, tested); set `device_solver = False` in the tracker config "
                      "(or WOFT_FUSED=0) to run the config's own callables instead" if probed else ""))
        if spec is None:
            return None
        from .presets import sobol_points
        n_draw = spec["n_draw"]
        spec["sobol_u"] = torch.from_numpy(sobol_points(n_draw).astype(np.float32)).to(self.device) if n_draw else None
        return spec

    def _fused_buffers(self, n_grid):
        if getattr(self, "_fb_key", None) != n_grid:
            cap = 1024 if self._fused["n_draw"] else n_grid
            self._fb = dict(ws=ops.tc_select_ws(n_grid), pa=torch.empty(cap, 2, device=self.device),
                            pb=torch.empty(cap, 2, device=self.device), w=torch.empty(cap, device=self.device),
                            res=torch.zeros(16, dtype=torch.float32, device=self.device),
                            fit_ws=ops.hfit_ws(self.device) if cap > ops.HFIT_SINGLE_MAX else None)
            self._fb_key = n_grid
        return self._fb

    # ---- public surface (TRK:19-57) -----------------------------------------------------------------
    def init(self, img, mask, img_identifier=None):
        k = self.C.downscale_inputs
        if k:                                                        # TRK:27-30
            img = ops.resize_by_factor_u8(_device_u8(img), k)
            mask = ops.resize_by_factor_u8(_device_u8(mask), k)
            img_identifier = None
        mask_np = mask.cpu().numpy() if isinstance(mask, torch.Tensor) else np.asarray(mask)
        inside = mask_np > 0
        assert _count_components(inside) == 1                        # TRK:36-37 (single contour)
        self.template_img = img
        self.np_template_mask = mask_np
        self._local_chain = None
        self.template_mask = torch.from_numpy(inside).to(self.device)
        self._template_mask_u8 = torch.from_numpy(inside.astype(np.uint8) * 255).to(self.device)
        if hasattr(self.flower, "pin_source"):
            self.flower.pin_source(self.template_img)
            # Only correspondences that start inside the template mask survive the keep rule (TRK:287-312): the flow
            # weights of the other template pixels are never read by this tracker, and the weight head has no
            # cross-pixel terms (weighted_raft.py:363-383), so for flows FROM the template it is evaluated on the mask's
            # pixels only -- bit-identical weights there, identical homographies (tested), ~25 % fewer milliseconds per
            # frame at a quarter-frame mask.  Only the tracker's own calls ask for the region (`weight_region=True`, _flow
            # below): `compute_flow` called directly returns the full weight map; config key mask_weight_head = False
            # evaluates the head everywhere for the tracker too, as the reference's network does.
            if hasattr(self.flower, "pin_weight_region"):
                self.flower.pin_weight_region(inside if self._mask_weight_head() else None)
            # ... and with a subsampler in front of the fit (the default config draws 500 correspondences), only the weights
            # of the DRAWN correspondences are read, and the draw does not depend on the weights: the head is evaluated after
            # the selection, on the windows under the drawn pixels' upsampling support (config key sparse_weight_head =
            # False / env WOFT_SPARSE_WH=0: on the whole mask region, as above).  Identical homographies (tested).
            v = self.C.sparse_weight_head
            on = (os.environ.get("WOFT_SPARSE_WH", "1") != "0") if (isinstance(v, type(self.C)) or v is None) else bool(v)
            self._sparse_weights = bool(on and self._fused is not None and self._fused["n_draw"] and self._mask_weight_head()
                                        and hasattr(self.flower, "finish_weights"))
        self._set_pose(_EYE.copy(), good=True)
        self.prev_img, self.prev_img_identifier = img, img_identifier
        self.lost, self.N_lost = False, 0
        self._replay = None

    def set_fast_meta(self, meta):
        """Next track() call replays a stored result instead of computing flow (TRK:49-55)."""
        if self.C.downscale_inputs:
            raise NotImplementedError("Fastforward not compatible with input downscaling yet.")
        self._replay = meta

    @property
    def fast_forward(self):
        return self._replay is not None

    @fast_forward.setter
    def fast_forward(self, value):
        """Reference-style `tracker.fast_forward = False` (TRK:47,55,64) cancels a pending replay; True without a stored
        meta has nothing to replay (set_fast_meta is the way in)."""
        if not value:
            self._replay = None
        elif self._replay is None:
            raise ValueError("fast_forward = True needs set_fast_meta(meta)")

    def track(self, input_img, debug=False, img_identifier=None):
        if self.C.downscale_inputs:                                  # TRK:60-61
            input_img = ops.resize_by_factor_u8(_device_u8(input_img), self.C.downscale_inputs)
        if self._replay is not None:                                 # TRK:63-76
            meta, self._replay = self._replay, None
            H_cur2init = meta.estim_H_current2template
            self.lost, self.N_lost = False, 0
            self.prev_H2init = self.last_good_H2init = H_cur2init
            self.prev_img = input_img.clone() if isinstance(input_img, torch.Tensor) else np.array(input_img)
            self.prev_img_identifier = img_identifier
            return H_cur2init, meta

        meta = SimpleNamespace()
        self._n_tracked += 1
        if self.C.no_prewarp_after_N and self.N_lost > self.C.no_prewarp_after_N:
            self.last_good_H2init = _EYE.copy()                      # TRK:78-79
        meta.last_good_H2init = self.last_good_H2init.copy()
        # (becomes prev_img: the reference keeps a copy.  Host frames go through the pinned double buffer)
        frame = _device_u8(input_img, copy=True) if isinstance(input_img, torch.Tensor) else self._upload(input_img)

        prewarp_H = self.last_good_H2init
        fit = self._global_stage(frame, prewarp_H)
        H_global = compose_H(prewarp_H, fit.H)
        meta.H_global_cur2init = H_global.copy()
        logger.debug(f"global_H_success: {fit.success}")
        if fit.success:
            self.lost, self.N_lost = False, 0
            H_cur2init = H_global
        else:
            self.lost, self.N_lost = True, self.N_lost + 1
            H_cur2init = H_global
            if not self.C.no_local_H:
                H_cur2init = meta.H_local_cur2init = self._local_stage(frame)
        if debug:
            logger.debug("debug visualisation (TRK:210-264) needs the OpenCV GUI and is not part of the HIP path")

        self.prev_img, self.prev_img_identifier = frame, img_identifier
        self._set_pose(H_cur2init.copy(), good=not self.lost)
        meta.lost, meta.N_lost, meta.global_H_success = self.lost, self.N_lost, fit.success
        if not self._announced:           # first tracked frame: which arithmetic and which solver produced these results
            self._announced = True
            meta.precision = getattr(self.flower, "precision", None)
            meta.precision_source = getattr(self.flower, "precision_source", None)
            meta.solver_decision = self.solver_decision
        k = self.C.downscale_inputs
        if k:                                                        # TRK:280-283
            H_cur2init = compose_H(np.diag([1.0 / k, 1.0 / k, 1.0]), H_cur2init, np.diag([float(k), float(k), 1.0]))
        return H_cur2init, meta

    def _mask_weight_head(self):
        if self.C.post_hoc_weights_postprocessing_fn:     # (a spatial filter of the weight MAP would read outside the mask)
            return False
        v = self.C.mask_weight_head
        if isinstance(v, type(self.C)) or v is None:      # key absent (Config returns an empty, falsy Config): default on
            return os.environ.get("WOFT_MASK_WEIGHT_HEAD", "1") != "0"
        return bool(v)

    def _set_pose(self, H, good):
        self.prev_H2init = H
        if good:
            self.last_good_H2init = H.copy()

    # ---- the two flow stages ------------------------------------------------------------------------
    def _flow(self, src, dst, src_is_previous_dst=False):
        """-> (grid coords (2, n) int64, target coords (2, n) f32, weights (1, n) f32 | None, (gh, gw)); borrowed
        buffers of the provider: consumed before the next flow."""
        # (borrowed buffers, and -- only for THIS caller -- weights restricted to the region pinned in init(): a direct
        #  compute_flow() call by anybody else returns the full weight map, as the reference's does)
        kw = {"borrow": True, "weight_region": True} if hasattr(self.flower, "pin_source") else {}
        if self._sparse_weights:
            # both stages: the template flow on its mask region, the frame t-1 -> t flow of a lost frame on the pixels of
            # the carried mask (TRK:314-327) -- either way only the drawn correspondences' weights are read (_solve_device)
            kw["defer_weights"] = int(self._fused["n_draw"])
        if src_is_previous_dst and "borrow" in kw:
            kw["src_is_previous_dst"] = True
        src_xy, dst_xy, w = self.flower.compute_flow(src, dst, mode="TC", vis=False, src_img_identifier=None,
                                                     do_sigmoid=True, **kw)
        s = getattr(self.flower, "last_flow_shape", None)
        return src_xy, dst_xy, w, ((s["H"], s["W"]) if s else None)

    def _global_stage(self, frame, prewarp_H):
        """Template -> frame pre-warped by the last good homography (TRK:85-162).  Kept: correspondences that start
        in the template mask, land inside the frame and (unless do_not_mask_TCs_by_prewarped) on a pixel the pre-warp
        actually filled."""
        valid = None
        if np.array_equal(prewarp_H, _EYE):
            prewarped = frame                                        # identity warp: same image, every pixel filled
        else:
            # (two per-size buffers, reused: consumed by the flow / the selection before the next frame's warp is enqueued on
            #  the same stream; allocating them per frame was ~15 us of host time in front of the frame's first kernel)
            key = tuple(frame.shape)
            if getattr(self, "_pw_key", None) != key:
                self._pw_buf = (torch.empty_like(frame), torch.empty(frame.shape[:2], dtype=torch.uint8, device=self.device))
                self._pw_key = key
            prewarped, valid = self._pw_buf
            ops.warp_perspective_u8(frame, prewarp_H, prewarped, valid)
        if self.C.do_not_mask_TCs_by_prewarped:
            valid = None
        src_xy, dst_xy, w, grid = self._flow(self.template_img, prewarped)
        return self._solve(src_xy, dst_xy, w, grid, frame.shape[:2], self._template_mask_u8, valid, bounds=True,
                           judge=True)

    def _local_stage(self, frame):
        """Frame t-1 -> frame t, chained onto the previous pose (TRK:171-207).  Kept: correspondences that start in
        the template mask carried to frame t-1 (nearest-neighbour warp by inv(prev_H2init), TRK:314-327).  A failed
        fit keeps the previous pose."""
        # consecutive lost frames: frame t-1 was the TARGET of the previous local flow and is the SOURCE of this one -- the provider
        # then takes its feature map from that flow instead of encoding the same image again (identical values)
        chained = (self._local_chain is not None and self._local_chain[0] is self.prev_img and self._local_chain[1] == self._n_tracked - 1
                   and os.environ.get("WOFT_LOCAL_REUSE", "1") != "0")
        src_xy, dst_xy, w, grid = self._flow(self.prev_img, frame, src_is_previous_dst=chained)
        self._local_chain = (frame, self._n_tracked)
        if np.array_equal(self.prev_H2init, _EYE):
            prev_mask = self._template_mask_u8
        else:
            prev_mask = torch.empty_like(self._template_mask_u8)
            ops.warp_perspective_u8(self._template_mask_u8, np.linalg.inv(self.prev_H2init), prev_mask, None,
                                    nearest=True)
        try:
            fit = self._solve(src_xy, dst_xy, w, grid, frame.shape[:2], prev_mask, None, bounds=False, judge=False)
            if not np.all(np.isfinite(fit.H)):
                raise FloatingPointError("singular homography system")
            return compose_H(fit.H, self.prev_H2init)
        except Exception as ex:
            logger.warning(f"frame-to-frame homography failed ({type(ex).__name__}): pose of the previous frame kept")
            return self.prev_H2init

    # ---- the solver ---------------------------------------------------------------------------------
    def _solve(self, src_xy, dst_xy, w, grid, frame_hw, src_mask_u8, dst_valid_u8, bounds, judge):
        """Prune the dense field and fit H (target -> source coordinates); judge: also run the re-detection test."""
        Hh, Ww = frame_hw
        grid = grid or (Hh, Ww)
        if self._fused is not None:
            return self._solve_device(dst_xy, w, grid, (Hh, Ww), src_mask_u8, dst_valid_u8, bounds)
        return self._solve_callables(src_xy, dst_xy, w, grid, (Hh, Ww), src_mask_u8, dst_valid_u8, bounds, judge)

    def _solve_device(self, dst_xy, w, grid, frame_hw, src_mask_u8, dst_valid_u8, bounds):
        F, b = self._fused, self._fused_buffers(grid[0] * grid[1])
        res = b["res"]
        ires = res.view(torch.int32)
        weighted = F.get("weighted", True)
        if w is None and getattr(self.flower, "weights_deferred", False):
            # The correspondences the fit will read are decided by the flow alone (masks, bounds, Sobol draw): select them
            # first, then let the provider evaluate the weight head on the windows under THEIR upsampling support only and
            # hand back the weights of exactly these pixels (exact: the head has no cross-pixel terms) -- identical fit.
            # An UNWEIGHTED estimator (the reference's "plainLSq" configs call the library with weights=None) reads no weight at
            # all: the deferred head is then simply never evaluated.
            ops.tc_select(dst_xy, None, src_mask_u8, dst_valid_u8, frame_hw[0], frame_hw[1], bounds, F["sobol_u"], b["ws"],
                          b["pa"], b["pb"], b["w"], ires[12:14], grid=grid)
            if weighted:
                self.flower.finish_weights(b["pb"], ires[12:13], b["pb"].shape[0], out=b["w"])      # (pb: the source pixels)
        else:
            ops.tc_select(dst_xy, w, src_mask_u8, dst_valid_u8, frame_hw[0], frame_hw[1], bounds, F["sobol_u"], b["ws"],
                          b["pa"], b["pb"], b["w"], ires[12:14], grid=grid)
        ops.hfit(b["pa"], b["pb"], b["w"] if weighted else None, res[0:9], ires[10:11], count=ires[12:13], reweight=F["reweight"],
                 huber_k=F["huber_k"], n_irls=F["n_irls"], ws=b["fit_ws"])
        ops.inlier_frac(b["pa"], b["pb"], res[0:9], res[9:10], thr=F["thr"], count=ires[12:13])
        # the flow's single device->host read: into a pinned buffer (no staging copy, no allocation), then wait for it
        if getattr(self, "_host_res", None) is None:
            self._host_res = torch.empty(16, dtype=torch.float32).pin_memory()
            self._host_ev = torch.cuda.Event()
        host = self._host_res
        host.copy_(res, non_blocking=True)
        self._host_ev.record()
        t0 = time.perf_counter()
        self._host_ev.synchronize()
        self.host_wait_s += time.perf_counter() - t0      # (bench: wall time minus this = the launch loop's host time per frame)
        ih = host.view(torch.int32)
        if int(ih[10]) == 1:
            raise AssertionError(torch.Size([1, int(ih[12]), 2]))    # least_squares_H.py:162 (fewer than 4 points)
        verdict = F.get("const_verdict")       # (a re-detection test that is `return True` / `return False`: the reference's ablations)
        if verdict is None:
            verdict = bool(np.float32(host[9]) > np.float32(F["min_frac"]))
        return _Fit(H=host[0:9].numpy().astype(np.float64).reshape(3, 3), success=bool(verdict))   # (astype: a copy; the verdict in float32, as torch compares a float32 mean with a Python float)

    def _solve_callables(self, src_xy, dst_xy, w, grid, frame_hw, src_mask_u8, dst_valid_u8, bounds, judge):
        C = self.C
        post = None
        if C.post_hoc_weights_postprocessing_fn:
            post = self.flower.postprocess_weights(w.clone(), C.post_hoc_weights_postprocessing_fn)
        keep = ops.tc_flags(dst_xy if bounds else None, src_mask_u8, dst_valid_u8, frame_hw[0], frame_hw[1], bounds,
                            grid=grid)
        pick = lambda t: None if t is None else t[:, keep]
        src_xy, dst_xy, w, post = pick(src_xy), pick(dst_xy), pick(w), pick(post)
        if judge:
            src_xy = src_xy.float()                                  # TRK:131 (global stage); the local stage hands the
                                                                     # subsampler the int64 grid coordinates (TRK:186-193)
        if C.flow_numpy_out and not judge:                           # (the reference asks numpy of the local flow only)
            to_np = lambda t: None if t is None else t.cpu().numpy()
            src_xy, dst_xy, w, post = to_np(src_xy), to_np(dst_xy), to_np(w), to_np(post)
        if C.subsampler_fn:
            src_xy, dst_xy, w, post = C.subsampler_fn(src_xy, dst_xy, w, post)
        if not judge:
            src_xy = src_xy.float() if isinstance(src_xy, torch.Tensor) else src_xy.astype(np.float32)
        H = C.H_estimator(dst_xy.T[None], src_xy.T[None], w)
        H = H.float() if isinstance(H, torch.Tensor) else torch.as_tensor(np.asarray(H)).float()
        fit = _Fit(H=H.detach().cpu().numpy()[0].astype(np.float64), success=None)
        if judge:
            fit.success = bool(C.redet_success_fn(H, src_xy, dst_xy, post if post is not None else w))
        return fit
