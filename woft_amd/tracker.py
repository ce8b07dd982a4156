"""WOFT tracker on the HIP flow / fit operators: the counterpart of the reference's
YAOFTrackerSingleControl (/root/reference/pytracking/tracker/YAOF_tracker_single_control.py,
TRK below) with the same constructor / init / track / set_fast_meta surface, the same config keys
and the same `meta` fields, so reference-style config files and WOFT_demo.py drive it unchanged.

Differences that are not visible in the results: frames live on the GPU (the two
cv2.warpPerspective calls of TRK:89-95 are one HIP kernel), the template's feature / context
tensors are computed once (the flow provider pins the template image), and there is a single
device->host read per frame for the homography and one for the re-detection test.
"""
import logging
import os
from inspect import signature
from types import SimpleNamespace

import numpy as np
import torch

from . import ops
from .homography import compose_H

logger = logging.getLogger(__name__)


def _count_components(mask_bool):
    from scipy import ndimage
    _, n = ndimage.label(mask_bool, structure=np.ones((3, 3), dtype=bool))   # 8-connected, as findContours
    return n


def make_forward_compatible(subsampler_fn):
    """3-argument subsamplers get a 4th (post-hoc weights) argument (TRK:344-362)."""
    if len(signature(subsampler_fn).parameters) == 3:
        def new_fn(coords_a, coords_b, weights, post_weights):
            if post_weights is not None:
                raise NotImplementedError("Using post-hoc weights post-processing with a subsampler that takes only 3 arguments")
            return subsampler_fn(coords_a, coords_b, weights) + (None,)
        if hasattr(subsampler_fn, "woft_spec"):
            new_fn.woft_spec = subsampler_fn.woft_spec
        return new_fn
    return subsampler_fn


def _to_gpu_u8(img):
    if isinstance(img, torch.Tensor):
        return img if img.is_cuda else img.cuda()
    return torch.from_numpy(np.ascontiguousarray(img)).cuda()


class YAOFTrackerSingleControl:
    def __init__(self, config):
        self.C = config
        if self.C.subsampler_fn:
            self.C.subsampler_fn = make_forward_compatible(self.C.subsampler_fn)
        self.flower = config.flow_config.of_class(config.flow_config)
        self.device = "cuda"
        self._fused = self._fused_specs()

    # ---- fused device-side path --------------------------------------------------------------
    def _fused_specs(self):
        """When the config's estimator / subsampler / re-detection callables are the tagged ones of
        woft_amd.presets, masking + Sobol selection + H fit + inlier test run as HIP kernels with ONE
        device->host read per flow (results identical to calling the callables).  Any other config takes
        the generic path that calls them as the reference tracker does."""
        C = self.C
        if os.environ.get("WOFT_FUSED", "1") == "0":
            return None
        est = getattr(C.H_estimator, "woft_spec", None)
        red = getattr(C.redet_success_fn, "woft_spec", None)
        sub = getattr(C.subsampler_fn, "woft_spec", None) if C.subsampler_fn else ("none", 0)
        if est is None or red is None or sub is None or red[0] != "inliers":
            return None
        if C.post_hoc_weights_postprocessing_fn or C.flow_numpy_out:
            return None
        if not hasattr(self.flower, "pin_source") or self.flower.C.raft_type != "weighted":
            return None
        n_draw = int(sub[1]) if sub[0] == "sobol" else 0
        if n_draw > 1024:
            return None
        from .presets import sobol_points
        return dict(reweight=int(est[1]), huber_k=float(est[2]), n_irls=int(est[3]), thr=float(red[1]),
                    min_frac=float(red[2]), n_draw=n_draw,
                    sobol_u=torch.from_numpy(sobol_points(n_draw).astype(np.float32)).cuda() if n_draw else None)

    def _fused_buffers(self, h, w):
        key = (h, w)
        if getattr(self, "_fb_key", None) != key:
            cap = 1024 if self._fused["n_draw"] else h * w
            self._fb = dict(ws=ops.tc_select_ws(h * w), pa=torch.empty(cap, 2, device=self.device),
                            pb=torch.empty(cap, 2, device=self.device), w=torch.empty(cap, device=self.device),
                            res=torch.zeros(16, dtype=torch.float32, device=self.device))
            self._fb_key = key
        return self._fb

    def _fused_fit(self, dst, weights, tmask_u8, pwmask_u8, h, w, check_dst):
        """-> (H 3x3 float64 mapping dst -> src, success flag, #kept, #selected)."""
        F, b = self._fused, self._fused_buffers(h, w)
        res = b["res"]
        ires = res.view(torch.int32)
        ops.tc_select(dst, weights, tmask_u8, pwmask_u8, h, w, check_dst, F["sobol_u"], b["ws"], b["pa"], b["pb"],
                      b["w"], ires[12:14])
        ops.hfit(b["pa"], b["pb"], b["w"], res[0:9], ires[10:11], count=ires[12:13], reweight=F["reweight"],
                 huber_k=F["huber_k"], n_irls=F["n_irls"])
        ops.inlier_frac(b["pa"], b["pb"], res[0:9], res[9:10], thr=F["thr"], count=ires[12:13])
        host = res.cpu()                                             # the frame's single device->host read
        ih = host.view(torch.int32)
        status, n_sel, n_kept = int(ih[10]), int(ih[12]), int(ih[13])
        if status == 1:
            raise AssertionError(torch.Size([1, n_sel, 2]))          # least_squares_H.py:162 (fewer than 4 points)
        Hm = host[0:9].numpy().astype(np.float64).reshape(3, 3)
        return Hm, bool(float(host[9]) > F["min_frac"]), n_kept, n_sel

    def init(self, img, mask, img_identifier=None):
        if self.C.downscale_inputs:                                  # TRK:27-30
            k = self.C.downscale_inputs
            img = ops.resize_by_factor_u8(_to_gpu_u8(img), k)
            mask = ops.resize_by_factor_u8(_to_gpu_u8(np.ascontiguousarray(mask) if not isinstance(mask, torch.Tensor)
                                                      else mask), k)
            img_identifier = None
        mask_np = mask.cpu().numpy() if isinstance(mask, torch.Tensor) else np.asarray(mask)
        self.template_img = img
        self.template_mask = torch.from_numpy(mask_np > 0).to(self.device)
        self.np_template_mask = mask_np
        self._template_mask_u8 = torch.from_numpy((mask_np > 0).astype(np.uint8) * 255).to(self.device)
        assert _count_components(mask_np > 0) == 1                   # TRK:36-37 (single contour)
        if hasattr(self.flower, "pin_source"):
            self.flower.pin_source(self.template_img)
            # opt-in (config key mask_weight_head = True): only correspondences that start inside the template mask
            # survive _mask_coords (TRK:287-312), so the flow weights of the other template pixels are never read and
            # the weight head can skip them -- identical tracks, ~25 % faster at a quarter-frame mask.  Default: the
            # head is evaluated on every pixel, as the reference's network does.
            if hasattr(self.flower, "pin_weight_region") and self.C.mask_weight_head:
                self.flower.pin_weight_region(mask_np > 0)
        self.prev_H2init = np.eye(3)
        self.last_good_H2init = np.eye(3)
        self.prev_img_identifier = img_identifier
        self.prev_img = img
        self.fast_forward = False
        self.lost = False
        self.N_lost = 0

    def set_fast_meta(self, meta):
        self.fast_forward = True
        self.fast_forward_H2init = meta.estim_H_current2template
        self.fast_forward_meta = meta
        if self.C.downscale_inputs:
            raise NotImplementedError("Fastforward not compatible with input downscaling yet.")

    def track(self, input_img, debug=False, img_identifier=None):
        meta = SimpleNamespace()
        if self.C.downscale_inputs:                                  # TRK:60-61
            input_img = ops.resize_by_factor_u8(_to_gpu_u8(input_img), self.C.downscale_inputs)
        if self.fast_forward:                                        # TRK:63-76
            H_cur2init = self.fast_forward_H2init
            meta = self.fast_forward_meta
            self.last_good_H2init = H_cur2init
            self.lost, self.N_lost = False, 0
            self.prev_img_identifier = img_identifier
            self.prev_img = input_img
            self.prev_H2init = H_cur2init
            self.fast_forward = False
            return H_cur2init, meta

        if self.C.no_prewarp_after_N and self.N_lost > self.C.no_prewarp_after_N:
            self.last_good_H2init = np.eye(3)
        meta.last_good_H2init = self.last_good_H2init.copy()

        # 'global' flow: template -> current frame pre-warped by the last good homography (TRK:85-102)
        prewarp_H = self.last_good_H2init
        frame = _to_gpu_u8(input_img)
        Hh, Ww = frame.shape[:2]
        valid = None
        if np.array_equal(prewarp_H, np.eye(3)):
            prewarped, pw_mask = frame, None                         # identity warp: same image, mask all-true
        else:
            prewarped = torch.empty_like(frame)
            valid = torch.empty(Hh, Ww, dtype=torch.uint8, device=self.device)
            ops.warp_perspective_u8(frame, prewarp_H, prewarped, valid)
            pw_mask = valid > 0
        template_coords, cur_pw_coords, weights = self.flower.compute_flow(
            self.template_img, prewarped, mode="TC", vis=False, do_sigmoid=True)
        if self._fused is not None:
            return self._track_fused(meta, frame, prewarp_H, cur_pw_coords, weights,
                                     None if pw_mask is None or self.C.do_not_mask_TCs_by_prewarped else valid,
                                     img_identifier)
        post_hoc_weights = None
        if self.C.post_hoc_weights_postprocessing_fn:
            post_hoc_weights = self.flower.postprocess_weights(weights.clone(), self.C.post_hoc_weights_postprocessing_fn)
        if pw_mask is None:
            pw_mask = torch.ones(Hh, Ww, dtype=torch.bool, device=self.device)
        template_coords, cur_pw_coords, weights, post_hoc_weights, _ = self._mask_coords(
            template_coords, cur_pw_coords, weights, post_hoc_weights, pw_mask,
            do_pw_mask=not self.C.do_not_mask_TCs_by_prewarped)
        template_coords = template_coords.float()
        if self.C.subsampler_fn:
            template_coords, cur_pw_coords, weights, post_hoc_weights = self.C.subsampler_fn(
                template_coords, cur_pw_coords, weights, post_hoc_weights)

        H_prewarped2init = self.C.H_estimator(cur_pw_coords.t()[None], template_coords.t()[None], weights).float()
        np_H_prewarped2init = H_prewarped2init.detach().cpu().numpy()[0].astype(np.float64)
        H_global_cur2init = compose_H(prewarp_H, np_H_prewarped2init)
        meta.H_global_cur2init = H_global_cur2init.copy()
        global_H_success = bool(self.C.redet_success_fn(
            H_prewarped2init, template_coords, cur_pw_coords,
            post_hoc_weights if post_hoc_weights is not None else weights))
        logger.debug(f"global_H_success: {global_H_success}")

        if global_H_success:
            H_cur2init = H_global_cur2init
            self.lost, self.N_lost = False, 0
        else:
            self.lost = True
            self.N_lost += 1
            if self.C.no_local_H:
                H_cur2init = H_global_cur2init
            else:                                                    # local flow t-1 -> t (TRK:178-207)
                prev_coords, cur_coords, weights = self.flower.compute_flow(
                    self.prev_img, frame, mode="TC", src_img_identifier=None, do_sigmoid=True,
                    numpy_out=bool(self.C.flow_numpy_out))
                post_hoc_weights = None
                if self.C.post_hoc_weights_postprocessing_fn:
                    post_hoc_weights = self.flower.postprocess_weights(weights.clone(), self.C.post_hoc_weights_postprocessing_fn)
                prev_coords, cur_coords, weights, post_hoc_weights = self._mask_coords_flow(
                    prev_coords, cur_coords, weights, post_hoc_weights)
                if self.C.subsampler_fn:
                    prev_coords, cur_coords, weights, post_hoc_weights = self.C.subsampler_fn(
                        prev_coords, cur_coords, weights, post_hoc_weights)
                try:
                    H_flow = self.C.H_estimator(cur_coords.t()[None], prev_coords.float().t()[None], weights)
                    H_flow = H_flow.detach().cpu().numpy()[0].astype(np.float64)
                    if not np.all(np.isfinite(H_flow)):
                        raise FloatingPointError("singular homography system")
                    H_local_cur2init = compose_H(H_flow, self.prev_H2init)
                except Exception:
                    logger.warning("local flow RANSAC failed")
                    H_local_cur2init = self.prev_H2init
                meta.H_local_cur2init = H_local_cur2init.copy()
                H_cur2init = H_local_cur2init

        if debug:
            logger.debug("debug visualisation (TRK:210-264) needs the OpenCV GUI and is not part of the HIP path")

        self.prev_img_identifier = img_identifier
        self.prev_img = frame
        self.prev_H2init = H_cur2init.copy()
        if not self.lost:
            self.last_good_H2init = H_cur2init.copy()
        meta.lost = self.lost
        meta.N_lost = self.N_lost
        meta.global_H_success = global_H_success
        if self.C.downscale_inputs:                                  # TRK:280-283
            k = self.C.downscale_inputs
            H_cur2init = compose_H(np.diag([1.0 / k, 1.0 / k, 1.0]), H_cur2init, np.diag([float(k), float(k), 1.0]))
        return H_cur2init, meta

    def _track_fused(self, meta, frame, prewarp_H, cur_pw_coords, weights, pw_valid_u8, img_identifier):
        """TRK:134-278 with masking / selection / fit / inlier test on the device."""
        Hh, Ww = frame.shape[:2]
        Hpw, success, _, _ = self._fused_fit(cur_pw_coords, weights, self._template_mask_u8, pw_valid_u8, Hh, Ww, 1)
        H_global_cur2init = compose_H(prewarp_H, Hpw)
        meta.H_global_cur2init = H_global_cur2init.copy()
        if success:
            H_cur2init = H_global_cur2init
            self.lost, self.N_lost = False, 0
        else:
            self.lost = True
            self.N_lost += 1
            if self.C.no_local_H:
                H_cur2init = H_global_cur2init
            else:
                _, cur_coords, weights = self.flower.compute_flow(self.prev_img, frame, mode="TC",
                                                                  src_img_identifier=None, do_sigmoid=True)
                if np.array_equal(self.prev_H2init, np.eye(3)):
                    prev_mask_u8 = self._template_mask_u8
                else:
                    prev_mask_u8 = torch.empty_like(self._template_mask_u8)
                    ops.warp_perspective_u8(self._template_mask_u8, np.linalg.inv(self.prev_H2init), prev_mask_u8,
                                            None, nearest=True)
                try:
                    H_flow, _, _, _ = self._fused_fit(cur_coords, weights, prev_mask_u8, None, Hh, Ww, 0)
                    if not np.all(np.isfinite(H_flow)):
                        raise FloatingPointError("singular homography system")
                    H_local_cur2init = compose_H(H_flow, self.prev_H2init)
                except Exception:
                    logger.warning("local flow RANSAC failed")
                    H_local_cur2init = self.prev_H2init
                meta.H_local_cur2init = H_local_cur2init.copy()
                H_cur2init = H_local_cur2init
        self.prev_img_identifier = img_identifier
        self.prev_img = frame
        self.prev_H2init = H_cur2init.copy()
        if not self.lost:
            self.last_good_H2init = H_cur2init.copy()
        meta.lost = self.lost
        meta.N_lost = self.N_lost
        meta.global_H_success = success
        if self.C.downscale_inputs:
            k = self.C.downscale_inputs
            H_cur2init = compose_H(np.diag([1.0 / k, 1.0 / k, 1.0]), H_cur2init, np.diag([float(k), float(k), 1.0]))
        return H_cur2init, meta

    def _mask_coords(self, template_coords, cur_coords, weights, post_weights, pw_mask=None, do_pw_mask=True):
        """TRK:287-312."""
        in_template_mask = self.template_mask[template_coords[1, :], template_coords[0, :]]
        if pw_mask is not None:
            H, W = pw_mask.shape
            cur_coords_int = cur_coords.round().long()
            cur_coords_oob = torch.logical_or(
                torch.any(cur_coords < 0, dim=0),
                torch.logical_or(cur_coords_int[0, :] >= W, cur_coords_int[1, :] >= H))
            in_pw_mask = ~cur_coords_oob
            if do_pw_mask:
                cx = cur_coords_int[0].clamp(0, W - 1)
                cy = cur_coords_int[1].clamp(0, H - 1)
                in_pw_mask = in_pw_mask & pw_mask[cy, cx]
            in_mask = torch.logical_and(in_template_mask, in_pw_mask)
        else:
            in_mask = in_template_mask
        template_coords = template_coords[:, in_mask]
        cur_coords = cur_coords[:, in_mask]
        if weights is not None:
            weights = weights[:, in_mask]
        if post_weights is not None:
            post_weights = post_weights[:, in_mask]
        return template_coords, cur_coords, weights, post_weights, in_mask

    def _mask_coords_flow(self, prev_coords, cur_coords, weights, post_weights):
        """TRK:314-327: template mask carried to frame t-1 by inv(prev_H2init), nearest neighbour."""
        Hm = np.linalg.inv(self.prev_H2init)
        if np.array_equal(self.prev_H2init, np.eye(3)):
            prev_mask = self.template_mask
        else:
            warped = torch.empty_like(self._template_mask_u8)
            ops.warp_perspective_u8(self._template_mask_u8, Hm, warped, None, nearest=True)
            prev_mask = warped > 0
        if not isinstance(prev_coords, torch.Tensor):
            prev_mask = prev_mask.cpu().numpy()
        in_mask = prev_mask[prev_coords[1, :], prev_coords[0, :]]
        prev_coords = prev_coords[:, in_mask]
        cur_coords = cur_coords[:, in_mask]
        if weights is not None:
            weights = weights[:, in_mask]
        if post_weights is not None:
            post_weights = post_weights[:, in_mask]
        return prev_coords, cur_coords, weights, post_weights
