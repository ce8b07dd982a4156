"""ORACLE (test infrastructure, not product code): CPU restatement of the WOFT tracker's
per-frame control flow, /root/reference/pytracking/tracker/YAOF_tracker_single_control.py
(TRK), driven by the oracle flow (oracle/raft_ref.py) and H fit (oracle/hfit_ref.py).

OpenCV is absent from the image, so `cv2.warpPerspective` (TRK:89-94, 315-317) is replaced
by the float bilinear / nearest warps below; **parity with OpenCV's fixed-point
INTER_LINEAR is unpinned** (expect +-1 grey level).  The product's HIP warp implements
exactly the arithmetic written here.
"""
from types import SimpleNamespace

import numpy as np
import torch

from . import hfit_ref, raft_ref


def warp_linear(img, Hm, fill=0.0):
    """dst(x,y) = bilinear src(H^-1 (x,y)), constant 0 outside (cv2.warpPerspective, INTER_LINEAR,
    BORDER_CONSTANT).  float64 coordinates; float32 interpolation weights."""
    H, W = img.shape[:2]
    Hi = np.linalg.inv(Hm)
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    d = Hi[2, 0] * xs + Hi[2, 1] * ys + Hi[2, 2]
    sx = (Hi[0, 0] * xs + Hi[0, 1] * ys + Hi[0, 2]) / d
    sy = (Hi[1, 0] * xs + Hi[1, 1] * ys + Hi[1, 2]) / d
    x0 = np.floor(sx)
    y0 = np.floor(sy)
    fx = (sx - x0).astype(np.float32)
    fy = (sy - y0).astype(np.float32)
    x0 = x0.astype(np.int64)
    y0 = y0.astype(np.int64)
    src = img.astype(np.float32)
    if src.ndim == 2:
        src = src[..., None]
    fx, fy = fx[..., None], fy[..., None]

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        return src[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)] * ok[..., None].astype(np.float32)

    top = tap(y0, x0) * (1 - fx) + tap(y0, x0 + 1) * fx
    bot = tap(y0 + 1, x0) * (1 - fx) + tap(y0 + 1, x0 + 1) * fx
    out = top * (1 - fy) + bot * fy
    return out[..., 0] if img.ndim == 2 else out


def warp_linear_u8(img, Hm):
    return np.clip(np.rint(warp_linear(img, Hm)), 0, 255).astype(np.uint8)


def warp_nearest(img, Hm):
    """cv2.warpPerspective(..., INTER_NEAREST): src(round(H^-1 x)), 0 outside."""
    H, W = img.shape[:2]
    Hi = np.linalg.inv(Hm)
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    d = Hi[2, 0] * xs + Hi[2, 1] * ys + Hi[2, 2]
    sx = np.rint((Hi[0, 0] * xs + Hi[0, 1] * ys + Hi[0, 2]) / d).astype(np.int64)
    sy = np.rint((Hi[1, 0] * xs + Hi[1, 1] * ys + Hi[1, 2]) / d).astype(np.int64)
    ok = (sy >= 0) & (sy < H) & (sx >= 0) & (sx < W)
    return img[np.clip(sy, 0, H - 1), np.clip(sx, 0, W - 1)] * ok


def resize_linear_u8(img, factor):
    """cv2.resize(img, None, fx=1/factor, fy=1/factor), INTER_LINEAR geometry, float arithmetic (TRK:27-30)."""
    h, w = img.shape[:2]
    ho, wo = int(round(h / factor)), int(round(w / factor))
    fy = (np.arange(ho, dtype=np.float32) + np.float32(0.5)) * np.float32(factor) - np.float32(0.5)
    fx = (np.arange(wo, dtype=np.float32) + np.float32(0.5)) * np.float32(factor) - np.float32(0.5)
    y0 = np.floor(fy).astype(np.int64)
    x0 = np.floor(fx).astype(np.int64)
    wy = (fy - y0).astype(np.float32)
    wx = (fx - x0).astype(np.float32)
    wy[y0 < 0] = 0
    wx[x0 < 0] = 0
    y0 = np.clip(y0, 0, h - 1)
    x0 = np.clip(x0, 0, w - 1)
    wy[y0 >= h - 1] = 0
    wx[x0 >= w - 1] = 0
    y1 = np.minimum(y0 + 1, h - 1)
    x1 = np.minimum(x0 + 1, w - 1)
    src = img.astype(np.float32)
    if src.ndim == 2:
        src = src[..., None]
    wxx = wx[None, :, None]
    top = src[y0][:, x0] * (1 - wxx) + src[y0][:, x1] * wxx
    bot = src[y1][:, x0] * (1 - wxx) + src[y1][:, x1] * wxx
    wyy = wy[:, None, None]
    out = np.clip(np.rint(top * (1 - wyy) + bot * wyy), 0, 255).astype(np.uint8)
    return out[..., 0] if img.ndim == 2 else out


class TrackerRef:
    """YAOFTrackerSingleControl restated (default WOFT config: QR estimator, Sobol-500,
    no_prewarp_after_N = 10; configs/YAOFT_single_control_repRAFT_sub500_noreliableinl_wLSq.py:56-71)."""

    def __init__(self, sd, iters=12, estimator="qr", subsample=500, no_prewarp_after_N=10,
                 no_local_H=False, small=False, downscale=None, padding_mode="nopad"):
        self.downscale, self.padding_mode = downscale, padding_mode
        self.sd, self.iters, self.small = sd, iters, small
        self.subsample = subsample
        self.no_prewarp_after_N = no_prewarp_after_N
        self.no_local_H = no_local_H
        self.force_fail = ()          # frame indices (0-based track() calls) whose re-detection test is made to fail
        self._n_tracked = 0
        if estimator == "qr":
            self.H_estimator = hfit_ref.find_homography_nonhomogeneous_QR
        elif estimator == "plain_qr":          # configs/..._plainLSq.py:16-20: the library called with weights=None
            self.H_estimator = lambda a, b, w: hfit_ref.find_homography_nonhomogeneous_QR(a, b, None)
        elif estimator == "irls_huber2":
            self.H_estimator = lambda a, b, w: hfit_ref.find_homography_IRLSq_QR(
                a, b, w, reweighting_fn=lambda r: hfit_ref.IRLSq_Huber(r, k=2))
        else:
            raise ValueError(estimator)

    def init(self, img, mask):                                   # TRK:26-47
        if self.downscale:
            img, mask = resize_linear_u8(img, self.downscale), resize_linear_u8(mask, self.downscale)
        self.template_img = img
        self.template_mask = torch.from_numpy(mask > 0)
        self.np_template_mask = mask
        self.prev_H2init = np.eye(3)
        self.last_good_H2init = np.eye(3)
        self.prev_img = img
        self.lost = False
        self.N_lost = 0

    def _flow(self, a, b):
        return raft_ref.compute_flow(self.sd, a, b, self.iters, mode="TC", small=self.small,
                                     weighted=True, do_sigmoid=True, padding_mode=self.padding_mode)

    def _subsample(self, a, b, w):                               # configs/..wLSq.py:31-53
        if not self.subsample:
            return a, b, w
        m = hfit_ref.sobol_subsample_mask(a.shape[1], self.subsample)
        return a[:, m], b[:, m], w[:, m]

    def track(self, img):                                        # TRK:57-285
        meta = SimpleNamespace()
        if self.downscale:
            img = resize_linear_u8(img, self.downscale)
        if self.no_prewarp_after_N and self.N_lost > self.no_prewarp_after_N:
            self.last_good_H2init = np.eye(3)
        meta.last_good_H2init = self.last_good_H2init.copy()
        prewarp_H = self.last_good_H2init
        prewarped = warp_linear_u8(img, prewarp_H)
        pw_mask = torch.from_numpy(warp_linear(np.ones(img.shape[:2]), prewarp_H) > 0)

        tc, cur, w = self._flow(self.template_img, prewarped)
        tc, cur, w = self._mask_coords(tc, cur, w, pw_mask)
        tc = tc.float()
        tc, cur, w = self._subsample(tc, cur, w)
        Hpw = self.H_estimator(cur.t()[None], tc.t()[None], w).float()
        H_global = hfit_ref.compose_H(prewarp_H, Hpw[0].numpy())
        meta.H_global_cur2init = H_global.copy()
        ok = hfit_ref.redet_success(Hpw, tc, cur)
        if self._n_tracked in self.force_fail:                   # (config-level hook of the golden runs, gen_golden.py)
            ok = False
        self._n_tracked += 1
        if ok:
            H_cur = H_global
            self.lost, self.N_lost = False, 0
        else:
            self.lost = True
            self.N_lost += 1
            if self.no_local_H:
                H_cur = H_global
            else:
                pc, cc, w = self._flow(self.prev_img, img)
                pm = torch.from_numpy(warp_nearest(self.np_template_mask, np.linalg.inv(self.prev_H2init)) > 0)
                keep = pm[pc[1], pc[0]]                         # TRK:314-327
                pc, cc, w = pc[:, keep], cc[:, keep], w[:, keep]
                pc, cc, w = self._subsample(pc, cc, w)
                try:
                    Hf = self.H_estimator(cc.t()[None], pc.float().t()[None], w)[0].numpy()
                    H_local = hfit_ref.compose_H(Hf, self.prev_H2init)
                except Exception:
                    H_local = self.prev_H2init
                meta.H_local_cur2init = H_local.copy()
                H_cur = H_local
        self.prev_img = img.copy()
        self.prev_H2init = H_cur.copy()
        if not self.lost:
            self.last_good_H2init = H_cur.copy()
        meta.lost, meta.N_lost, meta.global_H_success = self.lost, self.N_lost, ok
        if self.downscale:                                       # TRK:280-283
            k = self.downscale
            H_cur = hfit_ref.compose_H(np.diag([1.0 / k, 1.0 / k, 1.0]), H_cur, np.diag([float(k), float(k), 1.0]))
        return H_cur, meta

    def _mask_coords(self, tc, cur, w, pw_mask):                 # TRK:287-312
        in_t = self.template_mask[tc[1], tc[0]]
        H, W = pw_mask.shape
        ci = cur.round().long()
        oob = torch.any(cur < 0, dim=0) | (ci[0] >= W) | (ci[1] >= H)
        in_pw = ~oob
        in_pw[in_pw.clone()] = pw_mask[ci[1, in_pw], ci[0, in_pw]]
        keep = in_t & in_pw
        return tc[:, keep], cur[:, keep], w[:, keep]
