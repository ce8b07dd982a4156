"""Homography estimators with the reference's operator API
(/root/reference/pytracking/utils/least_squares_H.py): same function names, argument meaning,
return shapes and AssertionError behaviour; the arithmetic runs in the HIP fit kernels
(csrc/hfit.hip) on the device the points live on.
"""
import numpy as np
import torch

from . import ops
from .probe import recorder


class _Probe:
    """Sentinel passed through user re-weighting callables to recognise the built-in losses."""


def IRLSq_L1(residuals, eps=1e-8):
    """least_squares_H.py:268-269."""
    if isinstance(residuals, _Probe):
        return ("l1", 0.0, eps)
    return 1 / (torch.abs(residuals) + eps)


def IRLSq_Huber(residuals, k=1, eps=1e-8):
    """L2 up to +-k, then L1 (least_squares_H.py:272-277)."""
    if isinstance(residuals, _Probe):
        return ("huber", float(k), eps)
    abs_res = torch.abs(residuals)
    weights = 1 / (abs_res + eps)
    weights[abs_res < k] = 1
    return weights


def _check(points1, points2):
    if points1.shape != points2.shape:
        raise AssertionError(points1.shape)
    if not (len(points1.shape) >= 1 and points1.shape[-1] == 2):
        raise AssertionError(points1.shape)
    if points1.shape[1] < 4:
        raise AssertionError(points1.shape)


def _operands(points1, points2, weights, b):
    pa = points1[b].float().contiguous()
    pb = points2[b].float().contiguous()
    w = weights[b].float().reshape(-1).contiguous() if weights is not None else None
    if w is not None and w.numel() != pa.shape[0]:
        raise AssertionError(weights.shape)
    return pa, pb, w


def _fit(points1, points2, weights, reweight, huber_k, n_irls):
    if not points1.is_cuda:
        # The reference's QR estimator takes tensors of any device (least_squares_H.py:142-210 has no device check; only the IRLS
        # variant asserts, :292-293 -- and so does find_homography_IRLSq_QR below).  Host tensors are copied to the HIP device,
        # fitted by the same kernel and the result handed back on the caller's device: there is no CPU solver here.
        dev = torch.device("cuda")
        out = _fit(points1.to(dev), points2.to(dev), None if weights is None else weights.to(dev), reweight, huber_k, n_irls)
        return out.to(points1.device)
    B = points1.shape[0]
    out = torch.empty(B, 3, 3, dtype=torch.float32, device=points1.device)
    status = torch.zeros(B, dtype=torch.int32, device=points1.device)
    for b in range(B):
        pa, pb, w = _operands(points1, points2, weights, b)
        ops.hfit(pa, pb, w, out[b].view(9), status[b:b + 1], reweight=reweight, huber_k=huber_k, n_irls=n_irls)
    return out


def _fit_callable(points1, points2, weights, reweighting_fn, n_iter):
    """IRLS with an arbitrary loss (least_squares_H.py:323-337): every pass is one device-side re-weighted solve
    (woft_hfit_step) that also returns the residuals A x - b of its solution on the weighted system; the user's
    callable sees them as a (1, 2N, 1) device tensor -- two consecutive rows per correspondence, the reference's
    row order -- and its result, square-rooted, re-weights the rows of the next pass."""
    B, N = points1.shape[0], points1.shape[1]
    dev = points1.device
    out = torch.empty(B, 3, 3, dtype=torch.float32, device=dev)
    status = torch.zeros(B, dtype=torch.int32, device=dev)
    res = torch.empty(2 * N, dtype=torch.float32, device=dev)
    for b in range(B):
        pa, pb, w = _operands(points1, points2, weights, b)
        rew = None
        for it in range(n_iter + 1):
            ops.hfit_step(pa, pb, w, rew, it == 0, res, out[b].view(9), status[b:b + 1])
            rew = torch.sqrt(reweighting_fn(res.view(1, 2 * N, 1))).to(dtype=torch.float32, device=dev)
            rew = rew.expand(1, 2 * N, 1).reshape(-1).contiguous()
    return out


def find_homography_nonhomogeneous_QR(points1, points2, weights=None):
    """Weighted inhomogeneous DLT, h33 = 1 (least_squares_H.py:142-210).
    points (B,N,2), weights (B,N) -> (B,3,3) mapping points1 -> points2."""
    rec = recorder()
    if rec is not None:                  # (woft_amd.probe: the tracker is finding out what a config's estimator does)
        return rec.fit("lsq", points1, points2, weights)
    _check(points1, points2)
    return _fit(points1, points2, weights, 0, 0.0, 0)


def find_homography_IRLSq_QR(points1, points2, weights=None, reweighting_fn=IRLSq_L1, n_iter=5):
    """IRLS m-estimator (least_squares_H.py:280-346): n_iter + 1 solves, per-row re-weighting
    sqrt(reweighting_fn(A x - b)) from the weighted algebraic residual.  Losses built from IRLSq_L1 / IRLSq_Huber
    (what the reference's configs use, configs/..._wIRLSq.py:24-31) run in ONE launch; any other callable is
    driven pass by pass on device tensors (_fit_callable)."""
    rec = recorder()
    if rec is not None:
        return rec.fit("irls", points1, points2, weights, reweighting_fn, n_iter)
    _check(points1, points2)
    if not points1.is_cuda:
        raise AssertionError("correspondences should be on GPU")
    try:
        kind = reweighting_fn(_Probe())
    except Exception:
        kind = None
    if isinstance(kind, tuple) and len(kind) == 3 and kind[2] == 1e-8:
        if kind[0] == "l1":
            return _fit(points1, points2, weights, 1, 0.0, n_iter)
        if kind[0] == "huber":
            return _fit(points1, points2, weights, 2, kind[1], n_iter)
    return _fit_callable(points1, points2, weights, reweighting_fn, n_iter)


def torch_proj_errors(GT_H, pts_A, pts_B):
    """L2 distance between H * pts_A and pts_B (least_squares_H.py:474-489).
    GT_H (B,3,3); pts (B,2,N) -> (B,N)."""
    rec = recorder()
    if rec is not None:
        return rec.proj(GT_H, pts_A, pts_B)
    ones = torch.ones_like(pts_A[:, :1])
    proj = torch.matmul(GT_H, torch.cat([pts_A, ones], dim=1))
    z = proj[:, 2:3]
    scale = torch.where(z.abs() > 1e-8, 1.0 / (z + 1e-8), torch.ones_like(z))
    return torch.sqrt(torch.square(scale * proj[:, :2] - pts_B).sum(dim=1))


def compose_H(*Hs):
    """Compose homographies: compose_H(H1, ..., Hk) = normalise(Hk ... H1)
    (/root/reference/pytracking/utils/geom_utils.py:365-373)."""
    for H in Hs:
        if H is None:
            return None
    result = np.eye(3)
    for H in Hs:
        result = np.dot(H, result)
    return result / result[2, 2]
