"""Building blocks for tracker config files: the estimator / subsampler / re-detection callables the
reference's configs define inline (pytracking/configs/YAOFT_single_control_repRAFT_sub500_*.py),
provided once so that config modules stay declarative.  Each factory returns a plain callable with
the signature the tracker expects, tagged with `.woft_spec` so the tracker can recognise it and run
the fused device-side path instead of calling it."""
import numpy as np
import torch

from .homography import (IRLSq_Huber, IRLSq_L1, find_homography_IRLSq_QR, find_homography_nonhomogeneous_QR,
                         torch_proj_errors)


def sobol_points(n):
    """First n points of the unscrambled 1-D Sobol sequence, float32 (what
    torch.quasirandom.SobolEngine(dimension=1).draw(n) returns)."""
    return torch.quasirandom.SobolEngine(dimension=1).draw(n).cpu().numpy().flatten()


def sobol_subsampler(to_draw=500):
    """Keep the correspondences whose rank is round(N * u_k), u = 1-D Sobol points
    (configs/..._wLSq.py:31-53): original order, duplicates collapse, no-op when N <= to_draw."""
    def subsampler(coords_a, coords_b, weights):
        assert coords_a.shape == coords_b.shape
        n_pts = coords_a.shape[1]
        assert weights.shape == (1, n_pts)
        if to_draw >= n_pts:
            return coords_a, coords_b, weights
        keep = np.zeros(n_pts) > 0
        keep[np.round(n_pts * sobol_points(to_draw)).astype(np.int32)] = True
        return coords_a[:, keep], coords_b[:, keep], weights[:, keep]
    subsampler.woft_spec = ("sobol", to_draw)
    return subsampler


def redetection_by_inliers(threshold_px=5.0, min_fraction=0.2):
    """Success = more than `min_fraction` of the correspondences re-project within `threshold_px`
    (configs/..._wLSq.py:14-21)."""
    def redet_success_fn(H_prewarped2init, template_coords, cur_pw_coords, weights):
        errs = torch_proj_errors(H_prewarped2init, cur_pw_coords[None], template_coords[None])
        return torch.mean((errs <= threshold_px).float()) > min_fraction
    redet_success_fn.woft_spec = ("inliers", threshold_px, min_fraction)
    return redet_success_fn


def estimator_weighted_lsq():
    """Weighted least-squares homography (configs/..._wLSq.py:24-28)."""
    def find_homography(pts_A, pts_B, weights=None):
        return find_homography_nonhomogeneous_QR(pts_A, pts_B, weights=weights)
    find_homography.woft_spec = ("lsq", 0, 0.0, 0)
    return find_homography


def estimator_irls(loss="huber", k=2.0, n_iter=5):
    """IRLS homography, Huber(k) or L1 loss (configs/..._wIRLSq.py:24-31)."""
    fn = (lambda r: IRLSq_Huber(r, k=k)) if loss == "huber" else IRLSq_L1

    def find_homography(pts_A, pts_B, weights=None):
        return find_homography_IRLSq_QR(pts_A, pts_B, weights=weights, reweighting_fn=fn, n_iter=n_iter)
    find_homography.woft_spec = ("irls", 2 if loss == "huber" else 1, float(k), n_iter)
    return find_homography
