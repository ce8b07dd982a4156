"""The IRLS variant of tests/configs/inline_wlsq.py, again in the REFERENCE's form (configs/..._wIRLSq.py / ablation_08.py:14-53): the
estimator wraps the library's IRLS with a nested re-weighting function around IRLSq_Huber(k = 2) and the default iteration count."""
from pathlib import Path

import numpy as np
import torch

from pytracking.tracker.YAOF_tracker_single_control import YAOFTrackerSingleControl
from pytracking.utils.config import Config, load_config
from pytracking.utils.least_squares_H import IRLSq_Huber, find_homography_IRLSq_QR, torch_proj_errors


def inlier_test(H_prewarped2init, template_coords, cur_pw_coords, weights):
    e = torch_proj_errors(H_prewarped2init, cur_pw_coords[None], template_coords[None])
    return torch.mean((e <= 5).float()) > 0.2


def robust_fit(pts_A, pts_B, weights=None):
    def huber2(residuals):
        return IRLSq_Huber(residuals, k=2)
    return find_homography_IRLSq_QR(pts_A, pts_B, weights=weights, reweighting_fn=huber2)


def sobol_500(coords_a, coords_b, weights):
    n = coords_a.shape[1]
    assert coords_a.shape == coords_b.shape and weights.shape == (1, n)
    if 500 >= n:
        return coords_a, coords_b, weights
    keep = np.zeros(n) > 0
    keep[np.round(n * torch.quasirandom.SobolEngine(dimension=1).draw(500).cpu().numpy().flatten()).astype(np.int32)] = True
    return coords_a[:, keep], coords_b[:, keep], weights[:, keep]


def get_config():
    root = Path(__file__).resolve().parents[2]
    conf = Config()
    conf.tracker_class = YAOFTrackerSingleControl
    conf.flow_config = load_config(root / 'pytracking' / 'optical_flow' / 'configs' / 'v2_SNOB_large_g05_RAFT.py')
    conf.flow_config.weights_postprocessing_fn = None
    conf.flow_numpy_out = False
    conf.H_estimator = robust_fit
    conf.redet_success_fn = inlier_test
    conf.subsampler_fn = sobol_500
    conf.pw_mask = True
    conf.no_prewarp_after_N = 10
    return conf
