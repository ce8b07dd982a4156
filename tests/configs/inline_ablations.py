"""Two of the reference's ablation configs in their inline form, selected by get_config(kind):
  "plain": configs/..._noreliableinl_plainLSq.py -- the estimator hands the library weights=None; the usual inlier test
  "never": configs/..._neverwarp_wLSq.py         -- weighted LSq; the re-detection test is `return False`"""
from pathlib import Path

import numpy as np
import torch

from pytracking.tracker.YAOF_tracker_single_control import YAOFTrackerSingleControl
from pytracking.utils.config import Config, load_config
from pytracking.utils.least_squares_H import find_homography_nonhomogeneous_QR, torch_proj_errors


def inlier_test(H_prewarped2init, template_coords, cur_pw_coords, weights):
    errs = torch_proj_errors(H_prewarped2init, cur_pw_coords[None], template_coords[None])
    return torch.mean((errs <= 5).float()) > 0.2


def never(H_prewarped2init, template_coords, cur_pw_coords, weights):
    return False


def plain_fit(pts_A, pts_B, weights=None):
    return find_homography_nonhomogeneous_QR(pts_A, pts_B, weights=None)


def weighted_fit(pts_A, pts_B, weights=None):
    return find_homography_nonhomogeneous_QR(pts_A, pts_B, weights=weights)


def subsampler(coords_a, coords_b, weights):
    n = coords_a.shape[1]
    if 500 >= n:
        return coords_a, coords_b, weights
    keep = np.zeros(n) > 0
    keep[np.round(n * torch.quasirandom.SobolEngine(dimension=1).draw(500).cpu().numpy().flatten()).astype(np.int32)] = True
    return coords_a[:, keep], coords_b[:, keep], weights[:, keep]


def get_config(kind="plain"):
    root = Path(__file__).resolve().parents[2]
    conf = Config()
    conf.tracker_class = YAOFTrackerSingleControl
    conf.flow_config = load_config(root / 'pytracking' / 'optical_flow' / 'configs' / 'v2_SNOB_large_g05_RAFT.py')
    conf.flow_config.weights_postprocessing_fn = None
    conf.flow_numpy_out = False
    conf.H_estimator = plain_fit if kind == "plain" else weighted_fit
    conf.redet_success_fn = inlier_test if kind == "plain" else never
    conf.subsampler_fn = subsampler
    conf.pw_mask = True
    conf.no_prewarp_after_N = 10
    return conf
