"""Reference-form config of the "plain least squares, always re-detected" ablation (configs/..._alwayswarp_plainLSq.py:12-21): the
estimator hands the library weights=None, the re-detection test is `return True`.  woft_amd.probe recognises both; the device solver
then never evaluates the flow-reliability head (nobody reads a weight)."""
from pathlib import Path

import numpy as np
import torch

from pytracking.tracker.YAOF_tracker_single_control import YAOFTrackerSingleControl
from pytracking.utils.config import Config, load_config
from pytracking.utils.least_squares_H import find_homography_nonhomogeneous_QR


def redet_success_fn(H_prewarped2init, template_coords, cur_pw_coords, weights):
    return True


def find_homography(pts_A, pts_B, weights=None):
    return find_homography_nonhomogeneous_QR(pts_A, pts_B, weights=None)


def subsampler(coords_a, coords_b, weights):
    n = coords_a.shape[1]
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
    conf.H_estimator = find_homography
    conf.redet_success_fn = redet_success_fn
    conf.subsampler_fn = subsampler
    conf.pw_mask = True
    conf.no_prewarp_after_N = 10
    return conf
