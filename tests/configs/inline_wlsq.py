"""A tracker config in the REFERENCE's form: estimator, subsampler and re-detection test written inline as plain functions
that call the library through the reference's import paths (what configs/YAOFT_single_control_repRAFT_sub500_noreliableinl_wLSq.py
does), no woft_amd import, no tags, and a flow config without a `precision` key (get_config(precision=...) adds one for the
bench's second pass).  woft_amd.probe must recognise the three callables and give this config the device back end."""
from pathlib import Path

import numpy as np
import torch

from pytracking.tracker.YAOF_tracker_single_control import YAOFTrackerSingleControl
from pytracking.utils.config import Config, load_config
from pytracking.utils.least_squares_H import find_homography_nonhomogeneous_QR, torch_proj_errors


def redetected(H_prewarped2init, template_coords, cur_pw_coords, weights):
    errors = torch_proj_errors(H_prewarped2init, cur_pw_coords.unsqueeze(0), template_coords.unsqueeze(0))
    fraction = (errors <= 5).float().mean()
    return fraction > 0.2


def fit_homography(pts_A, pts_B, weights=None):
    return find_homography_nonhomogeneous_QR(pts_A, pts_B, weights=weights)


def draw_500(coords_a, coords_b, weights):
    assert coords_a.shape == coords_b.shape
    n = coords_a.shape[1]
    assert weights.shape == (1, n)
    if n <= 500:
        return coords_a, coords_b, weights
    picked = np.zeros(n, dtype=bool)
    u = torch.quasirandom.SobolEngine(dimension=1).draw(500).cpu().numpy().flatten()
    picked[np.round(n * u).astype(np.int32)] = True
    return coords_a[:, picked], coords_b[:, picked], weights[:, picked]


def get_config(precision=None):
    root = Path(__file__).resolve().parents[2]
    conf = Config()
    conf.tracker_class = YAOFTrackerSingleControl
    conf.flow_config = load_config(root / 'pytracking' / 'optical_flow' / 'configs' / 'v2_SNOB_large_g05_RAFT.py')
    del conf.flow_config.precision             # a reference flow config has no such key: the provider's built-in default decides
    if precision is not None:
        conf.flow_config.precision = precision
    conf.flow_config.weights_postprocessing_fn = None
    conf.flow_numpy_out = False
    conf.H_estimator = fit_homography
    conf.redet_success_fn = redetected
    conf.subsampler_fn = draw_500
    conf.pw_mask = True
    conf.no_prewarp_after_N = 10
    return conf
