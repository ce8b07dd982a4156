"""Default WOFT tracker configuration (reference-format config module): weighted-RAFT flow,
weighted least-squares homography on 500 Sobol-picked correspondences, re-detection test at
20 % inliers within 5 px, pre-warp dropped after 10 lost frames."""
from pathlib import Path

from pytracking.tracker.YAOF_tracker_single_control import YAOFTrackerSingleControl
from pytracking.utils.config import Config, load_config
from woft_amd import presets


def get_config():
    here = Path(__file__).resolve().parent
    conf = Config()
    conf.tracker_class = YAOFTrackerSingleControl
    conf.flow_config = load_config(here.parent / 'optical_flow' / 'configs' / 'v2_SNOB_large_g05_RAFT.py')
    conf.flow_config.weights_postprocessing_fn = None
    conf.flow_numpy_out = False
    conf.H_estimator = presets.estimator_weighted_lsq()
    conf.redet_success_fn = presets.redetection_by_inliers(5.0, 0.2)
    conf.subsampler_fn = presets.sobol_subsampler(500)
    conf.pw_mask = True
    conf.no_prewarp_after_N = 10
    return conf
