"""Compatibility shim: the reference's `pytracking.*` import paths, backed by woft_amd (MI355X/HIP).

WOFT_demo.py and reference-style config files import
  pytracking.utils.config.{Config, load_config}
  pytracking.utils.least_squares_H.{find_homography_nonhomogeneous_QR, find_homography_IRLSq_QR, IRLSq_Huber, IRLSq_L1, torch_proj_errors}
  pytracking.optical_flow.raft.RAFTWrapper
  pytracking.tracker.YAOF_tracker_single_control.YAOFTrackerSingleControl
  pytracking.utils.{io, vis_utils, geom_utils, misc}
(SURVEY.md 8b.1).  Every module here is our own code; none of the reference's files is shipped.
"""
