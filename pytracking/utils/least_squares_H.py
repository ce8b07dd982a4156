from woft_amd.homography import (IRLSq_Huber, IRLSq_L1, find_homography_IRLSq_QR,  # noqa: F401
                                 find_homography_nonhomogeneous_QR, torch_proj_errors)
