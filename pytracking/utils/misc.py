"""Feature-map coordinate grids (the reference's utils/misc.py:27-68 surface): (2, H*W) x,y, row-major."""
import numpy as np
import torch


def get_featuremap_coords(feature_map):
    H, W = feature_map if isinstance(feature_map, tuple) else feature_map.shape[-2:]
    idx = np.arange(H * W)
    return np.stack((idx % W, idx // W), axis=0)


def torch_get_featuremap_coords(feature_map, device=None, keep_shape=False):
    if isinstance(feature_map, tuple):
        H, W = feature_map
        assert device is not None
    else:
        H, W = feature_map.shape[-2:]
        device = device or feature_map.device
    idx = torch.arange(H * W, device=device)
    xy = torch.stack([idx % W, torch.div(idx, W, rounding_mode="floor")], dim=0)
    return xy.reshape(2, H, W) if keep_shape else xy
