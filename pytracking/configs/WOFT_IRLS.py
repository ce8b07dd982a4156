"""WOFT with the IRLS (Huber, k = 2) homography estimator -- the reference's ablation_08 variant."""
from pytracking.utils.config import load_config
from pathlib import Path
from woft_amd import presets


def get_config():
    conf = load_config(Path(__file__).resolve().parent / 'WOFT.py')
    conf.H_estimator = presets.estimator_irls("huber", k=2.0, n_iter=5)
    return conf
