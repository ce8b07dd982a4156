"""WOFT on 2x-downscaled inputs with RAFT replicate padding (the reference's WOFT_downscale_2x variant)."""
from pathlib import Path

from pytracking.utils.config import load_config


def get_config():
    conf = load_config(Path(__file__).resolve().parent / 'WOFT.py')
    conf.flow_config.padding_mode = 'RAFT'
    conf.downscale_inputs = 2
    return conf
