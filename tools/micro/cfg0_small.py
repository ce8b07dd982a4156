"""BASELINE config 0 on the HIP path: one 480x640 pair, RAFT-small, 4 iterations -- flow vs the CPU oracle."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np
import torch

from oracle import raft_ref
from woft_amd import synth
from woft_amd.config import Config
from woft_amd.flow_provider import RAFTWrapper

H, W = 480, 640
sd = synth.make_state_dict(seed=3, small=True, weighted=False)
a = synth.make_template(H, W, seq_id=1)
b = synth.make_frame(a, 2)
ref = raft_ref.compute_flow(sd, a, b, 4, mode="flow", small=True, weighted=False, padding_mode="nopad")[0]
for prec in ("fp32", "bf16x3", "bf16"):
    c = Config(); c.of_class = RAFTWrapper; c.raft_type = "orig"; c.class_params = Config(); c.class_params.small = True
    c.model = sd; c.iters = 4; c.padding_mode = "nopad"; c.precision = prec
    p = RAFTWrapper(c)
    fl, _ = p.compute_flow(a, b, mode="flow", numpy_out=True)
    e = np.sqrt(((fl - np.asarray(ref)) ** 2).sum(0))
    print(f"{prec:7s} corr={p.engine.corr:6s} EPE vs oracle mean {e.mean():.2e} max {e.max():.2e}")
