"""Odd input sizes through the operator (RAFT replicate padding): default split-bf16 / volume-free path vs the exact
fp32 / volume path of the same engine -- flow EPE and sigmoid(weights) differences."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np

from woft_amd import synth
from woft_amd.config import Config
from woft_amd.flow_provider import RAFTWrapper

sd = synth.make_state_dict(seed=9)
for (H, W) in [(137, 203), (200, 312), (481, 643), (721, 1283)]:
    a = synth.make_template(H, W, seq_id=H)
    b = synth.make_frame(a, 3)
    res = {}
    for prec in ("fp32", "bf16x3"):
        c = Config(); c.of_class = RAFTWrapper; c.raft_type = "weighted"; c.class_params = Config()
        c.model = sd; c.iters = 6; c.padding_mode = "RAFT"; c.precision = prec
        p = RAFTWrapper(c)
        res[prec] = p.compute_flow(a, b, mode="flow", numpy_out=True, do_sigmoid=True)
        del p
    e = np.sqrt(((res["fp32"][0] - res["bf16x3"][0]) ** 2).sum(0))
    dw = np.abs(res["fp32"][1] - res["bf16x3"][1]).max()
    print(f"{H}x{W}: EPE bf16x3/otf vs fp32/volume mean {e.mean():.2e} max {e.max():.2e}; max |d sigma(w)| {dw:.2e}; "
          f"finite {np.isfinite(res['bf16x3'][0]).all()}")
