"""Micro-benchmark of the correlation lookup at full-size shapes (default 1080p: P = 135 x 240).
Reports the average launch time and the achieved algorithmic bandwidth (2896 B per pixel)."""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from woft_amd import ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hf", type=int, default=135)
    ap.add_argument("--wf", type=int, default=240)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--flow", type=float, default=5.0, help="uniform +-flow magnitude at 1/8 res")
    ap.add_argument("--storage", default="fp32", choices=["fp32", "bf16"], help="element type of the volume")
    a = ap.parse_args()
    hf, wf = a.hf, a.wf
    P = hf * wf
    vols, dims = [], []
    h, w = hf, wf
    for l in range(4):
        vols.append(torch.randn(P, ops.tiled_dims(h, w)[2], device="cuda").to(torch.bfloat16 if a.storage == "bf16" else torch.float32))
        dims.append((h, w))
        h, w = h // 2, w // 2
    idx = torch.arange(P, device="cuda")
    coords = torch.stack([idx % wf, idx // wf], 1).float() + (torch.rand(P, 2, device="cuda") * 2 - 1) * a.flow
    out = torch.zeros(P, 352, device="cuda")
    lp = ops.make_lookup_params(vols, dims, coords.contiguous(), out, 4)
    import os
    lp.ablate = int(os.environ.get("LOOKUP_ABL", "0"))     # developer ablation bits: 1 = no volume reads, 2 = no output
    for _ in range(3):
        ops.run_lookup(lp)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.reps)]
    for s, e in ev:
        s.record()
        ops.run_lookup(lp)
        e.record()
    torch.cuda.synchronize()
    ms = sorted(s.elapsed_time(e) for s, e in ev)
    med = ms[len(ms) // 2]
    algo = (2896 if a.storage == "fp32" else 2096) * P
    print(f"lookup {hf}x{wf} ({a.storage} volume, 4x4 tiles): median {med*1e3:.1f} us  min {ms[0]*1e3:.1f} us  -> {algo/med/1e6:.0f} GB/s algorithmic "
          f"({algo/med/1e6/8000*100:.1f}% of 8 TB/s)")


if __name__ == "__main__":
    main()
