"""Micro-benchmark of the volume-free lookup (woft_corr_lookup_otf) at 1080p feature size, smooth random flow."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from woft_amd import ops
from tools.bench_conv import bench


def main():
    hf, wf, c = 135, 240, 256
    P = hf * wf
    flow = float(os.environ.get("FLOW", "3.0"))
    mk = lambda n: torch.randn(n, c, device="cuda") * 0.3

    def split(t):
        o = torch.zeros(t.shape[0], 2 * c, dtype=torch.bfloat16, device="cuda")
        ops.split_bf16_lines(t, o)
        return o
    f1s = split(mk(P))
    f2s, dims = [], []
    h, w = hf, wf
    for _ in range(4):
        f2s.append(split(mk(h * w)))
        dims.append((h, w))
        h, w = h // 2, w // 2
    idx = torch.arange(P, device="cuda")
    base = torch.stack([idx % wf, idx // wf], 1).float()
    # smooth flow: a global shift + small noise (SMOOTH=0: independent per pixel)
    noise = (torch.rand(P, 2, device="cuda") * 2 - 1) * (flow if os.environ.get("SMOOTH", "1") == "0" else 0.3)
    coords = (base + flow + noise).contiguous()
    out = torch.zeros(P, 352, device="cuda")
    lp = ops.make_lookup_otf_params(f1s, f2s, dims, hf, wf, c, coords, out, 4, 3)
    ms = bench(lambda: ops.run_lookup_otf(lp), reps=20)
    print(f"volume-free lookup {hf}x{wf}, flow {flow}: {ms * 1e3:8.1f} us")


if __name__ == "__main__":
    main()
