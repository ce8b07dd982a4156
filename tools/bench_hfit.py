"""H fit on every pixel of a frame (the configs without a subsampler): N = H*W correspondences through the
streaming multi-workgroup fit vs the single-workgroup kernel, weighted LSq and IRLS (6 solves)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from woft_amd import _lib, ops
from tools.bench_conv import bench


def main():
    lib = _lib.load()
    for (H, W) in ((1080, 1920), (2160, 3840), (128, 160), (64, 64), (32, 64)):
        n = H * W
        idx = torch.arange(n, device="cuda")
        a = torch.stack([idx % W, idx // W], 1).float().contiguous()
        b = (a * 1.01 + torch.tensor([3.0, -2.0], device="cuda") + torch.randn(n, 2, device="cuda") * 0.3).contiguous()
        w = (torch.rand(n, device="cuda") * 0.9 + 0.1).contiguous()
        Hd, st = torch.zeros(9, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
        ws = ops.hfit_ws()
        for name, rew, nir in (("weighted LSq", 0, 0), ("IRLS Huber, 6 solves", 2, 5), ("IRLS L1, 6 solves", 1, 5)):
            row = f"{H}x{W} N={n:8d} {name:22s}"
            for label, wsp in (("streaming", ws.data_ptr()), ("one workgroup", None)):
                if wsp is None and n > 3_000_000:
                    continue
                fn = lambda: _lib.check(lib.woft_hfit(a.data_ptr(), b.data_ptr(), w.data_ptr(), n, None, rew, 0.01, nir, wsp,
                                                      Hd.data_ptr(), st.data_ptr(), _lib.stream_ptr()), "woft_hfit")
                ms = bench(fn, reps=7)
                passes = (2 + (nir + 1 if rew else 1))
                row += f" | {label}: {ms * 1e3:9.1f} us ({20.0 * n * passes / ms / 1e6:7.1f} GB/s algorithmic)"
            print(row, flush=True)


if __name__ == "__main__":
    main()
