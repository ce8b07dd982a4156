"""Micro-benchmark / ablation of the pre-split correlation GEMM (woft_corr_gemm_bf16) at the 1080p level-0 shape.
ABL bits (woft_set_tuning key 2): 1 no global stores, 2 no epilogue, 4 no operand DMA after the prologue,
8 no LDS reads / MFMA."""
import math
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from woft_amd import _lib, ops
from tools.bench_conv import bench


def main():
    hf, wf = (int(v) for v in os.environ.get("HW", "135,240").split(","))
    P = hf * wf
    n = ops.tiled_dims(hf, wf)[2]
    terms = int(os.environ.get("TERMS", "3"))
    mk = lambda rows: (torch.randn(ops._round_up(rows, 256), 256, device="cuda") * 0.1)
    a, b = mk(P), mk(n)
    if terms == 3:
        sa, sb = (torch.zeros(x.shape[0], 512, dtype=torch.bfloat16, device="cuda") for x in (a, b))
        ops.split_bf16_lines(a, sa)
        ops.split_bf16_lines(b, sb)
    else:
        sa, sb = (torch.zeros_like(x, dtype=torch.bfloat16) for x in (a, b))
        ops.split_bf16(a, sa, None)
        ops.split_bf16(b, sb, None)
    vol = torch.zeros(P, n, device="cuda")
    lib = _lib.load()
    flops = 2.0 * P * n * 256 * terms
    for abl in [int(v) for v in os.environ.get("ABL", "0,1,2,4,8,6,12").split(",")]:
        lib.woft_set_tuning(2, abl)
        ms = bench(lambda: ops.corr_gemm_bf16(sa, sb, P, n, 1 / 16.0, vol, terms))
        print(f"abl {abl:2d}: {ms * 1e3:8.1f} us  {flops / ms / 1e9:7.1f} TF/s (of the full work)  "
              f"write {P * n * 4 / ms / 1e9:6.2f} TB/s", flush=True)
    lib.woft_set_tuning(2, 0)


if __name__ == "__main__":
    main()
