"""Micro-benchmark of the implicit-GEMM conv kernel on the layer shapes that dominate a 1080p frame."""
import argparse
import os
import math
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from woft_amd import ops


def bench(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in ev:
        s.record()
        fn()
        e.record()
    torch.cuda.synchronize()
    ms = sorted(s.elapsed_time(e) for s, e in ev)
    return ms[len(ms) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="bf16x3")
    ap.add_argument("--tiles", default=None, help="e.g. 128,128")
    ap.add_argument("--halo", type=int, default=None)
    a = ap.parse_args()
    tiles = tuple(int(t) for t in a.tiles.split(",")) if a.tiles else None
    hf, wf = 135, 240
    P = hf * wf
    cases = []

    def conv_case(name, n, h, w, cin, cout, kh, kw, x2c=0):
        wt = torch.randn(cout, cin + x2c, kh, kw) * 0.05
        pc = ops.pack_conv(wt, torch.zeros(cout), padding=(kh // 2, kw // 2))
        x = ops.new_act(n, h, w, cin, zero=False)
        x.t.normal_()
        x2 = None
        if x2c:
            x2 = ops.new_act(n, h, w, x2c)
            x2.t.normal_()
        out = ops.new_act(n, h, w, cout, cs=ops._round_up(cout, 4), zero=True)
        p = ops.conv_params(x, pc, out, x2=x2, c_split=cin if x2c else 0, epi=ops._lib.EPI_RELU,
                            precision=a.precision, tiles=tiles, halo=a.halo)
        p.out_w = int(os.environ.get("ABL", "0"))
        flops = 2.0 * n * h * w * (cin + x2c) * kh * kw * cout
        cases.append((name + f" [halo {p.halo}]", lambda: ops.run_conv(p), flops, (p.tile_m, p.tile_n)))

    if os.environ.get("ONLYVOL"):
        conv_case = lambda *a, **k: None
    conv_case("gru zr 1x5 (384->256)", 1, hf, wf, 128, 256, 1, 5, x2c=256)
    conv_case("gru q 1x5 (384->128)", 1, hf, wf, 128, 128, 1, 5, x2c=256)
    conv_case("convc2 3x3 (256->192)", 1, hf, wf, 256, 192, 3, 3)
    conv_case("convc1 1x1 (352->256)", 1, hf, wf, 352, 256, 1, 1)
    conv_case("flow head 3x3 (128->256)", 1, hf, wf, 128, 256, 3, 3)
    conv_case("wh 3x3 128->128 on 9x9 patches", P, 9, 9, 128, 128, 3, 3)
    conv_case("fnet l1 3x3 64->64 @1/2", 1, 540, 960, 64, 64, 3, 3)
    conv_case("fnet l3 3x3 128->128 @1/8", 1, hf, wf, 128, 128, 3, 3)
    # volume level 0
    f1 = ops.new_act(1, hf, wf, 256)
    f1.t.normal_()
    n = ops.tiled_dims(hf, wf)[2]
    rows = torch.randn(ops._round_up(n, 128), 256, device="cuda")
    hi, lo = torch.zeros_like(rows, dtype=torch.bfloat16), torch.zeros_like(rows, dtype=torch.bfloat16)
    ops.split_bf16(rows, hi, lo)
    vol = torch.zeros(P, n, device="cuda")
    pv = ops.corr_volume(f1, rows, n, vol, 1 / 16.0, precision=a.precision, f2_hi=hi, f2_lo=lo)
    if tiles:
        pv.tile_m, pv.tile_n = tiles
    cases.append(("volume L0 (PxP, K=256)", lambda: ops.run_conv(pv), 2.0 * P * n * 256, (pv.tile_m, pv.tile_n)))

    if os.environ.get("NOSTORE"):
        pv.out_w = -12345
    # plain copy kernels as a write-bandwidth yardstick for the 4.2 GB volume
    big = torch.empty(P, n, device="cuda")
    cases.append(("torch fill 4.2 GB", lambda: big.fill_(1.0), 0.0, (0, 0)))
    cases.append(("torch copy 4.2 GB", lambda: big.copy_(vol), 0.0, (0, 0)))
    tot = 0.0
    only = os.environ.get("ONLY")
    if only:
        cases = [c for c in cases if only in c[0]]
    tune = os.environ.get("TUNE1")                       # A/B of a developer knob (woft_set_tuning key 1)
    for name, fn, flops, t in cases:
        ms = bench(fn)
        tot += ms
        extra = ""
        if tune:
            ops._lib.load().woft_set_tuning(1, int(tune))
            extra = f"   tuned({tune}) {bench(fn)*1e3:9.1f} us"
            ops._lib.load().woft_set_tuning(1, 0)
        print(f"{name:44s} tiles {t}  {ms*1e3:9.1f} us   {flops/ms/1e9:8.1f} TFLOP/s (useful){extra}")
    print(f"sum {tot:.3f} ms  [{a.precision}]")


if __name__ == "__main__":
    main()
