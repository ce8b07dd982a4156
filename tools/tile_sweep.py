"""Tile / kernel choice sweep for the layer shapes of a 1080p frame, all candidates of a layer timed interleaved in
one process (medians of 9; the choice in ops.conv_params is the first column).  python tools/tile_sweep.py [precision] [hf,wf]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from woft_amd import ops


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
    hf, wf = (int(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 else (135, 240)   # 1/8-resolution grid
    layers = [("gru zr 1x5 256->256", 1, hf, wf, 256, 256, 1, 5), ("gru zr 5x1 256->256", 1, hf, wf, 256, 256, 5, 1),
              ("gru q 1x5 256->128", 1, hf, wf, 256, 128, 1, 5), ("convc2 3x3 256->192", 1, hf, wf, 256, 192, 3, 3),
              ("convf2 3x3 128->64", 1, hf, wf, 128, 64, 3, 3), ("conv 3x3 256->126", 1, hf, wf, 256, 126, 3, 3),
              ("fh1 3x3 128->256", 1, hf, wf, 128, 256, 3, 3), ("fnet 3x3 64->64 @1/2", 1, 4 * hf, 4 * wf, 64, 64, 3, 3),
              ("fnet 3x3 96->96 @1/4", 1, 2 * hf, 2 * wf, 96, 96, 3, 3), ("fnet 3x3 128->128 @1/8", 1, hf, wf, 128, 128, 3, 3)]
    cands = [None, (1, 128), (1, 64), (4, 128), (4, 64), (6, 128)]
    for name, n, h, w, cin, cout, kh, kw in layers:
        wt = torch.randn(cout, cin, kh, kw) * 0.05
        pc = ops.pack_conv(wt, torch.zeros(cout), padding=(kh // 2, kw // 2))
        x = ops.new_act(n, h, w, cin, zero=False)
        x.t.normal_()
        out = ops.new_act(n, h, w, cout, cs=ops._round_up(cout, 4), zero=True)
        ps = []
        for c in cands:
            try:
                if c is None:
                    p = ops.conv_params(x, pc, out, epi=ops._lib.EPI_RELU, precision=prec)
                    tag = f"auto(h{p.halo},n{p.tile_n})"
                else:
                    if c[1] == 128 and pc.cout_pad % 128:
                        continue
                    p = ops.conv_params(x, pc, out, epi=ops._lib.EPI_RELU, precision=prec, tiles=(128, c[1]), halo=c[0])
                    if c[1] == 64:
                        p.cout_pad = ops._round_up(cout, 64)
                    tag = f"h{c[0]},n{c[1]}"
                ops.run_conv(p)
                torch.cuda.synchronize()
                ps.append((tag, p))
            except Exception as e:                       # (combination not instantiated)
                continue
        times = {t: [] for t, _ in ps}
        for _ in range(9):
            for t, p in ps:
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                ops.run_conv(p)
                e.record()
                torch.cuda.synchronize()
                times[t].append(s.elapsed_time(e) * 1e3)
        print(f"{name:26s} " + "  ".join(f"{t} {sorted(v)[4]:6.1f}" for t, v in times.items()))


if __name__ == "__main__":
    main()
