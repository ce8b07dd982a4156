"""conv_regb_kernel (halo 8: weights streamed global -> registers) against conv_halo_bf16_kernel (halo 1) on the
update-block layer shapes: bit equality (same products, same accumulation order) and timing."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from woft_amd import ops, _lib
from tools.bench_conv import bench


def main():
    hf, wf = 135, 240
    rows = []
    for prec in ("bf16x3", "bf16"):
        for name, cin, x2c, cout, kh, kw, epi in (
                ("gru zr 1x5", 128, 128, 256, 1, 5, _lib.EPI_RELU), ("gru q 5x1", 128, 128, 128, 5, 1, _lib.EPI_TANH),
                ("convc2 3x3", 256, 0, 192, 3, 3, _lib.EPI_RELU), ("fh1 3x3", 128, 0, 256, 3, 3, _lib.EPI_RELU),
                ("conv 3x3 256->126", 256, 0, 126, 3, 3, _lib.EPI_RELU), ("convf2 3x3 128->64", 128, 0, 64, 3, 3, _lib.EPI_RELU),
                ("convc1 1x1 352->256", 352, 0, 256, 1, 1, _lib.EPI_RELU), ("mask 1x1 256->576", 256, 0, 576, 1, 1, _lib.EPI_LINEAR)):
            wt = torch.randn(cout, cin + x2c, kh, kw) * 0.05
            pc = ops.pack_conv(wt, torch.randn(cout) * 0.1, padding=(kh // 2, kw // 2))
            x = ops.new_act(1, hf, wf, cin)
            x.t.normal_()
            x2 = None
            if x2c:
                x2 = ops.new_act(1, hf, wf, x2c)
                x2.t.normal_()
            outs, ts, tl = [], [], []
            for halo, tiles in (((0 if kh * kw == 1 else 1), None), (8, None), (8, (128, 64))):
                out = ops.new_act(1, hf, wf, cout, cs=ops._round_up(cout, 4), zero=True)
                p = ops.conv_params(x, pc, out, x2=x2, c_split=cin if x2c else 0, epi=epi, precision=prec, halo=halo,
                                    tiles=tiles)
                ops.run_conv(p)
                torch.cuda.synchronize()
                outs.append(out.t.clone())
                ts.append(bench(lambda: ops.run_conv(p), reps=20) * 1e3)
                tl.append((p.halo, p.tile_n))
            eq = [bool(torch.equal(outs[0], o)) for o in outs[1:]]
            d = [float((outs[0] - o).abs().max()) for o in outs[1:]]
            flops = 2.0 * hf * wf * (cin + x2c) * kh * kw * cout * (3 if prec == "bf16x3" else 1)
            print(f"{prec:7s} {name:20s} base{tl[0]} {ts[0]:7.1f} us | regb{tl[1]} {ts[1]:7.1f} us eq={eq[0]} d={d[0]:.1e} | "
                  f"regb{tl[2]} {ts[2]:7.1f} us eq={eq[1]} d={d[1]:.1e} | issue {flops / min(ts) / 1e6 / 2500:.2f} of peak",
                  flush=True)


if __name__ == "__main__":
    main()
