"""Block-tile sweep of the gather-kernel launches of a 1080p frame (engine launch programs re-timed with other tiles;
timing only -- layers with InstanceNorm statistics need their tile_m for the statistics rows).  python tools/gather_sweep.py"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from woft_amd import ops, synth
from woft_amd.engine import RaftEngine


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
    h, w = 1080, 1920
    eng = RaftEngine(synth.make_state_dict(seed=7), small=False, weighted=True, precision=prec, corr=("volume" if prec == "fp32" else "otf"))
    plan = eng.plan(h, w)
    img = torch.randint(0, 255, (h, w, 3), dtype=torch.uint8, device="cuda")
    plan.load_image(0, img, 0, 0)
    plan.load_image(1, img, 0, 0)
    plan.encode_source()
    plan.flow(2, (0, 0), h, w, flow_up=torch.zeros(2, h, w, device="cuda"), dst=torch.zeros(2, h * w, device="cuda"),
              wout=torch.zeros(1, h * w, device="cuda"))
    torch.cuda.synchronize()
    progs = [("f_dst", plan.prog_f_dst), ("iter", plan.prog_iter), ("mask", [("conv", p) for p in plan.prog_mask])]
    for name, prog in progs:
        for idx, (kind, p) in enumerate(prog):
            if kind != "conv" or p.halo != 0:
                continue
            keep = (p.tile_m, p.tile_n, p.cout_pad)
            res = []
            cands = [(keep[0], keep[1])] + [c for c in ((128, 128), (128, 64), (64, 128), (64, 64)) if c != keep[:2]]
            times = {c: [] for c in cands}
            for c in cands:
                if c[1] == 128 and ops._round_up(p.cout, 128) > keep[2] and keep[1] == 64:
                    times.pop(c)
            for _ in range(7):
                for c in list(times):
                    p.tile_m, p.tile_n = c
                    p.cout_pad = ops._round_up(p.cout, c[1]) if p.stat_sum is None else keep[2]
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    ops.run_conv(p)
                    e.record()
                    torch.cuda.synchronize()
                    times[c].append(s.elapsed_time(e) * 1e3)
            p.tile_m, p.tile_n, p.cout_pad = keep
            print(f"{name}[{idx:2d}] {p.taps_y}x{p.taps_x} s{p.stride} cin {p.cin_pad:3d} -> {p.cout:3d} stats {int(p.stat_sum is not None)}  "
                  + "  ".join(f"{c[0]}x{c[1]} {sorted(v)[3]:6.1f}" for c, v in times.items()))


if __name__ == "__main__":
    main()
