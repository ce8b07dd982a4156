"""Per-layer time table of one 1080p frame: every entry of the engine's launch programs timed on its own
(median of 5, HIP events).  python tools/layer_times.py [--precision bf16x3] [--hw 1080,1920]"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from woft_amd import ops, synth
from woft_amd.engine import RaftEngine


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="bf16x3")
    ap.add_argument("--hw", default="1080,1920")
    ap.add_argument("--corr", default=None, help="volume | otf (default: otf)")
    a = ap.parse_args()
    h, w = (int(v) for v in a.hw.split(","))
    corr = a.corr or "otf"
    eng = RaftEngine(synth.make_state_dict(seed=7), small=False, weighted=True, precision=a.precision, corr=corr)
    plan = eng.plan(h, w) if hasattr(eng, "plan") else None
    if plan is None:
        raise SystemExit("engine has no plan()")
    img = torch.randint(0, 255, (h, w, 3), dtype=torch.uint8, device="cuda")
    plan.load_image(0, img, 0, 0)
    plan.load_image(1, img, 0, 0)
    plan.encode_source()
    plan.flow(2, (0, 0), h, w, flow_up=torch.zeros(2, h, w, device="cuda"), dst=torch.zeros(2, h * w, device="cuda"),
              wout=torch.zeros(1, h * w, device="cuda"))
    torch.cuda.synchronize()
    progs = [("f_dst", plan.prog_f_dst, 1), ("c_src", plan.prog_c_src, 0), ("volume", plan.prog_volume, 1),
             ("iter", plan.prog_iter, 12), ("mask", [("conv", p) for p in plan.prog_mask], 1),
             ("wh", [("conv", p) for p in plan.prog_wh], 1)]
    total = 0.0
    if getattr(plan, "wh0_direct", False) and not getattr(plan, "wh0_fused", False):
        from woft_amd import _lib
        lib = _lib.load()
        n = eng.spec.nwin
        progs.append(("wh0", [("call", lambda: _lib.check(lib.woft_wh_conv0(
            _lib.ptr(plan.corr.t), plan.corr.cs, _lib.ptr(plan.wmean), plan.P, n, _lib.ptr(plan.wh0_t),
            _lib.ptr(eng.wh0.bias), _lib.ptr(plan.a1.t), None, _lib.stream_ptr()), "wh_conv0"))], 1))
    if getattr(plan, "wh0_direct", False) and not getattr(plan, "wh_fused", False):
        from woft_amd import _lib
        lib = _lib.load()
        n = eng.spec.nwin
        progs.append(("whred", [("call", lambda: _lib.check(lib.woft_wh_reduce(
            _lib.ptr(plan.a1.t), 128, n * n, _lib.ptr(eng.wh6_w), eng.wh6_b, plan.P, _lib.ptr(plan.wlow),
            _lib.stream_ptr()), "wh_reduce"))], 1))
    for name, prog, mult in progs:
        sub = 0.0
        for idx, ent in enumerate(prog):
            kind, arg = ent[0], ent[1]
            ts = []
            for _ in range(5):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                arg() if kind == "call" else plan.run([(kind, arg)])
                e.record()
                torch.cuda.synchronize()
                ts.append(s.elapsed_time(e))
            t = sorted(ts)[2] * 1e3
            sub += t
            desc = kind
            if kind in ("conv", "conv2"):
                desc = " + ".join(f"conv {p.taps_y}x{p.taps_x} s{p.stride} {p.n_img}x{p.h}x{p.w} cin {p.cin_pad} -> {p.cout}"
                                  f" tile {p.tile_m}x{p.tile_n} halo {p.halo} epi {p.epi}{' flat' if p.flat else ''}{' mx' if p.precision == 4 else ''}"
                                  for p in (arg if kind == "conv2" else [arg]))
            print(f"  {name:7s}[{idx:2d}] {t:9.1f} us  {desc}")
        print(f"{name}: {sub / 1e3:.3f} ms x {mult}")
        total += sub * mult
    print(f"sum over a tracked frame (source cached): {total / 1e3:.2f} ms")


if __name__ == "__main__":
    main()
