"""Volume-free lookup: the kernel variants against each other at 1080p feature size (bit equality + time + phase stamps).
  python tools/otf_variants.py            (on the GPU box)
Variant byte = bits 8..15 of woft_lookup_otf_params.ablate: 0 = the 2 x 2-wave kernel of rounds 1-5, 1 = four autonomous waves
(window drop after each tile, 6-stage rings), 3 = drop of tile i - 1 in slices pinned between the MFMAs of tile i, 4 / 5 = as 3 with 4 / 3 stages."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from woft_amd import ops
from tools.bench_conv import bench


def inputs(hf, wf, c, flow, smooth, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    P = hf * wf
    mk = lambda n: torch.randn(n, c, device="cuda", generator=g) * 0.3

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
    noise = (torch.rand(P, 2, device="cuda", generator=g) * 2 - 1) * (0.3 if smooth else flow)
    coords = (base + (flow if smooth else 0.0) + noise).contiguous()
    return f1s, f2s, dims, coords


def run(variant, abl, f1s, f2s, dims, hf, wf, c, coords, reps=20, time_it=True):
    out = torch.zeros(hf * wf, 352, device="cuda")
    lp = ops.make_lookup_otf_params(f1s, f2s, dims, hf, wf, c, coords, out, 4, 3)
    lp.ablate = (variant << 8) | abl
    ops.run_lookup_otf(lp)
    torch.cuda.synchronize()
    ms = bench(lambda: ops.run_lookup_otf(lp), reps=reps) if time_it else float("nan")
    return out, ms


def main():
    hf, wf, c = 135, 240, 256
    variants = [int(v) for v in os.environ.get("VARIANTS", "0,1,3,4,5").split(",")]
    cases = [("smooth 3.0", 3.0, True), ("scatter +-3", 3.0, False), ("scatter +-8", 8.0, False), ("smooth shift 40 (clipped boxes)", 40.0, True),
             ("smooth shift -300 (windows leave the map)", -300.0, True)]
    for name, flow, smooth in cases:
        f1s, f2s, dims, coords = inputs(hf, wf, c, flow, smooth)
        ref = None
        for v in variants:
            try:
                out, ms = run(v, 0, f1s, f2s, dims, hf, wf, c, coords)
            except Exception as ex:
                print(f"{name:42s} variant {v}: {type(ex).__name__}: {ex}")
                continue
            if ref is None:
                ref = out
                eq = "reference"
            else:
                same = torch.equal(out[:, :324], ref[:, :324])
                nbad = int((out[:, :324] != ref[:, :324]).sum().item())
                eq = "bit-identical" if same else f"DIFFERENT in {nbad} of {ref[:, :324].numel()} values (max |d| {float((out[:, :324] - ref[:, :324]).abs().max()):.3e})"
            print(f"{name:42s} variant {v}: {ms * 1e3:8.1f} us   {eq}", flush=True)
    # odd sizes: ragged blocks at the right / bottom border
    for hf2, wf2 in ((17, 25), (67, 120)):
        f1s, f2s, dims, coords = inputs(hf2, wf2, c, 2.0, True, seed=3)
        ref = None
        for v in variants:
            try:
                out, _ = run(v, 0, f1s, f2s, dims, hf2, wf2, c, coords, time_it=False)
            except Exception as ex:
                print(f"{hf2}x{wf2} variant {v}: {type(ex).__name__}: {ex}")
                continue
            if ref is None:
                ref = out
            else:
                print(f"{hf2}x{wf2} variant {v}: {'bit-identical' if torch.equal(out[:, :324], ref[:, :324]) else 'DIFFERENT'}", flush=True)
    # ablations and stamps of the autonomous-wave kernel on the smooth case
    f1s, f2s, dims, coords = inputs(hf, wf, c, 3.0, True)
    for v, abl in ((1, 2), (1, 6), (1, 14), (1, 15), (3, 8)):
        try:
            _, ms = run(v, abl, f1s, f2s, dims, hf, wf, c, coords)
            print(f"variant {v} ablate {abl:2d} (1 no stream, 2 no MFMA, 4 no drop, 8 no sampling): {ms * 1e3:8.1f} us", flush=True)
        except Exception as ex:
            print(f"variant {v} ablate {abl}: {type(ex).__name__}: {ex}")
    for v in (1, 3):
        try:
            out, ms = run(v, 16, f1s, f2s, dims, hf, wf, c, coords, reps=5)
        except Exception as ex:
            print(f"variant {v} stamps: {type(ex).__name__}: {ex}")
            continue
        o = out.view(hf, wf, 352)
        print(f"variant {v} with stamps: {ms * 1e3:.1f} us; median cycles between consecutive stamps, per wave "
              "(start | A rows + centres | A fragments + boxes | cleared + primed | first fragments | ... phase boundary ... | tiles done | all waves done | sampled):")
        for wv in range(4):
            rows = o[wv::8, ::8, 324:348].contiguous().view(torch.int32).cpu().numpy().astype(np.int64).reshape(-1, 24)
            n = int(np.median((rows != 0).sum(axis=1)))
            rows = rows[(rows != 0).sum(axis=1) == n]
            d = np.diff(rows[:, :n], axis=1) & 0xffffffff
            print(f"  wave {wv} ({len(rows)} blocks, {n} stamps): " + ", ".join(str(int(np.median(d[:, k]))) for k in range(n - 1))
                  + f"; total {int(np.median((rows[:, n - 1] - rows[:, 0]) & 0xffffffff))}")


if __name__ == "__main__":
    main()
