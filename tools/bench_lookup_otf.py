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
    lp.ablate = int(os.environ.get("OTF_ABL", "0"))     # developer instances (csrc/lookup_otf.hip: 1, 2, 3, 8, 15, 16 = phase stamps, 64 = stamps in one chunk)
    ms = bench(lambda: ops.run_lookup_otf(lp), reps=20)
    print(f"volume-free lookup {hf}x{wf}, flow {flow}, ablate {lp.ablate}: {ms * 1e3:8.1f} us")
    import numpy as np
    if lp.ablate in (16, 64):
        torch.cuda.synchronize()
        rows = out.view(hf, wf, 352)[::8, ::8, 324:348].contiguous().view(torch.int32).cpu().numpy().astype(np.int64).reshape(-1, 24)
    if lp.ablate == 16:     # per-workgroup timeline (thread 0): start | A fragments + all-level set-up | per level: windows cleared, chunks, next level primed + samples written
        n = int(np.median((rows != 0).sum(axis=1)))
        rows = rows[(rows != 0).sum(axis=1) == n]
        d = np.diff(rows[:, :n], axis=1) & 0xffffffff
        print(f"  start -> set-up barrier (centres, origins / boxes of all levels, A fragments requested) {int(np.median(d[:, 0]))}")
        for l in range((n - 2) // 3):
            print(f"  level {l}: {'box + stream primed + ' if l == 0 else ''}windows cleared {int(np.median(d[:, 1 + 3 * l]))}, chunks {int(np.median(d[:, 2 + 3 * l]))}, "
                  f"next level primed + samples written {int(np.median(d[:, 3 + 3 * l]))}")
        print(f"  workgroup total (median cycles): {int(np.median((rows[:, n - 1] - rows[:, 0]) & 0xffffffff))}")
    if lp.ablate == 64:     # one chunk (level 0, second chunk), per two-K-step group: [top, stream landed, barrier, first fragments, reads + MFMAs (+ DMA issue) done] ... drop
        rows = rows[rows[:, 20] != 0]
        d = np.diff(rows[:, :21], axis=1) & 0xffffffff
        names = ["wait stream", "barrier", "to first fragments", "reads + MFMAs + DMA issue", "to next group / drop"]
        for g in range(4):
            print(f"  group {g}: " + ", ".join(f"{n} {int(np.median(d[:, 5 * g + k]))}" for k, n in enumerate(names)))
        print(f"  chunk total (median cycles): {int(np.median((rows[:, 20] - rows[:, 0]) & 0xffffffff))}  ({len(rows)} workgroups)")


if __name__ == "__main__":
    main()
