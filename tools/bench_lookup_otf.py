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
    lp.ablate = int(os.environ.get("OTF_ABL", "0")) | (int(os.environ.get("OTF_VARIANT", "0")) << 8)     # developer ablation bits / kernel variant (include/woft_hip.h)
    ms = bench(lambda: ops.run_lookup_otf(lp), reps=20)
    print(f"volume-free lookup {hf}x{wf}, flow {flow}, ablate {lp.ablate}: {ms * 1e3:8.1f} us")
    if (lp.ablate & 16) and not (lp.ablate >> 8):      # per-workgroup timeline: [start | per level: coords written, synced, stream primed, chunks done, samples written]
        import numpy as np
        torch.cuda.synchronize()
        rows = out.view(hf, wf, 352)[::8, ::8, 324:348].contiguous().view(torch.int32).cpu().numpy().astype(np.int64).reshape(-1, 24)
        d = np.diff(rows[:, :21], axis=1) & 0xffffffff
        names = ["coords", "sync", "bbox+zero+prime", "chunks", "samples"]
        print("  A fragments + setup -> first level: included in level 0 'coords'")
        for l in range(4):
            print(f"  level {l}: " + ", ".join(f"{n} {int(np.median(d[:, 5 * l + k]))}" for k, n in enumerate(names)))
        print(f"  workgroup total (median cycles): {int(np.median((rows[:, 20] - rows[:, 0]) & 0xffffffff))}")
    if (lp.ablate & 32) and not (lp.ablate >> 8):      # one chunk (level 0, second chunk), per K-step group: [start, stream landed, barrier, issued, MFMAs done] ... drop
        import numpy as np
        torch.cuda.synchronize()
        rows = out.view(hf, wf, 352)[::8, ::8, 324:348].contiguous().view(torch.int32).cpu().numpy().astype(np.int64).reshape(-1, 24)
        rows = rows[rows[:, 20] != 0]
        d = np.diff(rows[:, :22], axis=1) & 0xffffffff
        names = ["wait stream", "barrier", "issue", "reads+MFMAs", "to next group"]
        for g in range(4):
            print(f"  group {g}: " + ", ".join(f"{n} {int(np.median(d[:, 5 * g + k]))}" for k, n in enumerate(names) if 5 * g + k < d.shape[1]))
        print(f"  chunk total (median cycles): {int(np.median((rows[:, 20] - rows[:, 0]) & 0xffffffff))}  ({len(rows)} workgroups)")


if __name__ == "__main__":
    main()
