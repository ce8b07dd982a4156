"""Host-side (Python) time of a tracked frame: cProfile over N track() calls at 1080p -- what stands between the
per-frame device->host read and the next frame's first launches.  python tools/host_profile.py [frames]"""
import cProfile
import pstats
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from woft_amd import synth
from pytracking.utils.config import load_config

ROOT = Path(__file__).resolve().parent.parent


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    H, W = 1080, 1920
    conf = load_config(ROOT / "pytracking" / "configs" / "WOFT.py")
    conf.flow_config.model = synth.make_state_dict(seed=7)
    conf.flow_config.iters = 12
    conf.flow_config.precision = "bf16x3"
    template = synth.make_template(H, W, seq_id=0)
    mask = synth.make_init_mask(H, W)
    trk = conf.tracker_class(conf)
    trk.init(template, mask)
    frames = [torch.from_numpy(synth.make_frame(template, t)).cuda() for t in range(1, 9)]
    for f in frames[:3]:
        trk.track(f)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pr = cProfile.Profile()
    pr.enable()
    for i in range(n):
        trk.track(frames[i % len(frames)])
    pr.disable()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{n} frames, {1e3 * dt / n:.2f} ms per frame (under cProfile)")
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(28)

    # GPU idle time between frames: event before a frame's first enqueue (the frame copy) and after its last kernel
    # (the inlier test, just before the device->host read)
    from woft_amd import ops, tracker as trk_mod
    firsts, lasts = [], []
    orig_dev, orig_inl = trk_mod._device_u8, ops.inlier_frac

    def dev_u8(img, copy=False):
        if copy:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            firsts.append(e)
        return orig_dev(img, copy=copy)

    def inl(*a, **k):
        r = orig_inl(*a, **k)
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        lasts.append(e)
        return r
    trk_mod._device_u8, ops.inlier_frac = dev_u8, inl
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        trk.track(frames[i % len(frames)])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    trk_mod._device_u8, ops.inlier_frac = orig_dev, orig_inl
    busy = [a.elapsed_time(b) for a, b in zip(firsts, lasts)]
    idle = [lasts[i].elapsed_time(firsts[i + 1]) for i in range(len(firsts) - 1)]
    print(f"wall {1e3 * dt / n:.3f} ms per frame; first enqueue -> last kernel done {np.median(busy):.3f} ms; "
          f"last kernel done -> next frame's first enqueue {np.median(idle) * 1e3:.0f} us (median), {np.mean(idle) * 1e3:.0f} us (mean)")


if __name__ == "__main__":
    main()
