#!/usr/bin/env python
"""Headless run of the reference's demo sequence of calls against the shim package.

The reference's only entry point, WOFT_demo.py (run: lines 36-83), needs an OpenCV GUI (ROI selection, imshow).  This
driver makes the same calls in the same order through the same import paths --

    load_config(--config) -> config.tracker_class(config) -> GeneralVideoCapture(video).read()
    -> init mask (rectangle, DEMO:86-96 rule: rows y..y+h, columns x..x+w inclusive)
    -> tracker.init(frame, mask) -> per frame: tracker.track(frame) [exception -> identity, DEMO:66-72]
    -> overlay: init mask warped by inv(H) (nearest) + blend_mask outline + draw_text (DEMO:99-111)

-- with the rectangle given on the command line instead of cv2.selectROI and the overlays written as PNG files
instead of shown.  `--weights synthetic[:seed]` substitutes the seeded synthetic checkpoint for the trained one that the
reference snapshot does not contain (.MISSING_LARGE_BLOBS).

    python tools/woft_demo_headless.py frames_dir --roi 40,32,80,64 --out overlays/ --weights synthetic
"""
import argparse
import logging
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

from pytracking.utils.config import load_config  # noqa: E402
from pytracking.utils import vis_utils as vu  # noqa: E402
from pytracking.utils import io as io_utils  # noqa: E402

logger = logging.getLogger(__name__)


def rect_mask(img, x, y, w, h):
    """The demo's mask from an (x, y, w, h) rectangle (DEMO:93-95)."""
    mask = np.zeros(img.shape[:2], dtype=np.uint8)
    mask[y:y + h + 1, x:x + w + 1] = 255
    return mask


def overlay(frame, init_mask, H_2init, label=None):
    """DEMO:99-111 with the mask warp on the device (cv2.warpPerspective INTER_NEAREST -> woft_warp_perspective_u8)."""
    import torch
    from woft_amd import ops
    m = torch.from_numpy(np.ascontiguousarray(init_mask)).cuda()
    cur = torch.empty_like(m)
    ops.warp_perspective_u8(m, np.linalg.inv(H_2init), cur, None, nearest=True)
    vis = vu.blend_mask(frame, cur.cpu().numpy(), color=(0, 255, 0), fill=False, contour_thickness=2)
    if label is not None:
        vis = vu.draw_text(vis, label, pos="tl", size=1, thickness=2)
    return vis


def run(video, config_path, roi, out_dir=None, weights=None, iters=None, max_frames=None, flow_overrides=None):
    """-> list of (H_2init 3x3 float64, meta | None) per tracked frame."""
    config = load_config(config_path)
    if weights and str(weights).startswith("synthetic"):
        from woft_amd import synth
        seed = int(str(weights).split(":")[1]) if ":" in str(weights) else 7
        config.flow_config.model = synth.make_state_dict(seed=seed)
    elif weights:
        config.flow_config.model = weights
    if iters:
        config.flow_config.iters = int(iters)
    for k, v in (flow_overrides or {}).items():
        setattr(config.flow_config, k, v)
    tracker = config.tracker_class(config)

    cap = io_utils.GeneralVideoCapture(video)
    success, frame = cap.read()
    if success is not True:
        raise SystemExit(f"Reading frame from {video} failed.")
    init_mask = rect_mask(frame, *roi)
    tracker.init(frame, init_mask)

    if out_dir is not None:
        out_dir = Path(out_dir)
        out_dir.mkdir(parents=True, exist_ok=True)
    results = []
    while max_frames is None or len(results) < max_frames:
        ret, frame = cap.read()
        if frame is None:
            break
        last_H = np.eye(3)
        meta = None
        try:
            H_2init, meta = tracker.track(frame)
            last_H = H_2init.copy()
        except Exception:
            logger.exception("Tracker exception")
            H_2init = last_H.copy()
        results.append((H_2init, meta))
        if out_dir is not None:
            from PIL import Image
            vis = overlay(frame.copy(), init_mask.copy(), H_2init.copy(), label=f"#{len(results)}")
            Image.fromarray(np.ascontiguousarray(vis[:, :, ::-1])).save(out_dir / f"{len(results):05d}.png")
    cap.release()
    return results


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("video", type=Path, help="directory of extracted frames (or a video file when OpenCV is installed)")
    ap.add_argument("--config", type=Path, default=ROOT / "pytracking" / "configs" / "WOFT.py")
    ap.add_argument("--roi", required=True, help="x,y,w,h of the target in the first frame (replaces cv2.selectROI)")
    ap.add_argument("--out", type=Path, default=None, help="directory for the overlay PNGs")
    ap.add_argument("--weights", default=None, help="checkpoint path, or 'synthetic[:seed]'")
    ap.add_argument("--iters", type=int, default=None)
    ap.add_argument("--max-frames", type=int, default=None)
    ap.add_argument("-v", "--verbose", action="store_true")
    a = ap.parse_args()
    logging.basicConfig(level=logging.DEBUG if a.verbose else logging.INFO,
                        format="[%(asctime)s] %(levelname)s:%(name)s:%(message)s")
    roi = tuple(int(v) for v in a.roi.split(","))
    res = run(a.video, a.config, roi, a.out, a.weights, a.iters, a.max_frames)
    for i, (H, meta) in enumerate(res, 1):
        lost = getattr(meta, "lost", None)
        print(f"frame {i}: lost={lost} H_2init={np.array2string(np.asarray(H), precision=4, suppress_small=True).replace(chr(10), ' ')}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
