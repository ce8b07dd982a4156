"""CPU tests of the demo-side shim modules WOFT_demo.py imports (DEMO:9-11): frame source ordering
(utils/io.py:145-169) and the overlay helpers (utils/vis_utils.py:593-621, 316-369), plus the rectangle-mask rule of
DEMO:86-96 used by the headless driver."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))


def _write(path, value, size=(12, 16)):
    from PIL import Image
    img = np.full(size + (3,), value, np.uint8)
    img[0, 0] = (255, 0, 0)                                   # RGB red -> BGR (0, 0, 255)
    Image.fromarray(img).save(path)


def test_general_video_capture_reads_sorted_images_as_bgr(tmp_path):
    from pytracking.utils.io import GeneralVideoCapture
    for name, v in (("b_002.png", 20), ("a_010.png", 10), ("b_001.jpg", 15), ("notes.txt", 0), ("c.PNG", 30)):
        if name.endswith(".txt"):
            (tmp_path / name).write_text("x")
        else:
            _write(tmp_path / name, v)
    cap = GeneralVideoCapture(tmp_path)
    seen = []
    while True:
        ok, img = cap.read()
        if not ok:
            assert img is None
            break
        assert img.dtype == np.uint8 and img.shape == (12, 16, 3)
        seen.append((cap.frame_src, int(img[5, 5, 1])))
    assert [s for s, _ in seen] == ["a_010.png", "b_001.jpg", "b_002.png", "c.PNG"]      # sorted file names, images only
    assert [v for _, v in seen if v in (10, 20, 30)] == [10, 20, 30]
    cap = GeneralVideoCapture(tmp_path)
    _, img = cap.read()
    assert tuple(img[0, 0]) == (0, 0, 255)                    # BGR channel order, as cv2.imread
    assert cap.release() is None
    rev = GeneralVideoCapture(tmp_path, reverse=True)
    assert rev.read()[0] and rev.frame_src == "c.PNG"


def test_blend_mask_and_draw_text():
    from pytracking.utils.vis_utils import blend_mask, draw_text
    img = np.full((40, 50, 3), 100, np.uint8)
    mask = np.zeros((40, 50), np.uint8)
    mask[10:30, 15:40] = 255
    out = blend_mask(img, mask, color=(0, 255, 0), fill=False, contour_thickness=2)
    assert out is not img and (img == 100).all()              # input untouched
    green = (out == np.array([0, 255, 0])).all(-1)
    assert green[10, 20] and green[29, 20] and green[20, 15] and green[20, 39]      # outline on the border
    assert not green[20, 27] and not green[2, 2]              # interior and background untouched
    assert (out[20, 27] == 100).all()
    filled = blend_mask(img, mask, color=(0, 255, 0), alpha=0.5, fill=True)
    assert tuple(filled[20, 27]) == (50, 177, 50) and (filled[2, 2] == 100).all()
    t = draw_text(img, "seq #3", pos="tl", size=1, thickness=2)
    assert t.shape == img.shape and t.dtype == img.dtype


def test_rect_mask_rule():
    import woft_demo_headless as demo
    img = np.zeros((48, 64, 3), np.uint8)
    m = demo.rect_mask(img, 10, 8, 20, 12)
    assert m.dtype == np.uint8 and m.shape == (48, 64)
    ys, xs = np.nonzero(m)
    assert (ys.min(), ys.max(), xs.min(), xs.max()) == (8, 20, 10, 30) and set(np.unique(m)) == {0, 255}
