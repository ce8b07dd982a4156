"""The demo call sequence (WOFT_demo.py:36-111) driven end to end, headless, against the shim package:
tools/woft_demo_headless.py on a directory of synthetic frames -- config loading through the reference's import paths,
frame source, rectangle mask, init, track loop, mask overlay."""
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))

from woft_amd import synth  # noqa: E402


def test_headless_demo_run(tmp_path):
    from PIL import Image
    import woft_demo_headless as demo
    from pytracking.utils.config import load_config
    H, W, n = 128, 160, 4
    template = synth.make_template(H, W, seq_id=3)
    frames = [template] + [synth.make_frame(template, t) for t in range(1, n + 1)]
    src = tmp_path / "frames"
    src.mkdir()
    for i, f in enumerate(frames):
        Image.fromarray(np.ascontiguousarray(f[:, :, ::-1])).save(src / f"{i:04d}.png")      # lossless, RGB on disk
    roi = (W // 4, H // 4, W // 2 - 1, H // 2 - 1)               # -> rows H/4 .. 3H/4-1, the SURVEY 8d rectangle
    cfg = ROOT / "pytracking" / "configs" / "WOFT.py"
    res = demo.run(src, cfg, roi, out_dir=tmp_path / "vis", weights="synthetic:7", iters=3)
    assert len(res) == n and all(m is not None for _, m in res)
    # the same sequence through the tracker directly
    conf = load_config(cfg)
    conf.flow_config.model = synth.make_state_dict(seed=7)
    conf.flow_config.iters = 3
    trk = conf.tracker_class(conf)
    mask = synth.make_init_mask(H, W)
    assert np.array_equal(demo.rect_mask(template, *roi), mask)
    trk.init(template, mask)
    for (Hd, md), f in zip(res, frames[1:]):
        Ht, mt = trk.track(f)
        assert np.array_equal(Hd, Ht) and md.lost == mt.lost
    outs = sorted((tmp_path / "vis").glob("*.png"))
    assert [o.name for o in outs] == [f"{i:05d}.png" for i in range(1, n + 1)]
    vis = np.asarray(Image.open(outs[0]))[:, :, ::-1]
    green = (vis == np.array([0, 255, 0])).all(-1)
    assert 100 < int(green.sum()) < H * W // 4                   # an outline, not a filled region
    ys, xs = np.nonzero(green)
    assert abs(ys.mean() - H / 2) < 24 and abs(xs.mean() - W / 2) < 24
    # a tracker exception inside the loop is survived with the identity, as DEMO:66-72
    class Boom(type(trk)):
        def track(self, *a, **k):
            raise RuntimeError("boom")
    conf2 = load_config(cfg)
    conf2.flow_config.model = synth.make_state_dict(seed=7)
    conf2.flow_config.iters = 2
    conf2.tracker_class = Boom
    import unittest.mock as um
    with um.patch.object(demo, "load_config", lambda p: conf2):
        res2 = demo.run(src, cfg, roi, max_frames=2)
    assert len(res2) == 2 and all(np.array_equal(Hm, np.eye(3)) and m is None for Hm, m in res2)
