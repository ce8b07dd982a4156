"""GPU parity of the tracker (flow + masking + Sobol subsampling + H fit + state machine) against
the CPU oracle tracker on a short synthetic sequence, driven through a reference-format config
file (pytracking/configs/WOFT.py) exactly as WOFT_demo.py does."""
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tracker_ref  # noqa: E402  (checker only)
from woft_amd import synth  # noqa: E402

ROOT = Path(__file__).resolve().parent.parent


def _corners_err(Ha, Hb, H, W):
    c = np.array([[W / 4, H / 4, 1], [3 * W / 4, H / 4, 1], [3 * W / 4, 3 * H / 4, 1], [W / 4, 3 * H / 4, 1.0]]).T
    pa, pb = np.linalg.inv(Ha) @ c, np.linalg.inv(Hb) @ c
    return np.abs(pa[:2] / pa[2] - pb[:2] / pb[2]).max()


@pytest.mark.parametrize("cfg,estimator", [("WOFT.py", "qr"), ("WOFT_IRLS.py", "irls_huber2")])
def test_tracker_matches_oracle(cfg, estimator):
    from pytracking.utils.config import load_config
    H, W, iters, nframes = 128, 160, 4, 4
    sd = synth.make_state_dict(seed=7)
    template = synth.make_template(H, W, seq_id=3)
    frames = [synth.make_frame(template, t) for t in range(1, nframes + 1)]
    mask = synth.make_init_mask(H, W)

    conf = load_config(ROOT / "pytracking" / "configs" / cfg)
    conf.flow_config.model = sd
    conf.flow_config.iters = iters
    tracker = conf.tracker_class(conf)
    tracker.init(template, mask)
    ref = tracker_ref.TrackerRef(sd, iters=iters, estimator=estimator)
    ref.init(template, mask)
    for t, f in enumerate(frames):
        Hg, mg = tracker.track(f)
        Hr, mr = ref.track(f)
        assert Hg.shape == (3, 3) and Hg.dtype == np.float64
        assert mg.lost == mr.lost and mg.N_lost == mr.N_lost and bool(mg.global_H_success) == bool(mr.global_H_success)
        err = _corners_err(Hg, Hr, H, W)
        assert err < 1.0, f"frame {t}: box corners differ by {err:.3f} px"
        assert np.allclose(mg.last_good_H2init, mr.last_good_H2init, atol=1e-2)


def test_tracker_lost_branch_and_interfaces():
    """Force the re-detection test to fail: the local (t-1 -> t) branch must run and agree with
    the oracle; also checks set_fast_meta replay and the single-component assertion."""
    from types import SimpleNamespace
    from pytracking.utils.config import load_config
    from woft_amd import presets
    H, W, iters = 128, 160, 4
    sd = synth.make_state_dict(seed=7)
    template = synth.make_template(H, W, seq_id=4)
    frames = [synth.make_frame(template, t) for t in (1, 2)]
    mask = synth.make_init_mask(H, W)
    conf = load_config(ROOT / "pytracking" / "configs" / "WOFT.py")
    conf.flow_config.model = sd
    conf.flow_config.iters = iters
    conf.redet_success_fn = presets.redetection_by_inliers(1e-6, 0.999)      # never satisfied
    tracker = conf.tracker_class(conf)
    tracker.init(template, mask)

    class Ref(tracker_ref.TrackerRef):
        pass
    ref = Ref(sd, iters=iters)
    ref.init(template, mask)
    import oracle.hfit_ref as hr
    orig = hr.redet_success
    hr.redet_success = lambda *a, **k: False
    try:
        for f in frames:
            Hg, mg = tracker.track(f)
            Hr, mr = ref.track(f)
            assert mg.lost and mr.lost and mg.N_lost == mr.N_lost
            assert _corners_err(Hg, Hr, H, W) < 1.0
            assert _corners_err(mg.H_local_cur2init, mr.H_local_cur2init, H, W) < 1.0
    finally:
        hr.redet_success = orig
    # fast-forward replay (TRK:49-76)
    meta = SimpleNamespace(estim_H_current2template=np.diag([1.0, 1.0, 1.0]) * 1.0)
    tracker.set_fast_meta(meta)
    Hf, mf = tracker.track(frames[0])
    assert Hf is meta.estim_H_current2template and mf is meta and not tracker.lost
    two = mask.copy()
    two[:8, :8] = 255
    with pytest.raises(AssertionError):
        tracker.init(template, two)


def test_tracker_downscaled_inputs():
    """downscale_inputs = 2 with RAFT replicate padding (configs/WOFT_downscale_2x.py, TRK:27-30,60-61,280-283)."""
    from pytracking.utils.config import load_config
    H, W, iters = 268, 332, 3                  # -> 134 x 166 after the resize: not a multiple of 8
    sd = synth.make_state_dict(seed=7)
    template = synth.make_template(H, W, seq_id=6)
    frames = [synth.make_frame(template, t) for t in (1, 2)]
    mask = synth.make_init_mask(H, W)
    conf = load_config(ROOT / "pytracking" / "configs" / "WOFT_downscale_2x.py")
    conf.flow_config.model = sd
    conf.flow_config.iters = iters
    tracker = conf.tracker_class(conf)
    tracker.init(template, mask)
    ref = tracker_ref.TrackerRef(sd, iters=iters, downscale=2, padding_mode="RAFT")
    ref.init(template, mask)
    d = np.abs(tracker.template_img.cpu().numpy().astype(np.int32) - ref.template_img.astype(np.int32))
    assert d.max() <= 1
    for f in frames:
        Hg, mg = tracker.track(f)
        Hr, mr = ref.track(f)
        assert mg.lost == mr.lost
        assert _corners_err(Hg, Hr, H, W) < 1.0
