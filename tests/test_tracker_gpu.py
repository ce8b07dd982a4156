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


@pytest.mark.parametrize("cfg,estimator,precision", [("WOFT.py", "qr", None), ("WOFT_IRLS.py", "irls_huber2", None),
                                                     ("WOFT.py", "qr", "fp32"), ("WOFT.py", "qr", "f16mx8")])
def test_tracker_matches_oracle(cfg, estimator, precision):
    """precision None: what the SHIPPED flow config selects (bf16x3: split-bf16 MFMA emulating fp32 -- the path a drop-in
    user and the bench run); "fp32": exact fp32 MFMA products.  Volume-free correlation lookup in both."""
    from pytracking.utils.config import load_config
    H, W, iters, nframes = 128, 160, 4, 4
    sd = synth.make_state_dict(seed=7)
    template = synth.make_template(H, W, seq_id=3)
    frames = [synth.make_frame(template, t) for t in range(1, nframes + 1)]
    mask = synth.make_init_mask(H, W)

    conf = load_config(ROOT / "pytracking" / "configs" / cfg)
    conf.flow_config.model = sd
    conf.flow_config.iters = iters
    if precision:
        conf.flow_config.precision = precision
    tracker = conf.tracker_class(conf)
    assert tracker.flower.engine.corr == "otf"         # (every precision, exact fp32 included)
    assert tracker.flower.precision == (precision or "bf16x3")
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
    tracker.set_fast_meta(meta)                      # reference-style cancellation: `tracker.fast_forward = False`
    assert tracker.fast_forward
    tracker.fast_forward = False
    assert not tracker.fast_forward and tracker.track(frames[0])[1] is not meta
    with pytest.raises(ValueError):
        tracker.fast_forward = True
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


def test_fused_path_equals_generic_path(monkeypatch):
    """The device-side selection / fit / inlier kernels (one host read per flow) must give exactly the
    result of calling the config's callables as the reference tracker does."""
    from pytracking.utils.config import load_config
    from woft_amd import presets
    H, W, iters, nframes = 136, 200, 3, 4
    sd = synth.make_state_dict(seed=7)
    template = synth.make_template(H, W, seq_id=8)
    frames = [synth.make_frame(template, t) for t in range(1, nframes + 1)]
    mask = synth.make_init_mask(H, W)

    def run(fused, always_lost=False, cfg="WOFT.py"):
        monkeypatch.setenv("WOFT_FUSED", "1" if fused else "0")
        conf = load_config(ROOT / "pytracking" / "configs" / cfg)
        conf.flow_config.model = sd
        conf.flow_config.iters = iters
        if always_lost:
            conf.redet_success_fn = presets.redetection_by_inliers(1e-6, 0.999)
        trk = conf.tracker_class(conf)
        assert (trk._fused is not None) == fused
        trk.init(template, mask)
        return [trk.track(f) for f in frames]

    for kw in (dict(), dict(always_lost=True), dict(cfg="WOFT_IRLS.py")):
        a, b = run(True, **kw), run(False, **kw)
        for (Ha, ma), (Hb, mb) in zip(a, b):
            assert ma.lost == mb.lost and ma.N_lost == mb.N_lost and bool(ma.global_H_success) == bool(mb.global_H_success)
            assert np.array_equal(Ha, Hb), np.abs(Ha - Hb).max()
            assert np.array_equal(ma.H_global_cur2init, mb.H_global_cur2init)


def test_tc_select_kernel_semantics(golden_dir):
    """Sobol ranks / order / duplicates of the device-side selection against the reference's subsampler output."""
    from woft_amd import ops, presets
    g = np.load(golden_dir / "sobol.npz")
    for (h, w) in ((20, 30), (25, 24), (50, 40), (720, 720)):
        n = h * w
        rs = np.random.RandomState(n)
        tmask = (rs.uniform(size=(h, w)) < 0.97).astype(np.uint8) * 255
        dst = torch.from_numpy(np.stack([rs.uniform(-3, w + 3, n), rs.uniform(-3, h + 3, n)]).astype(np.float32)).cuda()
        wts = torch.from_numpy(rs.uniform(size=n).astype(np.float32)).cuda()
        pw = (rs.uniform(size=(h, w)) < 0.9).astype(np.uint8)
        u = torch.from_numpy(presets.sobol_points(500).astype(np.float32)).cuda()
        pa, pb, wo = torch.zeros(1024, 2, device="cuda"), torch.zeros(1024, 2, device="cuda"), torch.zeros(1024, device="cuda")
        cnt = torch.zeros(2, dtype=torch.int32, device="cuda")
        ops.tc_select(dst, wts, torch.from_numpy(tmask).cuda(), torch.from_numpy(pw).cuda(), h, w, 1, u,
                      ops.tc_select_ws(n), pa, pb, wo, cnt)
        torch.cuda.synchronize()
        d = dst.cpu().numpy()
        idx = np.arange(n)
        keep = (tmask.reshape(-1) > 0) & ~((d[0] < 0) | (d[1] < 0) | (np.rint(d[0]) >= w) | (np.rint(d[1]) >= h))
        ri = np.clip(np.rint(d[1]).astype(np.int64), 0, h - 1) * w + np.clip(np.rint(d[0]).astype(np.int64), 0, w - 1)
        keep &= pw.reshape(-1)[ri] > 0
        kept = idx[keep]
        N = len(kept)
        sel = np.arange(N) if N <= 500 else np.unique(np.round(N * presets.sobol_points(500)).astype(np.int32))
        if N == 518400:
            assert np.array_equal(sel, g["n518400"])
        chosen = kept[sel]
        m, nk = int(cnt[0]), int(cnt[1])
        assert nk == N and m == len(chosen)
        assert np.array_equal(pb[:m].cpu().numpy(), np.stack([chosen % w, chosen // w], 1).astype(np.float32))
        assert np.array_equal(pa[:m].cpu().numpy(), d[:, chosen].T)
        assert np.array_equal(wo[:m].cpu().numpy(), wts.cpu().numpy()[chosen])


def test_weight_head_on_mask_region_only():
    """The tracker reads flow weights only inside its template mask (TRK:287-312): evaluating the weight head on
    those 1/8-res pixels (+ the upsampling support) instead of everywhere gives the same weights there, bit for bit,
    and therefore identical homographies -- on both tracker paths and for a mask touching the border."""
    from pytracking.utils.config import load_config
    H, W, iters = 136, 200, 3
    sd = synth.make_state_dict(seed=5)
    template = synth.make_template(H, W, seq_id=7)
    frames = [synth.make_frame(template, t) for t in (1, 2, 3)]
    mask = np.zeros((H, W), np.uint8)
    mask[0:70, 30:120] = 255
    outs = {}
    for full in (True, False):
        conf = load_config(ROOT / "pytracking" / "configs" / "WOFT.py")
        conf.flow_config.model = sd
        conf.flow_config.iters = iters
        conf.flow_config.precision = "bf16x3"
        conf.flow_config.padding_mode = "RAFT"
        conf.mask_weight_head = not full
        trk = conf.tracker_class(conf)
        trk.init(template, mask)
        outs[full] = [trk.track(f)[0] for f in frames]
        plan = trk.flower.engine.plan(*[(d + 7) // 8 * 8 for d in (H, W)])
        assert (plan.wh_region is None) == full
        assert trk._mask_weight_head() == (not full)
        if not full:
            n_sel = int(plan.wh_region[0].numel())
            assert 0 < n_sel < plan.P // 2
            # the region belongs to the TRACKER's calls (weight_region=True); anybody else's compute_flow gets the
            # reference's full weight map -- after init(), too
            _, _, w_reg = trk.flower.compute_flow(template, frames[0], mode="TC", do_sigmoid=True, numpy_out=True,
                                                  weight_region=True)
            _, _, w_full = trk.flower.compute_flow(template, frames[0], mode="TC", do_sigmoid=True, numpy_out=True)
            sel = (mask > 0).reshape(-1)
            assert np.array_equal(w_reg[0, sel], w_full[0, sel])
            assert not np.array_equal(w_reg[0, ~sel], w_full[0, ~sel])       # (outside: 0.5 = sigmoid(0) vs real weights)
            assert float(np.abs(w_full[0, ~sel] - 0.5).max()) > 1e-3
            trk.flower.pin_weight_region(None)
            _, _, w_none = trk.flower.compute_flow(template, frames[0], mode="TC", do_sigmoid=True, numpy_out=True,
                                                   weight_region=True)
            assert np.array_equal(w_none, w_full)
    for a, b in zip(outs[True], outs[False]):
        assert np.array_equal(a, b)
    # the key's default: on (the tracker never reads the other weights); off when a post-hoc filter of the weight map is set
    conf = load_config(ROOT / "pytracking" / "configs" / "WOFT.py")
    conf.flow_config.model, conf.flow_config.iters = sd, 1
    trk = conf.tracker_class(conf)
    assert trk._mask_weight_head()
    conf.post_hoc_weights_postprocessing_fn = lambda w: w
    assert not conf.tracker_class(conf)._mask_weight_head()


@pytest.mark.parametrize("cfg", ["WOFT.py", "WOFT_IRLS.py"])
def test_weight_head_on_the_drawn_correspondences_only(cfg):
    """With a subsampler in front of the fit only the weights of the drawn correspondences are read, and the draw is decided
    by the flow alone (masks, bounds, Sobol points): the tracker selects first and has the weight head evaluated on the
    windows under the drawn pixels' upsampling support (woft_wh_needed + negative window-list entries).  Same weights at the
    drawn pixels, bit for bit; identical homographies, lost flags included (a forced-lost frame runs the local stage, whose
    flow is not from the pinned template and keeps the full head)."""
    from pytracking.utils.config import load_config
    H, W, iters = 544, 960, 3        # (large enough that the 500 drawn pixels leave windows of the region unneeded)
    sd = synth.make_state_dict(seed=5)
    template = synth.make_template(H, W, seq_id=7)
    frames = [synth.make_frame(template, t) for t in (1, 2, 3, 4)]
    mask = np.zeros((H, W), np.uint8)
    mask[0:300, 100:700] = 255
    outs, sel = {}, {}
    for sparse in (False, True):
        conf = load_config(ROOT / "pytracking" / "configs" / cfg)
        conf.flow_config.model = sd
        conf.flow_config.iters = iters
        conf.flow_config.precision = "bf16x3"
        conf.flow_config.padding_mode = "RAFT"
        conf.sparse_weight_head = sparse
        trk = conf.tracker_class(conf)
        trk.init(template, mask)
        assert trk._sparse_weights == sparse and trk._fused is not None
        trk.flower.defer_min_ratio = 0        # (this region, ~3 000 windows, is below the provider's pay-off rule: force it)
        res = []
        for t, f in enumerate(frames):
            Hm, meta = trk.track(f)
            res.append((Hm, meta.lost, meta.N_lost))
            assert trk.flower.weights_deferred == sparse          # (the last flow of the frame was the template's)
            if t == 1:
                b = trk._fb
                n = int(b["res"].view(torch.int32)[12])
                sel[sparse] = (b["pa"][:n].clone(), b["pb"][:n].clone(), b["w"][:n].clone())
        outs[sparse] = res
        if sparse:
            plan = trk.flower.engine.plan(*[(d + 7) // 8 * 8 for d in (H, W)])
            dyn, _, _, n_needed = next(iter(plan._wh_dyn.values()))
            n_region = int(plan.wh_region[0].numel())
            assert 0 < int(n_needed) < n_region and int((dyn >= 0).sum()) == int(n_needed)
            print(f"{cfg}: {int(n_needed)} of {n_region} windows of the mask region evaluated")
    for k in range(3):
        assert torch.equal(sel[False][k], sel[True][k])           # same draw, same targets, same weights
    for a, b in zip(outs[False], outs[True]):
        assert np.array_equal(a[0], b[0]) and a[1:] == b[1:]
    # off when there is nothing to draw from (no subsampler), and by the key / the environment
    conf = load_config(ROOT / "pytracking" / "configs" / cfg)
    conf.flow_config.model, conf.flow_config.iters = sd, 1
    conf.subsampler_fn = None
    trk = conf.tracker_class(conf)
    trk.init(template, mask)
    assert not trk._sparse_weights


def test_lost_frames_keep_the_template_resident_and_defer_the_local_weights():
    """The lost branch (TRK:167-207) runs the frame t-1 -> t flow in a SECOND buffer set, so the template's feature /
    context / gate-bias tensors stay resident (the reference alternates template and frame t-1 as sources, TRK:101,181) and
    the frame after a lost one costs what any frame costs; its weight head, like the global stage's, is evaluated only under
    the correspondences the fit draws (which start inside the carried mask, TRK:314-327; the draw does not depend on the
    weights).  Identical homographies with and without the deferral, frame for frame; the template is encoded once."""
    from pytracking.utils.config import load_config
    H, W, iters = 544, 960, 3
    sd = synth.make_state_dict(seed=5)
    template = synth.make_template(H, W, seq_id=3)
    frames = [synth.make_frame(template, t) for t in (1, 2, 3, 4, 5)]
    mask = np.zeros((H, W), np.uint8)
    mask[100:400, 200:800] = 255
    lost_at = {1, 2, 4}
    outs = {}
    for sparse in (False, True):
        conf = load_config(ROOT / "pytracking" / "configs" / "WOFT.py")
        conf.flow_config.model, conf.flow_config.iters, conf.flow_config.precision = sd, iters, "bf16x3"
        conf.sparse_weight_head = sparse
        trk = conf.tracker_class(conf)
        trk.init(template, mask)
        trk.flower.defer_min_ratio = 0
        inner, k = trk._global_stage, {"i": -1}

        def overruled(frame, prewarp_H, inner=inner, k=k):
            fit = inner(frame, prewarp_H)
            k["i"] += 1
            if k["i"] in lost_at:
                fit.success = False
            return fit
        trk._global_stage = overruled
        plan0 = trk.flower.engine.plan(H, W)
        encodes = {"n": 0}
        enc0 = plan0.encode_source

        def counted(enc0=enc0, encodes=encodes, **kw):
            encodes["n"] += 1
            return enc0(**kw)
        plan0.encode_source = counted
        res, deferred_local = [], []
        for t, f in enumerate(frames):
            Hm, meta = trk.track(f)
            res.append((Hm, meta.lost, meta.N_lost, getattr(meta, "H_local_cur2init", None)))
            if meta.lost:
                deferred_local.append(trk.flower.weights_deferred)       # (the frame's last flow was frame t-1 -> t)
        assert [r[1] for r in res] == [t in lost_at for t in range(5)]
        assert encodes["n"] == 1 and plan0.source_tag is trk.flower        # template encoded once, never evicted
        plan1 = trk.flower.engine.plan(H, W, 1)
        assert plan1 is not plan0 and plan1.source_tag is None
        assert deferred_local == [sparse] * 3
        outs[sparse] = res
    for a, b in zip(outs[False], outs[True]):
        assert np.array_equal(a[0], b[0]) and a[1:3] == b[1:3]
        assert (a[3] is None) == (b[3] is None) and (a[3] is None or np.array_equal(a[3], b[3]))


@pytest.mark.parametrize("precision,host_frames", [("bf16x3", False), ("fp32", False), ("bf16x3", True)])
def test_consecutive_lost_frames_reuse_the_previous_target_features(precision, host_frames, monkeypatch):
    """Frames lost in a row (TRK:167-207): frame t was the TARGET of the t-1 -> t flow and is the SOURCE of the t -> t+1 flow, so
    the provider takes its feature map from the previous call's target pyramid instead of a second fnet pass over the same image
    (round 6; `compute_flow(..., src_is_previous_dst=True)`, engine.encode_source(reuse_target=True)).  Same launch program on the
    same pixels: homographies bit-identical with the reuse switched off (WOFT_LOCAL_REUSE=0), frame for frame; reused exactly on
    the second and later frames of a lost run, also for host (numpy) frames, whose device copies alternate between two buffers."""
    from pytracking.utils.config import load_config
    H, W, iters = 256, 320, 3
    sd = synth.make_state_dict(seed=5)
    template = synth.make_template(H, W, seq_id=4)
    frames = [synth.make_frame(template, t) for t in range(1, 8)]
    if not host_frames:
        frames = [torch.from_numpy(f).cuda() for f in frames]
    mask = synth.make_init_mask(H, W)
    lost_at = {1, 2, 3, 5}                     # runs: frames 1-3 (reuse at 2 and 3), frame 5 alone (no reuse: frame 4 was not a local target)
    outs = {}
    for reuse in ("1", "0"):
        monkeypatch.setenv("WOFT_LOCAL_REUSE", reuse)
        conf = load_config(ROOT / "pytracking" / "configs" / "WOFT.py")
        conf.flow_config.model, conf.flow_config.iters, conf.flow_config.precision = sd, iters, precision
        trk = conf.tracker_class(conf)
        trk.init(template, mask)
        inner, k = trk._global_stage, {"i": -1}

        def overruled(frame, prewarp_H, inner=inner, k=k):
            fit = inner(frame, prewarp_H)
            k["i"] += 1
            if k["i"] in lost_at:
                fit.success = False
            return fit
        trk._global_stage = overruled
        res, reused = [], []
        for t, f in enumerate(frames):
            trk.flower.source_features_reused = False
            Hm, meta = trk.track(f)
            res.append((Hm, meta.lost, getattr(meta, "H_local_cur2init", None)))
            reused.append(bool(meta.lost and trk.flower.source_features_reused))
        assert [r[1] for r in res] == [t in lost_at for t in range(len(frames))]
        assert reused == [reuse == "1" and t in (2, 3) for t in range(len(frames))], reused
        outs[reuse] = res
    for a, b in zip(outs["1"], outs["0"]):
        assert np.array_equal(a[0], b[0]) and a[1] == b[1]
        assert (a[2] is None) == (b[2] is None) and (a[2] is None or np.array_equal(a[2], b[2]))


@pytest.mark.parametrize("name,cfg", [("woft", "WOFT.py"), ("lost", "WOFT.py"), ("irls", "WOFT_IRLS.py"),
                                      ("woft", "inline_wlsq.py"), ("lost", "inline_wlsq.py"), ("irls", "inline_irls.py")])
@pytest.mark.parametrize("backend", ["device", "callables"])
def test_tracker_vs_reference_tracker_runs(golden_dir, monkeypatch, name, cfg, backend):
    """SURVEY 8c fixture (7): the HIP tracker against runs of the REFERENCE's own YAOFTrackerSingleControl
    (oracle/gen_golden.py: gen_tracker -- reference configs WOFT.py / ablation_08.py, functional cv2 stub): per frame
    the homography (box corners < 1 px), lost / N_lost / global_H_success and the local-branch result, incl. the frames
    whose re-detection test was made to fail (lost -> local flow -> recovery).  Both solver back ends; the shipped presets
    configs and configs in the reference's inline form (tests/configs/inline_*.py: recognised by woft_amd.probe)."""
    from pytracking.utils.config import load_config
    from woft_amd import presets
    g = np.load(golden_dir / "tracker_ref_runs.npz")
    monkeypatch.setenv("WOFT_FUSED", "1" if backend == "device" else "0")
    sd = synth.make_state_dict(seed=int(g["seed"]))
    inline = cfg.startswith("inline")
    conf = load_config(ROOT / ("tests/configs" if inline else "pytracking/configs") / cfg)
    conf.flow_config.model = sd
    conf.flow_config.iters = int(g["iters"])
    tracker = conf.tracker_class(conf)
    assert (tracker._fused is not None) == (backend == "device")
    if backend == "device":
        assert ("probed" in tracker.solver_decision) == inline
        assert (tracker._fused["reweight"], tracker._fused["huber_k"], tracker._fused["n_irls"]) == ((2, 2.0, 5) if name == "irls" else (0, 0.0, 0))
    mask = g[f"{name}_mask"]
    H, W = mask.shape
    tracker.init(g[f"{name}_template"], mask)
    fail = set(int(i) for i in g[f"{name}_force_fail"])
    normal, never = conf.redet_success_fn, presets.redetection_by_inliers(1e-6, 0.999)
    for i, f in enumerate(g[f"{name}_frames"]):
        tracker.C.redet_success_fn = never if i in fail else normal           # config-level hook, as in the golden run
        tracker._fused = tracker._fused_specs()
        Hg, mg = tracker.track(f)
        lost, n_lost, ok, has_local = g[f"{name}_meta"][i]
        assert (mg.lost, mg.N_lost, bool(mg.global_H_success)) == (bool(lost), int(n_lost), bool(ok)), (name, i)
        assert _corners_err(Hg, g[f"{name}_H"][i], H, W) < 1.0, (name, i)
        assert _corners_err(mg.H_global_cur2init, g[f"{name}_Hglobal_{i}"], H, W) < 1.0
        assert np.allclose(mg.last_good_H2init, g[f"{name}_lastgood_{i}"], atol=1e-2)
        assert hasattr(mg, "H_local_cur2init") == bool(has_local)
        if has_local:
            assert _corners_err(mg.H_local_cur2init, g[f"{name}_Hlocal_{i}"], H, W) < 1.0


@pytest.mark.parametrize("backend", ["device", "callables"])
def test_tracker_crop_padding_mode(monkeypatch, backend):
    """padding_mode 'crop' (optical_flow/raft.py:235-247) on frames that are not multiples of 8: the flow grid is
    smaller than the frame, the masks are not -- both back ends must index them as the reference does (the oracle
    tracker does it with the reference's own indexing expressions)."""
    from pytracking.utils.config import load_config
    H, W, iters = 133, 171, 3                                  # flow grid 128 x 168
    monkeypatch.setenv("WOFT_FUSED", "1" if backend == "device" else "0")
    sd = synth.make_state_dict(seed=7)
    template = synth.make_template(H, W, seq_id=9)
    frames = [synth.make_frame(template, t) for t in (1, 2, 3)]
    mask = synth.make_init_mask(H, W)
    conf = load_config(ROOT / "pytracking" / "configs" / "WOFT.py")
    conf.flow_config.model = sd
    conf.flow_config.iters = iters
    conf.flow_config.padding_mode = "crop"
    tracker = conf.tracker_class(conf)
    tracker.init(template, mask)
    ref = tracker_ref.TrackerRef(sd, iters=iters, padding_mode="crop")
    ref.init(template, mask)
    for f in frames:
        Hg, mg = tracker.track(f)
        Hr, mr = ref.track(f)
        assert tracker.flower.last_flow_shape["H"] == 128 and tracker.flower.last_flow_shape["W"] == 168
        assert mg.lost == mr.lost and mg.N_lost == mr.N_lost
        assert _corners_err(Hg, Hr, H, W) < 1.0


def test_tracker_without_subsampler_fits_every_kept_correspondence(monkeypatch):
    """Configs without `subsampler_fn` (N up to the in-mask count) fit all kept correspondences: the
    device back end streams them through the multi-workgroup fit; it must equal the callable back end."""
    from pytracking.utils.config import load_config
    H, W, iters = 256, 320, 3                                  # mask 128 x 160 = 20480 correspondences > 8192
    sd = synth.make_state_dict(seed=7)
    template = synth.make_template(H, W, seq_id=10)
    frames = [synth.make_frame(template, t) for t in (1, 2)]
    mask = synth.make_init_mask(H, W)
    outs = {}
    for fused in (True, False):
        monkeypatch.setenv("WOFT_FUSED", "1" if fused else "0")
        conf = load_config(ROOT / "pytracking" / "configs" / "WOFT_IRLS.py")
        conf.flow_config.model = sd
        conf.flow_config.iters = iters
        conf.flow_config.precision = "bf16x3"
        conf.subsampler_fn = None
        trk = conf.tracker_class(conf)
        assert (trk._fused is not None) == fused
        trk.init(template, mask)
        outs[fused] = [trk.track(f) for f in frames]
        if fused:
            assert trk._fb["fit_ws"] is not None and trk._fb["pa"].shape[0] == H * W
    for (Ha, ma), (Hb, mb) in zip(outs[True], outs[False]):
        assert ma.lost == mb.lost and bool(ma.global_H_success) == bool(mb.global_H_success)
        assert np.array_equal(Ha, Hb), np.abs(Ha - Hb).max()


def test_operator_outputs_are_not_aliased():
    """compute_flow returns tensors of its own (the reference returns fresh tensors): a forward / backward pair must
    not share memory; borrow=True (the tracker's private fast path) may."""
    from pytracking.utils.config import load_config
    sd = synth.make_state_dict(seed=7)
    conf = load_config(ROOT / "pytracking" / "configs" / "WOFT.py")
    fc = conf.flow_config
    fc.model, fc.iters = sd, 2
    flower = fc.of_class(fc)
    a = synth.make_template(128, 160, seq_id=1)
    b = synth.make_frame(a, 2)
    f, wf = flower.compute_flow(a, b, mode="flow")
    f0 = f.clone()
    gflow, wg = flower.compute_flow(b, a, mode="flow")
    assert f.data_ptr() != gflow.data_ptr() and wf.data_ptr() != wg.data_ptr()
    assert torch.equal(f, f0) and not torch.equal(f, gflow)
    _, d1, w1 = flower.compute_flow(a, b, mode="TC", do_sigmoid=True)
    _, d2, w2 = flower.compute_flow(b, a, mode="TC", do_sigmoid=True)
    assert d1.data_ptr() != d2.data_ptr() and w1.data_ptr() != w2.data_ptr()
    _, d3, _ = flower.compute_flow(a, b, mode="TC", do_sigmoid=True, borrow=True)
    _, d4, _ = flower.compute_flow(b, a, mode="TC", do_sigmoid=True, borrow=True)
    assert d3.data_ptr() == d4.data_ptr()
    with pytest.raises(TypeError):
        flower.compute_flow(a.astype(np.float32), b.astype(np.float32))


@pytest.mark.parametrize("backend", ["device", "callables"])
def test_tracker_on_real_720p_frames_vs_reference_tracker(golden_dir, monkeypatch, backend):
    """BASELINE config 2 (720p sequence, RAFT full 12 iterations + IRLS homography) at its real size on real frames: the
    HIP tracker with the IRLS config against the REFERENCE's own tracker (config ablation_08.py) on three frames of the
    reference's demo sequence (tests/golden/real_720p.npz): box corners < 1 px, same lost flags.  Default arithmetic
    (bf16x3, volume-free correlation, weight head under the drawn correspondences on the device back end)."""
    from pytracking.utils.config import load_config
    g = np.load(golden_dir / "real_720p.npz")
    monkeypatch.setenv("WOFT_FUSED", "1" if backend == "device" else "0")
    sd = synth.make_state_dict(seed=int(g["seed"]))
    conf = load_config(ROOT / "pytracking" / "configs" / "WOFT_IRLS.py")
    conf.flow_config.model, conf.flow_config.iters, conf.flow_config.precision = sd, int(g["iters"]), "bf16x3"
    tracker = conf.tracker_class(conf)
    assert (tracker._fused is not None) == (backend == "device")
    tracker.init(g["frame1"], g["mask"])
    H, W = g["mask"].shape
    for i, f in enumerate((g["frame2"], g["frame3"])):
        Hg, mg = tracker.track(f)
        lost, n_lost, ok = g["track_meta"][i]
        assert (mg.lost, mg.N_lost, bool(mg.global_H_success)) == (bool(lost), int(n_lost), bool(ok)), i
        err = _corners_err(Hg, g["track_H"][i], H, W)
        print(f"real 720p frame {i}: corners within {err:.3f} px of the reference tracker")
        assert err < 1.0, (i, err)


def test_host_frames_through_pinned_staging_equal_device_frames():
    """track() on numpy frames (what WOFT_demo.py:61-78 hands it: the pinned double-buffered upload of
    woft_amd.tracker._FrameUploader, incl. a non-contiguous view) == track() on frames already on the device, bit for bit --
    also across forced-lost frames, whose local stage reads the PREVIOUS frame's device buffer (TRK:181-184)."""
    from pytracking.utils.config import load_config
    H, W, iters = 128, 160, 3
    sd = synth.make_state_dict(seed=7)
    template = synth.make_template(H, W, seq_id=2)
    frames = [synth.make_frame(template, t) for t in range(1, 7)]
    mask = synth.make_init_mask(H, W)

    def run(as_host):
        conf = load_config(ROOT / "pytracking" / "configs" / "WOFT.py")
        conf.flow_config.model, conf.flow_config.iters = sd, iters
        trk = conf.tracker_class(conf)
        inner, seen = trk._global_stage, {"i": 0}

        def overruled(frame, prewarp_H):
            fit = inner(frame, prewarp_H)
            seen["i"] += 1
            if seen["i"] in (3, 4):
                fit.success = False
            return fit
        trk._global_stage = overruled
        trk.init(template, mask)
        out = []
        for k, f in enumerate(frames):
            if as_host:
                x = np.ascontiguousarray(f[:, ::-1])[:, ::-1] if k == 1 else f      # (k == 1: a negative-stride view)
            else:
                x = torch.from_numpy(f).cuda()
            Hc, m = trk.track(x)
            out.append((Hc.copy(), bool(m.lost)))
        return out

    a, b = run(True), run(False)
    assert [l for _, l in a] == [l for _, l in b] and any(l for _, l in a)
    for (Ha, _), (Hb, _) in zip(a, b):
        assert np.array_equal(Ha, Hb)


@pytest.mark.gpu
@torch.no_grad()
def test_graph_replay_keeps_the_deferred_weight_head():
    """Flow config key `graph` with the tracker's default weight head (evaluated after the draw, under the drawn correspondences only):
    the flow's launch list up to the upsampling is replayed as one hipGraph, the deferred head follows eagerly -- same poses as the
    all-eager tracker, bit for bit.  (Until round 4 the graph path switched the deferral off: its flows did the head on the whole
    mask region, which is what made `alt_graph` read 5 % below the eager run.)"""
    from pytracking.utils.config import load_config
    H, W, iters = 128, 160, 3
    sd = synth.make_state_dict(seed=7)
    template = synth.make_template(H, W, seq_id=2)
    frames = [synth.make_frame(template, t) for t in range(1, 7)]
    mask = synth.make_init_mask(H, W)
    outs = {}
    for graph in (False, True):
        conf = load_config(ROOT / "pytracking" / "configs" / "WOFT.py")
        conf.flow_config.model, conf.flow_config.iters, conf.flow_config.graph = sd, iters, graph
        trk = conf.tracker_class(conf)
        trk.flower.defer_min_ratio = 0                   # (a frame this small has too few windows for the deferral to be chosen)
        assert trk.flower.use_graph == graph
        trk.init(template, mask)
        res = []
        for f in frames:
            Hc, m = trk.track(torch.from_numpy(f).cuda())
            assert trk.flower.weights_deferred and not m.lost
            res.append(Hc.copy())
        outs[graph] = res
        if graph:
            plan = next(iter(trk.flower.engine._plans.values()))
            assert any(g is not None for g in plan._graphs.values()), "nothing was replayed"
    for a, b in zip(outs[False], outs[True]):
        assert np.array_equal(a, b)


def test_reference_form_config_takes_the_device_solver_with_identical_results(monkeypatch):
    """A config written like the reference's default (inline, untagged estimator / subsampler / re-detection callables:
    tests/configs/inline_wlsq.py; configs/..._wLSq.py:14-53) is recognised by woft_amd.probe and runs the device back end --
    with the homographies, lost flags and metas the SAME config produces on the callable back end (WOFT_FUSED=0), and those of
    the shipped presets config."""
    from pytracking.utils.config import load_config
    H, W, iters, nframes = 128, 160, 4, 5
    sd = synth.make_state_dict(seed=7)
    template = synth.make_template(H, W, seq_id=3)
    frames = [synth.make_frame(template, t) for t in range(1, nframes + 1)]
    mask = synth.make_init_mask(H, W)

    def run(path, fused, **kw):
        monkeypatch.setenv("WOFT_FUSED", "1" if fused else "0")
        conf = load_config(path)
        conf.flow_config.model, conf.flow_config.iters = sd, iters
        for k, v in kw.items():
            if v is not None:
                setattr(conf.flow_config, k, v)
        trk = conf.tracker_class(conf)
        trk.init(template, mask)
        return trk, [trk.track(f) for f in frames]

    inline = ROOT / "tests" / "configs" / "inline_wlsq.py"
    t_dev, r_dev = run(inline, True, precision="bf16x3")
    t_call, r_call = run(inline, False, precision="bf16x3")
    t_pre, r_pre = run(ROOT / "pytracking" / "configs" / "WOFT.py", True)
    assert t_dev._fused is not None and "probed" in t_dev.solver_decision and t_dev.solver_decision.count("probed") == 3
    assert t_dev._fused["n_draw"] == 500 and t_dev._fused["thr"] == 5.0 and t_dev._fused["min_frac"] == 0.2
    assert t_call._fused is None and t_pre._fused is not None and "tagged" in t_pre.solver_decision
    for (Ha, ma), (Hb, mb), (Hc, mc) in zip(r_dev, r_call, r_pre):
        assert np.array_equal(Ha, Hc) and ma.lost == mc.lost == mb.lost and ma.N_lost == mb.N_lost
        assert _corners_err(Ha, Hb, H, W) < 1e-2           # (callable back end: torch's QR-free path on the same kernels, fp32 H)
    # no `precision` key (an unmodified reference flow config): bf16x3 by the built-in default -- the very same run --, and exact
    # fp32 products one key away
    t_def, r_def = run(inline, True, precision=None)
    assert t_def.flower.precision == "bf16x3" and "built-in default" in t_def.flower.precision_source and t_def._fused is not None
    t_fp32, r_fp32 = run(inline, True, precision="fp32")
    assert t_fp32.flower.precision == "fp32"
    for (Ha, _), (Hb, _), (Hc, _) in zip(r_fp32, r_dev, r_def):
        assert _corners_err(Ha, Hb, H, W) < 0.05 and np.array_equal(Hb, Hc)


def test_host_frames_keep_the_previous_frame_intact():
    """_FrameUploader's lifetime contract: the device frame a host-frame track() call keeps as prev_img survives the NEXT call
    (two alternating buffers) -- the local stage of a lost frame reads it -- and is reused by the call after that; frames and
    results equal those of device-resident frames."""
    from pytracking.utils.config import load_config
    H, W, iters = 128, 160, 3
    sd = synth.make_state_dict(seed=7)
    template = synth.make_template(H, W, seq_id=3)
    frames = [synth.make_frame(template, t) for t in range(1, 5)]
    mask = synth.make_init_mask(H, W)

    def tracker():
        conf = load_config(ROOT / "pytracking" / "configs" / "WOFT.py")
        conf.flow_config.model, conf.flow_config.iters = sd, iters
        conf.redet_success_fn = lambda *a: False            # every frame lost: the local stage reads frame t - 1
        t = conf.tracker_class(conf)
        t.init(template, mask)
        return t
    a, b = tracker(), tracker()
    kept = []
    for f in frames:
        Ha, ma = a.track(f)                                  # numpy frame per call
        Hb, mb = b.track(torch.from_numpy(f).cuda())         # device frame
        assert np.array_equal(Ha, Hb) and ma.lost and mb.lost
        kept.append((a.prev_img, f))
        assert torch.equal(a.prev_img.cpu(), torch.from_numpy(f))
        if len(kept) >= 2:                                   # the frame before is still intact after this call
            assert torch.equal(kept[-2][0].cpu(), torch.from_numpy(kept[-2][1]))
    assert kept[0][0].data_ptr() == kept[2][0].data_ptr() != kept[1][0].data_ptr()


def test_unweighted_estimator_and_constant_verdict_on_the_device_solver(monkeypatch):
    """The reference's "plain LSq / always re-detected" ablation in its inline form (tests/configs/inline_plain_always.py): the probe
    finds an UNWEIGHTED fit and a constant verdict; the device back end then skips the weight head (nobody reads a weight) and must
    give the homographies of the callable back end, which evaluates everything and calls the config's functions."""
    from pytracking.utils.config import load_config
    H, W, iters = 128, 160, 4
    sd = synth.make_state_dict(seed=7)
    template = synth.make_template(H, W, seq_id=3)
    frames = [synth.make_frame(template, t) for t in range(1, 5)]
    mask = synth.make_init_mask(H, W)
    res = {}
    for fused in (True, False):
        monkeypatch.setenv("WOFT_FUSED", "1" if fused else "0")
        conf = load_config(ROOT / "tests" / "configs" / "inline_plain_always.py")
        conf.flow_config.model, conf.flow_config.iters = sd, iters
        trk = conf.tracker_class(conf)
        assert (trk._fused is not None) == fused
        if fused:
            assert trk._fused["weighted"] is False and trk._fused["const_verdict"] is True
        trk.init(template, mask)
        res[fused] = [trk.track(f) for f in frames]
    for (Ha, ma), (Hb, mb) in zip(res[True], res[False]):
        assert _corners_err(Ha, Hb, H, W) < 1e-2 and not ma.lost and not mb.lost and ma.global_H_success and mb.global_H_success


@pytest.mark.parametrize("name", ["plain", "never"])
@pytest.mark.parametrize("backend", ["device", "callables"])
def test_tracker_vs_reference_tracker_ablation_runs(golden_dir, monkeypatch, name, backend):
    """Runs of the REFERENCE's tracker with its plain-LSq and never-re-detected ablation configs (gen_tracker (d)) against the
    HIP tracker driven by the same configs in inline form (tests/configs/inline_ablations.py): on the device back end the probe
    maps them to an unweighted fit / a constant verdict, on the callable back end the functions themselves run."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("inline_ablations", str(ROOT / "tests" / "configs" / "inline_ablations.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    g = np.load(golden_dir / "tracker_ref_runs_ablations.npz")
    monkeypatch.setenv("WOFT_FUSED", "1" if backend == "device" else "0")
    conf = mod.get_config(name)
    conf.flow_config.model, conf.flow_config.iters = synth.make_state_dict(seed=int(g["seed"])), int(g["iters"])
    tracker = conf.tracker_class(conf)
    assert (tracker._fused is not None) == (backend == "device")
    if backend == "device":
        assert tracker._fused["weighted"] == (name != "plain") and tracker._fused["const_verdict"] == (None if name == "plain" else False)
    mask = g[f"{name}_mask"]
    H, W = mask.shape
    tracker.init(g[f"{name}_template"], mask)
    for i, f in enumerate(g[f"{name}_frames"]):
        Hg, mg = tracker.track(f)
        lost, n_lost, ok, has_local = g[f"{name}_meta"][i]
        assert (mg.lost, mg.N_lost, bool(mg.global_H_success)) == (bool(lost), int(n_lost), bool(ok)), (name, i)
        assert _corners_err(Hg, g[f"{name}_H"][i], H, W) < 1.0, (name, i)
        assert hasattr(mg, "H_local_cur2init") == bool(has_local)
        if has_local:
            assert _corners_err(mg.H_local_cur2init, g[f"{name}_Hlocal_{i}"], H, W) < 1.0
