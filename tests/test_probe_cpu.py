"""woft_amd.probe: which reference-format callables get the tracker's device back end (no GPU: the probe never launches a kernel)."""
import importlib.util
import sys
from pathlib import Path

import numpy as np
import torch

# one test below loads config files that live in the READ-ONLY reference tree: no __pycache__ may be written next to them (SURVEY
# line 67) -- switched off for the whole interpreter before the first import machinery runs on such a file, and the loader below
# compiles from source without touching any cache
sys.dont_write_bytecode = True

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from woft_amd import presets, probe  # noqa: E402
from woft_amd.homography import (IRLSq_Huber, IRLSq_L1, find_homography_IRLSq_QR,  # noqa: E402
                                 find_homography_nonhomogeneous_QR, torch_proj_errors)
from woft_amd.tracker import make_forward_compatible  # noqa: E402


def _module(path):
    spec = importlib.util.spec_from_file_location(path.stem, str(path))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _load_config_without_cache(path):
    """pytracking.utils.config.load_config's semantics (module.get_config()) for a file in a read-only tree: the source is compiled
    in memory, nothing is read from or written to a __pycache__ directory."""
    import types
    m = types.ModuleType("tracker_config")
    m.__file__ = str(path)
    exec(compile(Path(path).read_text(), str(path), "exec"), m.__dict__)
    return m.get_config()


def test_reference_form_config_is_recognised():
    m = _module(ROOT / "tests" / "configs" / "inline_wlsq.py")
    spec, how = probe.solver_spec(m.fit_homography, make_forward_compatible(m.draw_500), m.redetected)
    assert spec == dict(reweight=0, huber_k=0.0, n_irls=0, weighted=True, thr=5.0, min_frac=0.2, const_verdict=None, n_draw=500), (spec, how)
    assert how.count("probed") == 3
    m = _module(ROOT / "tests" / "configs" / "inline_irls.py")          # nested re-weighting function around IRLSq_Huber(k = 2)
    spec, how = probe.solver_spec(m.robust_fit, make_forward_compatible(m.sobol_500), m.inlier_test)
    assert spec == dict(reweight=2, huber_k=2.0, n_irls=5, weighted=True, thr=5.0, min_frac=0.2, const_verdict=None, n_draw=500), (spec, how)
    m = _module(ROOT / "tests" / "configs" / "inline_plain_always.py")   # weights=None handed to the library, `return True`
    spec, how = probe.solver_spec(m.find_homography, make_forward_compatible(m.subsampler), m.redet_success_fn)
    assert spec is not None and spec["weighted"] is False and spec["const_verdict"] is True and spec["n_draw"] == 500, (spec, how)


def test_presets_are_taken_by_tag_and_probe_agrees():
    for est, want in ((presets.estimator_weighted_lsq(), (0, 0.0, 0, True)), (presets.estimator_irls("huber", 2.0, 5), (2, 2.0, 5, True)),
                      (presets.estimator_irls("l1", n_iter=3), (1, 0.0, 3, True))):
        fn = lambda a, b, weights=None, _f=est: _f(a, b, weights)          # (an untagged wrapper: probed)
        assert probe.probe_estimator(fn) == want
    sub = presets.sobol_subsampler(300)
    assert probe.probe_subsampler(make_forward_compatible(lambda a, b, w: sub(a, b, w))) == 300
    red = presets.redetection_by_inliers(3.5, 0.35)
    assert probe.probe_redetection(lambda H, t, c, w: red(H, t, c, w)) == (3.5, 0.35)
    spec, how = probe.solver_spec(presets.estimator_weighted_lsq(), make_forward_compatible(presets.sobol_subsampler(500)),
                                  presets.redetection_by_inliers(5.0, 0.2))
    assert spec["n_draw"] == 500 and how.count("tagged") == 3


def test_irls_losses_by_behaviour():
    huber = lambda r: torch.where(r.abs() < 1.5, torch.ones_like(r), 1 / (r.abs() + 1e-8))      # (no library call inside)
    est = lambda a, b, weights=None: find_homography_IRLSq_QR(a, b, weights=weights, reweighting_fn=huber, n_iter=4)
    assert probe.probe_estimator(est) == (2, 1.5, 4, True)
    est = lambda a, b, weights=None: find_homography_IRLSq_QR(a, b, weights=weights, reweighting_fn=lambda r: IRLSq_Huber(r, k=2), n_iter=5)
    assert probe.probe_estimator(est) == (2, 2.0, 5, True)
    est = lambda a, b, weights=None: find_homography_IRLSq_QR(a, b, weights=weights, reweighting_fn=IRLSq_L1)
    assert probe.probe_estimator(est) == (1, 0.0, 5, True)
    cauchy = lambda r: 1 / (1 + r * r)
    est = lambda a, b, weights=None: find_homography_IRLSq_QR(a, b, weights=weights, reweighting_fn=cauchy)
    assert probe.probe_estimator(est) is None                       # (arbitrary loss: callable back end, woft_hfit_step)


def test_callables_that_do_something_else_keep_the_callable_back_end():
    # estimators: drops the weights / rescales the points / post-processes the result / calls the library twice
    lsq = find_homography_nonhomogeneous_QR
    assert probe.probe_estimator(lambda a, b, weights=None: lsq(a, b)) == (0, 0.0, 0, False)        # (an UNWEIGHTED fit: recognised as such)
    assert probe.probe_estimator(lambda a, b, weights=None: lsq(a, b, weights=weights * 2)) is None
    assert probe.probe_estimator(lambda a, b, weights=None: lsq(a * 2, b, weights=weights)) is None
    assert probe.probe_estimator(lambda a, b, weights=None: lsq(a, b, weights=weights) * 1.0) is None
    assert probe.probe_estimator(lambda a, b, weights=None: (lsq(a, b, weights=weights), lsq(a, b, weights=weights))[1]) is None
    assert probe.probe_estimator(lambda a, b, weights=None: torch.eye(3)[None]) is None
    # subsamplers: every second point / a weight-dependent draw / a random draw / scrambled Sobol / reordered output
    fc = make_forward_compatible
    assert probe.probe_subsampler(fc(lambda a, b, w: (a[:, ::2], b[:, ::2], w[:, ::2]))) is None

    def by_weight(a, b, w):
        keep = w[0] > 0.5
        return a[:, keep], b[:, keep], w[:, keep]
    assert probe.probe_subsampler(fc(by_weight)) is None

    def scrambled(a, b, w):
        n = a.shape[1]
        if n <= 100:
            return a, b, w
        u = torch.quasirandom.SobolEngine(dimension=1, scramble=True, seed=1).draw(100).numpy().flatten()
        keep = np.zeros(n, dtype=bool)
        keep[np.minimum(np.round(n * u).astype(np.int32), n - 1)] = True
        return a[:, keep], b[:, keep], w[:, keep]
    assert probe.probe_subsampler(fc(scrambled)) is None

    sob = presets.sobol_subsampler(200)
    assert probe.probe_subsampler(fc(lambda a, b, w: tuple(t.flip(1) for t in sob(a, b, w)))) is None
    # re-detection tests: swapped point sets / another statistic / weights used / no library call
    tpe = torch_proj_errors
    assert probe.probe_redetection(lambda H, t, c, w: (tpe(H, t[None], c[None]) <= 5).float().mean() > 0.2) is None
    assert probe.probe_redetection(lambda H, t, c, w: tpe(H, c[None], t[None]).median() < 5) is None
    assert probe.probe_redetection(lambda H, t, c, w: ((tpe(H, c[None], t[None]) <= 5).float() * w).sum() / w.sum() > 0.2) is None
    assert probe.probe_redetection(lambda H, t, c, w: True) == ("const", True)                      # (the reference's alwayswarp ablation)
    assert probe.probe_redetection(lambda H, t, c, w: bool(w.mean() > 0.5)) is None


def test_strict_and_non_strict_comparisons_are_told_apart():
    tpe = torch_proj_errors
    # `<` instead of `<=`: the largest passing error is the float below 5 -- an equivalent `<=` rule with that threshold
    got = probe.probe_redetection(lambda H, t, c, w: (tpe(H, c[None], t[None]) < 5).float().mean() > 0.2)
    assert got is not None and got[0] == float(np.nextafter(np.float32(5), np.float32(0))) and got[1] == 0.2
    # `>=` on the fraction: succeeds AT 0.2 -- not the device rule (`>`): refused
    assert probe.probe_redetection(lambda H, t, c, w: (tpe(H, c[None], t[None]) <= 5).float().mean() >= 0.2) is None


def test_the_references_own_config_files_through_the_shim():
    """In the build container (where /root/reference exists; skipped elsewhere): the reference's OWN tracker config files, loaded
    unmodified through the `pytracking` shim of this repository -- which back end each gets.  The default WOFT.py (= ..._wLSq.py), its
    downscale variants, the IRLS config and the ablations built from them are recognised; RANSAC / LiteFlowNet configs are out of scope."""
    import pytest
    ref = Path("/root/reference/pytracking/configs")
    if not ref.exists():
        pytest.skip("reference tree not present on this machine")
    from woft_amd.tracker import YAOFTrackerSingleControl
    load_config = _load_config_without_cache
    before = sorted(str(q) for q in ref.rglob("*.pyc"))
    want = {"WOFT.py": (0, True, None), "WOFT_downscale_2x.py": (0, True, None), "ablation_08.py": (2, True, None),
            "YAOFT_single_control_repRAFT_sub500_noreliableinl_wIRLSq.py": (2, True, None),
            "YAOFT_single_control_repRAFT_sub500_noreliableinl_plainLSq.py": (0, False, None),
            "YAOFT_single_control_repRAFT_sub500_alwayswarp_wLSq.py": (0, True, True),
            "YAOFT_single_control_repRAFT_sub500_neverwarp_plainLSq.py": (0, False, False)}
    for name, (rew, weighted, const) in want.items():
        conf = load_config(ref / name)
        assert conf.tracker_class is YAOFTrackerSingleControl and not conf.flow_config.precision
        spec, how = probe.solver_spec(conf.H_estimator, make_forward_compatible(conf.subsampler_fn), conf.redet_success_fn)
        assert spec is not None and how.count("probed") == 3, (name, how)
        assert (spec["reweight"], spec["weighted"], spec["const_verdict"], spec["n_draw"]) == (rew, weighted, const, 500), (name, spec)
        if const is None:
            assert (spec["thr"], spec["min_frac"]) == (5.0, 0.2)
    assert sorted(str(q) for q in ref.rglob("*.pyc")) == before, "the test wrote bytecode into the read-only reference tree"
