"""CPU tests: the oracle (torch-CPU restatement) against golden vectors produced by the
imported reference (oracle/gen_golden.py).  Tolerances: SURVEY 8(d) -- restatement vs
reference EPE mean <= 1e-4 px, max <= 1e-3 px; weights <= 1e-5 (sigmoid) / 1e-4 (logits)."""
import json

import numpy as np
import pytest
import torch

from oracle import hfit_ref, raft_ref, tracker_ref
from woft_amd import synth


def _t(a):
    return torch.from_numpy(a[:, :, ::-1].copy()).permute(2, 0, 1).float()[None]


def _epe(a, b):
    d = torch.as_tensor(a) - torch.as_tensor(b)
    e = torch.sqrt((d ** 2).sum(dim=-3))
    return float(e.mean()), float(e.max())


def test_state_dict_keys(golden_dir):
    keys = json.loads((golden_dir / "state_dict_keys.json").read_text())
    for name, kw in (("weighted_full", dict(small=False, weighted=True)),
                     ("plain_small", dict(small=True, weighted=False)),
                     ("weighted_small", dict(small=True, weighted=True))):
        sd = synth.make_state_dict(seed=1, **kw)
        assert {k: list(v.shape) for k, v in sd.items()} == keys[name]
    assert keys["weighted_full_nparams"] == 5558721
    assert keys["plain_small_nparams"] == 990162


@torch.no_grad()
def test_full_model_with_intermediates(golden_dir):
    g = np.load(golden_dir / "flow_full_128x160_it4.npz")
    sd = synth.make_state_dict(seed=int(g["seed"]), small=False, weighted=True)
    tr = {}
    out = raft_ref.raft_forward(sd, _t(g["img1"]), _t(g["img2"]), int(g["iters"]), trace=tr)
    assert np.abs(tr["fmap1"].numpy() - g["fmap1"]).max() < 2e-5
    assert np.abs(tr["fmap2"].numpy() - g["fmap2"]).max() < 2e-5
    assert np.abs(tr["net0"].numpy() - g["net0"]).max() < 1e-5
    assert np.abs(tr["inp"].numpy() - g["inp"]).max() < 1e-5
    for l in range(4):
        assert np.abs(tr["pyr"][l][:24, 0].numpy() - g[f"pyr{l}_rows"]).max() < 1e-4
    assert np.abs(tr["lookups"][0].numpy() - g["lookup0"]).max() < 1e-4
    assert np.abs(tr["nets"][0].numpy() - g["net1"]).max() < 1e-5
    m, mx = _epe(out["flow_up"], g["flow_up"])
    assert m < 1e-4 and mx < 1e-3, (m, mx)
    m, mx = _epe(out["flow_low"], g["flow_low"])
    assert m < 1e-4 and mx < 1e-3
    assert np.abs(out["weights_up"].numpy() - g["w_up"]).max() < 1e-4
    assert np.abs(out["weights_low"].numpy() - g["w_low"]).max() < 1e-4


@torch.no_grad()
@pytest.mark.parametrize("name,small,weighted", [
    ("flow_full_136x200_it12", False, True),
    ("flow_full_136x200_it32", False, True),
    ("flow_small_128x160_it4", True, False),
    ("flow_wsmall_128x160_it4", True, True),
])
def test_model_outputs(golden_dir, name, small, weighted):
    g = np.load(golden_dir / f"{name}.npz")
    sd = synth.make_state_dict(seed=int(g["seed"]), small=small, weighted=weighted)
    out = raft_ref.raft_forward(sd, _t(g["img1"]), _t(g["img2"]), int(g["iters"]), small=small, weighted=weighted)
    m, mx = _epe(out["flow_up"], g["flow_up"])
    assert m < 1e-4 and mx < 1e-3, (m, mx)
    if weighted:
        assert np.abs(out["weights_up"].numpy() - g["w_up"]).max() < 2e-4


def test_lookup_handmade(golden_dir):
    g = np.load(golden_dir / "lookup_handmade.npz")
    v = torch.from_numpy(g["vol0"])
    pyr = [v]
    for _ in range(3):
        v = torch.nn.functional.avg_pool2d(v, 2, stride=2)
        pyr.append(v)
    coords = torch.from_numpy(g["coords"])[None]
    out = raft_ref.corr_lookup(pyr, coords, 4)
    assert np.abs(out.numpy() - g["out"]).max() < 1e-4
    # x-major window: source pixel 0 sits at (x=8, y=6) on the pure ramp 100*y + x; window
    # element (i=0, j=1) samples x-4, y-3 -> 304 (SURVEY 8c fixture plan (3))
    assert abs(float(out[0, 0 * 9 + 1, 0, 0]) - 304.0) < 1e-3
    assert abs(float(out[0, 1 * 9 + 0, 0, 0]) - 205.0) < 1e-3
    # the direct pixel-coordinate formula (what the HIP kernel implements) agrees to ~5e-5 relative to O(1000) values
    out2 = raft_ref.lookup_direct(pyr, coords, 4)
    tol = 1e-5 * np.abs(g["out"]).max()         # ramp values reach ~1.7e3: relative fp32 tolerance
    assert np.abs(out2.numpy() - g["out"]).max() < tol
    assert np.abs(out2.numpy() - out.numpy()).max() < tol


@torch.no_grad()
def test_wrapper_boundary(golden_dir):
    g = np.load(golden_dir / "wrapper_tc_128x160_it4.npz")
    sd = synth.make_state_dict(seed=int(g["seed"]))
    src, dst, w = raft_ref.compute_flow(sd, g["img1"], g["img2"], int(g["iters"]), mode="TC", do_sigmoid=True)
    assert src.dtype == torch.int64 and tuple(src.shape) == (2, 128 * 160)
    assert np.array_equal(src.numpy(), g["src"])
    assert np.abs(dst.numpy() - g["dst"]).max() < 1e-3
    assert np.abs(w.numpy() - g["w"]).max() < 1e-5
    fl, wl = raft_ref.compute_flow(sd, g["img1"], g["img2"], int(g["iters"]), mode="flow")
    assert np.abs(fl.numpy() - g["flow"]).max() < 1e-3
    assert np.abs(wl.numpy() - g["w_logit"]).max() < 1e-4
    a2, b2 = g["img1"][:125, :157].copy(), g["img2"][:125, :157].copy()
    s2, d2, w2 = raft_ref.compute_flow(sd, a2, b2, int(g["iters"]), mode="TC", do_sigmoid=True, padding_mode="RAFT")
    assert np.array_equal(s2.numpy(), g["src_pad"])
    assert np.abs(d2.numpy() - g["dst_pad"]).max() < 1e-3
    assert np.abs(w2.numpy() - g["w_pad"]).max() < 1e-5
    a4, b4 = g["img1"][:, :157].copy(), g["img2"][:, :157].copy()       # 'crop': outputs keep the cropped 128x152
    s4, d4, w4 = raft_ref.compute_flow(sd, a4, b4, int(g["iters"]), mode="TC", do_sigmoid=True, padding_mode="crop")
    assert np.array_equal(s4.numpy(), g["src_crop"]) and tuple(s4.shape) == (2, 128 * 152)
    assert np.abs(d4.numpy() - g["dst_crop"]).max() < 1e-3
    assert np.abs(w4.numpy() - g["w_crop"]).max() < 1e-5


def _corner_err(Ha, Hb):
    c = np.array([[100, 80, 1], [1800, 80, 1], [1800, 1000, 1], [100, 1000, 1.0]]).T
    pa = Ha @ c
    pb = Hb @ c
    return np.abs(pa[:2] / pa[2] - pb[:2] / pb[2]).max()


@pytest.mark.parametrize("case", ["n4", "n500", "n4096", "degen"])
def test_hfit(golden_dir, case):
    g = np.load(golden_dir / "hfit.npz")
    a, b, w = (torch.from_numpy(g[f"{case}_{k}"]) for k in "abw")
    tol = 1e-3 if case != "degen" else 5.0     # degenerate set: ill-conditioned by construction
    H = hfit_ref.find_homography_nonhomogeneous_QR(a, b, w)[0].numpy().astype(np.float64)
    assert _corner_err(H, g[f"{case}_qr_w"][0].astype(np.float64)) < tol
    H = hfit_ref.find_homography_nonhomogeneous_QR(a, b, None)[0].numpy().astype(np.float64)
    assert _corner_err(H, g[f"{case}_qr_now"][0].astype(np.float64)) < tol
    H = hfit_ref.find_homography_IRLSq_QR(a, b, w)[0].numpy().astype(np.float64)
    assert _corner_err(H, g[f"{case}_irls_l1"][0].astype(np.float64)) < max(tol, 2e-2)
    Hh = hfit_ref.find_homography_IRLSq_QR(a, b, w, reweighting_fn=lambda r: hfit_ref.IRLSq_Huber(r, k=2))
    assert _corner_err(Hh[0].numpy().astype(np.float64), g[f"{case}_irls_huber2"][0].astype(np.float64)) < tol
    Hh = hfit_ref.find_homography_IRLSq_QR(a, b, w, reweighting_fn=lambda r: hfit_ref.IRLSq_Huber(r, k=0.01))
    assert _corner_err(Hh[0].numpy().astype(np.float64), g[f"{case}_irls_huber001"][0].astype(np.float64)) < max(tol, 2e-2)
    Hq = torch.from_numpy(g[f"{case}_qr_w"])
    e = hfit_ref.torch_proj_errors(Hq, a.permute(0, 2, 1), b.permute(0, 2, 1))
    assert np.allclose(e.numpy(), g[f"{case}_projerr"], rtol=1e-5, atol=1e-4)


def test_hfit_small_pieces(golden_dir):
    g = np.load(golden_dir / "hfit.npz")
    r = torch.from_numpy(g["huber_in"])
    assert np.array_equal(hfit_ref.IRLSq_Huber(r.clone(), k=1).numpy(), g["huber_k1"])
    assert np.array_equal(hfit_ref.IRLSq_L1(r.clone()).numpy(), g["l1"])
    assert np.allclose(hfit_ref.compose_H(g["compose_in1"], g["compose_in2"]), g["compose_12"], rtol=0, atol=1e-12)
    assert np.allclose(hfit_ref.compose_H(g["compose_in1"], g["compose_in2"], g["compose_in1"]), g["compose_121"], rtol=0, atol=1e-12)
    with pytest.raises(AssertionError):
        hfit_ref.find_homography_nonhomogeneous_QR(torch.zeros(1, 3, 2), torch.zeros(1, 3, 2))


def test_sobol(golden_dir):
    g = np.load(golden_dir / "sobol.npz")
    for N in (400, 501, 600, 2000, 518400):
        m = hfit_ref.sobol_subsample_mask(N, 500)
        assert np.array_equal(np.nonzero(m)[0], g[f"n{N}"])


def _box_err(Ha, Hb, H, W):
    c = np.array([[W / 4, H / 4, 1], [3 * W / 4, H / 4, 1], [3 * W / 4, 3 * H / 4, 1], [W / 4, 3 * H / 4, 1.0]]).T
    pa, pb = np.linalg.inv(Ha) @ c, np.linalg.inv(Hb) @ c
    return np.abs(pa[:2] / pa[2] - pb[:2] / pb[2]).max()


@torch.no_grad()
@pytest.mark.parametrize("name,estimator", [("woft", "qr"), ("lost", "qr"), ("irls", "irls_huber2")])
def test_tracker_state_machine_vs_reference_runs(golden_dir, name, estimator):
    """SURVEY 8c fixture (7): oracle/tracker_ref.py against runs of the reference's own YAOFTrackerSingleControl
    (reference configs WOFT.py / ablation_08.py, functional numpy cv2 stub; oracle/gen_golden.py: gen_tracker) --
    per frame the homography, lost / N_lost / global_H_success, the pre-warp used and the local-branch result."""
    g = np.load(golden_dir / "tracker_ref_runs.npz")
    sd = synth.make_state_dict(seed=int(g["seed"]))
    ref = tracker_ref.TrackerRef(sd, iters=int(g["iters"]), estimator=estimator)
    ref.force_fail = tuple(int(i) for i in g[f"{name}_force_fail"])
    ref.init(g[f"{name}_template"], g[f"{name}_mask"])
    H, W = g[f"{name}_mask"].shape
    for i, f in enumerate(g[f"{name}_frames"]):
        Hc, m = ref.track(f)
        lost, n_lost, ok, has_local = g[f"{name}_meta"][i]
        assert (m.lost, m.N_lost, bool(m.global_H_success)) == (bool(lost), int(n_lost), bool(ok)), (name, i)
        assert _box_err(Hc, g[f"{name}_H"][i], H, W) < 0.02, (name, i)
        assert _box_err(m.H_global_cur2init, g[f"{name}_Hglobal_{i}"], H, W) < 0.02
        assert np.allclose(m.last_good_H2init, g[f"{name}_lastgood_{i}"], atol=1e-3)
        assert hasattr(m, "H_local_cur2init") == bool(has_local)
        if has_local:
            assert _box_err(m.H_local_cur2init, g[f"{name}_Hlocal_{i}"], H, W) < 0.02


@torch.no_grad()
@pytest.mark.parametrize("name,estimator,fail_all", [("plain", "plain_qr", False), ("never", "qr", True)])
def test_tracker_state_machine_vs_reference_ablation_runs(golden_dir, name, estimator, fail_all):
    """Runs of the reference's tracker with two of its ablation configs (gen_tracker (d)): the unweighted fit of
    ..._noreliableinl_plainLSq.py and the `return False` re-detection test of ..._neverwarp_wLSq.py (every frame takes the
    frame-to-frame branch)."""
    g = np.load(golden_dir / "tracker_ref_runs_ablations.npz")
    sd = synth.make_state_dict(seed=int(g["seed"]))
    ref = tracker_ref.TrackerRef(sd, iters=int(g["iters"]), estimator=estimator)
    n = len(g[f"{name}_frames"])
    ref.force_fail = tuple(range(n)) if fail_all else ()
    ref.init(g[f"{name}_template"], g[f"{name}_mask"])
    H, W = g[f"{name}_mask"].shape
    for i, f in enumerate(g[f"{name}_frames"]):
        Hc, m = ref.track(f)
        lost, n_lost, ok, has_local = g[f"{name}_meta"][i]
        assert (m.lost, m.N_lost, bool(m.global_H_success)) == (bool(lost), int(n_lost), bool(ok)), (name, i)
        assert bool(lost) == fail_all
        assert _box_err(Hc, g[f"{name}_H"][i], H, W) < 0.02, (name, i)
        assert hasattr(m, "H_local_cur2init") == bool(has_local)


DEGENERATE = ["constant", "constant_vs_texture", "saturated", "identical"]


@torch.no_grad()
@pytest.mark.parametrize("name", DEGENERATE)
def test_degenerate_inputs(golden_dir, name):
    """Inputs on which InstanceNorm's variance is zero (extractor.py:28-32 on a constant image), large saturated regions,
    identical frames: the oracle against the REFERENCE's outputs on them."""
    g = np.load(golden_dir / "degenerate_128x160_it4.npz")
    sd = synth.make_state_dict(seed=int(g["seed"]), small=False, weighted=True)
    out = raft_ref.raft_forward(sd, _t(g[f"{name}_img1"]), _t(g[f"{name}_img2"]), int(g["iters"]))
    m, mx = _epe(out["flow_up"], g[f"{name}_flow_up"])
    assert m < 1e-4 and mx < 1e-3, (m, mx)
    assert np.abs(out["weights_up"].numpy() - g[f"{name}_w_up"]).max() < 2e-4


@torch.no_grad()
def test_real_frames_720p(golden_dir):
    """BASELINE config 2 at its real size on real frames (the reference's demo sequence): oracle flow vs the reference's."""
    g = np.load(golden_dir / "real_720p.npz")
    sd = synth.make_state_dict(seed=int(g["seed"]), small=False, weighted=True)
    out = raft_ref.raft_forward(sd, _t(g["frame1"]), _t(g["frame3"]), int(g["iters"]))
    s = int(g["stride"])
    m, mx = _epe(out["flow_low"], g["flow_low"])
    assert m < 1e-4 and mx < 2e-3, (m, mx)
    m, mx = _epe(out["flow_up"][..., ::s, ::s], g["flow_up_s4"])
    assert m < 1e-4 and mx < 2e-3, (m, mx)
    assert np.abs(out["flow_up"].double().mean(-1).numpy() - g["flow_up_rowmean"]).max() < 1e-4
    assert np.abs(out["weights_up"][..., ::s, ::s].numpy() - g["w_up_s4"]).max() < 3e-4
    assert np.abs(out["weights_up"].double().mean(-1).numpy() - g["w_up_rowmean"]).max() < 1e-4


def _crc(a):
    import zlib
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


@torch.no_grad()
def test_config1_small_480x640(golden_dir):
    """BASELINE configs[0] (plain RAFT-small, 4 iterations, 480 x 640): oracle vs the reference's outputs."""
    from oracle.gen_golden import pair
    g = np.load(golden_dir / "cfg0_small_480x640_it4.npz")
    a, b = pair(int(g["H"]), int(g["W"]), seed=int(g["pair_seed"]), shift=tuple(int(v) for v in g["shift"]))
    assert _crc(a) == int(g["crc_img1"]) and _crc(b) == int(g["crc_img2"])
    sd = synth.make_state_dict(seed=int(g["seed"]), small=True, weighted=False)
    out = raft_ref.raft_forward(sd, _t(a), _t(b), int(g["iters"]), small=True, weighted=False)
    s = int(g["stride"])
    m, mx = _epe(out["flow_low"], g["flow_low"])
    assert m < 1e-4 and mx < 1e-3, (m, mx)
    m, mx = _epe(out["flow_up"][..., ::s, ::s], g["flow_up_s"])
    assert m < 1e-4 and mx < 1e-3, (m, mx)
    assert np.abs(out["flow_up"].double().mean(-1).numpy() - g["flow_up_rowmean"]).max() < 1e-4


@torch.no_grad()
def test_metric_resolution_1080p(golden_dir):
    """The metric's own configuration (1080 x 1920, WeightedRAFT-full, 12 iterations): oracle vs the reference's outputs on
    the regenerated synthetic pair (CRC-pinned).  ~30-60 s on 8 cores: the one slow test of the CPU suite."""
    g = np.load(golden_dir / "metric_1080p_it12.npz")
    H, W = int(g["H"]), int(g["W"])
    a = synth.make_template(H, W, seq_id=int(g["seq_id"]))
    b = synth.make_frame(a, int(g["t"]))
    assert _crc(a) == int(g["crc_img1"]) and _crc(b) == int(g["crc_img2"])
    sd = synth.make_state_dict(seed=int(g["seed"]), small=False, weighted=True)
    out = raft_ref.raft_forward(sd, _t(a), _t(b), int(g["iters"]))
    s = int(g["stride"])
    m, mx = _epe(out["flow_low"], g["flow_low"])
    assert m < 1e-4 and mx < 2e-3, (m, mx)
    m, mx = _epe(out["flow_up"][..., ::s, ::s], g["flow_up_s"])
    assert m < 1e-4 and mx < 2e-3, (m, mx)
    assert np.abs(out["flow_up"].double().mean(-1).numpy() - g["flow_up_rowmean"]).max() < 1e-4
    assert np.abs(out["weights_up"][..., ::s, ::s].numpy() - g["w_up_s"]).max() < 3e-4
    assert np.abs(out["weights_up"].double().mean(-1).numpy() - g["w_up_rowmean"]).max() < 1e-4


@torch.no_grad()
def test_weight_heads_other_than_the_shipped_one(golden_dir):
    """class_params.weight_head_structure (weighted_raft.py:318-345): the oracle's head follows the state-dict's layers --
    (channels, kernel) tuples with any odd kernel, plain channel counts (3x3) -- against the reference built with each structure."""
    g = np.load(golden_dir / "weight_heads_128x160_it3.npz")
    from oracle.gen_golden import HEAD_STRUCTURES
    assert sorted(HEAD_STRUCTURES) == list(g["names"])
    for name, st in HEAD_STRUCTURES.items():
        sd = synth.make_state_dict(seed=int(g["seed"]), weight_head_structure=st)
        out = raft_ref.raft_forward(sd, _t(g["img1"]), _t(g["img2"]), int(g["iters"]))
        m, mx = _epe(out["flow_up"], g[f"{name}_flow_up"])
        assert m < 1e-4 and mx < 1e-3, (name, m, mx)
        assert np.abs(out["weights_low"].numpy() - g[f"{name}_w_low"]).max() < 1e-4, name
        assert np.abs(out["weights_up"].numpy() - g[f"{name}_w_up"]).max() < 1e-4, name
