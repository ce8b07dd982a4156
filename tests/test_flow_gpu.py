"""GPU parity of the whole flow operator (encoders -> volume -> refinement -> weight head ->
upsampling -> TC epilogue) against the CPU oracle and the golden vectors of the imported
reference.  Tolerances: GPU fp32 vs oracle EPE mean <= 1e-3 px, max <= 1e-2 px,
sigmoid(weights) <= 1e-4."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import raft_ref  # noqa: E402  (checker only)
from woft_amd import synth  # noqa: E402


def _t(a):
    return torch.from_numpy(a[:, :, ::-1].copy()).permute(2, 0, 1).float()[None]


def _flow_config(sd, iters, raft_type="weighted", padding_mode="nopad", small=False, precision="fp32"):
    from woft_amd.config import Config
    from woft_amd.flow_provider import RAFTWrapper
    c = Config()
    c.of_class = RAFTWrapper
    c.raft_type = raft_type
    c.class_params = Config()
    c.class_params.small = small
    c.class_params.mixed_precision = False
    c.class_params.alternate_corr = False
    c.class_params.weight_head_structure = [(128, 3)] * 3
    c.model = sd
    c.iters = iters
    c.padding_mode = padding_mode
    if precision:
        c.precision = precision
    return c


def _epe(a, b):
    d = a.detach().cpu().float() - b.detach().cpu().float()
    e = torch.sqrt((d ** 2).sum(dim=-3))
    return float(e.mean()), float(e.max())


@torch.no_grad()
def test_stages_against_oracle(golden_dir):
    """Stage-by-stage comparison on the 128x160 golden pair (first failing stage is the culprit)."""
    g = np.load(golden_dir / "flow_full_128x160_it4.npz")
    sd = synth.make_state_dict(seed=int(g["seed"]))
    tr = {}
    ref = raft_ref.raft_forward(sd, _t(g["img1"]), _t(g["img2"]), 4, trace=tr)
    from woft_amd.engine import RaftEngine
    eng = RaftEngine(sd)
    plan = eng.plan(128, 160)
    plan.load_image(0, torch.from_numpy(g["img1"]).cuda(), 0, 0)
    plan.load_image(1, torch.from_numpy(g["img2"]).cuda(), 0, 0)
    plan.encode_source()
    snaps = []

    def trace(p, it):
        snaps.append(dict(lookup=p.corr.nchw().cpu(), net=p.hB.nchw().cpu(),
                          coords=p.coords.cpu().reshape(p.hf, p.wf, 2).permute(2, 0, 1).clone()))
    fu, dst, wo = (torch.zeros(2, 128, 160, device="cuda"), torch.zeros(2, 128 * 160, device="cuda"),
                   torch.zeros(1, 128 * 160, device="cuda"))
    plan.flow(4, (0, 0), 128, 160, flow_up=fu, dst=dst, wout=wo, do_sigmoid=False, trace=trace)
    torch.cuda.synchronize()

    def close(a, b, tol, what):
        err = float((a.cpu() - b).abs().max())
        assert err <= tol, f"{what}: max err {err:.3e} > {tol:.1e} (ref max {float(b.abs().max()):.2e})"

    close(plan.f1.nchw(), tr["fmap1"], 2e-4, "fmap1")
    close(plan.f2act[0].nchw(), tr["fmap2"], 2e-4, "fmap2")
    close(torch.from_numpy(g["fmap1"]), tr["fmap1"], 1e-4, "oracle vs golden fmap1")
    close(plan.net0.nchw(), tr["net0"], 5e-5, "net0")
    close(plan.xbuf.nchw()[:, :128], tr["inp"], 5e-5, "inp")
    for l in range(4):
        hl, wl = plan.dims[l]
        from woft_amd import ops
        close(ops.untile_planes(plan.vol[l], hl, wl), tr["pyr"][l][:, 0], 3e-4, f"volume level {l}")
    close(snaps[0]["lookup"], tr["lookups"][0], 5e-4, "lookup 0")
    close(snaps[0]["net"], tr["nets"][0], 2e-4, "net after iter 0")
    for it in range(4):
        close(snaps[it]["coords"], tr["coords"][it][0], 2e-3, f"coords after iter {it}")
    m, mx = _epe(fu, ref["flow_up"][0])
    assert m < 1e-3 and mx < 1e-2, ("flow_up vs oracle", m, mx)
    m, mx = _epe(fu, torch.from_numpy(g["flow_up"])[0])
    assert m < 1e-3 and mx < 1e-2, ("flow_up vs golden", m, mx)
    close(plan.wlow.reshape(1, 1, plan.hf, plan.wf), ref["weights_low"], 2e-3, "weight logits (1/8 res)")
    close(wo.reshape(1, 1, 128, 160), ref["weights_up"], 2e-3, "weight logits (full res)")


@torch.no_grad()
@pytest.mark.parametrize("name,small,raft_type", [
    ("flow_full_136x200_it12", False, "weighted"),
    ("flow_small_128x160_it4", True, "orig"),          # config-1 family: plain RAFT-small (raft.py:49-56)
    ("flow_wsmall_128x160_it4", True, "weighted"),
])
def test_operator_vs_golden(golden_dir, name, small, raft_type):
    g = np.load(golden_dir / f"{name}.npz")
    weighted = raft_type == "weighted"
    sd = synth.make_state_dict(seed=int(g["seed"]), small=small, weighted=weighted)
    fc = _flow_config(sd, int(g["iters"]), raft_type=raft_type, small=small)
    flower = fc.of_class(fc)
    flow, w = flower.compute_flow(g["img1"], g["img2"], mode="flow", do_sigmoid=False)
    torch.cuda.synchronize()
    H, W = g["img1"].shape[:2]
    assert tuple(flow.shape) == (2, H, W)
    m, mx = _epe(flow, torch.from_numpy(g["flow_up"])[0])
    assert m < 1e-3 and mx < 1e-2, (m, mx)
    if weighted:
        assert tuple(w.shape) == (1, H, W)
        assert float((torch.sigmoid(w.cpu()) - torch.sigmoid(torch.from_numpy(g["w_up"])[0])).abs().max()) < 1e-4
    else:
        assert w is None
        src, dst, ww = flower.compute_flow(g["img1"], g["img2"], mode="TC")
        assert ww is None and tuple(dst.shape) == (2, H * W)


@torch.no_grad()
def test_operator_tc_boundary(golden_dir):
    """RAFTWrapper.compute_flow contract: types, shapes, TC mode, replicate padding, pinned source."""
    g = np.load(golden_dir / "wrapper_tc_128x160_it4.npz")
    sd = synth.make_state_dict(seed=int(g["seed"]))
    fc = _flow_config(sd, int(g["iters"]))
    flower = fc.of_class(fc)
    flower.pin_source(g["img1"])
    for rep in range(2):                                   # second call hits the template cache
        src, dst, w = flower.compute_flow(g["img1"], g["img2"], mode="TC", do_sigmoid=True)
        torch.cuda.synchronize()
        assert src.dtype == torch.int64 and tuple(src.shape) == (2, 128 * 160) and src.is_cuda
        assert dst.dtype == torch.float32 and tuple(dst.shape) == (2, 128 * 160)
        assert tuple(w.shape) == (1, 128 * 160)
        assert np.array_equal(src.cpu().numpy(), g["src"])
        assert np.abs(dst.cpu().numpy() - g["dst"]).max() < 1e-2
        assert np.abs(w.cpu().numpy() - g["w"]).max() < 1e-4
    assert flower.last_flow_shape == {"batch": 1, "delta": 2, "H": 128, "W": 160}
    with pytest.raises(AssertionError):
        flower.compute_flow(g["img1"], g["img2"][:-8], mode="TC")
    with pytest.raises(AssertionError):
        flower.compute_flow(g["img1"], g["img2"], mode="bogus")
    with pytest.raises(AssertionError):                    # nopad insists on multiples of 8 (raft.py:223-226)
        flower.compute_flow(g["img1"][:125], g["img2"][:125], mode="TC")
    fc2 = _flow_config(sd, int(g["iters"]), padding_mode="RAFT")
    fl2 = fc2.of_class(fc2)
    a2, b2 = g["img1"][:125, :157].copy(), g["img2"][:125, :157].copy()
    s2, d2, w2 = fl2.compute_flow(a2, b2, mode="TC", do_sigmoid=True)
    torch.cuda.synchronize()
    assert np.array_equal(s2.cpu().numpy(), g["src_pad"])
    assert np.abs(d2.cpu().numpy() - g["dst_pad"]).max() < 1e-2
    assert np.abs(w2.cpu().numpy() - g["w_pad"]).max() < 1e-4
    fc3 = _flow_config(sd, int(g["iters"]), padding_mode="crop")
    fl3 = fc3.of_class(fc3)
    s4, d4, w4 = fl3.compute_flow(g["img1"][:, :157].copy(), g["img2"][:, :157].copy(), mode="TC", do_sigmoid=True)
    torch.cuda.synchronize()
    assert np.array_equal(s4.cpu().numpy(), g["src_crop"])
    assert np.abs(d4.cpu().numpy() - g["dst_crop"]).max() < 1e-2
    assert np.abs(w4.cpu().numpy() - g["w_crop"]).max() < 1e-4
    with pytest.raises(NotImplementedError):
        fc4 = _flow_config(sd, 2, padding_mode="Michal")
        fc4.of_class(fc4).compute_flow(g["img1"], g["img2"], mode="TC")


@torch.no_grad()
@pytest.mark.parametrize("precision,epe_mean,epe_max,wtol", [("bf16x3", 1e-3, 1e-2, 1e-4), ("bf16", 5e-2, 0.5, 5e-3),
                                                               ("fp16", 1e-2, 0.1, 1e-3)])
def test_reduced_precision_operating_points(golden_dir, precision, epe_mean, epe_max, wtol):
    """Split-bf16 (fp32-emulating), plain bf16 and fp16 MFMA paths against the reference's golden flow.
    Stated budgets: bf16x3 keeps the fp32 tolerances; bf16 EPE mean <= 0.05 px; fp16 -- the reference's
    `mixed_precision` scoping: fp16 convolutions in the encoders and the update block, fp32-class correlation, weight head
    and upsampling (weighted_raft.py:204-219,233-234,258-290) -- EPE mean <= 0.01 px, max <= 0.1 px, sigmoid(w) <= 1e-3."""
    g = np.load(golden_dir / "flow_full_136x200_it12.npz")
    sd = synth.make_state_dict(seed=int(g["seed"]))
    fc = _flow_config(sd, int(g["iters"]), precision=precision)
    flower = fc.of_class(fc)
    flow, w = flower.compute_flow(g["img1"], g["img2"], mode="flow", do_sigmoid=True)
    torch.cuda.synchronize()
    m, mx = _epe(flow, torch.from_numpy(g["flow_up"])[0])
    print(f"{precision}: EPE mean {m:.2e} max {mx:.2e}")
    assert m < epe_mean and mx < epe_max, (m, mx)
    assert float((w.cpu() - torch.sigmoid(torch.from_numpy(g["w_up"])[0])).abs().max()) < wtol


@torch.no_grad()
@pytest.mark.parametrize("precision,corr,epe_mean,epe_max,wtol", [
    ("fp32", "volume", 1e-3, 1e-2, 1e-4), ("fp32", "otf", 1e-3, 1e-2, 1e-4), ("bf16x3", "otf", 1e-3, 1e-2, 1e-4),
    ("bf16x3", "volume", 1e-3, 1e-2, 1e-4),
    ("bf16", "otf", 0.15, 1.0, 5e-3), ("bf16", "volume", 0.15, 1.0, 5e-3), ("fp16", "otf", 0.03, 0.3, 2e-3),
    ("fp16", "volume", 0.03, 0.3, 2e-3)])
def test_32_iterations_vs_reference_golden(golden_dir, precision, corr, epe_mean, epe_max, wtol):
    """BASELINE config 3 (32 refinement iterations; bf16 operating point) against the REFERENCE's flow at 32
    iterations (weighted_raft.py:228-237 run 32 times; tests/golden/flow_full_136x200_it32.npz).  Budgets of SURVEY
    8d: fp32-class arithmetic (fp32, bf16x3) EPE mean <= 1e-3 px / max <= 1e-2 px; bf16 EPE mean <= 0.15 px @ 32 it,
    sigmoid(w) <= 5e-3; fp16 (mixed_precision scoping) EPE mean <= 0.03 px @ 32 it, sigmoid(w) <= 2e-3."""
    g = np.load(golden_dir / "flow_full_136x200_it32.npz")
    assert int(g["iters"]) == 32
    sd = synth.make_state_dict(seed=int(g["seed"]))
    fc = _flow_config(sd, 32, precision=precision)
    fc.corr = corr
    flower = fc.of_class(fc)
    assert flower.engine.corr == corr and flower.engine.precision == precision
    flow, w = flower.compute_flow(g["img1"], g["img2"], mode="flow", do_sigmoid=True)
    torch.cuda.synchronize()
    m, mx = _epe(flow, torch.from_numpy(g["flow_up"])[0])
    print(f"32 it, {precision}/{corr}: EPE mean {m:.2e} max {mx:.2e} (mean |flow| "
          f"{float(np.sqrt((g['flow_up'] ** 2).sum(1)).mean()):.2f} px)")
    assert m < epe_mean and mx < epe_max, (m, mx)
    assert float((w.cpu() - torch.sigmoid(torch.from_numpy(g["w_up"])[0])).abs().max()) < wtol


@torch.no_grad()
def test_cached_flow_wire_format(tmp_path):
    """Pre-computed flow (utils/caching.py:53-59: '<i>-<i+1>.npz' with fp16 'half_flow' / 'half_weights') goes
    through the same TC / sigmoid post-processing as a computed one (raft.py:92-109,152-195); a missing file falls
    back to computing the flow."""
    h, w = 128, 160
    rng = np.random.RandomState(5)
    flow = (rng.randn(2, h, w) * 3).astype(np.float16)
    wts = rng.randn(1, h, w).astype(np.float16)
    d = tmp_path / "ds" / "seq"
    d.mkdir(parents=True)
    np.savez(d / "7-8.npz", half_flow=flow, half_weights=wts)
    sd = synth.make_state_dict(seed=3)
    c = _flow_config(sd, 2)
    c.flow_cache_dir = tmp_path
    prov = c.of_class(c)
    img = synth.make_template(h, w, seq_id=1)
    src, dst, wout = prov.compute_flow(img, img, mode="TC", src_img_identifier=("ds", "seq", 7), do_sigmoid=True,
                                       numpy_out=True)
    ys, xs = np.mgrid[0:h, 0:w]
    grid = np.stack([xs.ravel(), ys.ravel()]).astype(np.float32)
    assert np.array_equal(src, grid.astype(src.dtype))
    assert np.array_equal(dst, grid + flow.astype(np.float32).reshape(2, -1))
    np.testing.assert_allclose(wout, 1 / (1 + np.exp(-wts.astype(np.float32).reshape(1, -1))), rtol=0, atol=1e-6)
    fl, wl = prov.compute_flow(img, img, mode="flow", src_img_identifier=("ds", "seq", 7), numpy_out=True)
    assert np.array_equal(fl, flow.astype(np.float32)) and np.array_equal(wl, wts.astype(np.float32))
    # no file for this pair: the flow is computed, exactly as without an identifier
    _, dst2, w2 = prov.compute_flow(img, img, mode="TC", src_img_identifier=("ds", "seq", 9), numpy_out=True)
    _, dst3, w3 = prov.compute_flow(img, img, mode="TC", numpy_out=True)
    assert dst2.shape == (2, h * w) and np.array_equal(dst2, dst3) and np.array_equal(w2, w3)
    # unreadable caches fall back the same way (the reference computes the flow on ANY exception, raft.py:108-109):
    # a truncated archive, an object placeholder instead of the weight array, a missing array
    blob = (d / "7-8.npz").read_bytes()
    (d / "10-11.npz").write_bytes(blob[:len(blob) // 2])
    np.savez(d / "11-12.npz", half_flow=flow, half_weights=np.array(None, dtype=object))
    np.savez(d / "12-13.npz", half_flow=flow)
    (d / "13-14.npz").write_bytes(b"")
    for i in (10, 11, 12, 13):
        _, dst4, w4 = prov.compute_flow(img, img, mode="TC", src_img_identifier=("ds", "seq", i), numpy_out=True)
        assert np.array_equal(dst4, dst3) and np.array_equal(w4, w3), i


@torch.no_grad()
def test_mixed_precision_key_selects_the_fp16_operating_point():
    """class_params.mixed_precision = True (weighted_raft.py:204,215,233: autocast around fnet, cnet, update block) ->
    fp16 convolutions there, fp32-class correlation and weight head; `precision` / WOFT_PRECISION override it."""
    sd = synth.make_state_dict(seed=3)
    c = _flow_config(sd, 2, precision=None)
    c.class_params.mixed_precision = True
    prov = c.of_class(c)
    e = prov.engine
    assert (prov.precision, e.precision, e.prec_corr, e.prec_wh, e.corr) == ("fp16", "fp16", "bf16x3", "bf16x3", "otf")
    plan = e.plan(128, 160)
    convs = [ent[1] for ent in plan.prog_iter if ent[0] == "conv"] + [p for ent in plan.prog_iter if ent[0] == "conv2" for p in ent[1]]
    assert convs and all(p.precision == 3 for p in convs) and all(p.precision == 3 for p in plan.prog_mask)
    assert all(p.precision == 1 for p in plan.prog_wh) and plan.lookup.terms == 3
    c2 = _flow_config(sd, 2, precision="bf16x3")
    c2.class_params.mixed_precision = True
    assert c2.of_class(c2).precision == "bf16x3"
    # no key at all (an unmodified reference flow config): the built-in default, fp32-emulating bf16x3 (flow_provider.py: why)
    c3 = _flow_config(sd, 2, precision=None)
    p3 = c3.of_class(c3)
    assert p3.precision == "bf16x3" and "built-in default" in p3.precision_source


@torch.no_grad()
def test_weights_postprocessing_fn_and_backbone_model(tmp_path):
    """Flow config keys the reference reads: `weights_postprocessing_fn` -- a callable on the (1, 1, H, W)
    weight LOGITS, before the sigmoid (raft.py:152-159), for computed and for cached flows -- and `backbone_model`
    (raft.py:58-62: the fnet / cnet / update_block tensors of `model` are dropped; here they come from that checkpoint)."""
    h, w = 128, 160
    sd = synth.make_state_dict(seed=3)
    a = synth.make_template(h, w, seq_id=2)
    b = synth.make_frame(a, 2)
    base = _flow_config(sd, 2, precision="bf16x3")
    _, dst0, logit0 = base.of_class(base).compute_flow(a, b, mode="TC")
    seen = {}

    def post(wmap):
        seen["shape"] = tuple(wmap.shape)
        return 2.0 * torch.nn.functional.avg_pool2d(wmap, 3, stride=1, padding=1)     # a spatial map -> map function
    c = _flow_config(sd, 2, precision="bf16x3")
    c.weights_postprocessing_fn = post
    prov = c.of_class(c)
    prov.pin_source(a)
    prov.pin_weight_region(np.ones((h, w), bool))
    for kw in ({}, {"weight_region": True, "defer_weights": 10}):         # (the callable may read any pixel: full map)
        _, dst1, w1 = prov.compute_flow(a, b, mode="TC", do_sigmoid=True, **kw)
        assert seen["shape"] == (1, 1, h, w) and not prov.weights_deferred
        want = torch.sigmoid(post(logit0.reshape(1, 1, h, w))).reshape(1, -1)
        assert torch.equal(dst1, dst0) and torch.allclose(w1, want, rtol=0, atol=1e-6)
    fl, wl = prov.compute_flow(a, b, mode="flow")
    assert torch.allclose(wl, post(logit0.reshape(1, 1, h, w)).reshape(1, h, w), rtol=0, atol=1e-6)
    # cached flow: the same callable on the stored logits
    rng = np.random.RandomState(1)
    d = tmp_path / "ds" / "s"
    d.mkdir(parents=True)
    wts = rng.randn(1, h, w).astype(np.float32)
    np.savez(d / "0-1.npz", half_flow=rng.randn(2, h, w).astype(np.float32), half_weights=wts)
    c.flow_cache_dir = tmp_path
    _, _, wc = prov.compute_flow(a, b, mode="TC", src_img_identifier=("ds", "s", 0), do_sigmoid=True)
    want = torch.sigmoid(post(torch.from_numpy(wts)[None].cuda())).reshape(1, -1)
    assert torch.allclose(wc, want, rtol=0, atol=1e-6)
    # backbone_model: weight head of `model`, backbone of the second checkpoint
    other = synth.make_state_dict(seed=9)
    path = tmp_path / "backbone.pth"
    torch.save({"module." + k: v for k, v in other.items()}, path)
    merged = {k: (other[k] if any(t in k for t in ("fnet", "cnet", "update_block")) else v) for k, v in sd.items()}
    assert any(not torch.equal(merged[k], sd[k]) for k in sd) and any(torch.equal(merged[k], sd[k]) for k in sd)
    ref = _flow_config(merged, 2, precision="bf16x3")
    _, dst_ref, w_ref = ref.of_class(ref).compute_flow(a, b, mode="TC")
    for src in (path, other):
        c2 = _flow_config(sd, 2, precision="bf16x3")
        c2.backbone_model = src
        _, dst2, w2 = c2.of_class(c2).compute_flow(a, b, mode="TC")
        assert torch.equal(dst2, dst_ref) and torch.equal(w2, w_ref)


@torch.no_grad()
@pytest.mark.parametrize("precision,small", [("bf16x3", False), ("bf16", False), ("bf16x3", True), ("fp32", False), ("fp32", True)])
def test_volume_free_correlation_matches_volume(precision, small):
    """corr='otf' (the lookup computed from the feature maps, no P x P volume; what the reference's alternate_corr
    selects, corr.py:72-100) gives the same flow and weights as the volume path of the same precision."""
    sd = synth.make_state_dict(seed=11, small=small, weighted=not small)
    rt = "orig" if small else "weighted"
    h, w = 136, 200
    a, b = synth.make_template(h, w, seq_id=2), synth.make_template(h, w, seq_id=3)
    outs = {}
    for corr in ("volume", "otf"):
        c = _flow_config(sd, 5, raft_type=rt, padding_mode="RAFT", small=small, precision=precision)
        c.corr = corr
        c.volume_storage = "fp32"       # (plain bf16 stores its volume in bf16 by default: one more rounding -- below)
        prov = c.of_class(c)
        assert prov.engine.corr == corr
        fl, wt = prov.compute_flow(a, b, mode="flow", numpy_out=True)
        outs[corr] = (fl, wt)
    # identical correlation values and identical interpolation arithmetic: the whole flow is bit-identical
    assert np.array_equal(outs["otf"][0], outs["volume"][0])
    if not small:
        assert np.array_equal(outs["otf"][1], outs["volume"][1])
    # it is the default in every precision (exact fp32 since round 3: fp32-MFMA instantiation of the volume-free lookup)
    c = _flow_config(sd, 5, raft_type=rt, padding_mode="RAFT", small=small, precision=precision)
    assert c.of_class(c).engine.corr == "otf"
    if precision == "bf16":
        # the bf16-storage volume (default of precision "bf16" + corr "volume"): correlation values rounded to bf16 once
        # more -- a different, still bf16-class flow (budget vs the reference: test_32_iterations_vs_reference_golden)
        c = _flow_config(sd, 5, raft_type=rt, padding_mode="RAFT", small=small, precision=precision)
        c.corr = "volume"
        prov = c.of_class(c)
        fl, _ = prov.compute_flow(a, b, mode="flow", numpy_out=True)
        assert prov.engine.volume_storage == "bf16"
        assert all(v.dtype == torch.bfloat16 for pl in prov.engine._plans.values() for v in pl.vol)
        d = np.sqrt(((fl - outs["volume"][0]) ** 2).sum(0))
        print(f"bf16-storage volume vs fp32-storage volume: EPE mean {d.mean():.2e} max {d.max():.2e}")
        assert 0 < d.mean() < 0.05
    c = _flow_config(sd, 5, raft_type=rt, padding_mode="RAFT", small=small, precision="fp32")
    assert c.of_class(c).engine.corr == "otf"


@torch.no_grad()
@pytest.mark.parametrize("precision,small", [("bf16x3", False), ("fp32", False), ("bf16x3", True)])
def test_graph_replay_matches_eager(precision, small):
    """Flow config key `graph`: the launch list of a flow captured once into a hipGraph (at the second call of a
    shape) and replayed afterwards -- same kernels, same buffers: bit-identical outputs, call after call, also with
    a pinned source image and a weight region."""
    sd = synth.make_state_dict(seed=12, small=small, weighted=not small)
    rt = "orig" if small else "weighted"
    h, w = 136, 200
    a = synth.make_template(h, w, seq_id=4)
    frames = [synth.make_frame(a, t) for t in (1, 2, 3, 4, 5)]
    outs = {}
    for graph in (False, True):
        c = _flow_config(sd, 4, raft_type=rt, padding_mode="RAFT", small=small, precision=precision)
        c.graph = graph
        prov = c.of_class(c)
        assert prov.use_graph == graph
        prov.pin_source(a)
        res = []
        for k, f in enumerate(frames):
            if k == 3 and not small:                     # a different launch list from here on: new capture
                m = np.zeros((h, w), bool)
                m[40:100, 60:150] = True
                prov.pin_weight_region(m)
            src, dst, wt = prov.compute_flow(a, f, mode="TC", do_sigmoid=True, weight_region=True)
            res.append((dst.cpu().numpy().copy(), None if wt is None else wt.cpu().numpy().copy()))
        outs[graph] = res
        if graph:
            plan = next(iter(prov.engine._plans.values()))
            assert any(g is not None for g in plan._graphs.values())       # something WAS replayed
    for (d0, w0), (d1, w1) in zip(outs[False], outs[True]):
        assert np.array_equal(d0, d1)
        if w0 is not None:
            assert np.array_equal(w0, w1)


@torch.no_grad()
@pytest.mark.parametrize("precision,epe_mean,epe_max,wtol", [("fp32", 1e-3, 1e-2, 2e-4), ("bf16x3", 1e-3, 1e-2, 2e-4),
                                                               ("bf16", 5e-2, 0.5, 1e-2), ("fp16", 1e-2, 0.1, 2e-3)])
@pytest.mark.parametrize("name", ["constant", "constant_vs_texture", "saturated", "identical"])
def test_degenerate_inputs_vs_reference(golden_dir, name, precision, epe_mean, epe_max, wtol):
    """Constant image (InstanceNorm variance 0, extractor.py:28-32: rstd = 1/sqrt(eps) = 316 on a channel that holds
    nothing but rounding noise), half-saturated frames with a black bar, identical frames -- against the REFERENCE's own
    outputs on these inputs (tests/golden/degenerate_128x160_it4.npz), in all three arithmetic modes."""
    g = np.load(golden_dir / "degenerate_128x160_it4.npz")
    sd = synth.make_state_dict(seed=int(g["seed"]))
    fc = _flow_config(sd, int(g["iters"]), precision=precision)
    flower = fc.of_class(fc)
    flow, w = flower.compute_flow(g[f"{name}_img1"], g[f"{name}_img2"], mode="flow", do_sigmoid=False)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(flow).all()) and bool(torch.isfinite(w).all())
    m, mx = _epe(flow, torch.from_numpy(g[f"{name}_flow_up"])[0])
    dw = float((w.cpu() - torch.from_numpy(g[f"{name}_w_up"])[0]).abs().max())
    print(f"{name} / {precision}: EPE mean {m:.2e} max {mx:.2e}, weight logits {dw:.2e}")
    assert m < epe_mean and mx < epe_max, (m, mx)
    assert dw < wtol


@torch.no_grad()
@pytest.mark.parametrize("precision,corr,epe_mean,epe_max,wtol", [("bf16x3", "otf", 1e-3, 1e-2, 1e-4), ("fp32", "otf", 1e-3, 1e-2, 1e-4),
                                                                    ("bf16", "otf", 5e-2, 0.5, 5e-3), ("fp16", "otf", 1e-2, 0.1, 1e-3),
                                                                    ("f16mx8", "otf", 1e-3, 1e-2, 1e-4)])
def test_real_frames_720p_vs_reference(golden_dir, precision, corr, epe_mean, epe_max, wtol):
    """BASELINE config 2 at its REAL size on REAL frames: a 720 x 1280 pair of the reference's demo sequence (decoded
    frames stored in tests/golden/real_720p.npz), 12 iterations, against the reference's flow and weight logits -- 1/8
    resolution in full, full resolution on the stored stride-4 lattice and through the per-row means."""
    g = np.load(golden_dir / "real_720p.npz")
    sd = synth.make_state_dict(seed=int(g["seed"]))
    fc = _flow_config(sd, int(g["iters"]), precision=precision)
    fc.corr = corr
    flower = fc.of_class(fc)
    flow, w = flower.compute_flow(g["frame1"], g["frame3"], mode="flow", do_sigmoid=False)
    torch.cuda.synchronize()
    s = int(g["stride"])
    assert tuple(flow.shape) == (2, 720, 1280)
    m, mx = _epe(flow[:, ::s, ::s], torch.from_numpy(g["flow_up_s4"])[0])
    dw = float((w[:, ::s, ::s].cpu() - torch.from_numpy(g["w_up_s4"])[0]).abs().max())
    rm = float((flow.double().mean(-1).cpu() - torch.from_numpy(g["flow_up_rowmean"])[0]).abs().max())
    dws = float((torch.sigmoid(w[:, ::s, ::s].cpu()) - torch.sigmoid(torch.from_numpy(g["w_up_s4"])[0])).abs().max())
    print(f"720p real frames, {precision}/{corr}: EPE mean {m:.2e} max {mx:.2e}; row means {rm:.2e}; weight logits {dw:.2e} "
          f"(mean |flow| {float(np.sqrt((g['flow_up_s4'] ** 2).sum(1)).mean()):.2f} px)")
    assert m < epe_mean and mx < epe_max, (m, mx)
    assert rm < epe_max and dws < wtol            # (wtol: on the sigmoid, the quantity SURVEY 8d states the budget for)


@torch.no_grad()
@pytest.mark.parametrize("small", [False, True])
def test_launch_merging_switches_are_bit_identical(monkeypatch, small):
    """Round-3 launch merging -- the motion encoder's branches in shared launches (woft_conv2d_pair), the last iteration's
    flow-head and mask-head convs in one launch, the flow-head gather of iteration k inside the lookup launch of iteration
    k + 1 -- changes which launch does the work, not the operations or their order: flows and weights are bit-identical with
    every switch off."""
    from woft_amd import engine
    sd = synth.make_state_dict(seed=21, small=small, weighted=not small)
    rt = "orig" if small else "weighted"
    a = synth.make_template(136, 200, seq_id=6)
    b = synth.make_frame(a, 3)
    outs = []
    for pair, fold in ((True, True), (False, True), (True, False), (False, False)):
        monkeypatch.setattr(engine, "PAIR_BRANCHES", pair)
        monkeypatch.setattr(engine, "FOLD_GATHER", fold)
        c = _flow_config(sd, 5, raft_type=rt, padding_mode="nopad", small=small, precision="bf16x3")
        prov = c.of_class(c)
        flow, w = prov.compute_flow(a, b, mode="flow")
        plan = prov.engine.plan(136, 200)
        assert (plan._fold is not None) == (fold and plan.prog_iter[-1][0] == "fh_gather")
        n_launch = len(plan._fold[id(plan.prog_iter)]) if plan._fold is not None else len(plan.prog_iter)
        outs.append((flow.clone(), None if w is None else w.clone(), n_launch))
    for f, w, _ in outs[1:]:
        assert torch.equal(f, outs[0][0]) and (w is None or torch.equal(w, outs[0][1]))
    if not small:
        # launches per refinement iteration: everything merged / one per layer
        assert outs[0][2] == 9 and outs[3][2] == 12


def _crc(a):
    import zlib
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


@torch.no_grad()
@pytest.mark.parametrize("precision,epe_mean,epe_max,wtol", [("bf16x3", 1e-3, 1e-2, 1e-4), ("fp32", 1e-3, 1e-2, 1e-4),
                                                             ("bf16", 5e-2, 0.5, 5e-3), ("fp16", 1e-2, 0.1, 1e-3),
                                                             ("f16mx8", 1e-3, 1e-2, 1e-4)])
def test_metric_resolution_1080p_vs_reference(golden_dir, precision, epe_mean, epe_max, wtol):
    """The metric's own configuration (BASELINE.json: 1080 x 1920, WeightedRAFT-full, 12 iterations) against the REFERENCE's
    outputs on the same pair (tests/golden/metric_1080p_it12.npz, oracle/gen_golden.py: gen_metric -- weighted_raft.py:186-290
    run on the CPU of the build container): 1/8-resolution flow in full, full-resolution flow and weight logits on the stored
    stride-8 lattice and through the per-row means.  The pair is regenerated from its seeds; the stored CRC32s pin it."""
    g = np.load(golden_dir / "metric_1080p_it12.npz")
    H, W = int(g["H"]), int(g["W"])
    a = synth.make_template(H, W, seq_id=int(g["seq_id"]))
    b = synth.make_frame(a, int(g["t"]))
    assert _crc(a) == int(g["crc_img1"]) and _crc(b) == int(g["crc_img2"]), "synthetic pair drifted from the fixture's"
    sd = synth.make_state_dict(seed=int(g["seed"]))
    fc = _flow_config(sd, int(g["iters"]), precision=precision)
    flower = fc.of_class(fc)
    assert flower.engine.corr == "otf"
    flow, w = flower.compute_flow(a, b, mode="flow", do_sigmoid=False)
    torch.cuda.synchronize()
    s = int(g["stride"])
    assert tuple(flow.shape) == (2, H, W)
    m, mx = _epe(flow[:, ::s, ::s], torch.from_numpy(g["flow_up_s"])[0])
    rm = float((flow.double().mean(-1).cpu() - torch.from_numpy(g["flow_up_rowmean"])[0]).abs().max())
    dws = float((torch.sigmoid(w[:, ::s, ::s].cpu()) - torch.sigmoid(torch.from_numpy(g["w_up_s"])[0])).abs().max())
    wrm = float((w.double().mean(-1).cpu() - torch.from_numpy(g["w_up_rowmean"])[0]).abs().max())
    print(f"1080p metric pair, {precision}: EPE mean {m:.2e} max {mx:.2e}; row means {rm:.2e}; sigmoid(w) {dws:.2e}; "
          f"w row means {wrm:.2e} (mean |flow| {float(np.sqrt((g['flow_up_s'] ** 2).sum(1)).mean()):.2f} px)")
    assert m < epe_mean and mx < epe_max, (m, mx)
    assert rm < epe_max and dws < wtol and wrm < 50 * wtol


@torch.no_grad()
@pytest.mark.parametrize("precision,epe_mean,epe_max", [("bf16x3", 1e-3, 1e-2), ("fp32", 1e-3, 1e-2), ("bf16", 5e-2, 0.5),
                                                        ("fp16", 1e-2, 0.1)])
def test_config1_small_480x640_vs_reference(golden_dir, precision, epe_mean, epe_max):
    """BASELINE configs[0]: plain RAFT-small, 4 iterations, 480 x 640 (raft.py:169-262 -> raft_core/raft.py:95-167,
    extractor.py:244-267, update.py:71-77,23-31,106-112; `upflow8` bilinear x8) against the reference's outputs on the same
    pair (tests/golden/cfg0_small_480x640_it4.npz): flow_low in full, flow_up on the stride-4 lattice + row means."""
    from oracle.gen_golden import pair                      # (the fixture's input generator; checker side only)
    g = np.load(golden_dir / "cfg0_small_480x640_it4.npz")
    H, W = int(g["H"]), int(g["W"])
    a, b = pair(H, W, seed=int(g["pair_seed"]), shift=tuple(int(v) for v in g["shift"]))
    assert _crc(a) == int(g["crc_img1"]) and _crc(b) == int(g["crc_img2"]), "synthetic pair drifted from the fixture's"
    sd = synth.make_state_dict(seed=int(g["seed"]), small=True, weighted=False)
    fc = _flow_config(sd, int(g["iters"]), raft_type="orig", small=True, precision=precision)
    flower = fc.of_class(fc)
    flow, w = flower.compute_flow(a, b, mode="flow")
    torch.cuda.synchronize()
    assert w is None and tuple(flow.shape) == (2, H, W)
    s = int(g["stride"])
    m, mx = _epe(flow[:, ::s, ::s], torch.from_numpy(g["flow_up_s"])[0])
    rm = float((flow.double().mean(-1).cpu() - torch.from_numpy(g["flow_up_rowmean"])[0]).abs().max())
    print(f"480x640 RAFT-small 4 it, {precision}: EPE mean {m:.2e} max {mx:.2e}; row means {rm:.2e} "
          f"(mean |flow| {float(np.sqrt((g['flow_up_s'] ** 2).sum(1)).mean()):.2f} px)")
    assert m < epe_mean and mx < epe_max, (m, mx)
    assert rm < epe_max


@torch.no_grad()
@pytest.mark.parametrize("layers", ["all", "auto"])
def test_f16mx8_operating_point_vs_reference(golden_dir, monkeypatch, layers):
    """precision "f16mx8" (the update block's convolutions in two matrix-pipe passes per product: fp16 main term + two block-scaled
    fp8 cross terms; everything else bf16x3) against the REFERENCE's flow and weights at 12 and 32 iterations: inside the fp32
    budget of SURVEY 8d (EPE mean <= 1e-3 px, max <= 1e-2 px, sigmoid(w) <= 1e-4)."""
    from woft_amd import ops
    monkeypatch.setattr(ops, "MX_LAYERS", layers)            # every multi-tap layer / the default: the 3x3 layers with 256 input channels
    for name in ("flow_full_136x200_it12", "flow_full_136x200_it32"):
        g = np.load(golden_dir / f"{name}.npz")
        sd = synth.make_state_dict(seed=int(g["seed"]))
        fc = _flow_config(sd, int(g["iters"]), precision="f16mx8")
        flower = fc.of_class(fc)
        plan = flower.engine.plan(136, 200)
        n_mx = sum(q.precision == 4 for e in plan.prog_iter if e[0] in ("conv", "conv2") for q in (e[1] if e[0] == "conv2" else (e[1],)))
        assert n_mx >= (5 if layers == "all" else 3), f"{n_mx} update-block layers run in f16mx8"
        flow, w = flower.compute_flow(g["img1"], g["img2"], mode="flow", do_sigmoid=False)
        torch.cuda.synchronize()
        m, mx = _epe(flow, torch.from_numpy(g["flow_up"])[0])
        dws = float((torch.sigmoid(w.cpu()) - torch.sigmoid(torch.from_numpy(g["w_up"])[0])).abs().max())
        print(f"{name} f16mx8: EPE mean {m:.2e} max {mx:.2e}; sigmoid(w) {dws:.2e}")
        scale = 1.0 if int(g["iters"]) <= 12 else 3.0
        assert m < 1e-3 * scale and mx < 1e-2 * scale and dws < 1e-4 * scale, (m, mx, dws)


@torch.no_grad()
def test_fast_gate_functions_drift_is_bounded(monkeypatch, golden_dir):
    """The split-bf16 / fp16 precisions evaluate the GRU's sigmoid / tanh on the hardware exp2 / rcp (common.h: absolute error <= 3e-7 per
    value; tanh's RELATIVE error is unbounded near 0).  Over 32 refinement iterations the flow must stay within 1e-4 px of the flow
    with the library functions (WOFT_SLOW_GATES) -- an order below the fp32 budget the precision is held to."""
    from woft_amd import ops
    g = np.load(golden_dir / "flow_full_136x200_it32.npz")
    sd = synth.make_state_dict(seed=int(g["seed"]))
    flows = []
    for slow in (False, True):
        monkeypatch.setattr(ops, "SLOW_GATES", slow)
        fc = _flow_config(sd, 32, precision="bf16x3")
        fl = fc.of_class(fc)
        flow, _ = fl.compute_flow(g["img1"], g["img2"], mode="flow")
        flows.append(flow.clone())
    m, mx = _epe(flows[0], flows[1])
    print(f"fast vs library gate functions, 32 iterations: EPE mean {m:.2e} max {mx:.2e}")
    assert 0 < mx < 1e-4, (m, mx)


@torch.no_grad()
@pytest.mark.parametrize("precision", ["bf16x3", "fp32"])
def test_weight_heads_other_than_the_shipped_one(golden_dir, precision):
    """class_params.weight_head_structure other than [(128, 3)] * 3 (weighted_raft.py:318-345; round-4 review: a hard refusal):
    the engine follows the state-dict's layers -- here 3x3 / 5x5 / 7x7 / 1x1 kernels, 16 to 160 channels, plain-int entries --
    layer by layer; flow and weights against the REFERENCE built with each structure."""
    from oracle.gen_golden import HEAD_STRUCTURES
    g = np.load(golden_dir / "weight_heads_128x160_it3.npz")
    for name, st in HEAD_STRUCTURES.items():
        sd = synth.make_state_dict(seed=int(g["seed"]), weight_head_structure=st)
        fc = _flow_config(sd, int(g["iters"]), precision=precision)
        fc.class_params.weight_head_structure = st
        flower = fc.of_class(fc)
        assert not flower.engine.wh_std
        flow, w = flower.compute_flow(g["img1"], g["img2"], mode="flow", do_sigmoid=False)
        torch.cuda.synchronize()
        m, mx = _epe(flow, torch.from_numpy(g[f"{name}_flow_up"])[0])
        dw = float((torch.sigmoid(w.cpu()) - torch.sigmoid(torch.from_numpy(g[f"{name}_w_up"])[0])).abs().max())
        dl = float((w.cpu() - torch.from_numpy(g[f"{name}_w_up"])[0]).abs().max())
        print(f"{name} {precision}: EPE {m:.2e} / {mx:.2e}; sigmoid(w) {dw:.2e}; logits {dl:.2e}")
        assert m < 1e-3 and mx < 1e-2 and dw < 1e-4 and dl < 2e-3, (name, m, mx, dw, dl)
        # through the tracker-facing surface too: a pinned source with a weight region falls back to the full map
        flower.pin_source(g["img1"])
        flower.pin_weight_region(np.ones(g["img1"].shape[:2], dtype=bool))
        _, _, wt = flower.compute_flow(g["img1"], g["img2"], mode="TC", do_sigmoid=False, weight_region=True, defer_weights=500)
        assert wt is not None and float((wt.reshape(-1).cpu() - w.reshape(-1).cpu()).abs().max()) == 0.0
