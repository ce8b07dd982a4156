"""Generate tests/golden/* by running the REFERENCE itself (imported from /root/reference).

Run only in the build container (the GPU box has no /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.gen_golden

The fixtures are data (inputs + the reference's outputs); no reference source is stored.
cv2 / ipdb / kornia are absent from the image: cv2 and ipdb are stubbed with inert
modules, kornia with the three functions restated in oracle/hfit_ref.py (third-party,
parity unpinned -- see that file's header).  'cuda' device strings are mapped to 'cpu'.
"""
import json
import os
import sys
import types
from pathlib import Path
from types import SimpleNamespace
from unittest import mock

sys.dont_write_bytecode = True
import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import hfit_ref  # noqa: E402
from woft_amd import synth  # noqa: E402

GOLD = ROOT / "tests" / "golden"
REF = Path("/root/reference")


def install_stubs():
    sys.modules["cv2"] = mock.MagicMock(name="cv2")
    ipdb = types.ModuleType("ipdb")
    ipdb.iex = lambda f: f
    ipdb.post_mortem = lambda *a, **k: None
    sys.modules["ipdb"] = ipdb
    names = ["kornia", "kornia.geometry", "kornia.geometry.epipolar",
             "kornia.geometry.conversions", "kornia.geometry.homography"]
    mods = {n: types.ModuleType(n) for n in names}
    mods["kornia"].geometry = mods["kornia.geometry"]
    mods["kornia.geometry"].epipolar = mods["kornia.geometry.epipolar"]
    mods["kornia.geometry"].conversions = mods["kornia.geometry.conversions"]
    mods["kornia.geometry"].homography = mods["kornia.geometry.homography"]
    mods["kornia.geometry.epipolar"].normalize_points = hfit_ref.normalize_points
    mods["kornia.geometry.conversions"].convert_points_to_homogeneous = hfit_ref.to_homogeneous
    mods["kornia.geometry.conversions"].convert_points_from_homogeneous = hfit_ref.from_homogeneous
    sys.modules.update(mods)
    _t, _m = torch.Tensor.to, torch.nn.Module.to

    def fix(a):
        return tuple("cpu" if isinstance(x, str) and x.startswith("cuda") else x for x in a)

    def fixk(k):
        return {kk: ("cpu" if kk == "device" and isinstance(v, str) and v.startswith("cuda") else v)
                for kk, v in k.items()}
    torch.Tensor.to = lambda self, *a, **k: _t(self, *fix(a), **fixk(k))
    torch.nn.Module.to = lambda self, *a, **k: _m(self, *fix(a), **k)
    sys.path[:0] = [str(REF), str(REF / "pytracking/external/RAFT")]


class FakeCuda(torch.Tensor):
    is_cuda = property(lambda s: True)


def ref_args(small, weighted=True, weight_head_structure=None):
    return SimpleNamespace(small=small, mixed_precision=False, alternate_corr=False,
                           weight_head_structure=weight_head_structure or [(128, 3)] * 3, mask_estimation=False)


def pair(H, W, seed, shift=(3, -2)):
    """Two related uint8 BGR images: a smooth-ish texture and a shifted + perturbed copy."""
    t = synth.make_template(H + 16, W + 16, seq_id=seed)
    a = t[8:8 + H, 8:8 + W]
    b = t[8 + shift[1]:8 + shift[1] + H, 8 + shift[0]:8 + shift[0] + W]
    rs = np.random.RandomState(seed)
    b = np.clip(b.astype(np.int32) + rs.randint(-6, 7, b.shape), 0, 255).astype(np.uint8)
    return np.ascontiguousarray(a), np.ascontiguousarray(b)


def to_t(a):
    return torch.from_numpy(a[:, :, ::-1].copy()).permute(2, 0, 1).float()[None]


@torch.no_grad()
def gen_flow():
    from raft_core.weighted_raft import WeightedRAFT
    from raft_core.raft import RAFT
    from raft_core.corr import CorrBlock

    keys = {}
    # ---- full weighted model ------------------------------------------------------
    sd = synth.make_state_dict(seed=7, small=False, weighted=True)
    net = WeightedRAFT(ref_args(False)).eval()
    net.load_state_dict(sd, strict=True)
    keys["weighted_full"] = {k: list(v.shape) for k, v in net.state_dict().items()}
    keys["weighted_full_nparams"] = int(sum(p.numel() for p in net.parameters()))

    # case A: 128x160, 4 iters, with intermediates
    a, b = pair(128, 160, seed=11)
    i1, i2 = to_t(a), to_t(b)
    flow_low, flow_up, vol, w_low, w_up = net(i1, i2, iters=4, test_mode=True)
    n1 = 2 * (i1 / 255.0) - 1.0
    n2 = 2 * (i2 / 255.0) - 1.0
    f1, f2 = net.fnet([n1, n2])
    cb = CorrBlock(f1.float(), f2.float(), radius=4, num_levels=4)
    c = net.cnet(n1)
    h0, inp = torch.tanh(c[:, :128]), torch.relu(c[:, 128:])
    from raft_core.utils.utils import coords_grid
    coords0 = coords_grid(1, 16, 20, device="cpu")
    look0 = cb(coords0)
    net1, mask1, d1 = net.update_block(h0, inp, look0, coords0 - coords0)
    np.savez_compressed(
        GOLD / "flow_full_128x160_it4.npz", img1=a, img2=b, seed=7, iters=4,
        flow_low=flow_low.numpy(), flow_up=flow_up.numpy(), w_low=w_low.numpy(), w_up=w_up.numpy(),
        fmap1=f1.numpy(), fmap2=f2.numpy(), net0=h0.numpy(), inp=inp.numpy(),
        pyr0_rows=cb.corr_pyramid[0][:24, 0].numpy(), pyr1_rows=cb.corr_pyramid[1][:24, 0].numpy(),
        pyr2_rows=cb.corr_pyramid[2][:24, 0].numpy(), pyr3_rows=cb.corr_pyramid[3][:24, 0].numpy(),
        lookup0=look0.numpy(), net1=net1.numpy(), mask1=mask1.numpy(), delta1=d1.numpy())

    # case B: 136x200 (odd pyramid sizes 17x25 -> 8x12 -> 4x6 -> 2x3), 12 iters, outputs only
    a, b = pair(136, 200, seed=12, shift=(-4, 3))
    flow_low, flow_up, vol, w_low, w_up = net(to_t(a), to_t(b), iters=12, test_mode=True)
    np.savez_compressed(GOLD / "flow_full_136x200_it12.npz", img1=a, img2=b, seed=7, iters=12,
                        flow_low=flow_low.numpy(), flow_up=flow_up.numpy(),
                        w_low=w_low.numpy(), w_up=w_up.numpy())
    # case C: the same pair at 32 iterations (BASELINE config 3: the refinement loop of weighted_raft.py:228-237
    # run to the depth the bf16 EPE budget of SURVEY 8d is stated at); outputs only
    flow_low, flow_up, vol, w_low, w_up = net(to_t(a), to_t(b), iters=32, test_mode=True)
    np.savez_compressed(GOLD / "flow_full_136x200_it32.npz", img1=a, img2=b, seed=7, iters=32,
                        flow_low=flow_low.numpy(), flow_up=flow_up.numpy(),
                        w_low=w_low.numpy(), w_up=w_up.numpy())

    # ---- small plain RAFT (config 1 family) -----------------------------------------
    sds = synth.make_state_dict(seed=8, small=True, weighted=False)
    nets = RAFT(ref_args(True)).eval()
    nets.load_state_dict(sds, strict=True)
    keys["plain_small"] = {k: list(v.shape) for k, v in nets.state_dict().items()}
    keys["plain_small_nparams"] = int(sum(p.numel() for p in nets.parameters()))
    a, b = pair(128, 160, seed=13)
    fl, fu = nets(to_t(a), to_t(b), iters=4, test_mode=True)
    np.savez_compressed(GOLD / "flow_small_128x160_it4.npz", img1=a, img2=b, seed=8, iters=4,
                        flow_low=fl.numpy(), flow_up=fu.numpy())

    # ---- small weighted RAFT ---------------------------------------------------------
    sdw = synth.make_state_dict(seed=9, small=True, weighted=True)
    netw = WeightedRAFT(ref_args(True)).eval()
    netw.load_state_dict(sdw, strict=True)
    keys["weighted_small"] = {k: list(v.shape) for k, v in netw.state_dict().items()}
    fl, fu, _, wl, wu = netw(to_t(a), to_t(b), iters=4, test_mode=True)
    np.savez_compressed(GOLD / "flow_wsmall_128x160_it4.npz", img1=a, img2=b, seed=9, iters=4,
                        flow_low=fl.numpy(), flow_up=fu.numpy(), w_low=wl.numpy(), w_up=wu.numpy())

    (GOLD / "state_dict_keys.json").write_text(json.dumps(keys, indent=0))

    # ---- lookup-only cases on hand-made pyramids --------------------------------------
    rs = np.random.RandomState(5)
    h1, w1 = 6, 7
    P = h1 * w1
    H2, W2 = 17, 25
    ramp = (100.0 * np.arange(H2)[:, None] + np.arange(W2)[None, :]).astype(np.float32)
    vol0 = np.tile(ramp[None, None], (P, 1, 1, 1)) + rs.uniform(-1, 1, (P, 1, H2, W2)).astype(np.float32)
    vol0[0, 0] = ramp                                   # row 0: pure ramp pins the x-major window order
    cbk = CorrBlock.__new__(CorrBlock)
    cbk.num_levels, cbk.radius = 4, 4
    v = torch.from_numpy(vol0)
    cbk.corr_pyramid = [v]
    for _ in range(3):
        v = torch.nn.functional.avg_pool2d(v, 2, stride=2)
        cbk.corr_pyramid.append(v)
    coords = np.stack([rs.uniform(-6, W2 + 5, (h1, w1)), rs.uniform(-6, H2 + 5, (h1, w1))], 0).astype(np.float32)
    coords[:, 0, 0] = (8.0, 6.0)                       # on-grid probe
    coords[:, 0, 1] = (8.25, 6.5)
    coords[:, 0, 2] = (-3.5, -2.25)                    # negative / partly outside
    coords[:, 0, 3] = (W2 + 3.0, H2 + 2.0)             # fully beyond the border at level 0
    out = cbk(torch.from_numpy(coords)[None])
    np.savez_compressed(GOLD / "lookup_handmade.npz", vol0=vol0, coords=coords, out=out.numpy(), radius=4)


HEAD_STRUCTURES = {"mixed": [(32, 3), (48, 5), 24], "one_1x1": [(64, 1)], "wide_7": [(16, 7), (160, 3)]}


@torch.no_grad()
def gen_heads():
    """Weight heads other than the shipped [(128, 3)] * 3 (class_params.weight_head_structure, weighted_raft.py:318-345):
    the reference network built with each structure, 3 iterations at 128 x 160 -> flow and weight logits."""
    from raft_core.weighted_raft import WeightedRAFT
    a, b = pair(128, 160, seed=21, shift=(2, 3))
    out = dict(img1=a, img2=b, seed=17, iters=3, names=np.array(sorted(HEAD_STRUCTURES)))
    for name, st in HEAD_STRUCTURES.items():
        sd = synth.make_state_dict(seed=17, weight_head_structure=st)
        net = WeightedRAFT(ref_args(False, weight_head_structure=st)).eval()
        net.load_state_dict(sd, strict=True)
        flow_low, flow_up, vol, w_low, w_up = net(to_t(a), to_t(b), iters=3, test_mode=True)
        out[f"{name}_flow_up"], out[f"{name}_w_low"], out[f"{name}_w_up"] = flow_up.numpy(), w_low.numpy(), w_up.numpy()
    np.savez_compressed(GOLD / "weight_heads_128x160_it3.npz", **out)


@torch.no_grad()
def gen_wrapper():
    """Operator boundary: RAFTWrapper.compute_flow through the reference's own config loader."""
    from pytracking.utils.config import load_config
    import tempfile
    fc = load_config(REF / "pytracking/optical_flow/configs/v2_SNOB_large_g05_RAFT.py")
    fc.weights_postprocessing_fn = None
    sd = synth.make_state_dict(seed=7, small=False, weighted=True)
    with tempfile.TemporaryDirectory() as td:
        fc.model = os.path.join(td, "sd.pth")
        torch.save(sd, fc.model)
        fc.iters = 4
        flower = fc.of_class(fc)
        a, b = pair(128, 160, seed=11)
        src, dst, w = flower.compute_flow(a, b, mode="TC", do_sigmoid=True)
        fl, wf = flower.compute_flow(a, b, mode="flow", do_sigmoid=False)
        # RAFT replicate padding on a non-multiple-of-8 input
        fc.padding_mode = "RAFT"
        a2, b2 = a[:125, :157].copy(), b[:125, :157].copy()
        src2, dst2, w2 = flower.compute_flow(a2, b2, mode="TC", do_sigmoid=True)
        # ('Michal' mode cannot be pinned: the reference's MichalPadder.unpad(None) raises AttributeError for
        #  raft_type 'orig' / 'weighted' (raft.py:148-150,264-265) -- it is unreachable in every shipped config)
        fc.padding_mode = "crop"            # crop to a multiple of 8 from the right/bottom (raft.py:235-247)
        a4, b4 = a[:, :157].copy(), b[:, :157].copy()
        src4, dst4, w4 = flower.compute_flow(a4, b4, mode="TC", do_sigmoid=True)
    np.savez_compressed(GOLD / "wrapper_tc_128x160_it4.npz", img1=a, img2=b, seed=7, iters=4,
                        src=src.numpy(), dst=dst.numpy(), w=w.numpy(), flow=fl.numpy(), w_logit=wf.numpy(),
                        src_pad=src2.numpy(), dst_pad=dst2.numpy(), w_pad=w2.numpy(),
                        src_crop=src4.numpy(), dst_crop=dst4.numpy(), w_crop=w4.numpy())


@torch.no_grad()
def gen_hfit():
    import pytracking.utils.least_squares_H as L
    import pytracking.utils.geom_utils as G
    out = {}
    rs = np.random.RandomState(3)

    def make(N, outlier_frac=0.1, noise=0.3, degenerate=False):
        Hgt = np.array([[1.02, 0.03, 12.0], [-0.02, 0.98, -7.0], [2e-5, -1e-5, 1.0]])
        a = np.stack([rs.uniform(100, 1800, N), rs.uniform(80, 1000, N)], 1)
        if degenerate:
            a[:, 1] = 300 + 0.001 * a[:, 0] + rs.uniform(-0.5, 0.5, N)
        ah = np.concatenate([a, np.ones((N, 1))], 1) @ Hgt.T
        b = ah[:, :2] / ah[:, 2:] + rs.normal(0, noise, (N, 2))
        no = int(outlier_frac * N)
        if no:
            b[:no] += rs.uniform(-80, 80, (no, 2))
        w = rs.uniform(0.05, 1.0, N)
        w[:no] *= 0.2
        return (torch.from_numpy(a.astype(np.float32))[None], torch.from_numpy(b.astype(np.float32))[None],
                torch.from_numpy(w.astype(np.float32))[None])

    for name, N, kw in (("n4", 4, dict(outlier_frac=0.0, noise=0.0)), ("n500", 500, {}),
                        ("n4096", 4096, {}), ("degen", 300, dict(degenerate=True, outlier_frac=0.0))):
        a, b, w = make(N, **kw)
        out[f"{name}_a"], out[f"{name}_b"], out[f"{name}_w"] = a.numpy(), b.numpy(), w.numpy()
        out[f"{name}_qr_w"] = L.find_homography_nonhomogeneous_QR(a, b, w).numpy()
        out[f"{name}_qr_now"] = L.find_homography_nonhomogeneous_QR(a, b, None).numpy()
        ac = a.as_subclass(FakeCuda)
        out[f"{name}_irls_l1"] = torch.Tensor(L.find_homography_IRLSq_QR(ac, b, w)).numpy()
        out[f"{name}_irls_huber2"] = torch.Tensor(L.find_homography_IRLSq_QR(
            ac, b, w, reweighting_fn=lambda r: L.IRLSq_Huber(r, k=2))).numpy()
        out[f"{name}_irls_huber001"] = torch.Tensor(L.find_homography_IRLSq_QR(
            ac, b, w, reweighting_fn=lambda r: L.IRLSq_Huber(r, k=0.01))).numpy()
        Hq = torch.from_numpy(out[f"{name}_qr_w"])
        out[f"{name}_projerr"] = L.torch_proj_errors(Hq, a.permute(0, 2, 1), b.permute(0, 2, 1)).numpy()
    r = torch.from_numpy(np.linspace(-3, 3, 25).astype(np.float32))
    out["huber_in"] = r.numpy()
    out["huber_k1"] = L.IRLSq_Huber(r.clone(), k=1).numpy()
    out["l1"] = L.IRLSq_L1(r.clone()).numpy()
    H1 = np.array([[1, 0.1, 3], [0, 1.1, -2], [1e-4, 0, 1.0]])
    H2 = np.array([[0.9, 0, 1], [0.05, 1, 4], [0, 2e-4, 1.2]])
    out["compose_in1"], out["compose_in2"] = H1, H2
    out["compose_12"] = G.compose_H(H1, H2)
    out["compose_121"] = G.compose_H(H1, H2, H1)
    np.savez_compressed(GOLD / "hfit.npz", **out)

    # Sobol subsampler of the default config, run through the reference config module itself
    from pytracking.utils.config import load_config  # noqa: F401
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "woftcfg", REF / "pytracking/configs/YAOFT_single_control_repRAFT_sub500_noreliableinl_wLSq.py")
    cfg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cfg)
    sob = {}
    for N in (400, 501, 600, 2000, 518400):
        idx = torch.arange(N)[None].float()
        a, _, _ = cfg.subsampler(idx, idx, torch.ones(1, N))
        sob[f"n{N}"] = a[0].numpy().astype(np.int64)
    np.savez_compressed(GOLD / "sobol.npz", **sob)


def install_functional_cv2():
    """SURVEY 8c fixture (7): the reference TRACKER needs cv2.warpPerspective / resize / findContours to do real
    work.  OpenCV is absent from the image, so the stub below supplies them with the float-bilinear arithmetic of
    oracle/tracker_ref.py (which the HIP warp kernel implements): the fixture pins the reference's state machine,
    masking, subsampling, estimator plumbing and homography composition -- NOT OpenCV's fixed-point interpolation,
    which stays parity-unpinned."""
    from scipy import ndimage
    from oracle import tracker_ref as TR
    cv2 = sys.modules["cv2"]
    cv2.INTER_NEAREST, cv2.INTER_LINEAR = 0, 1
    cv2.RETR_EXTERNAL, cv2.CHAIN_APPROX_NONE = 0, 1

    def warpPerspective(src, M, dsize, flags=1, **kw):
        assert tuple(dsize) == (src.shape[1], src.shape[0]) and not kw
        if flags == cv2.INTER_NEAREST:
            return TR.warp_nearest(src, M)
        return TR.warp_linear_u8(src, M) if src.dtype == np.uint8 else TR.warp_linear(src, M)

    def resize(img, dsize, fx=None, fy=None, **kw):
        assert dsize is None and fx == fy and not kw
        return TR.resize_linear_u8(img, 1.0 / fx)

    def findContours(mask, mode, method):
        _, n = ndimage.label(mask > 0, structure=np.ones((3, 3), dtype=bool))     # external contours = 8-connected blobs
        ys, xs = np.nonzero(mask)
        return [np.stack([xs, ys], 1)[:, None, :].astype(np.int32)] * n, None      # (the tracker only counts them)

    cv2.warpPerspective, cv2.resize, cv2.findContours = warpPerspective, resize, findContours


@torch.no_grad()
def gen_tracker():
    """Runs of the reference's own YAOFTrackerSingleControl (tracker/YAOF_tracker_single_control.py:18-327) built
    from the reference's own config files, on short synthetic sequences: per frame H_cur2init and the meta fields."""
    import tempfile
    from pytracking.utils.config import load_config
    install_functional_cv2()
    sd = synth.make_state_dict(seed=7, small=False, weighted=True)
    H, W, iters = 128, 160, 4
    out = {}
    with tempfile.TemporaryDirectory() as td:
        model = os.path.join(td, "sd.pth")
        torch.save(sd, model)

        def run(name, cfg, seq_id, frame_ts, force_fail=(), mutate=None):
            conf = load_config(REF / "pytracking/configs" / cfg)
            conf.flow_config.model = model
            conf.flow_config.iters = iters
            if mutate:
                mutate(conf)
            state = {"i": 0}
            orig = conf.redet_success_fn

            def redet(*a):                       # config-level hook: the re-detection test fails on chosen frames
                ok = orig(*a)
                return ok if state["i"] not in force_fail else (ok & False)
            conf.redet_success_fn = redet
            trk = conf.tracker_class(conf)
            template = synth.make_template(H, W, seq_id=seq_id)
            mask = synth.make_init_mask(H, W)
            trk.init(template, mask)
            frames, Hs, meta = [], [], []
            for i, t in enumerate(frame_ts):
                state["i"] = i
                f = synth.make_frame(template, t)
                Hc, m = trk.track(f)
                frames.append(f)
                Hs.append(np.asarray(Hc, np.float64))
                loc = getattr(m, "H_local_cur2init", None)
                meta.append([float(bool(m.lost)), float(m.N_lost), float(bool(m.global_H_success)),
                             0.0 if loc is None else 1.0])
                out[f"{name}_Hglobal_{i}"] = np.asarray(m.H_global_cur2init, np.float64)
                out[f"{name}_lastgood_{i}"] = np.asarray(m.last_good_H2init, np.float64)
                if loc is not None:
                    out[f"{name}_Hlocal_{i}"] = np.asarray(loc, np.float64)
            out[f"{name}_template"], out[f"{name}_mask"] = template, mask
            out[f"{name}_frames"] = np.stack(frames)
            out[f"{name}_H"], out[f"{name}_meta"] = np.stack(Hs), np.asarray(meta)
            out[f"{name}_force_fail"] = np.asarray(sorted(force_fail), np.int64)

        # (a) default config, 5 frames; (b) the same with the re-detection test failing on frames 2 and 3: the
        #     lost / local-flow branch (TRK:171-207) and the recovery; (c) the IRLS config (ablation_08)
        run("woft", "WOFT.py", 3, [1, 2, 3, 4, 5])
        run("lost", "WOFT.py", 4, [1, 2, 3, 4, 5, 6], force_fail=(2, 3))
        def cuda_flag(conf):                     # the IRLS estimator insists on is_cuda (least_squares_H.py:292-293)
            est = conf.H_estimator
            conf.H_estimator = lambda a, b, w: torch.Tensor(est(a.as_subclass(FakeCuda), b, w))
        run("irls", "ablation_08.py", 5, [1, 2, 3], mutate=cuda_flag)
    out["iters"], out["seed"] = iters, 7
    if os.environ.get("GOLDEN_TRACKER_ABLATIONS") != "only":
        np.savez_compressed(GOLD / "tracker_ref_runs.npz", **out)
    # (d) round 5: two of the reference's ablation configs whose callables the HIP tracker's probe maps onto its device solver
    #     -- the UNWEIGHTED least-squares fit (..._noreliableinl_plainLSq.py hands the library weights=None) and a re-detection test
    #     that is `return False` (..._neverwarp_wLSq.py: every frame takes the frame-to-frame branch, TRK:171-207)
    out2 = {}
    out, keep = out2, out
    with tempfile.TemporaryDirectory() as td:
        model = os.path.join(td, "sd.pth")
        torch.save(sd, model)
        run("plain", "YAOFT_single_control_repRAFT_sub500_noreliableinl_plainLSq.py", 6, [1, 2, 3])
        run("never", "YAOFT_single_control_repRAFT_sub500_neverwarp_wLSq.py", 7, [1, 2, 3])
    out2["iters"], out2["seed"] = iters, 7
    np.savez_compressed(GOLD / "tracker_ref_runs_ablations.npz", **out2)


def demo_frame(i):
    """Frame i (1-based) of the reference's demo sequence demo/V24_7 (1280 x 720 JPEG), decoded with PIL -> BGR uint8
    (cv2.imread's channel order; the decoder differs from OpenCV's libjpeg build by at most a grey level, which is why
    the DECODED frames are stored in the fixture)."""
    from PIL import Image
    rgb = np.asarray(Image.open(REF / "demo" / "V24_7" / f"{i:08d}.jpg").convert("RGB"))
    return np.ascontiguousarray(rgb[:, :, ::-1])


@torch.no_grad()
def gen_real():
    """BASELINE config 2 at its real size on REAL frames: the reference's flow network on a 720 x 1280 pair of the demo
    sequence (12 iterations), and the reference's own tracker (config ablation_08.py = IRLS; functional cv2 stub) over
    three of its frames.  Stored: the decoded uint8 frames, the 1/8-resolution outputs in full, the full-resolution flow
    and weight logits on a stride-4 lattice (+ their per-row means in full: a checksum of what the lattice skips)."""
    import tempfile
    from raft_core.weighted_raft import WeightedRAFT
    from pytracking.utils.config import load_config
    sd = synth.make_state_dict(seed=7, small=False, weighted=True)
    net = WeightedRAFT(ref_args(False)).eval()
    net.load_state_dict(sd, strict=True)
    f1, f2, f3 = demo_frame(1), demo_frame(4), demo_frame(7)
    out = dict(frame1=f1, frame2=f2, frame3=f3, seed=7, iters=12, stride=4)
    flow_low, flow_up, _, w_low, w_up = net(to_t(f1), to_t(f3), iters=12, test_mode=True)
    out.update(flow_low=flow_low.numpy(), w_low=w_low.numpy(),
               flow_up_s4=flow_up[..., ::4, ::4].contiguous().numpy(), w_up_s4=w_up[..., ::4, ::4].contiguous().numpy(),
               flow_up_rowmean=flow_up.double().mean(-1).numpy(), w_up_rowmean=w_up.double().mean(-1).numpy())
    install_functional_cv2()
    mask = np.zeros(f1.shape[:2], np.uint8)
    mask[180:560, 360:960] = 255
    with tempfile.TemporaryDirectory() as td:
        model = os.path.join(td, "sd.pth")
        torch.save(sd, model)
        conf = load_config(REF / "pytracking/configs/ablation_08.py")
        conf.flow_config.model, conf.flow_config.iters = model, 12
        est = conf.H_estimator
        conf.H_estimator = lambda a, b, w: torch.Tensor(est(a.as_subclass(FakeCuda), b, w))
        trk = conf.tracker_class(conf)
        trk.init(f1, mask)
        Hs, meta = [], []
        for f in (f2, f3):
            Hc, m = trk.track(f)
            Hs.append(np.asarray(Hc, np.float64))
            meta.append([float(bool(m.lost)), float(m.N_lost), float(bool(m.global_H_success))])
    out.update(mask=mask, track_H=np.stack(Hs), track_meta=np.asarray(meta))
    np.savez_compressed(GOLD / "real_720p.npz", **out)


@torch.no_grad()
def gen_degenerate():
    """Inputs on which a normalisation or a division degenerates (extractor.py:28-32 InstanceNorm with zero variance;
    saturated regions; identical frames = zero flow), through the reference's network: 128 x 160, 4 iterations."""
    from raft_core.weighted_raft import WeightedRAFT
    sd = synth.make_state_dict(seed=7, small=False, weighted=True)
    net = WeightedRAFT(ref_args(False)).eval()
    net.load_state_dict(sd, strict=True)
    H, W = 128, 160
    a, b = pair(H, W, seed=21)
    const = np.full((H, W, 3), 117, np.uint8)
    sat = a.copy()
    sat[:, :W // 2] = 255                                 # half the frame saturated, a black bar, texture elsewhere
    sat[40:60] = 0
    sat2 = b.copy()
    sat2[:, :W // 2] = 255
    sat2[43:63] = 0
    cases = {"constant": (const, const.copy()), "constant_vs_texture": (const, b), "saturated": (sat, sat2),
             "identical": (a, a.copy())}
    out = dict(seed=7, iters=4)
    for name, (x, y) in cases.items():
        flow_low, flow_up, _, w_low, w_up = net(to_t(x), to_t(y), iters=4, test_mode=True)
        out[f"{name}_img1"], out[f"{name}_img2"] = x, y
        out[f"{name}_flow_up"], out[f"{name}_w_up"] = flow_up.numpy(), w_up.numpy()
        out[f"{name}_flow_low"], out[f"{name}_w_low"] = flow_low.numpy(), w_low.numpy()
        print(name, "flow |max|", float(flow_up.abs().max()), "finite", bool(torch.isfinite(flow_up).all()),
              "w range", float(w_up.min()), float(w_up.max()))
    np.savez_compressed(GOLD / "degenerate_128x160_it4.npz", **out)


def metric_pair(H=1080, W=1920, seq_id=0, t=3):
    """The bench's own kind of input at the metric's resolution: the SURVEY 8d template of sequence `seq_id` and its frame
    t (numpy warp, woft_amd/synth.py).  Regenerated from the seeds by the tests (12 MB of noise texture is not a fixture);
    the fixture stores CRC32s of both images so that a drift of the generator is caught before anything is compared."""
    template = synth.make_template(H, W, seq_id=seq_id)
    return template, synth.make_frame(template, t)


def lattice_summary(prefix, flow_up, w_up, stride):
    """Full-resolution outputs as a stride-`stride` lattice + per-row means in full (a checksum of what the lattice skips)."""
    d = {f"{prefix}flow_up_s": flow_up[..., ::stride, ::stride].contiguous().numpy(),
         f"{prefix}flow_up_rowmean": flow_up.double().mean(-1).numpy()}
    if w_up is not None:
        d[f"{prefix}w_up_s"] = w_up[..., ::stride, ::stride].contiguous().numpy()
        d[f"{prefix}w_up_rowmean"] = w_up.double().mean(-1).numpy()
    return d


@torch.no_grad()
def gen_metric():
    """(1) The metric's own configuration (BASELINE.json metric / configs[1]): WeightedRAFT-full, 12 iterations, on a
    1080 x 1920 pair of the synthetic sequence (weighted_raft.py:186-290; ~20 s and ~12 GB on the CPU here).  Stored: the
    1/8-resolution flow and weight logits in full, the full-resolution outputs on a stride-8 lattice + row means.
    (2) BASELINE configs[0]: plain RAFT-small, 4 iterations, 480 x 640 (raft.py:169-262 -> raft_core/raft.py,
    extractor.py:244-267); flow_low in full, flow_up on a stride-4 lattice + row means."""
    import zlib
    from raft_core.weighted_raft import WeightedRAFT
    from raft_core.raft import RAFT
    crc = lambda a: np.int64(zlib.crc32(np.ascontiguousarray(a).tobytes()))
    sd = synth.make_state_dict(seed=7, small=False, weighted=True)
    net = WeightedRAFT(ref_args(False)).eval()
    net.load_state_dict(sd, strict=True)
    a, b = metric_pair()
    flow_low, flow_up, _, w_low, w_up = net(to_t(a), to_t(b), iters=12, test_mode=True)
    out = dict(seed=7, iters=12, H=1080, W=1920, seq_id=0, t=3, stride=8, crc_img1=crc(a), crc_img2=crc(b),
               flow_low=flow_low.numpy(), w_low=w_low.numpy(), **lattice_summary("", flow_up, w_up, 8))
    np.savez_compressed(GOLD / "metric_1080p_it12.npz", **out)
    print("1080p: mean |flow|", float(flow_up.abs().mean()), "w logits", float(w_up.min()), float(w_up.max()))
    del net

    sds = synth.make_state_dict(seed=8, small=True, weighted=False)
    nets = RAFT(ref_args(True)).eval()
    nets.load_state_dict(sds, strict=True)
    a, b = pair(480, 640, seed=14, shift=(5, -3))
    fl, fu = nets(to_t(a), to_t(b), iters=4, test_mode=True)
    out = dict(seed=8, iters=4, H=480, W=640, pair_seed=14, shift=np.asarray((5, -3)), stride=4, crc_img1=crc(a), crc_img2=crc(b),
               flow_low=fl.numpy(), **lattice_summary("", fu, None, 4))
    np.savez_compressed(GOLD / "cfg0_small_480x640_it4.npz", **out)
    print("480x640 small: mean |flow|", float(fu.abs().mean()))


def main():
    GOLD.mkdir(parents=True, exist_ok=True)
    install_stubs()
    torch.manual_seed(0)
    only = os.environ.get("GOLDEN_ONLY")
    if only in (None, "flow"):
        gen_flow()
    if only in (None, "heads"):
        gen_heads()
    if only in (None, "wrapper"):
        gen_wrapper()
    if only in (None, "hfit"):
        gen_hfit()
    if only in (None, "tracker"):
        gen_tracker()
    if only in (None, "degenerate"):
        gen_degenerate()
    if only in (None, "real"):
        gen_real()
    if only in (None, "metric"):
        gen_metric()
    for p in sorted(GOLD.iterdir()):
        print(f"{p.name:40s} {p.stat().st_size/1024:9.1f} KB")


if __name__ == "__main__":
    main()
