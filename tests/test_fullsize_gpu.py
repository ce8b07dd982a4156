"""Size-independent properties at BASELINE.json's full size (1080p: P = 135 x 240 source pixels, a 5.6 GB
correlation pyramid) -- the CPU oracle needs ~10 s and 12 GB per frame there, so parity at full size is
checked through invariants of the path instead (plus the 1080p EPE-vs-oracle gate inside bench.py)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HF, WF, C = 135, 240, 256


@pytest.fixture(scope="module")
def ops():
    from woft_amd import _lib, ops as o
    _lib.load()
    return o


@pytest.fixture(scope="module")
def pyramid(ops):
    """Full-size fp32 pyramid built exactly as the engine builds it."""
    g = torch.Generator(device="cuda").manual_seed(1)
    f1 = ops.new_act(1, HF, WF, C)
    f2 = ops.new_act(1, HF, WF, C)
    f1.t.copy_(torch.randn(HF * WF, C, device="cuda", generator=g))
    f2.t.copy_(torch.randn(HF * WF, C, device="cuda", generator=g))
    vols, dims, maps = [], [], [f2]
    cur = f2
    for l in range(4):
        h, w = cur.h, cur.w
        n = ops.tiled_dims(h, w)[2]
        rows = torch.zeros(ops._round_up(n, 128), C, device="cuda")
        ops.tile_rows(cur, rows)
        vol = torch.zeros(HF * WF, n, device="cuda")
        ops.run_conv(ops.corr_volume(f1, rows, n, vol, 1.0 / math.sqrt(C)))
        vols.append(vol)
        dims.append((h, w))
        if l < 3:
            nxt = ops.new_act(1, h // 2, w // 2, C)
            ops.avgpool2(cur, nxt)
            cur = nxt
            maps.append(nxt)
    torch.cuda.synchronize()
    return f1, f2, vols, dims


def test_volume_entries_and_pooling_consistency(ops, pyramid):
    f1, f2, vols, dims = pyramid
    rs = np.random.RandomState(0)
    P = HF * WF
    # (1) spot entries of level 0 are the scaled dot products (fp64 reference on the host)
    ps, qs = rs.randint(0, P, 64), rs.randint(0, P, 64)
    a, b = f1.t[ps].double().cpu(), f2.t[qs].double().cpu()
    ref = (a * b).sum(1) / 16.0
    v0 = ops.untile_planes(vols[0][torch.from_numpy(ps).cuda()], HF, WF).reshape(64, -1).cpu()
    got = v0[torch.arange(64), torch.from_numpy(qs)]
    assert float((got.double() - ref).abs().max()) < 2e-4
    # (2) level l+1 equals the 2x2 average of level l (corr.py:25-27), floor sizes, for a sample of source pixels
    idx = torch.from_numpy(rs.randint(0, P, 48)).cuda()
    for l in range(3):
        h, w = dims[l]
        lo = ops.untile_planes(vols[l][idx], h, w)
        hi = ops.untile_planes(vols[l + 1][idx], h // 2, w // 2)
        pooled = torch.nn.functional.avg_pool2d(lo[:, None], 2, stride=2)[:, 0]
        assert float((pooled - hi).abs().max()) < 3e-4, l
    # (3) the tile padding of every plane is exactly zero
    for l in range(4):
        h, w = dims[l]
        part = vols[l][idx]
        assert torch.equal(ops.tile_planes(ops.untile_planes(part, h, w)), part)


def test_lookup_properties_fullsize(ops, pyramid):
    f1, f2, vols, dims = pyramid
    P = HF * WF
    idx = torch.arange(P, device="cuda")
    grid = torch.stack([idx % WF, idx // WF], 1).float()
    out = torch.zeros(P, 352, device="cuda")
    # (1) at integer coordinates the centre tap of level 0 (i = j = 4 -> channel 40) is vol[p][p + d]
    d = torch.tensor([3.0, -2.0], device="cuda")
    coords = (grid + d).contiguous()
    ops.run_lookup(ops.make_lookup_params(vols, dims, coords, out, 4))
    torch.cuda.synchronize()
    sample = torch.arange(0, P, 997, device="cuda")
    x, y = (sample % WF) + 3, (sample // WF) - 2
    ok = (x >= 0) & (x < WF) & (y >= 0) & (y < HF)
    planes = ops.untile_planes(vols[0][sample], HF, WF)
    ref = torch.where(ok, planes[torch.arange(len(sample)), y.clamp(0, HF - 1), x.clamp(0, WF - 1)], torch.zeros(()).cuda())
    assert float((out[sample, 40] - ref).abs().max()) == 0.0
    # (2) shifting the query by one level-0 pixel shifts the level-0 window by one tap (x-major: i*9 + j)
    out2 = torch.zeros(P, 352, device="cuda")
    ops.run_lookup(ops.make_lookup_params(vols, dims, (coords + torch.tensor([1.0, 0.0], device="cuda")).contiguous(), out2, 4))
    torch.cuda.synchronize()
    a = out[:, :81].reshape(P, 9, 9)[:, 1:, :]
    b = out2[:, :81].reshape(P, 9, 9)[:, :-1, :]
    assert torch.equal(a, b)
    # (3) far outside the map every tap is zero
    far = (grid + 1.0e4).contiguous()
    ops.run_lookup(ops.make_lookup_params(vols, dims, far, out2, 4))
    torch.cuda.synchronize()
    assert float(out2.abs().max()) == 0.0
    # (4) bilinearity: the sample at x + 0.5 is the mean of the samples at x and x + 1 (level 0)
    ops.run_lookup(ops.make_lookup_params(vols, dims, (coords + torch.tensor([0.5, 0.0], device="cuda")).contiguous(), out2, 4))
    torch.cuda.synchronize()
    mid = out2[:, :81].reshape(P, 9, 9)[:, :-1, :]
    assert float((mid - 0.5 * (out[:, :81].reshape(P, 9, 9)[:, :-1, :] + a)).abs().max()) < 1e-5


def test_hfit_recovers_known_homography_fullres(ops):
    """Every pixel of a 1080p frame as a correspondence (N = 2 073 600, the no-subsampler configs): exact
    correspondences of a known H must give that H back; IRLS must agree with plain LSq on clean data."""
    H, W = 1080, 1920
    n = H * W
    Hgt = np.array([[1.01, 0.02, 14.0], [-0.015, 0.99, -9.0], [1.5e-5, -1e-5, 1.0]])
    idx = torch.arange(n, device="cuda")
    a = torch.stack([idx % W, idx // W], 1).double()
    ah = torch.cat([a, torch.ones(n, 1, device="cuda", dtype=torch.float64)], 1) @ torch.from_numpy(Hgt).cuda().t()
    b = (ah[:, :2] / ah[:, 2:]).float().contiguous()
    a = a.float().contiguous()
    w = torch.rand(n, device="cuda") * 0.9 + 0.1
    Hd, st = torch.zeros(9, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
    c = np.array([[0, 0, 1], [W, 0, 1], [W, H, 1], [0, H, 1.0]]).T
    for kw in (dict(), dict(reweight=2, huber_k=2.0, n_irls=5)):
        ops.hfit(a, b, w, Hd, st, **kw)
        torch.cuda.synchronize()
        assert int(st.item()) == 0
        He = Hd.cpu().numpy().reshape(3, 3).astype(np.float64)
        pa, pb = He @ c, Hgt @ c
        assert np.abs(pa[:2] / pa[2] - pb[:2] / pb[2]).max() < 0.02
    fr = torch.zeros(1, device="cuda")
    ops.inlier_frac(a, b, Hd, fr, thr=0.5)
    torch.cuda.synchronize()
    assert float(fr.item()) == 1.0


@pytest.mark.parametrize("precision,corr,iters", [("bf16x3", "otf", 2), ("bf16", "otf", 32), ("bf16", "volume", 32)])
def test_flow_of_identical_frames_is_deterministic_fullsize(precision, corr, iters):
    """1080p end to end: two runs of the same pair are bit-identical (no atomics / no order dependence on the
    path), the pinned-template cache gives the same answer as a cold call, and the int64 source grid is exact.
    ("bf16", 32 iterations: BASELINE config 3's operating point at its full size, both correlation modes.)"""
    from woft_amd import synth
    from woft_amd.config import Config
    from woft_amd.flow_provider import RAFTWrapper
    H, W = 1080, 1920
    c = Config()
    c.of_class, c.raft_type, c.class_params = RAFTWrapper, "weighted", Config()
    c.class_params.small = False
    c.model, c.iters, c.padding_mode, c.precision, c.corr = synth.make_state_dict(seed=7), iters, "nopad", precision, corr
    fl = RAFTWrapper(c)
    assert fl.engine.precision == precision and fl.engine.corr == corr
    t = synth.make_template(H, W, seq_id=9)
    f = np.roll(t, (3, -5), axis=(0, 1)).copy()
    s1, d1, w1 = fl.compute_flow(t, f, mode="TC", do_sigmoid=True)
    d1, w1 = d1.clone(), w1.clone()
    fl.pin_source(t)
    fl.compute_flow(t, f, mode="TC", do_sigmoid=True)              # fills the cache
    s2, d2, w2 = fl.compute_flow(t, f, mode="TC", do_sigmoid=True)   # served from the cache
    torch.cuda.synchronize()
    assert torch.equal(d1, d2) and torch.equal(w1, w2)
    idx = torch.arange(H * W, device="cuda")
    assert torch.equal(s2[0], idx % W) and torch.equal(s2[1], idx // W) and s2.dtype == torch.int64
    assert bool(torch.isfinite(d2).all()) and float(w2.min()) >= 0.0 and float(w2.max()) <= 1.0


def test_volume_free_lookup_equals_volume_lookup_fullsize(ops):
    """1080p feature size, split-bf16: the on-the-fly lookup (no volume) returns bit for bit what the lookup in the
    5.6 GB volume built by the correlation GEMM returns -- for a smooth field, for independent per-pixel offsets and
    for windows hanging over every border."""
    g = torch.Generator(device="cuda").manual_seed(3)
    f1 = torch.randn(HF * WF, C, device="cuda", generator=g) * 0.3
    f2 = ops.new_act(1, HF, WF, C)
    f2.t.copy_(torch.randn(HF * WF, C, device="cuda", generator=g) * 0.3)

    def split(t):
        o = torch.zeros(t.shape[0], 2 * C, dtype=torch.bfloat16, device="cuda")
        ops.split_bf16_lines(t.contiguous(), o)
        return o
    a = torch.zeros(ops._round_up(HF * WF, 128), C, device="cuda")
    a[:HF * WF] = f1
    sa = split(a)
    vols, dims, f2s, cur = [], [], [], f2
    for l in range(4):
        h, w = cur.h, cur.w
        n = ops.tiled_dims(h, w)[2]
        rows = torch.zeros(ops._round_up(n, 128), C, device="cuda")
        ops.tile_rows(cur, rows)
        vol = torch.zeros(HF * WF, n, device="cuda")
        ops.corr_gemm_bf16(sa, split(rows), HF * WF, n, 1.0 / math.sqrt(C), vol, 3)
        vols.append(vol)
        dims.append((h, w))
        f2s.append(split(cur.t))
        if l < 3:
            nxt = ops.new_act(1, h // 2, w // 2, C)
            ops.avgpool2(cur, nxt)
            cur = nxt
    idx = torch.arange(HF * WF, device="cuda")
    grid = torch.stack([idx % WF, idx // WF], 1).float()
    fields = {
        "smooth": grid * 1.01 + torch.tensor([2.3, -1.7], device="cuda"),
        "scattered": grid + (torch.rand(HF * WF, 2, device="cuda", generator=g) * 2 - 1) * 12.0,
        "borders": grid * 1.2 - torch.tensor([20.0, 12.0], device="cuda"),
    }
    for name, coords in fields.items():
        coords = coords.contiguous()
        ref = torch.zeros(HF * WF, 352, device="cuda")
        out = torch.zeros(HF * WF, 352, device="cuda")
        ops.run_lookup(ops.make_lookup_params(vols, dims, coords, ref, 4))
        ops.run_lookup_otf(ops.make_lookup_otf_params(sa, f2s, dims, HF, WF, C, coords, out, 4, 3))
        torch.cuda.synchronize()
        assert torch.equal(out, ref), f"{name}: volume-free lookup differs from the lookup in the volume"
