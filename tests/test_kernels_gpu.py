"""GPU parity tests, kernel by kernel, through the C ABI (ctypes) against the CPU oracle /
plain PyTorch fp32 references of the same op.  Run on the MI355X box: pytest -m gpu."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import hfit_ref, raft_ref, tracker_ref  # noqa: E402  (checker only)


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    from woft_amd import _lib, ops as o
    _lib.load()
    return o


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def _close(a, b, atol, rtol=0.0, what=""):
    a, b = a.detach().cpu().float(), b.detach().cpu().float()
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    assert bool((err <= tol).all()), f"{what}: max err {float(err.max()):.3e} (tol {atol:.1e}), max ref {float(b.abs().max()):.3e}"


# ------------------------------------------------------------------------------------------
# convolution engine
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cin,cout,k,stride,h,w,tiles", [
    (64, 64, 3, 1, 20, 28, (64, 64)),
    (64, 96, 3, 2, 21, 27, (64, 128)),
    (96, 128, 3, 1, 17, 25, (128, 128)),
    (128, 256, 1, 1, 17, 25, (128, 64)),
    (324, 256, 1, 1, 9, 13, None),
    (256, 2, 3, 1, 17, 25, None),
    (64, 96, 1, 2, 20, 28, None),
])
@pytest.mark.parametrize("precision,tol", [("fp32", 2e-5), ("bf16x3", 1e-4), ("bf16", 3e-2), ("fp16", 4e-3)])
def test_conv_plain(ops, cin, cout, k, stride, h, w, tiles, precision, tol):
    x = _rand(2, cin, h, w, seed=1)
    wt = _rand(cout, cin, k, k, seed=2, scale=1.0 / math.sqrt(cin * k * k))
    b = _rand(cout, seed=3, scale=0.1)
    ref = F.conv2d(x, wt, b, stride=stride, padding=k // 2)
    pc = ops.pack_conv(wt, b, stride=stride)
    xa = ops.act_from_nchw(x, cs=ops._round_up(cin, 32))
    out = ops.conv2d(xa, pc, tiles=tiles, precision=precision)
    torch.cuda.synchronize()
    _close(out.nchw(), ref, tol, what=f"conv {precision}")
    # relu epilogue, written at a channel offset into a wider buffer
    big = ops.new_act(2, ref.shape[2], ref.shape[3], cout + 8, cs=ops._round_up(cout + 8, 4), zero=True)
    ops.run_conv(ops.conv_params(xa, pc, big, co_off=8, epi=ops._lib.EPI_RELU, tiles=tiles, precision=precision))
    torch.cuda.synchronize()
    _close(big.t[:, 8:8 + cout].reshape(2, ref.shape[2], ref.shape[3], cout).permute(0, 3, 1, 2), F.relu(ref), tol,
           what="conv relu offset")
    assert float(big.t[:, :8].abs().max()) == 0.0


@pytest.mark.parametrize("mode", [1, 2])
def test_conv_halo_norm_on_load(ops, mode):
    """InstanceNorm (+ ReLU) of the producer applied inside the consumer's LDS-halo loader (extractor.py:44-47):
    identical to normalising first (same fp32 operations), zero padding applied after the normalisation."""
    x = _rand(1, 96, 21, 37, seed=91, scale=2.0) + 0.3
    wt = _rand(96, 96, 3, 3, seed=92, scale=1 / math.sqrt(96 * 9))
    b = _rand(96, seed=93, scale=0.1)
    mean, rstd = _rand(96, seed=94, scale=0.5).cuda(), (_rand(96, seed=95).abs() + 0.5).cuda()
    pc = ops.pack_conv(wt, b)
    xa = ops.act_from_nchw(x)
    xn = ops.new_act(1, 21, 37, 96, zero=True)
    ops.inorm_apply(xa, mean, rstd, xn, mode - 1)
    o1, o2 = ops.new_act(1, 21, 37, 96, zero=True), ops.new_act(1, 21, 37, 96, zero=True)
    p1 = ops.conv_params(xa, pc, o1, precision="bf16x3", in_norm=mode, in_stats=(mean, rstd))
    p2 = ops.conv_params(xn, pc, o2, precision="bf16x3")
    assert p1.halo != 0 and p1.in_norm == mode
    ops.run_conv(p1)
    ops.run_conv(p2)
    torch.cuda.synchronize()
    assert torch.equal(o1.t, o2.t)
    ref = F.conv2d(F.relu((x - mean.cpu().view(1, -1, 1, 1)) * rstd.cpu().view(1, -1, 1, 1)) if mode == 2 else
                   (x - mean.cpu().view(1, -1, 1, 1)) * rstd.cpu().view(1, -1, 1, 1), wt, b, padding=1)
    _close(o1.nchw(), ref, 2e-4, what="norm-on-load conv")


@pytest.mark.parametrize("precision,tol", [("bf16x3", 2e-5), ("bf16", 2e-2)])
def test_wh_mean_epilogue(ops, precision, tol):
    """Last weight-head layer with ReLU + 1x1 conv + patch mean fused into the whole-patch kernel's epilogue
    (weighted_raft.py:340-341,378-383): one float per patch, the activation is never stored."""
    n_patch = 53
    x = F.relu(_rand(n_patch, 128, 9, 9, seed=81))
    wt = _rand(128, 128, 3, 3, seed=82, scale=1 / math.sqrt(128 * 9))
    b = _rand(128, seed=83, scale=0.1)
    w6, b6 = _rand(128, seed=84, scale=0.2), 0.37
    act = F.relu(F.conv2d(x.double(), wt.double(), b.double(), padding=1))
    ref = (act * w6.double().view(1, 128, 1, 1)).sum(1).mean(dim=(1, 2)) + b6
    pc = ops.pack_conv(wt, b)
    xa = ops.act_from_nchw(x)
    dummy = ops.new_act(n_patch, 9, 9, 128, zero=True)
    p = ops.conv_params(xa, pc, dummy, epi=ops._lib.EPI_RELU, precision=precision)
    assert p.halo == 2
    out = torch.full((n_patch,), -7.0, device="cuda")
    w6d, b6d = w6.cuda(), torch.tensor([b6], device="cuda")
    p.epi = ops._lib.EPI_WH_MEAN
    p.e0, p.e1 = ops.ptr(w6d), ops.ptr(b6d)
    p.out, p.ldo, p.co_off = ops.ptr(out), 1, 0
    ops.run_conv(p)
    torch.cuda.synchronize()
    _close(out, ref.float(), tol, what="wh mean epilogue")
    assert float(dummy.t.abs().max()) == 0.0              # nothing else was written


@pytest.mark.parametrize("n", [9, 7])
def test_wh_conv0(ops, n):
    """First weight-head conv (weighted_raft.py:336) straight from the lookup buffer: input channels are the
    (hp wp level) re-read of the level-major lookup row (weighted_raft.py:267-272) + the mean channel."""
    P = 37
    lookup = _rand(P, 4 * n * n + 28, seed=71, scale=3.0)
    mean = _rand(P, seed=72)
    w0 = _rand(128, 5, 3, 3, seed=73, scale=1 / math.sqrt(45))
    b0 = _rand(128, seed=74, scale=0.1)
    x = torch.cat([lookup[:, :4 * n * n].reshape(P, n, n, 4).permute(0, 3, 1, 2), mean.view(P, 1, 1, 1).expand(P, 1, n, n)], 1)
    ref = F.relu(F.conv2d(x.double(), w0.double(), b0.double(), padding=1)).float()
    pc = ops.pack_conv(w0, b0, flat_cs=8)
    wt = pc.wgt[:128].t().contiguous()
    out = torch.zeros(P, n, n, 128, device="cuda")
    lib = ops._lib.load()
    lk, mn = lookup.cuda().contiguous(), mean.cuda().contiguous()
    ops.check(lib.woft_wh_conv0(ops.ptr(lk), lk.shape[1], ops.ptr(mn), P, n, ops.ptr(wt), ops.ptr(pc.bias), ops.ptr(out),
                                None, ops.stream_ptr()), "woft_wh_conv0")
    torch.cuda.synchronize()
    _close(out.permute(0, 3, 1, 2), ref, 3e-6, rtol=3e-6, what="wh conv0")


@pytest.mark.parametrize("precision,tol", [("bf16x3", 3e-5), ("bf16", 3e-2)])
@pytest.mark.parametrize("subset", [False, True])
def test_wh_first_two_layers_one_launch(ops, precision, tol, subset):
    """weighted_raft.py:336-338: conv(5->128)+ReLU evaluated inside the launch of the following 128->128 layer,
    chunk by chunk on the matrix cores, straight from the lookup windows (woft_conv_params.wh0_*); with an index
    the launch runs on a subset of the source pixels."""
    P, n = 41, 9
    lookup = _rand(P, 4 * n * n + 28, seed=91, scale=3.0)
    mean = _rand(P, seed=92)
    w0 = _rand(128, 5, 3, 3, seed=93, scale=1 / math.sqrt(45))
    b0 = _rand(128, seed=94, scale=0.1)
    w1 = _rand(128, 128, 3, 3, seed=95, scale=1 / math.sqrt(128 * 9))
    b1 = _rand(128, seed=96, scale=0.1)
    x = torch.cat([lookup[:, :4 * n * n].reshape(P, n, n, 4).permute(0, 3, 1, 2), mean.view(P, 1, 1, 1).expand(P, 1, n, n)], 1)
    a1 = F.relu(F.conv2d(x.double(), w0.double(), b0.double(), padding=1))
    ref = F.relu(F.conv2d(a1, w1.double(), b1.double(), padding=1)).float()
    sel = torch.tensor([5, 0, 40, 17, 18, 33, 2], dtype=torch.int32) if subset else None
    n_win = int(sel.numel()) if subset else P
    if subset:
        ref = ref[sel.long()]
    pc1 = ops.pack_conv(w1, b1)
    frag = ops.pack_wh0_frags(w0, 2 if precision == "bf16x3" else 1)
    lk = ops.Act(lookup.cuda().contiguous(), 1, 1, P, 4 * n * n)
    unused = ops.new_act(n_win, n, n, 128, zero=True)
    out = ops.new_act(n_win, n, n, 128, zero=True)
    mn, b0d = mean.cuda().contiguous(), b0.cuda().contiguous()
    idx = sel.cuda() if subset else None
    p = ops.conv_params(unused, pc1, out, epi=ops._lib.EPI_RELU, precision=precision, wh0=(lk, mn, frag, b0d, idx))
    assert p.halo == 2
    ops.run_conv(p)
    torch.cuda.synchronize()
    _close(out.nchw(), ref, tol, rtol=tol, what="fused first layer")
    # ... and with the head's tail fused as well (one float per window)
    w6, b6 = _rand(128, seed=97, scale=0.2), -0.21
    w2 = _rand(128, 128, 3, 3, seed=98, scale=1 / math.sqrt(128 * 9))
    pc2 = ops.pack_conv(w2, b1)
    a2 = F.relu(F.conv2d(a1, w2.double(), b1.double(), padding=1))
    ref6 = ((a2 * w6.double().view(1, 128, 1, 1)).sum(1).mean(dim=(1, 2)) + b6).float()
    res = torch.full((P,), -7.0, device="cuda")
    w6d, b6d = w6.cuda(), torch.tensor([b6], device="cuda")
    q = ops.conv_params(unused, pc2, out, epi=ops._lib.EPI_RELU, precision=precision, wh0=(lk, mn, frag, b0d, idx))
    q.epi, q.e0, q.e1 = ops._lib.EPI_WH_MEAN, ops.ptr(w6d), ops.ptr(b6d)
    q.out, q.ldo, q.co_off = ops.ptr(res), 1, 0
    q.out_index = ops.ptr(idx) if subset else None
    ops.run_conv(q)
    torch.cuda.synchronize()
    if subset:
        _close(res[sel.long().cuda()], ref6[sel.long()], tol, rtol=tol, what="fused first layer + tail (subset)")
        rest = torch.ones(P, dtype=torch.bool); rest[sel.long()] = False
        assert bool((res.cpu()[rest] == -7.0).all())
    else:
        _close(res, ref6, tol, rtol=tol, what="fused first layer + tail")


@pytest.mark.parametrize("cin,cout,h,w", [(256, 2, 19, 37), (256, 1, 5, 16), (128, 2, 17, 33), (256, 2, 3, 3)])
def test_conv3x3_narrow(ops, cin, cout, h, w):
    """Flow head conv2 (update.py:10-17) on the vector ALUs: exact fp32 products, written at a channel offset."""
    x = F.relu(_rand(2, cin, h, w, seed=61))
    wt = _rand(cout, cin, 3, 3, seed=62, scale=1 / math.sqrt(cin * 9))
    b = _rand(cout, seed=63, scale=0.1)
    ref = F.conv2d(x.double(), wt.double(), b.double(), padding=1).float()
    pc = ops.pack_conv(wt, b)
    xa = ops.act_from_nchw(x)
    assert ops.narrow_ok(xa, pc)
    out = ops.new_act(2, h, w, 4, cs=4, zero=True)
    ops.conv3x3_narrow(xa, pc, out, co_off=1)
    torch.cuda.synchronize()
    _close(out.nchw()[:, 1:1 + cout], ref, 2e-6, rtol=2e-6, what="narrow conv")
    assert float(out.t[:, 0].abs().max()) == 0.0 and float(out.t[:, 1 + cout:].abs().max()) == 0.0


@pytest.mark.parametrize("cin,h,w", [(256, 19, 37), (128, 5, 16)])
def test_flow_head_update_one_launch(ops, cin, h, w):
    """update.py:10-17 + weighted_raft.py:236-237 in one launch: bit-identical to the narrow conv followed by
    woft_coords_update (same fp32 operations)."""
    x = F.relu(_rand(1, cin, h, w, seed=64))
    wt = _rand(2, cin, 3, 3, seed=65, scale=1 / math.sqrt(cin * 9))
    b = _rand(2, seed=66, scale=0.1)
    pc = ops.pack_conv(wt, b)
    xa = ops.act_from_nchw(x)
    start = (_rand(h * w, 2, seed=67, scale=30.0) + 15.0).cuda()
    res = []
    for fused in (False, True):
        coords = start.clone()
        delta = ops.new_act(1, h, w, 2, cs=4, zero=True)
        flow4 = torch.full((h * w, 4), -3.0, device="cuda")
        cat = torch.full((h * w, 12), -5.0, device="cuda")
        if fused:
            ops.flow_head_update(xa, pc, delta, coords, flow4, cat[:, 6:], 12)
        else:
            ops.conv3x3_narrow(xa, pc, delta)
            ops.coords_update(coords, delta.t, delta.cs, w, flow4, cat[:, 6:], 12)
        torch.cuda.synchronize()
        res.append((coords, delta.t.clone(), flow4, cat))
    for a, b_ in zip(res[0], res[1]):
        assert torch.equal(a, b_)
    ref = F.conv2d(x.double(), wt.double(), b.double(), padding=1).float()
    _close(res[1][1][:, :2].reshape(h, w, 2).permute(2, 0, 1)[None], ref, 2e-6, rtol=2e-6, what="flow head delta")
    assert bool((res[1][3][:, :6] == -5.0).all()) and bool((res[1][3][:, 8:] == -5.0).all())


@pytest.mark.parametrize("precision,tol", [("bf16x3", 3e-5), ("bf16", 2e-2), ("fp16", 3e-3)])
@pytest.mark.parametrize("cin,cmid,h,w", [(128, 256, 135, 240), (128, 256, 17, 25), (96, 128, 40, 64)])
def test_flow_head_in_one_conv_launch(ops, cin, cmid, h, w, precision, tol):
    """FlowHead (update.py:10-17) = conv2(relu(conv1(h))) with the 3x3 -> 2-channel conv2 folded into conv1's epilogue
    (WOFT_EPI_FLOWHEAD: 18 partial products per pixel and column tile on the matrix cores) + woft_flow_head_gather (3x3
    neighbour sum, bias, coords1 += delta, flow operands of the next iteration): against torch fp64 and against the
    two-launch path (conv1 -> woft_flow_head_update); sizes with ragged 8x16 tiles, both tile widths, the small model's
    channel counts."""
    x = torch.tanh(_rand(1, cin, h, w, seed=70))
    w1 = _rand(cmid, cin, 3, 3, seed=71, scale=1 / math.sqrt(cin * 9))
    b1 = _rand(cmid, seed=72, scale=0.1)
    w2 = _rand(2, cmid, 3, 3, seed=73, scale=1 / math.sqrt(cmid * 9))
    b2 = _rand(2, seed=74, scale=0.1)
    pc1, pc2 = ops.pack_conv(w1, b1), ops.pack_conv(w2, b2)
    xa = ops.act_from_nchw(x, cs=(cin + 31) // 32 * 32)
    frags = ops.pack_flowhead_frags(w2, 2 if precision == "bf16x3" else 1, f16=precision == "fp16")
    part = torch.full((4 * h * w, 20), float("nan"), device="cuda")
    p = ops.flowhead_params(xa, pc1, part, frags, precision=precision)
    assert p is not None and p.halo == 8
    start = (_rand(h * w, 2, seed=75, scale=30.0) + 15.0).cuda()
    coords = start.clone()
    delta = ops.new_act(1, h, w, 2, cs=4, zero=True)
    flow4 = torch.full((h * w, 4), -3.0, device="cuda")
    cat = torch.full((h * w, 12), -5.0, device="cuda")
    ops.run_conv(p)
    ops.flow_head_gather(part, p._n_planes, h, w, pc2.bias[:2].contiguous(), delta, coords, flow4, cat[:, 6:], 12)
    torch.cuda.synchronize()
    ref = F.conv2d(F.relu(F.conv2d(x.double(), w1.double(), b1.double(), padding=1)), w2.double(), b2.double(), padding=1)
    got = delta.t[:, :2].reshape(h, w, 2).permute(2, 0, 1)[None]
    _close(got, ref.float(), tol, rtol=tol, what=f"fused flow head {precision}")
    assert torch.equal(coords, start + delta.t[:, :2])
    gx = (torch.arange(h * w, device="cuda") % w).float()
    gy = torch.div(torch.arange(h * w, device="cuda"), w, rounding_mode="floor").float()
    assert torch.equal(flow4[:, 0], coords[:, 0] - gx) and torch.equal(flow4[:, 1], coords[:, 1] - gy)
    assert torch.equal(cat[:, 6:8], flow4[:, :2]) and bool((flow4[:, 2:] == 0).all())
    assert bool((cat[:, :6] == -5.0).all()) and bool((cat[:, 8:] == -5.0).all())
    # the two-launch path (conv1 with ReLU epilogue, then the exact-fp32 narrow conv): same delta up to the split-bf16
    # rounding of the second conv's products
    mid = ops.conv2d(xa, pc1, epi=1, precision=precision, c_out_stride=cmid)
    d2 = ops.new_act(1, h, w, 2, cs=4, zero=True)
    ops.flow_head_update(mid, pc2, d2, start.clone())
    torch.cuda.synchronize()
    _close(got, d2.t[:, :2].reshape(h, w, 2).permute(2, 0, 1)[None], tol, rtol=tol, what="fused vs two launches")


@pytest.mark.parametrize("precision", ["bf16x3", "bf16", "fp16"])
@pytest.mark.parametrize("h,w", [(135, 240), (17, 25)])
def test_two_layers_in_one_launch(ops, h, w, precision):
    """woft_conv2d_pair: two independent layers that select the same kernel instance share one launch -- the motion
    encoder's branches (update.py:91-95): convc1 (1x1, 324 -> 256) with convf1 (7x7 on the 2-channel flow, flat packing) on
    the per-tap kernel, convc2 (3x3, 256 -> 192) with convf2 (3x3, 128 -> 64) on the register-streamed kernel, each writing at
    a channel offset of a shared buffer.  Bit-identical to the two separate launches; layers on different kernels are refused."""
    corr = ops.act_from_nchw(_rand(1, 324, h, w, seed=80), cs=352)
    flow = ops.act_from_nchw(_rand(1, 2, h, w, seed=81, scale=5.0), cs=4)
    mk = lambda co, ci, k, seed, **kw: ops.pack_conv(_rand(co, ci, k, k, seed=seed, scale=1 / math.sqrt(ci * k * k)),
                                                      _rand(co, seed=seed + 1, scale=0.1), **kw)
    c1, f1 = mk(256, 324, 1, 82), mk(128, 2, 7, 84, flat_cs=4)
    c2, f2 = mk(192, 256, 3, 86), mk(64, 128, 3, 88)
    res = {}
    for paired in (False, True):
        a1 = ops.new_act(1, h, w, 256, zero=True)
        b1 = ops.new_act(1, h, w, 128, zero=True)
        cf = ops.new_act(1, h, w, 256, zero=True)
        pa = ops.conv_params(corr, c1, a1, epi=1, precision=precision)
        pb = ops.conv_params(flow, f1, b1, epi=1, precision=precision)
        assert ops.pair_ok(pa, pb) and pa.halo == pb.halo == 16          # (the streamed GEMM kernel, conv_1x1.hip)
        if paired:                                                       # ... and the same pair on the gather kernel: one launch too
            a0, b0 = ops.new_act(1, h, w, 256, zero=True), ops.new_act(1, h, w, 128, zero=True)
            pa0 = ops.conv_params(corr, c1, a0, epi=1, precision=precision, halo=0)
            pb0 = ops.conv_params(flow, f1, b0, epi=1, precision=precision, halo=0)
            assert ops.pair_ok(pa0, pb0) and not ops.pair_ok(pa, pb0)
            ops.run_conv_pair(pa0, pb0)
        if paired:
            ops.run_conv_pair(pa, pb)
        else:
            ops.run_conv(pa)
            ops.run_conv(pb)
        qa = ops.conv_params(a1, c2, cf, co_off=0, epi=1, precision=precision)
        qb = ops.conv_params(b1, f2, cf, co_off=192, epi=1, precision=precision)
        assert ops.pair_ok(qa, qb) and qa.halo == 8 and qa.tile_n == qb.tile_n == 64
        if paired:
            ops.run_conv_pair(qa, qb)
        else:
            ops.run_conv(qa)
            ops.run_conv(qb)
        torch.cuda.synchronize()
        res[paired] = (a1.t.clone(), b1.t.clone(), cf.t.clone())
        if paired:
            assert torch.equal(a0.t, a1.t) and torch.equal(b0.t, b1.t)
        assert not ops.pair_ok(pa, qa)
        if paired:
            with pytest.raises(Exception):
                ops.run_conv_pair(pa, qa)
    for x, y in zip(res[False], res[True]):
        assert torch.equal(x, y)
    assert float(res[True][2][:, :192].abs().max()) > 0 and float(res[True][2][:, 192:].abs().max()) > 0


@pytest.mark.parametrize("kh,kw", [(1, 5), (5, 1), (3, 3)])
@pytest.mark.parametrize("precision,tol", [("fp32", 1.0), ("bf16x3", 4.0)])
def test_conv_gru_epilogues(ops, kh, kw, precision, tol):
    """SepConvGRU half step (update.py:45-60) from two convs with two-source inputs
    (bf16x3 runs the LDS-halo kernel: 18x22 is not a multiple of the 8x16 tile)."""
    n, h, w = 1, 18, 22
    hprev = torch.tanh(_rand(n, 128, h, w, seed=4))
    xin = _rand(n, 256, h, w, seed=5)
    mk = lambda s: (_rand(128, 384, kh, kw, seed=s, scale=1 / math.sqrt(384 * kh * kw)), _rand(128, seed=s + 50, scale=0.1))
    (wz, bz), (wr, br), (wq, bq) = mk(6), mk(7), mk(8)
    pad = (kh // 2, kw // 2)
    hx = torch.cat([hprev, xin], 1)
    z = torch.sigmoid(F.conv2d(hx, wz, bz, padding=pad))
    r = torch.sigmoid(F.conv2d(hx, wr, br, padding=pad))
    q = torch.tanh(F.conv2d(torch.cat([r * hprev, xin], 1), wq, bq, padding=pad))
    ref = (1 - z) * hprev + z * q
    pzr = ops.pack_conv(torch.cat([wz, wr], 0), torch.cat([bz, br], 0), padding=pad)
    pq = ops.pack_conv(wq, bq, padding=pad)
    ha, xa = ops.act_from_nchw(hprev), ops.act_from_nchw(xin)
    zb, rh, hn = (ops.new_act(n, h, w, 128, zero=True) for _ in range(3))
    pzr_p = ops.conv_params(ha, pzr, zb, x2=xa, c_split=128, epi=ops._lib.EPI_GRU_ZR, split=128, e0=ha, out1=rh,
                            precision=precision)
    # default choice in the split-bf16 precisions: the kernel that streams the weights global -> registers (conv_regb)
    assert (pzr_p.halo in (8, 12)) == (precision != "fp32")
    ops.run_conv(pzr_p)
    ops.run_conv(ops.conv_params(rh, pq, hn, x2=xa, c_split=128, epi=ops._lib.EPI_GRU_Q, e0=ha, e1=zb,
                                 precision=precision))
    torch.cuda.synchronize()
    _close(zb.nchw(), z, 2e-5 * tol, what="z")
    _close(rh.nchw(), r * hprev, 2e-5 * tol, what="r*h")
    _close(hn.nchw(), ref, 3e-5 * tol, what="h")
    if precision != "fp32":
        # the LDS-staged-weights halo kernels (8x16 and 4x16 pixel tiles), both column widths of conv_regb and its
        # 4x16-pixel x 128-column layout (halo 12) compute the same products in the same order: bit-identical gate tensors
        for halo, tiles in ((1, None), (4, None), (8, (128, 64)), (8, (128, 128)), (12, None)):
            z2, rh2, h2 = (ops.new_act(n, h, w, 128, zero=True) for _ in range(3))
            ops.run_conv(ops.conv_params(ha, pzr, z2, x2=xa, c_split=128, epi=ops._lib.EPI_GRU_ZR, split=128, e0=ha,
                                         out1=rh2, precision=precision, halo=halo, tiles=tiles))
            ops.run_conv(ops.conv_params(rh2, pq, h2, x2=xa, c_split=128, epi=ops._lib.EPI_GRU_Q, e0=ha, e1=z2,
                                         precision=precision, halo=halo, tiles=tiles))
            torch.cuda.synchronize()
            assert torch.equal(z2.t, zb.t) and torch.equal(rh2.t, rh.t) and torch.equal(h2.t, hn.t), (halo, tiles)


@pytest.mark.parametrize("precision,tol", [("bf16x3", 2e-4), ("bf16", 5e-2)])
def test_conv_halo_patches_and_fallback(ops, precision, tol):
    """3x3 128->128 on batches of 9x9 patches (weight head layers 2/3, weighted_raft.py:336-340): the
    whole-patch LDS-halo kernel; and the same conv with the halo disabled must agree with it."""
    x = F.relu(_rand(41, 128, 9, 9, seed=40))
    wt = _rand(128, 128, 3, 3, seed=41, scale=1 / math.sqrt(128 * 9))
    b = _rand(128, seed=42, scale=0.1)
    ref = F.relu(F.conv2d(x, wt, b, padding=1))
    pc = ops.pack_conv(wt, b)
    xa = ops.act_from_nchw(x)
    o1, o2 = ops.new_act(41, 9, 9, 128, zero=True), ops.new_act(41, 9, 9, 128, zero=True)
    p1 = ops.conv_params(xa, pc, o1, epi=ops._lib.EPI_RELU, precision=precision)
    p2 = ops.conv_params(xa, pc, o2, epi=ops._lib.EPI_RELU, precision=precision, halo=0)
    assert p1.halo == 2 and p2.halo == 0
    ops.run_conv(p1)
    ops.run_conv(p2)
    torch.cuda.synchronize()
    _close(o1.nchw(), ref, tol, what="halo 9x9")
    _close(o2.nchw(), ref, tol, what="gather")
    _close(o1.nchw(), o2.nchw(), 2e-5, what="halo vs gather (same arithmetic, other order)")
    # image-sized input, ragged against the 8x16 tile, cout 126 written at an offset (update.py:86,97)
    x = _rand(1, 256, 19, 37, seed=43)
    wt = _rand(126, 256, 3, 3, seed=44, scale=1 / math.sqrt(256 * 9))
    b = _rand(126, seed=45, scale=0.1)
    ref = F.relu(F.conv2d(x, wt, b, padding=1))
    big = ops.new_act(1, 19, 37, 256, zero=True)
    p3 = ops.conv_params(ops.act_from_nchw(x), ops.pack_conv(wt, b), big, co_off=128, epi=ops._lib.EPI_RELU,
                         precision=precision)
    assert p3.halo == 8
    for other in (1, 4, 12):
        p4 = ops.conv_params(ops.act_from_nchw(x), ops.pack_conv(wt, b), ops.new_act(1, 19, 37, 126, cs=128, zero=True),
                             epi=ops._lib.EPI_RELU, precision=precision, halo=other)
        assert p4.halo == other
        ops.run_conv(p4)
        torch.cuda.synchronize()
        _close(p4._keep[3].nchw(), ref, tol, what="LDS-staged-weights halo kernel")
    ops.run_conv(p3)
    torch.cuda.synchronize()
    _close(big.nchw()[:, 128:254], ref, tol, what="halo 8x16 ragged")
    assert float(big.t[:, :128].abs().max()) == 0.0 and float(big.t[:, 254:].abs().max()) == 0.0


@pytest.mark.parametrize("precision,tol", [("fp32", 1.0), ("bf16x3", 6.0)])
def test_conv_flat_first_layer(ops, precision, tol):
    """7x7 stride-2 conv on a 3-channel image stored NHWC4 (extractor.py:132), flat K packing."""
    import functools
    conv2d = functools.partial(ops.conv2d, precision=precision)
    x = _rand(1, 3, 40, 56, seed=9)
    wt = _rand(64, 3, 7, 7, seed=10, scale=0.1)
    b = _rand(64, seed=11, scale=0.1)
    ref = F.conv2d(x, wt, b, stride=2, padding=3)
    pc = ops.pack_conv(wt, b, stride=2, padding=3, flat_cs=4)
    xa = ops.act_from_nchw(x, cs=4)
    out = conv2d(xa, pc)
    torch.cuda.synchronize()
    _close(out.nchw(), ref, 2e-5 * tol, what="conv7x7s2 flat")
    # 7x7 stride-1 on a 2-channel flow field (update.py:84)
    x = _rand(1, 2, 17, 25, seed=12, scale=3.0)
    wt = _rand(128, 2, 7, 7, seed=13, scale=0.1)
    b = _rand(128, seed=14, scale=0.1)
    ref = F.conv2d(x, wt, b, padding=3)
    out = conv2d(ops.act_from_nchw(x, cs=4), ops.pack_conv(wt, b, padding=3, flat_cs=4))
    torch.cuda.synchronize()
    _close(out.nchw(), ref, 2e-5 * tol, what="conv7x7 flow flat")
    # 3x3 on batches of 9x9 5-channel patches stored with 8 channels (weighted_raft.py:336-338)
    x = _rand(37, 5, 9, 9, seed=15, scale=5.0)
    wt = _rand(128, 5, 3, 3, seed=16, scale=0.2)
    b = _rand(128, seed=17, scale=0.1)
    ref = F.conv2d(x, wt, b, padding=1)
    out = conv2d(ops.act_from_nchw(x, cs=8), ops.pack_conv(wt, b, padding=1, flat_cs=8))
    torch.cuda.synchronize()
    _close(out.nchw(), ref, 5e-5 * tol, what="conv3x3 patches flat")


@pytest.mark.parametrize("precision", ["bf16x3", "bf16", "fp16"])
@pytest.mark.parametrize("cin,cout,h,w,two", [(324, 256, 135, 240, False), (324, 256, 17, 25, False), (128, 256, 9, 7, False),
                                              (96, 512, 33, 20, False), (160, 256, 24, 40, True), (32, 256, 8, 8, False)])
def test_conv_1x1_kernel_equals_gather_kernel(ops, cin, cout, h, w, two, precision):
    """Wide 1x1 layers on the streamed GEMM kernel (conv_1x1.hip, halo 16: 64 pixels x 256 columns per workgroup, activations
    converted once, weights global -> registers) against the per-tap gather kernel: every output bit-identical (update.py:89
    convc1 with its 324 -> 352 padded lookup rows, extractor.py:166); ragged pixel tiles, 1-11 chunks (fewer than the loader's
    prefetch depth included), two-source input, bias + ReLU at a channel offset."""
    x = _rand(1, cin, h, w, seed=70)
    wt = _rand(cout, cin, 1, 1, seed=71, scale=1.0 / math.sqrt(cin))
    b = _rand(cout, seed=72, scale=0.1)
    if two:                                   # channels [0, 96) from one tensor, [96, 160) from a second one (32-channel aligned split)
        pc = ops.pack_conv(wt, b)
        xa = ops.act_from_nchw(x[:, :96], cs=96)
        xb = ops.act_from_nchw(x[:, 96:], cs=64)
        kw = dict(x2=xb, c_split=96)
    else:
        pc = ops.pack_conv(wt, b)
        xa = ops.act_from_nchw(x, cs=ops._round_up(cin, 32))
        kw = {}
    outs = []
    for halo in (None, 0):
        out = ops.new_act(1, h, w, cout + 8, cs=cout + 8, zero=True)
        cp = ops.conv_params(xa, pc, out, co_off=8, epi=ops._lib.EPI_RELU, precision=precision, halo=halo, **kw)
        assert cp.halo == (16 if halo is None else 0), cp.halo
        ops.run_conv(cp)
        outs.append(out.t.clone())
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]), f"max diff {float((outs[0] - outs[1]).abs().max()):.3e}"
    assert float(outs[0][:, :8].abs().max()) == 0.0
    ref = F.relu(F.conv2d(x, wt, b))
    _close(outs[0][:, 8:].reshape(1, h, w, cout).permute(0, 3, 1, 2), ref, {"bf16x3": 1e-4, "bf16": 3e-2, "fp16": 4e-3}[precision],
           what="1x1 vs torch")


@pytest.mark.parametrize("precision", ["bf16x3", "bf16", "fp16"])
@pytest.mark.parametrize("h,w,k,flat_cs", [(135, 240, 7, 4), (17, 25, 7, 4), (8, 8, 7, 4), (5, 3, 7, 4), (24, 40, 3, 8)])
def test_flat_conv_on_the_1x1_kernel_and_shared_launch(ops, h, w, k, flat_cs, precision):
    """The motion encoder's convf1 (update.py:91: 7x7 on the 2-channel flow, "flat" packing: a K chunk = one tap row of the NHWC
    image) on the streamed GEMM kernel (conv_1x1.hip, halo 16): bit-identical to the per-tap gather kernel -- frames narrower than
    the kernel, rows / columns outside the image, a second flat form (3 taps x 8-channel pixels) -- and, with a wide 1x1 layer, in
    ONE launch (woft_conv2d_pair, either order of the two layers) bit-identical to the two separate launches."""
    cin = 2 if flat_cs == 4 else 7
    x = _rand(1, cin, h, w, seed=80)
    wt = _rand(128, cin, k, k, seed=81, scale=1.0 / math.sqrt(cin * k * k))
    b = _rand(128, seed=82, scale=0.1)
    pc = ops.pack_conv(wt, b, padding=k // 2, flat_cs=flat_cs)
    xa = ops.act_from_nchw(x, cs=flat_cs)
    outs, cps = [], []
    for halo in (None, 0):
        out = ops.new_act(1, h, w, 128, zero=True)
        cp = ops.conv_params(xa, pc, out, epi=ops._lib.EPI_RELU, precision=precision, halo=halo)
        assert cp.halo == (16 if halo is None else 0), cp.halo
        ops.run_conv(cp)
        outs.append(out)
        cps.append(cp)
    torch.cuda.synchronize()
    assert torch.equal(outs[0].t, outs[1].t), f"max diff {float((outs[0].t - outs[1].t).abs().max()):.3e}"
    _close(outs[0].nchw(), F.relu(F.conv2d(x, wt, b, padding=k // 2)), {"bf16x3": 1e-4, "bf16": 3e-2, "fp16": 4e-3}[precision], what="flat vs torch")
    # one launch with a wide 1x1 layer on the same pixels
    y = _rand(1, 96, h, w, seed=83)
    pw = ops.pack_conv(_rand(256, 96, 1, 1, seed=84, scale=0.1), _rand(256, seed=85, scale=0.1))
    ya = ops.act_from_nchw(y, cs=96)
    wide = ops.new_act(1, h, w, 256, zero=True)
    cw = ops.conv_params(ya, pw, wide, epi=ops._lib.EPI_RELU, precision=precision)
    assert cw.halo == 16 and ops.pair_ok(cw, cps[0]) and not ops.pair_ok(cw, cps[1])
    ops.run_conv(cw)
    torch.cuda.synchronize()
    want = (wide.t.clone(), outs[0].t.clone())
    for a, bb in ((cw, cps[0]), (cps[0], cw)):
        wide.t.fill_(float("nan"))
        outs[0].t.fill_(float("nan"))
        ops.run_conv_pair(a, bb)
        torch.cuda.synchronize()
        assert torch.equal(wide.t, want[0]) and torch.equal(outs[0].t, want[1]), (a is cw)


@pytest.mark.parametrize("precision", ["bf16x3", "bf16", "fp16"])
@pytest.mark.parametrize("h,w", [(75, 91), (64, 128), (272, 480), (13, 9)])
def test_conv_stem_kernel_equals_gather_kernel(ops, h, w, precision):
    """The encoders' 7x7 / stride-2 first layer on its own kernel (conv_stem.hip, halo 7: 8x16-pixel tiles, patch in LDS
    once, weights in registers) against the per-tap gather kernel: every output bit-identical (extractor.py:127-129,168);
    with InstanceNorm partial statistics (fnet) and with the BatchNorm-folded bias + ReLU epilogue (cnet)."""
    x = _rand(1, 3, h, w, seed=90)
    wt = _rand(64, 3, 7, 7, seed=91, scale=0.1)
    b = _rand(64, seed=92, scale=0.1)
    pc = ops.pack_conv(wt, b, stride=2, padding=3, flat_cs=4)
    xa = ops.act_from_nchw(x, cs=4)
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    outs, means = [], []
    for halo in (None, 0):
        for epi, with_stats in ((ops._lib.EPI_LINEAR, True), (ops._lib.EPI_RELU, False)):
            out = ops.new_act(1, ho, wo, 64, zero=True)
            stats = (torch.zeros(4096 * pc.cout_pad, device="cuda"), torch.zeros(4096 * pc.cout_pad, device="cuda")) if with_stats else None
            cp = ops.conv_params(xa, pc, out, epi=epi, stats=stats, precision=precision, halo=halo)
            assert cp.halo == (7 if halo is None else 0)
            ops.run_conv(cp)
            if with_stats:
                mean, rstd = torch.zeros(64, device="cuda"), torch.zeros(64, device="cuda")
                ops.inorm_finalize(stats, 2 * cp._m_tiles, pc.cout_pad, 64, ho * wo, mean, rstd)
                means.append((mean.clone(), rstd.clone()))
            outs.append(out.t.clone())
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[2]), f"raw: max diff {float((outs[0] - outs[2]).abs().max()):.3e}"
    assert torch.equal(outs[1], outs[3]), f"relu: max diff {float((outs[1] - outs[3]).abs().max()):.3e}"
    # (the partial sums are grouped by the kernels' own tiles: equal up to fp32 summation order)
    _close(means[0][0], means[1][0], 1e-6, what="mean")
    _close(means[0][1], means[1][1], 0.0, rtol=1e-5, what="rstd")
    ref = F.conv2d(x, wt, b, stride=2, padding=3)
    y = ops.Act(outs[0], 1, ho, wo, 64).nchw() if hasattr(ops, "Act") else None
    if y is not None:
        _close(y, ref, {"bf16x3": 1e-4, "bf16": 3e-2, "fp16": 4e-3}[precision], what="stem vs torch")


def test_residual_epilogue_and_bn_fold(ops):
    x = _rand(1, 64, 16, 24, seed=18)
    res = _rand(1, 64, 16, 24, seed=19)
    wt = _rand(64, 64, 3, 3, seed=20, scale=0.05)
    b = _rand(64, seed=21, scale=0.1)
    g, be = 1 + _rand(64, seed=22, scale=0.2), _rand(64, seed=23, scale=0.1)
    mu, var = _rand(64, seed=24, scale=0.1), 1 + _rand(64, seed=25, scale=0.3)
    y = F.batch_norm(F.conv2d(x, wt, b, padding=1), mu, var, g, be, False, 0.0, 1e-5)
    ref = F.relu(res + F.relu(y))
    wf, bf = ops.fold_bn(wt, b, g, be, mu, var)
    out = ops.new_act(1, 16, 24, 64, zero=True)
    ops.run_conv(ops.conv_params(ops.act_from_nchw(x), ops.pack_conv(wf, bf), out, epi=ops._lib.EPI_RELU_RES_RELU,
                                 e0=ops.act_from_nchw(res)))
    torch.cuda.synchronize()
    _close(out.nchw(), ref, 2e-5, what="bn-fold residual")


@pytest.mark.parametrize("tiles,precision", [((64, 64), "fp32"), ((128, 64), "fp32"), ((64, 64), "bf16x3")])
def test_instance_norm(ops, tiles, precision):
    x = _rand(1, 64, 30, 44, seed=26)
    wt = _rand(64, 64, 3, 3, seed=27, scale=0.05)
    b = _rand(64, seed=28, scale=0.3)
    res = F.relu(_rand(1, 64, 30, 44, seed=29))
    y = F.conv2d(x, wt, b, padding=1)
    pc = ops.pack_conv(wt, b)
    m = 30 * 44
    raw = ops.new_act(1, 30, 44, 64, zero=True)
    stats = (torch.zeros(64 * pc.cout_pad, device="cuda"), torch.zeros(64 * pc.cout_pad, device="cuda"))
    cp = ops.conv_params(ops.act_from_nchw(x), pc, raw, stats=stats, tiles=tiles, precision=precision)
    rows = 2 * cp._m_tiles
    ops.run_conv(cp)
    mean, rstd = torch.zeros(64, device="cuda"), torch.zeros(64, device="cuda")
    mean += 7.0
    ops.inorm_finalize(stats, rows, pc.cout_pad, 64, m, mean, rstd)
    out = ops.new_act(1, 30, 44, 64)
    for mode, ref in ((0, F.instance_norm(y)), (1, F.relu(F.instance_norm(y))),
                      (2, F.relu(res + F.relu(F.instance_norm(y))))):
        ops.inorm_apply(raw, mean, rstd, out, mode, res=ops.act_from_nchw(res) if mode == 2 else None)
        torch.cuda.synchronize()
        _close(out.nchw(), ref, 3e-5 if precision == "fp32" else 2e-4, what=f"instance norm mode {mode}")


@pytest.mark.parametrize("n_part,channels,ld", [(8100, 64, 64), (2040, 96, 128), (510, 128, 128), (37, 126, 128), (3, 64, 64),
                                                (5000, 256, 256)])
def test_instance_norm_finalize(ops, n_part, channels, ld):
    """woft_inorm_finalize: many workgroups + last-workgroup total (with the scratch), the one-workgroup path (without), and
    repeated calls on the same scratch all give the fp64 statistics of the partial sums; padding channels come out as zero."""
    g = torch.Generator().manual_seed(n_part)
    s1 = (torch.randn(n_part, ld, generator=g) * 40).cuda()
    s2 = (torch.rand(n_part, ld, generator=g) * 900 + 30).cuda()
    count = 64 * n_part
    mu = s1.double().sum(0) / count
    var = (s2.double().sum(0) / count - mu * mu).clamp_min(0)
    want_mean, want_rstd = mu.float(), (1.0 / torch.sqrt(var + 1e-5)).float()
    ws = ops.inorm_ws()
    outs = []
    for w in (ws, ws, None, ws):
        mean, rstd = torch.full((ld,), 7.0, device="cuda"), torch.full((ld,), 7.0, device="cuda")
        ops.inorm_finalize((s1, s2), n_part, ld, channels, count, mean, rstd, channels_pad=ld, ws=w)
        torch.cuda.synchronize()
        outs.append((mean.clone(), rstd.clone()))
        assert float((mean[:channels] - want_mean[:channels]).abs().max()) <= 1e-6 * float(want_mean.abs().max()) + 1e-7
        assert float(((rstd[:channels] - want_rstd[:channels]) / want_rstd[:channels]).abs().max()) <= 2e-6
        assert float(mean[channels:].abs().max() if channels < ld else 0.0) == 0.0
        assert float(rstd[channels:].abs().max() if channels < ld else 0.0) == 0.0
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[3][1])      # deterministic, scratch reusable
    assert int(ws.view(torch.int32)[-16:].abs().sum()) == 0                                  # the ticket is back at zero


def test_preprocess_and_pool(ops):
    rs = np.random.RandomState(0)
    img = rs.randint(0, 256, (37, 45, 3), dtype=np.uint8)
    t = torch.from_numpy(img).cuda()
    out = ops.new_act(1, 40, 48, 3, cs=4)
    ops.preprocess(t, out, 40, 48, 1, 1)
    torch.cuda.synchronize()
    x = torch.from_numpy(img[:, :, ::-1].copy()).permute(2, 0, 1).float()[None]
    ref = 2 * (F.pad(x, [1, 2, 1, 2], mode="replicate") / 255.0) - 1.0
    _close(out.nchw(), ref, 0.0, what="preprocess (bit exact)")
    f = _rand(1, 64, 17, 25, seed=30)
    fa = ops.act_from_nchw(f)
    o = ops.new_act(1, 8, 12, 64)
    ops.avgpool2(fa, o)
    torch.cuda.synchronize()
    _close(o.nchw(), F.avg_pool2d(f, 2, stride=2), 1e-6, what="avgpool")


@pytest.mark.parametrize("terms", [3, 1])
@pytest.mark.parametrize("h,w,c,levels", [(135, 240, 256, 4), (17, 25, 256, 4), (30, 44, 128, 4), (9, 8, 64, 3), (16, 16, 32, 1)])
def test_feature_pyramid_one_launch(ops, h, w, c, levels, terms):
    """woft_feature_pyramid (pooled maps + split operands of all levels in one launch) against the chained
    woft_avgpool2_nhwc / woft_split_bf16(_lines) calls it replaces: bit-identical (corr.py:25-27,77-81)."""
    f = _rand(1, c, h, w, seed=31, scale=3.0)
    split_of = lambda t: torch.zeros(t.shape[0], t.shape[1] * (2 if terms == 3 else 1), dtype=torch.bfloat16, device="cuda")
    ref_maps, hh, ww = [ops.act_from_nchw(f)], h, w
    for l in range(1, levels):
        hh, ww = hh // 2, ww // 2
        ref_maps.append(ops.new_act(1, hh, ww, c, zero=True))
        ops.avgpool2(ref_maps[l - 1], ref_maps[l])
    ref_split = [split_of(m.t) for m in ref_maps]
    for m, sp in zip(ref_maps, ref_split):
        if terms == 3:
            ops.split_bf16_lines(m.t, sp)
        else:
            ops.split_bf16(m.t, sp, None)
    maps = [ref_maps[0]] + [ops.new_act(1, m.h, m.w, c, zero=True) for m in ref_maps[1:]]
    splits = [split_of(m.t) for m in maps]
    ops.feature_pyramid(ops.PyramidArgs(maps, splits, terms))
    torch.cuda.synchronize()
    for l in range(levels):
        assert torch.equal(maps[l].t, ref_maps[l].t), f"pooled map of level {l}"
        assert torch.equal(splits[l].view(torch.int16), ref_split[l].view(torch.int16)), f"split operand of level {l}"
    _close(maps[-1].nchw(), F.avg_pool2d(f, 2 ** (levels - 1)) if levels > 1 else f, 1e-5, what="pyramid top vs torch") \
        if (h % (2 ** (levels - 1)) == 0 and w % (2 ** (levels - 1)) == 0) else None


# ------------------------------------------------------------------------------------------
# correlation volume + lookup
# ------------------------------------------------------------------------------------------
def _build_pyramid_gpu(ops, f1, f2, precision="fp32", presplit=False, vol_dtype=torch.float32):
    """f1, f2: (1, C, H, W) cpu tensors -> tiled volumes [P][ht*wt*16] on the GPU via tile_rows + conv/GEMM."""
    _, c, h, w = f1.shape
    a1, a2 = ops.act_from_nchw(f1), ops.act_from_nchw(f2)
    vols, dims = [], []
    cur = a2
    for l in range(4):
        hl, wl = cur.h, cur.w
        n = ops.tiled_dims(hl, wl)[2]
        rows = torch.zeros(ops._round_up(n, 128), c, device="cuda")
        ops.tile_rows(cur, rows)
        vol = torch.zeros(h * w, n, device="cuda", dtype=vol_dtype)
        hi, lo = torch.zeros_like(rows, dtype=torch.bfloat16), torch.zeros_like(rows, dtype=torch.bfloat16)
        if precision != "fp32":
            ops.split_bf16(rows, hi, lo)
        if presplit:            # the engine's path: both operands pre-split, woft_corr_gemm_bf16
            arows = torch.zeros(ops._round_up(h * w, 128), c, device="cuda")
            arows[:h * w] = a1.t
            if precision == "bf16x3":
                sa = torch.zeros(arows.shape[0], 2 * c, dtype=torch.bfloat16, device="cuda")
                sb = torch.zeros(rows.shape[0], 2 * c, dtype=torch.bfloat16, device="cuda")
                ops.split_bf16_lines(arows, sa)
                ops.split_bf16_lines(rows, sb)
                # line format: per 32 values [32 hi | 32 lo]
                assert torch.equal(sb.view(-1, 2, 32)[:, 0].reshape(-1, c), hi)
                assert torch.equal(sb.view(-1, 2, 32)[:, 1].reshape(-1, c), lo)
            else:
                sa, sb = torch.zeros_like(arows, dtype=torch.bfloat16), hi
                ops.split_bf16(arows, sa, None)
            ops.corr_gemm_bf16(sa, sb, h * w, n, 1.0 / math.sqrt(c), vol, 3 if precision == "bf16x3" else 1)
        else:
            ops.run_conv(ops.corr_volume(a1, rows, n, vol, 1.0 / math.sqrt(c), precision=precision, f2_hi=hi, f2_lo=lo))
        vols.append(vol)
        dims.append((hl, wl))
        if l < 3:
            nxt = ops.new_act(1, hl // 2, wl // 2, c)
            ops.avgpool2(cur, nxt)
            cur = nxt
    torch.cuda.synchronize()
    return vols, dims


@pytest.mark.parametrize("h,w,precision,tol", [(16, 20, "fp32", 3e-5), (17, 25, "fp32", 3e-5), (17, 25, "bf16x3", 2e-4),
                                               (16, 20, "bf16", 5e-2)])
def test_corr_volume_and_lookup(ops, h, w, precision, tol):
    f1, f2 = _rand(1, 256, h, w, seed=31), _rand(1, 256, h, w, seed=32)
    pyr = raft_ref.corr_pyramid(f1, f2)
    vols, dims = _build_pyramid_gpu(ops, f1, f2, precision)
    for l in range(4):
        hl, wl = dims[l]
        _close(ops.untile_planes(vols[l], hl, wl), pyr[l][:, 0], tol, what=f"volume level {l}")
        # the padding entries of the tiled layout are exact zeros (the lookup relies on it)
        full = ops.tile_planes(ops.untile_planes(vols[l], hl, wl))
        assert torch.equal(full, vols[l][:, :full.shape[1]])
    if precision != "fp32":
        return
    coords = raft_ref.coords_grid(1, h, w) + _rand(1, 2, h, w, seed=33, scale=6.0)
    coords[0, :, 0, 0] = torch.tensor([-7.3, 2.2])
    coords[0, :, 0, 1] = torch.tensor([w + 9.5, h + 3.0])
    ref = raft_ref.corr_lookup(pyr, coords, 4)
    cg = coords[0].permute(1, 2, 0).reshape(h * w, 2).contiguous().cuda()
    out = torch.zeros(h * w, 352, device="cuda")
    ops.run_lookup(ops.make_lookup_params(vols, dims, cg, out, 4))
    torch.cuda.synchronize()
    got = out[:, :324].reshape(1, h, w, 324).permute(0, 3, 1, 2)
    _close(got, ref, 1e-4, what="lookup")
    assert float(out[:, 324:].abs().max()) == 0.0


@pytest.mark.parametrize("h,w,precision,tol", [(17, 25, "bf16x3", 2e-4), (24, 40, "bf16x3", 2e-4), (16, 20, "bf16", 5e-2)])
def test_corr_gemm_presplit(ops, h, w, precision, tol):
    """woft_corr_gemm_bf16 (both operands pre-split, LDS-DMA fed) == the per-tap conv kernel's volume in the same
    precision mode (same products; the accumulation order inside a K step differs in bf16 mode), and == the oracle
    within tolerance."""
    f1, f2 = _rand(1, 256, h, w, seed=41), _rand(1, 256, h, w, seed=42)
    pyr = raft_ref.corr_pyramid(f1, f2)
    vols, dims = _build_pyramid_gpu(ops, f1, f2, precision, presplit=True)
    ref, _ = _build_pyramid_gpu(ops, f1, f2, precision)
    for l in range(4):
        hl, wl = dims[l]
        _close(ops.untile_planes(vols[l], hl, wl), pyr[l][:, 0], tol, what=f"volume level {l}")
        _close(vols[l], ref[l], tol * 0.1, what=f"level {l}: pre-split GEMM vs the conv kernel")


def test_lookup_on_the_fly_wanted_blocks(ops):
    """woft_lookup_otf_params.need: 8x8 blocks of source pixels without a wanted pixel are skipped (their output rows stay
    as they were), the others are computed in full -- the same bits as without the map."""
    h, w, c = 24, 40, 256
    f1, f2 = _rand(1, c, h, w, seed=51), _rand(1, c, h, w, seed=52)
    coords = raft_ref.coords_grid(1, h, w) + _rand(1, 2, h, w, seed=53, scale=3.0)
    cg = coords[0].permute(1, 2, 0).reshape(h * w, 2).contiguous().cuda()
    a1, a2 = ops.act_from_nchw(f1), ops.act_from_nchw(f2)

    def split(t):
        o = torch.zeros(t.shape[0], 2 * c, dtype=torch.bfloat16, device="cuda")
        ops.split_bf16_lines(t, o)
        return o
    f2s, dims, cur = [], [], a2
    for l in range(4):
        f2s.append(split(cur.t))
        dims.append((cur.h, cur.w))
        if l < 3:
            nxt = ops.new_act(1, cur.h // 2, cur.w // 2, c)
            ops.avgpool2(cur, nxt)
            cur = nxt
    full, part = torch.zeros(h * w, 352, device="cuda"), torch.full((h * w, 352), 3.0, device="cuda")
    ops.run_lookup_otf(ops.make_lookup_otf_params(split(a1.t), f2s, dims, h, w, c, cg, full, 4, 3))
    need = torch.zeros(h, w, dtype=torch.int32, device="cuda")
    need[3, 5] = 1            # block (0, 0)
    need[23, 39] = 1          # block (2, 4)
    p = ops.make_lookup_otf_params(split(a1.t), f2s, dims, h, w, c, cg, part, 4, 3)
    p.need = need.data_ptr()
    ops.run_lookup_otf(p)
    torch.cuda.synchronize()
    blk = torch.zeros(h, w, dtype=torch.bool)
    blk[0:8, 0:8] = True
    blk[16:24, 32:40] = True
    blk = blk.reshape(-1)
    assert torch.equal(part.cpu()[blk][:, :324], full.cpu()[blk][:, :324])
    assert float((part.cpu()[~blk] - 3.0).abs().max()) == 0.0


@pytest.mark.parametrize("h,w,dtype", [(17, 25, torch.float32), (24, 40, torch.float32), (33, 47, torch.bfloat16),
                                       (8, 9, torch.bfloat16)])
def test_lookup_tile_shapes(ops, h, w, dtype):
    """Tiled volume layout (4 x 4 elements): tile_planes / untile_planes round-trip, lookups on hand-made planes run (smooth,
    scattered and out-of-map coordinates), and woft_tile_rows orders the GEMM's B rows the way tile_planes orders a plane."""
    P = h * w
    g = torch.Generator().manual_seed(7)
    dims, planes = [], []
    hl, wl = h, w
    for l in range(4):
        dims.append((hl, wl))
        planes.append(torch.randn(P, hl, wl, generator=g).to(dtype).cuda())
        hl, wl = max(hl // 2, 1), max(wl // 2, 1)
    coords = raft_ref.coords_grid(1, h, w) + _rand(1, 2, h, w, seed=71, scale=9.0)
    coords[0, :, 0, 0] = torch.tensor([-7.3, 2.2])
    coords[0, :, h - 1, w - 1] = torch.tensor([w + 9.5, h + 3.0])
    coords[0, :, 1, 1] = torch.tensor([-30.0, -30.0])
    cg = coords[0].permute(1, 2, 0).reshape(P, 2).contiguous().cuda()
    vols = [ops.tile_planes(pl) for pl in planes]
    for v, pl, (a, b) in zip(vols, planes, dims):
        assert torch.equal(ops.untile_planes(v, a, b), pl)
    out = torch.zeros(P, 352, device="cuda")
    ops.run_lookup(ops.make_lookup_params(vols, dims, cg, out, 4))
    torch.cuda.synchronize()
    assert float(out[:, :324].abs().max()) > 0
    # woft_tile_rows: row (tile, dy, dx) of the output = pixel (4 ty + dy, 4 tx + dx) of the map, zero rows outside
    x = ops.new_act(1, h, w, 8)
    x.t.normal_()
    ht, wt, n = ops.tiled_dims(h, w)
    rows = torch.full((n, 8), 7.0, device="cuda")
    ops.tile_rows(x, rows)
    want = ops.tile_planes(x.t.reshape(h, w, 8).permute(2, 0, 1).contiguous())      # (8, n)
    assert torch.equal(rows.t().contiguous(), want)


@pytest.mark.parametrize("h,w,precision", [(17, 25, "bf16"), (24, 40, "bf16"), (16, 20, "bf16x3")])
def test_bf16_storage_volume(ops, h, w, precision):
    """bf16-STORAGE volume (the plain-bf16 operating point, SURVEY 8d: 2096 B per pixel and lookup): the correlation GEMM
    with a bf16 output == its fp32 output rounded to nearest even once (bit for bit, padding zeros included), and the lookup
    in the bf16 volume == the lookup in that volume widened back to fp32 (same interpolation, fp32 output), bit for bit;
    against the oracle within bf16 rounding of the correlation values."""
    f1, f2 = _rand(1, 256, h, w, seed=61), _rand(1, 256, h, w, seed=62)
    v32, dims = _build_pyramid_gpu(ops, f1, f2, precision, presplit=True)
    v16, _ = _build_pyramid_gpu(ops, f1, f2, precision, presplit=True, vol_dtype=torch.bfloat16)
    for l in range(4):
        assert v16[l].dtype == torch.bfloat16
        assert torch.equal(v16[l], v32[l].to(torch.bfloat16)), f"level {l}"
    coords = raft_ref.coords_grid(1, h, w) + _rand(1, 2, h, w, seed=63, scale=6.0)
    coords[0, :, 0, 0] = torch.tensor([-7.3, 2.2])
    coords[0, :, 0, 1] = torch.tensor([w + 9.5, h + 3.0])
    cg = coords[0].permute(1, 2, 0).reshape(h * w, 2).contiguous().cuda()
    got, wide = torch.zeros(h * w, 352, device="cuda"), torch.zeros(h * w, 352, device="cuda")
    p16 = ops.make_lookup_params(v16, dims, cg, got, 4)
    assert p16.vol_bf16 == 1
    ops.run_lookup(p16)
    ops.run_lookup(ops.make_lookup_params([v.float() for v in v16], dims, cg, wide, 4))
    torch.cuda.synchronize()
    assert torch.equal(got, wide)
    assert float(got[:, 324:].abs().max()) == 0.0
    ref = raft_ref.corr_lookup(raft_ref.corr_pyramid(f1, f2), coords, 4)
    tol = 2.0 ** -8 * float(ref.abs().max()) + (5e-2 if precision == "bf16" else 2e-4)
    _close(got[:, :324].reshape(1, h, w, 324).permute(0, 3, 1, 2), ref, tol, what="lookup in the bf16 volume")
    with pytest.raises(AssertionError):     # mixed storage types are refused
        ops.make_lookup_params([v16[0], v32[1], v32[2], v32[3]], dims, cg, got, 4)


@pytest.mark.parametrize("h,w,precision,spread", [(17, 25, "bf16x3", 2.0), (24, 40, "bf16x3", 30.0), (16, 20, "bf16", 3.0),
                                                  (9, 11, "bf16x3", 200.0), (17, 25, "fp32", 2.0), (24, 40, "fp32", 30.0),
                                                  (9, 11, "fp32", 200.0),
                                                  # round 6: maps with interior blocks (boxes not clipped: windows not cleared, every cell
                                                  # written) next to clipped ones, a smooth field and a one-sided shift
                                                  (48, 72, "bf16x3", 0.3), (48, 72, "fp32", 0.3), (48, 72, "bf16", 0.3), (40, 64, "bf16x3", -11.0)])
def test_lookup_on_the_fly(ops, h, w, precision, spread):
    """Volume-free lookup (woft_corr_lookup_otf) == lookup in the volume built by the correlation GEMM in the same
    arithmetic, bit for bit (identical correlation values, identical interpolation), for smooth, scattered and
    far-out-of-map coordinates."""
    c = 256
    f1, f2 = _rand(1, c, h, w, seed=51), _rand(1, c, h, w, seed=52)
    vols, dims = _build_pyramid_gpu(ops, f1, f2, precision, presplit=precision != "fp32")     # (fp32: the fp32-MFMA GEMM)
    if spread < 0:              # a global shift (windows leave the map on one side) + a little noise
        coords = raft_ref.coords_grid(1, h, w) + spread + _rand(1, 2, h, w, seed=53, scale=0.4)
    else:
        coords = raft_ref.coords_grid(1, h, w) + _rand(1, 2, h, w, seed=53, scale=spread)
    coords[0, :, 0, 0] = torch.tensor([-7.3, 2.2])
    coords[0, :, h - 1, w - 1] = torch.tensor([w + 9.5, h + 3.0])
    cg = coords[0].permute(1, 2, 0).reshape(h * w, 2).contiguous().cuda()
    ref = torch.zeros(h * w, 352, device="cuda")
    ops.run_lookup(ops.make_lookup_params(vols, dims, cg, ref, 4))
    # the same operands, row-major and split
    x3 = precision == "bf16x3"
    a1, a2 = ops.act_from_nchw(f1), ops.act_from_nchw(f2)

    def split(t):
        if precision == "fp32":             # terms = 0: the fp32 feature rows themselves
            return t
        o = torch.zeros(t.shape[0], c * (2 if x3 else 1), dtype=torch.bfloat16, device="cuda")
        ops.split_bf16_lines(t, o) if x3 else ops.split_bf16(t, o, None)
        return o
    f2s, cur = [], a2
    for l in range(4):
        f2s.append(split(cur.t.contiguous()))
        if l < 3:
            nxt = ops.new_act(1, cur.h // 2, cur.w // 2, c)
            ops.avgpool2(cur, nxt)
            cur = nxt
    out = torch.full((h * w, 352), 7.0, device="cuda")
    terms = 3 if x3 else (0 if precision == "fp32" else 1)
    ops.run_lookup_otf(ops.make_lookup_otf_params(split(a1.t.contiguous()), f2s, dims, h, w, c, cg, out, 4, terms))
    torch.cuda.synchronize()
    assert torch.equal(out[:, :324], ref[:, :324]), "on-the-fly lookup differs from the lookup in the volume"
    assert float((out[:, 324:] - 7.0).abs().max()) == 0.0


@pytest.mark.parametrize("radius", [4, 3])
def test_lookup_golden_handmade(ops, golden_dir, radius):
    g = np.load(golden_dir / "lookup_handmade.npz")
    v = torch.from_numpy(g["vol0"])
    P = v.shape[0]
    pyr, vols, dims = [], [], []
    for l in range(4):
        pyr.append(v)
        vols.append(ops.tile_planes(v[:, 0]).cuda())
        dims.append(tuple(v.shape[-2:]))
        v = F.avg_pool2d(v, 2, stride=2)
    coords = torch.from_numpy(g["coords"])                       # (2, h1, w1)
    h1, w1 = coords.shape[1:]
    cg = coords.permute(1, 2, 0).reshape(P, 2).contiguous().cuda()
    nch = 4 * (2 * radius + 1) ** 2
    out = torch.zeros(P, nch, device="cuda")
    ops.run_lookup(ops.make_lookup_params(vols, dims, cg, out, radius))
    torch.cuda.synchronize()
    got = out.reshape(1, h1, w1, nch).permute(0, 3, 1, 2)
    if radius == 4:         # golden vector from the imported reference
        ref = torch.from_numpy(g["out"])
        assert abs(float(got[0, 1, 0, 0]) - 304.0) < 1e-3      # x-major window order pin
    else:                   # radius 3 (small model): the oracle's lookup on the same pyramid
        ref = raft_ref.corr_lookup(pyr, coords[None], 3)
    _close(got, ref, 1e-5 * float(ref.abs().max()), what="lookup golden")


def test_coords_update(ops):
    hf, wf = 7, 9
    c = torch.zeros(hf * wf, 2, device="cuda")
    f4 = torch.ones(hf * wf, 4, device="cuda")
    cat = torch.zeros(hf * wf, 16, device="cuda")
    ops.coords_init(c, hf, wf, f4, cat[:, 14:], 16)
    torch.cuda.synchronize()
    ref = raft_ref.coords_grid(1, hf, wf)[0].permute(1, 2, 0).reshape(-1, 2)
    assert torch.equal(c.cpu(), ref) and float(f4.abs().max()) == 0.0
    d = _rand(hf * wf, 64, seed=34).cuda()
    ops.coords_update(c, d, 64, wf, f4, cat[:, 14:], 16)
    torch.cuda.synchronize()
    assert torch.equal(c.cpu(), ref + d[:, :2].cpu())
    fl = (ref + d[:, :2].cpu()) - ref
    assert torch.equal(f4[:, :2].cpu(), fl) and torch.equal(cat[:, 14:16].cpu(), fl)
    assert float(cat[:, :14].abs().max()) == 0.0


# ------------------------------------------------------------------------------------------
# upsampling / boundary epilogue / warp
# ------------------------------------------------------------------------------------------
def test_convex_upsample(ops):
    hf, wf = 9, 11
    flow = _rand(1, 2, hf, wf, seed=35, scale=4.0)
    wl = _rand(1, 1, hf, wf, seed=36, scale=3.0)
    mask = _rand(1, 576, hf, wf, seed=37, scale=2.0)
    fu = raft_ref.convex_upsample(flow, mask)
    wu = raft_ref.convex_upsample(wl, mask) / 8
    coords = (raft_ref.coords_grid(1, hf, wf) + flow)[0].permute(1, 2, 0).reshape(-1, 2).contiguous().cuda()
    mk = mask[0].permute(1, 2, 0).reshape(-1, 576).contiguous().cuda()
    H, W = 8 * hf, 8 * wf
    f_o, d_o, w_o = torch.zeros(2, H, W, device="cuda"), torch.zeros(2, H * W, device="cuda"), torch.zeros(H * W, device="cuda")
    ops.convex_upsample(coords, wl.reshape(-1).cuda(), mk, hf, wf, (0, 0), H, W, f_o, d_o, w_o, do_sigmoid=True)
    torch.cuda.synchronize()
    _close(f_o, fu[0], 2e-5, what="flow_up")
    _close(w_o.reshape(H, W), torch.sigmoid(wu[0, 0]), 1e-6, what="sigmoid(w_up)")
    idx = torch.arange(H * W)
    src = torch.stack([idx % W, idx // W]).float()
    _close(d_o, src + fu[0].reshape(2, -1), 3e-5, what="dst coords")
    # cropped (un-padded) window, raw logits
    h, w, top, left = H - 5, W - 3, 2, 1
    f_c, w_c = torch.zeros(2, h, w, device="cuda"), torch.zeros(h * w, device="cuda")
    ops.convex_upsample(coords, wl.reshape(-1).cuda(), mk, hf, wf, (top, left), h, w, f_c, None, w_c, do_sigmoid=False)
    torch.cuda.synchronize()
    _close(f_c, fu[0, :, top:top + h, left:left + w], 2e-5, what="flow_up crop")
    _close(w_c.reshape(h, w), wu[0, 0, top:top + h, left:left + w], 2e-5, what="w_up crop")


def test_weights_at_points_and_needed_windows(ops):
    """The sparse weight head's helpers.  woft_convex_weights_at: the weight woft_convex_upsample writes at a pixel, computed
    for a list of pixels only -- bit for bit, with crop offsets, raw and sigmoid, count on the device.  woft_wh_needed: the
    window list rewritten to the windows under the 3x3 upsampling support of those pixels' 1/8-res cells (numpy reference)."""
    hf, wf = 9, 11
    flow = _rand(1, 2, hf, wf, seed=35, scale=4.0)
    wl = _rand(1, 1, hf, wf, seed=36, scale=3.0).reshape(-1).cuda()
    mask = _rand(1, 576, hf, wf, seed=37, scale=2.0)
    coords = (raft_ref.coords_grid(1, hf, wf) + flow)[0].permute(1, 2, 0).reshape(-1, 2).contiguous().cuda()
    mk = mask[0].permute(1, 2, 0).reshape(-1, 576).contiguous().cuda()
    rs = np.random.RandomState(3)
    for (top, left, h, w), sig in (((0, 0, 8 * hf, 8 * wf), True), ((2, 1, 8 * hf - 5, 8 * wf - 3), False)):
        w_full = torch.zeros(h * w, device="cuda")
        ops.convex_upsample(coords, wl, mk, hf, wf, (top, left), h, w, None, None, w_full, do_sigmoid=sig)
        n_max, n = 64, 50
        pts = np.stack([rs.randint(0, w, n_max), rs.randint(0, h, n_max)], 1).astype(np.float32)
        pts[0], pts[1] = (0, 0), (w - 1, h - 1)
        pts_d = torch.from_numpy(pts).cuda()
        count = torch.tensor([n], dtype=torch.int32, device="cuda")
        wsel = torch.full((n_max,), -7.0, device="cuda")
        ops.convex_weights_at(pts_d, count, n_max, wl, mk, hf, wf, (top, left), wsel, do_sigmoid=sig)
        torch.cuda.synchronize()
        idx = (pts[:n, 1] * w + pts[:n, 0]).astype(np.int64)
        assert torch.equal(wsel[:n].cpu(), w_full.cpu()[idx])
        assert float((wsel[n:] + 7.0).abs().max()) == 0.0                      # beyond the count: untouched
        # needed windows of a window list (here: every second 1/8-res pixel) for these points
        index = torch.arange(0, hf * wf, 2, dtype=torch.int32, device="cuda")
        bitmap = torch.full((hf * wf,), 5, dtype=torch.int32, device="cuda")    # (scratch: cleared by the call)
        dyn = torch.zeros_like(index)
        n_needed = torch.zeros(1, dtype=torch.int32, device="cuda")
        ops.wh_needed(pts_d, count, n_max, top, left, hf, wf, index, bitmap, dyn, n_needed)
        torch.cuda.synchronize()
        need = np.zeros((hf, wf), bool)
        for x, y in pts[:n]:
            cy, cx = (int(y) + top) >> 3, (int(x) + left) >> 3
            need[max(cy - 1, 0):cy + 2, max(cx - 1, 0):cx + 2] = True
        want = np.where(need.reshape(-1)[index.cpu().numpy()], index.cpu().numpy(), -1)
        assert np.array_equal(dyn.cpu().numpy(), want) and int(n_needed) == int((want >= 0).sum())
        assert np.array_equal(bitmap.cpu().numpy() != 0, need.reshape(-1))


def test_upflow8(ops):
    hf, wf = 6, 9
    flow = _rand(1, 2, hf, wf, seed=38, scale=4.0)
    wl = _rand(1, 1, hf, wf, seed=39, scale=3.0)
    coords = (raft_ref.coords_grid(1, hf, wf) + flow)[0].permute(1, 2, 0).reshape(-1, 2).contiguous().cuda()
    H, W = 8 * hf, 8 * wf
    f_o, w_o = torch.zeros(2, H, W, device="cuda"), torch.zeros(H * W, device="cuda")
    ops.upflow8(coords, wl.reshape(-1).cuda(), hf, wf, (0, 0), H, W, f_o, None, w_o)
    torch.cuda.synchronize()
    _close(f_o, raft_ref.upflow8(flow)[0], 3e-5, what="upflow8")
    _close(w_o.reshape(H, W), (raft_ref.upflow8(wl) / 8)[0, 0], 1e-5, what="upflow8 weights")


def test_warp(ops):
    rs = np.random.RandomState(1)
    img = rs.randint(0, 256, (60, 84, 3), dtype=np.uint8)
    Hm = np.array([[1.02, 0.05, 3.5], [-0.03, 0.97, -2.25], [1e-4, -2e-4, 1.0]])
    t = torch.from_numpy(img).cuda()
    out, valid = torch.zeros_like(t), torch.zeros(60, 84, dtype=torch.uint8, device="cuda")
    ops.warp_perspective_u8(t, Hm, out, valid)
    torch.cuda.synchronize()
    ref = tracker_ref.warp_linear_u8(img, Hm)
    d = np.abs(out.cpu().numpy().astype(np.int32) - ref.astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3, (d.max(), (d > 0).mean())
    refm = tracker_ref.warp_linear(np.ones((60, 84)), Hm) > 0
    assert (valid.cpu().numpy().astype(bool) != refm).mean() < 1e-3
    m = (rs.uniform(size=(60, 84)) > 0.5).astype(np.uint8) * 255
    mt = torch.from_numpy(m).cuda()
    mo = torch.zeros_like(mt)
    ops.warp_perspective_u8(mt, Hm, mo, None, nearest=True)
    torch.cuda.synchronize()
    assert (mo.cpu().numpy() != tracker_ref.warp_nearest(m, Hm)).mean() < 2e-3


# ------------------------------------------------------------------------------------------
# homography fit
# ------------------------------------------------------------------------------------------
def _corner_err(Ha, Hb):
    c = np.array([[100, 80, 1], [1800, 80, 1], [1800, 1000, 1], [100, 1000, 1.0]]).T
    pa, pb = Ha @ c, Hb @ c
    return np.abs(pa[:2] / pa[2] - pb[:2] / pb[2]).max()


@pytest.mark.parametrize("case", ["n4", "n500", "n4096"])
def test_hfit_vs_golden_and_oracle(ops, golden_dir, case):
    g = np.load(golden_dir / "hfit.npz")
    a, b, w = (torch.from_numpy(g[f"{case}_{k}"]) for k in "abw")
    pa, pb, pw = a[0].contiguous().cuda(), b[0].contiguous().cuda(), w[0].contiguous().cuda()
    Hd, st = torch.zeros(9, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")

    def run(**kw):
        ops.hfit(pa, pb, kw.pop("w", pw), Hd, st, **kw)
        torch.cuda.synchronize()
        assert int(st.item()) == 0
        return Hd.cpu().numpy().reshape(3, 3).astype(np.float64)

    tol = 0.05            # px at the corners of a 1700x920 box (H corners <= 0.05 px in fp32)
    assert _corner_err(run(), g[f"{case}_qr_w"][0].astype(np.float64)) < tol
    assert _corner_err(run(w=None), g[f"{case}_qr_now"][0].astype(np.float64)) < tol
    assert _corner_err(run(reweight=2, huber_k=2.0, n_irls=5), g[f"{case}_irls_huber2"][0].astype(np.float64)) < tol
    if case != "n4":
        assert _corner_err(run(reweight=1, n_irls=5), g[f"{case}_irls_l1"][0].astype(np.float64)) < 0.2
        assert _corner_err(run(reweight=2, huber_k=0.01, n_irls=5), g[f"{case}_irls_huber001"][0].astype(np.float64)) < 0.2
    # inlier fraction against the oracle's torch_proj_errors
    Hq = torch.from_numpy(g[f"{case}_qr_w"])
    e = hfit_ref.torch_proj_errors(Hq, a.permute(0, 2, 1), b.permute(0, 2, 1))
    fr = torch.zeros(1, device="cuda")
    ops.inlier_frac(pa, pb, Hq[0].reshape(9).contiguous().cuda(), fr, thr=5.0)
    torch.cuda.synchronize()
    assert abs(float(fr.item()) - float((e <= 5).float().mean())) < 2.0 / a.shape[1]


def test_hfit_too_few_points(ops):
    pa = torch.zeros(3, 2, device="cuda")
    Hd, st = torch.zeros(9, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.hfit(pa, pa, None, Hd, st)
    torch.cuda.synchronize()
    assert int(st.item()) == 1


def test_abi_rejects_bad_arguments(ops):
    from woft_amd import _lib
    import ctypes as C
    p = _lib.ConvParams()
    assert _lib.load().woft_conv2d(C.byref(p), None) == -1
    lp = _lib.LookupParams()
    assert _lib.load().woft_corr_lookup(C.byref(lp), None) == -1


# ------------------------------------------------------------------------------------------
# how far the "fp32-emulating" claim of the split-bf16 mode carries
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", ["large_activations", "cancelling_weights", "wide_dynamic_range", "tiny_values"])
def test_bf16x3_stress_against_fp64(ops, case):
    """bf16x3 = hi*hi + hi*lo + lo*hi with fp32 accumulation: every product carries ~2^-16 relative error (the lo*lo term
    is dropped, and hi/lo are 8-bit mantissas), so the error of a sum is bounded by ~2^-16 * sum |a_k b_k| -- like fp32's
    2^-24 * sum |a_k b_k| (accumulation order aside), with a 2^8 larger constant.  The test pins that bound where trained
    weights could hurt: activations x1e3, weights that cancel to 1e-3 of their magnitude, operands spanning
    eight decades, operands near the bf16 denormal range -- against an fp64 convolution, side by side with the exact-fp32
    MFMA mode."""
    g = torch.Generator().manual_seed(11)
    n, cin, cout, h, w = 1, 256, 128, 24, 32
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    if case == "large_activations":
        x = x * 1.0e3 + 2.0e3                               # large offset + spread: post-ReLU features of a hot layer
    elif case == "cancelling_weights":
        x = x.abs() + 5.0                                   # all-positive activations ...
        wt = wt - wt.mean(dim=(1, 2, 3), keepdim=True)       # ... against zero-sum filters: the output is the small residue
        wt = wt + 1e-3 * wt.abs().mean()
    elif case == "wide_dynamic_range":
        x = x * torch.pow(10.0, torch.randint(-4, 5, (1, cin, 1, 1), generator=g).float())
        wt = wt * torch.pow(10.0, -torch.randint(-4, 5, (1, cin, 1, 1), generator=g).float())
    elif case == "tiny_values":
        x, wt = x * 1e-18, wt * 1e-12                       # products ~1e-30: lo parts fall into bf16's subnormal range
    ref = F.conv2d(x.double(), wt.double(), padding=1)
    mag = F.conv2d(x.double().abs(), wt.double().abs(), padding=1)          # sum |a_k b_k| per output
    pc = ops.pack_conv(wt, torch.zeros(cout))
    xa = ops.act_from_nchw(x)
    errs = {}
    for prec in ("fp32", "bf16x3"):
        out = ops.conv2d(xa, pc, precision=prec)
        torch.cuda.synchronize()
        assert bool(torch.isfinite(out.t).all())
        errs[prec] = float(((out.nchw().double().cpu() - ref).abs() / mag).max())
    print(f"{case}: max |err| / sum|a b|   fp32 {errs['fp32']:.2e}   bf16x3 {errs['bf16x3']:.2e}")
    assert errs["fp32"] < 2.0 ** -18                        # fp32 accumulation of 2304 exact products (8 decades apart in one case)
    if case == "tiny_values":
        # the lo planes underflow (bf16 has fp32's exponent range but the residue x - hi is 2^-8 smaller): the mode
        # degrades gracefully to plain-bf16 accuracy there, it does not blow up
        assert errs["bf16x3"] < 2.0 ** -7
    else:
        assert errs["bf16x3"] < 2.0 ** -14, "split-bf16 products lost more than the dropped lo*lo term explains"
        # relative to the OUTPUT the error is amplified by the cancellation ratio, exactly as in fp32 (256 x smaller there)
        assert errs["bf16x3"] < 600 * max(errs["fp32"], 2.0 ** -26)


@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
@pytest.mark.parametrize("kh,kw,h,w", [(1, 5, 18, 22), (5, 1, 18, 22), (1, 5, 8, 9), (5, 1, 8, 9), (3, 3, 5, 7)])
def test_gru_convs_on_the_per_tap_kernel(ops, kh, kw, h, w, precision):
    """The GRU's two-source convs with their gate epilogues on the PER-TAP kernel (halo 0) -- the kernel they fall back to on
    feature maps too small for the pixel-tile kernels -- against torch fp32, and against the pixel-tile kernel where that one
    exists (18 x 22)."""
    E = ops._lib
    n = 1
    hprev = torch.tanh(_rand(n, 128, h, w, seed=4))
    xin = _rand(n, 128, h, w, seed=5)
    mk = lambda s: (_rand(128, 256, kh, kw, seed=s, scale=1 / math.sqrt(256 * kh * kw)), _rand(128, seed=s + 50, scale=0.1))
    (wz, bz), (wr, br), (wq, bq) = mk(6), mk(7), mk(8)
    pad = (kh // 2, kw // 2)
    hx = torch.cat([hprev, xin], 1)
    z = torch.sigmoid(F.conv2d(hx, wz, bz, padding=pad))
    r = torch.sigmoid(F.conv2d(hx, wr, br, padding=pad))
    q = torch.tanh(F.conv2d(torch.cat([r * hprev, xin], 1), wq, bq, padding=pad))
    ref = (1 - z) * hprev + z * q
    pzr = ops.pack_conv(torch.cat([wz, wr], 0), torch.cat([bz, br], 0), padding=pad)
    pq = ops.pack_conv(wq, bq, padding=pad)
    ha, xa = ops.act_from_nchw(hprev), ops.act_from_nchw(xin)
    outs = {}
    for halo in (0, 8) if (h >= 8 and w >= 16) else (0,):
        zb, rh, hn = (ops.new_act(n, h, w, 128, zero=True) for _ in range(3))
        a = ops.conv_params(ha, pzr, zb, x2=xa, c_split=128, epi=E.EPI_GRU_ZR, split=128, e0=ha, out1=rh, precision=precision,
                            halo=halo)
        b = ops.conv_params(rh, pq, hn, x2=xa, c_split=128, epi=E.EPI_GRU_Q, e0=ha, e1=zb, precision=precision, halo=halo)
        assert a.halo == halo and b.halo == halo
        ops.run_conv(a)
        ops.run_conv(b)
        torch.cuda.synchronize()
        outs[halo] = (zb.t.clone(), rh.t.clone(), hn.t.clone())
    tol = 8e-5 if precision == "bf16x3" else 3e-2
    _close(ops.Act(outs[0][0], n, h, w, 128).nchw(), z, tol, what="z")
    _close(ops.Act(outs[0][2], n, h, w, 128).nchw(), ref, tol, what="h")
    if 8 in outs:        # (different accumulation order -- tap-major vs chunk-major K steps: close, not bit-identical)
        for u, v_ in zip(outs[0], outs[8]):
            _close(u, v_, tol, what="per-tap vs pixel-tile kernel")


# ------------------------------------------------------------------------------------------
# precision "f16mx8": fp16 main term + two block-scaled fp8 cross terms (woft_conv_params.wgt_mx, round 4)
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kh,kw,cin,cout,h,w,tiles", [(3, 3, 256, 192, 24, 40, None), (3, 3, 128, 64, 17, 37, None), (1, 5, 128, 128, 24, 40, None),
                                                      (5, 1, 128, 128, 9, 16, None), (3, 3, 256, 126, 135, 240, None),
                                                      (3, 3, 128, 256, 24, 40, (128, 128)), (1, 5, 64, 128, 19, 37, None)])
def test_conv_f16mx8(ops, monkeypatch, kh, kw, cin, cout, h, w, tiles):
    """The two-pass fp32-emulating product on the register-streamed kernel against fp64: error scale of bf16x3 (measured 2.2-2.3 x
    its error on the matrix cores, tools/micro/mx_split_probe.hip) -- far below fp16's (67 x) and bf16's (530 x); ReLU-like
    activations with a wide per-pixel amplitude range (the block scales' job), zero rows (scale byte 0) and ragged tiles."""
    E = ops._lib
    monkeypatch.setattr(ops, "MX_LAYERS", "all")             # (the engine's default keeps f16mx8 to the layers it is faster on)
    g = torch.Generator().manual_seed(7)
    amp = torch.exp(torch.rand(1, 1, h, w, generator=g) * 8 - 6)                    # per-pixel amplitude over 3.5 decades
    x = torch.relu(torch.randn(1, cin, h, w, generator=g)) * amp
    x[:, :, 2, 3] = 0.0                                                             # an all-zero pixel: block maximum 0
    wt = torch.randn(cout, cin, kh, kw, generator=g) / math.sqrt(cin * kh * kw)
    b = _rand(cout, seed=13, scale=0.1)
    ref = F.conv2d(x.double(), wt.double(), b.double(), padding=(kh // 2, kw // 2))
    norm = F.conv2d(x.double().abs(), wt.double().abs(), None, padding=(kh // 2, kw // 2)) + 1e-30
    pc = ops.pack_conv(wt, b, padding=(kh // 2, kw // 2))
    xa = ops.act_from_nchw(x)
    errs = {}
    for prec in ("bf16x3", "f16mx8", "fp16"):
        out = ops.new_act(1, h, w, cout, cs=ops._round_up(cout, 4), zero=True)
        p = ops.conv_params(xa, pc, out, precision=prec, tiles=tiles, halo=8 if tiles else None)
        assert p.halo in (8, 12) and p.precision == ops.PRECISION[prec]
        ops.run_conv(p)
        torch.cuda.synchronize()
        e = (out.nchw().double().cpu() - ref) / norm
        errs[prec] = float(torch.sqrt((e ** 2).mean()))
    print(f"{kh}x{kw} {cin}->{cout} @{h}x{w}: rms error / sum|a||w|  bf16x3 {errs['bf16x3']:.2e}  f16mx8 {errs['f16mx8']:.2e} "
          f"({errs['f16mx8'] / errs['bf16x3']:.1f} x)  fp16 {errs['fp16']:.2e}")
    assert errs["f16mx8"] < 6 * errs["bf16x3"] and errs["f16mx8"] < 0.2 * errs["fp16"]


def test_conv_f16mx8_tiny_blocks_stay_finite(ops, monkeypatch):
    """Round-4 advisor finding: a 32-channel block whose largest |a| is non-zero but below fp16's normal range (2^-14) has
    remainders a - fp16(a) of up to 2^-25 REGARDLESS of the block maximum; with the remainder's scale derived as (block scale - 11)
    and no floor, the scaled remainder exceeded e4m3's 448 and v_cvt_pk_fp8_f32 (which does not saturate) produced NaNs that the
    scaled MFMA spread.  The loader now floors the block scale at 105 (conv_regb_body.h): ReLU-like blocks of tiny values and
    zeros must give finite outputs with the usual error against fp64."""
    E = ops._lib
    monkeypatch.setattr(ops, "MX_LAYERS", "all")
    g = torch.Generator().manual_seed(3)
    h, w, cin, cout = 24, 40, 128, 128
    mag = torch.exp(torch.rand(1, cin, h, w, generator=g) * (math.log(2e-5) - math.log(1e-7)) + math.log(1e-7))   # 1e-7 .. 2e-5
    x = mag * (torch.rand(1, cin, h, w, generator=g) > 0.4)                        # ReLU-like: 40 % zeros
    x[:, :, 5:9] *= 1e3                                                            # some rows in fp16's normal range beside them
    x[:, :32, 0, 0] = 1e-30                                                        # a block far below every grid
    wt = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    ref = F.conv2d(x.double(), wt.double(), None, padding=1)
    norm = F.conv2d(x.double().abs(), wt.double().abs(), None, padding=1) + 1e-30
    pc = ops.pack_conv(wt, None)
    xa = ops.act_from_nchw(x)
    errs = {}
    for prec in ("bf16x3", "f16mx8"):
        out = ops.new_act(1, h, w, cout, zero=True)
        p = ops.conv_params(xa, pc, out, precision=prec)
        assert p.halo in (8, 12) and p.precision == ops.PRECISION[prec]
        ops.run_conv(p)
        torch.cuda.synchronize()
        o = out.nchw().double().cpu()
        assert bool(torch.isfinite(o).all()), prec
        e = (o - ref) / norm
        errs[prec] = float(torch.sqrt((e ** 2).mean()))
    print(f"tiny blocks: rms error / sum|a||w|  bf16x3 {errs['bf16x3']:.2e}  f16mx8 {errs['f16mx8']:.2e}")
    # (values below fp16's subnormal grid reach the product through the fp8 image of a alone: 2^-4 relative on those terms)
    assert errs["f16mx8"] < 0.05


def test_gru_half_step_f16mx8(ops, monkeypatch):
    """Two-source GRU convs with their gate epilogues in f16mx8 against the bf16x3 launches."""
    E = ops._lib
    monkeypatch.setattr(ops, "MX_LAYERS", "all")
    n, h, w, kh, kw = 1, 24, 40, 1, 5
    hprev = torch.tanh(_rand(n, 128, h, w, seed=4))
    xin = torch.relu(_rand(n, 128, h, w, seed=5))
    mk = lambda s: _rand(128, 256, kh, kw, seed=s, scale=1 / math.sqrt(256 * kh * kw))
    pzr = ops.pack_conv(torch.cat([mk(6), mk(7)], 0), None, padding=(0, 2))
    pq = ops.pack_conv(mk(8), None, padding=(0, 2))
    gz, gq = ops.act_from_nchw(_rand(n, 256, h, w, seed=9, scale=0.3)), ops.act_from_nchw(_rand(n, 128, h, w, seed=10, scale=0.3))
    ha, xa = ops.act_from_nchw(hprev), ops.act_from_nchw(xin)
    res = {}
    for prec in ("bf16x3", "f16mx8"):
        z, rh, hn = (ops.new_act(n, h, w, 128, zero=True) for _ in range(3))
        a = ops.conv_params(ha, pzr, z, x2=xa, c_split=128, epi=E.EPI_GRU_ZR, split=128, e0=ha, out1=rh, bias_map=gz, precision=prec)
        b = ops.conv_params(rh, pq, hn, x2=xa, c_split=128, epi=E.EPI_GRU_Q, e0=ha, e1=z, bias_map=gq, precision=prec)
        assert a.precision == ops.PRECISION[prec] and b.precision == ops.PRECISION[prec]
        ops.run_conv(a)
        ops.run_conv(b)
        torch.cuda.synchronize()
        res[prec] = (z.t.clone(), hn.t.clone())
    _close(res["f16mx8"][0], res["bf16x3"][0], 2e-5, what="z")
    _close(res["f16mx8"][1], res["bf16x3"][1], 3e-5, what="h")


