"""BASELINE config 5: a 4K (3840 x 2160) pair.  P = 270 x 480 = 129 600 source pixels; level 0 of the all-pairs
volume is P x 130 560 fp32 = 1.69e10 elements (67.7 GB), i.e. FOUR times past the 2^31-element mark and past 2^32 as
well -- exactly where 32-bit indexing of the correlation GEMM's stores or the lookup's loads would wrap (the
reference's own CUDA extension uses 32-bit accessors, alt_cuda_corr/correlation_kernel.cu:20-23).  The CPU oracle
cannot run at this size (it would need ~180 GB and minutes), so parity is checked through size-independent
properties: spot entries against fp64 dot products at source pixels on both sides of the 2^31 / 2^32 element offsets,
lookup identities there, bit equality of the volume-free lookup with the lookup in the volume over the whole frame, and
a whole 4K flow (finite, deterministic, both correlation modes bit-identical)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HF, WF, C = 270, 480, 256
P = HF * WF


@pytest.fixture(scope="module")
def ops():
    from woft_amd import _lib, ops as o
    _lib.load()
    if torch.cuda.get_device_properties(0).total_memory < 200 * 2 ** 30:
        pytest.skip("the 4K volume needs the 288 GB of an MI355X")
    return o


@pytest.fixture(scope="module")
def volume4k(ops):
    """Level 0..3 of the 4K pyramid built by the correlation GEMM (split-bf16 operands) + the operands of the
    volume-free lookup; 90 GB, freed when the module's tests are done."""
    g = torch.Generator(device="cuda").manual_seed(11)
    f1 = torch.zeros(ops._round_up(P, 128), C, device="cuda")
    f1[:P] = torch.randn(P, C, device="cuda", generator=g) * 0.3
    f2 = ops.new_act(1, HF, WF, C)
    f2.t.copy_(torch.randn(P, C, device="cuda", generator=g) * 0.3)

    def split(t):
        o = torch.zeros(t.shape[0], 2 * C, dtype=torch.bfloat16, device="cuda")
        ops.split_bf16_lines(t.contiguous(), o)
        return o
    sa = split(f1)
    vols, dims, f2s, maps, cur = [], [], [], [], f2
    for l in range(4):
        h, w = cur.h, cur.w
        n = ops.tiled_dims(h, w)[2]
        rows = torch.zeros(ops._round_up(n, 128), C, device="cuda")
        ops.tile_rows(cur, rows)
        vol = torch.zeros(P, n, device="cuda")
        ops.corr_gemm_bf16(sa, split(rows), P, n, 1.0 / math.sqrt(C), vol, 3)
        vols.append(vol)
        dims.append((h, w))
        f2s.append(split(cur.t))
        maps.append(cur)
        if l < 3:
            nxt = ops.new_act(1, h // 2, w // 2, C)
            ops.avgpool2(cur, nxt)
            cur = nxt
    torch.cuda.synchronize()
    assert vols[0].numel() > 2 ** 33 and vols[0].numel() * 4 > 67e9
    yield f1[:P], maps, sa, f2s, vols, dims
    del vols, f2s, sa
    torch.cuda.empty_cache()


def _probe_pixels():
    """Source pixels whose level-0 planes straddle the 2^31 and 2^32 ELEMENT offsets and the 2^32 / 2^34 / 2^36 BYTE
    offsets, plus the first and last ones."""
    plane = 68 * 120 * 16                                   # floats per source pixel at level 0
    marks = [2 ** 31, 2 ** 32, 2 ** 33, 2 ** 30, 2 ** 34 // 4 * 4, (2 ** 36) // 4]
    ps = {0, 1, P - 1, P - 2, P // 2}
    for m in marks:
        p = m // plane
        ps.update(q for q in (p - 1, p, p + 1) if 0 <= q < P)
    return sorted(ps)


def test_volume_entries_beyond_2_31_elements(ops, volume4k):
    f1, maps, sa, f2s, vols, dims = volume4k
    rs = np.random.RandomState(0)
    ps = np.array(_probe_pixels() + rs.randint(0, P, 24).tolist())
    assert (ps.astype(np.int64) * vols[0].shape[1] > 2 ** 32).sum() >= 8
    pt = torch.from_numpy(ps).cuda()
    for l in (0, 1, 3):
        h, w = dims[l]
        planes = ops.untile_planes(vols[l][pt], h, w).reshape(len(ps), -1).double().cpu()      # (n, h*w)
        qs = rs.randint(0, h * w, (len(ps), 16))
        got = torch.gather(planes, 1, torch.from_numpy(qs))
        a = f1[pt].double().cpu()                                                           # (n, C)
        b = maps[l].t.double().cpu()[torch.from_numpy(qs)]                                   # (n, 16, C)
        ref = (a[:, None, :] * b).sum(-1) / 16.0
        assert float((got - ref).abs().max()) < 2e-4, l
        # the whole plane of the probe pixels, and its tile padding is exactly zero
        full = (a @ maps[l].t.double().cpu().t()) / 16.0
        assert float((planes - full).abs().max()) < 2e-4, l
        part = vols[l][pt]
        assert torch.equal(ops.tile_planes(ops.untile_planes(part, h, w)), part)
    # untouched neighbours: the last plane ends exactly at the end of the buffer, the first starts at 0
    assert bool(torch.isfinite(vols[0][-1]).all()) and float(vols[0][-1].abs().max()) > 0


def test_lookup_identities_beyond_2_31_elements(ops, volume4k):
    f1, maps, sa, f2s, vols, dims = volume4k
    idx = torch.arange(P, device="cuda")
    grid = torch.stack([idx % WF, idx // WF], 1).float()
    d = torch.tensor([3.0, -2.0], device="cuda")
    coords = (grid + d).contiguous()
    out = torch.zeros(P, 352, device="cuda")
    ops.run_lookup(ops.make_lookup_params(vols, dims, coords, out, 4))
    torch.cuda.synchronize()
    # centre tap of level 0 at integer coordinates = vol[p][p + d], for pixels on both sides of every offset mark
    sample = torch.from_numpy(np.array(_probe_pixels() + list(range(0, P, 1009)))).cuda()
    x, y = (sample % WF) + 3, (sample // WF) - 2
    ok = (x >= 0) & (x < WF) & (y >= 0) & (y < HF)
    planes = ops.untile_planes(vols[0][sample], HF, WF)
    ref = torch.where(ok, planes[torch.arange(len(sample)), y.clamp(0, HF - 1), x.clamp(0, WF - 1)], torch.zeros(()).cuda())
    assert float((out[sample, 40] - ref).abs().max()) == 0.0
    # a one-pixel shift of the query shifts the level-0 window by one tap, everywhere
    out2 = torch.zeros(P, 352, device="cuda")
    ops.run_lookup(ops.make_lookup_params(vols, dims, (coords + torch.tensor([1.0, 0.0], device="cuda")).contiguous(), out2, 4))
    torch.cuda.synchronize()
    assert torch.equal(out[:, :81].reshape(P, 9, 9)[:, 1:, :], out2[:, :81].reshape(P, 9, 9)[:, :-1, :])
    # half-pixel query = mean of the two integer queries (bilinearity), level 0
    ops.run_lookup(ops.make_lookup_params(vols, dims, (coords + torch.tensor([0.5, 0.0], device="cuda")).contiguous(), out2, 4))
    torch.cuda.synchronize()
    mid = out2[:, :81].reshape(P, 9, 9)[:, :-1, :]
    a = out[:, :81].reshape(P, 9, 9)
    assert float((mid - 0.5 * (a[:, :-1, :] + a[:, 1:, :])).abs().max()) < 1e-5


def test_volume_free_lookup_equals_volume_lookup_4k(ops, volume4k):
    """Bit equality over the whole 4K frame, incl. every source pixel whose plane lies beyond 2^31 / 2^32 elements:
    the volume-free kernel has no P^2 addressing at all, so equality pins the 64-bit addressing of GEMM and lookup."""
    f1, maps, sa, f2s, vols, dims = volume4k
    g = torch.Generator(device="cuda").manual_seed(5)
    idx = torch.arange(P, device="cuda")
    grid = torch.stack([idx % WF, idx // WF], 1).float()
    fields = {
        "smooth": grid * 1.01 + torch.tensor([2.3, -1.7], device="cuda"),
        "scattered": grid + (torch.rand(P, 2, device="cuda", generator=g) * 2 - 1) * 12.0,
        "borders": grid * 1.2 - torch.tensor([40.0, 24.0], device="cuda"),
    }
    for name, coords in fields.items():
        coords = coords.contiguous()
        ref = torch.zeros(P, 352, device="cuda")
        out = torch.zeros(P, 352, device="cuda")
        ops.run_lookup(ops.make_lookup_params(vols, dims, coords, ref, 4))
        ops.run_lookup_otf(ops.make_lookup_otf_params(sa, f2s, dims, HF, WF, C, coords, out, 4, 3))
        torch.cuda.synchronize()
        assert torch.equal(out, ref), f"{name}: volume-free lookup differs from the lookup in the volume"
        assert float(ref[P - 1].abs().max()) > 0 or name == "borders"


def test_exact_fp32_volume_free_lookup_4k(ops, volume4k):
    """The fp32-MFMA instantiation of the volume-free lookup (terms = 0: the fp32 feature rows themselves) at 4K feature size,
    where the exact-fp32 volume would be another 90 GB: centre taps at integer coordinates against fp64 dot products (fp32
    accumulation error only), the whole output within the split-bf16 lookup's 2^-16 class, shift identity."""
    f1, maps, sa, f2s, vols, dims = volume4k
    idx = torch.arange(P, device="cuda")
    grid = torch.stack([idx % WF, idx // WF], 1).float()
    coords = (grid + torch.tensor([3.0, -2.0], device="cuda")).contiguous()
    rows = [m.t.contiguous() for m in maps]
    out = torch.zeros(P, 352, device="cuda")
    ops.run_lookup_otf(ops.make_lookup_otf_params(f1, rows, dims, HF, WF, C, coords, out, 4, 0))
    ref3 = torch.zeros(P, 352, device="cuda")
    ops.run_lookup_otf(ops.make_lookup_otf_params(sa, f2s, dims, HF, WF, C, coords, ref3, 4, 3))
    torch.cuda.synchronize()
    sample = torch.from_numpy(np.array(list(range(0, P, 257)))).cuda()
    x, y = (sample % WF) + 3, (sample // WF) - 2
    ok = (x >= 0) & (x < WF) & (y >= 0) & (y < HF)
    q = (y.clamp(0, HF - 1) * WF + x.clamp(0, WF - 1))
    dot = (f1[sample].double() * rows[0][q].double()).sum(1) / math.sqrt(C)
    ref = torch.where(ok, dot, torch.zeros((), dtype=torch.float64, device="cuda"))
    assert float((out[sample, 40].double() - ref).abs().max()) < 2e-6
    assert float((out[:, :324] - ref3[:, :324]).abs().max()) < 2e-4          # (values of order 1: 2^-16 per product, 256 products)
    out2 = torch.zeros(P, 352, device="cuda")
    ops.run_lookup_otf(ops.make_lookup_otf_params(f1, rows, dims, HF, WF, C, (coords + torch.tensor([1.0, 0.0], device="cuda")).contiguous(),
                                                  out2, 4, 0))
    torch.cuda.synchronize()
    assert torch.equal(out[:, :81].reshape(P, 9, 9)[:, 1:, :], out2[:, :81].reshape(P, 9, 9)[:, :-1, :])


def test_whole_4k_flow_both_correlation_modes(ops, volume4k):
    """One full 4K compute_flow per correlation mode through the operator: finite, weights in [0, 1], exact int64
    grid, run-to-run bit-identical, and the two modes bit-identical to each other (104 GB volume vs none).
    (The module fixture's 90 GB stay allocated next to the engine's 104 GB: 288 GB of HBM hold both.)"""
    from woft_amd import synth
    from woft_amd.config import Config
    from woft_amd.flow_provider import RAFTWrapper
    H, W = 2160, 3840
    del volume4k
    torch.cuda.empty_cache()
    sd = synth.make_state_dict(seed=7)
    t = synth.make_template(H, W, seq_id=9)
    f = np.roll(t, (5, -9), axis=(0, 1)).copy()
    res = {}
    for corr in ("otf", "volume"):
        c = Config()
        c.of_class, c.raft_type, c.class_params = RAFTWrapper, "weighted", Config()
        c.class_params.small = False
        c.model, c.iters, c.padding_mode, c.precision, c.corr = sd, 2, "nopad", "bf16x3", corr
        fl = RAFTWrapper(c)
        s1, d1, w1 = fl.compute_flow(t, f, mode="TC", do_sigmoid=True)
        s2, d2, w2 = fl.compute_flow(t, f, mode="TC", do_sigmoid=True)
        torch.cuda.synchronize()
        assert torch.equal(d1, d2) and torch.equal(w1, w2)
        idx = torch.arange(H * W, device="cuda")
        assert torch.equal(s1[0], idx % W) and torch.equal(s1[1], idx // W) and s1.dtype == torch.int64
        assert bool(torch.isfinite(d1).all()) and float(w1.min()) >= 0.0 and float(w1.max()) <= 1.0
        res[corr] = (d1.cpu(), w1.cpu())
        del fl, d1, d2, w1, w2
        import gc
        gc.collect()
        torch.cuda.empty_cache()
    assert torch.equal(res["otf"][0], res["volume"][0]) and torch.equal(res["otf"][1], res["volume"][1])
