"""The persistent update-block kernel (woft_update_pk, csrc/update_pk.hip): the register-streamed conv layers of a refinement
iteration as ONE launch of resident workgroups with in-launch tile hand-offs must be BIT-identical to the per-layer launches
(same tile code, same products, same order) -- at kernel level on dependent layer chains (ragged tiles, every tile shape, fresh
data every launch so that a stale read cannot hide), and at flow level in every precision that has the kernel."""
import math
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

pytestmark = pytest.mark.gpu

from woft_amd import synth  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    from woft_amd import ops as o
    return o


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _flow_config(sd, iters, precision):
    from woft_amd.config import Config
    from woft_amd.flow_provider import RAFTWrapper
    c = Config()
    c.of_class, c.raft_type = RAFTWrapper, "weighted"
    c.class_params = Config()
    c.class_params.small = c.class_params.mixed_precision = c.class_params.alternate_corr = False
    c.class_params.weight_head_structure = [(128, 3)] * 3
    c.model, c.iters, c.padding_mode, c.precision = sd, iters, "nopad", precision
    return c


def _chain(ops, h, w, precision):
    """A miniature update block on (h, w) feature pixels: 3x3 (two layers writing one tensor) -> 3x3 -> z|r 1x5 -> q 1x5 ->
    z|r 5x1 -> q 5x1 -> 3x3, with the GRU epilogues and their operands -> (layers, input tensors to refresh, outputs)."""
    E = ops._lib
    A = lambda c, s: ops.act_from_nchw(_rand(1, c, h, w, seed=s))
    mk = lambda co, ci, kh, kw, s: ops.pack_conv(_rand(co, ci, kh, kw, seed=s, scale=1 / math.sqrt(ci * kh * kw)), _rand(co, seed=s + 1, scale=0.1),
                                                 padding=(kh // 2, kw // 2))
    x0, x1, h0 = A(128, 1), A(256, 2), ops.act_from_nchw(torch.tanh(_rand(1, 128, h, w, seed=3)))
    gz = [A(256, 4), A(256, 5)]
    gq = [A(128, 6), A(128, 7)]
    cat = ops.new_act(1, h, w, 256, zero=True)
    mot = ops.new_act(1, h, w, 128, zero=True)
    z = [ops.new_act(1, h, w, 128, zero=True) for _ in range(2)]
    rh = [ops.new_act(1, h, w, 128, zero=True) for _ in range(2)]
    hA, hB = ops.new_act(1, h, w, 128, zero=True), ops.new_act(1, h, w, 128, zero=True)
    fin = ops.new_act(1, h, w, 256, zero=True)
    kw = dict(precision=precision)
    L = [ops.conv_params(x0, mk(64, 128, 3, 3, 10), cat, co_off=192, epi=E.EPI_RELU, **kw),
         ops.conv_params(x1, mk(192, 256, 3, 3, 12), cat, co_off=0, epi=E.EPI_RELU, **kw),
         ops.conv_params(cat, mk(128, 256, 3, 3, 14), mot, epi=E.EPI_RELU, **kw)]
    states = [h0, hA, hB]
    for k, (kh, kw_) in enumerate(((1, 5), (5, 1))):
        hi, ho = states[k], states[k + 1]
        pzr = ops.pack_conv(_rand(256, 256, kh, kw_, seed=20 + k, scale=1 / math.sqrt(256 * 5)), None, padding=(kh // 2, kw_ // 2))
        pq = ops.pack_conv(_rand(128, 256, kh, kw_, seed=30 + k, scale=1 / math.sqrt(256 * 5)), None, padding=(kh // 2, kw_ // 2))
        L.append(ops.conv_params(hi, pzr, z[k], x2=mot, c_split=128, epi=E.EPI_GRU_ZR, split=128, e0=hi, out1=rh[k], bias_map=gz[k], **kw))
        L.append(ops.conv_params(rh[k], pq, ho, x2=mot, c_split=128, epi=E.EPI_GRU_Q, e0=hi, e1=z[k], bias_map=gq[k], **kw))
    L.append(ops.conv_params(hB, mk(256, 128, 3, 3, 40), fin, epi=E.EPI_RELU, **kw))
    return L, (x0, x1, h0), (cat, mot, z[0], rh[0], hA, z[1], rh[1], hB, fin)


@pytest.mark.parametrize("precision", ["bf16x3"])      # (the kernel is instantiated for the shipped arithmetic only)
@pytest.mark.parametrize("h,w", [(135, 240), (17, 25), (40, 64), (9, 16), (8, 33)])
def test_persistent_launch_equals_layer_launches(ops, h, w, precision):
    L, ins, outs = _chain(ops, h, w, precision)
    assert all(p.halo in (8, 12) for p in L), [p.halo for p in L]
    table = ops.PkTable(L)
    assert table.n_items == sum(math.ceil(h / (4 if p.halo == 12 else 8)) * math.ceil(w / 16) * (p.cout_pad // p.tile_n) for p in L)
    for rep in range(6):
        for t in ins:                                  # fresh inputs every round: a stale tile would not match
            t.t.mul_(-0.7).add_(0.01 * rep)
        for p in L:
            ops.run_conv(p)
        torch.cuda.synchronize()
        want = [o.t.clone() for o in outs]
        for o in outs:
            o.t.fill_(float("nan"))
        table.run()
        torch.cuda.synchronize()
        assert table.status() == 0
        for k, (o, wnt) in enumerate(zip(outs, want)):
            assert torch.equal(o.t, wnt), f"round {rep}: output {k} differs ({int((o.t != wnt).sum())} values)"


def test_persistent_launch_back_to_back_and_under_load(ops):
    """40 launches in a row without a host sync (the state resets itself), with another stream keeping the memory system busy:
    the hand-off must not depend on timing or placement."""
    h, w = 72, 112
    L, ins, outs = _chain(ops, h, w, "bf16x3")
    table = ops.PkTable(L)
    for p in L:
        ops.run_conv(p)
    torch.cuda.synchronize()
    want = [o.t.clone() for o in outs]
    side = torch.cuda.Stream()
    big = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    stop = torch.cuda.Event()
    with torch.cuda.stream(side):
        for _ in range(60):
            big.add_(1)
    for rep in range(40):
        if rep % 8 == 0:
            for o in outs:
                o.t.fill_(float("nan"))
        table.run()
    stop.record()
    torch.cuda.synchronize()
    assert table.status() == 0
    for k, (o, wnt) in enumerate(zip(outs, want)):
        assert torch.equal(o.t, wnt), f"output {k}"


def test_table_checks(ops):
    """What the host refuses: a tensor written twice in one launch, a read before a later layer's write, foreign kernels."""
    E = ops._lib
    h, w = 24, 40
    x = ops.act_from_nchw(_rand(1, 128, h, w, seed=1))
    y = ops.new_act(1, h, w, 128, zero=True)
    pc = ops.pack_conv(_rand(128, 128, 3, 3, seed=2, scale=0.03), None)
    a = ops.conv_params(x, pc, y, epi=E.EPI_RELU, precision="bf16x3")
    b = ops.conv_params(y, pc, x, epi=E.EPI_RELU, precision="bf16x3")          # overwrites a's input
    with pytest.raises(AssertionError):
        ops.PkTable([a, b])
    with pytest.raises(AssertionError):
        ops.PkTable([a, ops.conv_params(x, pc, y, epi=E.EPI_RELU, precision="bf16x3")])      # y twice
    p1 = ops.pack_conv(_rand(128, 128, 1, 1, seed=3, scale=0.1), None)
    z = ops.new_act(1, h, w, 128, zero=True)
    with pytest.raises(ops._lib.WoftHipError):
        ops.PkTable([a, ops.conv_params(y, p1, z, precision="bf16x3")])        # 1x1: the per-tap kernel's layer
    with pytest.raises(ops._lib.WoftHipError):
        ops.PkTable([ops.conv_params(x, pc, y, epi=E.EPI_RELU, precision="fp32")])
    with pytest.raises(ops._lib.WoftHipError):
        ops.PkTable([ops.conv_params(x, pc, y, epi=E.EPI_RELU, precision="bf16")])     # (bf16x3 only)


@torch.no_grad()
@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
@pytest.mark.parametrize("h,w,iters", [(136, 200, 5), (72, 136, 1), (264, 392, 2)])
def test_flow_with_persistent_update_block_is_bit_identical(monkeypatch, precision, h, w, iters):
    from woft_amd import engine
    sd = synth.make_state_dict(seed=21)
    a = synth.make_template(h, w, seq_id=6)
    b = synth.make_frame(a, 3)
    outs = []
    for pk in ("1", "0"):
        monkeypatch.setattr(engine, "UPDATE_PK", pk)
        c = _flow_config(sd, iters, precision=precision)
        prov = c.of_class(c)
        flow, wts = prov.compute_flow(a, b, mode="flow")
        flow2, wts2 = prov.compute_flow(b, a, mode="flow")           # (second call on the same buffers)
        torch.cuda.synchronize()
        plan = prov.engine.plan(h, w)
        if pk == "1":
            progs = [v for v in plan._pk.values() if v is not None]
            if precision != "bf16x3":                    # no instance of the kernel: the per-layer launches ran
                assert not progs
            else:
                assert progs and all(ent[1].status() == 0 for pr in progs for ent in pr if ent[0] == "pk")
                assert sum(ent[0] == "pk" for ent in progs[0]) == 1 and len(progs[0]) == 3      # lookup, convc1 | convf1, the rest
        outs.append((flow.clone(), wts.clone(), flow2.clone(), wts2.clone()))
    for x, y in zip(*outs):
        assert torch.equal(x, y)
