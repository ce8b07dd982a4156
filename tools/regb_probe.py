"""One update-block layer, one kernel variant, a few launches -- the target of PMC passes.
  python tools/regb_probe.py <case: zr|fh1|q> <halo: 0 (default choice) | 8> [tile_n]"""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from woft_amd import ops, _lib

case, halo = sys.argv[1], int(sys.argv[2])
tn = int(sys.argv[3]) if len(sys.argv) > 3 else None
prec = sys.argv[4] if len(sys.argv) > 4 else "bf16x3"
hf, wf = int(os.environ.get("HF", 135)), int(os.environ.get("WF", 240))
cin, x2c, cout, kh, kw = {"zr": (128, 128, 256, 1, 5), "q": (128, 128, 128, 5, 1), "fh1": (128, 0, 256, 3, 3),
                          "c2": (256, 0, 192, 3, 3), "c1": (352, 0, 256, 1, 1), "f1": (128, 0, 256, 1, 1), "e64": (64, 0, 64, 3, 3), "e96": (96, 0, 96, 3, 3)}[case]
wt = torch.randn(cout, cin + x2c, kh, kw) * 0.05
pc = ops.pack_conv(wt, torch.randn(cout) * 0.1, padding=(kh // 2, kw // 2))
x = ops.new_act(1, hf, wf, cin)
x.t.normal_()
x2 = None
if x2c:
    x2 = ops.new_act(1, hf, wf, x2c)
    x2.t.normal_()
out = ops.new_act(1, hf, wf, cout, cs=ops._round_up(cout, 4), zero=True)
stats = None
if os.environ.get("STATS"):          # encoder layer with InstanceNorm partial statistics (HF=540 WF=960 e64 8)
    stats = (torch.zeros(2 * 8192 * 128, device="cuda"), torch.zeros(2 * 8192 * 128, device="cuda"))
p = ops.conv_params(x, pc, out, x2=x2, c_split=cin if x2c else 0, epi=_lib.EPI_LINEAR if stats else _lib.EPI_RELU, precision=prec,
                    halo=halo or None, tiles=(128, tn) if tn else None, stats=stats)
import os
if os.environ.get("NOSTORE"):
    p.out_w = -12345
if os.environ.get("ABL"):            # conv_1x1.hip: 1 no activation loads, 2 no weight loads, 4 no stores
    p.out_w = -12350 - int(os.environ["ABL"])
stamps = None
if os.environ.get("STAMPS"):
    stamps = torch.zeros(4096 * 32, dtype=torch.int64, device="cuda")
    p.in_mean, p.in_rstd = 1, stamps.data_ptr()
_lib.load().woft_set_tuning(3, int(os.environ.get("DYN_LDS", "0")))
import time
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(12)]
for _ in range(3):
    ops.run_conv(p)
cold = torch.zeros(int(os.environ.get("COLD", "0")) * 2**18, device="cuda") if os.environ.get("COLD") else None   # MiB
for s_, e_ in ev:
    if cold is not None:
        cold.add_(1.0)          # evicts the L2s (weights stay in the 256-MiB memory-side cache, as inside a frame)
    s_.record(); ops.run_conv(p); e_.record()
torch.cuda.synchronize()
ts = sorted(s_.elapsed_time(e_) for s_, e_ in ev)
print(f"median {ts[6]*1e3:.1f} us  (grid {p._m_tiles * (p.cout_pad // p.tile_n)})")
for _ in range(0):
    ops.run_conv(p)
torch.cuda.synchronize()
print("done", p.halo, p.tile_n)

if stamps is not None:
    import numpy as np
    nb = p._m_tiles * (p.cout_pad // p.tile_n)
    st = stamps.cpu().numpy().reshape(-1, 32)[:nb]
    t0 = st[:, 0].min()
    start, pro, end = st[:, 0] - t0, st[:, 1] - st[:, 0], st[:, 15] - t0
    nch = p.cin_pad // 32
    chunks = np.diff(st[:, 1:1 + nch], axis=1)
    epi = st[:, 15] - st[:, 14]
    print(f"kernel span {end.max()} ticks; WG start min/med/max {start.min()}/{int(np.median(start))}/{start.max()}; "
          f"WG duration med {int(np.median(st[:,15]-st[:,0]))} max {int((st[:,15]-st[:,0]).max())}")
    print(f"  first chunk (incl. prologue) med {int(np.median(pro))}; later chunks med {int(np.median(chunks))} "
          f"p10 {int(np.percentile(chunks,10))} p90 {int(np.percentile(chunks,90))}; epilogue med {int(np.median(epi))}")
    print(f"  ticks per us: {end.max() / (ts[6]*1e3):.1f}")
    if prec == "f16mx8":              # (its chunk loop also stamps "MFMAs of chunk c issued" at [16 + c]: epilogue phases are not stamped)
        c = np.arange(1, min(nch, 8))
        issued = st[:, 16 + c] - st[:, c]
        tail = st[:, 1 + c] - st[:, 16 + c]
        print(f"  chunk = issue phase (loads, halo rows, MFMAs) med {int(np.median(issued))} p10 {int(np.percentile(issued, 10))} p90 "
              f"{int(np.percentile(issued, 90))} + tail (barrier) med {int(np.median(tail))} p10 {int(np.percentile(tail, 10))} p90 {int(np.percentile(tail, 90))}")
        raise SystemExit
    ep = st[:, 16:26]
    d = np.diff(np.concatenate([st[:, 14:15], ep], axis=1), axis=1)
    print("  epilogue phases (median cycles): entry->", [int(np.median(d[:, k])) for k in range(d.shape[1]) if ep[:, k].max() > 0])
    rt = (st[:, 31] - st[:, 30]).astype(np.float64)          # s_memrealtime: constant 100 MHz
    cyc = (st[:, 15] - st[:, 0]).astype(np.float64)
    ok = rt > 0
    print(f"  shader clock while the workgroups ran: {np.median(cyc[ok] / rt[ok]) * 100:.0f} MHz (s_memtime / s_memrealtime); "
          f"median workgroup {np.median(rt[ok]) / 100:.1f} us of a {ts[6]*1e3:.1f} us launch; "
          f"first start -> last end {(st[:,31].max() - st[:,30].min()) / 100:.1f} us")
