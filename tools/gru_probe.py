"""One SepConvGRU half step at 1/8 of 1080p: the two-launch path (EPI_GRU_ZR, EPI_GRU_Q) vs woft_gru_halfstep, with the
per-workgroup phase timeline of the fused kernel (s_memtime stamps).  python tools/gru_probe.py [h|v] [precision]"""
import math
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from woft_amd import ops, _lib

kind = sys.argv[1] if len(sys.argv) > 1 else "h"
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16x3"
kh, kw = (1, 5) if kind == "h" else (5, 1)
hf, wf = 135, 240
pad = (kh // 2, kw // 2)
dyn = [(0, 128, 0), (256, 384, 128)]
wzr = torch.randn(256, 384, kh, kw) / math.sqrt(384 * 5)
wq = torch.randn(128, 384, kh, kw) / math.sqrt(384 * 5)
zr_dyn, q_dyn = ops.pack_conv(wzr, None, padding=pad, cin_layout=dyn), ops.pack_conv(wq, None, padding=pad, cin_layout=dyn)
ha = ops.new_act(1, hf, wf, 128); ha.t.normal_().tanh_()
xbuf = ops.new_act(1, hf, wf, 256); xbuf.t.normal_().relu_()
gz, gq = ops.new_act(1, hf, wf, 256), ops.new_act(1, hf, wf, 128)
gz.t.normal_(); gq.t.normal_()
zb, rh, h1, h2 = (ops.new_act(1, hf, wf, 128, zero=True) for _ in range(4))
mk = lambda out: (ops.conv_params(ha, zr_dyn, zb, x2=xbuf, x2_off=128, c_split=128, epi=_lib.EPI_GRU_ZR, split=128, e0=ha, out1=rh,
                                  bias_map=gz, precision=prec),
                  ops.conv_params(rh, q_dyn, out, x2=xbuf, x2_off=128, c_split=128, epi=_lib.EPI_GRU_Q, e0=ha, e1=zb, bias_map=gq,
                                  precision=prec))
pzr, pq2 = mk(h2)
_, pq1 = mk(h1)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for s_, e_ in ev:
        s_.record(); fn(); e_.record()
    torch.cuda.synchronize()
    return sorted(s_.elapsed_time(e_) for s_, e_ in ev)[n // 2] * 1e3


t2 = timeit(lambda: (ops.run_conv(pzr), ops.run_conv(pq2)))
t1 = timeit(lambda: ops.run_gru_halfstep(pzr, pq1))
torch.cuda.synchronize()
print(f"half step {kh}x{kw} {prec}: two launches {t2:.1f} us, one launch {t1:.1f} us, identical {bool(torch.equal(h1.t, h2.t))}")
nwg = math.ceil(hf / 8) * math.ceil(wf / 16)
stamps = torch.zeros(nwg * 16, dtype=torch.int64, device="cuda")
pq1.in_mean, pq1.in_rstd = 1, stamps.data_ptr()
ops.run_gru_halfstep(pzr, pq1)
torch.cuda.synchronize()
st = stamps.cpu().numpy().reshape(nwg, 16)
d = np.diff(st[:, :7], axis=1)
names = ["prologue", "phase 1 (z|r main loop)", "epilogue 1 (this wave)", "... wait for all waves", "phase 2 (q main loop)", "epilogue 2"]
for k, n in enumerate(names):
    print(f"  {n:28s} median {int(np.median(d[:, k])):7d} cycles  (p10 {int(np.percentile(d[:, k], 10))}, p90 {int(np.percentile(d[:, k], 90))})")
print(f"  workgroup total median {int(np.median(st[:, 6] - st[:, 0]))} cycles; start skew {int(st[:, 0].max() - st[:, 0].min())}; "
      f"first start -> last end {int(st[:, 6].max() - st[:, 0].min())} cycles = {t1:.1f} us -> {int((st[:, 6].max() - st[:, 0].min()) / t1)} cycles per us")
