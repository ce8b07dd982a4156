"""Per-item timeline of the persistent update-block kernel (csrc/update_pk.hip) at 1/8 of a frame: for every (layer, tile) work
item the 100 MHz clock at its start, after its dependency wait and at its end, and the XCD it ran on.  -> per layer: items, mean
wait, mean tile time, first start / last end; for the launch: slot occupancy, time in waits, items per XCD.
  python tools/pk_timeline.py [H W] [precision]        (WOFT_PK_OPTIONS / WOFT_PK_PLAIN=1: ablations)"""
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
os.environ.setdefault("WOFT_UPDATE_PK", "1")
from woft_amd import engine, ops, synth  # noqa: E402
from woft_amd.config import Config  # noqa: E402
from woft_amd.flow_provider import RAFTWrapper  # noqa: E402

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1080, 1920)
prec = sys.argv[3] if len(sys.argv) > 3 else "bf16x3"
engine.UPDATE_PK = "1"
c = Config()
c.of_class, c.raft_type = RAFTWrapper, "weighted"
c.class_params = Config()
c.class_params.small = c.class_params.mixed_precision = c.class_params.alternate_corr = False
c.model, c.iters, c.padding_mode, c.precision = synth.make_state_dict(seed=7), 4, "nopad", prec
prov = c.of_class(c)
a = synth.make_template(H, W, seq_id=0)
b = synth.make_frame(a, 3)
prov.compute_flow(a, b, mode="flow")
torch.cuda.synchronize()
plan = prov.engine.plan(H, W)
prog = plan._pk[(False, 1, False)]
table = next(e[1] for e in prog if e[0] == "pk")
if os.environ.get("WOFT_PK_PLAIN") == "1":            # ablation: ordinary stores (NOT a valid hand-off: timing only)
    import ctypes as C
    for i in range(table.n):
        table.host[i].conv.out_w = -12347
    table.dev.copy_(torch.frombuffer(bytearray(bytes(table.host)), dtype=torch.uint8))
tl = torch.zeros(table.n_items, 4, dtype=torch.int64, device="cuda")
ptr = tl.data_ptr()
for rep in range(3):
    table.state[4], table.state[5] = ptr & 0xffffffff if (ptr & 0xffffffff) < 2 ** 31 else (ptr & 0xffffffff) - 2 ** 32, ptr >> 32
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    table.run()
    e.record()
    torch.cuda.synchronize()
table.state[4] = 0
table.state[5] = 0
assert table.status() == 0
t = tl.cpu().numpy().astype(np.float64)
t0 = t[:, 0].min()
us = lambda x: x / 100.0                                # 100 MHz ticks -> us
print(f"{H}x{W} {prec}: launch {1e3 * s.elapsed_time(e):.1f} us (events), items {table.n_items}, "
      f"span {us(t[:, 2].max() - t0):.1f} us, options {table.options}, plain stores {os.environ.get('WOFT_PK_PLAIN') == '1'}")
names = ["convf2", "convc2", "convm", "zr1", "q1", "zr2", "q2", "fh1", "mk1"]
tot_tile = tot_wait = 0.0
for l in range(table.n):
    L = table.host[l]
    n = L.n_ty * L.n_tx * L.n_nt
    r = t[L.item0:L.item0 + n]
    wait, tile = us(r[:, 1] - r[:, 0]), us(r[:, 2] - r[:, 1])
    tot_tile += tile.sum()
    tot_wait += wait.sum()
    q = table.layers[l]
    print(f"  {names[l]:7s} {q.taps_y}x{q.taps_x} {q.cin_pad:3d}->{q.cout:3d} tile {L.ty}x16x{q.tile_n:3d} items {n:4d}: wait mean {wait.mean():6.2f} max {wait.max():6.1f} us "
          f"(>1 us: {100.0 * (wait > 1).mean():4.1f} %), tile mean {tile.mean():6.2f} p10 {np.percentile(tile, 10):6.2f} p90 {np.percentile(tile, 90):6.2f} us, "
          f"start {us(r[:, 0].min() - t0):6.1f} .. end {us(r[:, 2].max() - t0):6.1f} us")
span = us(t[:, 2].max() - t0)
nslots = 2 * torch.cuda.get_device_properties(0).multi_processor_count
print(f"  slot time: tiles {100 * tot_tile / (nslots * span):.1f} %, dependency waits {100 * tot_wait / (nslots * span):.1f} % of {nslots} slots x {span:.1f} us")
xcc = t[:, 3].astype(int)
print("  items per XCD:", np.bincount(xcc, minlength=8).tolist())
# the per-layer launches of the same layers, for comparison
for p in table.layers:
    ops.run_conv(p)
torch.cuda.synchronize()
for l, p in enumerate(table.layers):
    evs = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        ops.run_conv(p)
        e.record()
        evs.append((s, e))
    torch.cuda.synchronize()
    print(f"  launch {names[l]:7s}: {1e3 * np.median([s.elapsed_time(e) for s, e in evs]):6.1f} us")
