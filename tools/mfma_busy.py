"""MFMA-busy fraction per kernel from ONE rocprofv3 counter pass
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -- python bench.py ...
busy = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (sum(GRBM_GUI_ACTIVE) / 8 XCDs * 1024 SIMDs): fraction of the cycles the chip
actually ran (DVFS-limited clock).  python tools/mfma_busy.py <counter_collection.csv> <precision tag> [> table.csv]"""
import csv
import sys
from collections import defaultdict

path, tag = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "bf16x3")
acc = defaultdict(lambda: defaultdict(float))
launches = defaultdict(set)
for r in csv.DictReader(open(path)):
    acc[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
    launches[r["Kernel_Name"]].add(r["Dispatch_Id"])
rows = []
for k, c in acc.items():
    busy, act = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0)
    if busy > 0 and act > 0:
        rows.append((k, len(launches[k]), act, busy / (act / 8.0 * 1024.0), busy))
tot_act = sum(r[2] for r in rows)
tot_busy = sum(r[4] for r in rows)
print("precision,kernel,launches,share_of_mfma_kernel_cycles,mfma_busy_fraction")
for k, n, act, frac, _ in sorted(rows, key=lambda r: -r[2]):
    print(f'{tag},"{k}",{n},{act / tot_act:.4f},{frac:.4f}')
print(f'{tag},"ALL MFMA KERNELS (time weighted)",{sum(r[1] for r in rows)},1.0000,{tot_busy / (tot_act / 8.0 * 1024.0):.4f}')
