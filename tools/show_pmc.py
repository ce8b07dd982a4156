"""Per-kernel means of the counters in rocprofv3 counter_collection CSVs: show_pmc.py <csv>... [--match substr]"""
import csv
import sys
from collections import defaultdict

files = [a for a in sys.argv[1:] if not a.startswith("--")]
match = None
if "--match" in sys.argv:
    match = sys.argv[sys.argv.index("--match") + 1]
    files = [f for f in files if f != match]
acc = defaultdict(lambda: defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if match and match not in name:
            continue
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in acc.items():
    print(name[:150])
    for c, v in sorted(cs.items()):
        print(f"    {c:32s} mean {sum(v) / len(v):16.1f}   (n={len(v)})")
