"""Print the top kernels of a rocprofv3 *_kernel_stats.csv (share of GPU time, average launch)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
top = int(sys.argv[2]) if len(sys.argv) > 2 else 16
for r in rows[:top]:
    print(f"{r['Name'][:100]:100s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:9.1f} us "
          f"{float(r['TotalDurationNs']) / tot * 100:5.1f}%")
print(f"total {tot / 1e6:.1f} ms")
