"""Ordered launch list of ONE steady-state tracked frame from a rocprofv3 kernel trace (tools/trace_gaps.sh):
start offset, duration and the gap to the previous dispatch's end, small helper launches (fills, copies, torch kernels) flagged.
    python tools/frame_sequence.py gpurun_out/<tag>_kernel_trace.csv [frame index from the end, default 2]"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:96]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
    # a frame starts at its first warp_kernel after an inlier_frac_kernel (the frame's last kernel)
    starts, armed = [], True
    for i, (s, e, n) in enumerate(ev):
        if "inlier_frac_kernel" in n:
            armed = True
        elif armed and "warp_kernel" in n:
            starts.append(i)
            armed = False
    if len(starts) < back + 1:
        print("not enough frames in the trace", len(starts))
        return
    a, b = starts[-back - 1], starts[-back]
    t0, prev_end, busy, small_t, small_n = ev[a][0], ev[a][0], 0, 0, 0
    for s, e, n in ev[a:b]:
        tiny = not any(k in n for k in ("conv_", "lookup", "gru", "hfit"))
        flag = "*" if (tiny and e - s < 20000) else " "
        if flag == "*":
            small_t += e - s
            small_n += 1
        print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f} {flag} {short(n)}")
        busy += e - s
        prev_end = max(prev_end, e)
    span = ev[b][0] - t0
    print(f"frame: {b - a} dispatches, span {span / 1e3:.1f} us, sum of kernel times {busy / 1e3:.1f} us, "
          f"{small_n} small helper dispatches = {small_t / 1e3:.1f} us")


if __name__ == "__main__":
    main()
