#!/usr/bin/env python3
"""Development aid: prints the last N ms of a rocprofv3 kernel trace as a per-queue timeline (kernel, start, duration,
waves), to see how the kernels of several steps in flight overlap.  usage: trace_timeline.py <kernel_trace.csv> [ms] [ms before the end of the trace where the window ends]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
span = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], int(r["Queue_Id"]),
       int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) // 64) for r in rows if r["Kernel_Name"].startswith("k_")]
back = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
t_end = max(k[1] for k in ks) - int(back * 1e6)
t0 = t_end - int(span * 1e6)
sel = sorted(k for k in ks if k[0] >= t0 and k[0] < t_end)
for s, e, n, q, w in sel:
    if e - s < 20000:
        continue
    print("%8.3f  +%6.3f ms  q%-2d %-28s %6d waves" % ((s - t0) / 1e6, (e - s) / 1e6, q, n, w))
# busy integral: sum over kernels of duration * min(waves, 2048) / 2048  (2 waves per SIMD = full)
tot = sum((e - s) * min(w, 2048) / 2048.0 for s, e, n, q, w in sel)
print("wave-slot occupancy over the window: %.2f" % (tot / (t_end - t0)))
