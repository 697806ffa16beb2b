#!/usr/bin/env python3
"""Timeline of one MSM step from a rocprofv3 kernel trace (…_kernel_trace.csv): per kernel its start offset, duration and the
gap since the previous kernel ended -- where a step's device time goes that is not inside a kernel.
Usage: trace_gaps.py <kernel_trace.csv> [anchor kernel substring = k_coarse_hist] [step index from the end = 2]"""
import csv
import sys

path = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_coarse_hist"
back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if anchor in r[2]]
if len(starts) < back + 1:
    sys.exit("not enough steps in the trace")
lo, hi = starts[-back - 1], starts[-back]
t0 = rows[lo][0]
prev_end = None
busy = 0
print("%-58s %10s %9s %8s" % ("kernel", "start us", "dur us", "gap us"))
for s, e, name in rows[lo:hi]:
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%-58s %10.1f %9.1f %8.1f" % (name.replace("void mlhip::", "").replace("mlhip::", "")[:58], (s - t0) / 1e3, (e - s) / 1e3, gap))
    busy += e - s
    prev_end = max(prev_end, e) if prev_end is not None else e
span = (prev_end - t0) / 1e3
print("step: first kernel start to last kernel end %.1f us, sum of kernel durations %.1f us, next step starts %.1f us after this one's last kernel"
      % (span, busy / 1e3, (rows[hi][0] - prev_end) / 1e3))
