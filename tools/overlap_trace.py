#!/usr/bin/env python3
"""tools/overlap_trace.py <rocprofv3 output dir>: from a --kernel-trace of tools/bench_sharded_local.py, the last pipelined pass's
kernels as a timeline (start, duration, name) and how much of the wall time had two kernels running at once."""
import csv
import glob
import os
import sys

rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
short = lambda n: ("emit" if "lane_kernel<true, 1" in n else "apply" if "lane_kernel<true, 2" in n else "lookup" if "lookup_coop" in n
                   else "copy" if "list_copy" in n else "prefix" if "list_prefix" in n else None)
rows = [(a, b, short(n)) for a, b, n in rows if short(n)]
# the last 6 emits and everything from the first of them on
emits = [i for i, r in enumerate(rows) if r[2] == "emit"]
rows = rows[emits[-6]:]
t0 = rows[0][0]
for a, b, n in rows:
    print(f"{(a - t0) / 1e6:9.3f} ms  +{(b - a) / 1e6:7.3f}  {n}")
ev = sorted([(a, 1) for a, b, n in rows] + [(b, -1) for a, b, n in rows])
busy1 = busy2 = 0
depth, last = 0, ev[0][0]
for t, d in ev:
    if depth >= 1:
        busy1 += t - last
    if depth >= 2:
        busy2 += t - last
    depth += d
    last = t
print(f"wall {(rows[-1][1] - t0) / 1e6:.2f} ms, some kernel running {busy1 / 1e6:.2f} ms, two or more running {busy2 / 1e6:.2f} ms")
