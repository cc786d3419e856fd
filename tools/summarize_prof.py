#!/usr/bin/env python3
"""Condense a tools/profile.sh output directory: per-kernel stats from the kernel trace and per-kernel PMC sums."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats:", os.path.relpath(f, out))
    rows = list(csv.DictReader(open(f)))
    for r in rows[:12]:
        print("  {:<70.70} calls {:>6} total_ms {:>12.3f} avg_ms {:>10.3f} pct {:>6}".format(
            r.get("Name", ""), r.get("Calls", ""), float(r.get("TotalDurationNs", 0)) / 1e6,
            float(r.get("AverageNs", 0)) / 1e6, r.get("Percentage", "")))
for d in sorted(glob.glob(os.path.join(out, "pmc*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(float))
        cnt = defaultdict(int)
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "")
            acc[name][r.get("Counter_Name", "")] += float(r.get("Counter_Value", 0) or 0)
            cnt[(name, r.get("Counter_Name", ""))] += 1
        print("== pmc:", os.path.relpath(f, out))
        for name, ctrs in acc.items():
            if "lane" not in name and "fused" not in name and "segment" not in name and "probe" not in name and "scan" not in name and "classify" not in name:
                continue
            for c, v in sorted(ctrs.items()):
                n = cnt[(name, c)]
                print("  {:<60.60} {:<24} per-dispatch {:>18.1f}  (dispatches {})".format(name, c, v / n, n))
