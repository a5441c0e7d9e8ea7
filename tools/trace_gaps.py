#!/usr/bin/env python3
"""Per-kernel duration and inter-kernel gap statistics from a rocprofv3 kernel_trace.csv (run on any box)."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = defaultdict(list); gap = defaultdict(list)
prev_end = None
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void avn::", "")[:48]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[k].append(e - s)
    if prev_end is not None:
        gap[k].append(s - prev_end)
    prev_end = e
print(f"{'kernel':50s} {'n':>6s} {'dur_avg':>8s} {'dur_min':>8s} {'gap_before_avg':>14s} {'gap_med':>8s}")
for k in sorted(dur, key=lambda k: -sum(dur[k])):
    d = dur[k]; g = sorted(gap.get(k, [0]))
    print(f"{k:50s} {len(d):6d} {sum(d)/len(d)/1e3:8.2f} {min(d)/1e3:8.2f} {sum(g)/len(g)/1e3:14.2f} {g[len(g)//2]/1e3:8.2f}")
