#!/usr/bin/env python3
"""Per-colour launch durations of the contact passes from a rocprofv3 kernel trace (colour = position in the pass)."""
import csv, sys
from collections import defaultdict
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ncol = int(sys.argv[2])
per = defaultdict(lambda: defaultdict(list))
cnt = defaultdict(int)
for r in rows:
    k = r["Kernel_Name"]
    if "k_color_pass<float, " not in k: continue
    p = k.split("k_color_pass<float, ")[1][0]
    per[p][cnt[p] % ncol].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    cnt[p] += 1
for p in sorted(per):
    print("pass", p, " ".join(f"{sum(v)/len(v)/1e3:.1f}" for _, v in sorted(per[p].items())))
