#!/usr/bin/env python3
"""Per-kernel device time of the closed loop's steps [first, last) from a rocprofv3 kernel trace of tools/time_closed_loop.py
(step boundaries = k_update_aabb launches).  usage: closed_loop_breakdown.py <kernel_trace.csv> [first=24] [top=30]"""
import collections
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    st = [i for i, r in enumerate(rows) if "k_update_aabb" in r["Kernel_Name"]]
    lo = st[first]
    n = len(st) - first
    agg = collections.defaultdict(lambda: [0, 0])
    for r in rows[lo:]:
        k = r["Kernel_Name"].split("(")[0].replace("void avn::", "").replace("avn::", "")
        agg[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); agg[k][1] += 1
    tot = sum(v[0] for v in agg.values())
    span = int(rows[-1]["End_Timestamp"]) - int(rows[lo]["Start_Timestamp"])
    print(f"steps {first}..{first + n - 1}: kernel sum {tot / n / 1e6:.3f} ms/step, span {span / n / 1e6:.3f} ms/step")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
        print(f"{v[0] / n / 1e3:9.1f} us/step {v[1] / n:7.1f} calls/step {v[0] / v[1] / 1e3:8.1f} us avg  {k[:100]}")


if __name__ == "__main__":
    main()
