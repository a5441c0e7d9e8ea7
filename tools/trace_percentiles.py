#!/usr/bin/env python3
"""Per-kernel duration percentiles from a rocprofv3 kernel_trace.csv, and for the colour launches one row per grid size (= per colour):
the `--stats` average of k_color_pass is skewed by the launches that overlap the broad phase on its own stream, so the median per colour
is the figure to price against the roofline.   python tools/trace_percentiles.py trace.csv [substring-of-kernels-to-split-by-grid]"""
import csv
import sys
from collections import defaultdict


def pct(v, q):
    return v[min(len(v) - 1, int(q * len(v)))]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    split = sys.argv[2] if len(sys.argv) > 2 else "k_color_pass"
    dur = defaultdict(list)
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("void avn::", "")[:44]
        if split in k:
            k += f" grid={int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))}"
        dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print(f"{'kernel':64s} {'n':>6s} {'mean':>9s} {'p10':>9s} {'p50':>9s} {'p90':>9s} {'max':>10s}  (us)")
    for k in sorted(dur, key=lambda k: -sum(dur[k])):
        d = sorted(dur[k])
        print(f"{k:64s} {len(d):6d} {sum(d) / len(d) / 1e3:9.2f} {pct(d, .1) / 1e3:9.2f} {pct(d, .5) / 1e3:9.2f} {pct(d, .9) / 1e3:9.2f} {d[-1] / 1e3:10.2f}")


if __name__ == "__main__":
    main()
