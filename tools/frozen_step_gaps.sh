#!/bin/bash
# where the main stream idles inside / between frozen-manifold cfg2 steps (rocprofv3 kernel trace of the bench): gaps >= 3 us of the busiest queue over three steps
R=$(cd $(dirname $0)/.. && pwd); O=$R/gpurun_out/frozen_gaps; mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o p -- python $R/bench.py --no-cpu-baseline --no-pcie --no-closed-loop --no-traffic --no-iters8 > $O/run.log 2>&1)
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python - "$f" > $O/gaps.txt <<'PY'
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
st = [i for i, r in enumerate(rows) if "k_update_aabb" in r["Kernel_Name"]]
a, b = st[20], st[23]
sel = rows[a:b]
q = collections.Counter(r["Queue_Id"] for r in sel).most_common(1)[0][0]
main = [r for r in sel if r["Queue_Id"] == q]
t0 = int(sel[0]["Start_Timestamp"]); last = None; tot = 0
print("steps 20..22: span per step %.1f us; main queue %s, %d kernels" % ((int(sel[-1]["End_Timestamp"]) - t0) / 3e3, q, len(main)))
for r in main:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if last is not None:
        g = (s - last[1]) / 1e3
        tot += max(g, 0)
        if g >= 3.0:
            print("%9.1f us  gap %6.1f us  after %-40s before %s" % ((s - t0) / 1e3, g, last[0][:40], r["Kernel_Name"].split("(")[0].replace("void avn::", "")[:40]))
    last = (r["Kernel_Name"].split("(")[0].replace("void avn::", ""), e)
print("sum of all gaps on the main queue: %.1f us over 3 steps; kernel time %.1f us" % (tot, sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in main) / 1e3))
PY
rm -rf $O/prof; cat $O/gaps.txt
