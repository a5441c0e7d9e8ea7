#!/bin/bash
# kernel timeline of ONE closed-loop step of cfg2 (rocprofv3 kernel trace): start offset, duration, gap to the previous kernel's end.  usage: bash tools/step_timeline.sh [step=110]
R=$(cd $(dirname $0)/.. && pwd); O=$R/gpurun_out/timeline; mkdir -p $O; STEP=${1:-110}
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o p -- python $R/tools/time_closed_loop.py 50 40 50 $((STEP+3)) > $O/run.log 2>&1)
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python - "$f" $STEP > $O/timeline.txt <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
st = [i for i, r in enumerate(rows) if "k_update_aabb" in r["Kernel_Name"]]
s = int(sys.argv[2])
sel = rows[st[s]:st[s + 1]]
t0 = int(sel[0]["Start_Timestamp"]); last_end = t0
run = None
for r in sel:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    k = r["Kernel_Name"].split("(")[0].replace("void avn::", "").replace("avn::", "")[:48]
    q = r.get("Queue_Id", "?")
    if "k_color_pass" in k or "k_overflow_flow" in k:   # collapse the colour launches
        if run is None: run = [a, b, 1, k]
        else: run[1] = b; run[2] += 1
        last_end = max(last_end, b); continue
    if run: print(f"{(run[0]-t0)/1e3:9.1f} us  +{(run[1]-run[0])/1e3:8.1f} us           [{run[2]} colour / overflow launches]"); run = None
    g = r.get("Grid_Size_X") or r.get("Grid_Size") or "?"
    print(f"{(a-t0)/1e3:9.1f} us  +{(b-a)/1e3:8.1f} us  gap {(a-last_end)/1e3:7.1f}  q{q} {k}  [{g} threads]")
    last_end = max(last_end, b)
if run: print(f"{(run[0]-t0)/1e3:9.1f} us  +{(run[1]-run[0])/1e3:8.1f} us           [{run[2]} colour / overflow launches]")
print(f"step span {(last_end-t0)/1e3:.1f} us")
PY
rm -rf $O/prof; cat $O/timeline.txt
