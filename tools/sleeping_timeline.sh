#!/bin/bash
# kernel timeline of ONE step of the sleeping scene (tools/time_sleeping.py) late in the run, when most islands sleep.  usage: bash tools/sleeping_timeline.sh [step=230]
R=$(cd $(dirname $0)/.. && pwd); O=$R/gpurun_out/sleeping_timeline; mkdir -p $O; STEP=${1:-230}
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o p -- python $R/tools/time_sleeping.py $(( (STEP + 60) / 50 )) < /dev/null > $O/run.log 2>&1)
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python - "$f" $STEP > $O/timeline.txt <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
st = [i for i, r in enumerate(rows) if "k_update_aabb" in r["Kernel_Name"]]
s = int(sys.argv[2])
sel = rows[st[s]:st[s + 1]]
t0 = int(sel[0]["Start_Timestamp"]); last_end = t0
for r in sel:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    k = r["Kernel_Name"].split("(")[0].replace("void avn::", "").replace("avn::", "")[:56]
    print(f"{(a-t0)/1e3:9.1f} us  +{(b-a)/1e3:8.1f} us  gap {(a-last_end)/1e3:7.1f}  q{r.get('Queue_Id','?')} {k}")
    last_end = max(last_end, b)
print(f"step span {(last_end-t0)/1e3:.1f} us, {len(sel)} launches")
PY
rm -rf $O/prof; cat $O/run.log | tail -8; cat $O/timeline.txt
