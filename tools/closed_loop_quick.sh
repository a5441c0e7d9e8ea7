#!/bin/bash
# quick A/B aid: closed-loop cfg2 under rocprofv3, per-kernel breakdown of the steady window.  usage: bash tools/closed_loop_quick.sh <tag>
R=$(cd $(dirname $0)/.. && pwd); O=$R/gpurun_out/quick_$1; mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $R/tools/time_closed_loop.py 50 40 50 120 > $O/run.log 2>&1)
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python - "$f" > $O/breakdown.txt <<'PY'
import collections, csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
st = [i for i, r in enumerate(rows) if "k_update_aabb" in r["Kernel_Name"]]
for a, b in ((4, 24), (24, 44), (100, 120)):
    agg = collections.defaultdict(lambda: [0, 0])
    sel = rows[st[a]:st[b] if b < len(st) else len(rows)]
    for r in sel:
        k = r["Kernel_Name"].split("(")[0].replace("void avn::", "").replace("avn::", "")
        agg[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); agg[k][1] += 1
    n = b - a
    span = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / n / 1e6
    print(f"\n== steps {a}..{b - 1}: kernel sum {sum(v[0] for v in agg.values()) / n / 1e6:.3f} ms/step, span {span:.3f} ms/step")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:22]:
        print(f"{v[0] / n / 1e3:9.1f} us/step {v[1] / n:7.1f} calls/step {v[0] / v[1] / 1e3:8.1f} us avg  {k[:100]}")
PY
rm -rf $O/prof
tail -3 $O/run.log; cat $O/breakdown.txt
