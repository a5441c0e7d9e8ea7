#!/bin/bash
# every kernel (all queues) around one step boundary of the frozen-manifold cfg2 bench (rocprofv3 kernel trace)
R=$(cd $(dirname $0)/.. && pwd); O=$R/gpurun_out/frozen_gaps; mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o p -- python $R/tools/wall_per_step.py > $O/run2.log 2>&1)
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python - "$f" > $O/boundary.txt <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
st = [i for i, r in enumerate(rows) if "k_store_contact_impulses" in r["Kernel_Name"]]
i = st[40]
t0 = int(rows[i]["End_Timestamp"])
for r in rows[i - 3:i + 40]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  +%7.1f us  q%s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r["Queue_Id"], r["Kernel_Name"].split("(")[0].replace("void avn::", "")[:50]))
PY
rm -rf $O/prof; cat $O/boundary.txt; tail -3 $O/run2.log
