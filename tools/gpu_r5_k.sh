#!/bin/bash
# round 5, batch K: what does ONE hop of the overflow colour's dataflow pass cost?  depth of the list after every step against the per-launch durations of a kernel trace
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5k; mkdir -p $O; cd $R; export TMPDIR=/tmp; exec </dev/null
python tools/tail_depth.py 48 per-step 2>&1 | grep "after step" > $O/depth.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o p -- python $R/tools/time_closed_loop.py 50 40 50 48 > $O/run.log 2>&1)
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python - "$f" $O/depth.txt > $O/hop_fit.txt <<'PY'
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
st = [i for i, r in enumerate(rows) if "k_update_aabb" in r["Kernel_Name"]]
depth = {}
for l in open(sys.argv[2]):
    m = re.match(r"after step (\d+): overflow colour (\d+) manifolds, depth (\d+)", l)
    depth[int(m.group(1))] = (int(m.group(2)), int(m.group(3)))
print("step  manifolds  depth  flow launches: mean us (bias | relax)   us per level")
xs, ys = [], []
for s in range(len(st)):
    sel = rows[st[s]:st[s + 1] if s + 1 < len(st) else len(rows)]
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in sel if "k_overflow_flow_tag<1>" in r["Kernel_Name"] or "k_overflow_flow_tag<2>" in r["Kernel_Name"]]
    if not d or s not in depth: continue
    n, dep = depth[s]
    mean = sum(d) / len(d)
    print(f"{s:4d} {n:9d} {dep:6d}   {mean:8.1f}   {mean / max(dep, 1):6.2f}")
    xs.append(dep); ys.append(mean)
import numpy as np
A = np.vstack([np.ones(len(xs)), xs]).T
c, *_ = np.linalg.lstsq(A, np.array(ys), rcond=None)
print(f"least squares over {len(xs)} steps: launch = {c[0]:.1f} us + {c[1]:.2f} us x depth")
PY
rm -rf $O/prof; cat $O/hop_fit.txt
