#!/bin/bash
# kernel timeline of TWO consecutive settled closed-loop steps of cfg2 with avn_sleeping_enable (one of them splits the pile's island): start offset, duration, gap,
# queue.  usage: bash tools/sleeping_cfg2_timeline.sh [step=110]
R=$(cd $(dirname $0)/.. && pwd); O=$R/gpurun_out/slp_timeline; mkdir -p $O; STEP=${1:-110}
cat > $O/drive.py <<PY
import sys, numpy as np
sys.path.insert(0, "$R")
import avian_amd
from avian_amd import _ffi as F, scenes
lib = avian_amd.load_library(); sc = scenes.box_stack(50, 40, 50)
w = F.World(lib, F.default_config(32, substeps=4))
w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs()); w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=0.5)
w.pipeline_enable(); w.sleeping_enable()
for _ in range($((STEP+4))):
    w.step(); w.synchronize()
PY
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o p -- python $O/drive.py > $O/run.log 2>&1)
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python - "$f" $STEP > $O/timeline.txt <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
st = [i for i, r in enumerate(rows) if "k_update_aabb" in r["Kernel_Name"]]
s = int(sys.argv[2])
for step in (s, s + 1):
    sel = rows[st[step]:st[step + 1]]
    t0 = int(sel[0]["Start_Timestamp"]); last_end = t0
    run = None
    print(f"---- step {step} ----")
    for r in sel:
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        k = r["Kernel_Name"].split("(")[0].replace("void avn::", "").replace("avn::", "")[:48]
        q = r.get("Queue_Id", "?")
        if "k_color_pass" in k or "k_overflow_flow" in k:
            if run is None: run = [a, b, 1, k]
            else: run[1] = b; run[2] += 1
            last_end = max(last_end, b); continue
        if run: print(f"{(run[0]-t0)/1e3:9.1f} us  +{(run[1]-run[0])/1e3:8.1f} us           [{run[2]} colour / overflow launches]"); run = None
        print(f"{(a-t0)/1e3:9.1f} us  +{(b-a)/1e3:8.1f} us  gap {(a-last_end)/1e3:7.1f}  q{q} {k}")
        last_end = max(last_end, b)
    if run: print(f"{(run[0]-t0)/1e3:9.1f} us  +{(run[1]-run[0])/1e3:8.1f} us           [{run[2]} colour / overflow launches]")
    print(f"step span {(last_end-t0)/1e3:.1f} us")
PY
rm -rf $O/prof; cat $O/timeline.txt
