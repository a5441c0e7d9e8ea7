#!/bin/bash
# where the narrow phase's time goes: the cut-offs of AVN_NP_DEBUG applied to ONE step (AVN_NP_DEBUG_STEP) of the cfg2 closed loop, kernel durations from rocprofv3
R=$(cd $(dirname $0)/.. && pwd); O=$R/gpurun_out/np_phases; mkdir -p $O; STEP=${1:-60}
# AVN_NP_DEBUG / AVN_NP_DEBUG_STEP only exist in the measurement build of the library (-DAVN_MEASURE)
[ -f $R/avian_amd/csrc/measure/libavian_mi355x.so ] || make -C $R/avian_amd/csrc -j8 measure
export AVN_LIB_PATH=$R/avian_amd/csrc/measure/libavian_mi355x.so
for k in 0 1 2 3 4 5; do
  (cd /tmp && export TMPDIR=/tmp && AVN_NP_DEBUG=$k AVN_NP_DEBUG_STEP=$STEP timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/p$k -o p -- python $R/tools/time_closed_loop.py 50 40 50 $((STEP+2)) > $O/run$k.log 2>&1)
  f=$(find $O/p$k -name "*kernel_trace.csv" | head -1)
  python - "$f" $k $STEP <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
st = [i for i, r in enumerate(rows) if "k_update_aabb" in r["Kernel_Name"]]
step = int(sys.argv[3])
for s in (step - 1, step):
    sel = rows[st[s]:st[s + 1] if s + 1 < len(st) else len(rows)]
    out = []
    for r in sel:
        if "narrow_phase" in r["Kernel_Name"]:
            out.append(("heavy" if "heavy" in r["Kernel_Name"] else "light", round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1), r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
    print(f"np_debug={sys.argv[2]} step {s}: {out}")
PY
  rm -rf $O/p$k
done
