#!/bin/bash
# The round's measurement batch (run on the GPU box from the repo root): PMC traffic, the bench line, the rocprofv3 kernel
# summaries of the bench and of the reference scenes, the reference-scene comparison, and an N = 2 smoke of the bench's
# multi-rank path on one device.  Everything lands in gpurun_out/final/; copy what is judged into profiles/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 400 python tools/pmc_traffic.py $O/pmc_traffic.json > $O/pmc.log 2>&1
cp $O/pmc_traffic.json $R/profiles/r01_pmc_traffic.json 2>/dev/null      # bench.py reads `traffic` from here
for c in fetch write; do f=$(find $R/gpurun_out/pmc_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python - "$f" "$O/pmc_${c}_size_per_kernel.csv" <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    acc[(r["Kernel_Name"], r["Counter_Name"])][0] += float(r["Counter_Value"]); acc[(r["Kernel_Name"], r["Counter_Name"])][1] += 1
w = csv.writer(open(sys.argv[2], "w")); w.writerow(["kernel", "counter", "launches", "mean_value_kib"])
for (k, c), (s, n) in sorted(acc.items()): w.writerow([k, c, n, round(s / n, 3)])
PY
done
timeout 600 python bench.py > $O/bench_cfg2.json 2> $O/bench.err
tail -c 600 $O/bench_cfg2.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o b -- python $R/bench.py --no-cpu-baseline --no-pcie --no-closed-loop > $O/bench_under_rocprof.json 2> $O/prof_bench.err)
cp $(find $O/prof_bench -name "*kernel_stats.csv" | head -1) $O/kernel_stats_cfg2.csv
for s in many large; do
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$s -o p -- python $R/tools/profile_reference_scene.py $s > $O/prof_$s.log 2>&1)
  cp $(find $O/prof_$s -name "*kernel_stats.csv" | head -1) $O/kernel_stats_scene_$s.csv
done
find $O -name "*kernel_trace.csv" -delete
timeout 300 python tools/bench_reference_scenes.py 300 4 $O/reference_scenes.json > $O/reference_scenes.log 2>&1; tail -2 $O/reference_scenes.log
AVN_BENCH_SINGLE_DEVICE=1 AVN_BENCH_DIST_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 > $O/bench_n2_single_device.json 2> $O/bench_n2.err; tail -c 400 $O/bench_n2_single_device.json
rm -rf $O/prof_bench $O/prof_many $O/prof_large
ls -la $O
