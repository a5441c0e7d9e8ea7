#!/bin/bash
# round 5, batch G: handle lists + constraint generation enqueued before the counters' read-back -- parity slice + A/B on one box
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5g; mkdir -p $O; cd $R; export TMPDIR=/tmp; exec </dev/null
M=$R/avian_amd/csrc/measure/libavian_mi355x.so
timeout 900 python -m pytest -x -q -m gpu -p no:cacheprovider tests/test_gpu_graph.py tests/test_gpu_pipeline.py tests/test_gpu_reference_benches.py > $O/tests.txt 2>&1
tail -4 $O/tests.txt
{
for k in 1 2; do
  echo "== early (measure lib), run $k"; AVN_LIB_PATH=$M python tools/time_closed_loop.py 50 40 50 120 2>&1 | python tools/window_means.py
  echo "== AVN_NO_EARLY_PREPARE=1, run $k"; AVN_NO_EARLY_PREPARE=1 AVN_LIB_PATH=$M python tools/time_closed_loop.py 50 40 50 120 2>&1 | python tools/window_means.py
done
} > $O/ab_early.txt 2>&1
cat $O/ab_early.txt
bash tools/step_timeline.sh 110 > /dev/null 2>&1; cp $R/gpurun_out/timeline/timeline.txt $O/step110_timeline.txt 2>/dev/null; grep -v "k_color_pass\|k_overflow\|k_body_warm\|k_integrate" $O/step110_timeline.txt | sed -n 28,70p
