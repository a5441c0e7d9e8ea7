#!/bin/bash
# round 5, batch A: the solver's body-sorted order inside colours 0..22 -- parity slice, A/B against the list order on ONE box, timeline, PMC traffic
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5a; mkdir -p $O; cd $R; export TMPDIR=/tmp; exec </dev/null
M=$R/avian_amd/csrc/measure/libavian_mi355x.so
timeout 1200 python -m pytest -x -q -m gpu -p no:cacheprovider tests/test_gpu_graph.py tests/test_gpu_closed_loop_configs.py tests/test_gpu_despawn.py tests/test_gpu_sleeping.py tests/test_gpu_pipeline.py tests/test_gpu_pipeline_edges.py tests/test_gpu_islands.py tests/test_gpu_reference_benches.py tests/test_gpu_configs_stepped.py > $O/tests.txt 2>&1
tail -3 $O/tests.txt
{
for k in 1 2; do
  echo "== sorted (measure lib), run $k"; AVN_LIB_PATH=$M python tools/time_closed_loop.py 50 40 50 120 2>&1 | python tools/window_means.py
  echo "== list order (AVN_NO_HANDLE_SORT=1), run $k"; AVN_NO_HANDLE_SORT=1 AVN_LIB_PATH=$M python tools/time_closed_loop.py 50 40 50 120 2>&1 | python tools/window_means.py
done
echo "== sorted + lane-per-body warm start"; AVN_WS_LANE_PER_BODY=1 AVN_LIB_PATH=$M python tools/time_closed_loop.py 50 40 50 120 2>&1 | python tools/window_means.py
echo "== product lib"; python tools/time_closed_loop.py 50 40 50 120 2>&1 | python tools/window_means.py
} > $O/ab.txt 2>&1
cat $O/ab.txt
bash tools/step_timeline.sh 110 > /dev/null 2>&1; cp $R/gpurun_out/timeline/timeline.txt $O/timeline_sorted.txt
bash tools/closed_loop_quick.sh r5a_sorted > /dev/null 2>&1; cp $R/gpurun_out/quick_r5a_sorted/breakdown.txt $O/breakdown_sorted.txt
AVN_NO_HANDLE_SORT=1 AVN_LIB_PATH=$M bash tools/closed_loop_quick.sh r5a_list > /dev/null 2>&1; cp $R/gpurun_out/quick_r5a_list/breakdown.txt $O/breakdown_list.txt
timeout 400 python tools/pmc_closed_loop_tail.py $O/pmc_closed_loop_settled.json 120 20 > $O/pmc.txt 2>&1
tail -16 $O/pmc.txt
sed -n 1,60p $O/timeline_sorted.txt
