#!/bin/bash
# development batch (round 4, second half): parity of the working tree, then A/B of the four-lanes-per-body warm start on the same box
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/dev4b; rm -rf $O; mkdir -p $O; cd $R; export TMPDIR=/tmp
exec < /dev/null
timeout 1000 python -m pytest tests/test_gpu_graph.py tests/test_gpu_closed_loop_configs.py tests/test_gpu_sleeping.py tests/test_gpu_despawn.py tests/test_gpu_pipeline_edges.py tests/test_gpu_reference_benches.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for k in 1 2; do
  echo "== closed loop, quad warm start, run $k"; timeout 120 python tools/time_closed_loop.py 50 40 50 120 2>&1 | python tools/window_means.py
  echo "== closed loop, AVN_WS_LANE_PER_BODY=1, run $k"; AVN_WS_LANE_PER_BODY=1 timeout 120 python tools/time_closed_loop.py 50 40 50 120 2>&1 | python tools/window_means.py
done
for s in many large; do echo "== reference scene $s: quad / lane"; timeout 100 python tools/profile_reference_scene.py $s 2>&1 | tail -1 | cut -c1-200; AVN_WS_LANE_PER_BODY=1 timeout 100 python tools/profile_reference_scene.py $s 2>&1 | tail -1 | cut -c1-200; done
bash tools/step_timeline.sh 110 > /dev/null 2>&1; cp $R/gpurun_out/timeline/timeline.txt $O/timeline110.txt; grep warm_start $O/timeline110.txt | head -2; tail -1 $O/timeline110.txt
