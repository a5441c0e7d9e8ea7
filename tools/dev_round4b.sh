#!/bin/bash
# development batch (round 4, second half): parity of the working tree, then A/B against the committed HEAD's library on the same box
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/dev4b; rm -rf $O; mkdir -p $O; cd $R; export TMPDIR=/tmp
exec < /dev/null
timeout 900 python -m pytest tests/test_gpu_narrow.py tests/test_gpu_graph.py tests/test_gpu_closed_loop_configs.py tests/test_gpu_pipeline.py tests/test_gpu_pipeline_edges.py tests/test_gpu_despawn.py tests/test_gpu_sleeping.py tests/test_gpu_reference_benches.py tests/test_gpu_sharded_closed_loop.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
H=$R/avian_amd/csrc/ab/libavian_head.so
for k in 1 2; do
  echo "== closed loop, working tree, run $k"; timeout 120 python tools/time_closed_loop.py 50 40 50 120 2>&1 | tee $O/cl_tree_$k.log | python tools/window_means.py
  echo "== closed loop, HEAD library, run $k"; AVN_AB_OLDER_LIBRARY=1 AVN_LIB_PATH=$H timeout 120 python tools/time_closed_loop.py 50 40 50 120 2>&1 | python tools/window_means.py
done
for s in many large; do echo "== reference scene $s: tree / HEAD"; timeout 100 python tools/profile_reference_scene.py $s 2>&1 | tail -1 | cut -c1-200; AVN_AB_OLDER_LIBRARY=1 AVN_LIB_PATH=$H timeout 100 python tools/profile_reference_scene.py $s 2>&1 | tail -1 | cut -c1-200; done
timeout 300 python tools/pmc_closed_loop_tail.py $O/pmc_tail.json 120 20 2>&1 | head -8
bash tools/step_timeline.sh 110 > /dev/null 2>&1; cp $R/gpurun_out/timeline/timeline.txt $O/timeline110.txt
