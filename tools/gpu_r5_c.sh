#!/bin/bash
# round 5, batch C: the eight-lanes-per-manifold colour pass -- closed-loop parity slice on it, A/B against the lane form on one box (measure build: AVN_NO_OCT),
# the reference scenes (launch-bound: small colours), kernel durations from a rocprofv3 trace; then the tests that batch B left
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5c; mkdir -p $O; cd $R; export TMPDIR=/tmp; exec </dev/null
M=$R/avian_amd/csrc/measure/libavian_mi355x.so
timeout 900 python -m pytest -x -q -m gpu -p no:cacheprovider tests/test_gpu_graph.py tests/test_gpu_despawn.py tests/test_gpu_reference_benches.py tests/test_gpu_parity.py tests/test_gpu_pipeline.py > $O/tests_oct.txt 2>&1
tail -4 $O/tests_oct.txt
{
for k in 1 2; do
  echo "== oct (measure lib), run $k"; AVN_LIB_PATH=$M python tools/time_closed_loop.py 50 40 50 120 2>&1 | python tools/window_means.py
  echo "== lane form only (AVN_NO_OCT=1), run $k"; AVN_NO_OCT=1 AVN_LIB_PATH=$M python tools/time_closed_loop.py 50 40 50 120 2>&1 | python tools/window_means.py
done
} > $O/ab_oct.txt 2>&1
cat $O/ab_oct.txt
bash tools/closed_loop_quick.sh r5c_oct > /dev/null 2>&1; cp $R/gpurun_out/quick_r5c_oct/breakdown.txt $O/breakdown_oct.txt; sed -n 38,62p $O/breakdown_oct.txt
for e in 0 1; do echo "== reference scenes, AVN_NO_OCT=$e"; if [ $e = 1 ]; then export AVN_NO_OCT=1; else unset AVN_NO_OCT; fi; AVN_LIB_PATH=$M timeout 300 python tools/bench_reference_scenes.py 200 2 $O/ref_scenes_nooct$e.json 2>&1 | tail -3; done > $O/ref_scenes.txt 2>&1
unset AVN_NO_OCT
cat $O/ref_scenes.txt
