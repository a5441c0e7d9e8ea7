#!/bin/bash
# round 5, batch D: the oct kernel with DPP subtractions -- parity slice + A/B on one box
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5d; mkdir -p $O; cd $R; export TMPDIR=/tmp; exec </dev/null
M=$R/avian_amd/csrc/measure/libavian_mi355x.so
timeout 600 python -m pytest -x -q -m gpu -p no:cacheprovider tests/test_gpu_graph.py tests/test_gpu_reference_benches.py tests/test_gpu_parity.py tests/test_gpu_sharded_closed_loop.py -s > $O/tests.txt 2>&1
tail -4 $O/tests.txt; grep "replicated bookkeeping" $O/tests.txt
{
for k in 1 2; do
  echo "== oct (measure lib), run $k"; AVN_LIB_PATH=$M python tools/time_closed_loop.py 50 40 50 120 2>&1 | python tools/window_means.py
  echo "== lane form only (AVN_NO_OCT=1), run $k"; AVN_NO_OCT=1 AVN_LIB_PATH=$M python tools/time_closed_loop.py 50 40 50 120 2>&1 | python tools/window_means.py
done
} > $O/ab_oct.txt 2>&1
cat $O/ab_oct.txt
