#!/bin/bash
# round 5, batch E: overflow colour handed over through tags in the velocity records -- parity slice + A/B against the ticket form on one box
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5e; mkdir -p $O; cd $R; export TMPDIR=/tmp; exec </dev/null
M=$R/avian_amd/csrc/measure/libavian_mi355x.so
timeout 900 python -m pytest -x -q -m gpu -p no:cacheprovider tests/test_gpu_graph.py tests/test_gpu_pipeline.py tests/test_gpu_reference_benches.py tests/test_gpu_parity.py tests/test_gpu_closed_loop_configs.py > $O/tests.txt 2>&1
tail -4 $O/tests.txt
{
for k in 1 2; do
  echo "== tags (measure lib), run $k"; AVN_LIB_PATH=$M python tools/time_closed_loop.py 50 40 50 120 2>&1 | python tools/window_means.py
  echo "== tickets (AVN_OVF_TICKETS=1), run $k"; AVN_OVF_TICKETS=1 AVN_LIB_PATH=$M python tools/time_closed_loop.py 50 40 50 120 2>&1 | python tools/window_means.py
done
} > $O/ab_tags.txt 2>&1
cat $O/ab_tags.txt
