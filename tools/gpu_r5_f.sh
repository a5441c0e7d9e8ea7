#!/bin/bash
# round 5, batch F: tagged overflow flow -- polling interval sweep + per-kernel breakdown
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5f; mkdir -p $O; cd $R; export TMPDIR=/tmp; exec </dev/null
M=$R/avian_amd/csrc/measure/libavian_mi355x.so
{
for sl in 8 4 2 1 0 16; do
  echo "== AVN_OVF_POLL_SLEEP=$sl"; AVN_OVF_POLL_SLEEP=$sl AVN_LIB_PATH=$M python tools/time_closed_loop.py 50 40 50 120 2>&1 | python tools/window_means.py
done
} > $O/poll_sleep.txt 2>&1
cat $O/poll_sleep.txt
bash tools/closed_loop_quick.sh r5f_tags > /dev/null 2>&1; cp $R/gpurun_out/quick_r5f_tags/breakdown.txt $O/breakdown_tags.txt; sed -n 1,26p $O/breakdown_tags.txt
