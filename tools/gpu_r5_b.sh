#!/bin/bash
# round 5, batch B: the full -m gpu suite on the tree (sorted handles, ADVICE fixes, new-pair ids / closed-loop changes), then the closed loop's windows
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5b; mkdir -p $O; cd $R; export TMPDIR=/tmp; exec </dev/null
timeout 1500 python -m pytest -x -q -m gpu -p no:cacheprovider tests > $O/tests.txt 2>&1
tail -15 $O/tests.txt
python tools/time_closed_loop.py 50 40 50 120 2>&1 | python tools/window_means.py > $O/closed_loop.txt; cat $O/closed_loop.txt
