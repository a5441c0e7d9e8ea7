#!/bin/bash
# rocprofv3 kernel stats of cfg5 (500 000 cuboids, f64) in the device closed loop, steps 0..12 (the collapse): where the 85 ms per step go
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/prof_cfg5; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
(cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o p -- python $R/tools/time_cfg5.py 0 > $O/run.log 2>&1)
cp $(find $O -name "*kernel_stats.csv" | head -1) $R/gpurun_out/kernel_stats_cfg5_closed_loop.csv
find $O -name "*.csv" -delete
head -25 $R/gpurun_out/kernel_stats_cfg5_closed_loop.csv | cut -c1-200
