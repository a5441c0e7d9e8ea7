#!/bin/bash
# round 6, final tree: the local-acceleration tests, smoke(), the default bench line (with its PMC side passes) and the rocprofv3 kernel stats of the same command
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/final_short; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp; cd $R; exec < /dev/null
timeout 600 python -m pytest -x -q -m gpu tests/test_gpu_local_accelerations.py tests/test_gpu_golden.py > $O/tests.txt 2>&1; tail -3 $O/tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 900 python bench.py > $O/bench_cfg2.json 2> $O/bench.err; tail -c 300 $O/bench_cfg2.json; echo
cp $R/gpurun_out/bench_pmc_traffic.json $O/pmc_traffic.json 2>/dev/null
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg2 -o p -- python $R/bench.py --no-cpu-baseline --no-pcie --no-closed-loop --no-traffic --no-iters8 > $O/prof_cfg2.log 2>&1)
cp $(find $O/prof_cfg2 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_cfg2.csv 2>/dev/null
grep "^{" $O/prof_cfg2.log | tail -1 > $O/bench_under_rocprof.json
rm -rf $O/prof_cfg2 $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
head -5 $O/kernel_stats_cfg2.csv
