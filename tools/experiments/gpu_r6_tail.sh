#!/bin/bash
# round 6: the tail colours as one dataflow launch -- parity slice, then a same-box A/B through the measure build (AVN_NO_TAIL_FLOW=1 = the colour launches)
# NOTE: the switch AVN_NO_TAIL_FLOW only exists with profiles/r06_tail_flow_not_kept.patch applied (git apply it on top of 7ea1b91, make + make measure): the experiment was not kept.
R=$(cd $(dirname $0)/../.. && pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_graph.py tests/test_gpu_overflow_stress.py tests/test_gpu_pipeline_edges.py "tests/test_gpu_closed_loop_configs.py::test_cfg2_closed_loop_to_the_steady_window_steps_100_119" > $O/tail_tests.txt 2>&1
tail -5 $O/tail_tests.txt
M=$R/avian_amd/csrc/measure/libavian_mi355x.so
{
for i in 1 2 3; do
  echo "== tail flow (default), run $i";   AVN_LIB_PATH=$M timeout 300 python tools/time_closed_loop.py 50 40 50 120 nosync 2>&1 | tail -3
  echo "== AVN_NO_TAIL_FLOW=1, run $i";     AVN_LIB_PATH=$M AVN_NO_TAIL_FLOW=1 timeout 300 python tools/time_closed_loop.py 50 40 50 120 nosync 2>&1 | tail -3
done
} > $O/tail_ab.txt 2>&1
cat $O/tail_ab.txt
