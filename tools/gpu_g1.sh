set -u
O=gpurun_out/g1; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_graph.py -x -q 2>&1 | tail -40 > $O/graph_tests.log; tail -12 $O/graph_tests.log
timeout 600 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_graph.py 2>&1 | tail -15 > $O/gpu_tests.log; tail -8 $O/gpu_tests.log
timeout 600 python tools/time_closed_loop.py 50 40 50 24 $O/closed_cfg2.json > $O/closed_cfg2.log 2>&1; tail -3 $O/closed_cfg2.log
