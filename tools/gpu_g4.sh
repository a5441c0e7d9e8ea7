set -u
O=gpurun_out/g4; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/gpu_tests.log; tail -12 $O/gpu_tests.log
( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2>&1 | tail -3
python - <<'PY'
import json
d=json.load(open('gpurun_out/g4/bench.json'))
for k in ('value','ms_per_step','roofline','solver_iterations_8','pcie_inclusive','closed_loop'):
    print(k, json.dumps(d.get(k))[:1500])
print('cpu', json.dumps(d.get('cpu_baseline'))[:600])
PY
tail -5 $O/bench.err
