#!/bin/bash
# Sanitizer runs of everything that executes on the CPU (GPU AddressSanitizer is not available on the pool: host code only).
#   1. the oracle (test infrastructure) built with g++ -fsanitize=address,undefined, the whole `-m "not gpu"` suite through it (AVO_LIB_PATH);
#   2. the PRODUCT library's host code (island manager, shard bookkeeping, level-2 planners, the ABI layer) built with
#      hipcc -fsanitize=address -fno-gpu-sanitize, the CPU tests that call into it (AVN_LIB_PATH).
# Nothing is written into the tree; builds go to ${OUT:-/tmp/avn_sanitize}.  Last run: profiles/r06_sanitizers_cpu.txt.
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${OUT:-/tmp/avn_sanitize}
mkdir -p "$OUT/prod"
cd "$R/oracle"
g++ -O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=default -fsanitize=address,undefined -fno-omit-frame-pointer -shared -o "$OUT/liboracle.so" oracle_capi.cpp -pthread
cd "$R"
echo "== oracle under ASan + UBSan: pytest -m 'not gpu'"
AVO_LIB_PATH="$OUT/liboracle.so" LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 \
  UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0 python -m pytest tests/ -q -m "not gpu" -s -p no:cacheprovider > "$OUT/oracle_suite.log" 2>&1 || true
tail -1 "$OUT/oracle_suite.log"; echo "UBSan 'runtime error' lines: $(grep -c 'runtime error' "$OUT/oracle_suite.log" || true); ASan reports: $(grep -c 'ERROR: AddressSanitizer' "$OUT/oracle_suite.log" || true)"
cd "$R/avian_amd/csrc"
F="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -fsanitize=address -fno-gpu-sanitize -shared-libasan -fno-omit-frame-pointer -Wno-unused-result"
objs=""
for s in k_bodies.hip k_contacts.hip k_xpbd.hip k_broadphase.hip k_narrow.hip k_graph.hip k_islands.hip k_transfer.hip avn_world.hip avn_comm.cpp avn_level2.cpp avn_islands.cpp avn_shard.cpp avn_abi.cpp; do
  o="$OUT/prod/${s%.*}.o"; objs="$objs $o"
  case $s in *.cpp) /opt/rocm/bin/hipcc $F -x hip -c $s -o $o & ;; *) /opt/rocm/bin/hipcc $F -c $s -o $o & ;; esac
  while [ "$(jobs -r | wc -l)" -ge 4 ]; do sleep 1; done
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address -fno-gpu-sanitize -shared-libasan -o "$OUT/prod/libavian_mi355x.so" $objs -ldl -Wl,-rpath,/opt/rocm/lib
cd "$R"
echo "== product host code under ASan: the CPU tests that call into libavian_mi355x.so"
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
AVN_LIB_PATH="$OUT/prod/libavian_mi355x.so" LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 python -m pytest tests/test_abi_cpu.py tests/test_colouring_second_opinion.py \
  tests/test_islands_cpu.py tests/test_level2_cpu.py tests/test_shard_gloo.py tests/test_reference_fixtures.py tests/test_sharded_closed_loop_cpu.py -q -m "not gpu" -s -p no:cacheprovider > "$OUT/product_suite.log" 2>&1 || true
tail -1 "$OUT/product_suite.log"; echo "ASan reports: $(grep -c 'ERROR: AddressSanitizer' "$OUT/product_suite.log" || true)"
