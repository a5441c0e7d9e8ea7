#!/bin/bash
# The round's measurement batch (run on the GPU box from the repo root): the bench line (with its in-run PMC traffic pass), rocprofv3
# kernel summaries of the bench, of the closed loop, of cfg3 and cfg5 and of the reference's own scenes, the colour pass against its
# memory skeleton, the other configurations' step times, the reference-scene comparison, and smokes of the multi-rank paths on one
# device.  Everything lands in gpurun_out/final/; copy what is judged into profiles/ (tools/collect_profiles.sh).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd $R
exec < /dev/null   # (nothing here may ever wait on stdin: a tool that did cost round 4 twenty GPU-minutes)
M=$R/avian_amd/csrc/measure/libavian_mi355x.so   # the only build that reads AVN_* switches (round 5): every A/B below loads it through AVN_LIB_PATH
timeout 900 python bench.py > $O/bench_cfg2.json 2> $O/bench.err
tail -c 400 $O/bench_cfg2.json; echo
cp $R/gpurun_out/bench_pmc_traffic.json $O/pmc_traffic.json 2>/dev/null
for c in fetch write; do f=$(find $R/gpurun_out/pmc_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python - "$f" "$O/pmc_${c}_size_per_kernel.csv" <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    acc[(r["Kernel_Name"], r["Counter_Name"])][0] += float(r["Counter_Value"]); acc[(r["Kernel_Name"], r["Counter_Name"])][1] += 1
w = csv.writer(open(sys.argv[2], "w")); w.writerow(["kernel", "counter", "launches", "mean_value_kib"])
for (k, c), (s, n) in sorted(acc.items()): w.writerow([k, c, n, round(s / n, 3)])
PY
done
rm -rf $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
prof() {  # prof <tag> <cmd...>: rocprofv3 kernel-trace + stats, keep the stats csv only
  tag=$1; shift
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o p -- "$@" > $O/prof_$tag.log 2>&1)
  cp $(find $O/prof_$tag -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$tag.csv 2>/dev/null
}
prof cfg2 python $R/bench.py --no-cpu-baseline --no-pcie --no-closed-loop --no-traffic --no-iters8
grep "^{" $O/prof_cfg2.log | tail -1 > $O/bench_under_rocprof.json
prof cfg2_closed_loop python $R/tools/time_closed_loop.py 50 40 50 120
f=$(find $O/prof_cfg2_closed_loop -name "*kernel_trace.csv" | head -1)
python tools/closed_loop_breakdown.py $f 4 40 > $O/closed_loop_breakdown.txt 2>&1
python - "$f" >> $O/closed_loop_breakdown.txt <<'PY'
import collections, csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
st = [i for i, r in enumerate(rows) if "k_update_aabb" in r["Kernel_Name"]]
for a, b in ((4, 24), (24, 44), (100, 120)):
    agg = collections.defaultdict(lambda: [0, 0])
    for r in rows[st[a]:st[b] if b < len(st) else len(rows)]:
        k = r["Kernel_Name"].split("(")[0].replace("void avn::", "").replace("avn::", "")
        agg[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); agg[k][1] += 1
    n = b - a
    print(f"\n== steps {a}..{b - 1}: kernel sum {sum(v[0] for v in agg.values()) / n / 1e6:.3f} ms/step")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:16]:
        print(f"{v[0] / n / 1e3:9.1f} us/step {v[1] / n:7.1f} calls/step {v[0] / v[1] / 1e3:8.1f} us avg  {k[:100]}")
PY
prof cfg3 python $R/tools/profile_config.py cfg3 20
# cfg3 with and without the island streams (A/B on this box), the PCIe-inclusive step call by call
for e in 0 1; do AVN_LIB_PATH=$M AVN_NO_ISLAND_STREAMS=$e python $R/tools/profile_config.py cfg3 30 2>/dev/null | tail -1 | sed "s/^/island_streams_off=$e /"; done > $O/cfg3_island_streams_ab.txt
timeout 120 python tools/time_pcie.py 20 > $O/pcie_calls.json 2> $O/pcie.err
# the closed loop: one steady step's kernel timeline, the narrow phase's cut-off timings, A/B of the round's switches on this box
bash tools/step_timeline.sh 110 > /dev/null 2>&1; cp $R/gpurun_out/timeline/timeline.txt $O/closed_loop_step110_timeline.txt 2>/dev/null
for e in "" AVN_NO_OCT=1 AVN_OVF_TICKETS=1 AVN_NO_HANDLE_SORT=1 AVN_NO_EARLY_PREPARE=1 "AVN_NO_OCT=1 AVN_OVF_TICKETS=1 AVN_NO_HANDLE_SORT=1 AVN_NO_EARLY_PREPARE=1"; do echo "== ${e:-default}"; env AVN_LIB_PATH=$M $e python tools/time_closed_loop.py 50 40 50 120 2>&1 | python tools/window_means.py; done > $O/closed_loop_switches_ab.txt
# this tree against round 5's library ON THIS BOX (boxes of the pool differ by 10-15 %): avian_amd/csrc/ab/libavian_r5.so is the round-5 tree (git archive of the round-5 commit) built next to this one
if [ -f $R/avian_amd/csrc/ab/libavian_r5.so ]; then
  for k in 1 2; do
    echo "== round 6 (this tree), run $k"; python tools/time_closed_loop.py 50 40 50 120 2>&1 | python tools/window_means.py
    echo "== round 5 library, run $k"; AVN_AB_OLDER_LIBRARY=1 AVN_LIB_PATH=$R/avian_amd/csrc/ab/libavian_r5.so python tools/time_closed_loop.py 50 40 50 120 2>&1 | python tools/window_means.py
  done > $O/closed_loop_r5_vs_r6_same_box.txt 2>&1
  ( echo "== round 6 (this tree), avn_sleeping_enable at cfg2's scale"; timeout 200 python tools/time_sleeping_cfg2.py 60 2>/dev/null | grep "steps\|ratio"
    echo "== round 5 library, same tool"; AVN_AB_OLDER_LIBRARY=1 AVN_LIB_PATH=$R/avian_amd/csrc/ab/libavian_r5.so timeout 300 python tools/time_sleeping_cfg2.py 60 2>&1 | grep "steps\|ratio\|rror" ) > $O/sleeping_cfg2_r5_vs_r6_same_box.txt 2>&1
fi
# round 6: avn_sleeping_enable at cfg2's scale -- window by window next to the sleeping-off run, the host phases of every step, each of the round's mechanisms switched off
# alone on this box, a two-step kernel timeline (a split step and its neighbour); the bench's sleeping scene; cfg3 / cfg5 closed loops; the sharded device loop's cost model
AVN_LIB_PATH=$M AVN_SLP_TRACE=1 timeout 200 python tools/time_sleeping_cfg2.py 140 > $O/sleeping_cfg2_windows.txt 2> $O/sleeping_cfg2_host_phases.txt
for e in "" AVN_SLP_NO_FAST=1 AVN_SLP_HOST_SPLIT=1 AVN_SLP_SYNC_SPLIT=1; do echo "== ${e:-default}"; env AVN_LIB_PATH=$M $e timeout 300 python tools/time_sleeping_cfg2.py 140 2>/dev/null | grep "sleeping=1 steps\|ratio"; done > $O/sleeping_cfg2_switches_ab.txt
bash tools/sleeping_cfg2_timeline.sh 110 > /dev/null 2>&1; cp $R/gpurun_out/slp_timeline/timeline.txt $O/sleeping_cfg2_steps110_111_timeline.txt 2>/dev/null
timeout 120 python tools/time_sleeping.py 9 > $O/sleeping_many_pyramids_windows.txt 2>&1
bash tools/sleeping_timeline.sh 230 > /dev/null 2>&1; cp $R/gpurun_out/sleeping_timeline/timeline.txt $O/sleeping_step230_timeline.txt 2>/dev/null
for e in 0 1; do AVN_LIB_PATH=$M $( [ $e = 1 ] && echo env AVN_NO_JOINT_LDS=1 ) timeout 200 python tools/time_cfg3.py 2>/dev/null | sed "s/^/joint_lds_off=$e /"; done > $O/cfg3_closed_loop_joint_lds_ab.txt
timeout 300 python tools/time_cfg5.py 1 > $O/cfg5_closed_loop_windows.txt 2>/dev/null
timeout 300 python tools/time_dshard.py 120 > $O/dshard_cost_model.txt 2>/dev/null
# round 6: host shapes -- a compiled host whose capsules exist only in its two callbacks, next to cfg2's box stack in the closed loop: cost per step with 1 000 / 10 000 of them
( cd $R/examples && make -s >/dev/null 2>&1; for c in 1000 10000; do timeout 120 ./host_shapes_demo 50 40 50 $c 140; done ) > $O/host_shapes_demo.txt 2>&1
timeout 300 python tools/pmc_closed_loop_tail.py $O/pmc_closed_loop_settled.json 120 20 > $O/pmc_closed_loop_settled.txt 2>&1
for s in many large; do prof scene_$s python $R/tools/profile_reference_scene.py $s; done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*domain_stats.csv" -delete; find $O -name "*agent_info.csv" -delete
for t in cfg2 cfg2_closed_loop cfg3 scene_many scene_large; do rm -rf $O/prof_$t; done
timeout 200 python tools/measure_floor.py > $O/color_pass_floor.json 2> $O/floor.err; tail -c 300 $O/color_pass_floor.json; echo
timeout 400 python tools/time_configs.py $O/other_configs.json > $O/time_configs.log 2>&1
timeout 400 python tools/bench_reference_scenes.py 300 4 $O/reference_scenes.json > $O/reference_scenes.log 2>&1; tail -2 $O/reference_scenes.log
AVN_BENCH_SINGLE_DEVICE=1 AVN_BENCH_DIST_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 > $O/bench_n2_single_device.json 2> $O/bench_n2.err; tail -c 300 $O/bench_n2_single_device.json; echo
timeout 200 python tools/level2_multi_gpu.py --steps 10 2> $O/level2.err | grep '^{' > $O/level2_world1.json; tail -c 300 $O/level2_world1.json; echo
# round 6: the level-2 leg on the closed loop's own manifolds (overflow colour included), one rank: a smoke of the leg bench.py --gpus N runs, and the unsplit step time of that set
timeout 300 python tools/level2_multi_gpu.py --dims 100 50 100 --bits 64 --substeps 8 --steps 3 --warmup 1 --closed-loop-steps 6 2>> $O/level2.err | grep '^{' > $O/level2_cfg5_closed_loop_manifolds_world1.json
ls -la $O
