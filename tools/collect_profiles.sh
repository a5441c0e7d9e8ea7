#!/bin/bash
# copy what is judged from gpurun_out/final/ (tools/final_measure.sh) into profiles/ under this round's prefix.  usage: bash tools/collect_profiles.sh r03
set -u
P=${1:?round prefix}; R=$(cd $(dirname $0)/.. && pwd); O=$R/gpurun_out/final; D=$R/profiles
cpf() { [ -s "$O/$1" ] && cp "$O/$1" "$D/${P}_$2"; }
cpf bench_cfg2.json bench_cfg2.json
cpf bench_under_rocprof.json bench_under_rocprof.json
cpf kernel_stats_cfg2.csv kernel_stats_cfg2.csv
cpf kernel_stats_cfg2_closed_loop.csv kernel_stats_cfg2_closed_loop.csv
cpf kernel_stats_cfg3.csv kernel_stats_cfg3.csv
cpf kernel_stats_cfg5.csv kernel_stats_cfg5.csv
cpf kernel_stats_scene_large.csv kernel_stats_scene_large_pyramid.csv
cpf kernel_stats_scene_many.csv kernel_stats_scene_many_pyramids.csv
cpf pmc_traffic.json pmc_traffic.json
cpf pmc_fetch_size_per_kernel.csv pmc_fetch_size_per_kernel.csv
cpf pmc_write_size_per_kernel.csv pmc_write_size_per_kernel.csv
cpf closed_loop_breakdown.txt closed_loop_breakdown.txt
cpf closed_loop_step110_timeline.txt closed_loop_step110_timeline.txt
cpf closed_loop_switches_ab.txt closed_loop_switches_ab.txt
cpf closed_loop_r4_vs_r5_same_box.txt closed_loop_r4_vs_r5_same_box.txt
cpf closed_loop_r5_vs_r6_same_box.txt closed_loop_r5_vs_r6_same_box.txt
cpf sleeping_cfg2_r5_vs_r6_same_box.txt sleeping_cfg2_r5_vs_r6_same_box.txt
cpf level2_cfg5_closed_loop_manifolds_world1.json level2_cfg5_closed_loop_manifolds_world1.json
cpf sleeping_step230_timeline.txt sleeping_step230_timeline.txt
cpf sleeping_cfg2_windows.txt sleeping_cfg2_windows.txt
cpf sleeping_cfg2_host_phases.txt sleeping_cfg2_host_phases.txt
cpf sleeping_cfg2_switches_ab.txt sleeping_cfg2_switches_ab.txt
cpf sleeping_cfg2_steps110_111_timeline.txt sleeping_cfg2_steps110_111_timeline.txt
cpf sleeping_many_pyramids_windows.txt sleeping_many_pyramids_windows.txt
cpf cfg3_closed_loop_joint_lds_ab.txt cfg3_closed_loop_joint_lds_ab.txt
cpf cfg5_closed_loop_windows.txt cfg5_closed_loop_windows.txt
cpf dshard_cost_model.txt dshard_cost_model.txt
cpf host_shapes_demo.txt host_shapes_demo.txt
cpf narrow_phase_cutoffs.txt narrow_phase_cutoffs.txt
cpf pmc_narrow_phase.json pmc_narrow_phase.json
cpf pmc_closed_loop_settled.json pmc_closed_loop_settled.json
cpf cfg5_percentiles.txt cfg5_percentiles.txt
cpf cfg3_island_streams_ab.txt cfg3_island_streams_ab.txt
cpf color_pass_floor.json color_pass_floor.json
cpf other_configs.json other_configs.json
cpf reference_scenes.json reference_scenes.json
cpf pcie_calls.json pcie_calls.json
cpf bench_n2_single_device.json bench_n2_one_device_gloo.json
cpf level2_world1.json level2_world1.json
ls -la $D | grep " ${P}_" | wc -l
