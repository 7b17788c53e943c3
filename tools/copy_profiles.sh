#!/bin/bash
# gpurun_out/<tag>/ (tools/make_profiles.sh) -> profiles/<tag>_* (the committed summaries)
set -u
T=${1:-r04}
S=gpurun_out/$T
P=profiles
cp $S/bench.json $P/${T}_bench.json
cp $S/bench_driver_settings.json $P/${T}_bench_driver_settings.json
cp $S/bench_8kly.json $P/${T}_bench_8kly_no_c5_entries.json
cp $S/bench_under_rocprof_8kly.json $P/${T}_bench_under_rocprof.json
cp $S/bench_c5-shard.json $P/${T}_bench_c5shard.json
cp $S/bench_under_rocprof_c5-shard.json $P/${T}_bench_c5shard_under_rocprof.json
cp $S/bench_c5_full.json $P/${T}_bench_c5_full_residency.json
cat $S/check_roofline_8kly.txt $S/check_roofline_c5-shard.txt > $P/${T}_check_roofline.txt
cp $S/kernel_stats_summary_8kly.txt $P/${T}_kernel_stats_summary.txt
cp $S/kernel_stats_summary_c5-shard.txt $P/${T}_kernel_stats_summary_c5-shard.txt
cp $S/sweep_timeline_c5-shard.txt $P/${T}_c5shard_sweep_timeline.txt 2>/dev/null
cp $S/rocprofv3_kernel_stats_8kly.csv $P/${T}_rocprofv3_kernel_stats.csv
cp $S/rocprofv3_kernel_stats_c5-shard.csv $P/${T}_rocprofv3_kernel_stats_c5shard.csv
cp $S/pmc_summary.json $P/${T}_pmc_likelihood_kernels.json
cp $S/pmc_summary_c5-shard.json $P/${T}_pmc_c5shard_kernels.json 2>/dev/null
cp $S/pipe_occupancy_c5-shard.txt $P/${T}_pipe_occupancy_c5shard.txt 2>/dev/null
cp $S/pipe_occupancy_head_fused.txt $P/${T}_pipe_occupancy_head_fused.txt 2>/dev/null
cp $S/mfma_utilisation.txt $P/${T}_mfma_utilisation_workloads.txt
cp $S/workloads.txt $P/${T}_workloads.txt
cp $S/storage_formats.txt $P/${T}_storage_formats.txt
cp $S/dp_overhead.txt $P/${T}_dp_overhead_one_rank.txt
ls -la $P | grep ${T}_ | wc -l
