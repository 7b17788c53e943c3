#!/bin/bash
# Round profiles (run on the GPU box: `gpurun -- bash tools/make_profiles.sh r03`): the default bench line, rocprofv3
# kernel-trace / stats of the SAME command (without the entries of the other width: rocprofv3 averages by kernel name), the
# roofline check of both against the profile, the same at the C5-shard width, PMC traffic passes for the likelihood kernels,
# MFMA utilisation at the wider workloads, the other workloads' bench lines + step timelines, the data-parallel overheads.
# Writes under gpurun_out/<tag>/ (copy the summaries to profiles/).
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r03}
O=$R/gpurun_out/$TAG
PART=${2:-all}   # a gpurun call is limited to 20 minutes: `make_profiles.sh r05 1`, then `make_profiles.sh r05 2` (outputs merge under gpurun_out/<tag>/)
[ "$PART" != 2 ] && rm -rf $O
mkdir -p $O   # (gpurun merges into an existing gpurun_out/: remove stale trace dirs locally before copying)
cd $R
if [ "$PART" != 2 ]; then
# ---- the headline: default flags as the driver runs them, then a long run ----
python3 bench.py --steps 20 --warmup 5 > $O/bench_driver_settings.json 2> $O/bench_driver_settings.err
python3 bench.py --steps 300 --warmup 30 > $O/bench.json 2> $O/bench.err
# ---- rocprofv3 of the same command at each width + the check that the roofline fractions follow from it ----
for w in 8kly c5-shard; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$w -- python3 bench.py --workload $w --steps 300 --warmup 30 --no-cpu-baseline --no-c5-entry > $O/bench_under_rocprof_$w.json 2> $O/rocprof_$w.err
  python3 tools/prof_summary.py $O/trace_$w > $O/kernel_stats_summary_$w.txt 2>&1
  cp $(find $O/trace_$w -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_$w.csv
  # (the timed steps of a wide panel run the heads' optimiser update as a background sweep on a second queue: their timeline, queue by queue)
  [ $w = c5-shard ] && python3 tools/sweep_timeline.py $O/trace_$w > $O/sweep_timeline_$w.txt 2>&1
  rm -rf $O/trace_$w
  python3 bench.py --workload $w --steps 300 --warmup 30 --no-cpu-baseline --no-c5-entry > $O/bench_$w.json 2> $O/bench_$w.err
  { echo "## $w: bench line of the profiled run itself vs the rocprofv3 summary of that run (the check)"; python3 tools/check_roofline.py $O/bench_under_rocprof_$w.json $O/rocprofv3_kernel_stats_$w.csv $w;
    echo "## $w: bench line of a run without the profiler vs the same summary (informational: the profiler serialises back-to-back dispatches)"; python3 tools/check_roofline.py $O/bench_$w.json $O/rocprofv3_kernel_stats_$w.csv $w --info; } > $O/check_roofline_$w.txt 2>&1
done
# ---- PMC traffic of the likelihood kernels (separate FETCH_SIZE / WRITE_SIZE passes, calibrated in the same passes) ----
./tools/pmc_pass.sh > $O/pmc_summary.txt 2>&1
cp gpurun_out/pmc/pmc_summary.json $O/ 2>/dev/null; rm -rf gpurun_out/pmc
./tools/pmc_pass.sh c5-shard > $O/pmc_summary_c5-shard.txt 2>&1
cp gpurun_out/pmc/pmc_summary.json $O/pmc_summary_c5-shard.json 2>/dev/null; rm -rf gpurun_out/pmc
fi
if [ "$PART" != 1 ]; then
# ---- pipe occupancy (SQ counters in separate passes) at the C5 width and of the fused output head by itself ----
./tools/pipe_pass.sh c5-shard > /dev/null 2>&1; cp gpurun_out/pipe_c5-shard/pipe_util.txt $O/pipe_occupancy_c5-shard.txt 2>/dev/null
./tools/hf_pipe.sh zinb > /dev/null 2>&1; cp gpurun_out/hfpipe_zinb/pipe_util.txt $O/pipe_occupancy_head_fused.txt 2>/dev/null
# ---- MFMA utilisation of the product kernels (separate PMC pass + a stats pass) at the benchmark size and at the C5 shard ----
: > $O/mfma_utilisation.txt
for w in 8kly c5-shard; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/mfma_$w -- python3 bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline --no-c5-entry > /dev/null 2> $O/mfma_$w.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/mfst_$w -- python3 bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline --no-c5-entry > /dev/null 2>> $O/mfma_$w.err
  echo "## $w" >> $O/mfma_utilisation.txt
  python3 tools/mfma_util.py $O/mfma_$w $O/mfst_$w >> $O/mfma_utilisation.txt 2>&1
  rm -rf $O/mfma_$w $O/mfst_$w
done
# ---- the other single-GPU workloads: bench line + one step's timeline ----
: > $O/workloads.txt
for w in 8kly-scvi eccly-sisua 8kly-2layer cortex-base; do
  python3 bench.py --workload $w --steps 300 --warmup 30 --no-cpu-baseline --no-c5-entry > $O/bench_$w.json 2> $O/bench_$w.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$w -- python3 bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline --no-c5-entry > /dev/null 2> $O/rocprof_$w.err
  echo "## $w" >> $O/workloads.txt
  python3 - $O/bench_$w.json >> $O/workloads.txt <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(f"{d['value']:.0f} cells/s, {1e3 * d['ms_per_step']:.1f} us per step (batch {d['config']['global_batch']}); kernel_us {d['kernel_us']}")
PY
  python3 tools/prof_summary.py $O/trace_$w | sed -n '/one step/,$p' >> $O/workloads.txt 2>&1
  rm -rf $O/trace_$w
done
# ---- BASELINE configs[4] at its real residency (1e6 x 20 000 generated on the device) and the storage formats ----
python3 bench.py --workload c5 --steps 300 --warmup 30 > $O/bench_c5_full.json 2> $O/bench_c5_full.err
for s in f32 u16 csr; do python3 bench.py --storage $s --steps 300 --warmup 30 --no-cpu-baseline --no-c5-entry 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$s', d['ms_per_step'], d['final_loss'])"; done > $O/storage_formats.txt
# ---- data-parallel overheads on one rank: RCCL vs the peer-to-peer exchange, one collective vs two buckets ----
{ python3 tools/dp_overhead.py 8kly; python3 tools/dp_overhead.py c5-shard; } > $O/dp_overhead.txt 2>&1
fi
ls -la $O
