#!/bin/bash
# Round profiles: rocprofv3 kernel-trace/stats of the default bench command, PMC traffic passes for the
# loss kernel, loss-kernel roofline at larger sizes.  Writes under gpurun_out/ (copy the summaries to profiles/).
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r02}
O=$R/gpurun_out/$TAG
rm -rf $O; mkdir -p $O   # (gpurun merges into an existing gpurun_out/: remove stale trace dirs locally before copying)
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof.err
python3 tools/prof_summary.py $O/trace > $O/kernel_stats_summary.txt 2>&1
python3 bench.py --steps 300 --warmup 30 > $O/bench.json 2> $O/bench.err
./tools/pmc_pass.sh > $O/pmc_summary.txt 2>&1
cp gpurun_out/pmc/pmc_summary.json $O/ 2>/dev/null
python3 tools/loss_roofline.py > $O/loss_roofline.txt 2>&1
# MFMA utilisation of the product kernels (separate PMC pass + the stats above)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/mfma -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2> $O/mfma.err
python3 tools/mfma_util.py $O/mfma $O/trace > $O/mfma_utilisation.txt 2>&1
rm -rf $O/mfma
python3 tools/divergence_event.py > $O/divergence_event.txt 2>&1
TRAJ_STEPS=400 python3 tools/divergence_trace.py > $O/divergence_trace.txt 2>&1
# the other single-GPU workloads (BASELINE configs C3 / C4 on one GPU, the C5 shard, the two-layer variant): bench line + one step's timeline
: > $O/workloads.txt
for w in 8kly-scvi eccly-sisua c5-shard 8kly-2layer cortex-base; do
  python3 bench.py --workload $w --steps 300 --warmup 30 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$w -- python3 bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2> $O/rocprof_$w.err
  echo "## $w" >> $O/workloads.txt
  python3 - $O/bench_$w.json >> $O/workloads.txt <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(f"{d['value']:.0f} cells/s, {1e3 * d['ms_per_step']:.1f} us per step (batch {d['config']['global_batch']}); kernel_us {d['kernel_us']}")
PY
  python3 tools/prof_summary.py $O/trace_$w | sed -n '/one step/,$p' >> $O/workloads.txt 2>&1
  rm -rf $O/trace_$w
done
rm -rf $O/trace/*/*kernel_trace.csv   # keep the stats, drop the bulky trace
ls -la $O
