#!/bin/bash
# MFMA utilisation of the product kernels at the wider / larger workloads (two rocprofv3 passes each: --pmc
# SQ_VALU_MFMA_BUSY_CYCLES with kernel-trace only, and --kernel-trace --stats); writes gpurun_out/mfma_workloads.txt
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/mfma_w
rm -rf $O; mkdir -p $O
: > $R/gpurun_out/mfma_workloads.txt
for w in c5-shard 8kly-scvi eccly-sisua; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_$w -- python3 bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2> $O/pmc_$w.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$w -- python3 bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2> $O/st_$w.err
  echo "## $w" >> $R/gpurun_out/mfma_workloads.txt
  python3 tools/mfma_util.py $O/pmc_$w $O/st_$w >> $R/gpurun_out/mfma_workloads.txt 2>&1
done
rm -rf $O
