#!/bin/bash
# HBM traffic of the loss kernel from PMC counters, as MI355X_MICROARCH.md prescribes: separate passes for
# FETCH_SIZE and WRITE_SIZE, kernel-trace only, plus a calibration launch of known byte count per access width.
# usage: pmc_pass.sh [workload]   (default 8kly; c5-shard: the fused output head of smx_headfused.hip and its neighbours)
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
W=${1:-8kly}
OUT=$R/gpurun_out/pmc
rm -rf $OUT; mkdir -p $OUT
[ -x $R/tools/ubench ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 $R/tools/ubench.hip -o $R/tools/ubench
for c in FETCH_SIZE WRITE_SIZE; do
  UBENCH_CALIB=1 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/calib_$c -- $R/tools/ubench > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/bench_$c -- python3 $R/bench.py --workload $W --steps 30 --warmup 5 --no-cpu-baseline --no-c5-entry > $OUT/bench_$c.json 2> $OUT/bench_$c.err
done
python3 $R/tools/pmc_summary.py $OUT
