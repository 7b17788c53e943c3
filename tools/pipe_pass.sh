#!/bin/bash
# pipe occupancy of the wide-panel kernels: SQ counters in separate --pmc passes + a stats pass, `bench.py --workload $1`
set -u
export TMPDIR=/tmp
W=${1:-c5-shard}
O=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/pipe_$W
rm -rf $O; mkdir -p $O
ARGS="bench.py --workload $W --steps 30 --warmup 5 --no-cpu-baseline --no-c5-entry"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $ARGS > /dev/null 2> $O/stats.err
i=0
for C in "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES" "SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_WAVE_CYCLES SQ_WAVES"; do
  i=$((i + 1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc$i -- python3 $ARGS > /dev/null 2> $O/pmc$i.err
done
python3 tools/pipe_util.py $O/stats $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4 $O/pmc5 > $O/pipe_util.txt 2>&1
rm -rf $O/stats $O/pmc?
cat $O/pipe_util.txt
