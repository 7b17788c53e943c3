#!/bin/bash
# pipe occupancy of the fused output head by itself: SQ counters in separate --pmc passes + a stats pass over tools/headfused_try.py
set -u
export TMPDIR=/tmp
LK=${1:-zinb}
O=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/hfpipe_$LK
rm -rf $O; mkdir -p $O
ARGS="tools/headfused_try.py --time-only $LK --reps 20"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $ARGS > $O/stats.out 2> $O/stats.err
i=0
for C in "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES" "SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_WAVE_CYCLES SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_WAIT_INST_LDS SQ_INSTS_VALU" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT" "SQ_INSTS_SALU SQ_INSTS_MFMA"; do
  i=$((i + 1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc$i -- python3 $ARGS > /dev/null 2> $O/pmc$i.err
done
python3 tools/pipe_util.py $O/stats $(ls -d $O/pmc? | tr '\n' ' ') > $O/pipe_util.txt 2>&1
rm -rf $O/stats $O/pmc?
cat $O/pipe_util.txt
