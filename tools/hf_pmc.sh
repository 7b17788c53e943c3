#!/bin/bash
# development: SQ counters of the one-launch output head by itself (tools/headfused_try.py at 128 x 20 000 zinb), one --pmc pass per pair,
# Prints per-launch means of every counter (round 4's kernel beside it: profiles/r05_head_fused_experiments.txt).
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/hf_pmc
rm -rf $O; mkdir -p $O
for V in v2; do
  i=0
  for C in "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_IFETCH SQ_WAIT_ANY"; do
    i=$((i + 1))
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$V$i -- python3 $R/tools/headfused_try.py --time-only ${1:-zinb} --reps 10 > /dev/null 2> $O/$V$i.err
  done
done
python3 - $O <<'PY' | tee $O/summary.txt
import csv, glob, sys, collections
o = sys.argv[1]
for v in ("v2",):
  acc = collections.defaultdict(list)
  for p in glob.glob(f"{o}/{v}*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
      if "head_fused" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
  print(v, {k: round(sum(x) / len(x)) for k, x in sorted(acc.items())})
PY
