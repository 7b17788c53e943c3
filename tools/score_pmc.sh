#!/bin/bash
# SQ counters of the scoring head by itself (tools/score_pmc.py: marginal_log_prob of 128 cells x DRAWS posterior draws at the 8kly shape), one --pmc pass per
# counter pair, kernel-trace only.  Prints per-launch means of every counter for the walk (score_walk_kernel) and, with SMX_TUNING=score_walk=0, for the
# tile-per-workgroup form; profiles/r05_scoring_pmc.txt is written from this output.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/score_pmc
rm -rf $O; mkdir -p $O
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
  i=$((i + 1))
  DRAWS=${DRAWS:-100} CALLS=4 timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/p$i -- python3 $R/tools/score_pmc.py > /dev/null 2> $O/p$i.err || exit 1
done
python3 - $O <<'PY' | tee $O/summary.txt
import csv, glob, sys, collections
o = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for p in glob.glob(f"{o}/p*/**/*counter_collection.csv", recursive=True):
  for r in csv.DictReader(open(p)):
    for fam in ("score_walk_kernel", "score_head_kernel"):
      if fam in r["Kernel_Name"]:
        acc[fam][r["Counter_Name"]].append(float(r["Counter_Value"]))
for p in glob.glob(f"{o}/p*/**/*kernel_trace.csv", recursive=True):
  for r in csv.DictReader(open(p)):
    for fam in ("score_walk_kernel", "score_head_kernel"):
      if fam in r["Kernel_Name"]:
        dur[fam].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for fam in acc:
  print(fam, "launches per pass", len(next(iter(acc[fam].values()))), "duration under the counters (us): mean %.1f min %.1f" % (sum(dur[fam]) / len(dur[fam]), min(dur[fam])))
  for k, x in sorted(acc[fam].items()):
    print(f"  {k:28s} {sum(x) / len(x):14.0f}")
PY
