export TMPDIR=/tmp
O=gpurun_out/spmc; rm -rf $O; mkdir -p $O
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  DRAWS=100 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/p$i -- python3 tools/score_pmc.py > /dev/null 2> $O/p$i.err
done
python3 - $O <<'PY'
import csv, glob, sys, collections
o = sys.argv[1]
acc = collections.defaultdict(list)
for p in glob.glob(f"{o}/p*/**/*counter_collection.csv", recursive=True):
  for r in csv.DictReader(open(p)):
    if "score_head_kernel" in r["Kernel_Name"]:
      acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
  print(f"{k:28s} mean per launch {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
