#!/bin/bash
# the fixed cost of a smx_train_steps call (tools/dev/train_tail.py) and the HIP calls / copies behind the last step of a traced call
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 python3 $GRAFT_REPO_ROOT/tools/dev/train_tail.py > $GRAFT_REPO_ROOT/gpurun_out/train_tail.log 2>&1
rm -rf /tmp/tt_ht
KS="20" REPS=4 timeout -k 10 200 rocprofv3 --hip-runtime-trace --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tt_ht -- python3 $GRAFT_REPO_ROOT/tools/dev/train_tail.py > /dev/null 2>&1
python3 - >> $GRAFT_REPO_ROOT/gpurun_out/train_tail.log 2>&1 <<'PY'
import csv, glob
k = sorted(csv.DictReader(open(glob.glob("/tmp/tt_ht/**/*kernel_trace.csv", recursive=True)[0])), key=lambda r: int(r["Start_Timestamp"]))
h = sorted(csv.DictReader(open(glob.glob("/tmp/tt_ht/**/*hip_api_trace.csv", recursive=True)[0])), key=lambda r: int(r["Start_Timestamp"]))
mc = glob.glob("/tmp/tt_ht/**/*memory_copy_trace.csv", recursive=True)
m = sorted(csv.DictReader(open(mc[0])), key=lambda r: int(r["Start_Timestamp"])) if mc else []
ad = [r for r in k if "adam_update" in r["Kernel_Name"]]
# the last adam launch that is followed by API calls (the call before the last one)
a = int(ad[-21]["Start_Timestamp"])
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:50]) for r in k] + [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "H " + r["Function"]) for r in h] + [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "M " + str(r.get("Direction", "copy")) + " " + str(r.get("Bytes", ""))) for r in m]
print("around the last step's optimiser launch of the call before the last:")
n_print = 0
for s, e, n in sorted(ev):
  if s >= a - 1000 and not n.startswith("H hipGetLastError") and n_print < 40:
    print(f"{(s - a) / 1e3:9.2f} us  +{(e - s) / 1e3:8.2f}  {n}")
    n_print += 1
PY
tail -50 $GRAFT_REPO_ROOT/gpurun_out/train_tail.log
