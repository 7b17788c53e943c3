#!/bin/bash
# the host's side of one marginal_log_prob call: HIP API calls (start, duration) between two score_walk launches, beside the kernels
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sc_ht
CALLS=6 DRAWS=100 timeout -k 10 300 rocprofv3 --hip-runtime-trace --kernel-trace --output-format csv -d /tmp/sc_ht -- python3 $GRAFT_REPO_ROOT/tools/score_pmc.py > $GRAFT_REPO_ROOT/gpurun_out/score_hiptrace.log 2>&1
python3 - >> $GRAFT_REPO_ROOT/gpurun_out/score_hiptrace.log 2>&1 <<'PY'
import csv, glob
k = sorted(csv.DictReader(open(glob.glob("/tmp/sc_ht/**/*kernel_trace.csv", recursive=True)[0])), key=lambda r: int(r["Start_Timestamp"]))
h = sorted(csv.DictReader(open(glob.glob("/tmp/sc_ht/**/*hip_api_trace.csv", recursive=True)[0])), key=lambda r: int(r["Start_Timestamp"]))
w = [r for r in k if "score_walk" in r["Kernel_Name"]]
a, b = int(w[-3]["End_Timestamp"]), int(w[-2]["End_Timestamp"])
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:60]) for r in k] + [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "H " + r["Function"]) for r in h]
for s, e, n in sorted(ev):
  if a - 2000 <= s <= b + 60000:
    print(f"{(s - a) / 1e3:9.2f} us  +{(e - s) / 1e3:8.2f}  {n}")
PY
tail -80 $GRAFT_REPO_ROOT/gpurun_out/score_hiptrace.log
