#!/usr/bin/env python3
"""One training step of a rocprofv3 --kernel-trace CSV with EVERY column that could explain a dispatch gap (queue, stream,
LDS / scratch sizes, workgroup shape): `python tools/gap_probe.py <trace dir>`."""
import csv, glob, sys
from collections import Counter
trace = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
print("# columns:", list(rows[0].keys()))
ends = [i for i, r in enumerate(rows) if "adam_update" in r["Kernel_Name"]]
counts = Counter(ends[i + 1] - ends[i] for i in range(len(ends) - 1))
modal = counts.most_common(1)[0][0]
cand = [i for i in range(len(ends) - 1) if ends[i + 1] - ends[i] == modal]
for pick in (cand[len(cand) // 4], cand[len(cand) // 2]):
  a, b = ends[pick] + 1, ends[pick + 1] + 1
  prev = None
  print(f"# step at row {a}")
  for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    keys = ("Queue_Id", "Stream_Id", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Workgroup_Size_X", "Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z", "Dispatch_Id")
    print(f"  {r['Kernel_Name'][:48]:48s} dur={(e - s) / 1e3:6.2f} gap={gap:5.2f} " + " ".join(f"{k.split('_')[0][:5]}{k.split('_')[1][:1] if '_' in k else ''}={r.get(k, '?')}" for k in keys))
    prev = e
