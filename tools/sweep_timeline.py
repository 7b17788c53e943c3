#!/usr/bin/env python3
"""Timeline around one background optimiser sweep (adam_sweep_kernel, smx_step.hip: head_sweep_start) from a rocprofv3
--kernel-trace CSV directory: every kernel that overlaps the window between two consecutive output-head launches, with its
queue -- what the main stream's launches cost while the sweep runs beside them."""
import csv
import glob
import sys

d = sys.argv[1]
trace = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
sw = [i for i, r in enumerate(rows) if "adam_sweep" in r["Kernel_Name"]]
if len(sw) < 8:
  print("no sweeps in this trace")
  sys.exit(0)
i = sw[len(sw) // 2]
# the output head before this sweep and the one after it
a = max(j for j in range(i) if "head_fused" in rows[j]["Kernel_Name"])
b = min(j for j in range(i + 1, len(rows)) if "head_fused" in rows[j]["Kernel_Name"])
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b + 1]:
  s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
  print(f"  q={r.get('Queue_Id', '?'):>3s} {r['Kernel_Name'][:64]:64s} t={(s - t0) / 1e3:7.2f} .. {(e - t0) / 1e3:7.2f}  dur={(e - s) / 1e3:6.2f}us  grid={r['Grid_Size_X']}")
print(f"  head to head = {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.2f}us")
# gaps between back-to-back launches of the output head (bench.py's kernel_times repeats it inside one event pair)
g = [(int(rows[j + 1]["Start_Timestamp"]) - int(rows[j]["End_Timestamp"])) / 1e3 for j in range(len(rows) - 1)
     if "head_fused" in rows[j]["Kernel_Name"] and "head_fused" in rows[j + 1]["Kernel_Name"]]
if g:
  g.sort()
  print(f"  back-to-back output heads: {len(g)} gaps, median {g[len(g) // 2]:.2f}us, min {g[0]:.2f}us")
for name in ("adam_update", "bigk_kernel<0, 1, 1>", "wgrad_panel"):
  g = [(int(rows[j + 1]["Start_Timestamp"]) - int(rows[j]["End_Timestamp"])) / 1e3 for j in range(len(rows) - 1) if name in rows[j]["Kernel_Name"]]
  if g:
    g.sort()
    print(f"  gap after {name}: median {g[len(g) // 2]:.2f}us, min {g[0]:.2f}us")
