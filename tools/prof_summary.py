#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats CSV directory: per-kernel calls / avg / share,
and one step's timeline (durations + gaps) from the middle of the trace."""
import csv
import glob
import sys

d = sys.argv[1]
stats = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
trace = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
print("# kernel stats:", stats)
for r in csv.DictReader(open(stats)):
  print(f"{r['Name'][:86]:86s} calls={int(r['Calls']):6d} avg={float(r['AverageNs']) / 1e3:8.2f}us  {float(r['Percentage']):6.2f}%")
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
starts = [i for i, n in enumerate(names) if "adam_update" in n]   # last kernel of a step
if len(starts) > 12:
  # a step of the TIMED region: the most common kernel count per step (the later timing passes of bench.py
  # repeat the loss kernel inside an event pair and would show up as longer steps)
  from collections import Counter
  counts = Counter(starts[i + 1] - starts[i] for i in range(len(starts) - 1))
  modal = counts.most_common(1)[0][0]
  cand = [i for i in range(len(starts) - 1) if starts[i + 1] - starts[i] == modal]
  pick = cand[len(cand) // 2]
  a, b = starts[pick] + 1, starts[pick + 1] + 1
  print(f"# one step: {b - a} kernels" + (" (the modal step of the trace; with a background optimiser sweep in the trace -- adam_sweep_kernel -- the timed steps' own timeline is tools/sweep_timeline.py's)" if any("adam_sweep" in n for n in names) else ""))
  prev = None
  t0 = int(rows[a]["Start_Timestamp"])
  for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"  {r['Kernel_Name'][:70]:70s} t={(s - t0) / 1e3:7.2f} dur={(e - s) / 1e3:6.2f}us gap={gap:5.2f} grid={r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']} vgpr={r['VGPR_Count']}")
    prev = e
  print(f"  step wall = {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.2f}us")
